"""Randomised sweep of pb_pwconv_fwd_bf16w (all three kernels) against the CPU mirror: shapes, groups, statistics, epilogues."""
import os, sys, random
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_err
from oracle import kernel_ref as R
from picklebot_amd import kernels as K
DEV = "cuda:0"; bf = torch.bfloat16
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
for it in range(n_cases):
    Kd = 8 * rng.choice([1, 2, 3, 5, 8, 9, 14, 15, 20, 21, 23, 25, 30, 42, 60, 84, 96, 120, 128])
    N = 8 * rng.choice([1, 2, 3, 5, 9, 10, 12, 14, 15, 20, 23, 30, 60, 84, 120])
    G = rng.choice([0, 0, 0, 2, 3, 7])
    rows = rng.choice([65, 127, 300, 1031, 2048, 2500, 4099, 9001])
    M = rows * (G if G else rng.choice([1, 2, 5]))
    if M * max(Kd, N) > 40e6:
        continue
    x = (torch.randn(M, Kd) * 1.0).to(bf); w = torch.randn(N, Kd) * Kd ** -0.5
    kw = dict(row_scale=torch.rand(G, Kd) + 0.5, rows_per_group=M // G) if G else {}
    dkw = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
    mode = rng.choice(["plain", "stats", "epi", "res"])
    tag = f"M{M} K{Kd} N{N} G{G} {mode}"
    try:
        if mode == "stats":
            st_ref = torch.zeros(R.REPL, 2, N); want = R.pw_fwd(x, w, stats=st_ref, **kw)
            st = torch.zeros(K.REPL, 2, N, device=DEV); got = K.pw_fwd(x.to(DEV), w.to(DEV), stats=st, **dkw)
            e2 = rel_err(st.sum(0).cpu().numpy(), st_ref.sum(0).numpy())
        elif mode == "epi":
            sc, sh = torch.randn(N) * 0.2 + 1, torch.randn(N) * 0.1
            act = rng.choice(["relu", "hardswish", "silu", "none"])
            want = R.pw_fwd(x, w, out_scale=sc, out_shift=sh, out_act=act, **kw)
            got = K.pw_fwd(x.to(DEV), w.to(DEV), out_scale=sc.to(DEV), out_shift=sh.to(DEV), out_act=act, **dkw); e2 = 0
        elif mode == "res" and not G:
            r = torch.randn(M, N).to(bf); want = R.pw_fwd(x, w, residual=r); got = K.pw_fwd(x.to(DEV), w.to(DEV), residual=r.to(DEV)); e2 = 0
        else:
            want = R.pw_fwd(x, w, **kw); got = K.pw_fwd(x.to(DEV), w.to(DEV), **dkw); e2 = 0
        torch.cuda.synchronize()
        e = rel_err(got.float().cpu().numpy(), want.float().numpy())
        fam = K._bf16w_family(M, Kd, N, G if G else 1, mode == "stats") if M > 64 else "-"
        ok = e <= 1.2e-2 and e2 <= 1.2e-2
        if not ok: bad += 1
        print(("ok  " if ok else "BAD ") + f"{tag:40s} {fam:24s} err {e:.2e} stats {e2:.2e}", flush=True)
    except Exception as ex:
        bad += 1; print("EXC " + tag, type(ex).__name__, str(ex)[:200], flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
