"""Actual error of the HIP path against the fp64 goldens on the gradient fixtures (tests/golden/grad_*.npz), per model and
mode, with NOTHING widened: distribution of the per-key relative errors (norm, projections) and the worst keys.
    python tools/grad_err.py [case ...] [--bf16]"""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import picklebot_amd  # noqa: E402
from conftest import check_model_result, golden_mode, load_golden  # noqa: E402
from oracle import protocol  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
bf16 = "--bf16" in sys.argv
for case in (args or list(protocol.GRAD_CASES)):
    gold_all = load_golden("grad_" + case)
    for mode in ("evalgrad", "train"):
        gold = golden_mode(gold_all, mode)
        with contextlib.redirect_stdout(io.StringIO()):
            res = protocol.run_grad_case(picklebot_amd, case, mode, device="cuda:0", autocast=bf16)
        rows = []
        for k in res:
            bad, worst = check_model_result({k: res[k]}, gold, 0.0, 0.0, dev="ref32_err", factor=0.0, samp_factor=1.0)
            rows.append((bad[0][1] if bad else 0.0, k))
        rows.sort(reverse=True)
        for kind in ("gnorm", "gproj", "gsamp"):
            g = np.array([e for e, k in rows if k.startswith(kind + "/")])
            print(f"[{case}/{mode}{' bf16' if bf16 else ''}] {kind}: median {np.median(g):.2e} p90 {np.percentile(g, 90):.2e} max {g.max():.2e}")
        if bf16:      # ratio of our per-tensor projection error to the reference's own mean bf16-autocast deviation
            rr = []
            for e, k in rows:
                if k.startswith("gproj/"):
                    ref = gold.get("refbf16_mean/grad/" + k[6:])
                    if ref is not None and float(ref[0]) > 2e-3:
                        rr.append((e / float(ref[0]), k))
            if rr:
                r = np.array([v for v, _ in rr])
                print(f"[{case}/{mode} bf16] ours / reference bf16 deviation over {len(r)} tensors: median {np.median(r):.2f} p90 {np.percentile(r, 90):.2f} "
                      f"p99 {np.percentile(r, 99):.2f} max {r.max():.2f} ({max(rr)[1]})")
            if "--by-depth" in sys.argv:      # in parameter order (input side first): our error, the reference's, the ratio
                order = [k for k in res if k.startswith("gproj/")]
                d = {k: e for e, k in rows}
                for k in order:
                    ref = gold.get("refbf16_mean/grad/" + k[6:])
                    if ref is not None:
                        print("      %-58s ours %.2e  ref %.2e  ratio %5.2f" % (k[6:], d[k], float(ref[0]), d[k] / max(float(ref[0]), 1e-12)))
        fwd = [(e, k) for e, k in rows if k.split("/")[0] not in ("gnorm", "gproj", "gsamp")]
        print(f"   forward keys: {[(k, '%.1e' % e) for e, k in fwd]}")
        print(f"   worst: {[(k, '%.1e' % e) for e, k in rows[:6]]}", flush=True)
