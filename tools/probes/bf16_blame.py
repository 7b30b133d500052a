"""Where does the bf16 gradient deviation of MobileViT come from?  Re-runs the eval-mode gradient fixture under bf16 autocast
with single operators swapped for their fp32 kernels (inputs widened, result rounded back to bf16) and prints, per stage
of the network, the mean ratio of our per-tensor projection error to the reference's own bf16 deviation.
    python tools/probes/bf16_blame.py [case]"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import picklebot_amd  # noqa: E402
from picklebot_amd import kernels as K  # noqa: E402
from conftest import check_model_result, golden_mode, load_golden  # noqa: E402
from oracle import protocol  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "mobilevit_s"
gold = golden_mode(load_golden("grad_" + case), "evalgrad")
bf = torch.bfloat16


def widen(fn, outs_bf16=True):
    def wrapped(*args, **kw):
        was = [isinstance(a, torch.Tensor) and a.dtype == bf for a in args]
        args = [a.float() if w else a for a, w in zip(args, was)]
        res = fn(*args, **kw)
        if not any(was):
            return res
        if isinstance(res, tuple):
            return tuple(r.to(bf) if isinstance(r, torch.Tensor) and r.dtype == torch.float32 and r.shape[-1] != 8 else r for r in res)
        return res.to(bf)
    return wrapped


def run(tag, patches):
    saved = {name: getattr(K, name) for name in patches}
    for name in patches:
        setattr(K, name, widen(saved[name]))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            res = protocol.run_grad_case(picklebot_amd, case, "evalgrad", device="cuda:0", autocast=True)
    finally:
        for name, fn in saved.items():
            setattr(K, name, fn)
    groups = {}
    for k in res:
        if not k.startswith("gproj/"):
            continue
        ref = gold.get("refbf16_mean/grad/" + k[6:])
        if ref is None or float(ref[0]) < 2e-3:
            continue
        bad, _ = check_model_result({k: res[k]}, gold, 0.0, 0.0, dev="ref32_err", factor=0.0, samp_factor=1.0)
        e = bad[0][1] if bad else 0.0
        name = k[6:]
        grp = name.split(".")[0] if not name.startswith("trunk") else ".".join(name.split(".")[:2])
        groups.setdefault(grp, []).append(e / float(ref[0]))
    print("%-34s " % tag + "  ".join("%s %.2f" % (g, np.mean(v)) for g, v in groups.items()), flush=True)


def fp32_token_stream():
    """Transformer.forward with the token residual stream kept in fp32 (what autocast does in the reference: LayerNorm and
    the residual adds run in fp32, only the Linear inputs are bf16)."""
    from picklebot_amd import mobilevit as MV
    orig = MV.Transformer.forward

    def forward(self, x):
        b, p, n, d = x.shape
        s32 = x.reshape(-1, d).float()
        for attn, ff in self.layers:
            s32 = s32 + attn.rows(s32.to(bf), b * p, n, residual=False).float()
            s32 = s32 + ff.rows(s32.to(bf), residual=False).float()
        return s32.to(bf).view(b, p, n, d)
    MV.Transformer.forward = forward
    return lambda: setattr(MV.Transformer, "forward", orig)


run("all bf16", [])
undo = fp32_token_stream()
run("fp32 token residual stream", [])
undo()
run("attention fwd+bwd fp32", ["attention_fwd", "attention_bwd"])
run("attention bwd fp32", ["attention_bwd"])
run("layernorm fp32", ["layernorm_fwd", "layernorm_bwd"])
