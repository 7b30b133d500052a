"""Worst error / allowed error of the forward keys of every bf16-autocast model golden case (how much margin the
bf16 parity test has): python tools/margin_check.py [runs]"""
import contextlib
import io
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import picklebot_amd
from oracle import protocol
from conftest import check_model_result, load_golden  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
only = sys.argv[2] if len(sys.argv) > 2 else None
for case in protocol.MODEL_CASES:
    if only and case != only:
        continue
    gold = load_golden("model_" + case)
    for it in range(runs):
        with contextlib.redirect_stdout(io.StringIO()):
            res = protocol.run_model_case(picklebot_amd, case, device="cuda:0", dtype=torch.bfloat16, autocast=True)
        fwd = {k: v for k, v in res.items() if k.split("/")[0] not in ("gnorm", "gsamp", "gproj", "blockstat")}
        bad, worst = check_model_result(fwd, gold, 1e-2, 1e-2, dev="refbf16_err", factor=2.0)
        print(f"{case} run {it}: {len(bad)} forward keys off; worst {worst[0]} err {worst[1]:.5f} allowed {worst[2]:.5f} ratio {worst[1] / worst[2]:.2f}", flush=True)
