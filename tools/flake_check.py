"""Run one model golden case several times in one process and print how many keys exceed the fp32 bound each time
(diagnosis of run-to-run variation: python tools/flake_check.py movinet_a2 6)."""
import contextlib
import io
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import picklebot_amd
from oracle import protocol
from conftest import check_model_result  # noqa: E402

case, n = sys.argv[1], int(sys.argv[2])
from conftest import load_golden
gold = load_golden("model_" + case)
for it in range(n):
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_model_case(picklebot_amd, case, device="cuda:0")
    bad, worst = check_model_result(res, gold, 1e-3, 1e-3, dev="ref32_err", factor=4.0)
    fwd_bad = [b for b in bad if b[0].split("/")[0] not in ("gnorm", "gsamp", "gproj")]
    print(f"run {it}: {len(bad)} keys off ({len(fwd_bad)} forward), worst {worst[0]} {worst[1]:.3e}", flush=True)
