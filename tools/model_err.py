"""Print the largest error/allowed ratios of one model case on the GPU (fp32 protocol) - debugging aid."""
import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import picklebot_amd
from oracle import protocol
from conftest import check_model_result

case = sys.argv[1] if len(sys.argv) > 1 else "movinet_a2"
gold = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model_%s.npz" % case)))
for rep in range(2):
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_model_case(picklebot_amd, case, device="cuda")
    rows = []
    for k in res:
        bad, worst = check_model_result({k: res[k]}, gold, 0.0, 0.0, dev="ref32_err", factor=0.0)
        if bad:
            rows.append((bad[0][1], k))
    rows.sort(reverse=True)
    print("run", rep, "top errors:", [(round(e, 6), k) for e, k in rows[:8]])
    g = [e for e, k in rows if k.startswith("gproj/")]
    print("   gproj median %.2e  p90 %.2e  max %.2e" % (np.median(g), np.percentile(g, 90), max(g)))
