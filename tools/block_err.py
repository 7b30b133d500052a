"""Debug aid: run one block case in bf16 autocast on the GPU and print per-tensor relative errors vs golden."""
import contextlib, io, os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import picklebot_amd
from oracle import protocol
case = sys.argv[1]
gold = dict(np.load(os.path.join(root, "tests", "golden", "blocks.npz")))
with contextlib.redirect_stdout(io.StringIO()):
    res = protocol.run_block_case(picklebot_amd, case, device="cuda", autocast=True)
for k, v in res.items():
    g = gold.get(case + "/" + k)
    if g is None:
        continue
    v = np.asarray(v, np.float64); g = np.asarray(g, np.float64)
    print("%-50s rel err %.4g  nan=%d" % (k, np.linalg.norm(v - g) / max(np.linalg.norm(g), 1e-30), int(np.isnan(v).sum())))
