"""Tile plans of the fused expand+depthwise kernel for the MobileNetLarge3D bottlenecks (no GPU needed)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picklebot_amd import _C, kernels as K
L = [(16, 16, (8, 112, 112), 3, 1), (16, 64, (10, 112, 112), 3, 2), (24, 72, (6, 56, 56), 3, 1), (24, 72, (8, 56, 56), 5, 2),
     (40, 120, (6, 28, 28), 5, 1), (40, 120, (10, 28, 28), 5, 1), (40, 240, (14, 28, 28), 3, 2), (80, 240, (8, 14, 14), 3, 1),
     (80, 184, (10, 14, 14), 3, 1), (80, 184, (12, 14, 14), 3, 1), (80, 480, (14, 14, 14), 3, 1), (112, 672, (16, 14, 14), 3, 1),
     (112, 672, (18, 14, 14), 5, 2), (160, 960, (11, 7, 7), 5, 1), (160, 960, (15, 7, 7), 5, 1)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lib = _C.load()
for Cin, E, thw, k, s in L:
    kernel, stride, pad = (1, k, k), (s, s, s), (k // 2,) * 3
    out = K.dw_out_shape(thw, kernel, stride, pad)
    sh = K._dw_shape(N, E, thw, out, kernel, stride, pad)
    for wg in (0, 1):
        v = (ctypes.c_int * 12)()
        ok = lib.pb_xdw_plan(ctypes.byref(sh), Cin, wg, v)
        names = "TH TW EC nchunk esplit lds gridx nseg nbands band_rows HP KS".split()
        print(f"Cin{Cin:4d} E{E:4d} {thw}->{out} k{k} s{s} {'wgrad' if wg else 'fwd  '}:", "NOT TAKEN" if not ok else " ".join(f"{n}={x}" for n, x in zip(names, v)))
