"""Micro-benchmarks of single kernels at the MobileNetLarge3D (batch 32) layer shapes.
    python tools/kbench.py [filter]
Prints us/launch and algorithmic GB/s (HIP events, median of several launches)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picklebot_amd import kernels as K  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


_FLUSH = None


def timeit(fn, iters=8, warm=2):
    """Median time of one launch right after a 512 MB write (keeps the clocks up and evicts the operands
    from L2 / Infinity Cache, as in a training step where they were produced many kernels earlier)."""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    for _ in range(warm):
        fn()
    ts = []
    for i in range(iters):
        if not os.environ.get("PB_KB_NOFLUSH"):
            _FLUSH.fill_(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def report(name, us, nbytes):
    print("%-70s %9.1f us %8.1f GB/s" % (name, us, nbytes / 1e3 / us), flush=True)


def pw(M, Kd, N, trans=False, rs=False, stats=False, G=32):
    x = torch.randn(M, Kd, device=DEV).to(BF)
    w = torch.randn((Kd, N) if trans else (N, Kd), device=DEV)
    row_scale = torch.rand(G, Kd, device=DEV) if rs else None
    st = torch.zeros(K.REPL, 2, N, device=DEV) if stats else None
    fn = lambda: K.pw_fwd(x, w, w_transposed=trans, row_scale=row_scale, rows_per_group=M // G if rs else 0, stats=st)
    report(f"pw_fwd M{M} K{Kd} N{N} trans={int(trans)} rs={int(rs)} stats={int(stats)}", timeit(fn), (M * Kd + M * N) * 2 + N * Kd * 4)


def wg(M, Kd, N, rs=False, G=32):
    dy = torch.randn(M, N, device=DEV).to(BF)
    x = torch.randn(M, Kd, device=DEV).to(BF)
    dw = torch.zeros(N, Kd, device=DEV)
    row_scale = torch.rand(G, Kd, device=DEV) if rs else None
    fn = lambda: K.pw_bwd_weight(dy, x, dw, row_scale=row_scale, rows_per_group=M // G if rs else 0)
    report(f"pw_wgrad M{M} K{Kd} N{N} rs={int(rs)}", timeit(fn), (M * Kd + M * N) * 2 + N * Kd * 4)


def dw(C, thw, k, s, N=32, what=("fwd", "bwd_data", "bwd_weight"), se=False):
    kernel, stride, pad = (1, k, k), (s, s, s), (k // 2,) * 3
    x = torch.randn(N, *thw, C, device=DEV).to(BF)
    w = torch.randn(C, k * k, device=DEV)
    pool = torch.zeros(N, C, device=DEV) if se else None
    y = K.dw_fwd(x, w, kernel, stride, pad, pool=pool)
    gs = torch.rand(N, C, device=DEV) if se else None
    gh = torch.rand(N, C, device=DEV) if se else None
    tag = f"C{C} {thw}->{tuple(y.shape[1:4])} k{k} s{s} se={int(se)}"
    nb = (x.numel() + y.numel()) * 2
    if "fwd" in what:
        report("dw_fwd " + tag, timeit(lambda: K.dw_fwd(x, w, kernel, stride, pad, pool=pool)), nb)
    if "bwd_data" in what:
        report("dw_bwd_data " + tag, timeit(lambda: K.dw_bwd_data(y, w, thw, kernel, stride, pad, gs, gh)), nb)
    if "bwd_weight" in what:
        dwt = torch.zeros_like(w)
        report("dw_bwd_weight " + tag, timeit(lambda: K.dw_bwd_weight(y, x, dwt, kernel, stride, pad, gs, gh)), nb)


def ew(M, C, G=32):
    x = torch.randn(M, C, device=DEV).to(BF)
    z = torch.randn(M, C, device=DEV).to(BF)
    sc, sh = torch.rand(C, device=DEV), torch.rand(C, device=DEV)
    rs = torch.rand(G, C, device=DEV)
    sums = torch.zeros(K.REPL, 2, C, device=DEV)
    report(f"affine_act_fwd M{M} C{C}", timeit(lambda: K.affine_act_fwd(x, sc, sh, "hardswish", 0.0, rs, M // G)), 2 * M * C * 2)
    report(f"act_bwd_reduce M{M} C{C}", timeit(lambda: K.act_bwd_reduce(x, z, scale=sc, shift=sh, act="hardswish", row_scale=rs,
           rows_per_group=M // G, mean=sh, rstd=sc, sums=sums)), 3 * M * C * 2)
    report(f"affine2 M{M} C{C}", timeit(lambda: K.affine2(x, z, sc, sh, sh)), 3 * M * C * 2)


CASES = {
    "pw_big": lambda: [pw(112896, 672, 112), pw(112896, 672, 112, rs=True), pw(112896, 672, 112, rs=True, stats=True),
                       pw(112896, 112, 672), pw(112896, 112, 672, trans=True), pw(112896, 672, 112, trans=True),
                       pw(29792, 960, 160, rs=True, stats=True), pw(29792, 160, 960), wg(112896, 672, 112, rs=True),
                       wg(112896, 112, 672), wg(29792, 960, 160, rs=True)],
    "wg_only": lambda: [wg(112896, 672, 112, rs=True), wg(112896, 112, 672), wg(29792, 960, 160, rs=True), wg(100352, 480, 112),
                        wg(351232, 40, 240), wg(250880, 120, 40)],
    "wg_small": lambda: [wg(8192, 192, 768), wg(8192, 128, 192), wg(2048, 960, 240), wg(25088, 960, 160, rs=True),
                         wg(25088, 160, 960), wg(100352, 112, 480)],
    "wg_floor": lambda: [wg(256, 128, 192), wg(1024, 128, 192), wg(8192, 128, 192), wg(256, 192, 768), wg(8192, 192, 768),
                         wg(65536, 192, 768)],
    "pw_small": lambda: [pw(4014080, 16, 64), pw(4014080, 16, 16, stats=True), pw(4014080, 64, 16, trans=True),
                         pw(602112, 64, 24, stats=True), pw(802816, 24, 72), pw(351232, 40, 240), wg(4014080, 16, 64),
                         wg(4014080, 16, 16), wg(802816, 24, 72)],
    "dw": lambda: [dw(16, (8, 112, 112), 3, 1), dw(64, (10, 112, 112), 3, 2), dw(72, (6, 56, 56), 3, 1),
                   dw(72, (8, 56, 56), 5, 2, se=True), dw(120, (10, 28, 28), 5, 1, se=True), dw(240, (8, 14, 14), 3, 1),
                   dw(184, (12, 14, 14), 3, 1), dw(672, (16, 14, 14), 3, 1, se=True), dw(672, (18, 14, 14), 5, 2, se=True),
                   dw(960, (15, 7, 7), 5, 1, se=True)],
    "dw_small": lambda: [dw(960, (15, 7, 7), 5, 1, se=True), dw(120, (10, 28, 28), 5, 1, se=True),
                         dw(672, (16, 14, 14), 3, 1, se=True)],
    "ew_small": lambda: [ew(17248, 160), ew(50176, 80), ew(100352, 112), ew(150528, 40)],
    "ew": lambda: [ew(4014080, 16), ew(602112, 24), ew(351232, 40), ew(112896, 112), ew(29792, 160), ew(29792, 960)],
}

if __name__ == "__main__":
    flt = sys.argv[1:] or list(CASES)
    for name in flt:
        print("==", name)
        CASES[name]()
