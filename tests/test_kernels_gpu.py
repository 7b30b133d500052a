"""GPU: every HIP kernel (through the C ABI) against its CPU mirror (oracle/kernel_ref.py) on the
same seeded inputs.  Tolerances: fp32 1e-4 relative per tensor (accumulation order differs),
bf16 1e-2 (outputs rounded to bf16).  Edge cases: ragged M, channel counts that are
not multiples of the vector/MFMA tile, N=13 class heads, zero frames of the temporal growth quirk,
stride 2, k=5, prologue/epilogue combinations, broadcast gradients."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import kernel_ref as R

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = {torch.float32: 2e-4, torch.bfloat16: 1.2e-2}


@pytest.fixture(scope="module")
def K():
    from picklebot_amd import kernels
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return kernels


def rnd(shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 9973)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def dev(t):
    return None if t is None else t.to(DEV)


def check(name, got, want, dtype, tol_scale=1.0):
    e = rel_err(got.detach().float().cpu().numpy(), want.detach().float().numpy())
    assert e <= TOL[dtype] * tol_scale, f"{name}: rel err {e:.3e} > {TOL[dtype] * tol_scale:.1e}"


DTYPES = [torch.float32, torch.bfloat16]

PW_CASES = [  # M, K, N
    (300, 16, 16), (1000, 16, 64), (257, 64, 24), (129, 24, 72), (640, 72, 40), (513, 120, 40), (200, 240, 80),
    (130, 184, 80), (260, 672, 160), (140, 960, 160), (150, 160, 960), (64, 960, 1280), (32, 1280, 13), (5, 13, 1280),
    (1, 8, 8), (4096, 40, 120),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Kd,N", PW_CASES)
def test_pw_fwd_plain_and_transposed(K, M, Kd, N, dtype):
    x, w = rnd((M, Kd), dtype, 1), rnd((N, Kd), seed=2, scale=Kd ** -0.5)
    check("pw_fwd", K.pw_fwd(dev(x), dev(w)), R.pw_fwd(x, w), dtype)
    wt = w.t().contiguous()
    check("pw_fwd_T", K.pw_fwd(dev(x), dev(wt), w_transposed=True), R.pw_fwd(x, wt, w_transposed=True), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Kd,N,G", [(6 * 49, 72, 40, 6), (4 * 100, 16, 16, 4), (2 * 333, 120, 48, 2), (3 * 64, 960, 160, 3)])
def test_pw_fwd_prologue_epilogue_stats(K, M, Kd, N, G, dtype):
    x, w = rnd((M, Kd), dtype, 3), rnd((N, Kd), seed=4, scale=Kd ** -0.5)
    kw = dict(in_scale=rnd((Kd,), seed=5) * 0.2 + 1, in_shift=rnd((Kd,), seed=6) * 0.1, in_act="hardswish",
              row_scale=torch.rand(G, Kd), rows_per_group=M // G, out_scale=rnd((N,), seed=7) * 0.2 + 1,
              out_shift=rnd((N,), seed=8) * 0.1, out_act="relu")
    st_ref = torch.zeros(R.REPL, 2, N)
    want = R.pw_fwd(x, w, stats=st_ref, **kw)
    st = torch.zeros(K.REPL, 2, N, device=DEV)
    got = K.pw_fwd(dev(x), dev(w), stats=st, **{k: (dev(v) if torch.is_tensor(v) else v) for k, v in kw.items()})
    check("pw_fwd_fused", got, want, dtype)
    check("pw_fwd_stats", st.sum(0), st_ref.sum(0), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Kd,N", [(300, 16, 16), (1000, 16, 64), (257, 64, 24), (513, 120, 40), (260, 672, 160),
                                    (150, 160, 960), (32, 1280, 13), (33, 13, 1280), (5000, 24, 72), (70000, 16, 16), (4100, 16, 16)])
def test_pw_bwd_weight(K, M, Kd, N, dtype):
    dy, x = rnd((M, N), dtype, 9), rnd((M, Kd), dtype, 10)
    G = 1
    rs = torch.rand(G, Kd)
    want = R.pw_bwd_weight(dy, x, torch.zeros(N, Kd), row_scale=rs, rows_per_group=M, in_act="hardswish")
    got = K.pw_bwd_weight(dev(dy), dev(x), torch.zeros(N, Kd, device=DEV), row_scale=dev(rs), rows_per_group=M,
                          in_act="hardswish")
    check("pw_bwd_weight", got, want, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Kd,N,G", [(4 * 196, 672, 112, 4), (3 * 130, 120, 40, 3), (2 * 3136, 72, 24, 2), (5 * 128, 960, 160, 5),
                                      (2 * 1000, 240, 80, 2), (6 * 200, 184, 80, 6)])
def test_pw_fwd_training_projection(K, M, Kd, N, G, dtype):
    """The projection conv of a training step: squeeze-excite scale alone on the input (bf16: folded into per-clip
    weights, LDS-DMA GEMM with clip-aligned row panels) + BatchNorm statistics of the output; and its
    backward-data twin (plain GEMM on the transposed weights)."""
    x, w = rnd((M, Kd), dtype, 31), rnd((N, Kd), seed=32, scale=Kd ** -0.5)
    rs = torch.rand(G, Kd) + 0.5
    st_ref = torch.zeros(R.REPL, 2, N)
    want = R.pw_fwd(x, w, row_scale=rs, rows_per_group=M // G, stats=st_ref)
    st = torch.zeros(K.REPL, 2, N, device=DEV)
    got = K.pw_fwd(dev(x), dev(w), row_scale=dev(rs), rows_per_group=M // G, stats=st)
    check("pw_fwd_se", got, want, dtype)
    check("pw_fwd_se_stats", st.sum(0), st_ref.sum(0), dtype)
    dy = rnd((M, N), dtype, 33)
    check("pw_fwd_dgrad", K.pw_fwd(dev(dy), dev(w), w_transposed=True), R.pw_fwd(dy, w, w_transposed=True), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Kd,N,G", [(48, 16, 384, 0), (32, 128, 16, 0), (96, 128, 128, 0), (40, 72, 24, 0), (200, 16, 384, 0),
                                      (150, 160, 960, 0), (5000, 24, 72, 4), (3000, 672, 112, 4), (7000, 112, 672, 7),
                                      (1024, 960, 160, 2), (777, 40, 120, 0)])
def test_pw_bwd_weight_training_paths(K, M, Kd, N, G, dtype):
    """What the training step actually launches: no affine prologue (LDS-DMA ring kernel in bf16) and the
    squeeze-excite scale alone (clip-aligned ranges, scale applied to the accumulators)."""
    dy, x = rnd((M, N), dtype, 19), rnd((M, Kd), dtype, 20)
    want = R.pw_bwd_weight(dy, x, torch.zeros(N, Kd))
    got = K.pw_bwd_weight(dev(dy), dev(x), torch.zeros(N, Kd, device=DEV))
    check("pw_bwd_weight_plain", got, want, dtype)
    if G:
        rs = torch.rand(G, Kd) + 0.5
        want = R.pw_bwd_weight(dy, x, torch.zeros(N, Kd), row_scale=rs, rows_per_group=M // G)
        got = K.pw_bwd_weight(dev(dy), dev(x), torch.zeros(N, Kd, device=DEV), row_scale=dev(rs), rows_per_group=M // G)
        check("pw_bwd_weight_se", got, want, dtype)


@pytest.mark.parametrize("M,Kd,N,G", [
    (70000, 24, 72, 0), (7 * 9431, 72, 24, 7), (20000, 40, 120, 0), (20001, 40, 240, 0), (9000, 80, 184, 0), (9000, 80, 200, 0),
    (5 * 1801, 80, 480, 5), (12000, 112, 672, 0), (6 * 2001, 672, 112, 6), (5000, 160, 960, 0), (4 * 1250, 960, 160, 4), (3000, 480, 112, 0),
    (130, 200, 80, 0), (40000, 64, 16, 0), (129, 24, 48, 0), (70000, 16, 112, 0), (3 * 3528, 672, 160, 3), (8 * 131, 120, 40, 8),
    (4100, 192, 112, 0), (2500, 768, 192, 0), (3001, 760, 128, 0), (2111, 184, 96, 0), (4000, 200, 80, 0), (2600, 480, 240, 0),
    (32 * 343, 672, 112, 32), (5 * 777, 240, 112, 5), (2 * 1031, 392, 120, 2), (9000, 480, 520, 0), (2048, 1032, 16, 0),
    (70000, 16, 16, 0), (40004, 16, 24, 0), (4100, 16, 8, 0)])
def test_pw_fwd_dma_ring_gemm(K, M, Kd, N, G):
    """The three bf16 GEMM kernels behind pb_pwconv_fwd_bf16w on their own - the weight-stationary streaming kernel
    (K <= 160), its long-reduction form (columns over the waves, K <= 1024, also with per-clip weights) and the LDS-DMA ring
    kernel (everything else): every column-tile width, several column tiles / groups per row panel, persistent workgroups
    walking many panels, ragged last panels, clip-aligned panels with per-clip weights, statistics (one slot per
    workgroup), affine + activation epilogue, residual; K = 16 with N <= 24: four rows per line (pb_pwconv_fwd_fold4), the
    statistics of the four column replicas folded in the workgroup.  Bit-identical reruns."""
    bf = torch.bfloat16
    x, w = rnd((M, Kd), bf, 41), rnd((N, Kd), seed=42, scale=Kd ** -0.5)
    kw = {}
    if G:
        kw = dict(row_scale=torch.rand(G, Kd) + 0.5, rows_per_group=M // G)
    dkw = {k: (dev(v) if torch.is_tensor(v) else v) for k, v in kw.items()}
    st_ref = torch.zeros(R.REPL, 2, N)
    want = R.pw_fwd(x, w, stats=st_ref, **kw)
    st = torch.zeros(K.REPL, 2, N, device=DEV)
    got = K.pw_fwd(dev(x), dev(w), stats=st, **dkw)
    check("dma_stats_y", got, want, bf)
    check("dma_stats", st.sum(0), st_ref.sum(0), bf)
    st2 = torch.zeros(K.REPL, 2, N, device=DEV)
    got2 = K.pw_fwd(dev(x), dev(w), stats=st2, **dkw)
    assert torch.equal(got, got2) and torch.equal(st, st2)
    plain = K.pw_fwd(dev(x), dev(w), **dkw)
    assert torch.equal(plain, got)
    ep = dict(out_scale=rnd((N,), seed=43) * 0.2 + 1, out_shift=rnd((N,), seed=44) * 0.1, out_act="hardswish")
    check("dma_epilogue", K.pw_fwd(dev(x), dev(w), **dkw, **{k: (dev(v) if torch.is_tensor(v) else v) for k, v in ep.items()}),
          R.pw_fwd(x, w, **kw, **ep), bf)
    check("dma_bias", K.pw_fwd(dev(x), dev(w), out_shift=dev(ep["out_shift"])), R.pw_fwd(x, w, out_shift=ep["out_shift"]), bf)
    if not G:
        r = rnd((M, N), bf, 45)
        check("dma_residual", K.pw_fwd(dev(x), dev(w), residual=dev(r)), R.pw_fwd(x, w, residual=r), bf)
        dy = rnd((M, N), bf, 46)
        check("dma_dgrad", K.pw_fwd(dev(dy), dev(w), w_transposed=True), R.pw_fwd(dy, w, w_transposed=True), bf)


@pytest.mark.parametrize("M,Kd,N,trans", [(1024, 192, 144, False), (777, 144, 288, True), (96, 768, 192, False), (32, 128, 64, False)])
def test_pw_fwd_residual(K, M, Kd, N, trans):
    """y = x W^T + r in the GEMM's store pass (MobileViT residual connections); also the copy fallback for M <= 64."""
    x = rnd((M, Kd), torch.bfloat16, 61)
    w = rnd((Kd, N) if trans else (N, Kd), seed=62, scale=Kd ** -0.5)
    r = rnd((M, N), torch.bfloat16, 63)
    want = R.pw_fwd(x, w, w_transposed=trans, residual=r)
    got = K.pw_fwd(dev(x), dev(w), w_transposed=trans, residual=dev(r))
    check("pw_fwd_residual", got, want, torch.bfloat16)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Kd,N,trans", [(32, 960, 1280, False), (32, 1280, 13, False), (32, 13, 1280, True), (2, 1280, 960, True),
                                          (64, 200, 36, False), (7, 2048, 40, True)])
def test_pw_fwd_skinny(K, M, Kd, N, trans, dtype):
    """M <= 64 rows: the classifier head (bias + hardswish epilogue, backward-data with W read transposed)."""
    x = rnd((M, Kd), dtype, 21)
    w = rnd((Kd, N) if trans else (N, Kd), seed=22, scale=Kd ** -0.5)
    bias = rnd((N,), seed=23)
    want = R.pw_fwd(x, w, w_transposed=trans, out_shift=bias, out_act="hardswish")
    got = K.pw_fwd(dev(x), dev(w), w_transposed=trans, out_shift=dev(bias), out_act="hardswish")
    check("pw_fwd_skinny", got, want, dtype)


DW_CASES = [  # N, C, (T,H,W), kernel, stride, pad
    (2, 16, (3, 14, 14), (1, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 64, (4, 13, 13), (1, 3, 3), (2, 2, 2), (1, 1, 1)),
    (1, 72, (3, 9, 11), (1, 5, 5), (2, 2, 2), (2, 2, 2)),
    (2, 120, (2, 7, 7), (1, 5, 5), (1, 1, 1), (2, 2, 2)),
    (1, 184, (2, 6, 6), (1, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 960, (2, 7, 7), (1, 5, 5), (1, 1, 1), (2, 2, 2)),
    (1, 12, (2, 5, 5), (1, 3, 3), (1, 1, 1), (1, 1, 1)),         # C % 8 != 0
    (2, 40, (5, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1)),          # MoViNet generic path
    (1, 96, (4, 9, 9), (5, 3, 3), (1, 2, 2), (2, 1, 1)),
    (1, 48, (3, 9, 9), (1, 5, 5), (1, 2, 2), (0, 2, 2)),
    (1, 6, (3, 4, 4), (3, 3, 3), (1, 1, 1), (1, 1, 1)),           # scalar channel path
    (2, 40, (3, 7, 7), (1, 3, 3), (1, 1, 1), (1, 1, 1)),          # 7x7 whole-frame-in-registers kernels, k=3
    (3, 66, (4, 7, 7), (1, 5, 5), (2, 1, 1), (2, 2, 2)),          # ... with temporal stride/padding (zero frames)
    (2, 48, (12, 11, 11), (3, 3, 3), (1, 1, 1), (2, 1, 1)),       # (kt,3,3) time-walking kernels: MoViNet block padding
    (1, 32, (23, 6, 6), (5, 3, 3), (1, 1, 1), (4, 1, 1)),         # (kt-1 both sides => T grows), several time chunks
    (1, 24, (13, 10, 10), (5, 3, 3), (1, 2, 2), (4, 1, 1)),
    (2, 16, (7, 9, 9), (3, 3, 3), (1, 2, 2), (2, 1, 1)),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,C,thw,kernel,stride,pad", DW_CASES)
def test_dwconv_fwd_bwd(K, N, C, thw, kernel, stride, pad, dtype):
    taps = kernel[0] * kernel[1] * kernel[2]
    x = rnd((N, *thw, C), dtype, 11)
    w = rnd((C, taps), seed=12, scale=taps ** -0.5)
    pool_ref = torch.zeros(N, C)
    y_ref = R.dw_fwd(x, w, kernel, stride, pad, pool=pool_ref)
    pool = torch.zeros(N, C, device=DEV)
    y = K.dw_fwd(dev(x), dev(w), kernel, stride, pad, pool=pool)
    assert tuple(y.shape) == tuple(y_ref.shape)
    check("dw_fwd", y, y_ref, dtype)
    check("dw_pool", pool, pool_ref, dtype)
    dy = rnd(tuple(y_ref.shape), dtype, 13)
    gs, gh = torch.rand(N, C) + 0.5, rnd((N, C), seed=14) * 0.1
    for a, b in ((None, None), (gs, gh)):
        dx_ref = R.dw_bwd_data(dy, w, thw, kernel, stride, pad, a, b)
        dx = K.dw_bwd_data(dev(dy), dev(w), thw, kernel, stride, pad, dev(a), dev(b))
        check("dw_bwd_data", dx, dx_ref, dtype)
        dw_ref = R.dw_bwd_weight(dy, x, torch.zeros(C, taps), kernel, stride, pad, a, b)
        dw = K.dw_bwd_weight(dev(dy), dev(x), torch.zeros(C, taps, device=DEV), kernel, stride, pad, dev(a), dev(b))
        check("dw_bwd_weight", dw, dw_ref, dtype)


RING_CASES = [  # N, C, (T,H,W), k, s : bf16 shapes the LDS-ring depthwise kernel (csrc/dw_ring.hip) takes by default
    (2, 16, (2, 30, 45), 3, 1),      # 8 segments per wave (2 chunks), ragged last segment (right-aligned), narrow segments
    (1, 72, (3, 33, 29), 3, 2),      # 9 chunks: uneven channel slices, stride 2
    (2, 64, (2, 41, 57), 3, 2),      # 29 output columns
    (1, 120, (2, 30, 30), 5, 2),     # k5: 15 chunks, ring depth >= K
    (1, 72, (2, 37, 37), 5, 2),      # 19 output columns: 5 segments of 4
    (1, 184, (3, 29, 29), 3, 2),     # 23 chunks: three slices
    (1, 672, (2, 14, 14), 5, 2),     # MobileNetLarge3D's 14 -> 7 layer: 84 chunks
    (3, 24, (1, 100, 19), 3, 1),     # long frames: many row steps (prologue, steady state, tail), row bands
    (1, 240, (2, 28, 28), 3, 2),
    (2, 32, (2, 9, 16), 5, 1),       # fewer window rows than the steady state needs (tail from the start); k5 backward-data
    (2, 24, (2, 20, 33), 5, 1),
]


@pytest.mark.parametrize("N,C,thw,k,s", RING_CASES)
def test_dwconv_lds_ring_forward_and_backward_data(K, N, C, thw, k, s):
    """csrc/dw_ring.hip (bf16): forward with the squeeze-excite pool sums, and backward-data (flipped taps, with and without
    the (g_scale, g_shift) prologue) for stride 1, against the CPU mirror, on shapes that exercise several segments per wave,
    right-aligned ragged segments, uneven channel slices, row bands, and rings longer / shorter than the band."""
    from picklebot_amd import _C
    import ctypes
    dtype = torch.bfloat16
    kernel, stride, pad = (1, k, k), (1, s, s), (0, k // 2, k // 2)
    x = rnd((N, *thw, C), dtype, 21)
    w = rnd((C, k * k), seed=22, scale=1.0 / k)
    pool_ref = torch.zeros(N, C)
    y_ref = R.dw_fwd(x, w, kernel, stride, pad, pool=pool_ref)
    sh = K._dw_shape(N, C, thw, tuple(y_ref.shape[1:4]), kernel, stride, pad)
    assert _C.load().pb_dwconv3d_kernel_name(ctypes.byref(sh), _C.PB_BF16, 0) == b"dw_ring_fwd_kernel"
    pool = torch.zeros(N, C, device=DEV)
    y = K.dw_fwd(dev(x), dev(w), kernel, stride, pad, pool=pool)
    check("ring_fwd", y, y_ref, dtype)
    check("ring_pool", pool, pool_ref, dtype)
    if s == 1:
        dy = rnd(tuple(y_ref.shape), dtype, 23)
        gs, gh = torch.rand(N, C) + 0.5, rnd((N, C), seed=24) * 0.1
        for a, b in ((None, None), (gs, gh)):
            dx_ref = R.dw_bwd_data(dy, w, thw, kernel, stride, pad, a, b)
            dx = K.dw_bwd_data(dev(dy), dev(w), thw, kernel, stride, pad, dev(a), dev(b))
            check("ring_bwd_data", dx, dx_ref, dtype)


FUSED_CASES = [  # N, C, (T,H,W), k, pt
    (2, 16, (2, 30, 45), 3, 0), (1, 72, (3, 33, 29), 3, 0), (2, 120, (2, 28, 28), 5, 0), (1, 184, (3, 14, 14), 3, 0),
    (1, 672, (2, 14, 14), 3, 0), (3, 24, (1, 100, 19), 3, 0), (2, 40, (2, 9, 16), 5, 2), (1, 600, (3, 10, 11), 5, 0),
    (2, 64, (3, 17, 17), 3, 1),
]


@pytest.mark.parametrize("N,C,thw,k,pt", FUSED_CASES)
def test_dwconv_fused_backward(K, N, C, thw, k, pt):
    """pb_dwconv3d_bwd_fused (bf16, stride 1): data gradient and weight gradient from ONE walk over dy, with and without the
    squeeze-excite prologue g = dy * g_scale + g_shift, against the two mirrors - and against the two separate HIP
    kernels it replaces (the data gradient bit for bit: same taps in the same order)."""
    dtype = torch.bfloat16
    kernel, stride, pad = (1, k, k), (1, 1, 1), (pt, k // 2, k // 2)
    x = rnd((N, *thw, C), dtype, 31)
    w = rnd((C, k * k), seed=32, scale=1.0 / k)
    dy = rnd((N, thw[0] + 2 * pt, thw[1], thw[2], C), dtype, 33)
    assert K.dw_bwd_fused_ws_bytes(dev(dy), thw, kernel, stride, pad) > 0
    assert R.dw_bwd_fused_ws_bytes(dy, thw, kernel, stride, pad) > 0
    gs, gh = torch.rand(N, C) + 0.5, rnd((N, C), seed=34) * 0.1
    for a, b in ((None, None), (gs, gh)):
        dx_ref = R.dw_bwd_data(dy, w, thw, kernel, stride, pad, a, b)
        dw_ref = R.dw_bwd_weight(dy, x, torch.zeros(C, k * k), kernel, stride, pad, a, b)
        dw = torch.zeros(C, k * k, device=DEV)
        dx = K.dw_bwd_fused(dev(dy), dev(x), dev(w), dw, kernel, stride, pad, dev(a), dev(b))
        check("fused_dx", dx, dx_ref, dtype)
        check("fused_dw", dw, dw_ref, dtype)
        dx2 = K.dw_bwd_data(dev(dy), dev(w), thw, kernel, stride, pad, dev(a), dev(b))
        assert torch.equal(dx, dx2)
        dw2 = K.dw_bwd_weight(dev(dy), dev(x), torch.zeros(C, k * k, device=DEV), kernel, stride, pad, dev(a), dev(b))
        check("fused_dw_vs_separate", dw, dw2.cpu(), torch.float32, 5.0)
    # shapes the fused kernel leaves to the two-kernel path
    assert K.dw_bwd_fused_ws_bytes(dev(dy).float(), thw, kernel, stride, pad) == 0
    assert K.dw_bwd_fused_ws_bytes(dev(dy), thw, kernel, (1, 2, 2), pad) == 0


def test_dwconv_temporal_growth_zero_frames(K):
    """(1,k,k) kernel with padding on T: the new frames are exact zeros (SURVEY 'temporal growth quirk')."""
    x = rnd((1, 2, 5, 5, 16), torch.bfloat16, 15)
    w = rnd((16, 25), seed=16)
    y = K.dw_fwd(dev(x), dev(w), (1, 5, 5), (1, 1, 1), (2, 2, 2))
    assert y.shape[1] == 6
    assert float(y[:, :2].abs().max()) == 0.0 and float(y[:, 4:].abs().max()) == 0.0
    assert float(y[:, 2:4].abs().max()) > 0.0


@pytest.mark.parametrize("kernel,stride,pad,bias", [((3, 3, 3), (2, 2, 2), (1, 1, 1), True), ((1, 3, 3), (1, 2, 2), (0, 1, 1), False),
                                                    ((3, 3, 3), (1, 2, 2), (1, 1, 1), True)])
def test_stem_bf16_wide_rows(K, kernel, stride, pad, bias):
    """The reference's input format (bf16, channels-last-3d) on rows longer than one 32-pixel block: the
    slot-staged stem kernels (interior 12-byte loads, block seams, right edge)."""
    N, T, H, W, Cout = 2, 4, 9, 150, 16
    base = torch.rand(N, T, H, W, 3, generator=torch.Generator().manual_seed(6)).to(torch.bfloat16)
    x = base.permute(0, 4, 1, 2, 3)
    w = rnd((Cout, 3, *kernel), seed=27, scale=0.2)
    b = rnd((Cout,), seed=28, scale=0.1) if bias else None
    st_ref, st = torch.zeros(R.REPL, 2, Cout), torch.zeros(K.REPL, 2, Cout, device=DEV)
    y_ref = R.stem_fwd(x, w, b, kernel, stride, pad, torch.bfloat16, 1.0, st_ref)
    xd = base.to(DEV).permute(0, 4, 1, 2, 3)
    y = K.stem_fwd(xd, dev(w), dev(b), kernel, stride, pad, torch.bfloat16, 1.0, st)
    check("stem_fwd_wide", y, y_ref, torch.bfloat16)
    check("stem_stats_wide", st.sum(0), st_ref.sum(0), torch.bfloat16)
    dy = rnd(tuple(y_ref.shape), torch.bfloat16, 29)
    dw_ref, db_ref = torch.zeros_like(w), (torch.zeros(Cout) if bias else None)
    R.stem_bwd_weight(dy, x, dw_ref, db_ref, kernel, stride, pad, 1.0)
    dw, db = torch.zeros_like(w, device=DEV), (torch.zeros(Cout, device=DEV) if bias else None)
    K.stem_bwd_weight(dev(dy), xd, dw, db, kernel, stride, pad, 1.0)
    check("stem_dw_wide", dw, dw_ref, torch.bfloat16)
    if bias:
        check("stem_db_wide", db, db_ref, torch.bfloat16)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("xdtype,layout", [(torch.float32, "ndhwc"), (torch.bfloat16, "ndhwc"), (torch.uint8, "ndhwc"),
                                           (torch.float32, "ncdhw")])
@pytest.mark.parametrize("kernel,stride,pad,bias", [((3, 3, 3), (2, 2, 2), (1, 1, 1), True),
                                                    ((1, 3, 3), (1, 2, 2), (0, 1, 1), False)])
def test_stem(K, dtype, xdtype, layout, kernel, stride, pad, bias):
    N, T, H, W, Cout = 2, 5, 12, 10, 16
    if xdtype == torch.uint8:
        base = torch.randint(0, 256, (N, T, H, W, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(5))
        scale = 1.0 / 255
    else:
        base, scale = torch.rand(N, T, H, W, 3, generator=torch.Generator().manual_seed(5)).to(xdtype), 1.0
    x = base.permute(0, 4, 1, 2, 3)
    if layout == "ncdhw":
        x = x.contiguous()
    w = rnd((Cout, 3, *kernel), seed=17, scale=0.2)
    b = rnd((Cout,), seed=18, scale=0.1) if bias else None
    st_ref, st = torch.zeros(R.REPL, 2, Cout), torch.zeros(K.REPL, 2, Cout, device=DEV)
    y_ref = R.stem_fwd(x, w, b, kernel, stride, pad, dtype, scale, st_ref)
    xd = base.to(DEV).permute(0, 4, 1, 2, 3)
    if layout == "ncdhw":
        xd = xd.contiguous()
    y = K.stem_fwd(xd, dev(w), dev(b), kernel, stride, pad, dtype, scale, st)
    check("stem_fwd", y, y_ref, dtype)
    check("stem_stats", st.sum(0), st_ref.sum(0), dtype)
    dy = rnd(tuple(y_ref.shape), dtype, 19)
    dw_ref, db_ref = torch.zeros_like(w), (torch.zeros(Cout) if bias else None)
    R.stem_bwd_weight(dy, x, dw_ref, db_ref, kernel, stride, pad, scale)
    dw, db = torch.zeros_like(w, device=DEV), (torch.zeros(Cout, device=DEV) if bias else None)
    K.stem_bwd_weight(dev(dy), xd, dw, db, kernel, stride, pad, scale)
    check("stem_dw", dw, dw_ref, dtype)
    if bias:
        check("stem_db", db, db_ref, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C,G", [(6 * 50, 16, 6), (4 * 33, 72, 4), (2 * 97, 184, 2), (3 * 40, 960, 3), (8, 13, 8),
                                   (5 * 7, 2048, 5)])
def test_norm_act_family(K, M, C, G, dtype):
    x = rnd((M, C), dtype, 20)
    st_ref, st = torch.zeros(R.REPL, 2, C), torch.zeros(K.REPL, 2, C, device=DEV)
    R.chan_stats(x, st_ref); K.chan_stats(dev(x), st)
    check("chan_stats", st.sum(0), st_ref.sum(0), dtype)
    gamma, beta = rnd((C,), seed=21) * 0.1 + 1, rnd((C,), seed=22) * 0.1
    rm_ref, rv_ref = rnd((C,), seed=23) * 0.1, torch.rand(C) + 0.5
    rm, rv = dev(rm_ref.clone()), dev(rv_ref.clone())
    out_ref = R.bn_finalize(st_ref, M, gamma, beta, 1e-5, 0.1, rm_ref, rv_ref, False, C, "cpu")
    out = K.bn_finalize(st.clone(), M, dev(gamma), dev(beta), 1e-5, 0.1, rm, rv, False, C, DEV)
    # the GPU statistics differ in the last bits (summation order), so feed both finalizers their own stats
    for a, b, n in zip(out, out_ref, ("scale", "shift", "mean", "rstd")):
        check("bn_" + n, a, b, dtype, 2.0)
    check("running_mean", rm, rm_ref, dtype); check("running_var", rv, rv_ref, dtype)
    scale, shift, mean, rstd = out_ref
    rs, rsh = torch.rand(G, C) + 0.5, rnd((G, C), seed=24) * 0.1
    for act in ("none", "relu", "leaky_relu", "hardswish", "silu", "hardsigmoid"):
        s2_ref, s2 = torch.zeros(R.REPL, 2, C), torch.zeros(K.REPL, 2, C, device=DEV)
        y_ref = R.affine_act_fwd(x, scale, shift, act, 0.01, rs, M // G, s2_ref, rsh, mean=mean)
        y = K.affine_act_fwd(dev(x), dev(scale), dev(shift), act, 0.01, dev(rs), M // G, s2, dev(rsh), mean=dev(mean))
        check("affine_act_" + act, y, y_ref, dtype)
        check("affine_act_stats_" + act, s2.sum(0), s2_ref.sum(0), dtype, 2.0)
        dy = rnd((M, C), dtype, 25)
        sm_ref, sm = torch.zeros(R.REPL, 2, C), torch.zeros(K.REPL, 2, C, device=DEV)
        g_ref = R.act_bwd_reduce(dy, x, scale=scale, shift=shift, act=act, slope=0.01, row_scale=rs, rows_per_group=M // G,
                                 mean=mean, rstd=rstd, sums=sm_ref)
        g = K.act_bwd_reduce(dev(dy), dev(x), scale=dev(scale), shift=dev(shift), act=act, slope=0.01, row_scale=dev(rs),
                             rows_per_group=M // G, mean=dev(mean), rstd=dev(rstd), sums=sm)
        check("act_bwd_" + act, g, g_ref, dtype)
        check("act_bwd_sums_" + act, sm.sum(0), sm_ref.sum(0), dtype, 3.0)
    # broadcast gradient (pool backward)
    dyb = rnd((G, C), dtype, 26)
    sm_ref, sm = torch.zeros(R.REPL, 2, C), torch.zeros(K.REPL, 2, C, device=DEV)
    g_ref = R.act_bwd_reduce(dyb, x, scale=scale, shift=shift, act="hardswish", dy_rows_per_group=M // G, dy_mul=0.25,
                             mean=mean, rstd=rstd, sums=sm_ref)
    g = K.act_bwd_reduce(dev(dyb), dev(x), scale=dev(scale), shift=dev(shift), act="hardswish", dy_rows_per_group=M // G,
                         dy_mul=0.25, mean=dev(mean), rstd=dev(rstd), sums=sm)
    check("act_bwd_bcast", g, g_ref, dtype); check("act_bwd_bcast_sums", sm.sum(0), sm_ref.sum(0), dtype, 3.0)
    fin_ref = R.bn_bwd_finalize(sm_ref, M, gamma, mean, rstd, True, True)
    sm_dev = torch.zeros(K.REPL, 2, C, device=DEV)
    sm_dev[:R.REPL] = dev(sm_ref)
    fin = K.bn_bwd_finalize(sm_dev, M, dev(gamma), dev(mean), dev(rstd), True, True)
    for a, b, n in zip(fin, fin_ref, "ABCgb"):
        check("bn_bwd_fin_" + n, a, b, torch.float32)
    A, B, Cc = fin_ref[:3]
    check("affine2", K.affine2(dev(g_ref), dev(x), dev(A), dev(B), dev(Cc), dev(mean)), R.affine2(g_ref, x, A, B, Cc, mean), dtype)
    # the bottleneck's BatchNorm backward: sums without materialising g, mask applied by the final pass
    sm_ref, sm = torch.zeros(R.REPL, 2, C), torch.zeros(K.REPL, 2, C, device=DEV)
    assert R.act_bwd_reduce(dy, x, scale=scale, shift=shift, row_scale=rs, rows_per_group=M // G, mean=mean, rstd=rstd,
                            sums=sm_ref, want_g=False) is None
    assert K.act_bwd_reduce(dev(dy), dev(x), scale=dev(scale), shift=dev(shift), row_scale=dev(rs), rows_per_group=M // G,
                            mean=dev(mean), rstd=dev(rstd), sums=sm, want_g=False) is None
    check("act_bwd_sums_only", sm.sum(0), sm_ref.sum(0), dtype, 3.0)
    check("affine2_masked", K.affine2(dev(dy), dev(x), dev(A), dev(B), dev(Cc), dev(mean), row_scale=dev(rs), rows_per_group=M // G),
          R.affine2(dy, x, A, B, Cc, mean, row_scale=rs, rows_per_group=M // G), dtype)
    for act in ("hardswish", "relu"):
        check("affine2_act_" + act,
              K.affine2(dev(dy), dev(x), dev(A), dev(B), dev(Cc), dev(mean), row_scale=dev(rs), rows_per_group=M // G,
                        act_scale=dev(scale), act_shift=dev(shift), act=act, slope=0.01),
              R.affine2(dy, x, A, B, Cc, mean, row_scale=rs, rows_per_group=M // G, act_scale=scale, act_shift=shift,
                        act=act, slope=0.01), dtype)
    check("rowgroup_sum", K.rowgroup_reduce(dev(x), None, G), R.rowgroup_reduce(x, None, G), dtype)
    check("rowgroup_dot", K.rowgroup_reduce(dev(x), dev(dy), G), R.rowgroup_reduce(x, dy, G), dtype)


@pytest.mark.parametrize("G,C", [(2, 16), (3, 72), (4, 960), (2, 576)])
def test_se_fc(K, G, C):
    Cm = C // 4
    pool = rnd((G, C), seed=30) * 40
    w1, b1 = rnd((Cm, C), seed=31, scale=C ** -0.5), rnd((Cm,), seed=32, scale=0.1)
    w2, b2 = rnd((C, Cm), seed=33, scale=2 * Cm ** -0.5), rnd((C,), seed=34, scale=0.5)
    s_ref, h_ref = R.se_fc_fwd(pool, 1 / 50, w1, b1, w2, b2)
    s, h = K.se_fc_fwd(dev(pool), 1 / 50, dev(w1), dev(b1), dev(w2), dev(b2))
    check("se_scale", s, s_ref, torch.float32); check("se_hidden", h, h_ref, torch.float32)
    dscale = rnd((G, C), seed=35)
    out_ref = R.se_fc_bwd(dscale, pool, 1 / 50, s_ref, h_ref, w1, w2)
    out = K.se_fc_bwd(dev(dscale), dev(pool), 1 / 50, dev(s_ref), dev(h_ref), dev(w1), dev(w2))
    for a, b, n in zip(out, out_ref, ("g_shift", "dw1", "db1", "dw2", "db2")):
        check("se_bwd_" + n, a, b, torch.float32)


def test_bad_arguments_raise(K):
    x = torch.zeros(8, 16, device=DEV)
    with pytest.raises(ValueError):
        K.pw_fwd(x, torch.zeros(4, 8, device=DEV))                 # K mismatch
    with pytest.raises(TypeError):
        K.pw_fwd(x.double(), torch.zeros(4, 16, device=DEV))       # dtype
    with pytest.raises(RuntimeError):
        K.pw_fwd(torch.zeros(8, 16), torch.zeros(4, 16))           # CPU tensors: no fallback
    with pytest.raises(ValueError):
        K.pw_fwd(x.t(), torch.zeros(4, 8, device=DEV))             # non-contiguous


CONV_CASES = [  # N, Cin, Cout, (T,H,W), kernel, stride, pad
    (2, 16, 16, (3, 6, 6), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 96, 96, (4, 9, 9), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 192, 96, (3, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # 2C -> C (conv4 of MobileViTBlock), K stages 64+64+64
    (1, 48, 80, (5, 7, 7), (3, 3, 3), (1, 1, 1), (1, 1, 1)),        # partial channel stage
    (1, 8, 24, (7, 5, 5), (3, 3, 3), (1, 1, 1), (2, 1, 1)),         # causal-style front padding only on T (back pad 0)
    (2, 16, 8, (4, 9, 9), (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    (2, 64, 136, (4, 12, 12), (3, 3, 3), (1, 1, 1), (1, 1, 1)),     # several row panels x column tiles (last one partial)
    (3, 160, 160, (3, 8, 8), (3, 3, 3), (1, 1, 1), (1, 1, 1)),      # MobileViT-S last stage
    (4, 64, 72, (8, 32, 32), (3, 3, 3), (1, 1, 1), (1, 1, 1)),      # >= 256 voxel blocks: the 64-column halo kernel (second tile partial)
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,Cin,Cout,thw,kernel,stride,pad", CONV_CASES)
def test_conv3d_dense(K, N, Cin, Cout, thw, kernel, stride, pad, dtype):
    taps = kernel[0] * kernel[1] * kernel[2]
    pad_back = (0, pad[1], pad[2]) if pad[0] == 2 else pad
    x = rnd((N, *thw, Cin), dtype, 40)
    w = rnd((Cout, Cin, *kernel), seed=41, scale=(Cin * taps) ** -0.5)
    bias = rnd((Cout,), seed=42, scale=0.1)
    out_thw = R.dw_out_shape(thw, kernel, stride, pad, pad_back)
    wp_ref = R.conv_weight_prep(w.reshape(-1), 0, Cout, Cin, taps)
    wp = K.conv_weight_prep(dev(w).reshape(-1), 0, Cout, Cin, taps)
    check("conv_prep0", wp, wp_ref, torch.float32)
    st_ref, st = torch.zeros(R.REPL, 2, Cout), torch.zeros(K.REPL, 2, Cout, device=DEV)
    y_ref = R.conv3d_fwd(x, wp_ref, Cout, kernel, stride, pad, out_thw, out_shift=bias, stats=st_ref)
    y = K.conv3d_fwd(dev(x), wp, Cout, kernel, stride, pad, out_thw, out_shift=dev(bias), stats=st)
    check("conv3d_fwd", y, y_ref, dtype)
    check("conv3d_stats", st.sum(0), st_ref.sum(0), dtype, 2.0)
    dy = rnd(tuple(y_ref.shape), dtype, 43)
    dwp_ref = R.conv3d_bwd_weight(dy, x, kernel, stride, pad)
    dwp = K.conv3d_bwd_weight(dev(dy), dev(x), kernel, stride, pad)
    check("conv3d_wgrad", dwp, dwp_ref, dtype)
    check("conv_prep2", K.conv_weight_prep(dwp, 2, Cout, Cin, taps), R.conv_weight_prep(dwp_ref, 2, Cout, Cin, taps), dtype)
    if stride == (1, 1, 1):
        wt_ref = R.conv_weight_prep(w.reshape(-1), 1, Cout, Cin, taps)
        check("conv_prep1", K.conv_weight_prep(dev(w).reshape(-1), 1, Cout, Cin, taps), wt_ref, torch.float32)
        pad_t = tuple(k - 1 - p for k, p in zip(kernel, pad))
        dx_ref = R.conv3d_fwd(dy, wt_ref, Cin, kernel, (1, 1, 1), pad_t, thw)
        dx = K.conv3d_fwd(dev(dy), dev(wt_ref), Cin, kernel, (1, 1, 1), pad_t, thw)
        check("conv3d_bwd_data", dx, dx_ref, dtype)
        # and the mirror itself against autograd of the plain conv
        xg = x.float().permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
        pb = [max((o - 1) * s_ + k - i - pf, 0) for o, s_, k, i, pf in zip(out_thw, stride, kernel, thw, pad)]
        import torch.nn.functional as TF
        yy = TF.conv3d(TF.pad(xg, (pad[2], pb[2], pad[1], pb[1], pad[0], pb[0])), w.to(dtype).float(), stride=stride)
        (gx,) = torch.autograd.grad(yy[:, :, :out_thw[0], :out_thw[1], :out_thw[2]], xg, dy.float().permute(0, 4, 1, 2, 3))
        assert rel_err(dx_ref.float().numpy(), gx.permute(0, 2, 3, 4, 1).numpy()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,D,thw", [(2, 16, (3, 5, 5)), (1, 144, (6, 8, 8)), (1, 80, (5, 14, 14)), (2, 96, (4, 7, 7))])
def test_unfold_fold_layernorm_copy(K, B, D, thw, dtype):
    x = rnd((B, *thw, D), dtype, 50)
    tok_ref = R.unfold(x, (2, 2, 2))
    tok = K.unfold(dev(x), (2, 2, 2))
    assert torch.equal(tok.cpu(), tok_ref)
    assert torch.equal(K.fold(tok, thw, (2, 2, 2)).cpu(), R.fold(tok_ref, thw, (2, 2, 2)))
    assert torch.equal(R.fold(tok_ref, thw, (2, 2, 2)), x)                  # fold(unfold(x)) == x
    rows = tok_ref.reshape(-1, D)
    y_ref, st_ref = R.layernorm_fwd(rows, 1e-5)
    y, st = K.layernorm_fwd(dev(rows), 1e-5)
    check("layernorm", y, y_ref, dtype); check("layernorm_stat", st, st_ref, dtype)
    dy = rnd(tuple(rows.shape), dtype, 51)
    check("layernorm_bwd", K.layernorm_bwd(dev(rows), dev(dy), dev(st_ref), 1e-5), R.layernorm_bwd(rows, dy, st_ref, 1e-5), dtype)
    vec = 8 if dtype == torch.bfloat16 else 4
    dst_ref, dst = torch.zeros(rows.shape[0], 2 * D + vec, dtype=dtype), torch.zeros(rows.shape[0], 2 * D + vec, dtype=dtype, device=DEV)
    R.copy_cols(rows, 0, dst_ref, vec, D); K.copy_cols(dev(rows), 0, dst, vec, D)
    R.copy_cols(dy, 0, dst_ref, vec + D, D); K.copy_cols(dev(dy), 0, dst, vec + D, D)
    R.copy_cols(dy, 0, dst_ref, vec, D, True); K.copy_cols(dev(dy), 0, dst, vec, D, True)
    check("copy_cols", dst, dst_ref, dtype)


@pytest.mark.parametrize("M,D", [(37, 64), (1031, 144), (515, 240), (8, 96), (3001, 192)])
def test_wide_token_stream_kernels(K, M, D):
    """pb_layernorm_wide / pb_stream_add (round 4: the Transformer's fp32 residual stream between bf16 blocks) vs mirrors:
    the fp32 outputs to fp32 accuracy, the bf16 outputs bit for bit as roundings of the kernel's own fp32 result."""
    x32 = rnd((M, D), torch.float32, 70, scale=2.0) + 0.3
    b = rnd((M, D), torch.bfloat16, 71)
    xn_ref, st_ref = R.layernorm_wide_fwd(x32, 1e-5)
    xn, st = K.layernorm_wide_fwd(dev(x32), 1e-5)
    assert xn.dtype == torch.bfloat16
    check("ln_wide", xn, xn_ref, torch.bfloat16, 0.5); check("ln_wide_stat", st, st_ref, torch.float32)
    res = rnd((M, D), torch.float32, 72)
    g32_ref, _ = R.layernorm_wide_bwd(x32, b, st_ref, 1e-5, res)
    g32, gbf = K.layernorm_wide_bwd(dev(x32), dev(b), dev(st_ref), 1e-5, dev(res))
    check("ln_wide_bwd", g32, g32_ref, torch.float32)
    assert torch.equal(gbf, g32.to(torch.bfloat16))
    g32b, none = K.layernorm_wide_bwd(dev(x32), dev(b), dev(st_ref), 1e-5, dev(res), want_bf16=False)
    assert none is None and torch.equal(g32b, g32)
    for a in (x32, x32.to(torch.bfloat16)):
        o_ref, ob_ref = R.stream_add(a, b)
        o, ob = K.stream_add(dev(a), dev(b))
        assert torch.equal(o.cpu(), o_ref) and torch.equal(ob.cpu(), ob_ref)          # one fp32 add: exact
        o2, n2 = K.stream_add(dev(a), dev(b), want_bf16=False)
        n3, ob3 = K.stream_add(dev(a), dev(b), want32=False)
        assert n2 is None and n3 is None and torch.equal(o2, o) and torch.equal(ob3, ob)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("G,n", [(8, 6), (3, 32), (2, 147), (2, 128), (1, 784)])
def test_attention(K, G, n, dtype):
    qkv = rnd((G * n, 384), dtype, 60, scale=1.5)
    out_ref, lse_ref = R.attention_fwd(qkv, G, n, 0.25)
    out, lse = K.attention_fwd(dev(qkv), G, n, 0.25)
    check("attn_out", out, out_ref, dtype)
    check("attn_lse", lse, lse_ref, torch.float32, 5.0)
    # and the mirror against torch SDPA (what the reference calls)
    q, k, v = R._split_heads(qkv, G, n)
    sd = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(G * n, 128)
    assert rel_err(out_ref.float().numpy(), sd.numpy()) < TOL[dtype]
    dout = rnd((G * n, 128), dtype, 61)
    dq_ref = R.attention_bwd(qkv, out_ref, dout, lse_ref, G, n, 0.25)
    dq = K.attention_bwd(dev(qkv), dev(out_ref), dev(dout), dev(lse_ref), G, n, 0.25)
    check("attn_dqkv", dq, dq_ref, dtype, 1.5)


@pytest.mark.parametrize("G,n", [(3, 32), (2, 147), (2, 768), (1, 784)])
@pytest.mark.parametrize("amp", [0.5, 1.5, 6.0, "mixed", "outlier", "rising", "negative"])
def test_attention_lazy_running_maximum(K, G, n, amp):
    """bf16 one-pass forward: the running maximum rides in a spare k-slot of the score MFMA (as minus a bf16 value) and
    is only raised when some score exceeds it.  Against the oracle on inputs that exercise the update path: large
    amplitudes, every third query scaled by 8, one huge key per group, keys whose scores keep RISING along the sequence
    (an update in nearly every step), and all-negative scores (first reference below zero); the padding keys of the
    ragged sizes must come out as exact zeros whatever the reference is."""
    bf = torch.bfloat16
    qkv = rnd((G * n, 384), torch.float32, 62, scale=amp if isinstance(amp, float) else 1.0)
    if amp == "mixed":
        qkv[::3, :128] *= 8.0
    elif amp == "outlier":
        qkv[5::n, 128:256] *= 5.0
    elif amp == "rising":
        ramp = torch.linspace(0.2, 6.0, n).repeat(G)[:, None]
        qkv[:, :128] = qkv[:, :128].abs()
        qkv[:, 128:256] = qkv[:, 128:256].abs() * ramp
    elif amp == "negative":
        qkv[:, :128] = qkv[:, :128].abs() * 3.0
        qkv[:, 128:256] = -qkv[:, 128:256].abs() * 3.0
    qkv = qkv.to(bf)
    out_ref, lse_ref = R.attention_fwd(qkv, G, n, 0.25)
    out, lse = K.attention_fwd(dev(qkv), G, n, 0.25)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
    check("attn_out", out, out_ref, bf)
    # the log-sum-exp moves with the re-rounded, pre-scaled Q operand: 2^-9 of the largest score, absolute
    tol = 1e-3 * max(1.0, float(lse_ref.abs().max())) + 2.0 ** -8 * float(lse_ref.abs().max())
    assert float((lse.cpu() - lse_ref).abs().max()) <= tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("G,n,D", [(16, 12, 16), (3, 1, 40), (8, 147, 64), (5, 300, 96), (2, 784, 144), (1, 1000, 240), (7, 33, 256)])
def test_linear_attention(K, G, n, D, dtype):
    """mobilevitv2.py:37-52 on 3-D patch tokens: softmax over the patches, context vector, relu(v)*context; backward of
    all three column blocks.  Two launches of the same input give the same bits (fixed-order sums)."""
    qkv = rnd((G * n, 1 + 2 * D), dtype, 70, scale=1.5)
    out_ref, sc_ref, ctx_ref = R.linattn_fwd(qkv, G, n)
    out, sc, ctx = K.linattn_fwd(dev(qkv), G, n)
    check("linattn_scores", sc, sc_ref, torch.float32, 2.0)
    check("linattn_ctx", ctx, ctx_ref, torch.float32, 2.0)
    check("linattn_out", out, out_ref, dtype)
    out2, _, _ = K.linattn_fwd(dev(qkv), G, n)
    assert torch.equal(out, out2)
    dout = rnd((G * n, D), dtype, 71)
    dq_ref = R.linattn_bwd(qkv, sc_ref, ctx_ref, dout, G, n)
    dq = K.linattn_bwd(dev(qkv), dev(sc_ref), dev(ctx_ref), dev(dout), G, n)
    check("linattn_dqkv", dq, dq_ref, dtype, 1.5)
    assert torch.equal(dq, K.linattn_bwd(dev(qkv), dev(sc_ref), dev(ctx_ref), dev(dout), G, n))


@pytest.mark.parametrize("n", [4096, 1000003, 13])
def test_adamw_flat_and_grad_check(K, n):
    """Fused AdamW over a flat bucket vs torch.optim.AdamW's rule (CPU mirror), several steps incl. the ragged tail;
    unscale + skip-on-inf (GradScaler semantics)."""
    g = torch.Generator().manual_seed(90)
    p0 = torch.randn(n, generator=g)
    m0, v0 = torch.zeros(n), torch.zeros(n)
    p1, m1, v1 = dev(p0.clone()), dev(m0.clone()), dev(v0.clone())
    step_c, step_d = torch.zeros(()), torch.zeros((), device=DEV)
    for it in range(4):
        gr = torch.randn(n, generator=g) * (10.0 ** (it - 2))
        step_c += 1; step_d += 1
        R.adamw_flat(p0, gr, m0, v0, step_c, 3e-3, 0.9, 0.99, 1e-8, 5e-2)
        K.adamw_flat(p1, dev(gr), m1, v1, step_d, 3e-3, 0.9, 0.99, 1e-8, 5e-2)
        check("adamw_p", p1, p0, torch.float32)
        check("adamw_v", v1, v0, torch.float32)
    # unscale
    gr = torch.randn(n, generator=g)
    inv = torch.tensor(1.0 / 256.0)
    step_c += 1; step_d += 1
    R.adamw_flat(p0, gr * 256.0, m0, v0, step_c, 3e-3, 0.9, 0.99, 1e-8, 5e-2, inv_scale=inv)
    K.adamw_flat(p1, dev(gr * 256.0), m1, v1, step_d, 3e-3, 0.9, 0.99, 1e-8, 5e-2, inv_scale=dev(inv))
    check("adamw_unscale", p1, p0, torch.float32)
    # inf check + skip
    found = torch.zeros((), device=DEV)
    K.grad_check(dev(gr), found)
    assert float(found) == 0.0
    bad = gr.clone(); bad[n // 2] = float("nan")
    K.grad_check(dev(bad), found)
    assert float(found) == 1.0
    before = p1.clone()
    K.adamw_flat(p1, dev(bad), m1, v1, step_d, 3e-3, 0.9, 0.99, 1e-8, 5e-2, found_inf=found)
    assert torch.equal(before, p1)


def test_flat_adamw_lr_follows_scheduler_under_graph_replay(K):
    """A step captured into a HIP graph reads (lr, weight_decay) from device memory: changing param_groups[0]["lr"] and
    calling refresh_hyper() before a replay changes the update (the host float baked into the graph would not)."""
    import picklebot_amd
    torch.manual_seed(5)
    n = 4096 + 3
    p = torch.nn.Parameter(torch.randn(n, device=DEV))
    opt = picklebot_amd.FlatAdamW([p], lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
    assert isinstance(opt, torch.optim.Optimizer)
    g = torch.randn(n, device=DEV)
    opt.gflat[0][:n].copy_(g)
    ref_p, ref_m, ref_v = p.detach().cpu().clone(), torch.zeros(n), torch.zeros(n)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        opt.step(packed=True)                                  # warm-up step (eager), counted below
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    R.adamw_flat(ref_p, g.cpu(), ref_m, ref_v, torch.tensor(1.0), 1e-2, 0.9, 0.99, 1e-8, 1e-2)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt.step(packed=True)
    step = 1.0                                                 # (capture records the step, it does not run it)
    for lr in (1e-2, 1e-3, 5e-2):
        opt.param_groups[0]["lr"] = lr
        opt.refresh_hyper()
        graph.replay()
        step += 1.0
        R.adamw_flat(ref_p, g.cpu(), ref_m, ref_v, torch.tensor(step), lr, 0.9, 0.99, 1e-8, 1e-2)
        torch.cuda.synchronize()
        check(f"adamw_graph_lr{lr}", p.detach(), ref_p, torch.float32)
    assert float(opt.step_t) == step


@pytest.mark.parametrize("G,C,Cm,N,slots", [(3, 72, 18, 24, 5), (2, 960, 240, 160, 19), (4, 120, 30, 40, 40), (1, 16, 4, 8, 1),
                                            (2, 576, 144, 96, 33)])
def test_se_fused_forward_backward(K, G, C, Cm, N, slots):
    """The one-launch squeeze-excite gate (slot sum + two GEMVs + per-clip scaled bf16 projection weights) and the fused
    backward against the mirrors."""
    part = rnd((slots, G, C), seed=40) * 3
    w1, b1, w2, b2 = rnd((Cm, C), seed=41) * C ** -0.5, rnd((Cm,), seed=42) * 0.1, rnd((C, Cm), seed=43) * Cm ** -0.5, rnd((C,), seed=44) * 0.1
    wproj = rnd((N, C), seed=45)
    pool_r, scale_r, hid_r, _ = R.se_fused_fwd(part, 1 / 50, w1, b1, w2, b2)
    pool, scale, hid, wbs = K.se_fused_fwd(dev(part), 1 / 50, dev(w1), dev(b1), dev(w2), dev(b2), wproj=dev(wproj), want_scaled=True)
    check("se_pool", pool, pool_r, torch.float32); check("se_scale", scale, scale_r, torch.float32)
    check("se_hidden", hid, hid_r, torch.float32)
    assert wbs is not None and wbs.dtype == torch.bfloat16
    check("se_wbs", wbs, (wproj[None] * scale_r[:, None, :]), torch.bfloat16)
    dpart = rnd((slots, G, C), seed=46)
    out_r = R.se_fused_bwd(dpart, pool_r, 1 / 50, scale_r, hid_r, w1, w2)
    out = K.se_fused_bwd(dev(dpart), dev(pool_r), 1 / 50, dev(scale_r), dev(hid_r), dev(w1), dev(w2))
    for a, b, name in zip(out, out_r, ("g_shift", "dw1", "db1", "dw2", "db2")):
        check("se_fused_bwd_" + name, a, b, torch.float32, 2.0)
    # bit-reproducible
    again = K.se_fused_bwd(dev(dpart), dev(pool_r), 1 / 50, dev(scale_r), dev(hid_r), dev(w1), dev(w2))
    assert all(torch.equal(a, b) for a, b in zip(out, again))


@pytest.mark.parametrize("dtype", DTYPES)
def test_pool_and_rowgroup_partials_sum_to_the_totals(K, dtype):
    x = rnd((2, 5, 14, 14, 72), dtype, 50)
    w = rnd((72, 9), seed=51)
    y, part = K.dw_fwd_pool(dev(x), dev(w), (1, 3, 3), (1, 1, 1), (1, 1, 1))
    y_ref, part_ref = R.dw_fwd_pool(x, w, (1, 3, 3), (1, 1, 1), (1, 1, 1))
    check("dw_fwd_pool_y", y, y_ref, dtype); check("dw_fwd_pool_sum", part.sum(0), part_ref.sum(0), dtype)
    a, b = rnd((6 * 70, 40), dtype, 52), rnd((6 * 70, 40), dtype, 53)
    check("rowgroup_partials", K.rowgroup_partials(dev(a), dev(b), 6).sum(0), R.rowgroup_reduce(a, b, 6), dtype)


@pytest.mark.parametrize("G,n", [(2, 32), (3, 147), (2, 768), (1, 784), (5, 16), (1, 1000)])
def test_attention_fp8_forward(K, G, n):
    """fp8 (e4m3) MFMA operands, fp32 softmax statistics (BASELINE config 5).  The reference has no fp8 path: parity
    unpinned beyond the bf16 mirror.  Measured on MI355X against the fp32-arithmetic mirror on the same bf16 inputs:
    4-6e-2 relative per output tensor on these random inputs (e4m3 carries 3 mantissa bits: ~6 % per element, the
    near-uniform attention of random data averages little of it away); bound 8e-2.  On the real model the logits move by
    1e-2 (test_mobilevit_xxs_fp8_attention_forward)."""
    qkv = rnd((G * n, 384), torch.bfloat16, 60)
    want, lse_ref = R.attention_fwd(qkv, G, n, 0.25)
    got, lse = K.attention_fwd(dev(qkv), G, n, 0.25, fp8=True)
    e = rel_err(got.float().cpu().numpy(), want.float().numpy())
    assert e <= 8e-2, e
    assert rel_err(lse.cpu().numpy(), lse_ref.numpy()) <= 2e-2
    bf, _ = K.attention_fwd(dev(qkv), G, n, 0.25)
    assert rel_err(bf.float().cpu().numpy(), want.float().numpy()) <= 1.2e-2


def test_partials_reduce_more_than_eight_jobs_with_empty_and_accumulate(K):
    """pb_partials_reduce batches 8 jobs per launch; empty jobs are skipped and must not shift the batches (a job run
    twice would double-add with accumulate=1)."""
    g = torch.Generator().manual_seed(5)
    jobs, want, outs = [], [], []
    for j in range(13):
        n = [64, 0, 37, 128, 4, 0, 256, 19, 8, 1000, 12, 0, 333][j]
        split = [3, 2, 5, 1, 7, 4, 2, 9, 1, 3, 6, 1, 2][j]
        acc = j % 2
        part = torch.randn(split, max(n, 1), generator=g).to(DEV)
        out = torch.randn(max(n, 1), generator=g).to(DEV)
        ref = (out.double().cpu() if acc else 0) + part.double().cpu().sum(0)
        jobs.append((part, out, n, max(n, 1), split, acc))
        outs.append(out); want.append(ref if n else out.double().cpu().clone())
    K.partials_reduce(jobs, torch.device(DEV))
    for j, (o, w) in enumerate(zip(outs, want)):
        assert torch.allclose(o.double().cpu(), w, rtol=1e-5, atol=1e-5), f"job {j}"


# ---------------------------------------------------------------------------- fused expansion conv -> depthwise conv
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,T,frame,t0,step", [(2, 10, (5, 6, 8), 1, 2), (3, 8, (4, 4, 16), 0, 2), (1, 7, (3, 3, 24), 2, 3), (2, 5, (2, 2, 8), 4, 2)])
def test_frames_gather_scatter(K, N, T, frame, t0, step, dtype):
    """Temporally strided bottlenecks read every `step`-th frame only: gather == strided slice, scatter == its adjoint with
    exact zeros in the unread frames (bit-exact: pure data movement)."""
    x = rnd((N, T) + frame, dtype, 61)
    count = (T - 1 - t0) // step + 1
    got = K.frames_gather(dev(x), t0, step, count)
    assert torch.equal(got.cpu(), R.frames_gather(x, t0, step, count))
    back = K.frames_scatter(got, T, t0, step)
    assert torch.equal(back.cpu(), R.frames_scatter(R.frames_gather(x, t0, step, count), T, t0, step))
    unread = [t for t in range(T) if t < t0 or (t - t0) % step]
    assert all(float(back[:, t].abs().max()) == 0.0 for t in unread)
    fill = rnd((frame[-1],), dtype, 62)
    filled = K.frames_scatter(got, T, t0, step, fill=dev(fill))
    assert torch.equal(filled.cpu(), R.frames_scatter(R.frames_gather(x, t0, step, count), T, t0, step, fill=fill))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("G,full,data,lead,C", [(3, 40, 24, 8, 16), (2, 90, 54, 18, 72), (4, 35, 21, 0, 24), (2, 33, 11, 22, 160), (1, 12, 12, 0, 8)])
def test_norm_act_frame_map(K, G, full, data, lead, C, dtype):
    """BatchNorm-apply and its backward reduction on a COMPACT pre-activation (the data frames only; the frames a (1,k,k)
    depthwise conv pads in time read as exact zeros): full output rows, sums over all rows, compact g."""
    z = rnd((G * data, C), dtype, 71)
    dy = rnd((G * full, C), dtype, 72)
    scale, shift, mean = rnd((C,), seed=73) * 0.2 + 1, rnd((C,), seed=74) * 0.3, rnd((C,), seed=75) * 0.1
    rstd = torch.rand(C) + 0.5
    drop = (torch.rand(G, C) > 0.3).float() * 1.4
    fm = (full, data, lead)
    want = R.affine_act_fwd(z, scale, shift, "hardswish", 0.0, row_scale=drop, rows_per_group=full, mean=mean, frame_map=fm)
    got = K.affine_act_fwd(dev(z), dev(scale), dev(shift), "hardswish", 0.0, row_scale=dev(drop), rows_per_group=full, mean=dev(mean), frame_map=fm)
    assert tuple(got.shape) == (G * full, C)
    check("affine_act_fm", got, want, dtype)
    s_ref = torch.zeros(R.REPL, 2, C)
    g_ref = R.act_bwd_reduce(dy, z, scale=scale, shift=shift, act="relu", row_scale=drop, rows_per_group=full, mean=mean, rstd=rstd,
                             sums=s_ref, frame_map=fm)
    s = torch.zeros(K.REPL, 2, C, device=DEV)
    g = K.act_bwd_reduce(dev(dy), dev(z), scale=dev(scale), shift=dev(shift), act="relu", row_scale=dev(drop), rows_per_group=full,
                         mean=dev(mean), rstd=dev(rstd), sums=s, frame_map=fm)
    assert tuple(g.shape) == (G * data, C)
    check("act_bwd_fm_g", g, g_ref, dtype)
    check("act_bwd_fm_sums", s.sum(0), s_ref.sum(0), dtype, tol_scale=3.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("G,rows,C,Cm,Co,act", [(32, 49, 960, 1280, 13, "hardswish"), (5, 30, 576, 1024, 2, "hardswish"),
                                                (3, 17, 40, 24, 7, "relu")])
def test_pooled_head_kernels(K, G, rows, C, Cm, Co, act, dtype):
    """pool -> fc -> hardswish / ReLU -> fc (the MobileNet classifiers) on the squeeze-excite kernels with Co != C: forward,
    the data-gradient pair and the weight gradients against the mirror; bit-identical reruns."""
    x = rnd((G * rows, C), dtype, 71)
    w1, b1 = rnd((Cm, C), seed=72, scale=C ** -0.5), rnd((Cm,), seed=73) * 0.1
    w2, b2 = rnd((Co, Cm), seed=74, scale=Cm ** -0.5), rnd((Co,), seed=75) * 0.1
    pool_r, pre_r, lg_r = R.pool_head_fwd(R.rowgroup_partials(x, None, G), 1.0 / rows, w1, b1, w2, b2, act, dtype)
    part = K.rowgroup_partials(dev(x), None, G)
    pool, pre, lg = K.pool_head_fwd(part, 1.0 / rows, dev(w1), dev(b1), dev(w2), dev(b2), act, dtype)
    assert lg.dtype == dtype
    check("head_pool", pool, pool_r, torch.float32)
    check("head_pre1", pre, pre_r if act == "hardswish" else torch.relu(pre_r), torch.float32)     # (ReLU: its output is kept)
    check("head_logits", lg, lg_r, dtype)
    pool2, pre2, lg2 = K.pool_head_fwd(part, 1.0 / rows, dev(w1), dev(b1), dev(w2), dev(b2), act, dtype)
    assert torch.equal(pre, pre2) and torch.equal(lg, lg2)
    dl = rnd((G, Co), seed=76)
    row_r, work_r = R.pool_head_bwd(dl, 1.0 / rows, pre_r, w1, w2, act, dtype)
    row, work = K.pool_head_bwd(dev(dl), 1.0 / rows, pre, dev(w1), dev(w2), act, dtype)
    assert row.dtype == dtype
    check("head_row", row, row_r, dtype)
    check("head_work", work, work_r, torch.float32)
    gr = [torch.zeros(Cm, C), torch.zeros(Cm), torch.zeros(Co, Cm), torch.zeros(Co)]
    R.pool_head_bwd_w(work_r, pool_r, 1.0 / rows, pre_r, act, *gr)
    gg = [dev(torch.zeros_like(t)) for t in gr]
    K.pool_head_bwd_w(work, pool, 1.0 / rows, pre, act, *gg)
    for n, a_, b_ in zip(("dw1", "db1", "dw2", "db2"), gg, gr):
        check("head_" + n, a_, b_, torch.float32)
