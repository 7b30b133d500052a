"""CPU: every per-kernel mirror (oracle/kernel_ref.py - the oracle the GPU kernel tests compare the HIP kernels with)
pinned to the STOCK torch op the reference calls at that site: torch.nn.functional conv3d with groups /
strides / asymmetric padding, batch_norm, layer_norm, scaled_dot_product_attention, linear, torch.optim.AdamW - forward
values and, through autograd on the stock op, every backward mirror.  Includes the branches no model case reaches
(asymmetric pad_back, strided dense weight gradient, found_inf, broadcast gradients)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import kernel_ref as R

torch.manual_seed(7)


def close(a, b, tol=2e-5):
    a, b = a.double(), b.double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float((a - b).norm() / max(float(b.norm()), 1e-30))
    assert err <= tol, err


def ndhwc(t):
    return t.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(t):
    return t.permute(0, 4, 1, 2, 3)


ACTS = {"none": lambda x: x, "relu": F.relu, "leaky_relu": lambda x: F.leaky_relu(x, 0.01), "hardswish": F.hardswish,
        "hardsigmoid": F.hardsigmoid, "silu": F.silu}


@pytest.mark.parametrize("act", list(ACTS))
def test_pointwise_forward_and_gradients_vs_linear(act):
    M, K, N, G = 24, 10, 7, 3
    x, w = torch.randn(M, K), torch.randn(N, K)
    in_scale, in_shift, rs = torch.rand(K) + 0.5, torch.randn(K) * 0.1, torch.rand(G, K) + 0.5
    out_scale, out_shift = torch.rand(N) + 0.5, torch.randn(N) * 0.1
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    px = ACTS[act](xr * in_scale + in_shift) * rs.repeat_interleave(M // G, 0)
    y = ACTS[act](F.linear(px, wr) * out_scale + out_shift)
    st = torch.zeros(R.REPL, 2, N)
    got = R.pw_fwd(x, w, in_scale=in_scale, in_shift=in_shift, in_act=act, in_slope=0.01, row_scale=rs, rows_per_group=M // G,
                   out_scale=out_scale, out_shift=out_shift, out_act=act, out_slope=0.01, stats=st)
    close(got, y.detach())
    close(st.sum(0)[0], y.detach().sum(0)); close(st.sum(0)[1], (y.detach() ** 2).sum(0))
    # transposed weights (backward-data form) and the residual epilogue
    close(R.pw_fwd(x, w.t().contiguous(), w_transposed=True), F.linear(x, w))
    res = torch.randn(M, N)
    close(R.pw_fwd(x, w, residual=res), F.linear(x, w) + res)
    # weight gradient of the plain product with the same prologue
    dy = torch.randn(M, N)
    (gw,) = torch.autograd.grad(F.linear(px, wr), wr, dy)
    close(R.pw_bwd_weight(dy, x, torch.zeros(N, K), in_scale=in_scale, in_shift=in_shift, in_act=act, in_slope=0.01,
                          row_scale=rs, rows_per_group=M // G), gw)


@pytest.mark.parametrize("kernel,stride,pf,pb", [((1, 3, 3), (1, 1, 1), (1, 1, 1), None), ((1, 5, 5), (2, 2, 2), (2, 2, 2), None),
                                                 ((3, 3, 3), (1, 2, 2), (2, 1, 1), (0, 1, 1)), ((5, 3, 3), (1, 1, 1), (4, 1, 1), (0, 1, 1)),
                                                 ((2, 3, 1), (1, 1, 2), (1, 0, 0), (0, 2, 0))])
def test_depthwise_family_vs_conv3d(kernel, stride, pf, pb):
    N, C, thw = 2, 6, (5, 7, 6)
    x = torch.randn(N, C, *thw)
    w = torch.randn(C, 1, *kernel)
    pbb = pf if pb is None else pb
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xp = F.pad(xr, (pf[2], pbb[2], pf[1], pbb[1], pf[0], pbb[0]))
    y = F.conv3d(xp, wr, stride=stride, groups=C)
    pool = torch.zeros(N, C)
    got = R.dw_fwd(ndhwc(x), w.view(C, -1), kernel, stride, pf, pb, pool=pool)
    close(got, ndhwc(y.detach())); close(pool, y.detach().sum((2, 3, 4)))
    assert tuple(got.shape[1:4]) == R.dw_out_shape(thw, kernel, stride, pf, pb)
    # backward with the fused squeeze-excite terms g = dy*g_scale + g_shift
    dy = torch.randn_like(y)
    gs, gh = torch.rand(N, C) + 0.5, torch.randn(N, C) * 0.1
    g = dy * gs[:, :, None, None, None] + gh[:, :, None, None, None]
    gx, gw = torch.autograd.grad(y, (xr, wr), g)
    close(R.dw_bwd_data(ndhwc(dy), w.view(C, -1), thw, kernel, stride, pf, gs, gh), ndhwc(gx))
    close(R.dw_bwd_weight(ndhwc(dy), ndhwc(x), torch.zeros(C, w[0].numel()), kernel, stride, pf, gs, gh), gw.view(C, -1))
    # history-buffer form == the same conv on [history ++ chunk] with the front padding shortened
    if pf[0] >= 2:
        hist = torch.randn(N, C, 2, *thw[1:])
        want = F.conv3d(F.pad(torch.cat([hist, x], 2), (pf[2], pbb[2], pf[1], pbb[1], pf[0] - 2, pbb[0])), w, stride=stride, groups=C)
        close(R.dw_fwd_hist(ndhwc(x), ndhwc(hist), w.view(C, -1), kernel, stride, pf, pb), ndhwc(want))


@pytest.mark.parametrize("stride,pad", [((1, 1, 1), (1, 1, 1)), ((2, 2, 2), (1, 1, 1)), ((1, 2, 2), (0, 1, 1))])
def test_dense_conv_and_stem_vs_conv3d(stride, pad):
    N, cin, cout, kernel, thw = 2, 4, 8, (3, 3, 3), (5, 6, 7)
    x, w, b = torch.randn(N, cin, *thw), torch.randn(cout, cin, *kernel), torch.randn(cout)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv3d(xr, wr, b, stride=stride, padding=pad)
    out_thw = tuple(y.shape[2:])
    taps = 27
    wp = R.conv_weight_prep(w.reshape(-1), 0, cout, cin, taps)
    st = torch.zeros(R.REPL, 2, cout)
    got = R.conv3d_fwd(ndhwc(x), wp, cout, kernel, stride, pad, out_thw, out_shift=b, stats=st)
    close(got, ndhwc(y.detach())); close(st.sum(0)[0], y.detach().sum((0, 2, 3, 4)))
    dy = torch.randn_like(y)
    gx, gw = torch.autograd.grad(y, (xr, wr), dy)
    dwp = R.conv3d_bwd_weight(ndhwc(dy), ndhwc(x), kernel, stride, pad)
    close(R.conv_weight_prep(dwp, 2, cout, cin, taps).view_as(w), gw)
    if stride == (1, 1, 1):          # backward-data = the forward entry on dy with the transposed, tap-reversed filter
        wt = R.conv_weight_prep(w.reshape(-1), 1, cout, cin, taps)
        pad_t = tuple(k - 1 - p for k, p in zip(kernel, pad))
        close(R.conv3d_fwd(ndhwc(dy), wt, cin, kernel, (1, 1, 1), pad_t, thw), ndhwc(gx))
    # stem: the same conv on a strided NCDHW view, /255 folded in, bias and statistics
    u8 = torch.randint(0, 256, (N, *thw, 3), dtype=torch.uint8).permute(0, 4, 1, 2, 3)
    ws, bs = torch.randn(16, 3, *kernel), torch.randn(16)
    wsr = ws.clone().requires_grad_(True)
    ys = F.conv3d(u8.float() / 255, wsr, bs, stride=stride, padding=pad)
    st = torch.zeros(R.REPL, 2, 16)
    close(R.stem_fwd(u8, ws, bs, kernel, stride, pad, torch.float32, 1 / 255, st), ndhwc(ys.detach()), 1e-5)
    close(st.sum(0)[1], (ys.detach() ** 2).sum((0, 2, 3, 4)), 1e-5)
    dys = torch.randn_like(ys)
    (gws,) = torch.autograd.grad(ys, wsr, dys)
    dw, db = torch.zeros_like(ws), torch.zeros(16)
    R.stem_bwd_weight(ndhwc(dys), u8, dw, db, kernel, stride, pad, 1 / 255)
    close(dw, gws, 1e-5); close(db, dys.sum((0, 2, 3, 4)))


@pytest.mark.parametrize("act", ["none", "relu", "hardswish", "silu", "leaky_relu"])
@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_activation_family_vs_batch_norm(act, training):
    """chan_stats + bn_finalize + affine_act_fwd == F.batch_norm -> act (-> channel-dropout row scale), and
    act_bwd_reduce + bn_bwd_finalize + affine2 == its autograd gradients (dz, dgamma, dbeta), train and eval mode."""
    M, C, G = 40, 6, 4
    z = torch.randn(M, C) * 2 + 0.5
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.2
    rm, rv = torch.randn(C) * 0.1, torch.rand(C) + 0.5
    drop = (torch.rand(G, C) > 0.3).float() / 0.7
    zr, gr, br = z.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = ACTS[act](F.batch_norm(zr, rm_ref, rv_ref, gr, br, training, 0.1, 1e-5)) * drop.repeat_interleave(M // G, 0)
    st = torch.zeros(R.REPL, 2, C)
    R.chan_stats(z, st)
    rm2, rv2, nbt = rm.clone(), rv.clone(), torch.zeros((), dtype=torch.int64)
    scale, shift, mean, rstd = R.bn_finalize(None if not training else st, M, gamma, beta, 1e-5, 0.1, rm2, rv2, not training, C, "cpu",
                                             num_batches_tracked=nbt)
    got = R.affine_act_fwd(z, scale, beta, act, 0.01, row_scale=drop, rows_per_group=M // G, mean=mean)
    close(got, y.detach())
    close(rm2, rm_ref); close(rv2, rv_ref)
    assert int(nbt) == (1 if training else 0)
    dy = torch.randn(M, C)
    dz_ref, dg_ref, db_ref = torch.autograd.grad(y, (zr, gr, br), dy)
    sums = torch.zeros(R.REPL, 2, C)
    g = R.act_bwd_reduce(dy, z, scale=scale, shift=beta, act=act, slope=0.01, row_scale=drop, rows_per_group=M // G, mean=mean,
                         rstd=rstd, sums=sums)
    A, B, Cc, dgamma, dbeta = R.bn_bwd_finalize(sums, M, gamma, mean, rstd, training, True)
    dz = R.affine2(g, z, A, B, Cc, mean=mean) if training else R.affine2(g, None, A)
    close(dz, dz_ref, 1e-4); close(dgamma, dg_ref, 1e-4); close(dbeta, db_ref, 1e-4)
    # the no-materialised-g form: sums only, then the final pass recomputes mask and act'
    sums2 = torch.zeros(R.REPL, 2, C)
    assert R.act_bwd_reduce(dy, z, scale=scale, shift=beta, act=act, slope=0.01, row_scale=drop, rows_per_group=M // G, mean=mean,
                            rstd=rstd, sums=sums2, want_g=False) is None
    A, B, Cc, _, _ = R.bn_bwd_finalize(sums2, M, gamma, mean, rstd, training, True)
    dz2 = R.affine2(dy, z, A, B, Cc, mean=mean, row_scale=drop, rows_per_group=M // G, act_scale=scale, act_shift=beta, act=act, slope=0.01)
    close(dz2, dz_ref, 1e-4)
    # broadcast gradient (global-average-pool backward): dy [G,C] repeated over each row group, times dy_mul
    dyb = torch.randn(G, C)
    gb = R.act_bwd_reduce(dyb, z, scale=scale, shift=beta, act=act, slope=0.01, dy_rows_per_group=M // G, dy_mul=0.25, mean=mean, rstd=rstd)
    pre = ((z - mean) * scale + beta).requires_grad_(True)              # g is the gradient w.r.t. the BatchNorm output
    (want,) = torch.autograd.grad(ACTS[act](pre), pre, dyb.repeat_interleave(M // G, 0) * 0.25)
    close(gb, want, 1e-4)


def test_squeeze_excite_vs_conv1x1_autograd():
    """SEBlock3D (mobilenet.py:11-26): hardsigmoid(W2 relu(W1 mean + b1) + b2) and its gradients, incl. the g_shift term the
    depthwise backward adds to every element (d loss / d mean / count)."""
    G, C, Cm, R_ = 3, 8, 2, 11
    x = torch.randn(G, R_, C, requires_grad=True)
    w1, b1, w2, b2 = (torch.randn(Cm, C, requires_grad=True), torch.randn(Cm, requires_grad=True),
                      torch.randn(C, Cm, requires_grad=True), torch.randn(C, requires_grad=True))
    mean = x.mean(1)
    scale_ref = F.hardsigmoid(F.linear(F.relu(F.linear(mean, w1, b1)), w2, b2))
    pool = x.detach().sum(1)
    scale, hidden = R.se_fc_fwd(pool, 1 / R_, w1.detach(), b1.detach(), w2.detach(), b2.detach())
    close(scale, scale_ref.detach()); close(hidden, F.relu(F.linear(mean, w1, b1)).detach())
    dscale = torch.randn(G, C)
    gx, gw1, gb1, gw2, gb2 = torch.autograd.grad(scale_ref, (x, w1, b1, w2, b2), dscale)
    g_shift, dw1, db1, dw2, db2 = R.se_fc_bwd(dscale, pool, 1 / R_, scale, hidden, w1.detach(), w2.detach())
    close(g_shift, gx[:, 0]); close(dw1, gw1); close(db1, gb1); close(dw2, gw2); close(db2, gb2)
    close(R.rowgroup_reduce(x.detach().reshape(-1, C), None, G), pool)
    y = torch.randn(G * R_, C)
    close(R.rowgroup_reduce(x.detach().reshape(-1, C), y, G), (x.detach().reshape(-1, C) * y).view(G, R_, C).sum(1))


def test_token_ops_vs_einops_and_layer_norm():
    from einops import rearrange
    B, D, T, H, W = 2, 5, 3, 5, 4
    x5 = torch.randn(B, T, H, W, D)
    xp = F.pad(x5.permute(0, 4, 1, 2, 3), (0, 0, 0, 1, 0, 1))                    # mobilevit.py:129-132: pad to even
    want = rearrange(xp, "b d (t pt) (h ph) (w pw) -> b (pt ph pw) (t h w) d", pt=2, ph=2, pw=2)      # mobilevit.py:134
    tok = R.unfold(x5, (2, 2, 2))
    close(tok, want)
    back = rearrange(want, "b (pt ph pw) (t h w) d -> b d (t pt) (h ph) (w pw)", pt=2, ph=2, pw=2, t=2, h=3, w=2)[:, :, :T, :H, :W]
    close(R.fold(tok, (T, H, W), (2, 2, 2)), back.permute(0, 2, 3, 4, 1))
    rows = torch.randn(9, 16, requires_grad=True)
    y = F.layer_norm(rows, (16,), eps=1e-5)
    got, stat = R.layernorm_fwd(rows.detach(), 1e-5)
    close(got, y.detach())
    dy = torch.randn(9, 16)
    (gx,) = torch.autograd.grad(y, rows, dy)
    close(R.layernorm_bwd(rows.detach(), dy, stat, 1e-5), gx, 1e-4)
    dst = torch.zeros(9, 40)
    R.copy_cols(rows.detach(), 4, dst, 10, 8)
    assert torch.equal(dst[:, 10:18], rows.detach()[:, 4:12]) and float(dst[:, :10].abs().sum() + dst[:, 18:].abs().sum()) == 0
    R.copy_cols(rows.detach(), 4, dst, 10, 8, accumulate=True)
    close(dst[:, 10:18], 2 * rows.detach()[:, 4:12])


def test_attention_vs_scaled_dot_product_attention():
    """mobilevit.py:71-74 + the head merge of :82: qkv rows [G*n, 384] -> 'b p h n d' -> SDPA -> 'b p n (h d)'."""
    G, n = 3, 10
    qkv = torch.randn(G * n, 384, requires_grad=True)
    q, k, v = qkv.view(G, n, 3, 8, 16).permute(2, 0, 3, 1, 4)
    o = F.scaled_dot_product_attention(q, k, v)
    want = o.permute(0, 2, 1, 3).reshape(G * n, 128)
    got, lse = R.attention_fwd(qkv.detach(), G, n, 16 ** -0.5)
    close(got, want.detach(), 1e-5)
    dout = torch.randn(G * n, 128)
    (gq,) = torch.autograd.grad(want, qkv, dout)
    close(R.attention_bwd(qkv.detach(), got, dout, lse, G, n, 16 ** -0.5), gq, 1e-4)


def test_linear_attention_vs_torch_restatement_of_mobilevitv2():
    """mobilevitv2.py:37-52 between qkv_proj and out_proj, in the reference's (b, c, pixels, patches) layout."""
    b, P, n, D = 2, 3, 7, 5
    G = b * P
    qkv = torch.randn(G * n, 1 + 2 * D, requires_grad=True)
    x = qkv.view(b, P, n, 1 + 2 * D).permute(0, 3, 1, 2)                  # (b, 1+2D, P, n)
    q, k, v = torch.split(x, [1, D, D], dim=1)
    want = (F.relu(v) * (k * F.softmax(q, dim=-1)).sum(-1, keepdim=True)).permute(0, 2, 3, 1).reshape(G * n, D)
    got, scores, ctx = R.linattn_fwd(qkv.detach(), G, n)
    close(got, want.detach(), 1e-6)
    close(scores.view(G, n).sum(-1), torch.ones(G), 1e-6)
    dout = torch.randn(G * n, D)
    (gq,) = torch.autograd.grad(want, qkv, dout)
    close(R.linattn_bwd(qkv.detach(), scores, ctx, dout, G, n), gq, 1e-5)


def test_adamw_flat_vs_torch_optim_adamw_incl_skip():
    n = 37
    p0, g = torch.randn(n), torch.randn(n)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=5e-2)
    p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    for t in (1, 2, 3):
        ref.grad = g * t
        opt.step()
        R.adamw_flat(p, g * t * 128.0, m, v, torch.tensor(float(t)), 3e-3, 0.9, 0.99, 1e-8, 5e-2, inv_scale=torch.tensor(1 / 128.0),
                     found_inf=torch.tensor(0.0))
        close(p, ref.detach(), 1e-6)
    before = p.clone()
    R.adamw_flat(p, g, m, v, torch.tensor(4.0), 3e-3, 0.9, 0.99, 1e-8, 5e-2, found_inf=torch.tensor(1.0))      # skipped update
    assert torch.equal(p, before)
    found = torch.zeros(())
    R.grad_check(torch.tensor([1.0, float("inf")]), found)
    assert float(found) == 1.0
    # the device-side (lr, weight_decay) pair overrides the host values
    q, mq, vq = p0.clone(), torch.zeros(n), torch.zeros(n)
    R.adamw_flat(q, g, mq, vq, torch.tensor(1.0), 123.0, 0.9, 0.99, 1e-8, 9.0, lr_wd=torch.tensor([3e-3, 5e-2]))
    q2, m2, v2 = p0.clone(), torch.zeros(n), torch.zeros(n)
    R.adamw_flat(q2, g, m2, v2, torch.tensor(1.0), 3e-3, 0.9, 0.99, 1e-8, 5e-2)
    close(q, q2, 1e-6)
