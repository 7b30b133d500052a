"""Per-kernel CPU restatement (TEST INFRASTRUCTURE ONLY - never imported by picklebot_amd).

Function-for-function mirror of picklebot_amd/kernels.py in plain torch, fp32 arithmetic with
the output rounded to the activation dtype exactly where the HIP kernel rounds.  Two uses:
  * -m gpu tests compare every HIP kernel against its mirror on the same seeded inputs;
  * CPU tests monkeypatch these into picklebot_amd.kernels to check the fused forward/backward
    DATAFLOW (picklebot_amd/functional.py) against the oracle models' autograd without a GPU.
Each mirror is itself pinned by tests/test_kernel_ref_cpu.py against torch.nn.functional ops
(conv3d / batch_norm / ... i.e. the stock ops the reference calls).
"""
import torch
import torch.nn.functional as F

REPL = 4    # statistics buffers are [slots, 2, C] (PB_STAT_SLOTS in the product); the mirror adds into slot 0


class deferred_reductions:
    """Mirror of kernels.deferred_reductions (the product batches its fixed-order partial reductions; the mirror has
    no partials)."""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class held_reductions(deferred_reductions):
    """Mirror of kernels.held_reductions."""
    jobs = ()


def weight_grad_branch(ref, params, thunks, keep, tag="", held=(), now=False):
    """Mirror of kernels.weight_grad_branch (the product may run a block's weight-gradient launches on a side stream and
    join later; the mirror runs them in place)."""
    for t in thunks:
        t()


def join_weight_grads():
    pass


def zeros(shape, device):
    """Mirror of kernels.zeros (the product carves these from a pre-zeroed arena)."""
    return torch.zeros(shape, dtype=torch.float32, device=device)


def _act(x, act, slope=0.0):
    if act == "none":
        return x
    if act == "relu":
        return torch.relu(x)
    if act == "leaky_relu":
        return torch.where(x > 0, x, x * slope)
    if act == "hardswish":
        return x * torch.clamp(x + 3, 0, 6) / 6
    if act == "hardsigmoid":
        return torch.clamp(x / 6 + 0.5, 0, 1)
    if act == "silu":
        return x * torch.sigmoid(x)
    raise ValueError(act)


def _act_grad(x, act, slope=0.0):
    one, zero = torch.ones_like(x), torch.zeros_like(x)
    if act == "none":
        return one
    if act == "relu":
        return torch.where(x > 0, one, zero)
    if act == "leaky_relu":
        return torch.where(x > 0, one, one * slope)
    if act == "hardswish":
        return torch.where(x < -3, zero, torch.where(x <= 3, x / 3 + 0.5, one))
    if act == "hardsigmoid":
        return torch.where((x > -3) & (x < 3), one / 6, zero)
    if act == "silu":
        s = torch.sigmoid(x)
        return s * (1 + x * (1 - s))
    raise ValueError(act)


def _prologue(x, in_scale, in_shift, in_act, in_slope, row_scale, rows_per_group):
    xf = x.float()
    if in_scale is not None:
        xf = xf * in_scale
    if in_shift is not None:
        xf = xf + in_shift
    xf = _act(xf, in_act, in_slope)
    if row_scale is not None:
        xf = xf * row_scale.repeat_interleave(rows_per_group, 0)
    return xf.to(x.dtype).float()      # the kernel rounds the staged operand to the activation dtype


def pw_fwd(x, w, *, w_transposed=False, in_scale=None, in_shift=None, in_act="none", in_slope=0.0, row_scale=None,
           rows_per_group=0, out_scale=None, out_shift=None, out_act="none", out_slope=0.0, stats=None, residual=None,
           scaled_weights=None):
    xf = _prologue(x, in_scale, in_shift, in_act, in_slope, row_scale, rows_per_group)
    wl = (w.t() if w_transposed else w).to(x.dtype).float()
    acc = xf @ wl.t()
    if out_scale is not None:
        acc = acc * out_scale
    if out_shift is not None:
        acc = acc + out_shift
    y = _act(acc, out_act, out_slope).to(x.dtype)
    if stats is not None:
        stats[0, 0] += y.float().sum(0)
        stats[0, 1] += (y.float() ** 2).sum(0)
    if residual is not None:
        y = (y.float() + residual.float()).to(x.dtype)
    return y


def pw_bwd_weight(dy, x, dw, *, in_scale=None, in_shift=None, in_act="none", in_slope=0.0, row_scale=None,
                  rows_per_group=0):
    xf = _prologue(x, in_scale, in_shift, in_act, in_slope, row_scale, rows_per_group)
    dw += dy.float().t() @ xf
    return dw


def dw_out_shape(in_thw, kernel, stride, pad_front, pad_back=None):
    pad_back = pad_front if pad_back is None else pad_back
    return tuple((i + pf + pb - k) // s + 1 for i, k, s, pf, pb in zip(in_thw, kernel, stride, pad_front, pad_back))


def _pad_ncdhw(x5, pad_front, pad_back):
    """[N,T,H,W,C] -> padded NCDHW float."""
    xn = x5.permute(0, 4, 1, 2, 3).float()
    return F.pad(xn, (pad_front[2], pad_back[2], pad_front[1], pad_back[1], pad_front[0], pad_back[0]))


def dw_fwd(x, w, kernel, stride, pad_front, pad_back=None, pool=None):
    pad_back = pad_front if pad_back is None else pad_back
    C = x.shape[4]
    y = F.conv3d(_pad_ncdhw(x, pad_front, pad_back), w.view(C, 1, *kernel), stride=stride, groups=C)
    y = y.permute(0, 2, 3, 4, 1).contiguous().to(x.dtype)
    if pool is not None:
        pool.copy_(y.float().sum((1, 2, 3)))
    return y


def dw_fwd_pool(x, w, kernel, stride, pad_front, pad_back=None):
    """Mirror of kernels.dw_fwd_pool: one slot holding the whole sum."""
    pool = torch.zeros(x.shape[0], x.shape[4])
    y = dw_fwd(x, w, kernel, stride, pad_front, pad_back, pool=pool)
    return y, pool.unsqueeze(0)


def dw_fwd_hist(x, hist, w, kernel, stride, pad_front, pad_back=None):
    """Mirror of kernels.dw_fwd_hist: the history frames replace the last nhist frames of the front padding."""
    nhist = hist.shape[1]
    return dw_fwd(torch.cat([hist, x], 1), w, kernel, stride, (pad_front[0] - nhist, pad_front[1], pad_front[2]),
                  pad_front if pad_back is None else pad_back)


def dw_bwd_fused_ws_bytes(dy, in_thw, kernel, stride, pad_front):
    """Mirror of kernels.dw_bwd_fused_ws_bytes: which shapes the one-launch depthwise backward takes (bf16, (1,k,k), stride 1,
    "same" spatial padding, not the 7x7 frames)."""
    ok = (dy.dtype == torch.bfloat16 and kernel[0] == 1 and kernel[1] == kernel[2] and kernel[1] in (3, 5) and tuple(stride) == (1, 1, 1)
          and pad_front[1] == kernel[1] // 2 and pad_front[2] == kernel[2] // 2 and dy.shape[4] % 2 == 0 and dy.shape[4] >= 16
          and tuple(dy.shape[2:4]) == tuple(in_thw[1:]) and dy.shape[1] == in_thw[0] + 2 * pad_front[0] and tuple(in_thw[1:]) != (7, 7))
    return 4 if ok else 0


def dw_bwd_fused(dy, x, w, dw, kernel, stride, pad_front, g_scale=None, g_shift=None):
    dx = dw_bwd_data(dy, w, tuple(x.shape[1:4]), kernel, stride, pad_front, g_scale, g_shift)
    dw_bwd_weight(dy, x, dw, kernel, stride, pad_front, g_scale, g_shift)
    return dx


def act_row(shift, act, slope, dtype):
    return affine_act_fwd(shift.view(1, -1), act=act, slope=slope).to(dtype).view(-1)


def frames_gather(x, t0, tstep, count):
    return x[:, t0:t0 + (count - 1) * tstep + 1:tstep].contiguous()


def frames_scatter(xs, T, t0, tstep, fill=None):
    out = torch.zeros((xs.shape[0], T) + tuple(xs.shape[2:]), dtype=xs.dtype)
    if fill is not None:
        out[:] = fill
    out[:, t0:t0 + (xs.shape[1] - 1) * tstep + 1:tstep] = xs
    return out


def _g(dy, g_scale, g_shift):
    g = dy.float()
    if g_scale is not None:
        g = g * g_scale[:, None, None, None, :]
    if g_shift is not None:
        g = g + g_shift[:, None, None, None, :]
    return g


@torch.enable_grad()
def dw_bwd_data(dy, w, in_thw, kernel, stride, pad_front, g_scale=None, g_shift=None):
    N, To, Ho, Wo, C = dy.shape
    g = _g(dy, g_scale, g_shift).permute(0, 4, 1, 2, 3)
    xin = torch.zeros((N, C) + tuple(in_thw), requires_grad=True)
    pb = [(o - 1) * s + k - i - pf for o, s, k, i, pf in zip((To, Ho, Wo), stride, kernel, in_thw, pad_front)]
    pb = [max(p, 0) for p in pb]
    xp = F.pad(xin, (pad_front[2], pb[2], pad_front[1], pb[1], pad_front[0], pb[0]))
    y = F.conv3d(xp, w.view(C, 1, *kernel), stride=stride, groups=C)[:, :, :To, :Ho, :Wo]
    (dx,) = torch.autograd.grad(y, xin, g)
    return dx.permute(0, 2, 3, 4, 1).contiguous().to(dy.dtype)


@torch.enable_grad()
def dw_bwd_weight(dy, x, dw, kernel, stride, pad_front, g_scale=None, g_shift=None):
    N, To, Ho, Wo, C = dy.shape
    g = _g(dy, g_scale, g_shift).permute(0, 4, 1, 2, 3)
    wv = w_ = torch.zeros(C, 1, *kernel, requires_grad=True)
    in_thw = tuple(x.shape[1:4])
    pb = [max((o - 1) * s + k - i - pf, 0) for o, s, k, i, pf in zip((To, Ho, Wo), stride, kernel, in_thw, pad_front)]
    xp = F.pad(x.permute(0, 4, 1, 2, 3).float(), (pad_front[2], pb[2], pad_front[1], pb[1], pad_front[0], pb[0]))
    y = F.conv3d(xp, wv, stride=stride, groups=C)[:, :, :To, :Ho, :Wo]
    (gw,) = torch.autograd.grad(y, w_, g)
    dw += gw.view(C, -1)
    return dw


def stem_fwd(x, w, bias, kernel, stride, pad, out_dtype, x_scale=1.0, stats=None):
    xf = x.float() * x_scale
    y = F.conv3d(xf, w, bias, stride=stride, padding=pad).permute(0, 2, 3, 4, 1).contiguous().to(out_dtype)
    if stats is not None:
        yf = y.float().reshape(-1, y.shape[-1])
        stats[0, 0] += yf.sum(0)
        stats[0, 1] += (yf ** 2).sum(0)
    return y


@torch.enable_grad()
def stem_bwd_weight(dy, x, dw, dbias, kernel, stride, pad, x_scale=1.0):
    wv = torch.zeros_like(dw, requires_grad=True)
    y = F.conv3d(x.float() * x_scale, wv, None, stride=stride, padding=pad)
    (gw,) = torch.autograd.grad(y, wv, dy.float().permute(0, 4, 1, 2, 3))
    dw += gw
    if dbias is not None:
        dbias += dy.float().sum((0, 1, 2, 3))
    return dw


def chan_stats(x, stats):
    stats[0, 0] += x.float().sum(0)
    stats[0, 1] += (x.float() ** 2).sum(0)
    return stats


def bn_finalize(stats, count, gamma, beta, eps, momentum, running_mean, running_var, use_running, C, device,
                num_batches_tracked=None):
    if not use_running and num_batches_tracked is not None:
        num_batches_tracked += 1
    if use_running:
        mean, var = running_mean.double(), running_var.double()
    else:
        st = stats.double().sum(0)
        mean = st[0] / count
        var = (st[1] / count - mean * mean).clamp_min(0)
        if running_mean is not None:
            running_mean.copy_(((1 - momentum) * running_mean.double() + momentum * mean).float())
        if running_var is not None:
            unb = var * count / (count - 1) if count > 1 else var
            running_var.copy_(((1 - momentum) * running_var.double() + momentum * unb).float())
    rstd = 1.0 / torch.sqrt(var + eps)
    g = gamma.double() if gamma is not None else torch.ones(C, dtype=torch.float64)
    b = beta.double() if beta is not None else torch.zeros(C, dtype=torch.float64)
    return (g * rstd).float(), (b - mean * g * rstd).float(), mean.float(), rstd.float()


def _fm_expand(t, frame_map):
    """compact rows -> full rows with exact zeros on the padded ones (mirror of the kernels' frame map)."""
    full, data, lead = frame_map
    n = t.shape[0] // data
    out = torch.zeros((n, full, t.shape[1]), dtype=t.dtype)
    out[:, lead:lead + data] = t.view(n, data, -1)
    return out.view(n * full, -1)


def _fm_compact(t, frame_map):
    full, data, lead = frame_map
    n = t.shape[0] // full
    return t.view(n, full, -1)[:, lead:lead + data].reshape(n * data, -1).contiguous()


def affine_act_fwd(x, scale=None, shift=None, act="none", slope=0.0, row_scale=None, rows_per_group=0, stats=None,
                   row_shift=None, mean=None, frame_map=None):
    if frame_map is not None:
        x = _fm_expand(x, frame_map)
    xf = x.float()
    if mean is not None:
        xf = xf - mean
    if scale is not None:
        xf = xf * scale
    if shift is not None:
        xf = xf + shift
    y = _act(xf, act, slope)
    if row_scale is not None:
        y = y * row_scale.repeat_interleave(rows_per_group, 0)
    if row_shift is not None:
        y = y + row_shift.repeat_interleave(rows_per_group, 0)
    y = y.to(x.dtype)
    if stats is not None:
        stats[0, 0] += y.float().sum(0)
        stats[0, 1] += (y.float() ** 2).sum(0)
    return y


def act_bwd_reduce(dy, z, *, scale=None, shift=None, act="none", slope=0.0, row_scale=None, rows_per_group=0,
                   dy_rows_per_group=0, dy_mul=1.0, mean=None, rstd=None, sums=None, want_g=True, frame_map=None):
    if frame_map is not None:
        g = act_bwd_reduce(dy, _fm_expand(z, frame_map), scale=scale, shift=shift, act=act, slope=slope, row_scale=row_scale,
                           rows_per_group=rows_per_group, dy_rows_per_group=dy_rows_per_group, dy_mul=dy_mul, mean=mean, rstd=rstd,
                           sums=sums, want_g=True)
        return _fm_compact(g, frame_map) if want_g else None
    zf = z.float()
    pre = zf if mean is None else zf - mean
    if scale is not None:
        pre = pre * scale
    if shift is not None:
        pre = pre + shift
    d = dy.float()
    if dy_rows_per_group > 0:
        d = d.repeat_interleave(dy_rows_per_group, 0)
    g = d * dy_mul
    if row_scale is not None:
        g = g * row_scale.repeat_interleave(rows_per_group, 0)
    g = g * _act_grad(pre, act, slope)
    gr = g.to(z.dtype).float() if want_g else g
    if sums is not None:
        xhat = zf
        if mean is not None:
            xhat = xhat - mean
        if rstd is not None:
            xhat = xhat * rstd
        sums[0, 0] += gr.sum(0)
        sums[0, 1] += (gr * xhat).sum(0)
    return g.to(z.dtype) if want_g else None


def bn_bwd_finalize(sums, count, gamma, mean, rstd, training, want_affine_grads):
    C = mean.shape[0]
    g = gamma.double() if gamma is not None else torch.ones(C, dtype=torch.float64)
    r, mu = rstd.double(), mean.double()
    sg, sgx = sums.double().sum(0)
    A = g * r
    if training:
        B = -g * r * r * (sgx / count)
        Cc = -g * r * (sg / count)
    else:
        B, Cc = torch.zeros_like(A), torch.zeros_like(A)
    return (A.float(), B.float(), Cc.float(), sgx.float() if want_affine_grads else None,
            sg.float() if want_affine_grads else None)


def affine2(g, z, A, B=None, Cc=None, mean=None, row_scale=None, rows_per_group=0, act_scale=None, act_shift=None,
            act="none", slope=0.0):
    gf = g.float()
    if row_scale is not None:
        gf = gf * row_scale.repeat_interleave(rows_per_group, 0)
    if act != "none":
        pre = z.float() if mean is None else z.float() - mean
        if act_scale is not None:
            pre = pre * act_scale
        if act_shift is not None:
            pre = pre + act_shift
        gf = gf * _act_grad(pre, act, slope)
    out = gf * A
    if B is not None:
        zf = z.float() if mean is None else z.float() - mean
        out = out + zf * B
    if Cc is not None:
        out = out + Cc
    return out.to(g.dtype)


def rowgroup_reduce(a, b, G):
    M, C = a.shape
    v = a.float() if b is None else a.float() * b.float()
    return v.view(G, M // G, C).sum(1)


def se_fc_fwd(pool, inv_count, w1, b1, w2, b2):
    mean = pool * inv_count
    h = mean @ w1.t()
    if b1 is not None:
        h = h + b1
    h = torch.relu(h)
    pre = h @ w2.t()
    if b2 is not None:
        pre = pre + b2
    return torch.clamp(pre / 6 + 0.5, 0, 1), h


def se_fused_fwd(pool_part, inv_count, w1, b1, w2, b2, wproj=None, want_scaled=False):
    pool = pool_part.sum(0)
    scale, hidden = se_fc_fwd(pool, inv_count, w1, b1, w2, b2)
    return pool, scale, hidden, None


def _head_act(x, act):
    return F.hardswish(x) if act == "hardswish" else torch.relu(x)


def pool_head_fwd(pool_part, inv_count, w1, b1, w2, b2, act, out_dtype):
    """mirror of kernels.pool_head_fwd (fp32 throughout; logits rounded to out_dtype at the end)"""
    pool = pool_part.sum(0)
    pre1 = (pool * inv_count) @ w1.t() + (b1 if b1 is not None else 0)
    logits = _head_act(pre1, act) @ w2.t() + (b2 if b2 is not None else 0)
    return pool, pre1, logits.to(out_dtype)


def pool_head_bwd(dlogits, inv_count, pre1, w1, w2, act, row_dtype):
    h = pre1.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        a = _head_act(h, act)
    dpre1, = torch.autograd.grad(a, h, dlogits @ w2)
    row = (dpre1 @ w1) * inv_count
    return row.to(row_dtype), torch.cat([dlogits, dpre1], dim=1)


def pool_head_bwd_w(work, pool, inv_count, pre1, act, dw1, db1, dw2, db2):
    Cm = pre1.shape[1]
    Co = work.shape[1] - Cm
    dlogits, dpre1 = work[:, :Co], work[:, Co:]
    dw2 += dlogits.t() @ _head_act(pre1, act)
    dw1 += dpre1.t() @ (pool * inv_count)
    if db2 is not None:
        db2 += dlogits.sum(0)
    if db1 is not None:
        db1 += dpre1.sum(0)


def rowgroup_partials(a, b, G):
    return rowgroup_reduce(a, b, G).unsqueeze(0)


def se_fused_bwd(dscale_part, pool, inv_count, scale, hidden, w1, w2, want_bias=True, defer_weights=False):
    out = se_fc_bwd(dscale_part.sum(0), pool, inv_count, scale, hidden, w1, w2, want_bias)
    if defer_weights:          # (the product launches the FC weight gradients later, from the block's weight-gradient branch)
        def weights():
            pass
        weights.keep = ()
        return out + (weights,)
    return out


def se_fc_bwd(dscale, pool, inv_count, scale, hidden, w1, w2, want_bias=True):
    dpre2 = torch.where((scale > 0) & (scale < 1), dscale / 6, torch.zeros_like(dscale))
    dh = (dpre2 @ w2) * (hidden > 0).float()
    g_shift = (dh @ w1) * inv_count
    dw2 = dpre2.t() @ hidden
    dw1 = dh.t() @ (pool * inv_count)
    return g_shift, dw1, (dh.sum(0) if want_bias else None), dw2, (dpre2.sum(0) if want_bias else None)


def conv_weight_prep(w, mode, cout, cin, taps):
    if mode == 0:
        return w.reshape(cout, cin, taps).permute(0, 2, 1).contiguous().reshape(-1)
    if mode == 1:
        return w.reshape(cout, cin, taps).flip(2).permute(1, 2, 0).contiguous().reshape(-1)
    return w.reshape(cout, taps, cin).permute(0, 2, 1).contiguous().reshape(-1)


def conv3d_fwd(x, wp, cout, kernel, stride, pad_front, out_thw, *, out_scale=None, out_shift=None, out_act="none",
               out_slope=0.0, stats=None):
    N, T, H, W, cin = x.shape
    taps = kernel[0] * kernel[1] * kernel[2]
    w = wp.reshape(cout, taps, cin).permute(0, 2, 1).reshape(cout, cin, *kernel).to(x.dtype).float()
    pb = [max((o - 1) * s + k - i - pf, 0) for o, s, k, i, pf in zip(out_thw, stride, kernel, (T, H, W), pad_front)]
    xp = F.pad(x.permute(0, 4, 1, 2, 3).float(), (pad_front[2], pb[2], pad_front[1], pb[1], pad_front[0], pb[0]))
    acc = F.conv3d(xp, w, stride=stride)[:, :, :out_thw[0], :out_thw[1], :out_thw[2]].permute(0, 2, 3, 4, 1)
    if out_scale is not None:
        acc = acc * out_scale
    if out_shift is not None:
        acc = acc + out_shift
    y = _act(acc, out_act, out_slope).contiguous().to(x.dtype)
    if stats is not None:
        yf = y.float().reshape(-1, cout)
        stats[0, 0] += yf.sum(0)
        stats[0, 1] += (yf ** 2).sum(0)
    return y


@torch.enable_grad()
def conv3d_bwd_weight(dy, x, kernel, stride, pad_front):
    N, To, Ho, Wo, cout = dy.shape
    cin = x.shape[4]
    in_thw = tuple(x.shape[1:4])
    wv = torch.zeros(cout, cin, *kernel, requires_grad=True)
    pb = [max((o - 1) * s + k - i - pf, 0) for o, s, k, i, pf in zip((To, Ho, Wo), stride, kernel, in_thw, pad_front)]
    xp = F.pad(x.permute(0, 4, 1, 2, 3).float(), (pad_front[2], pb[2], pad_front[1], pb[1], pad_front[0], pb[0]))
    y = F.conv3d(xp, wv, stride=stride)[:, :, :To, :Ho, :Wo]
    (gw,) = torch.autograd.grad(y, wv, dy.float().permute(0, 4, 1, 2, 3))
    return gw.reshape(cout, cin, -1).permute(0, 2, 1).contiguous().reshape(-1)


def unfold(x5, patch):
    B, T, H, W, D = x5.shape
    pt, ph, pw = patch
    tp, hp, wp = (-T) % pt, (-H) % ph, (-W) % pw
    x = F.pad(x5, (0, 0, 0, wp, 0, hp, 0, tp))
    t, h, w = (T + tp) // pt, (H + hp) // ph, (W + wp) // pw
    x = x.reshape(B, t, pt, h, ph, w, pw, D).permute(0, 2, 4, 6, 1, 3, 5, 7)
    return x.reshape(B, pt * ph * pw, t * h * w, D).contiguous()


def fold(tok, thw, patch):
    B, P_, n, D = tok.shape
    T, H, W = thw
    pt, ph, pw = patch
    t, h, w = -(-T // pt), -(-H // ph), -(-W // pw)
    x = tok.reshape(B, pt, ph, pw, t, h, w, D).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(B, t * pt, h * ph, w * pw, D)
    return x[:, :T, :H, :W].contiguous()


def layernorm_fwd(x, eps):
    xf = x.float()
    mean = xf.mean(1, keepdim=True)
    rstd = torch.rsqrt(xf.var(1, unbiased=False, keepdim=True) + eps)
    return ((xf - mean) * rstd).to(x.dtype), torch.cat([mean, rstd], 1)


def layernorm_bwd(x, dy, stat, eps):
    xf, g = x.float(), dy.float()
    xh = (xf - stat[:, :1]) * stat[:, 1:]
    dx = stat[:, 1:] * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True))
    return dx.to(x.dtype)


def layernorm_wide_fwd(x32, eps):
    mean = x32.mean(1, keepdim=True)
    rstd = torch.rsqrt(x32.var(1, unbiased=False, keepdim=True) + eps)
    return ((x32 - mean) * rstd).to(torch.bfloat16), torch.cat([mean, rstd], 1)


def layernorm_wide_bwd(x32, dxn, stat, eps, res32, want_bf16=True):
    g = dxn.float()
    xh = (x32 - stat[:, :1]) * stat[:, 1:]
    g32 = res32 + stat[:, 1:] * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True))
    return g32, (g32.to(torch.bfloat16) if want_bf16 else None)


def stream_add(a, b, want32=True, want_bf16=True):
    o = a.float() + b.float()
    return (o if want32 else None), (o.to(torch.bfloat16) if want_bf16 else None)


def copy_cols(src, src_off, dst, dst_off, C, accumulate=False):
    if accumulate:
        dst[:, dst_off:dst_off + C] += src[:, src_off:src_off + C]
    else:
        dst[:, dst_off:dst_off + C] = src[:, src_off:src_off + C]
    return dst


def _split_heads(qkv, G, n):
    q, k, v = qkv.float().view(G, n, 3, 8, 16).permute(2, 0, 3, 1, 4)      # each [G, 8, n, 16]
    return q, k, v


def attention_fwd(qkv, G, n, scale, fp8=False):
    q, k, v = _split_heads(qkv, G, n)
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    lse = torch.logsumexp(s, -1)
    p = torch.exp(s - lse[..., None])
    if qkv.dtype == torch.bfloat16:
        p = p.to(torch.bfloat16).float()        # the kernel feeds P to the MFMA in bf16
    out = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(G * n, 128)
    return out.to(qkv.dtype), lse.permute(0, 2, 1).reshape(G * n, 8).contiguous()


def attention_bwd(qkv, out, dout, lse, G, n, scale):
    q, k, v = _split_heads(qkv, G, n)
    do = dout.float().view(G, n, 8, 16).permute(0, 2, 1, 3)
    o = out.float().view(G, n, 8, 16).permute(0, 2, 1, 3)
    L = lse.view(G, n, 8).permute(0, 2, 1)
    p = torch.exp(torch.matmul(q, k.transpose(-1, -2)) * scale - L[..., None])
    delta = (do * o).sum(-1, keepdim=True)
    dp = torch.matmul(do, v.transpose(-1, -2))
    ds = p * (dp - delta) * scale
    dv = torch.matmul(p.transpose(-1, -2), do)
    dq = torch.matmul(ds, k)
    dk = torch.matmul(ds.transpose(-1, -2), q)
    dqkv = torch.stack([dq, dk, dv], 0).permute(1, 3, 0, 2, 4).reshape(G * n, 384)
    return dqkv.to(qkv.dtype)


def linattn_fwd(qkv, G, n):
    """mobilevitv2.py:37-52 on token rows [G*n, 1+2D] (q | k | v)."""
    D = (qkv.shape[1] - 1) // 2
    x = qkv.float().view(G, n, 1 + 2 * D)
    scores = torch.softmax(x[..., 0], dim=-1)                       # over the n patches (:41)
    ctx = (x[..., 1:1 + D] * scores[..., None]).sum(1)              # [G, D] (:46-48)
    out = torch.relu(x[..., 1 + D:]) * ctx[:, None, :]              # (:52)
    return out.reshape(G * n, D).to(qkv.dtype), scores.reshape(G * n).contiguous(), ctx.contiguous()


def linattn_bwd(qkv, scores, ctx, dout, G, n):
    D = (qkv.shape[1] - 1) // 2
    x = qkv.float().view(G, n, 1 + 2 * D)
    k, v = x[..., 1:1 + D], x[..., 1 + D:]
    do = dout.float().view(G, n, D)
    a = scores.view(G, n)
    dctx = (do * torch.relu(v)).sum(1)                              # [G, D]
    dv = do * ctx[:, None, :] * (v > 0)
    dk = a[..., None] * dctx[:, None, :]
    da = (k * dctx[:, None, :]).sum(-1)                             # [G, n]
    dq = a * (da - (a * da).sum(-1, keepdim=True))
    return torch.cat([dq[..., None], dk, dv], -1).reshape(G * n, 1 + 2 * D).to(qkv.dtype)


def adamw_flat(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay, inv_scale=None, found_inf=None, lr_wd=None):
    """torch.optim.AdamW's single-tensor update (torch/optim/adamw.py) on flat buffers, in place.  lr_wd: the (lr,
    weight_decay) pair as a tensor, which the product's kernel reads from device memory instead of the host values."""
    if found_inf is not None and float(found_inf) != 0.0:
        return
    if lr_wd is not None:
        lr, weight_decay = float(lr_wd[0]), float(lr_wd[1])
    t = float(step)
    gg = g * (float(inv_scale) if inv_scale is not None else 1.0)
    p.mul_(1 - lr * weight_decay)
    m.lerp_(gg, 1 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** t, 1 - beta2 ** t
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def grad_check(g, found_inf):
    if not bool(torch.isfinite(g).all()):
        found_inf.fill_(1.0)
