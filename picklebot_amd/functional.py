"""Fused forward/backward dataflows as torch.autograd.Functions over the HIP kernels.

One Function per reference building block, so a block costs a handful of launches and the
intermediate tensors move through HBM the minimum number of times the grid-wide dependencies
(squeeze-excite mean, train-mode BatchNorm statistics) allow:

  bottleneck   Bottleneck3D / MoviNetBottleneck   (mobilenet.py:84-93, movinet.py:69-77)
  stem_bn_act  Conv3d(3,C) -> BN -> act            (mobilenet.py:140-144, movinet.py:91-95, mobilevit.py:165)
  pw_bn_act    Conv3d 1x1x1 [-> SE] -> BN -> act   (mobilenet.py:178-182, 244-249; movinet.py:139-144; mobilevit.py:12-17)
  pool_head    AdaptiveAvgPool3d -> 1x1x1 conv -> Hardswish -> 1x1x1 conv   (mobilenet.py:185-190)

All activations are contiguous NDHWC tensors [N,T,H,W,C] (float32 or bfloat16); parameters are
the reference-shaped fp32 tensors.  `K` is the kernel namespace (picklebot_amd.kernels): tests
swap it for the CPU mirror to check this file's dataflow without a GPU; the product never does.
"""
import torch

from . import kernels as K


def _zeros(shape, ref):
    return torch.zeros(shape, dtype=torch.float32, device=ref.device)


def _zeros_many(ref, *shapes):
    """Several zero-filled fp32 buffers from ONE allocation + fill (each 16-byte aligned): the per-step
    count of tiny fill kernels is what this saves (they were 0.6 ms of a 23 ms step)."""
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 3) // 4 * 4
    flat = torch.zeros(total, dtype=torch.float32, device=ref.device)
    return [flat[o:o + n].view(s) for o, n, s in zip(offs, sizes, shapes)]


def _contig5(t):
    return t if t.is_contiguous() else t.contiguous()


def _bn_state(bn):
    """(gamma, beta, running_mean, running_var, eps, momentum) from an nn.BatchNorm module or None."""
    if bn is None:
        return None
    if bn.momentum is None:
        raise NotImplementedError("picklebot_amd: BatchNorm with momentum=None (cumulative average) is not supported")
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps), float(bn.momentum))


def _bn_forward(z2, stats, count, gamma, beta, rmean, rvar, eps, momentum, training):
    """-> (scale, shift, mean, rstd); updates running stats in training."""
    use_running = not training
    if not use_running and rmean is None:
        rmean_, rvar_ = None, None
    else:
        rmean_, rvar_ = rmean, rvar
    C = (stats.shape[-1] if stats is not None else rmean.shape[0])
    return K.bn_finalize(None if use_running else stats, count, gamma, beta, eps, momentum, rmean_, rvar_,
                         use_running, C, z2.device)


class _BnCtx:
    """What BatchNorm backward needs."""
    __slots__ = ("scale", "shift", "mean", "rstd", "training", "count")


def _bn_act_backward(dy2, z2, bnc, gamma, act, slope, row_scale, rows_per_group, dy_rows_per_group=0, dy_mul=1.0):
    """Gradient through [row_scale *] act(BN(z)) -> (dz, dgamma, dbeta)."""
    C = z2.shape[1]
    if bnc is None:
        g = K.act_bwd_reduce(dy2, z2, act=act, slope=slope, row_scale=row_scale, rows_per_group=rows_per_group,
                             dy_rows_per_group=dy_rows_per_group, dy_mul=dy_mul)
        return g, None, None
    sums = _zeros((K.REPL, 2, C), z2)
    g = K.act_bwd_reduce(dy2, z2, scale=bnc.scale, shift=bnc.shift, act=act, slope=slope, row_scale=row_scale,
                         rows_per_group=rows_per_group, dy_rows_per_group=dy_rows_per_group, dy_mul=dy_mul,
                         mean=bnc.mean, rstd=bnc.rstd, sums=sums)
    A, B, Cc, dgamma, dbeta = K.bn_bwd_finalize(sums, bnc.count, gamma, bnc.mean, bnc.rstd, bnc.training,
                                                gamma is not None)
    dz = K.affine2(g, z2, A, B, Cc) if bnc.training else K.affine2(g, None, A)
    return dz, dgamma, dbeta


def _split_broadcast_grad(dy5):
    """A gradient that is an expand() of [N,1,1,1,C] (global-average-pool backward) is passed to the
    kernels as a per-clip row instead of being materialised."""
    if dy5.dim() == 5 and dy5.stride(1) == 0 and dy5.stride(2) == 0 and dy5.stride(3) == 0 and dy5.stride(4) == 1:
        N, T, H, W, C = dy5.shape
        rows = dy5[:, 0, 0, 0, :]
        rows = rows if rows.is_contiguous() else rows.contiguous()
        return rows, T * H * W
    return None, 0


# =============================================================================== bottleneck
class _Bottleneck(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x5, w1, wdw, w2, se_w1, se_b1, se_w2, se_b2, gamma, beta, rmean, rvar, drop, cfg):
        N, T, H, W, Cin = x5.shape
        E, Cout = w1.shape[0], w2.shape[0]
        kernel, stride, pad = cfg["kernel"], cfg["stride"], cfg["pad"]
        act, slope, training = cfg["act"], cfg["slope"], cfg["training"]
        has_bn, has_se = cfg["bn"] is not None, se_w1 is not None
        x2 = x5.view(-1, Cin)
        e2 = K.pw_fwd(x2, w1.view(E, Cin))
        e5 = e2.view(N, T, H, W, E)
        training_bn = has_bn and training
        pool, stats = _zeros_many(x5, (N, E) if has_se else (0,), (K.REPL, 2, Cout) if training_bn else (0,))
        pool = pool if has_se else None
        stats = stats if training_bn else None
        d5 = K.dw_fwd(e5, wdw.view(E, -1), kernel, stride, pad, pool=pool)
        To, Ho, Wo = d5.shape[1:4]
        R2 = To * Ho * Wo
        d2 = d5.view(-1, E)
        se_scale = hidden = None
        if has_se:
            Cm = se_w1.shape[0]
            se_scale, hidden = K.se_fc_fwd(pool, 1.0 / R2, se_w1.view(Cm, E), se_b1, se_w2.view(E, Cm), se_b2)
        need_grad = cfg["need_grad"]
        bnc = None
        if has_bn and not training and not need_grad:
            # inference: BatchNorm folds into the projection GEMM's epilogue, z is never materialised
            eps, momentum = cfg["bn"]
            scale, shift, _, _ = _bn_forward(d2, None, 1, gamma, beta, rmean, rvar, eps, momentum, False)
            y2 = K.pw_fwd(d2, w2.view(Cout, E), row_scale=se_scale, rows_per_group=R2, out_scale=scale,
                          out_shift=shift, out_act=act, out_slope=slope)
            return y2.view(N, To, Ho, Wo, Cout)
        z2 = K.pw_fwd(d2, w2.view(Cout, E), row_scale=se_scale, rows_per_group=R2, stats=stats)
        scale = shift = None
        if has_bn:
            eps, momentum = cfg["bn"]
            bnc = _BnCtx()
            bnc.scale, bnc.shift, bnc.mean, bnc.rstd = _bn_forward(z2, stats, z2.shape[0], gamma, beta, rmean, rvar,
                                                                   eps, momentum, training)
            bnc.training, bnc.count = training, z2.shape[0]
            scale, shift = bnc.scale, bnc.shift
        y2 = K.affine_act_fwd(z2, scale, shift, act, slope, row_scale=drop, rows_per_group=R2)
        if need_grad:
            ctx.cfg, ctx.bnc, ctx.dims = cfg, bnc, (N, T, H, W, Cin, E, Cout, To, Ho, Wo)
            ctx.save_for_backward(x5, e5, d5, z2, w1, wdw, w2, se_w1, se_w2, pool, se_scale, hidden, gamma, drop)
        return y2.view(N, To, Ho, Wo, Cout)

    @staticmethod
    def backward(ctx, dy5):
        x5, e5, d5, z2, w1, wdw, w2, se_w1, se_w2, pool, se_scale, hidden, gamma, drop = ctx.saved_tensors
        cfg, bnc = ctx.cfg, ctx.bnc
        N, T, H, W, Cin, E, Cout, To, Ho, Wo = ctx.dims
        kernel, stride, pad = cfg["kernel"], cfg["stride"], cfg["pad"]
        R2 = To * Ho * Wo
        has_se = se_w1 is not None
        rows, rpg = _split_broadcast_grad(dy5)
        if rows is not None:
            dz, dgamma, dbeta = _bn_act_backward(rows, z2, bnc, gamma, cfg["act"], cfg["slope"], drop, R2, rpg, 1.0)
        else:
            dy2 = _contig5(dy5).view(-1, Cout)
            dz, dgamma, dbeta = _bn_act_backward(dy2, z2, bnc, gamma, cfg["act"], cfg["slope"], drop, R2)
        d2 = d5.view(-1, E)
        dw2, dwdw, dw1 = _zeros_many(x5, (Cout, E), (E, wdw[0].numel()), (E, Cin))
        # projection conv: weight gradient sees the SE-scaled input, data gradient is w.r.t. it
        K.pw_bwd_weight(dz, d2, dw2, row_scale=se_scale, rows_per_group=R2)
        dS2 = K.pw_fwd(dz, w2.view(Cout, E), w_transposed=True)
        g_scale = g_shift = dse_w1 = dse_b1 = dse_w2 = dse_b2 = None
        if has_se:
            Cm = se_w1.shape[0]
            dscale = K.rowgroup_reduce(dS2, d2, N)
            g_shift, dse_w1, dse_b1, dse_w2, dse_b2 = K.se_fc_bwd(dscale, pool, 1.0 / R2, se_scale, hidden,
                                                                  se_w1.view(Cm, E), se_w2.view(E, Cm))
            g_scale = se_scale
            dse_w1, dse_w2 = dse_w1.view_as(se_w1), dse_w2.view_as(se_w2)
        dS5 = dS2.view(N, To, Ho, Wo, E)
        K.dw_bwd_weight(dS5, e5, dwdw, kernel, stride, pad, g_scale, g_shift)
        dE5 = K.dw_bwd_data(dS5, wdw.view(E, -1), (T, H, W), kernel, stride, pad, g_scale, g_shift)
        dE2 = dE5.view(-1, E)
        K.pw_bwd_weight(dE2, x5.view(-1, Cin), dw1)
        dx5 = None
        if ctx.needs_input_grad[0]:
            dx5 = K.pw_fwd(dE2, w1.view(E, Cin), w_transposed=True).view(N, T, H, W, Cin)
        return (dx5, dw1.view_as(w1), dwdw.view_as(wdw), dw2.view_as(w2), dse_w1, dse_b1, dse_w2, dse_b2, dgamma,
                dbeta, None, None, None, None)


def bottleneck(x5, w1, wdw, w2, se, bn, drop, *, kernel, stride, pad, act, slope, training):
    """x5 NDHWC.  se = (w1,b1,w2,b2) or None; bn = nn.BatchNorm3d or None; drop = fp32 [N,Cout] mask/(1-p) or None."""
    st = _bn_state(bn)
    gamma, beta, rmean, rvar = (st[0], st[1], st[2], st[3]) if st else (None, None, None, None)
    params = [w1, wdw, w2] + ([p for p in se] if se else []) + ([gamma, beta] if gamma is not None else [])
    need_grad = torch.is_grad_enabled() and (x5.requires_grad or any(p.requires_grad for p in params if p is not None))
    cfg = dict(kernel=tuple(kernel), stride=tuple(stride), pad=tuple(pad), act=act, slope=slope, training=training,
               bn=(st[4], st[5]) if st else None, need_grad=need_grad)
    se_w1, se_b1, se_w2, se_b2 = se if se else (None, None, None, None)
    return _Bottleneck.apply(x5, w1, wdw, w2, se_w1, se_b1, se_w2, se_b2, gamma, beta, rmean, rvar, drop, cfg)


# =============================================================================== stem conv + BN + act
class _StemBnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, rmean, rvar, cfg):
        kernel, stride, pad = cfg["kernel"], cfg["stride"], cfg["pad"]
        training, has_bn = cfg["training"], cfg["bn"] is not None
        Cout = w.shape[0]
        stats = _zeros((K.REPL, 2, Cout), x) if (has_bn and training) else None
        z5 = K.stem_fwd(x, w, bias, kernel, stride, pad, cfg["dtype"], cfg["x_scale"], stats)
        z2 = z5.view(-1, Cout)
        bnc, scale, shift = None, None, None
        if has_bn:
            eps, momentum = cfg["bn"]
            bnc = _BnCtx()
            bnc.scale, bnc.shift, bnc.mean, bnc.rstd = _bn_forward(z2, stats, z2.shape[0], gamma, beta, rmean, rvar,
                                                                   eps, momentum, training)
            bnc.training, bnc.count = training, z2.shape[0]
            scale, shift = bnc.scale, bnc.shift
        y2 = K.affine_act_fwd(z2, scale, shift, cfg["act"], cfg["slope"])
        if cfg["need_grad"]:
            ctx.cfg, ctx.bnc = cfg, bnc
            ctx.save_for_backward(x, z5, w, gamma)
            ctx.has_bias = bias is not None
        return y2.view(z5.shape)

    @staticmethod
    def backward(ctx, dy5):
        x, z5, w, gamma = ctx.saved_tensors
        cfg = ctx.cfg
        Cout = w.shape[0]
        z2 = z5.view(-1, Cout)
        dy2 = _contig5(dy5).view(-1, Cout)
        dz, dgamma, dbeta = _bn_act_backward(dy2, z2, ctx.bnc, gamma, cfg["act"], cfg["slope"], None, 0)
        dw = torch.zeros_like(w)
        dbias = _zeros((Cout,), w) if ctx.has_bias else None
        K.stem_bwd_weight(dz.view(z5.shape), x, dw, dbias, cfg["kernel"], cfg["stride"], cfg["pad"], cfg["x_scale"])
        return None, dw, dbias, dgamma, dbeta, None, None, None


def stem_bn_act(x, conv, bn, *, act, slope, training, dtype, x_scale=1.0):
    """x: logical (N,Cin,T,H,W), any strides, fp32/bf16/uint8.  Returns NDHWC [N,To,Ho,Wo,Cout] in `dtype`."""
    st = _bn_state(bn)
    gamma, beta, rmean, rvar = (st[0], st[1], st[2], st[3]) if st else (None, None, None, None)
    if conv.dilation != (1, 1, 1) or conv.groups != 1 or any(isinstance(p, str) for p in (conv.padding,)):
        raise NotImplementedError("stem conv: dilation/groups/string padding are not supported")
    params = [conv.weight, conv.bias, gamma, beta]
    need_grad = torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in params)
    cfg = dict(kernel=tuple(conv.kernel_size), stride=tuple(conv.stride), pad=tuple(conv.padding), act=act,
               slope=slope, training=training, bn=(st[4], st[5]) if st else None, need_grad=need_grad, dtype=dtype,
               x_scale=x_scale)
    return _StemBnAct.apply(x, conv.weight, conv.bias, gamma, beta, rmean, rvar, cfg)


# =============================================================================== 1x1x1 conv (+bias) [+SE] + BN + act
class _PwBnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x5, w, bias, se_w1, se_b1, se_w2, se_b2, gamma, beta, rmean, rvar, drop, cfg):
        lead = x5.shape[:-1]
        Cin, Cout = x5.shape[-1], w.shape[0]
        N = x5.shape[0]
        x2 = x5.reshape(-1, Cin)
        M = x2.shape[0]
        R = M // N
        training, has_bn, has_se = cfg["training"], cfg["bn"] is not None, se_w1 is not None
        act, slope, need_grad = cfg["act"], cfg["slope"], cfg["need_grad"]
        if has_bn and not training and not need_grad and not has_se:
            eps, momentum = cfg["bn"]
            scale, shift, _, _ = _bn_forward(x2, None, 1, gamma, beta, rmean, rvar, eps, momentum, False)
            if bias is not None:
                shift = torch.addcmul(shift, bias, scale)       # [C] fold of the conv bias (tiny, host-side glue)
            y2 = K.pw_fwd(x2, w.view(Cout, Cin), out_scale=scale, out_shift=shift, out_act=act, out_slope=slope)
            return y2.view(*lead, Cout)
        stats = _zeros((K.REPL, 2, Cout), x5) if (has_bn and training) else None
        z2 = K.pw_fwd(x2, w.view(Cout, Cin), out_shift=bias, stats=None if has_se else stats)
        u2, pool, se_scale, hidden = z2, None, None, None
        if has_se:   # MobileNetSmall3D block4: conv -> SE -> BN -> act (mobilenet.py:244-249)
            Cm = se_w1.shape[0]
            pool = K.rowgroup_reduce(z2, None, N)
            se_scale, hidden = K.se_fc_fwd(pool, 1.0 / R, se_w1.view(Cm, Cout), se_b1, se_w2.view(Cout, Cm), se_b2)
            u2 = K.affine_act_fwd(z2, row_scale=se_scale, rows_per_group=R, stats=stats)
        bnc, scale, shift = None, None, None
        if has_bn:
            eps, momentum = cfg["bn"]
            bnc = _BnCtx()
            bnc.scale, bnc.shift, bnc.mean, bnc.rstd = _bn_forward(u2, stats, M, gamma, beta, rmean, rvar, eps,
                                                                   momentum, training)
            bnc.training, bnc.count = training, M
            scale, shift = bnc.scale, bnc.shift
        y2 = K.affine_act_fwd(u2, scale, shift, act, slope, row_scale=drop, rows_per_group=R if drop is not None else 0)
        if need_grad:
            ctx.cfg, ctx.bnc, ctx.has_bias, ctx.lead = cfg, bnc, bias is not None, lead
            ctx.save_for_backward(x5, z2, u2 if has_se else None, w, se_w1, se_w2, pool, se_scale, hidden, gamma, drop)
        return y2.view(*lead, Cout)

    @staticmethod
    def backward(ctx, dy5):
        x5, z2, u2, w, se_w1, se_w2, pool, se_scale, hidden, gamma, drop = ctx.saved_tensors
        cfg = ctx.cfg
        Cin, Cout = x5.shape[-1], w.shape[0]
        N = x5.shape[0]
        x2 = x5.reshape(-1, Cin)
        M = x2.shape[0]
        R = M // N
        has_se = se_w1 is not None
        bn_in = u2 if has_se else z2
        rows, rpg = _split_broadcast_grad(dy5) if dy5.dim() == 5 else (None, 0)
        drop_rpg = R if drop is not None else 0
        if rows is not None:
            du, dgamma, dbeta = _bn_act_backward(rows, bn_in, ctx.bnc, gamma, cfg["act"], cfg["slope"], drop, drop_rpg,
                                                 rpg, 1.0)
        else:
            dy2 = (dy5 if dy5.is_contiguous() else dy5.contiguous()).view(-1, Cout)
            du, dgamma, dbeta = _bn_act_backward(dy2, bn_in, ctx.bnc, gamma, cfg["act"], cfg["slope"], drop, drop_rpg)
        dse_w1 = dse_b1 = dse_w2 = dse_b2 = None
        dz = du
        if has_se:
            Cm = se_w1.shape[0]
            dscale = K.rowgroup_reduce(du, z2, N)
            g_shift, dse_w1, dse_b1, dse_w2, dse_b2 = K.se_fc_bwd(dscale, pool, 1.0 / R, se_scale, hidden,
                                                                  se_w1.view(Cm, Cout), se_w2.view(Cout, Cm))
            dse_w1, dse_w2 = dse_w1.view_as(se_w1), dse_w2.view_as(se_w2)
            dz = K.affine_act_fwd(du, row_scale=se_scale, row_shift=g_shift, rows_per_group=R)
        dw = _zeros((Cout, Cin), x5)
        K.pw_bwd_weight(dz, x2, dw)
        dbias = None
        if ctx.has_bias:
            st = _zeros((K.REPL, 2, Cout), x5)
            K.chan_stats(dz, st)
            dbias = st[:, 0].sum(0)              # fold the stats replicas ([REPL,2,C] -> [C]; host-side glue)
        dx5 = None
        if ctx.needs_input_grad[0]:
            dx5 = K.pw_fwd(dz, w.view(Cout, Cin), w_transposed=True).view(x5.shape)
        return dx5, dw.view_as(w), dbias, dse_w1, dse_b1, dse_w2, dse_b2, dgamma, dbeta, None, None, None, None


def pw_bn_act(x5, weight, bias, bn, *, act, slope, training, se=None, drop=None):
    """[..., Cin] rows -> [..., Cout]: 1x1x1 conv / Linear (+bias) [-> SE] [-> BatchNorm] -> act [-> channel dropout].
    The leading dim is the clip index (SE / dropout groups)."""
    st = _bn_state(bn)
    gamma, beta, rmean, rvar = (st[0], st[1], st[2], st[3]) if st else (None, None, None, None)
    params = [weight, bias, gamma, beta] + (list(se) if se else [])
    need_grad = torch.is_grad_enabled() and (x5.requires_grad or any(p is not None and p.requires_grad for p in params))
    cfg = dict(act=act, slope=slope, training=training, bn=(st[4], st[5]) if st else None, need_grad=need_grad)
    se_w1, se_b1, se_w2, se_b2 = se if se else (None, None, None, None)
    x5 = x5 if x5.is_contiguous() else x5.contiguous()
    return _PwBnAct.apply(x5, weight, bias, se_w1, se_b1, se_w2, se_b2, gamma, beta, rmean, rvar, drop, cfg)


# =============================================================================== global average pool
class _PoolMean(torch.autograd.Function):
    """[N,T,H,W,C] -> [N,C] mean over (T,H,W).  Backward returns an expand()ed view that the
    producer's backward kernel reads as one row per clip (no [N,T,H,W,C] gradient is written)."""

    @staticmethod
    def forward(ctx, x5):
        N, C = x5.shape[0], x5.shape[-1]
        R = x5[0].numel() // C
        sums = K.rowgroup_reduce(x5.view(-1, C), None, N)
        ctx.shape, ctx.R = x5.shape, R
        return (sums * (1.0 / R)).to(x5.dtype)       # [N,C] cast: host-side glue on N*C values

    @staticmethod
    def backward(ctx, dp):
        N, C = dp.shape
        g = (dp.float() * (1.0 / ctx.R)).to(dp.dtype)
        return g.view(N, 1, 1, 1, C).expand(ctx.shape)


def pool_mean(x5):
    return _PoolMean.apply(_contig5(x5))


# =============================================================================== standalone SE block
class _SEBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x5, w1, b1, w2, b2):
        N, C = x5.shape[0], x5.shape[-1]
        x2 = x5.view(-1, C)
        R = x2.shape[0] // N
        Cm = w1.shape[0]
        pool = K.rowgroup_reduce(x2, None, N)
        scale, hidden = K.se_fc_fwd(pool, 1.0 / R, w1.view(Cm, C), b1, w2.view(C, Cm), b2)
        y2 = K.affine_act_fwd(x2, row_scale=scale, rows_per_group=R)
        ctx.save_for_backward(x5, w1, w2, pool, scale, hidden)
        ctx.has_bias = (b1 is not None, b2 is not None)
        return y2.view(x5.shape)

    @staticmethod
    def backward(ctx, dy5):
        x5, w1, w2, pool, scale, hidden = ctx.saved_tensors
        N, C = x5.shape[0], x5.shape[-1]
        x2 = x5.view(-1, C)
        R = x2.shape[0] // N
        Cm = w1.shape[0]
        dy2 = _contig5(dy5).view(-1, C)
        dscale = K.rowgroup_reduce(dy2, x2, N)
        g_shift, dw1, db1, dw2, db2 = K.se_fc_bwd(dscale, pool, 1.0 / R, scale, hidden, w1.view(Cm, C), w2.view(C, Cm))
        dx2 = K.affine_act_fwd(dy2, row_scale=scale, row_shift=g_shift, rows_per_group=R)
        return (dx2.view(x5.shape), dw1.view_as(w1), db1 if ctx.has_bias[0] else None, dw2.view_as(w2),
                db2 if ctx.has_bias[1] else None)


def se_block(x5, se):
    return _SEBlock.apply(_contig5(x5), *se)
