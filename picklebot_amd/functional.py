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
import functools

import os

import torch

from . import kernels as K


def _prepare_bn_folds(model):
    """Inference forwards: fold every eval-mode BatchNorm of `model` in ONE launch, valid for this forward only."""
    fs_cls = getattr(K, "BnFoldSet", None)
    if fs_cls is None or not isinstance(model, torch.nn.Module):
        return
    bns = model.__dict__.get("_pb_bn_mods")
    if bns is None:
        bns = [m for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        model.__dict__["_pb_bn_mods"] = bns
    todo = [(m.weight, m.bias, m.running_mean, m.running_var, float(m.eps)) for m in bns
            if not m.training and m.running_mean is not None and m.running_mean.is_cuda]
    if not todo:
        return
    fs = model.__dict__.get("_pb_foldset")
    if fs is None:
        fs = model.__dict__["_pb_foldset"] = fs_cls()
    fs.prepare(todo)


def scoped(fn):
    """Decorator for every module forward: the outermost one opens a new weight generation (kernels.weight_scope),
    so that the bf16 weight copies derived inside it and in its backward are never reused after an optimiser step.
    An outermost forward that needs no gradients also folds the module's eval-mode BatchNorms (one launch)."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        scope = getattr(K, "weight_scope", None)
        if scope is None:
            return fn(*args, **kwargs)
        outer = getattr(K._wscope_depth, "d", 0) == 0
        with scope():
            if outer and args and not torch.is_grad_enabled():
                _prepare_bn_folds(args[0])
            return fn(*args, **kwargs)
    return wrapper


def prepare_pointwise_weights(model, convs):
    """One batched bf16 conversion (both orientations) of the given 1x1x1 convolutions' weights; call inside the
    model's (scoped) forward when it computes in bf16 on the GPU."""
    ws_cls = getattr(K, "WeightSet", None)
    if ws_cls is None:
        return
    ws = model.__dict__.get("_pb_weightset")
    if ws is None:
        ws = ws_cls()
        model.__dict__["_pb_weightset"] = ws
    ws.prepare([c.weight.view(c.out_channels, c.in_channels) for c in convs])


def prepare_model_weights(model, x, dt):
    """prepare_pointwise_weights for every 1x1x1 convolution and Linear of `model` (list cached on the module);
    dt = the dtype the forward computes in."""
    if not (x.is_cuda and dt == torch.bfloat16):
        return
    mods = model.__dict__.get("_pb_pw_mods")
    if mods is None:
        mods = [m for m in model.modules()            # (the squeeze-excite and classifier layers carry a bias: not GEMM-kernel work)
                if (isinstance(m, torch.nn.Linear) or (isinstance(m, torch.nn.Conv3d) and m.groups == 1
                                                       and tuple(m.kernel_size) == (1, 1, 1))) and m.bias is None]
        model.__dict__["_pb_pw_mods"] = mods
    ws_cls = getattr(K, "WeightSet", None)
    if ws_cls is None:
        return
    ws = model.__dict__.get("_pb_weightset")
    if ws is None:
        ws = ws_cls()
        model.__dict__["_pb_weightset"] = ws
    ws.prepare([m.weight.view(m.weight.shape[0], -1) for m in mods])


def _zeros(shape, ref):
    return K.zeros(shape, ref.device)


def _zeros_many(ref, *shapes):
    """Several zero-filled fp32 buffers (each 16-byte aligned) from the kernels' zero arena."""
    return [K.zeros(s, ref.device) for s in shapes]


def _contig5(t):
    return t if t.is_contiguous() else t.contiguous()


def _bn_state(bn):
    """(gamma, beta, running_mean, running_var, eps, momentum, num_batches_tracked) from an nn.BatchNorm module or None."""
    if bn is None:
        return None
    if bn.momentum is None:
        raise NotImplementedError("picklebot_amd: BatchNorm with momentum=None (cumulative average) is not supported")
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps), float(bn.momentum),
            getattr(bn, "num_batches_tracked", None))


def _bn_forward(z2, stats, count, gamma, beta, rmean, rvar, eps, momentum, training, nbt=None, infer=False):
    """-> (scale, shift, mean, rstd); updates running stats (and the step counter nbt) in training.

    Inference (eval mode, no gradients needed): the outermost module forward has folded all its eval-mode BatchNorms in one
    launch (`scoped` -> kernels.BnFoldSet) - 29 finalize launches per MoViNet-A2 forward were 8 % of its kernel time.  That
    fold is valid for THIS top-level forward only: a training step replayed from a HIP graph changes running statistics and
    affine parameters without any Python-visible trace, so nothing is kept across forwards."""
    use_running = not training
    if not use_running and rmean is None:
        rmean_, rvar_ = None, None
    else:
        rmean_, rvar_ = rmean, rvar
    C = (stats.shape[-1] if stats is not None else rmean.shape[0])
    if infer and use_running and rmean is not None and rmean.is_cuda:
        lookup = getattr(K, "bn_fold_lookup", None)
        hit = lookup(gamma, beta, rmean, rvar, eps) if lookup is not None else None
        if hit is not None:
            return hit
    return K.bn_finalize(None if use_running else stats, count, gamma, beta, eps, momentum, rmean_, rvar_,
                         use_running, C, z2.device, num_batches_tracked=None if use_running else nbt)


class _BnCtx:
    """What BatchNorm backward needs."""
    __slots__ = ("scale", "shift", "mean", "rstd", "training", "count", "beta")


def _bn_act_backward(dy2, z2, bnc, gamma, act, slope, row_scale, rows_per_group, dy_rows_per_group=0, dy_mul=1.0, frame_map=None):
    """Gradient through [row_scale *] act(BN(z)) -> (dz, dgamma, dbeta).  frame_map (kernels.affine_act_fwd): z2 and the
    returned dz are compact (data frames only), dy2 is the full gradient; needs an activation (the g tensor carries the map)."""
    C = z2.shape[1]
    fmkw = dict(frame_map=frame_map) if frame_map is not None else {}
    if bnc is None:
        g = K.act_bwd_reduce(dy2, z2, act=act, slope=slope, row_scale=row_scale, rows_per_group=rows_per_group,
                             dy_rows_per_group=dy_rows_per_group, dy_mul=dy_mul, **fmkw)
        return g, None, None
    sums = _zeros((K.REPL, 2, C), z2)
    # Without an activation behind the BatchNorm, g = dy * dropout-mask is not materialised: the reduction keeps only
    # the two sums and the final pass applies the mask itself.  (With an activation, recomputing act' in the final
    # pass was measured slower on MI355X than writing g once - 0.79+0.65 ms vs 0.83+0.46 ms per step - so g stays.)
    direct = act == "none" and dy_rows_per_group == 0 and dy_mul == 1.0 and frame_map is None
    g = K.act_bwd_reduce(dy2, z2, scale=bnc.scale, shift=bnc.beta, act=act, slope=slope, row_scale=row_scale,
                         rows_per_group=rows_per_group, dy_rows_per_group=dy_rows_per_group, dy_mul=dy_mul,
                         mean=bnc.mean, rstd=bnc.rstd, sums=sums, want_g=not direct, **fmkw)
    A, B, Cc, dgamma, dbeta = K.bn_bwd_finalize(sums, bnc.count, gamma, bnc.mean, bnc.rstd, bnc.training,
                                                gamma is not None)
    if direct:
        kw = dict(row_scale=row_scale, rows_per_group=rows_per_group) if row_scale is not None else {}
        if act != "none":
            kw.update(act_scale=bnc.scale, act_shift=bnc.beta, act=act, slope=slope)
        if bnc.training:
            dz = K.affine2(dy2, z2, A, B, Cc, mean=bnc.mean, **kw)
        else:
            dz = K.affine2(dy2, z2 if act != "none" else None, A, mean=bnc.mean if act != "none" else None, **kw)
    else:
        dz = K.affine2(g, z2, A, B, Cc, mean=bnc.mean) if bnc.training else K.affine2(g, None, A)
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
        training_bn = has_bn and training
        stats = _zeros((K.REPL, 2, Cout), x5) if training_bn else None
        pool = se_scale = hidden = wbs = None
        # A (1,k,k) depthwise kernel whose scalar stride / padding also apply to T (mobilenet.py:67-75): an output frame reads
        # exactly ONE input frame - or none (temporal padding: exact-zero output frames, T grows by 2(k//2) at stride 1), and
        # with stride 2 half of the input frames are never read.  All EXPANDED-side tensors are therefore kept COMPACT: e, d, dS,
        # dE hold only the Tu frames that carry data (the reference expands every input frame and drags the zero frames
        # through the projection conv); only the block's narrow output z gets its zero frames back (frames_scatter) in
        # front of BatchNorm, whose statistics count them.
        To = Tu = None
        compact = None                                # (t0, tstep, Tu, lead): input frames t0 + j * tstep <-> output frame lead + j
        xs5, dkernel, dstride, dpad = x5, kernel, stride, pad
        if kernel[0] == 1 and hasattr(K, "frames_gather"):
            To_full = (T + 2 * pad[0] - 1) // stride[0] + 1
            used = [to for to in range(To_full) if 0 <= to * stride[0] - pad[0] < T]
            if used and (len(used) < To_full or stride[0] > 1):
                lead, Tu, To = used[0], len(used), To_full
                t0 = used[0] * stride[0] - pad[0]
                compact = (t0, stride[0], Tu, lead)
                if stride[0] > 1 or Tu < T:
                    xs5 = K.frames_gather(x5, t0, stride[0], Tu)
                dstride = (1,) + tuple(stride[1:])
                dpad = (0,) + tuple(pad[1:])
        Ts = xs5.shape[1]
        x2 = xs5.view(-1, Cin)
        e2 = K.pw_fwd(x2, w1.view(E, Cin))
        e5 = e2.view(N, Ts, H, W, E)
        if has_se:
            # squeeze partial sums ride on the depthwise kernel; ONE launch turns them into pool / hidden / scale and the
            # per-clip scaled bf16 projection weights (the squeeze-excite scale is folded into the GEMM's weights)
            d5, pool_part = K.dw_fwd_pool(e5, wdw.view(E, -1), dkernel, dstride, dpad)
        else:
            d5 = K.dw_fwd(e5, wdw.view(E, -1), dkernel, dstride, dpad)
        Td, Ho, Wo = d5.shape[1:4]                    # Td = frames of d (compact: the data frames only)
        if To is None:
            To = Td
        R2, Rd = To * Ho * Wo, Td * Ho * Wo           # rows per clip: of the block's output / of d
        d2 = d5.view(-1, E)

        def expand(t2, fill=None):                    # [N*Rd, Cout] rows of the data frames -> all To frames
            if compact is None or Td == To:
                return t2
            return K.frames_scatter(t2.view(N, Td, Ho, Wo, Cout), To, compact[3], 1, fill).view(-1, Cout)

        if has_se:
            Cm = se_w1.shape[0]
            want_scaled = d2.dtype == torch.bfloat16 and Rd >= 128
            pool, se_scale, hidden, wbs = K.se_fused_fwd(pool_part, 1.0 / R2, se_w1.view(Cm, E), se_b1, se_w2.view(E, Cm), se_b2,
                                                         wproj=w2.view(Cout, E), want_scaled=want_scaled)
        need_grad = cfg["need_grad"]
        bnc = None
        if has_bn and not training and not need_grad:
            # inference: BatchNorm folds into the projection GEMM's epilogue, z is never materialised
            eps, momentum = cfg["bn"]
            scale, shift, _, _ = _bn_forward(d2, None, 1, gamma, beta, rmean, rvar, eps, momentum, False, infer=True)
            y2 = K.pw_fwd(d2, w2.view(Cout, E), row_scale=se_scale, rows_per_group=Rd, out_scale=scale,
                          out_shift=shift, out_act=act, out_slope=slope, scaled_weights=wbs)
            if compact is not None and Td != To:      # the padded frames: z = 0 there, so y = act(shift) per channel
                y2 = expand(y2, K.act_row(shift, act, slope, y2.dtype))
            return y2.view(N, To, Ho, Wo, Cout)
        # z stays COMPACT too (the data frames only): the BatchNorm-apply and its backward passes walk the full output rows
        # and read z as 0 on the padded frames (frame map), so the zero frames are never written or read back
        z2 = K.pw_fwd(d2, w2.view(Cout, E), row_scale=se_scale, rows_per_group=Rd, stats=stats, scaled_weights=wbs)
        fmap = (R2, Rd, compact[3] * Ho * Wo) if (compact is not None and Td != To) else None
        rows_full = N * R2
        scale = shift = None
        if has_bn:
            eps, momentum = cfg["bn"]
            bnc = _BnCtx()
            bnc.scale, bnc.shift, bnc.mean, bnc.rstd = _bn_forward(z2, stats, rows_full, gamma, beta, rmean, rvar,
                                                                   eps, momentum, training, cfg.get("nbt"), infer=not training and not cfg["need_grad"])
            bnc.training, bnc.count, bnc.beta = training, rows_full, beta
            scale, shift = bnc.scale, beta
        fmkw = dict(frame_map=fmap) if fmap is not None else {}
        y2 = K.affine_act_fwd(z2, scale, shift, act, slope, row_scale=drop, rows_per_group=R2,
                              mean=bnc.mean if bnc is not None else None, **fmkw)
        if need_grad:
            ctx.cfg, ctx.bnc, ctx.dims = cfg, bnc, (N, T, H, W, Cin, E, Cout, To, Ho, Wo)
            ctx.compact, ctx.dconv, ctx.fmap = compact, (dkernel, dstride, dpad), fmap
            ctx.save_for_backward(xs5, e5, d5, z2, w1, wdw, w2, se_w1, se_w2, pool, se_scale, hidden, gamma, drop, se_b1, se_b2)
        return y2.view(N, To, Ho, Wo, Cout)

    @staticmethod
    def backward(ctx, dy5):
        x5, e5, d5, z2, w1, wdw, w2, se_w1, se_w2, pool, se_scale, hidden, gamma, drop, se_b1, se_b2 = ctx.saved_tensors
        cfg, bnc = ctx.cfg, ctx.bnc
        N, T, H, W, Cin, E, Cout, To, Ho, Wo = ctx.dims
        kernel, stride, pad = cfg["kernel"], cfg["stride"], cfg["pad"]
        compact = ctx.compact
        Td, Ts = d5.shape[1], x5.shape[1]             # frames of d / of the (gathered) block input
        R2, Rd = To * Ho * Wo, Td * Ho * Wo
        has_se = se_w1 is not None
        # z2 and dz are compact (data frames only): the padded output frames reach no input frame and d is zero there, so their
        # dz rows are never needed - only their share of the BatchNorm sums, which the frame-mapped reduction includes
        rows, rpg = _split_broadcast_grad(dy5)
        if rows is not None:
            dz, dgamma, dbeta = _bn_act_backward(rows, z2, bnc, gamma, cfg["act"], cfg["slope"], drop, R2, rpg, 1.0, frame_map=ctx.fmap)
        else:
            dy2 = _contig5(dy5).view(-1, Cout)
            dz, dgamma, dbeta = _bn_act_backward(dy2, z2, bnc, gamma, cfg["act"], cfg["slope"], drop, R2, frame_map=ctx.fmap)
        d2 = d5.view(-1, E)
        dw2, dwdw, dw1 = _zeros_many(x5, (Cout, E), (E, wdw[0].numel()), (E, Cin))
        g_scale = g_shift = dse_w1 = dse_b1 = dse_w2 = dse_b2 = None
        se_keep = ()
        # The chain (data gradients) runs here; the three weight gradients are collected and issued as one branch at the
        # end (K.weight_grad_branch: their partial tiles are added up by ONE launch; on a side stream when that is safe).
        wgrads = []
        # projection conv: weight gradient sees the SE-scaled input, data gradient is w.r.t. it
        wgrads.append(lambda: K.pw_bwd_weight(dz, d2, dw2, row_scale=se_scale, rows_per_group=Rd))
        dS2 = K.pw_fwd(dz, w2.view(Cout, E), w_transposed=True)
        if has_se:
            Cm = se_w1.shape[0]
            dpart = K.rowgroup_partials(dS2, d2, N)
            # (the FC weight / bias gradients are not on the chain: their launch joins the block's weight-gradient branch)
            g_shift, dse_w1, dse_b1, dse_w2, dse_b2, se_wg = K.se_fused_bwd(dpart, pool, 1.0 / R2, se_scale, hidden,
                                                                            se_w1.view(Cm, E), se_w2.view(E, Cm), defer_weights=True)
            wgrads.append(se_wg)
            se_keep = se_wg.keep
            g_scale = se_scale
            dse_w1, dse_w2 = dse_w1.view_as(se_w1), dse_w2.view_as(se_w2)
            # (fresh views for the biases too: the deferred launch still holds the buffers, and a gradient object somebody else
            # holds is CLONED by autograd - on the calling stream, at once)
            dse_b1, dse_b2 = dse_b1.view_as(se_b1), dse_b2.view_as(se_b2)
        dS5 = dS2.view(N, Td, Ho, Wo, E)
        dkernel, dstride, dpad = ctx.dconv       # (x5 here = the gathered frames when the block skips input frames)
        held = ()
        fuse = K.dw_bwd_fused_ws_bytes(dS5, (Ts, H, W), dkernel, dstride, dpad)
        if fuse:
            # stride-1 (1,k,k) blocks, bf16: data gradient AND weight gradient in one walk over dS (read once, one launch
            # fewer per block).  Measured on the MobileNetLarge3D step (nine such blocks): in line 7.20 -> 7.04 ms; with the
            # weight gradients on the side stream 6.70 either way (the separate weight gradient was hidden beside the next
            # block's chain) - kept on in both schedules: 9 launches and 1.1 GB fewer per step.  The partial weight
            # gradients are added up by the branch's single reduction launch.
            with K.held_reductions() as hr:
                dE5 = K.dw_bwd_fused(dS5, e5, wdw.view(E, -1), dwdw, dkernel, dstride, dpad, g_scale, g_shift)
            held = hr.jobs
        else:
            wgrads.append(lambda: K.dw_bwd_weight(dS5, e5, dwdw, dkernel, dstride, dpad, g_scale, g_shift))
            dE5 = K.dw_bwd_data(dS5, wdw.view(E, -1), (Ts, H, W), dkernel, dstride, dpad, g_scale, g_shift)
        dE2 = dE5.view(-1, E)
        wgrads.append(lambda: K.pw_bwd_weight(dE2, x5.view(-1, Cin), dw1))
        dx5 = None
        if ctx.needs_input_grad[0]:
            dx5 = K.pw_fwd(dE2, w1.view(E, Cin), w_transposed=True).view(N, Ts, H, W, Cin)
            if compact is not None and Ts != T:       # exact zeros in the input frames the block never read
                dx5 = K.frames_scatter(dx5, T, compact[0], compact[1])
        se_params = (se_w1, se_b1, se_w2, se_b2) if has_se else ()
        K.weight_grad_branch(dE2, (w1, wdw, w2) + se_params, wgrads,
                             (dz, d2, dS2, e5, x5, dE2, w1, se_scale, g_scale, g_shift, dw2, dwdw, dw1, se_keep), "bneck", held=held)
        return (dx5, dw1.view_as(w1), dwdw.view_as(wdw), dw2.view_as(w2), dse_w1, dse_b1, dse_w2, dse_b2, dgamma,
                dbeta, None, None, None, None)


def bottleneck(x5, w1, wdw, w2, se, bn, drop, *, kernel, stride, pad, act, slope, training):
    """x5 NDHWC.  se = (w1,b1,w2,b2) or None; bn = nn.BatchNorm3d or None; drop = fp32 [N,Cout] mask/(1-p) or None."""
    st = _bn_state(bn)
    gamma, beta, rmean, rvar = (st[0], st[1], st[2], st[3]) if st else (None, None, None, None)
    params = [w1, wdw, w2] + ([p for p in se] if se else []) + ([gamma, beta] if gamma is not None else [])
    need_grad = torch.is_grad_enabled() and (x5.requires_grad or any(p.requires_grad for p in params if p is not None))
    cfg = dict(kernel=tuple(kernel), stride=tuple(stride), pad=tuple(pad), act=act, slope=slope, training=training,
               bn=(st[4], st[5]) if st else None, nbt=st[6] if st else None, need_grad=need_grad)
    se_w1, se_b1, se_w2, se_b2 = se if se else (None, None, None, None)
    return _Bottleneck.apply(x5, w1, wdw, w2, se_w1, se_b1, se_w2, se_b2, gamma, beta, rmean, rvar, drop, cfg)


# =============================================================================== stem conv + BN + act
class _StemBnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, rmean, rvar, cfg):
        kernel, stride, pad = cfg["kernel"], cfg["stride"], cfg["pad"]
        training, has_bn = cfg["training"], cfg["bn"] is not None
        Cout = w.shape[0]
        stats = _zeros((getattr(K, "STEM_REPL", K.REPL), 2, Cout), x) if (has_bn and training) else None
        z5 = K.stem_fwd(x, w, bias, kernel, stride, pad, cfg["dtype"], cfg["x_scale"], stats)
        z2 = z5.view(-1, Cout)
        bnc, scale, shift = None, None, None
        if has_bn:
            eps, momentum = cfg["bn"]
            bnc = _BnCtx()
            bnc.scale, bnc.shift, bnc.mean, bnc.rstd = _bn_forward(z2, stats, z2.shape[0], gamma, beta, rmean, rvar,
                                                                   eps, momentum, training, cfg.get("nbt"), infer=not training and not cfg["need_grad"])
            bnc.training, bnc.count, bnc.beta = training, z2.shape[0], beta
            scale, shift = bnc.scale, beta
        y2 = K.affine_act_fwd(z2, scale, shift, cfg["act"], cfg["slope"], mean=bnc.mean if bnc is not None else None)
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
        dw = _zeros(w.shape, w)
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
               slope=slope, training=training, bn=(st[4], st[5]) if st else None, nbt=st[6] if st else None,
               need_grad=need_grad, dtype=dtype, x_scale=x_scale)
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
            scale, shift, _, _ = _bn_forward(x2, None, 1, gamma, beta, rmean, rvar, eps, momentum, False, infer=True)
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
                                                                   momentum, training, cfg.get("nbt"), infer=not training and not cfg["need_grad"])
            bnc.training, bnc.count, bnc.beta = training, M, beta
            scale, shift = bnc.scale, beta
        y2 = K.affine_act_fwd(u2, scale, shift, act, slope, row_scale=drop, rows_per_group=R if drop is not None else 0,
                              mean=bnc.mean if bnc is not None else None)
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
        dbias = None
        if ctx.has_bias:
            st = _zeros((K.REPL, 2, Cout), x5)
            K.chan_stats(dz, st)
            dbias = st[:, 0].sum(0)              # fold the stats replicas ([REPL,2,C] -> [C]; host-side glue)
        dx5 = None
        if ctx.needs_input_grad[0]:
            dx5 = K.pw_fwd(dz, w.view(Cout, Cin), w_transposed=True).view(x5.shape)
        K.weight_grad_branch(dz, (w,), [lambda: K.pw_bwd_weight(dz, x2, dw)], (dz, x2, x5, dw), "pw")
        return dx5, dw.view_as(w), dbias, dse_w1, dse_b1, dse_w2, dse_b2, dgamma, dbeta, None, None, None, None


def pw_bn_act(x5, weight, bias, bn, *, act, slope, training, se=None, drop=None):
    """[..., Cin] rows -> [..., Cout]: 1x1x1 conv / Linear (+bias) [-> SE] [-> BatchNorm] -> act [-> channel dropout].
    The leading dim is the clip index (SE / dropout groups)."""
    st = _bn_state(bn)
    gamma, beta, rmean, rvar = (st[0], st[1], st[2], st[3]) if st else (None, None, None, None)
    params = [weight, bias, gamma, beta] + (list(se) if se else [])
    need_grad = torch.is_grad_enabled() and (x5.requires_grad or any(p is not None and p.requires_grad for p in params))
    cfg = dict(act=act, slope=slope, training=training, bn=(st[4], st[5]) if st else None, nbt=st[6] if st else None,
               need_grad=need_grad)
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


class _PooledHead(torch.autograd.Function):
    """AdaptiveAvgPool3d(1) -> Conv3d(C, Cm, 1) -> Hardswish / ReLU -> Conv3d(Cm, Co, 1) (mobilenet.py `classifier`) on the
    squeeze-excite gate's kernels (K.pool_head_*: fp32 weights and sums): 3 launches forward, 2 (+ one cast of the incoming
    gradient) on the data-gradient chain backward, the weight gradients in the block's branch - the generic path (pool_mean +
    two pw_bn_act) ran 8 and ~15 launches of ~5 us each on [N x 1280] tensors.  Backward returns the expand()ed row view of
    _PoolMean (no [N,T,H,W,C] gradient is written)."""

    @staticmethod
    def forward(ctx, x5, w1, b1, w2, b2, act):
        N, C = x5.shape[0], x5.shape[-1]
        x2 = x5.view(-1, C)
        R = x2.shape[0] // N
        Cm, Co = w1.shape[0], w2.shape[0]
        part = K.rowgroup_partials(x2, None, N)
        pool, pre1, logits = K.pool_head_fwd(part, 1.0 / R, w1.view(Cm, C), b1, w2.view(Co, Cm), b2, act, x5.dtype)
        ctx.save_for_backward(pool, pre1, w1, b1, w2, b2)
        ctx.meta = (x5.shape, R, act, x5.dtype)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        pool, pre1, w1, b1, w2, b2 = ctx.saved_tensors
        shape, R, act, dt = ctx.meta
        N, C = pool.shape
        Cm, Co = w1.shape[0], w2.shape[0]
        dl = dlogits.float().contiguous()                  # [N, Co]: host-side glue
        row, work = K.pool_head_bwd(dl, 1.0 / R, pre1, w1.view(Cm, C), w2.view(Co, Cm), act, dt)
        dw1, db1, dw2, db2 = _zeros_many(pool, (Cm, C), (Cm,), (Co, Cm), (Co,))
        K.weight_grad_branch(row, (w1, b1, w2, b2),
                             [lambda: K.pool_head_bwd_w(work, pool, 1.0 / R, pre1, act, dw1, db1, dw2, db2)],
                             (work, pool, pre1, dw1, db1, dw2, db2), "head")
        # (fresh views: autograd adopts a gradient tensor only when nobody else holds the OBJECT - the branch keeps them)
        return (row.view(N, 1, 1, 1, C).expand(shape), dw1.view_as(w1), db1.view_as(b1) if b1 is not None else None,
                dw2.view_as(w2), db2.view_as(b2) if b2 is not None else None, None)


def pooled_head_ok(x5, fc1, fc2, act):
    """the fused head takes: hardswish / ReLU between two 1x1x1 convs with fp32 weights, bf16 or fp32 activations"""
    return (act in ("hardswish", "relu") and x5.dtype in (torch.bfloat16, torch.float32) and fc1.weight.dtype == torch.float32
            and fc2.weight.dtype == torch.float32 and fc1.weight[0].numel() == x5.shape[-1] and fc2.weight[0].numel() == fc1.weight.shape[0]
            and max(x5.shape[-1], fc1.weight.shape[0], fc2.weight.shape[0]) <= 8192 and hasattr(K, "pool_head_fwd")
            and os.environ.get("PB_POOLED_HEAD", "1") != "0")


def pooled_head(x5, fc1, fc2, act):
    """logits [N, classes] of `pool -> fc1 -> act -> fc2` (fc1, fc2: nn.Conv3d / nn.Linear with 1x1x1 kernels)"""
    return _PooledHead.apply(_contig5(x5), fc1.weight, fc1.bias, fc2.weight, fc2.bias, act)


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


# =============================================================================== standalone depthwise conv (+bias)
class _Depthwise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x5, w, bias, cfg):
        C = x5.shape[-1]
        y5 = K.dw_fwd(x5, w.view(C, -1), cfg["kernel"], cfg["stride"], cfg["pad_front"], cfg["pad_back"])
        if bias is not None:
            y5 = K.affine_act_fwd(y5.view(-1, C), shift=bias).view(y5.shape)
        ctx.cfg, ctx.has_bias = cfg, bias is not None
        ctx.save_for_backward(x5, w)
        return y5

    @staticmethod
    def backward(ctx, dy5):
        x5, w = ctx.saved_tensors
        cfg = ctx.cfg
        C = x5.shape[-1]
        dy5 = _contig5(dy5)
        dw = _zeros((C, w[0].numel()), x5)
        dx5 = None
        if ctx.needs_input_grad[0]:
            dx5 = K.dw_bwd_data(dy5, w.view(C, -1), tuple(x5.shape[1:4]), cfg["kernel"], cfg["stride"], cfg["pad_front"])
        dbias = None
        if ctx.has_bias:
            st = _zeros((K.REPL, 2, C), x5)
            K.chan_stats(dy5.view(-1, C), st)
            dbias = st[:, 0].sum(0)
        K.weight_grad_branch(dy5, (w,), [lambda: K.dw_bwd_weight(dy5, x5, dw, cfg["kernel"], cfg["stride"], cfg["pad_front"])],
                             (dy5, x5, dw), "dw")
        return dx5, dw.view_as(w), dbias, None


def depthwise(x5, weight, bias, kernel, stride, pad_front, pad_back=None):
    cfg = dict(kernel=tuple(kernel), stride=tuple(stride), pad_front=tuple(pad_front),
               pad_back=tuple(pad_back if pad_back is not None else pad_front))
    return _Depthwise.apply(_contig5(x5), weight, bias, cfg)


# =============================================================================== dense conv (+bias) [+BN] + act
class _ConvBnAct(torch.autograd.Function):
    """Conv3d (groups=1) as implicit GEMM -> [BatchNorm] -> activation: conv_nxn_bn of mobilevit.py:19-24 and a dense
    CausalConv3d (movinet.py:31).  Backward-data runs the same kernel on dy with the transposed, tap-reversed filter."""

    @staticmethod
    def forward(ctx, x5, w, bias, gamma, beta, rmean, rvar, cfg):
        N, T, H, W, Cin = x5.shape
        Cout = w.shape[0]
        kernel, stride, pf, pb = cfg["kernel"], cfg["stride"], cfg["pad_front"], cfg["pad_back"]
        taps = kernel[0] * kernel[1] * kernel[2]
        out_thw = K.dw_out_shape((T, H, W), kernel, stride, pf, pb)
        training, has_bn = cfg["training"], cfg["bn"] is not None
        wp = K.conv_weight_prep(w.reshape(-1), 0, Cout, Cin, taps)
        stats = _zeros((K.REPL, 2, Cout), x5) if (has_bn and training) else None
        z5 = K.conv3d_fwd(x5, wp, Cout, kernel, stride, pf, out_thw, out_shift=bias, stats=stats)
        z2 = z5.view(-1, Cout)
        bnc, scale, shift = None, None, None
        if has_bn:
            eps, momentum = cfg["bn"]
            bnc = _BnCtx()
            bnc.scale, bnc.shift, bnc.mean, bnc.rstd = _bn_forward(z2, stats, z2.shape[0], gamma, beta, rmean, rvar, eps,
                                                                   momentum, training, cfg.get("nbt"), infer=not training and not cfg["need_grad"])
            bnc.training, bnc.count, bnc.beta = training, z2.shape[0], beta
            scale, shift = bnc.scale, beta
        plain = not has_bn and cfg["act"] == "none"
        y5 = z5 if plain else K.affine_act_fwd(z2, scale, shift, cfg["act"], cfg["slope"],
                                               mean=bnc.mean if bnc is not None else None).view(z5.shape)
        if cfg["need_grad"]:
            ctx.cfg, ctx.bnc, ctx.has_bias, ctx.plain = cfg, bnc, bias is not None, plain
            ctx.save_for_backward(x5, z5, w, gamma)
        return y5

    @staticmethod
    def backward(ctx, dy5):
        x5, z5, w, gamma = ctx.saved_tensors
        cfg = ctx.cfg
        N, T, H, W, Cin = x5.shape
        Cout = w.shape[0]
        kernel, stride, pf = cfg["kernel"], cfg["stride"], cfg["pad_front"]
        taps = kernel[0] * kernel[1] * kernel[2]
        z2 = z5.view(-1, Cout)
        dy2 = _contig5(dy5).view(-1, Cout)
        dgamma = dbeta = None
        if ctx.plain:
            dz = dy2
        else:
            dz, dgamma, dbeta = _bn_act_backward(dy2, z2, ctx.bnc, gamma, cfg["act"], cfg["slope"], None, 0)
        dz5 = dz.view(z5.shape)
        wg = {}

        def weight_gradient():
            wg["dwp"] = K.conv3d_bwd_weight(dz5, x5, kernel, stride, pf)
            wg["dw"] = K.conv_weight_prep(wg["dwp"], 2, Cout, Cin, taps).view_as(w)
        dbias = None
        if ctx.has_bias:
            st = _zeros((K.REPL, 2, Cout), x5)
            K.chan_stats(dz, st)
            dbias = st[:, 0].sum(0)
        dx5 = None
        if ctx.needs_input_grad[0]:
            if tuple(stride) != (1, 1, 1):
                raise NotImplementedError("picklebot_amd: input gradient of a strided dense Conv3d is not implemented")
            wt = K.conv_weight_prep(w.reshape(-1), 1, Cout, Cin, taps)            # [Cin][taps reversed][Cout]
            pad_t = tuple(k - 1 - p for k, p in zip(kernel, pf))
            dx5 = K.conv3d_fwd(dz5, wt, Cin, kernel, (1, 1, 1), pad_t, (T, H, W))
        K.weight_grad_branch(dz5, (w,), [weight_gradient], (dz5, x5, wg), "conv", now=True)      # (the thunk creates its result)
        # (a fresh view: autograd adopts a gradient tensor only when nobody else holds the OBJECT - the branch keeps `wg` -
        # and would otherwise clone it on this stream at once, before the side stream has written it)
        return dx5, wg["dw"].view_as(w), dbias, dgamma, dbeta, None, None, None


def conv_bn_act(x5, conv_weight, conv_bias, bn, *, stride, pad_front, pad_back=None, act="none", slope=0.0, training=False):
    st = _bn_state(bn)
    if st is not None and st[2] is None:
        raise NotImplementedError("picklebot_amd: BatchNorm without running statistics")
    gamma, beta, rmean, rvar = (st[0], st[1], st[2], st[3]) if st else (None, None, None, None)
    params = [conv_weight, conv_bias, gamma, beta]
    need_grad = torch.is_grad_enabled() and (x5.requires_grad or any(p is not None and p.requires_grad for p in params))
    cfg = dict(kernel=tuple(conv_weight.shape[2:]), stride=tuple(stride), pad_front=tuple(pad_front),
               pad_back=tuple(pad_back if pad_back is not None else pad_front), act=act, slope=slope, training=training,
               bn=(st[4], st[5]) if st else None, nbt=st[6] if st else None, need_grad=need_grad)
    return _ConvBnAct.apply(_contig5(x5), conv_weight, conv_bias, gamma, beta, rmean, rvar, cfg)


def conv3d_dense(x5, weight, bias, stride, pad_front, pad_back=None):
    """Plain dense Conv3d.  The implicit-GEMM kernel wants channel counts that are multiples of the 16-byte
    vector; other counts (only the CausalConv3d corner, no reference model) are zero-padded here with torch
    layout ops - the convolution itself still runs in HIP."""
    vec = 8 if x5.dtype == torch.bfloat16 else 4
    cout, cin = weight.shape[0], weight.shape[1]
    pi, po = (-cin) % vec, (-cout) % vec
    if pi or po:
        x5 = torch.nn.functional.pad(x5, (0, pi))
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, 0, 0, pi, 0, po))
        bias = torch.nn.functional.pad(bias, (0, po)) if bias is not None else None
        y5 = conv_bn_act(x5, weight, bias, None, stride=stride, pad_front=pad_front, pad_back=pad_back)
        return y5[..., :cout]
    return conv_bn_act(x5, weight, bias, None, stride=stride, pad_front=pad_front, pad_back=pad_back)


# =============================================================================== MobileViT token blocks
class _Unfold(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x5, patch):
        ctx.thw, ctx.patch = tuple(x5.shape[1:4]), patch
        return K.unfold(x5, patch)

    @staticmethod
    def backward(ctx, dtok):
        return K.fold(dtok if dtok.is_contiguous() else dtok.contiguous(), ctx.thw, ctx.patch), None


class _Fold(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, thw, patch):
        ctx.patch = patch
        return K.fold(tok, thw, patch)

    @staticmethod
    def backward(ctx, dx5):
        return K.unfold(_contig5(dx5), ctx.patch), None, None


def unfold(x5, patch):
    return _Unfold.apply(_contig5(x5), tuple(patch))


def fold(tok, thw, patch):
    return _Fold.apply(tok if tok.is_contiguous() else tok.contiguous(), tuple(thw), tuple(patch))


class _AttnBlock(torch.autograd.Function):
    """x + to_out(attention(to_qkv(LayerNorm(x))))   (mobilevit.py:68-86 + the residual of :102), rows [G*n, D]."""

    @staticmethod
    def forward(ctx, x, w_qkv, w_out, G, n, scale, eps, residual=True):
        D = x.shape[1]
        xn, stat = K.layernorm_fwd(x, eps)
        qkv = K.pw_fwd(xn, w_qkv)
        o, lse = K.attention_fwd(qkv, G, n, scale)
        y = K.pw_fwd(o, w_out, residual=x if residual else None)
        ctx.save_for_backward(x, stat, xn, qkv, o, lse, w_qkv, w_out)
        ctx.meta = (G, n, scale, eps, residual)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stat, xn, qkv, o, lse, w_qkv, w_out = ctx.saved_tensors
        G, n, scale, eps, residual = ctx.meta
        D = x.shape[1]
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dw_out, dw_qkv = _zeros_many(x, tuple(w_out.shape), tuple(w_qkv.shape))
        do = K.pw_fwd(dy, w_out, w_transposed=True)
        dqkv = K.attention_bwd(qkv, o, do, lse, G, n, scale)
        dxn = K.pw_fwd(dqkv, w_qkv, w_transposed=True)
        dx = K.layernorm_bwd(x, dxn, stat, eps)
        if residual:
            K.copy_cols(dy, 0, dx, 0, D, accumulate=True)
        K.weight_grad_branch(dy, (w_qkv, w_out), [lambda: K.pw_bwd_weight(dy, o, dw_out), lambda: K.pw_bwd_weight(dqkv, xn, dw_qkv)],
                             (dy, o, dqkv, xn, dw_out, dw_qkv), "attn")
        return dx, dw_qkv.view_as(w_qkv), dw_out.view_as(w_out), None, None, None, None, None    # (fresh views, see _ConvBnAct)


class _FFBlock(torch.autograd.Function):
    """x + W2 silu(W1 x)   (mobilevit.py:28-40 + the residual of :103)."""

    @staticmethod
    def forward(ctx, x, w1, w2, residual=True):
        D = x.shape[1]
        ctx.residual = residual
        hpre = K.pw_fwd(x, w1)
        # silu in its own pass (12 MB) so that both GEMMs on it are the LDS-DMA kernels: the activation-prologue
        # GEMM measured 146 us against 21 + 5 on the [4096,768]x[768,192] layer
        h = K.affine_act_fwd(hpre, act="silu")
        y = K.pw_fwd(h, w2, residual=x if residual else None)
        ctx.save_for_backward(x, hpre, h, w1, w2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, hpre, h, w1, w2 = ctx.saved_tensors
        D = x.shape[1]
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dw1, dw2 = _zeros_many(x, tuple(w1.shape), tuple(w2.shape))
        dh = K.pw_fwd(dy, w2, w_transposed=True)
        dhpre = K.act_bwd_reduce(dh, hpre, act="silu")
        dx = K.pw_fwd(dhpre, w1, w_transposed=True, residual=dy if ctx.residual else None)
        K.weight_grad_branch(dy, (w1, w2), [lambda: K.pw_bwd_weight(dy, h, dw2), lambda: K.pw_bwd_weight(dhpre, x, dw1)],
                             (dy, h, dhpre, x, dw1, dw2), "ff")
        return dx, dw1.view_as(w1), dw2.view_as(w2), None                                          # (fresh views, see _ConvBnAct)


class _TransformerWide(torch.autograd.Function):
    """Transformer.forward (mobilevit.py:89-104: `x = attn(x) + x; x = ff(x) + x`, depth times) on bf16 token rows [G*n, D]
    with the RUNNING SUM kept in fp32 (round 4).  Every GEMM / attention operand is bf16 exactly as under the reference's
    autocast; what is wider than the reference is the residual stream between the blocks (autocast's `bf16 + bf16` add
    rounds it to bf16 after every block - measured on CUDA autocast, profiles/r04_autocast_dtypes.txt) and, in backward,
    the gradient stream.  The rounding of that stream was the systematic part of this path's bf16 gradient deviation
    towards the input.  One autograd node for the whole stack: the fp32 stream and its bf16 rounding
    travel together between the blocks, LayerNorm reads the fp32 stream directly and the residual add emits the next
    GEMM's bf16 input in the same pass (K.stream_add / K.layernorm_wide_*)."""

    @staticmethod
    def forward(ctx, x, G, n, scale, eps, *weights):
        L = len(weights) // 4
        s32, saved = None, []
        out = None
        for l in range(L):
            w_qkv, w_out, w1, w2 = weights[4 * l:4 * l + 4]
            s_in = s32
            if s32 is None:
                xn, stat = K.layernorm_fwd(x, eps)                    # the stack's input is the bf16 output of conv2
            else:
                xn, stat = K.layernorm_wide_fwd(s32, eps)
            qkv = K.pw_fwd(xn, w_qkv)
            o, lse = K.attention_fwd(qkv, G, n, scale)
            y = K.pw_fwd(o, w_out)
            s32, xb = K.stream_add(x if s32 is None else s32, y)      # attn(x) + x: fp32 stream + the feed-forward's bf16 input
            hpre = K.pw_fwd(xb, w1)
            h = K.affine_act_fwd(hpre, act="silu")
            y2 = K.pw_fwd(h, w2)
            last = l == L - 1
            s32, out = K.stream_add(s32, y2, want32=not last, want_bf16=last)   # ff(x) + x
            saved += [s_in, stat, xn, qkv, o, lse, xb, hpre, h]
        ctx.meta = (G, n, scale, eps, L)
        ctx.save_for_backward(x, *weights, *saved)
        return out

    @staticmethod
    def backward(ctx, dy):
        G, n, scale, eps, L = ctx.meta
        x = ctx.saved_tensors[0]
        weights = ctx.saved_tensors[1:1 + 4 * L]
        saved = ctx.saved_tensors[1 + 4 * L:]
        gb = dy if dy.is_contiguous() else dy.contiguous()            # bf16 rounding of the gradient stream
        g32 = None                                                    # the fp32 gradient stream (None: it IS gb)
        grads = [None] * (4 * L)
        dx = None
        for l in reversed(range(L)):
            w_qkv, w_out, w1, w2 = weights[4 * l:4 * l + 4]
            s_in, stat, xn, qkv, o, lse, xb, hpre, h = saved[9 * l:9 * l + 9]
            dw_qkv, dw_out, dw1, dw2 = _zeros_many(x, tuple(w_qkv.shape), tuple(w_out.shape), tuple(w1.shape), tuple(w2.shape))
            # feed-forward block
            dh = K.pw_fwd(gb, w2, w_transposed=True)
            dhpre = K.act_bwd_reduce(dh, hpre, act="silu")
            dxf = K.pw_fwd(dhpre, w1, w_transposed=True)
            K.weight_grad_branch(gb, (w1, w2), [lambda gb=gb, h=h, dw2=dw2: K.pw_bwd_weight(gb, h, dw2),
                                                lambda dhpre=dhpre, xb=xb, dw1=dw1: K.pw_bwd_weight(dhpre, xb, dw1)],
                                 (gb, h, dhpre, xb, dw1, dw2), "ff")
            g32, gb = K.stream_add(gb if g32 is None else g32, dxf)
            # attention block
            do = K.pw_fwd(gb, w_out, w_transposed=True)
            dqkv = K.attention_bwd(qkv, o, do, lse, G, n, scale)
            dxn = K.pw_fwd(dqkv, w_qkv, w_transposed=True)
            K.weight_grad_branch(gb, (w_qkv, w_out), [lambda gb=gb, o=o, dw_out=dw_out: K.pw_bwd_weight(gb, o, dw_out),
                                                      lambda dqkv=dqkv, xn=xn, dw_qkv=dw_qkv: K.pw_bwd_weight(dqkv, xn, dw_qkv)],
                                 (gb, o, dqkv, xn, dw_out, dw_qkv), "attn")
            if s_in is not None:
                g32, gb = K.layernorm_wide_bwd(s_in, dxn, stat, eps, g32)
            else:
                dxl = K.layernorm_bwd(x, dxn, stat, eps)
                _, dx = K.stream_add(g32, dxl, want32=False, want_bf16=True)
            grads[4 * l:4 * l + 4] = [dw_qkv.view_as(w_qkv), dw_out.view_as(w_out), dw1.view_as(w1), dw2.view_as(w2)]   # fresh views
        return (dx, None, None, None, None) + tuple(grads)


_VIT_WIDE = os.environ.get("PB_VIT_WIDE", "1") != "0"


def transformer_wide_ok(rows):
    """bf16 token rows (autocast) of a TRAINING forward take the wide-stream stack (what it buys is gradient accuracy:
    MobileViT's bf16 gradients from 2-2.4x to 0.8-1.5x of the reference's own deviation); fp32 rows are wide anyway;
    inference (no gradients) keeps the per-block bf16 stream - the fp32 stream's extra passes cost MobileViT-XXS forward at
    config 5's batch 5.7 % (8.34 -> 8.82 ms at 64 clips) for no measurable change of the logits."""
    return _VIT_WIDE and rows.dtype == torch.bfloat16 and rows.numel() % 8 == 0 and torch.is_grad_enabled()


def transformer_wide(rows, layer_weights, G, n, scale, eps):
    """layer_weights: [(w_qkv, w_out, w_ff1, w_ff2), ...] -> bf16 rows after all layers."""
    flat = [w for lw in layer_weights for w in lw]
    return _TransformerWide.apply(rows if rows.is_contiguous() else rows.contiguous(), G, n, scale, eps, *flat)


_fp8_attention = False


class fp8_attention:
    """Context manager: inside it, INFERENCE forwards (torch.no_grad, bf16 activations) run the patch attention with fp8
    (e4m3) MFMA operands - BASELINE config 5.  Training forwards are unaffected (there is no fp8 backward)."""

    def __enter__(self):
        global _fp8_attention
        self.prev, _fp8_attention = _fp8_attention, True
        return self

    def __exit__(self, *exc):
        global _fp8_attention
        _fp8_attention = self.prev
        return False


def attn_block(rows, w_qkv, w_out, G, n, scale, eps, residual=True):
    """residual=True: x + attn(x) (Transformer.forward, mobilevit.py:102); False: attn(x) alone (Attention.forward)."""
    rows = rows if rows.is_contiguous() else rows.contiguous()
    if _fp8_attention and not torch.is_grad_enabled() and rows.dtype == torch.bfloat16 and n <= 1024:
        xn, _ = K.layernorm_fwd(rows, eps)
        o, _ = K.attention_fwd(K.pw_fwd(xn, w_qkv), G, n, scale, fp8=True)
        return K.pw_fwd(o, w_out, residual=rows if residual else None)
    return _AttnBlock.apply(rows, w_qkv, w_out, G, n, scale, eps, residual)


def ff_block(rows, w1, w2, residual=True):
    return _FFBlock.apply(rows if rows.is_contiguous() else rows.contiguous(), w1, w2, residual)


class _LinAttn(torch.autograd.Function):
    """out_proj(relu(v) * sum_i softmax_i(q) k)  with (q | k | v) = qkv_proj(x)   (mobilevitv2.py:28-55), rows [G*n, D]:
    G = batch * pixels-in-patch groups of n patches (the 3-D unfold's token layout)."""

    @staticmethod
    def forward(ctx, x, w_qkv, b_qkv, w_out, b_out, G, n):
        qkv = K.pw_fwd(x, w_qkv, out_shift=b_qkv)
        o, scores, cvec = K.linattn_fwd(qkv, G, n)
        y = K.pw_fwd(o, w_out, out_shift=b_out)
        ctx.save_for_backward(x, qkv, o, scores, cvec, w_qkv, w_out)
        ctx.meta = (G, n, b_qkv is not None, b_out is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, qkv, o, scores, cvec, w_qkv, w_out = ctx.saved_tensors
        G, n, has_bq, has_bo = ctx.meta
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dw_out, dw_qkv = _zeros_many(x, tuple(w_out.shape), tuple(w_qkv.shape))
        K.pw_bwd_weight(dy, o, dw_out)
        do = K.pw_fwd(dy, w_out, w_transposed=True)
        dqkv = K.linattn_bwd(qkv, scores, cvec, do, G, n)
        K.pw_bwd_weight(dqkv, x, dw_qkv)
        db_out = db_qkv = None
        if has_bo:
            db_out = K.chan_stats(dy, _zeros((K.REPL, 2, dy.shape[1]), x))[:, 0].sum(0)
        if has_bq:
            db_qkv = K.chan_stats(dqkv, _zeros((K.REPL, 2, dqkv.shape[1]), x))[:, 0].sum(0)
        dx = K.pw_fwd(dqkv, w_qkv, w_transposed=True) if ctx.needs_input_grad[0] else None
        return dx, dw_qkv, db_qkv, dw_out, db_out, None, None


def linattn_block(rows, w_qkv, b_qkv, w_out, b_out, G, n):
    return _LinAttn.apply(rows if rows.is_contiguous() else rows.contiguous(), w_qkv, b_qkv, w_out, b_out, G, n)


class _Concat2(torch.autograd.Function):
    """torch.cat((a, b), channel) for NDHWC tensors (mobilevit.py:141)."""

    @staticmethod
    def forward(ctx, a5, b5):
        Ca, Cb = a5.shape[-1], b5.shape[-1]
        out = torch.empty(a5.shape[:-1] + (Ca + Cb,), dtype=a5.dtype, device=a5.device)
        K.copy_cols(a5.view(-1, Ca), 0, out.view(-1, Ca + Cb), 0, Ca)
        K.copy_cols(b5.view(-1, Cb), 0, out.view(-1, Ca + Cb), Ca, Cb)
        ctx.split = (Ca, Cb)
        return out

    @staticmethod
    def backward(ctx, dout):
        Ca, Cb = ctx.split
        dout = _contig5(dout)
        da = torch.empty(dout.shape[:-1] + (Ca,), dtype=dout.dtype, device=dout.device)
        db = torch.empty(dout.shape[:-1] + (Cb,), dtype=dout.dtype, device=dout.device)
        K.copy_cols(dout.view(-1, Ca + Cb), 0, da.view(-1, Ca), 0, Ca)
        K.copy_cols(dout.view(-1, Ca + Cb), Ca, db.view(-1, Cb), 0, Cb)
        return da, db


def concat_channels(a5, b5):
    return _Concat2.apply(_contig5(a5), _contig5(b5))
