"""Drop-in MI355X modules for the reference's movinet.py (CausalConv3d :7-39, MoviNetBottleneck :43-77,
MoViNetA2 :80-179): same class names, constructor signatures, attribute names and state_dict keys.
As in picklebot_amd.mobilenet the torch.nn layers are parameter containers; compute runs in the HIP
kernels (the bottleneck is the same fused Function as Bottleneck3D, with a (kt,k,k) depthwise filter,
stride (1,s,s), explicit symmetric padding and squeeze-excite always on)."""
import torch
import torch.nn as nn
from torch.nn import init

from . import arch
from . import functional as PF
from .mobilenet import SEBlock3D, act_of, channel_dropout_mask, compute_dtype, from_ndhwc, se_params, to_ndhwc


def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


class CausalConv3d(nn.Module):
    """Left-pad T by (k_t - 1) frames of `stream_buffer` (a constant; 0 by default), then a plain Conv3d
    (movinet.py:7-39).  `conv3d` carries the parameters (key `conv3d.weight`).  Depthwise (groups == channels) and
    dense filters both run in HIP.

    Streaming (the reference's "stream buffer" idea, which its own forward only ever fills with a constant): the causal
    left pad is exactly the (k_t - 1) frames that preceded the chunk, so `forward_stream(chunk, state)` evaluates a clip
    chunk by chunk and equals `forward(clip)` on the whole clip (SURVEY 8a-5 known answer).  The depthwise (kt,3,3)
    kernels read the history from the caller-owned state buffer directly (pb_dwconv3d_fwd_hist); other shapes
    concatenate history and chunk.  No reference MODEL uses CausalConv3d, and whole-model streaming is undefined there
    (squeeze-excite pools over all of T), so this stays a layer-level API."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, stream_buffer=None, **kwargs):
        super().__init__()
        ks = kernel_size if isinstance(kernel_size, tuple) else (kernel_size,) * 3
        dl = dilation if isinstance(dilation, tuple) else (dilation,) * 3
        self.stream_buffer = 0 if stream_buffer is None else stream_buffer
        kt = ks[0]
        left, right = ((kt - 2) // 2, kt // 2) if kt % 2 == 0 else ((kt - 1) // 2, (kt - 1) // 2)
        self.p_left_causal, self.p_right_causal = left + right, 0
        self.conv3d = nn.Conv3d(in_channels, out_channels, ks, stride=stride, dilation=dl, **kwargs)

    def _check(self):
        c = self.conv3d
        if c.dilation != (1, 1, 1) or isinstance(c.padding, str):
            raise NotImplementedError("picklebot_amd.CausalConv3d: dilation / string padding")
        if not (c.groups == c.in_channels == c.out_channels or c.groups == 1):
            raise NotImplementedError("picklebot_amd.CausalConv3d: grouped (non-depthwise) convolution")
        return c

    def _conv(self, x5, hist5):
        """x5 NDHWC chunk; hist5: None (zero causal pad) or the p_left_causal frames that precede it."""
        c = self.conv3d
        pt, ph, pw = c.padding
        pad_front, pad_back = (pt + self.p_left_causal, ph, pw), (pt, ph, pw)
        depthwise = c.groups == c.in_channels == c.out_channels
        if hist5 is not None and self.p_left_causal > 0:
            if depthwise and c.bias is None and not torch.is_grad_enabled():
                y5 = PF.K.dw_fwd_hist(x5, hist5, c.weight.view(c.out_channels, -1), c.kernel_size, c.stride, pad_front, pad_back)
                if y5 is not None:
                    return y5
            # any other shape (and training): history and chunk as one tensor, the conv's own zero padding in front of it
            x5 = torch.cat([hist5, x5], 1)
            pad_front = (pt, ph, pw)
        if depthwise:
            return PF.depthwise(x5, c.weight, c.bias, c.kernel_size, c.stride, pad_front, pad_back)
        return PF.conv3d_dense(x5, c.weight, c.bias, c.stride, pad_front, pad_back)

    @torch.compiler.disable
    @PF.scoped
    def forward(self, x):
        self._check()
        x5 = to_ndhwc(x, compute_dtype(x))
        hist5 = None
        if self.stream_buffer != 0 and self.p_left_causal > 0:        # F.pad(..., "constant", stream_buffer), movinet.py:36
            N, _, H, W, C = x5.shape
            hist5 = torch.full((N, self.p_left_causal, H, W, C), float(self.stream_buffer), dtype=x5.dtype, device=x5.device)
        return from_ndhwc(self._conv(x5, hist5))

    @torch.compiler.disable
    @PF.scoped
    def forward_stream(self, x, state=None):
        """One chunk of a clip: x (N,C,t,H,W) -> (y (N,C',t',H',W'), state).  `state` is the history buffer
        [N, k_t-1, H, W, C] (NDHWC, caller-owned; None for the first chunk = the `stream_buffer` constant) and is
        advanced in place to the last k_t-1 frames seen.  Temporal stride 1 only (every reference use)."""
        c = self._check()
        if c.stride[0] != 1:
            raise NotImplementedError("picklebot_amd.CausalConv3d.forward_stream: temporal stride must be 1")
        x5 = to_ndhwc(x, compute_dtype(x))
        nh = self.p_left_causal
        if nh == 0:
            return from_ndhwc(self._conv(x5, None)), state
        N, t, H, W, C = x5.shape
        if state is None:
            state = torch.full((N, nh, H, W, C), float(self.stream_buffer), dtype=x5.dtype, device=x5.device)
        elif tuple(state.shape) != (N, nh, H, W, C) or state.dtype != x5.dtype:
            raise ValueError(f"CausalConv3d.forward_stream: state {tuple(state.shape)} {state.dtype} does not fit the chunk")
        y5 = self._conv(x5, state)
        with torch.no_grad():                                         # advance the history (memory plumbing)
            if t >= nh:
                state.copy_(x5[:, t - nh:])
            else:
                state.copy_(torch.cat([state[:, t:], x5], 1))
        return from_ndhwc(y5), state


class MoviNetBottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, expanded_channels, kernel_size, stride=1, use_se=True,
                 batchnorm=True, nonlinearity=None, bias=False, dropout=0, padding=None, dilation=1):
        super().__init__()
        if bias or dilation != 1:
            raise NotImplementedError("picklebot_amd.MoviNetBottleneck: bias/dilation are never used by the reference models")
        e = expanded_channels
        self.expand = nn.Conv3d(in_channels, e, kernel_size=1, bias=False)
        if padding is None:      # movinet.py:49
            padding = ((kernel_size[0] - 1, kernel_size[1] // 2, kernel_size[2] // 2)
                       if isinstance(kernel_size, tuple) else kernel_size // 2)
        self.conv = nn.Conv3d(e, e, kernel_size=kernel_size, stride=stride, padding=padding, groups=e, bias=False)
        self.squeeze_excite = SEBlock3D(e) if use_se else None
        self.project = nn.Conv3d(e, out_channels, kernel_size=1, bias=False)
        self.batchnorm = nn.BatchNorm3d(out_channels) if batchnorm else None
        self.nonlinearity = nn.Hardswish() if nonlinearity is None else nonlinearity
        self.dropout = nn.Dropout3d(p=dropout)        # constructed, never applied (movinet.py:67 vs :69-77)

    @torch.compiler.disable
    @PF.scoped
    def forward(self, x):
        x5 = to_ndhwc(x, compute_dtype(x))
        dwc = self.conv
        act, slope = act_of(self.nonlinearity)
        bn = self.batchnorm
        se = se_params(self.squeeze_excite) if self.squeeze_excite is not None else None
        y5 = PF.bottleneck(x5, self.expand.weight, dwc.weight, self.project.weight, se, bn, None,
                           kernel=dwc.kernel_size, stride=dwc.stride, pad=dwc.padding, act=act, slope=slope,
                           training=bn.training if bn is not None else self.training)
        return from_ndhwc(y5)


class MoViNetA2(nn.Module):
    def __init__(self, num_classes=2, buffer_size=2):
        super().__init__()
        tb = arch.MOVINET_A2
        self.num_classes, self.buffer_size = num_classes, buffer_size
        self.block1 = nn.Sequential(nn.Conv3d(3, tb["stem_out"], kernel_size=(1, 3, 3), stride=(1, 2, 2),
                                              padding=(0, 1, 1), bias=False), nn.BatchNorm3d(tb["stem_out"]), nn.Hardswish())
        self._body = []
        for name, rows in tb["blocks"].items():
            setattr(self, name, nn.Sequential(*[
                MoviNetBottleneck(r.cin, r.cout, r.exp, kernel_size=r.kernel, stride=r.stride, padding=r.padding)
                for r in rows]))
            self._body.append(name)
        cin, cout = tb["last_conv"]
        self.conv = nn.Sequential(nn.Conv3d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm3d(cout), nn.Hardswish(),
                                  nn.Dropout3d(0.2))
        hid = tb["head_hidden"]
        self.classifier = nn.Sequential(nn.AdaptiveAvgPool3d((1, 1, 1)), nn.Flatten(), nn.Linear(cout, hid),
                                        nn.BatchNorm1d(hid), nn.Hardswish(), nn.Dropout(0.2), nn.Linear(hid, num_classes))

    @torch.compiler.disable
    @PF.scoped
    def forward(self, x):
        dt = compute_dtype(x)
        PF.prepare_model_weights(self, x, dt)
        conv, bn, act_m = self.block1
        act, slope = act_of(act_m)
        y5 = PF.stem_bn_act(x, conv, bn, act=act, slope=slope, training=bn.training, dtype=dt,
                            x_scale=(1.0 / 255.0) if x.dtype == torch.uint8 else 1.0)
        y = from_ndhwc(y5)
        for name in self._body:
            y = getattr(self, name)(y)
        conv, bn, act_m, drop_m = self.conv
        act, slope = act_of(act_m)
        y5 = to_ndhwc(y)
        drop = channel_dropout_mask(drop_m.p, drop_m.training, y5.shape[0], conv.out_channels, y5.device)
        y5 = PF.pw_bn_act(y5, conv.weight, None, bn, act=act, slope=slope, training=bn.training, drop=drop)
        pooled = PF.pool_mean(y5)
        fc1, bn1, act_m, drop_m, fc2 = (self.classifier[i] for i in (2, 3, 4, 5, 6))
        act, slope = act_of(act_m)
        drop = channel_dropout_mask(drop_m.p, drop_m.training, pooled.shape[0], fc1.out_features, pooled.device)
        h = PF.pw_bn_act(pooled, fc1.weight, fc1.bias, bn1, act=act, slope=slope, training=bn1.training, drop=drop)
        return PF.pw_bn_act(h, fc2.weight, fc2.bias, None, act="none", slope=0.0, training=self.training)

    def initialize_weights(self):      # movinet.py:167-179
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
                if m.bias is not None:
                    init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm3d):
                init.constant_(m.weight, 1)
                init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    init.constant_(m.bias, 0)
