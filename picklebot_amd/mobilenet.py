"""Drop-in MI355X modules for the reference's mobilenet.py 3-D classes.

Same class names, constructor signatures, attribute names and state_dict keys/shapes as
/root/reference/mobilenet.py (SEBlock3D :11-26, Bottleneck3D :47-93, MobileNetLarge3D :133-210,
MobileNetSmall3D :213-278), so train.py's registry (train.py:156-161), DDP, optimizers and
checkpoints work unchanged.  The torch.nn layers held here are parameter containers only: forward
never calls them - every op runs in the hand-written HIP kernels through
picklebot_amd.functional (no eager fallback; a missing library raises).

Tensors keep the reference's convention at the module boundary: logical (N,C,T,H,W), physically
channels-last-3d (what train.py:106 produces); a contiguous NCDHW input is accepted and converted
once.  Compute dtype = the autocast dtype when autocast is on (bf16, train.py:264), else the input's.
"""
import torch
import torch.nn as nn

from . import arch
from . import functional as PF

_ACT_NAMES = {nn.ReLU: "relu", nn.LeakyReLU: "leaky_relu", nn.Hardswish: "hardswish", nn.Hardsigmoid: "hardsigmoid",
              nn.SiLU: "silu", nn.Identity: "none"}
_ACT_CLASSES = {"relu": nn.ReLU, "leaky_relu": nn.LeakyReLU, "hardswish": nn.Hardswish}


def act_of(module):
    """(kernel activation name, negative slope) for an nn activation module; loud on anything else."""
    for cls, name in _ACT_NAMES.items():
        if type(module) is cls:
            return name, float(getattr(module, "negative_slope", 0.0))
    raise NotImplementedError(f"picklebot_amd: no HIP kernel for nonlinearity {module!r}")


def compute_dtype(x):
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda")
        if dt != torch.bfloat16:
            raise NotImplementedError(f"picklebot_amd: autocast dtype {dt} not supported (bf16 only, train.py:264)")
        return dt
    if x.dtype in (torch.float32, torch.bfloat16):
        return x.dtype
    if x.dtype == torch.uint8:
        return torch.float32
    raise TypeError(f"picklebot_amd: unsupported input dtype {x.dtype}")


def to_ndhwc(x, dtype=None):
    """logical (N,C,T,H,W) -> contiguous [N,T,H,W,C] (a free view for channels-last-3d inputs)."""
    if x.dim() != 5:
        raise ValueError(f"expected a (N,C,T,H,W) tensor, got shape {tuple(x.shape)}")
    x5 = x.permute(0, 2, 3, 4, 1)
    if dtype is not None and x5.dtype != dtype:
        x5 = x5.to(dtype)
    return x5 if x5.is_contiguous() else x5.contiguous()


def from_ndhwc(y5):
    return y5.permute(0, 4, 1, 2, 3)


def channel_dropout_mask(p, training, n, c, device):
    """Dropout3d (mobilenet.py:82): whole channels of a clip are zeroed with prob p, survivors scaled 1/(1-p)."""
    if not training or p <= 0.0:
        return None
    if p >= 1.0:
        return PF.K.zeros((n, c), device)
    return _mask_pool.take(float(p), n * c, torch.device(device)).view(n, c)


class _MaskPool:
    """Dropout3d keep-masks (Bernoulli(1-p) / (1-p)) drawn a million at a time: a training step needs 15
    masks of n*c <= 5k elements; drawing each one separately cost 3 tiny launches per block."""
    SIZE = 1 << 20

    CAPTURE_SIZE = 1 << 16           # masks drawn per (forward, p) while a HIP graph is being captured

    def __init__(self):
        self._pools = {}             # (device, p) -> [flat mask tensor, offset]
        self._cap = {}               # (device, p) -> [forward generation, flat mask tensor, offset]

    def take(self, p, numel, device):
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing and numel <= self.CAPTURE_SIZE // 4:
            # under HIP-graph capture a pooled slice would replay the same mask for ever: the masks of ONE forward are
            # drawn in-graph by one Bernoulli launch (each replay redraws them), not three tiny launches per block
            gen = getattr(PF.K, "_wgen", 0)
            ent = self._cap.get((device, p))
            if ent is None or ent[0] != gen or ent[2] + numel > self.CAPTURE_SIZE:
                # (bernoulli_(float): one launch - torch.full + torch.bernoulli(tensor) were two)
                ent = [gen, torch.empty((self.CAPTURE_SIZE,), dtype=torch.float32, device=device).bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p)), 0]
                self._cap[(device, p)] = ent
            off = ent[2]
            ent[2] = off + (numel + 3) // 4 * 4
            return ent[1][off:off + numel]
        if numel > self.SIZE // 4 or capturing:
            return torch.empty((numel,), dtype=torch.float32, device=device).bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))
        ent = self._pools.get((device, p))
        if ent is None or ent[1] + numel > self.SIZE:
            ent = [torch.empty((self.SIZE,), dtype=torch.float32, device=device).bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p)), 0]
            self._pools[(device, p)] = ent
        off = ent[1]
        ent[1] = off + (numel + 3) // 4 * 4          # keep every mask 16-byte aligned
        return ent[0][off:off + numel]


_mask_pool = _MaskPool()


def se_params(se_block):
    conv_a, conv_b = se_block.se[1], se_block.se[3]
    return conv_a.weight, conv_a.bias, conv_b.weight, conv_b.bias


class SEBlock3D(nn.Module):
    """x * hardsigmoid(W2 relu(W1 mean_THW(x) + b1) + b2)."""

    def __init__(self, channels):
        super().__init__()
        mid = channels // 4
        self.se = nn.Sequential(nn.AdaptiveAvgPool3d(1), nn.Conv3d(channels, mid, kernel_size=1), nn.ReLU(inplace=True),
                                nn.Conv3d(mid, channels, kernel_size=1), nn.Hardsigmoid())

    @torch.compiler.disable
    @PF.scoped
    def forward(self, x):
        x5 = to_ndhwc(x, compute_dtype(x))
        return from_ndhwc(PF.se_block(x5, se_params(self)))


class Bottleneck3D(nn.Module):
    def __init__(self, in_channels, out_channels, expanded_channels, stride=1, use_se=False, kernel_size=3,
                 nonlinearity=None, batchnorm=True, dropout=0, bias=False):
        super().__init__()
        if bias:
            raise NotImplementedError("picklebot_amd.Bottleneck3D: bias=True is never used by the reference models")
        e, k = expanded_channels, kernel_size
        self.pointwise_conv1 = nn.Conv3d(in_channels, e, kernel_size=1, bias=False)
        self.depthwise_conv = nn.Conv3d(e, e, groups=e, kernel_size=(1, k, k), stride=stride, padding=k // 2, bias=False)
        self.squeeze_excite = SEBlock3D(e) if use_se else None
        self.pointwise_conv2 = nn.Conv3d(e, out_channels, kernel_size=1, bias=False)
        self.batchnorm = nn.BatchNorm3d(out_channels) if batchnorm else None
        self.nonlinearity = nn.Hardswish() if nonlinearity is None else nonlinearity
        self.dropout = nn.Dropout3d(p=dropout)

    @torch.compiler.disable
    @PF.scoped
    def forward(self, x):
        x5 = to_ndhwc(x, compute_dtype(x))
        dwc = self.depthwise_conv
        act, slope = act_of(self.nonlinearity)
        bn = self.batchnorm
        training = bn.training if bn is not None else self.training
        drop = channel_dropout_mask(self.dropout.p, self.dropout.training, x5.shape[0],
                                    self.pointwise_conv2.out_channels, x5.device)
        se = se_params(self.squeeze_excite) if self.squeeze_excite is not None else None
        y5 = PF.bottleneck(x5, self.pointwise_conv1.weight, dwc.weight, self.pointwise_conv2.weight, se, bn, drop,
                           kernel=dwc.kernel_size, stride=dwc.stride, pad=dwc.padding, act=act, slope=slope,
                           training=training)
        return from_ndhwc(y5)


def _bneck_from_row(r):
    return Bottleneck3D(r.cin, r.cout, r.exp, stride=r.stride, use_se=r.se, kernel_size=r.k,
                        nonlinearity=_ACT_CLASSES[r.act](), dropout=r.drop)


class _MobileNet3D(nn.Module):
    TABLE = None

    def __init__(self, num_classes=2):
        super().__init__()
        tb = self.TABLE
        self.num_classes = num_classes
        self.block1 = nn.Sequential(nn.Conv3d(3, tb["stem_out"], kernel_size=3, stride=2, padding=1),
                                    nn.BatchNorm3d(tb["stem_out"]), nn.Hardswish())
        self._body = []
        for name, rows in tb["blocks"].items():
            setattr(self, name, nn.Sequential(*[_bneck_from_row(r) for r in rows]))
            self._body.append(name)
        cin, cout = tb["last_conv"]
        tail = [nn.Conv3d(cin, cout, kernel_size=1)]
        if tb["last_se"]:
            tail.append(SEBlock3D(channels=cout))
        tail += [nn.BatchNorm3d(cout), nn.Hardswish()]
        self._last = tb["last_name"]
        setattr(self, self._last, nn.Sequential(*tail))
        self.classifier = nn.Sequential(nn.AdaptiveAvgPool3d((1, 1, 1)), nn.Conv3d(cout, tb["head_hidden"], kernel_size=1),
                                        nn.Hardswish(), nn.Conv3d(tb["head_hidden"], num_classes, kernel_size=1))

    @torch.compiler.disable
    @PF.scoped
    def forward(self, x):
        dt = compute_dtype(x)
        if x.is_cuda and dt == torch.bfloat16:
            # bf16 copies (both orientations) of every pointwise weight for this forward/backward: one launch
            convs = self.__dict__.get("_pb_pw_convs")
            if convs is None:
                convs = [c for m in self.modules() if isinstance(m, Bottleneck3D) for c in (m.pointwise_conv1, m.pointwise_conv2)]
                convs.append(getattr(self, self._last)[0])
                self.__dict__["_pb_pw_convs"] = convs
            PF.prepare_pointwise_weights(self, convs)
        conv, bn, act_m = self.block1
        act, slope = act_of(act_m)
        y5 = PF.stem_bn_act(x, conv, bn, act=act, slope=slope, training=bn.training, dtype=dt,
                            x_scale=(1.0 / 255.0) if x.dtype == torch.uint8 else 1.0)
        y = from_ndhwc(y5)
        for name in self._body:
            y = getattr(self, name)(y)
        tail = getattr(self, self._last)
        conv, bn, act_m = tail[0], tail[-2], tail[-1]
        se = se_params(tail[1]) if len(tail) == 4 else None
        act, slope = act_of(act_m)
        y5 = PF.pw_bn_act(to_ndhwc(y), conv.weight, conv.bias, bn, act=act, slope=slope, training=bn.training, se=se)
        fc1, act_m, fc2 = self.classifier[1], self.classifier[2], self.classifier[3]
        act, slope = act_of(act_m)
        if PF.pooled_head_ok(y5, fc1, fc2, act):
            return PF.pooled_head(y5, fc1, fc2, act).view(x.shape[0], self.num_classes)
        pooled = PF.pool_mean(y5)
        h = PF.pw_bn_act(pooled, fc1.weight, fc1.bias, None, act=act, slope=slope, training=self.training)
        logits = PF.pw_bn_act(h, fc2.weight, fc2.bias, None, act="none", slope=0.0, training=self.training)
        return logits.view(x.shape[0], self.num_classes)


class MobileNetLarge3D(_MobileNet3D):
    TABLE = arch.MOBILENET_LARGE_3D

    def initialize_weights(self):
        # reference mobilenet.py:203-210 never matches a module (hasattr(conv, "nonlinearity") is
        # False), so it changes nothing; kept for API parity (train.py:177).
        return None


class MobileNetSmall3D(_MobileNet3D):
    TABLE = arch.MOBILENET_SMALL_3D

    def initialize_weights(self):
        # reference mobilenet.py:268-278: only the BatchNorm branch has an effect.
        for m in self.modules():
            if isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
