"""Golden-vector protocol (test infrastructure).

`run_model_case` / `run_block_case` take a *namespace of classes* (the real reference when
called from make_golden.py, the oracle restatement or the HIP-backed modules when called from
tests) and return a flat {key: ndarray} dict, so all three are compared on identical inputs:
formula weights (oracle/formula.py), formula clips, dropout disabled.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import formula

MOBILEVIT_CFG = {  # config/mobilevit_s.json:14-15, config/mobilevit_xxs.json:14-15
    "s": dict(dims=[144, 192, 240], channels=[16, 32, 64, 64, 96, 96, 128, 128, 160, 160, 640]),
    "xs": dict(dims=[96, 120, 144], channels=[16, 32, 48, 48, 64, 64, 80, 80, 96, 96, 384]),
    "xxs": dict(dims=[64, 80, 96], channels=[16, 16, 24, 24, 48, 48, 64, 64, 80, 80, 320]),
}

# name -> (class name, ctor kwargs, eval clip (N,T,H,W), train clip (N,T,H,W))
MODEL_CASES = {
    "small3d": ("MobileNetSmall3D", dict(num_classes=2), (2, 8, 112, 112), (2, 8, 112, 112)),
    "large3d": ("MobileNetLarge3D", dict(num_classes=13), (2, 16, 224, 224), (2, 16, 224, 224)),
    "movinet_a2": ("MoViNetA2", dict(num_classes=13), (1, 50, 172, 172), (2, 10, 96, 96)),
    "mobilevit_s": ("MobileViT", dict(num_classes=13, **MOBILEVIT_CFG["s"]), (1, 16, 256, 256), (1, 16, 128, 128)),
    "mobilevit_xxs": ("MobileViT", dict(num_classes=13, **MOBILEVIT_CFG["xxs"]), (1, 32, 224, 224), (1, 10, 96, 96)),
}

# Gradient fixtures (tests/golden/grad_*.npz), two protocols per model:
#   evalgrad: model.eval() (running-statistics BatchNorm, SURVEY 8a-T "in eval-mode"), the train clip of MODEL_CASES
#   train   : model.train() (batch-statistics BatchNorm, train.py:254-289) on the clip below
# The train clips were chosen by oracle/condition_screen.py (round 3): clip shape and input salt such that, in the REAL
# reference's fp64 run, no activation input lies within 1e-5 of a point where the activation's derivative jumps (ReLU 0,
# hardswish / hardsigmoid +-3) and the gradient moves smoothly under perturbation (kappa <= 2000).  A single element on the
# other side of such a kink changes every upstream gradient by ~1e-2 in these BatchNorm-coupled nets, whatever fp32
# implementation computes it - round 2's MobileNetLarge3D fixture (8x8x112x112, 13 M activations) had 17 elements within
# 1e-5, MoViNet-A2's sat 1 ulp from one.
GRAD_CASES = {   # name -> (N, T, H, W, input salt)
    "small3d": (8, 8, 64, 64, 2),
    "large3d": (4, 8, 64, 64, 30),
    "movinet_a2": (4, 8, 64, 64, 15),
    "mobilevit_s": (8, 8, 64, 64, 22),
    "mobilevit_xxs": (8, 8, 64, 64, 18),
}

# Larger train-mode clips, NOT screened: realistic spatial sizes for the compact-frame / frame-map paths (the
# screened clips above leave the last MobileNet stages with 2x2 maps).  An unscreened clip has activation inputs within
# rounding distance of a derivative jump, so any fp32 implementation - the reference's own included - lands on one side
# or the other by luck: these cases are bounded by the reference's OWN fp32 spread over its arithmetic variants
# (`ref32_spread`, stored per key), not by the flat 1e-3.
GRAD_CASES_LARGE = {   # name -> (N, T, H, W, input salt)
    "large3d": (4, 8, 112, 112, 7),
    "small3d": (8, 8, 112, 112, 7),
}

N_SAMPLE = 8


def disable_dropout(model):
    for m in model.modules():
        if isinstance(m, (nn.Dropout, nn.Dropout2d, nn.Dropout3d)):
            m.p = 0.0
        if hasattr(m, "dropout_p"):
            m.dropout_p = 0.0
    return model


def sample_index(name, numel):
    return (np.abs(formula.unit("idx:" + name, N_SAMPLE, lane=3)) * numel).astype(np.int64) % max(numel, 1)


def _np(t):
    return t.detach().float().cpu().numpy()


def _np64(t):
    return t.detach().double().cpu().numpy()


def build_model(ns, case, device="cpu"):
    cls_name, kwargs, _, _ = MODEL_CASES[case]
    model = getattr(ns, cls_name)(**kwargs)
    formula.fill_state_dict(model)
    disable_dropout(model)
    return model.to(device)


def collect_model_case(ns, case, device="cpu", dtype=torch.float32, do_eval=True, do_train=True, autocast=False):
    """Run the protocol and return RAW tensors (float64, on CPU):
    {"logits_eval", "blockstat/<child>", "logits_train", "loss", "grad/<param>", "buf/<name>"}."""
    cls_name, kwargs, eval_shape, train_shape = MODEL_CASES[case]
    ncls = kwargs["num_classes"]
    model = build_model(ns, case, "cpu")
    if dtype == torch.float64:
        model = model.double()      # fp64 "truth" runs; otherwise parameters stay fp32 (train.py keeps fp32 masters)
    model = model.to(device)
    out = {}
    dev_type = torch.device(device).type

    def fwd(x):
        if autocast:
            with torch.autocast(device_type=dev_type, dtype=torch.bfloat16):
                return model(x)
        return model(x)

    if do_eval:
        model.eval()
        x = formula.clip_features(*eval_shape, dtype=dtype).to(device)
        stats, hooks = {}, []
        for name, child in model.named_children():
            def hook(_m, _i, o, name=name):
                o = o.detach().double()
                stats[name] = torch.tensor([o.mean().item(), o.abs().mean().item(), o.norm().item()], dtype=torch.float64)
            hooks.append(child.register_forward_hook(hook))
        with torch.no_grad():
            out["logits_eval"] = fwd(x).detach().double().cpu()
        for h in hooks:
            h.remove()
        for name, v in stats.items():
            out["blockstat/" + name] = v

    if do_train:
        model.train()
        x = formula.clip_features(*train_shape, dtype=dtype).to(device)
        y = formula.labels(train_shape[0], ncls).to(device)
        model.zero_grad(set_to_none=True)
        logits = fwd(x)
        loss = F.cross_entropy(logits if logits.dtype == torch.float64 else logits.float(), y)
        loss.backward()
        out["logits_train"] = logits.detach().double().cpu()
        out["loss"] = loss.detach().double().cpu().reshape(1)
        for name, p in model.named_parameters():
            if p.grad is not None:
                out["grad/" + name] = p.grad.detach().double().cpu()
        bufs = [(k, v) for k, v in model.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")]
        for k, v in bufs[:2] + bufs[-2:]:
            out["buf/" + k] = v.detach().double().cpu()
    return out


def collect_grad_case(ns, case, mode, device="cpu", dtype=torch.float32, autocast=False, shape=None, salt=1234,
                      contiguous_input=False):
    """mode "evalgrad" | "train" (see GRAD_CASES) -> RAW tensors {"logits", "loss", "grad/<param>"} (float64, CPU).
    shape / salt override the clip (oracle/condition_screen.py); contiguous_input: NCDHW-contiguous instead of the
    channels-last-3d view train.py:106 produces (an arithmetic variant for the screen: other conv kernels)."""
    cls_name, kwargs, _, train_shape = MODEL_CASES[case]
    ncls = kwargs["num_classes"]
    model = build_model(ns, case, "cpu")
    if dtype == torch.float64:
        model = model.double()
    model = model.to(device)
    if mode == "evalgrad":
        model.eval()
        shape = shape or train_shape
    elif mode == "train":
        model.train()
        if shape is None:
            shape, salt = GRAD_CASES[case][:4], GRAD_CASES[case][4]
    else:
        raise ValueError(mode)
    x = formula.clip_features(*shape, dtype=dtype, salt=salt)
    if contiguous_input:
        x = x.contiguous()
    x = x.to(device)
    y = formula.labels(shape[0], ncls).to(device)
    model.zero_grad(set_to_none=True)
    if autocast:
        with torch.autocast(device_type=torch.device(device).type, dtype=torch.bfloat16):
            logits = model(x)
    else:
        logits = model(x)
    loss = F.cross_entropy(logits if logits.dtype == torch.float64 else logits.float(), y)
    loss.backward()
    out = {"logits": logits.detach().double().cpu(), "loss": loss.detach().double().cpu().reshape(1)}
    for name, p in model.named_parameters():
        if p.grad is not None:
            out["grad/" + name] = p.grad.detach().double().cpu()
    return out


def run_grad_case(ns, case, mode, **kw):
    return summarize(collect_grad_case(ns, case, mode, **kw))


N_PROJ = 8


def projection_signs(name, numel):
    """N_PROJ fixed +-1 vectors per parameter (formula-generated)."""
    return np.stack([np.sign(formula.unit("proj:" + name, numel, lane=11 + j)) for j in range(N_PROJ)]).astype(np.float64)


def summarize(raw):
    """Compact a raw result: a gradient tensor becomes its norm, N_SAMPLE sampled elements and
    N_PROJ random-sign projections (which see every element)."""
    out = {}
    for k, v in raw.items():
        v = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        if k.startswith("grad/"):
            name = k[5:]
            flat = v.reshape(-1).astype(np.float64)
            out["gnorm/" + name] = np.array([np.linalg.norm(flat)])
            out["gsamp/" + name] = flat[sample_index(name, flat.size)]
            out["gproj/" + name] = projection_signs(name, flat.size) @ flat
        else:
            out[k] = v.astype(np.float64)
    return out


def deviations(raw, truth):
    """Per-key relative error ||raw - truth|| / ||truth|| over the FULL tensors."""
    out = {}
    for k, t in truth.items():
        if k not in raw:
            continue
        d = float(torch.linalg.vector_norm(raw[k].double() - t.double()))
        n = float(torch.linalg.vector_norm(t.double()))
        out[k] = np.array([d / n if n > 0 else d])
    return out


def run_model_case(ns, case, **kw):
    return summarize(collect_model_case(ns, case, **kw))


# ---------------------------------------------------------------------------- tiny blocks
def _block_cases():
    cases = {}
    for stride in (1, 2):
        for k in (3, 5):
            for se in (False, True):
                cases[f"bneck_s{stride}_k{k}_se{int(se)}"] = (
                    "Bottleneck3D", dict(in_channels=8, out_channels=8, expanded_channels=16, stride=stride,
                                         use_se=se, kernel_size=k), (2, 8, 3, 6, 6))
    cases["bneck_relu"] = ("Bottleneck3D", dict(in_channels=8, out_channels=16, expanded_channels=24, stride=1,
                                                use_se=True, kernel_size=3, nonlinearity="relu"), (2, 8, 2, 5, 7))
    cases["bneck_leaky"] = ("Bottleneck3D", dict(in_channels=8, out_channels=16, expanded_channels=24, stride=2,
                                                 use_se=False, kernel_size=3, nonlinearity="leaky_relu"), (2, 8, 4, 9, 9))
    cases["se8"] = ("SEBlock3D", dict(channels=8), (2, 8, 3, 4, 4))
    cases["movinet_bneck_333"] = ("MoviNetBottleneck", dict(in_channels=8, out_channels=8, expanded_channels=16,
                                                            kernel_size=(3, 3, 3), stride=(1, 1, 1),
                                                            padding=(1, 1, 1)), (2, 8, 5, 6, 6))
    cases["movinet_bneck_155_s2"] = ("MoviNetBottleneck", dict(in_channels=8, out_channels=16, expanded_channels=24,
                                                               kernel_size=(1, 5, 5), stride=(1, 2, 2),
                                                               padding=(0, 2, 2)), (2, 8, 3, 9, 9))
    cases["movinet_bneck_533"] = ("MoviNetBottleneck", dict(in_channels=8, out_channels=8, expanded_channels=16,
                                                            kernel_size=(5, 3, 3), stride=(1, 2, 2),
                                                            padding=(2, 1, 1)), (2, 8, 6, 8, 8))
    cases["causal_333"] = ("CausalConv3d", dict(in_channels=4, out_channels=6, kernel_size=(3, 3, 3),
                                                padding=(0, 1, 1)), (1, 4, 7, 5, 5))
    cases["causal_dw_533"] = ("CausalConv3d", dict(in_channels=8, out_channels=8, kernel_size=(5, 3, 3),
                                                   padding=(0, 1, 1), groups=8, bias=False), (2, 8, 9, 6, 6))
    cases["vitblock"] = ("MobileViTBlock", dict(embed_dim=16, depth=2, channel=8, kernel_size=3,
                                                patch_size=(2, 2, 2), ffw_dim=32), (1, 8, 3, 5, 5))
    cases["attention"] = ("Attention", dict(embed_dim=16, heads=8, dim_head=16), (1, 8, 6, 16))
    cases["feedforward"] = ("FeedForward", dict(embed_dim=16, hidden_dim=32), (1, 8, 6, 16))
    return cases


BLOCK_CASES = _block_cases()
# mobilevitv2.py's separable attention, kept in its own fixture file (blocks_v2.npz): input in the
# reference layout (batch, embed_dim, pixels-in-patch, patches)
V2_CASES = {"linattn16": ("LinearSelfAttention", dict(embed_dim=16), (2, 16, 8, 12)),
            "linattn40": ("LinearSelfAttention", dict(embed_dim=40), (1, 40, 4, 30))}
_ALL_BLOCKS = {**BLOCK_CASES, **V2_CASES}
_ACT = {"relu": nn.ReLU, "leaky_relu": nn.LeakyReLU, "hardswish": nn.Hardswish}


def build_block(ns, case, device="cpu"):
    cls_name, kwargs, _ = _ALL_BLOCKS[case]
    kwargs = dict(kwargs)
    if isinstance(kwargs.get("nonlinearity"), str):
        kwargs["nonlinearity"] = _ACT[kwargs["nonlinearity"]]()
    blk = getattr(ns, cls_name)(**kwargs)
    formula.fill_state_dict(blk)
    return disable_dropout(blk).to(device)


def run_block_case(ns, case, device="cpu", dtype=torch.float32, train=True, autocast=False):
    """Full tensors (as float64 arrays): output, input gradient and every parameter gradient of
    L = sum(out * G) for a formula-generated G."""
    _, _, shape = _ALL_BLOCKS[case]
    blk = build_block(ns, case, "cpu")
    if dtype == torch.float64:
        blk = blk.double()
    blk = blk.to(device)
    blk.train(train)
    x = formula.tensor("x:" + case, shape, 1.0, dtype).to(device)
    if len(shape) == 5:   # physically NDHWC like the reference's training input (train.py:106)
        x = x.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)
    x.requires_grad_(True)
    if autocast:
        with torch.autocast(device_type=torch.device(device).type, dtype=torch.bfloat16):
            y = blk(x)
    else:
        y = blk(x)
    g = formula.tensor("g:" + case, tuple(y.shape), 1.0, torch.float64).to(device)
    (y.double() * g).sum().backward()
    out = {"out": _np64(y), "dx": _np64(x.grad)}
    for name, p in blk.named_parameters():
        if p.grad is not None:
            out["grad/" + name] = _np64(p.grad)
    for k, v in blk.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["buf/" + k] = _np64(v)
    return out


def run_causal_stream_case(ns, case="causal_dw_533", chunk=3, device="cpu", dtype=torch.float32):
    """Streaming known answer: feeding the clip in `chunk`-frame pieces, each
    prefixed by the previous (k_t-1) frames of history, equals the full-clip output."""
    _, kwargs, shape = BLOCK_CASES[case]
    blk = build_block(ns, case, "cpu")
    if dtype == torch.float64:
        blk = blk.double()
    blk = blk.to(device).eval()
    x = formula.tensor("x:" + case, shape, 1.0, dtype).to(device)
    with torch.no_grad():
        full = blk(x)
        kt = kwargs["kernel_size"][0]
        pieces = []
        for t0 in range(0, shape[2], chunk):
            lo = max(0, t0 - (kt - 1))
            y = blk(x[:, :, lo:t0 + chunk])
            pieces.append(y[:, :, (t0 - lo):])
        return {"full": _np64(full), "stream": _np64(torch.cat(pieces, 2))}
