"""GPU: the HIP-backed modules against the reference goldens (tests/golden, values = the real
reference's fp64 run; oracle/make_golden.py) under the same protocol the oracle is pinned with.

Tolerances (BASELINE north_star / SURVEY 8c): fp32 1e-3 relative, bf16 1e-2 relative, each
widened only to a multiple of the REFERENCE'S OWN deviation in that precision where that is
larger (conftest.check_model_result) - the reference's bf16-autocast gradients are themselves
~20 % off fp64 on these BatchNorm-heavy nets, so "1e-2 on every gradient" is not a bar the
reference meets against itself.  Also: size-independent properties at larger batch."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

import picklebot_amd
from conftest import check_model_result, compare_full, golden_mode, rel_err
from oracle import formula, protocol

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

BLOCKS = [c for c, (cls, _, _) in protocol.BLOCK_CASES.items() if hasattr(picklebot_amd, cls)]
MODELS = [c for c, (cls, _, _, _) in protocol.MODEL_CASES.items() if hasattr(picklebot_amd, cls)]


def _native_loaded():
    from picklebot_amd import _C
    assert _C.load() is not None


@pytest.mark.parametrize("case", BLOCKS)
def test_block_fp32_vs_golden(case, golden):
    _native_loaded()
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_block_case(picklebot_amd, case, device=DEV)
    bad, worst = compare_full(res, golden("blocks"), case + "/", 1e-3, 1e-3)
    assert not bad, f"{bad[:5]} worst {worst}"


@pytest.mark.parametrize("case", BLOCKS)
def test_block_bf16_autocast_vs_golden(case, golden):
    """fp32 input + autocast(bf16), exactly how the golden's refbf16_err (the reference's own bf16 CPU
    deviation) was produced.  Tiny blocks (8-24 channels, <=300 rows, BN over few rows) are noisy in
    bf16 (one draw of rounding noise each): allowed error = max(2e-2, 8 x the reference's own
    deviation) per tensor; the model-level tests below carry the north_star's bf16 bar."""
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_block_case(picklebot_amd, case, device=DEV, autocast=True)
    bad, worst = compare_full(res, golden("blocks"), case + "/", 2e-2, 2e-2, dev="refbf16_err", factor=8.0)
    assert not bad, f"{bad[:5]} worst {worst}"


@pytest.mark.parametrize("case", list(protocol.V2_CASES))
def test_v2_linear_attention_vs_golden(case, golden):
    """mobilevitv2.LinearSelfAttention on the HIP path: output, input gradient and all four parameter
    gradients against the real module's fp64 values; fp32 flat 1e-3, bf16-autocast within max(2e-2, 8 x the reference's
    own bf16 deviation) like the other tiny blocks."""
    _native_loaded()
    res = protocol.run_block_case(picklebot_amd, case, device=DEV)
    bad, worst = compare_full(res, golden("blocks_v2"), case + "/", 1e-3, 1e-3)
    assert not bad, f"{bad[:5]} worst {worst}"
    res = protocol.run_block_case(picklebot_amd, case, device=DEV, autocast=True)
    bad, worst = compare_full(res, golden("blocks_v2"), case + "/", 2e-2, 2e-2, dev="refbf16_err", factor=8.0)
    assert not bad, f"{bad[:5]} worst {worst}"


@pytest.mark.parametrize("case", MODELS)
def test_model_fp32_vs_golden(case, golden):
    """Every key (forward quantities, BN buffers, and per-parameter gradient norms / samples / full-tensor projections)
    within its bound on ONE run: no retry, no share of keys allowed over (the library has no floating-point atomics any
    more, so a run is reproducible bit for bit - test_two_runs_are_bit_identical).  The bound is the north_star's 1e-3,
    widened only where the REFERENCE'S OWN fp32 run is further than that from its fp64 run (train-mode BatchNorm over a
    2-clip batch; the well-conditioned gradient fixtures of test_gradients_* carry the flat 1e-3)."""
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_model_case(picklebot_amd, case, device=DEV)
    bad, worst = check_model_result(res, golden("model_" + case), 1e-3, 1e-3, dev="ref32_err", factor=4.0)
    assert not bad, f"{len(bad)} keys off, first {bad[:5]}, worst {worst}"


@pytest.mark.parametrize("case", MODELS)
def test_two_runs_are_bit_identical(case):
    """No floating-point atomics anywhere on the path (statistics, squeeze-excite pools and weight gradients go through
    per-workgroup slots added in a fixed order): the same binary on the same inputs gives the same bits - logits, loss
    and every parameter gradient, fp32 and bf16-autocast."""
    for kw in (dict(), dict(autocast=True)):
        with contextlib.redirect_stdout(io.StringIO()):
            a = protocol.collect_model_case(picklebot_amd, case, device=DEV, do_eval=False, **kw)
            b = protocol.collect_model_case(picklebot_amd, case, device=DEV, do_eval=False, **kw)
        assert a.keys() == b.keys()
        diff = [k for k in a if not torch.equal(a[k], b[k])]
        assert not diff, f"{len(diff)} of {len(a)} tensors differ between two runs ({kw}): {diff[:8]}"


@pytest.mark.parametrize("case", MODELS)
def test_model_bf16_autocast_vs_golden(case, golden):
    """The reference's own precision mode (train.py:264): fp32 parameters, bf16 activations."""
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_model_case(picklebot_amd, case, device=DEV, dtype=torch.bfloat16, autocast=True)
    res = {k: v for k, v in res.items() if not k.startswith("blockstat/")}
    gold = golden("model_" + case)
    # forward quantities (logits, loss, BN buffers): per key within max(1e-2, 2 x the reference's own
    # bf16 deviation from fp64)
    fwd = {k: v for k, v in res.items() if k.split("/")[0] not in ("gnorm", "gsamp", "gproj")}
    bad, worst = check_model_result(fwd, gold, 1e-2, 1e-2, dev="refbf16_err", factor=2.0)
    assert not bad, f"{len(bad)} keys off, first {bad[:5]}, worst {worst}"
    # (gradients at bf16: per tensor on the gradient fixtures, test_gradients_bf16_per_tensor - the N = 2 train-mode clips of
    #  these files carry the forward bar only)


GRADS = [c for c in protocol.GRAD_CASES if c in MODELS]


@pytest.mark.parametrize("mode", ["evalgrad", "train"])
@pytest.mark.parametrize("case", GRADS)
def test_gradients_fp32_flat_1e3(case, mode, golden):
    """Logits, loss and every per-parameter gradient (norm, 8 full-tensor projections, sampled elements) within a FLAT 1e-3
    of the reference's fp64 values, eval mode (running-statistics BatchNorm) AND train mode (batch statistics, train.py:254),
    all five models, nothing widened by anybody's deviation.  The train clips passed oracle/condition_screen.py: no
    activation input within 1e-5 of a derivative jump, smooth conditioning kappa <= 2000 (the verdict is stored in the
    fixture: train/meta/*)."""
    gold = golden_mode(golden("grad_" + case), mode)
    if mode == "train":
        assert int(gold["meta/kink_within_1e-5"][0]) == 0 and float(gold["meta/kappa_median_max"][1]) <= 2000
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_grad_case(picklebot_amd, case, mode, device=DEV)
    bad, worst = check_model_result(res, gold, 1e-3, 1e-3, dev="ref32_err", factor=0.0, samp_factor=4.0)
    assert not bad, f"{len(bad)} keys off, first {bad[:5]}, worst {worst}"


@pytest.mark.parametrize("case", [c for c in protocol.GRAD_CASES_LARGE if c in MODELS])
def test_train_gradients_fp32_at_112x112_within_the_references_own_spread(case, golden):
    """the screened train clips are 64x64 (the last MobileNet stages run 5x5 kernels on 2x2 maps), so this adds one
    train-mode case per MobileNet at 8x112x112 - realistic frame sizes for the compact-frame / frame-map / frames_scatter
    paths.  The clip is NOT screened: it has activation inputs within rounding distance of a derivative jump (16 / 2 within
    1e-5, stored in train/meta), so every fp32 implementation lands on one side or the other by luck - the reference's own
    four arithmetic variants are 1.2e-2 (median over the tensors) apart from fp64 on it.  Bound per key: 3 x the reference's
    own fp32 spread (`ref32_spread` = its worst variant), floor 1e-3; forward quantities likewise."""
    gold = golden_mode(golden("grad112_" + case), "train")
    shape, salt = protocol.GRAD_CASES_LARGE[case][:4], protocol.GRAD_CASES_LARGE[case][4]
    assert tuple(int(v) for v in gold["meta/clip"]) == tuple(shape) + (salt,)
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_grad_case(picklebot_amd, case, "train", device=DEV, shape=shape, salt=salt)
    bad, worst = check_model_result(res, gold, 1e-3, 1e-3, dev="ref32_spread", factor=3.0, samp_factor=4.0)
    print(f"{case} 8x112x112 train fp32: worst {worst}")
    assert not bad, f"{len(bad)} keys off, first {bad[:5]}, worst {worst}"


@pytest.mark.parametrize("case", ["movinet_a2", "large3d"])
def test_train_gradients_do_not_depend_on_the_summation_order(case, golden, monkeypatch):
    """with the squeeze-excite slot sums re-associated (PB_SE_ASSOC=1: tail into the last chain, chains added
    left to right - an equally valid fp32 order) the train-mode gradients must still sit within the flat 1e-3: the fixture
    pins the function, not one rounding order of it."""
    gold = golden_mode(golden("grad_" + case), "train")
    results = {}
    for assoc in ("0", "1"):
        monkeypatch.setenv("PB_SE_ASSOC", assoc)
        with contextlib.redirect_stdout(io.StringIO()):
            results[assoc] = protocol.collect_grad_case(picklebot_amd, case, "train", device=DEV)
        bad, worst = check_model_result(protocol.summarize(results[assoc]), gold, 1e-3, 1e-3, dev="ref32_err", factor=0.0, samp_factor=4.0)
        assert not bad, f"assoc {assoc}: {len(bad)} keys off, first {bad[:5]}, worst {worst}"
    moved = [k for k in results["0"] if not torch.equal(results["0"][k], results["1"][k])]
    assert moved, "the two associations gave identical bits: the knob is not reaching the kernels"


BF16_GRAD_FACTOR = 3.0        # bar for EVERY tensor (round 4; rounds 2-3: 5x / "90 % within 3x, all within 6x")


@pytest.mark.parametrize("mode", ["evalgrad", "train"])
@pytest.mark.parametrize("case", GRADS)
def test_gradients_bf16_per_tensor(case, mode, golden):
    """bf16 autocast (train.py:264), PER TENSOR (norm and every full-tensor projection of every parameter gradient), against
    the reference's OWN bf16-autocast deviation from fp64 for that tensor (mean over four arithmetic variants of the real
    reference, oracle/make_golden.py; floor = the north_star's 1e-2): forward quantities, gradient norms and gradient
    projections within 3x - ALL of them, all ten cases.

    3 = 2.2 x 1.35: the reference's deviation is its full-tensor error norm (exact); ours is measured through 8 random-sign
    projections, i.e. sqrt(chi2_8 / 8) x our true error - above 2.2 with probability 6e-6 per tensor at EQUAL accuracy;
    1.35 is this path's measured excess on single tensors (it keeps one more intermediate, g = dy * act', in bf16).

    Round 3 needed a quantile form (90 % within 3x, cap 6x) for the two MobileViT eval-mode cases, whose input-side tensors
    sat at 2-2.4x as a stage mean (single tensors up to 3.96x).  The systematic part of that was the rounding of the
    Transformer's residual stream to bf16 after every block in forward AND backward; the stream is fp32 now
    (functional._TransformerWide - WIDER than the reference, whose autocast adds are bf16 + bf16 -> bf16,
    profiles/r04_autocast_dtypes.txt).  Measured (profiles/r04_grad_err_bf16.txt), ours / reference over 86 tensors:
    MobileViT-S eval median 0.77, max 2.29 (was 1.21 / 3.96); XXS eval 0.90 / 1.85 (was 1.34 / 3.04); train mode
    0.82-0.86 / 1.34-1.76; the MobileNet / MoViNet cases are unchanged (max 1.2-2.7)."""
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_grad_case(picklebot_amd, case, mode, device=DEV, autocast=True)
    res = {k: v for k, v in res.items() if not k.startswith("gsamp/")}
    gold = golden_mode(golden("grad_" + case), mode)
    bad, worst = check_model_result(res, gold, 1e-2, 1e-2, dev="refbf16_mean", factor=BF16_GRAD_FACTOR)
    assert not bad, f"{len(bad)} keys beyond {BF16_GRAD_FACTOR}x, first {bad[:5]}, worst {worst}"


def _large(n, dtype=torch.bfloat16):
    torch.manual_seed(1234)
    m = protocol.build_model(picklebot_amd, "large3d", DEV)
    x = formula.clip_features(n, 16, 224, 224, dtype=dtype).to(DEV)
    return m, x


def test_eval_batch_independence_at_scale():
    """Eval-mode logits of a clip do not depend on its batch neighbours or position - bit for bit: every reduction on the
    path is per clip or goes through fixed-order slots, and the grids are sized per clip."""
    m, x = _large(8)
    m.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        full = m(x).float().cpu()
        one = torch.cat([m(x[i:i + 1]) for i in (0, 5, 7)]).float().cpu()
    assert torch.equal(one, full[[0, 5, 7]]), rel_err(one.numpy(), full[[0, 5, 7]].numpy())


def test_gradients_linear_in_loss_scale():
    """GradScaler multiplies the loss by 2^16 (train.py:224,269): gradients must scale with it
    (power-of-two scaling commutes with every rounding on the path).  Checked in fp32 on the
    well-conditioned MobileNetSmall3D (run-to-run atomic-order noise ~1e-5 there)."""
    m = protocol.build_model(picklebot_amd, "small3d", DEV).train()
    x = formula.clip_features(2, 8, 112, 112).to(DEV)
    y = formula.labels(2, 2).to(DEV)
    grads = []
    for scale in (1.0, 65536.0):
        m.zero_grad(set_to_none=True)
        (torch.nn.functional.cross_entropy(m(x), y) * scale).backward()
        grads.append({n: p.grad.detach().double().cpu() / scale for n, p in m.named_parameters()})
    gmax = max(float(g.norm()) for g in grads[0].values())
    worst = max(float((grads[1][n] - grads[0][n]).norm()) / max(float(grads[0][n].norm()), 1e-3 * gmax) for n in grads[0])
    assert worst <= 1e-3, worst


def test_state_dict_roundtrip_and_compile_flag():
    m = picklebot_amd.MobileNetSmall3D(2).to(DEV)
    sd = {"_orig_mod." + k: v for k, v in m.state_dict().items()}          # train.py:38-44 strips this prefix
    sd = {k.replace("_orig_mod.", ""): v for k, v in sd.items()}
    m2 = picklebot_amd.MobileNetSmall3D(2).to(DEV)
    m2.load_state_dict(sd)
    x = formula.clip_features(1, 8, 112, 112).to(DEV)
    m.eval(); m2.eval()
    with torch.no_grad():   # (no floating-point atomics anywhere: the same weights give the same bits)
        assert torch.equal(m(x), m2(x))


def test_uint8_clip_input_matches_float_path():
    """the stem reads the dataloader's uint8 clip directly (x/255 fused)."""
    m = protocol.build_model(picklebot_amd, "small3d", DEV).eval()
    u8 = formula.clip_u8(2, 8, 112, 112).to(DEV)
    with torch.no_grad():
        a = m(u8.permute(0, -1, 1, 2, 3))
        b = m(u8.permute(0, -1, 1, 2, 3).float() / 255)
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


def test_weight_copies_follow_a_fused_optimizer_step():
    """The LDS-DMA GEMMs read bf16 copies of the fp32 weights.  torch's fused optimisers update parameters without
    bumping their version counters, so the copies are re-derived per top-level forward (kernels.weight_scope): after
    an optimiser step the model must compute with the NEW weights, exactly like a fresh model loaded from them."""
    import copy
    torch.manual_seed(3)
    model = picklebot_amd.MobileNetSmall3D(5).to(DEV).train()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-2, fused=True)
    x = torch.randn(2, 3, 8, 64, 64, device=DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y1 = model(x).float()
    y1.square().mean().backward()
    opt.step()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y1b = model(x).float()                       # second training forward: must already see the new weights
    assert (y1b - y1).abs().max() > 1e-3, "the step did not change the output: stale weights?"
    fresh = picklebot_amd.MobileNetSmall3D(5).to(DEV)
    fresh.load_state_dict(copy.deepcopy(model.state_dict()))
    model.eval(); fresh.eval()                       # (dropout and batch statistics would differ between the two)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y2 = model(x).float()
        y2_fresh = fresh(x).float()
    assert torch.allclose(y2, y2_fresh, rtol=2e-2, atol=2e-2), (y2 - y2_fresh).abs().max()


def test_inference_batchnorm_fold_is_per_forward_and_never_stale():
    """Inference folds BatchNorm into (scale, shift) with ONE batched launch per top-level forward (29 finalize launches per
    MoViNet-A2 forward were 8 % of its kernel time).  Nothing is kept across forwards, so the fold follows everything that can
    change it: in-place buffer edits, a training forward (our kernel updates the running statistics without bumping versions),
    a fused optimiser step (no version bump either) - and a training step REPLAYED from a HIP graph, which never enters Python."""
    import copy
    torch.manual_seed(5)
    model = picklebot_amd.MobileNetSmall3D(5)
    formula.fill_state_dict(model)                   # (weights and running statistics under which the signal survives)
    model = model.to(DEV).eval()
    x = torch.randn(2, 3, 8, 64, 64, device=DEV)

    def fresh_output():
        fresh = picklebot_amd.MobileNetSmall3D(5).to(DEV).eval()
        fresh.load_state_dict(copy.deepcopy(model.state_dict()))
        with torch.no_grad():
            return fresh(x)

    from picklebot_amd import kernels as K
    K.start_profile()
    with torch.no_grad():
        y1 = model(x)
    prof = K.stop_profile()
    assert prof["bn_fold_batch"]["launches"] == 1 and "bn_finalize" not in prof      # one batched fold, no per-layer launches
    with torch.no_grad():
        y2 = model(x)
    assert torch.equal(y1, y2) and torch.equal(y1, fresh_output()) and not torch.equal(y1[0], y1[1])
    with torch.no_grad():
        bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm3d)]
        bns[3].running_mean.add_(0.25)
        y3 = model(x)
    assert not torch.equal(y3, y1) and torch.equal(y3, fresh_output())
    opt = torch.optim.AdamW(model.parameters(), lr=5e-2, fused=True)
    model.train()
    model(x).square().mean().backward()              # updates the running statistics
    opt.step()                                       # changes gamma / beta without bumping their versions
    model.eval()
    with torch.no_grad():
        y4 = model(x)
    assert not torch.equal(y4, y3) and torch.equal(y4, fresh_output())
    model(x).sum().backward()                        # eval mode, grad enabled (frozen-BN fine-tuning)
    opt.step()
    with torch.no_grad():
        y5 = model(x)
    assert torch.equal(y5, fresh_output())


def test_eval_after_hipgraph_replayed_training_steps_sees_current_statistics():
    """capture a training step, eval, replay K steps (no Python runs: running statistics, gamma/beta and all
    weights change through raw pointers), eval again - the second eval must equal a fresh model loaded from the CURRENT
    state_dict (train.py:123-153 interleaves estimate_loss with training exactly like this)."""
    import copy
    from picklebot_amd import kernels as K
    torch.manual_seed(11)
    model = picklebot_amd.MobileNetSmall3D(5).to(DEV)
    opt = picklebot_amd.FlatAdamW(model.parameters(), lr=1e-2, weight_decay=1e-3)
    x = torch.randn(4, 3, 8, 64, 64, device=DEV)
    y = torch.randint(0, 5, (4,), device=DEV)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(model(x).float(), y)
        loss.backward()
        opt.step()
        return loss

    def evaluate(m):
        m.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(x).float()
        return out

    def fresh_eval():
        fresh = picklebot_amd.MobileNetSmall3D(5).to(DEV)
        fresh.load_state_dict(copy.deepcopy(model.state_dict()))
        return evaluate(fresh)

    model.train()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    K.reset_zero_arena()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        step()
    K.reset_zero_arena()
    graph.replay()
    e1 = evaluate(model)
    assert torch.equal(e1, fresh_eval())
    for _ in range(5):                               # the module stays in eval mode: the replay does not care
        graph.replay()
    e2 = evaluate(model)
    assert not torch.equal(e2, e1)
    assert torch.equal(e2, fresh_eval())


@pytest.mark.parametrize("name", ["large", "movinet", "vit"])
def test_training_reduces_loss_on_a_fixed_batch(name):
    """Beyond one step: 25 AdamW steps on ONE synthetic batch (bf16 autocast, the reference's training mode) must
    bring the loss down - a wrong-signed or mis-scaled gradient somewhere would pass no per-step tolerance for long."""
    torch.manual_seed(0)
    if name == "large":
        model, shape = picklebot_amd.MobileNetLarge3D(num_classes=13), (4, 16, 112, 112, 3)
    elif name == "movinet":
        model, shape = picklebot_amd.MoViNetA2(num_classes=13), (2, 16, 96, 96, 3)
    else:
        with contextlib.redirect_stdout(io.StringIO()):
            model = picklebot_amd.MobileViT(num_classes=13, dims=[64, 80, 96], channels=[16, 16, 24, 24, 48, 48, 64, 64, 80, 80, 320])
        shape = (2, 16, 128, 128, 3)
    model = model.to(DEV).train()
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout3d)):
            m.p = 0.0
    x = torch.rand(shape, device=DEV).permute(0, 4, 1, 2, 3).to(torch.bfloat16)
    y = torch.randint(0, 13, (shape[0],), device=DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
    losses = []
    for _ in range(25):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16), contextlib.redirect_stdout(io.StringIO()):
            loss = torch.nn.functional.cross_entropy(model(x).float(), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(v == v for v in losses), losses
    assert min(losses[-5:]) < 0.6 * losses[0], losses


# ------------------------------------------------------------------------------ CausalConv3d streaming
@pytest.mark.parametrize("case", ["causal_dw_533", "causal_333"])
def test_causal_streaming_known_answer_on_hip(case, golden):
    """chunked evaluation with (k_t - 1) frames of carried history == the full clip, on the HIP path, against the
    reference's own values (blocks.npz: stream_<case>/full)."""
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_causal_stream_case(picklebot_amd, case, device=DEV)
    g = golden("blocks")
    assert rel_err(res["full"], g[f"stream_{case}/full"]) <= 1e-5
    assert rel_err(res["stream"], g[f"stream_{case}/full"]) <= 1e-5


def _torch_causal(w, b, x, value, pad):
    """The reference's CausalConv3d.forward in stock torch ops on the CPU (movinet.py:33-39)."""
    kt = w.shape[2]
    xp = torch.nn.functional.pad(x, (0, 0, 0, 0, kt - 1, 0), "constant", value)
    return torch.nn.functional.conv3d(xp, w, b, padding=pad, groups=x.shape[1] if w.shape[1] == 1 else 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,kt,thw,value", [(40, 3, (11, 22, 22), 0.0), (40, 3, (11, 22, 22), 0.5), (96, 5, (9, 11, 11), -0.25),
                                            (16, 5, (7, 6, 6), 0.0)])
def test_causal_depthwise_stateful_stream(C, kt, thw, value, dtype):
    """The stateful entry (pb_dwconv3d_fwd_hist: (kt,3,3) depthwise kernels reading a caller-owned history buffer):
    forward(clip) with a stream_buffer pad VALUE == F.pad(value) + conv3d, and forward_stream over chunks of 1, 2, 4
    frames == forward(clip) (fp32: 1e-5; bf16: the same kernel arithmetic per output, so also tight)."""
    torch.manual_seed(C + kt)
    layer = picklebot_amd.CausalConv3d(C, C, (kt, 3, 3), padding=(0, 1, 1), groups=C, bias=False, stream_buffer=value).to(DEV).eval()
    x = torch.randn(2, C, *thw)
    want = _torch_causal(layer.conv3d.weight.detach().cpu(), None, x, value, (0, 1, 1))
    xd = x.to(DEV).to(dtype)
    tol = 1e-5 if dtype == torch.float32 else 1.5e-2
    with torch.no_grad():
        full = layer(xd)
        assert rel_err(full.float().cpu().numpy(), want.numpy()) <= tol
        for chunk in (1, 2, 4):
            state, pieces = None, []
            for t0 in range(0, thw[0], chunk):
                y, state = layer.forward_stream(xd[:, :, t0:t0 + chunk], state)
                pieces.append(y)
            got = torch.cat(pieces, 2)
            assert rel_err(got.float().cpu().numpy(), full.float().cpu().numpy()) <= (1e-6 if dtype == torch.float32 else 1e-6), chunk


def test_causal_dense_stream_buffer_value():
    """Dense CausalConv3d with a non-zero stream_buffer value (constant history frames, concatenated in front)."""
    torch.manual_seed(2)
    layer = picklebot_amd.CausalConv3d(8, 16, (3, 3, 3), padding=(0, 1, 1), stream_buffer=0.3).to(DEV).eval()
    x = torch.randn(1, 8, 6, 9, 9)
    want = _torch_causal(layer.conv3d.weight.detach().cpu(), layer.conv3d.bias.detach().cpu(), x, 0.3, (0, 1, 1))
    with torch.no_grad():
        got = layer(x.to(DEV))
        assert rel_err(got.cpu().numpy(), want.numpy()) <= 1e-4
        state, pieces = None, []
        for t0 in range(0, 6, 2):
            y, state = layer.forward_stream(x.to(DEV)[:, :, t0:t0 + 2], state)
            pieces.append(y)
        assert rel_err(torch.cat(pieces, 2).cpu().numpy(), got.cpu().numpy()) <= 1e-5


# ------------------------------------------------------------------------------ callers around the path
def test_variable_length_batch_from_custom_collate():
    """dataloader.py:10-23: clips of different lengths are zero-padded on T to the longest.  The padded uint8 batch goes
    through the model as is (the stem folds /255) and must match the CPU oracle on the same padded batch."""
    from oracle import torch_ref
    g = torch.Generator().manual_seed(4)
    batch = [(torch.randint(0, 256, (t, 64, 64, 3), dtype=torch.uint8, generator=g), lab) for t, lab in ((8, 1), (5, 0), (3, 1))]
    clips, labels = picklebot_amd.custom_collate(batch)
    assert tuple(clips.shape) == (3, 8, 64, 64, 3) and int(clips[2, 3:].sum()) == 0
    m = protocol.build_model(picklebot_amd, "small3d", DEV).eval()
    ref = protocol.build_model(torch_ref, "small3d", "cpu").eval()
    feats, lab = picklebot_amd.extract_features_labels((clips, labels), DEV)
    with torch.no_grad():
        got = m(feats).float().cpu()
        want = ref(clips.permute(0, -1, 1, 2, 3).float() / 255)
    assert rel_err(got.numpy(), want.numpy()) <= 1e-3
    # a clip's logits do not depend on how much zero tail its neighbours forced on the batch... only on its own padding
    with torch.no_grad():
        alone = m(picklebot_amd.extract_features_labels((clips[1:2], labels[1:2]), DEV)[0]).float().cpu()
    assert rel_err(alone.numpy(), got[1:2].numpy()) <= 1e-5


def test_estimate_loss_and_prefetcher_on_gpu():
    """train.py:123-153 on the device: DevicePrefetcher (pinned H2D on a copy stream) feeding estimate_loss; same numbers
    as a plain per-batch loop."""
    torch.manual_seed(0)
    m = protocol.build_model(picklebot_amd, "small3d", DEV)
    g = torch.Generator().manual_seed(9)
    loader = [(torch.randint(0, 256, (b, 8, 64, 64, 3), dtype=torch.uint8, generator=g), torch.randint(0, 2, (b,), generator=g))
              for b in (3, 2, 4)]
    crit = torch.nn.CrossEntropyLoss()
    loss, acc = picklebot_amd.estimate_loss(m, picklebot_amd.DevicePrefetcher(loader, DEV), crit, DEV)
    tot, cor, n = 0.0, 0, 0
    m.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for clips, labels in loader:
            out = m(clips.to(DEV).permute(0, -1, 1, 2, 3))
            tot += float(crit(out, labels.to(DEV)))
            cor += int((out.argmax(1).cpu() == labels).sum())
            n += labels.numel()
    assert abs(loss - tot / len(loader)) <= 1e-6 * max(1.0, abs(loss)) and abs(acc - cor / n) < 1e-12


# ------------------------------------------------------------------------------ BASELINE configs at full size
def test_mobilenet_large3d_batch32_full_size_properties():
    """BASELINE config 2 at its full size (32 clips of 16x224x224, bf16 autocast, train mode): finite loss and gradients,
    bit-identical on a second run, and the gradient of a clip-summed loss is the sum of per-half-batch gradients in eval
    mode (linearity over the batch: no cross-clip leakage outside BatchNorm)."""
    torch.manual_seed(1234)
    m = picklebot_amd.MobileNetLarge3D(13).to(DEV).train()
    x = torch.randint(0, 256, (32, 16, 224, 224, 3), dtype=torch.uint8, device=DEV).permute(0, -1, 1, 2, 3).to(torch.bfloat16) / 255
    y = torch.randint(0, 13, (32,), device=DEV)

    def grads(model, xs, ys, reduction="mean"):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(model(xs).float(), ys, reduction=reduction)
        loss.backward()
        return float(loss), [p.grad.detach().clone() for p in model.parameters()]

    for mod in m.modules():
        if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout3d)):
            mod.p = 0.0
    l1, g1 = grads(m, x, y)
    bufs = {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
    m.load_state_dict({**m.state_dict(), **bufs})
    assert l1 == l1 and all(torch.isfinite(g).all() for g in g1)
    # same inputs, same statistics buffers -> same bits (only running stats moved, they do not enter train-mode math)
    l2, g2 = grads(m, x, y)
    assert l1 == l2 and all(torch.equal(a, b) for a, b in zip(g1, g2))
    m.eval()
    _, gs = grads(m, x, y, "sum")
    _, ga = grads(m, x[:16], y[:16], "sum")
    _, gb = grads(m, x[16:], y[16:], "sum")
    gmax = max(float(g.norm()) for g in gs)
    worst = max(float((s - (a + b)).norm()) / max(float(s.norm()), 1e-3 * gmax) for s, a, b in zip(gs, ga, gb))
    assert worst <= 2e-2, worst


def test_mobilevit_s_full_size_training_properties():
    """BASELINE config 4 at its full size per GPU (8 clips of 16x256x256, bf16 autocast): finite, loss decreases over AdamW
    steps on a fixed batch, and eval-mode logits of a clip do not depend on its batch neighbours."""
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = picklebot_amd.MobileViT(num_classes=13, **protocol.MOBILEVIT_CFG["s"]).to(DEV).train()
    x = torch.rand(8, 16, 256, 256, 3, device=DEV).permute(0, 4, 1, 2, 3).to(torch.bfloat16)
    y = torch.randint(0, 13, (8,), device=DEV)
    opt = picklebot_amd.FlatAdamW(m.parameters(), lr=1e-3)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16), contextlib.redirect_stdout(io.StringIO()):
            loss = torch.nn.functional.cross_entropy(m(x).float(), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(v == v for v in losses) and min(losses[-3:]) < losses[0], losses
    m.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        full = m(x).float().cpu()
        part = torch.cat([m(x[i:i + 1]) for i in (0, 5)]).float().cpu()
    assert rel_err(part.numpy(), full[[0, 5]].numpy()) <= 2e-2


def test_mobilevit_xxs_fp8_attention_forward(golden):
    """BASELINE config 5: MobileViT-XXS inference with fp8 MFMA attention (picklebot_amd.fp8_attention()).  Checked against
    the reference's fp64 logits (model_mobilevit_xxs.npz): the fp8 forward must stay within 3 x the reference's own
    bf16-autocast deviation or 5e-2, and the bf16 forward of the same build is reported beside it."""
    with contextlib.redirect_stdout(io.StringIO()):
        m = protocol.build_model(picklebot_amd, "mobilevit_xxs", DEV).eval()
    shape = protocol.MODEL_CASES["mobilevit_xxs"][2]
    x = formula.clip_features(*shape).to(DEV)
    gold = golden("model_mobilevit_xxs")
    want = np.asarray(gold["logits_eval"], np.float64)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16), contextlib.redirect_stdout(io.StringIO()):
        bf = m(x).float().cpu().numpy()
        with picklebot_amd.fp8_attention():
            f8 = m(x).float().cpu().numpy()
    e_bf, e_f8 = rel_err(bf, want), rel_err(f8, want)
    ref_bf = float(gold["refbf16_err/logits_eval"][0]) if "refbf16_err/logits_eval" in gold else 0.0
    print(f"mobilevit_xxs eval logits vs fp64 reference: bf16 {e_bf:.2e}, fp8-attention {e_f8:.2e} (reference's own bf16 {ref_bf:.2e})")
    assert e_f8 <= max(5e-2, 3 * ref_bf), (e_f8, ref_bf)
    assert not np.array_equal(bf, f8)            # the fp8 kernel really ran


def _a2_streaming_layers():
    from picklebot_amd import arch
    return sorted(set(l for l in arch.movinet_a2_depthwise_layers(172, 172) if l[1][0] > 1))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("E,kernel,stride,hw", _a2_streaming_layers())
def test_movinet_a2_depthwise_streaming_at_scale(E, kernel, stride, hw, dtype):
    """BASELINE config 3 / SURVEY 8a-5 at ITS OWN scale: every (kt,3,3) depthwise layer of MoViNet-A2 at its true shape in
    a 50-frame 172x172 clip ((40,50,43,43) k333 ... (480,50,6,6), movinet.py:98-137) as a causal stream-buffer layer:
    the clip fed in chunks of 1, 2 and 8 frames with the carried (kt-1)-frame history equals the full-clip output (fp32:
    1e-6, bf16: bit for bit - the same kernel computes each output frame from the same frames either way), and the full
    clip equals the CPU mirror."""
    from oracle import kernel_ref as R
    torch.manual_seed(E + kernel[0])
    layer = picklebot_amd.CausalConv3d(E, E, kernel, stride=stride, padding=(0, kernel[1] // 2, kernel[2] // 2), groups=E, bias=False)
    layer = layer.to(DEV).eval()
    x = torch.randn(1, E, 50, hw[0], hw[1], device=DEV).to(dtype).to(memory_format=torch.channels_last_3d)
    with torch.no_grad():
        full = layer(x)
        x5 = x.permute(0, 2, 3, 4, 1).contiguous().cpu()
        w = layer.conv3d.weight.detach().cpu().view(E, -1)
        want = R.dw_fwd(x5, w, kernel, stride, (kernel[0] - 1, kernel[1] // 2, kernel[2] // 2), (0, kernel[1] // 2, kernel[2] // 2))
        got5 = full.permute(0, 2, 3, 4, 1).float().cpu()
        assert rel_err(got5.numpy(), want.float().numpy()) <= (2e-4 if dtype == torch.float32 else 1.2e-2)
        for chunk in (1, 2, 8):
            state, pieces = None, []
            for t0 in range(0, 50, chunk):
                y, state = layer.forward_stream(x[:, :, t0:t0 + chunk], state)
                pieces.append(y)
            stream = torch.cat(pieces, 2)
            assert stream.shape == full.shape
            if dtype == torch.bfloat16:
                assert torch.equal(stream, full), f"chunk {chunk}"
            else:
                assert rel_err(stream.float().cpu().numpy(), full.float().cpu().numpy()) <= 1e-6, f"chunk {chunk}"


def _large3d_train_grads(steps=1, hook=False, region=True, wrap=None, set_to_none=False, node_hook=None):
    """Train-mode gradients of a small MobileNetLarge3D clip.  region: run inside `weight_grad_overlap()` (the caller's
    promise that nothing reads .grad during backward); wrap: e.g. stock DistributedDataParallel; set_to_none: zero_grad
    between the steps like train.py:289; node_hook(model): registers hooks before the first step."""
    from picklebot_amd import kernels as K
    torch.manual_seed(3)
    model = picklebot_amd.MobileNetLarge3D(num_classes=13).to(DEV).train()
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout3d)):
            m.p = 0.0
    if hook:
        model.block3[0].pointwise_conv1.weight.register_post_accumulate_grad_hook(lambda p: None)
    if node_hook is not None:
        node_hook(model)
    net = wrap(model) if wrap is not None else model
    x = torch.randn(4, 3, 8, 64, 64, device=DEV)
    y = torch.arange(4, device=DEV) % 13
    with (K.weight_grad_overlap() if region else contextlib.nullcontext()):
        for i in range(steps):
            if set_to_none and i:
                net.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = torch.nn.functional.cross_entropy(net(x).float(), y)
            loss.backward()
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in model.named_parameters()}


def test_weight_gradient_side_branch_is_bit_identical(monkeypatch):
    """kernels.weight_grad_branch: a block's weight gradients run on a second HIP stream and are joined when the backward
    pass ends.  Same kernels in the same order - the gradients must be bit-identical to the in-line schedule; a second
    backward into existing `.grad` tensors (autograd then ADDS, reading the new gradient on the calling stream at once) and
    a block whose parameter carries a foreign hook must fall back to the in-line schedule and still be exact."""
    from picklebot_amd import kernels as K
    monkeypatch.setattr(K, "_WGRAD_OVERLAP", False)
    ref = _large3d_train_grads()
    monkeypatch.setattr(K, "_WGRAD_OVERLAP", True)
    K.branch_counts.update(side=0, inline=0)
    plain = _large3d_train_grads(region=False)       # nobody opted in: every block in line (the DEFAULT, round 4)
    assert K.branch_counts["side"] == 0 and K.branch_counts["inline"] >= 15, K.branch_counts
    assert all(torch.equal(ref[k], plain[k]) for k in ref)
    K.branch_counts.update(side=0, inline=0)
    got = _large3d_train_grads()
    nb = K.branch_counts["side"]          # 15 bottlenecks + the 1x1x1 conv / classifier layers around them
    assert nb >= 15 and K.branch_counts["inline"] == 0, K.branch_counts
    assert not K._lag, "a weight-gradient branch was left unjoined after backward() returned"
    assert all(torch.equal(ref[k], got[k]) for k in ref), [k for k in ref if not torch.equal(ref[k], got[k])][:5]
    # accumulation: the second backward adds into .grad - every block in line, result = the first gradient twice... no:
    # BatchNorm's running statistics moved in between, so compare against the in-line schedule doing the same thing
    K.branch_counts.update(side=0, inline=0)
    acc = _large3d_train_grads(steps=2)
    assert K.branch_counts["side"] == nb and K.branch_counts["inline"] == nb, K.branch_counts
    monkeypatch.setattr(K, "_WGRAD_OVERLAP", False)
    acc_ref = _large3d_train_grads(steps=2)
    assert all(torch.equal(acc_ref[k], acc[k]) for k in acc_ref)
    monkeypatch.setattr(K, "_WGRAD_OVERLAP", True)
    K.branch_counts.update(side=0, inline=0)
    hooked = _large3d_train_grads(hook=True)
    assert K.branch_counts["side"] == nb - 1 and K.branch_counts["inline"] == 1, K.branch_counts
    assert all(torch.equal(ref[k], hooked[k]) for k in ref)


@pytest.mark.parametrize("case", ["large3d", "movinet_a2", "mobilevit_s"])
def test_weight_gradient_side_branch_on_whole_models(case, monkeypatch):
    """Every block kind that defers its weight gradients (bottleneck, 1x1x1 conv, depthwise, dense conv, attention and
    feed-forward blocks): the protocol's train-mode gradients with the side stream on are bit-identical to the in-line
    schedule, every branch ran on the side stream, and none is left unjoined when backward() has returned.  (This is also the
    guard for the adoption rule: a gradient tensor that autograd CLONES instead of adopting is read on the calling stream
    before the side stream has written it - it came out as zeros / partial sums before the backward functions returned
    fresh views.)"""
    from picklebot_amd import kernels as K
    res = {}
    for on in (False, True):
        monkeypatch.setattr(K, "_WGRAD_OVERLAP", on)
        K.branch_counts.update(side=0, inline=0)
        with contextlib.redirect_stdout(io.StringIO()), K.weight_grad_overlap():
            res[on] = protocol.collect_model_case(picklebot_amd, case, device=DEV, do_eval=False)
        assert not K._lag
        if on:
            assert K.branch_counts["side"] > 0 and K.branch_counts["inline"] == 0, K.branch_counts
    keys = [k for k in res[False] if k.startswith("grad/")]
    assert keys and all(torch.equal(res[False][k], res[True][k]) for k in keys), [k for k in keys if not torch.equal(res[False][k], res[True][k])][:5]


def test_stock_ddp_gradients_equal_the_unwrapped_models(tmp_path):
    """VERDICT r3 #1 / train.py:203-204, 269, 289: the reference wraps the model in stock DistributedDataParallel and calls
    zero_grad(set_to_none=True).  DDP's C++ hook on each AccumulateGrad node copies `.grad` into its bucket on the calling
    stream DURING backward; with a weight gradient still in flight on the side stream the bucket would get zeros or partial
    sums.  World-size-1 RCCL group, two steps: the gradients DDP leaves behind (bucket views, all-reduced) are torch.equal
    to the un-wrapped model's, and every block ran in line."""
    import torch.distributed as dist
    from picklebot_amd import kernels as K
    ref = _large3d_train_grads(steps=2, region=False, set_to_none=True)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        K.branch_counts.update(side=0, inline=0)
        ddp = lambda m: torch.nn.parallel.DistributedDataParallel(m, device_ids=[torch.cuda.current_device()])   # noqa: E731
        got = _large3d_train_grads(steps=2, region=False, set_to_none=True, wrap=ddp)
        assert K.branch_counts["side"] == 0 and K.branch_counts["inline"] >= 30, K.branch_counts
        bad = [k for k in ref if not torch.equal(ref[k], got[k])]
        assert not bad, bad[:5]
    finally:
        dist.destroy_process_group()


def test_accumulate_grad_node_hook_sees_the_final_gradient():
    """A hook on a parameter's AccumulateGrad NODE (what DDP registers from C++) is invisible on the parameter
    (`_backward_hooks`, `_post_accumulate_grad_hooks` stay empty) and fires during backward, right after `.grad` is set:
    what it reads on the calling stream must be the finished gradient."""
    seen, nodes = {}, []

    def node_hook(model):
        for name, p in model.named_parameters():
            if name.endswith(("pointwise_conv1.weight", "depthwise_conv.weight", "pointwise_conv2.weight", "se.1.weight")):
                node = p.view_as(p).grad_fn.next_functions[0][0]
                nodes.append(node)           # (the node must stay alive: parameters hold it weakly)
                node.register_hook(lambda gi, go, name=name, p=p: seen.__setitem__(name, p.grad.clone()))

    got = _large3d_train_grads(region=False, node_hook=node_hook)
    assert len(seen) >= 45
    bad = [k for k in seen if not torch.equal(seen[k], got[k])]
    assert not bad, bad[:5]
    assert all(float(seen[k].abs().sum()) > 0 for k in seen)


def test_parameter_with_two_weight_gradient_nodes_in_one_pass():
    """a module called twice (or tied weights) gets two gradients for one parameter in the same pass; autograd's
    input buffer adds them on the calling stream when the second arrives.  With the side stream opted in the second block
    must join the branch that still writes the first and run in line: same bits as the in-line schedule."""
    from picklebot_amd import kernels as K

    def run(region):
        torch.manual_seed(11)
        blk = picklebot_amd.Bottleneck3D(16, 16, 48, stride=1, use_se=True, kernel_size=3).to(DEV).train()
        x = torch.randn(2, 16, 4, 12, 12, device=DEV).to(memory_format=torch.channels_last_3d).requires_grad_()
        with (K.weight_grad_overlap() if region else contextlib.nullcontext()):
            blk(blk(x)).square().mean().backward()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in blk.named_parameters()}, x.grad.clone()

    ref, dx_ref = run(False)
    K.branch_counts.update(side=0, inline=0)
    got, dx = run(True)
    assert K.branch_counts == {"side": 1, "inline": 1}, K.branch_counts
    assert not K._lag and not K._seen
    assert torch.equal(dx_ref, dx) and all(torch.equal(ref[k], got[k]) for k in ref)


def test_bench_data_parallel_path_on_one_gpu(tmp_path):
    """The path the 8-GPU run takes, at world size 1 over RCCL (PB_BENCH_FORCE_DP=1): bench.py builds the reducer, captures
    forward + backward into a HIP graph BEFORE the process group exists (ProcessGroupNCCL's watchdog thread polling events
    during a capture aborted the process 1 run in 6-10 on this stack), binds the group, replays + all-reduces + steps, and
    prints ONE JSON line last on stdout."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PB_BENCH_FORCE_DP="1", MASTER_PORT="29533")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "4", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])                            # (RCCL prints a version banner first; the JSON line is the LAST line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["clips_per_gpu"] == 4
    assert "captured before the process group exists" in d["config"]["launch"], d["config"]["launch"]
    assert d["roofline"]["kernel"] and d["roofline"]["frac"] > 0
