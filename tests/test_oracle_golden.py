"""CPU: the oracle restatement (oracle/torch_ref.py, fp32) against the golden vectors that
oracle/make_golden.py produced from the real reference (fp64 values + the reference's own fp32
deviation).  Spec tolerance 1e-5 relative; see conftest.check_model_result for
how the reference's own fp32 noise floor enters."""
import contextlib
import io

import pytest

from conftest import check_model_result, compare_full, golden_mode, rel_err
from oracle import protocol, torch_ref

TOL = 1e-5


@pytest.mark.parametrize("case", list(protocol.MODEL_CASES))
def test_model_matches_reference_golden(case, golden):
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_model_case(torch_ref, case)
    gold = golden("model_" + case)
    expected = {k for k in gold if k.split("/")[0] not in ("meta", "ref32_err", "refbf16_err")}
    assert set(res) == expected
    bad, worst = check_model_result(res, gold, TOL, TOL)
    assert not bad, f"{len(bad)} keys off, first {bad[:5]}, worst {worst}"


@pytest.mark.parametrize("mode", ["evalgrad", "train"])
@pytest.mark.parametrize("case", list(protocol.GRAD_CASES))
def test_gradient_fixtures_match_reference_golden(case, mode, golden):
    """The gradient fixtures (eval-mode BatchNorm; train mode on the screened clips): the oracle restatement against
    the real reference's fp64 values."""
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_grad_case(torch_ref, case, mode)
    gold = golden_mode(golden("grad_" + case), mode)
    expected = {k for k in gold if k.split("/")[0] not in ("ref32_err", "ref32_spread", "refbf16_err", "refbf16_mean", "refbf16_bound", "meta")}
    assert set(res) == expected
    bad, worst = check_model_result(res, gold, TOL, TOL)
    assert not bad, f"{len(bad)} keys off, first {bad[:5]}, worst {worst}"


@pytest.mark.parametrize("case", list(protocol.GRAD_CASES_LARGE))
def test_large_unscreened_train_fixtures_match_reference_golden(case, golden):
    """The 8x112x112 train-mode fixtures (not screened: activation inputs sit within rounding distance of derivative jumps,
    the reference's own fp32 variants are ~1e-2 apart on them): the fp32 restatement within 3x the reference's own spread."""
    shape, salt = protocol.GRAD_CASES_LARGE[case][:4], protocol.GRAD_CASES_LARGE[case][4]
    with contextlib.redirect_stdout(io.StringIO()):
        res = protocol.run_grad_case(torch_ref, case, "train", shape=shape, salt=salt)
    gold = golden_mode(golden("grad112_" + case), "train")
    expected = {k for k in gold if k.split("/")[0] not in ("ref32_err", "ref32_spread", "meta")}
    assert set(res) == expected
    bad, worst = check_model_result(res, gold, 1e-3, 1e-3, dev="ref32_spread", factor=3.0, samp_factor=4.0)
    assert not bad, f"{len(bad)} keys off, first {bad[:5]}, worst {worst}"


@pytest.mark.parametrize("case", list(protocol.BLOCK_CASES))
def test_block_matches_reference_golden(case, golden):
    res = protocol.run_block_case(torch_ref, case)
    g = golden("blocks")
    assert len([k for k in g if k.startswith(case + "/") and "_err/" not in k]) == len(res)
    bad, worst = compare_full(res, g, case + "/", TOL, TOL)
    assert not bad, f"{bad[:5]} worst {worst}"


@pytest.mark.parametrize("case", list(protocol.V2_CASES))
def test_v2_linear_attention_matches_reference_golden(case, golden):
    """mobilevitv2.LinearSelfAttention: the restatement against the real module's fp64 values."""
    res = protocol.run_block_case(torch_ref, case)
    g = golden("blocks_v2")
    assert len([k for k in g if k.startswith(case + "/") and "_err/" not in k]) == len(res)
    bad, worst = compare_full(res, g, case + "/", TOL, TOL)
    assert not bad, f"{bad[:5]} worst {worst}"


@pytest.mark.parametrize("case", ["causal_dw_533", "causal_333"])
def test_causal_streaming_known_answer(case, golden):
    res = protocol.run_causal_stream_case(torch_ref, case)
    g = golden("blocks")
    assert rel_err(res["full"], g[f"stream_{case}/full"]) <= TOL
    # chunked-with-history == full clip, both for the oracle and in the golden
    assert rel_err(res["stream"], res["full"]) <= 1e-6
    assert rel_err(g[f"stream_{case}/stream"], g[f"stream_{case}/full"]) <= 1e-12


def test_state_dict_keys_match_survey_examples():
    m = torch_ref.MobileNetLarge3D(13)
    sd = m.state_dict()
    assert len(sd) == 170 and sum(p.numel() for p in m.parameters()) == 4205675
    assert tuple(sd["block3.0.depthwise_conv.weight"].shape) == (72, 1, 1, 5, 5)
    assert tuple(sd["block3.0.squeeze_excite.se.1.weight"].shape) == (18, 72, 1, 1, 1)
    a2 = torch_ref.MoViNetA2(13)
    assert len(a2.state_dict()) == 333 and sum(p.numel() for p in a2.parameters()) == 3992289
    assert tuple(a2.state_dict()["block2.1.conv.weight"].shape) == (40, 1, 3, 3, 3)
    with contextlib.redirect_stdout(io.StringIO()):
        vs = torch_ref.MobileViT(num_classes=13, **protocol.MOBILEVIT_CFG["s"])
    assert sum(p.numel() for p in vs.parameters()) == 8453136
    assert tuple(vs.state_dict()["trunk.0.1.transformer.layers.0.0.to_qkv.weight"].shape) == (384, 144)
