"""FlatAdamW host logic on CPU: the flat-bucket bookkeeping (parameters re-pointed into one buffer, gradients gathered
by one multi-tensor copy, device-side step count, GradScaler-style skip) with the HIP kernel replaced by its CPU mirror
(oracle/kernel_ref.py, test-only substitution) must reproduce torch.optim.AdamW step for step."""
import copy

import pytest
import torch

from oracle import kernel_ref
from picklebot_amd import optim as pb_optim


@pytest.fixture()
def cpu_kernels(monkeypatch):
    monkeypatch.setattr(pb_optim, "K", kernel_ref)


def _net():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.ReLU(), torch.nn.Linear(13, 5, bias=True),
                               torch.nn.LayerNorm(5))


def test_flat_adamw_matches_torch_adamw(cpu_kernels):
    a, b = _net(), None
    b = copy.deepcopy(a)
    ref = torch.optim.AdamW(a.parameters(), lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=5e-2)
    opt = pb_optim.FlatAdamW(b.parameters(), lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=5e-2)
    for p in b.parameters():                       # parameters live in the flat bucket, 64-byte aligned
        assert p.data_ptr() % 64 == opt.pflat[0].data_ptr() % 64
    x = torch.randn(11, 7)
    for it in range(5):
        ref.zero_grad(set_to_none=True)
        opt.zero_grad()
        a(x).square().mean().backward()
        b(x).square().mean().backward()
        ref.step()
        opt.step()
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            # (lr and weight_decay reach the kernel as fp32 - the device-side pair a replayed HIP graph reads - where
            # torch.optim.AdamW folds them in double: ~1e-5 relative on the first steps' tiny second moments)
            assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), (it, n, (p - q).abs().max())
    assert float(opt.step_t) == 5.0


def test_flat_adamw_skips_on_inf_and_unscales(cpu_kernels):
    a = _net()
    b = copy.deepcopy(a)
    ref = torch.optim.AdamW(a.parameters(), lr=1e-2)
    opt = pb_optim.FlatAdamW(b.parameters(), lr=1e-2)
    x = torch.randn(4, 7)
    scale = torch.tensor(1024.0)
    # scaled loss on ours, plain on the reference: the unscale must make them equal
    ref.zero_grad(set_to_none=True); opt.zero_grad()
    a(x).sum().backward()
    (b(x).sum() * scale).backward()
    ref.step()
    opt.step(inv_scale=1.0 / scale, check_inf=True)
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
    # an inf gradient: parameters and step count stay
    before = [q.detach().clone() for q in b.parameters()]
    opt.zero_grad()
    b(x).sum().backward()
    next(b.parameters()).grad[0, 0] = float("inf")
    opt.step(check_inf=True)
    assert float(opt.found_inf) == 1.0 and float(opt.step_t) == 1.0
    for q0, q in zip(before, b.parameters()):
        assert torch.equal(q0, q)


def test_flat_adamw_follows_a_torch_lr_scheduler(cpu_kernels):
    """The reference wraps its optimizer in CosineAnnealingLR (train.py:215): FlatAdamW is a torch.optim.Optimizer, and the
    (lr, weight_decay) pair the kernel reads (a tensor: device memory in the product) follows param_groups[0]."""
    a = _net()
    b = copy.deepcopy(a)
    ref = torch.optim.AdamW(a.parameters(), lr=1e-2, weight_decay=1e-2)
    opt = pb_optim.FlatAdamW(b.parameters(), lr=1e-2, weight_decay=1e-2)
    assert isinstance(opt, torch.optim.Optimizer)
    s_ref = torch.optim.lr_scheduler.CosineAnnealingLR(ref, T_max=6, eta_min=1e-4)
    s_opt = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=6, eta_min=1e-4)
    x = torch.randn(9, 7)
    for it in range(6):
        ref.zero_grad(set_to_none=True); opt.zero_grad()
        a(x).square().mean().backward()
        b(x).square().mean().backward()
        ref.step(); opt.step()
        s_ref.step(); s_opt.step()
        assert abs(ref.param_groups[0]["lr"] - opt.param_groups[0]["lr"]) < 1e-12
        for p, q in zip(a.parameters(), b.parameters()):
            assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), it
    # the pair the kernel reads lags one refresh behind the scheduler; refresh_hyper() (called by step(), or by a
    # graph-replay loop before each replay) brings it up to date
    opt.refresh_hyper()
    assert abs(float(opt.hyper[0]) - opt.param_groups[0]["lr"]) < 1e-9
