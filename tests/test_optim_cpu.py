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
            assert torch.allclose(p, q, rtol=1e-6, atol=1e-7), (it, n, (p - q).abs().max())
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
