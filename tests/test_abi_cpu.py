"""CPU: the C-ABI library builds, loads, and exports every symbol include/picklebot_amd.h declares;
the ctypes table covers the header; host-side argument validation fails loudly (no compute)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "picklebot_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pb_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_loads_and_exports_header_symbols():
    from picklebot_amd import _C, build
    build.build(verbose=False)
    lib = _C.load()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert set(names) == set(_C.SIGNATURES) | set(_C.STRING_SIGNATURES) | set(_C.SIZE_SIGNATURES) | set(_C.COUNT_SIGNATURES) | {"pb_last_error", "pb_abi_version"}
    assert lib.pb_abi_version() == _C.ABI_VERSION


def test_entry_points_reject_bad_arguments_without_a_gpu():
    from picklebot_amd import _C
    lib = _C.load()
    # null pointers: rejected by PB_CHECK_ARG before any HIP call
    rc = lib.pb_pwconv_fwd(None, None, None, 4, 8, 8, 8, 0, 1, None, None, 0, 0.0, None, 0, None, None, 0, 0.0, None, 0, None)
    assert rc != 0 and b"null pointer" in lib.pb_last_error()
    rc = lib.pb_chan_stats(None, 4, 8, 1, None, 0, None)
    assert rc != 0 and b"null" in lib.pb_last_error()


def test_product_path_has_no_cpu_fallback():
    import picklebot_amd
    m = picklebot_amd.MobileNetSmall3D(2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 8, 32, 32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "picklebot_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), fn


def test_weight_gradient_branch_only_defers_when_the_consumer_opted_in():
    """kernels.weight_grad_branch runs a block's weight gradients on a second stream and joins later - safe only when nothing
    reads the gradient tensor before the join.  The rule (no GPU needed): the DEFAULT is in line; a leaf parameter without
    .grad and without hooks defers only when its consumer opted in - per parameter (`_pb_join_aware`: FlatAdamW, GradReducer)
    or for a region (`weight_grad_overlap()`); CPU tensors always run in line."""
    from picklebot_amd import kernels as K
    import picklebot_amd
    p = torch.nn.Parameter(torch.zeros(4, 4))
    assert not K._adopting(p, False)              # nobody opted in: a C++ hook on the AccumulateGrad node could be reading
    assert K._adopting(p, True) and K._adopting(None, False)
    p.grad = torch.zeros(4, 4)
    assert not K._adopting(p, True)               # autograd would ADD into it on the calling stream
    p.grad = None
    assert not K._adopting(p * 2, True)           # non-leaf: the producing node consumes the gradient
    h = p.register_hook(lambda g: g)
    assert not K._adopting(p, True)
    h.remove()
    h = p.register_post_accumulate_grad_hook(lambda q: None)
    assert not K._adopting(p, True)               # a foreign hook reads .grad during backward, region or not
    p._pb_join_aware = True
    assert K._adopting(p, False)
    h.remove()
    assert K.weight_grad_overlap._depth == 0
    with picklebot_amd.weight_grad_overlap():
        assert K.weight_grad_overlap._depth == 1
    assert K.weight_grad_overlap._depth == 0
    ran = []
    K.branch_counts.update(side=0, inline=0)
    K.weight_grad_branch(torch.zeros(1), (p,), [lambda: ran.append(1)], ())
    assert ran == [1] and K.branch_counts == {"side": 0, "inline": 1}
    K.join_weight_grads()                         # nothing in flight: a no-op


def test_stock_ddp_parameters_are_not_opted_in(tmp_path):
    """DistributedDataParallel (train.py:203-204) hooks the AccumulateGrad node in C++; nothing on the
    parameter shows it.  So a DDP-wrapped model must find every parameter NOT opted in (weight gradients in line), and a
    FlatAdamW built in a process group without a GradReducer must not opt them in either."""
    import torch.distributed as dist
    from picklebot_amd import kernels as K
    import picklebot_amd
    dist.init_process_group("gloo", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1)
    try:
        m = torch.nn.Linear(16, 16)
        ddp = torch.nn.parallel.DistributedDataParallel(m)
        assert all(not K._adopting(p, False) for p in ddp.parameters())
        opt = picklebot_amd.FlatAdamW(m.parameters(), lr=1e-3)
        assert all(not getattr(p, "_pb_join_aware", False) for p in m.parameters())
        assert all(not K._adopting(p, False) for p in m.parameters())
        del opt, ddp
    finally:
        dist.destroy_process_group()
    opt = picklebot_amd.FlatAdamW(torch.nn.Linear(16, 16).parameters(), lr=1e-3)      # no process group: opted in
    assert all(p._pb_join_aware for p in opt.param_groups[0]["params"])
