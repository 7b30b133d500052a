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


def test_weight_gradient_branch_only_defers_when_autograd_adopts():
    """kernels.weight_grad_branch runs a block's weight gradients on a second stream and joins later - safe only when nothing
    reads the gradient tensor when the backward node returns.  The rule (no GPU needed): a leaf parameter without .grad and
    without hooks (the GradReducer's own hook, marked join-aware, excepted); and CPU tensors always run in line."""
    from picklebot_amd import kernels as K
    p = torch.nn.Parameter(torch.zeros(4, 4))
    assert K._adopting(p) and K._adopting(None)
    p.grad = torch.zeros(4, 4)
    assert not K._adopting(p)                     # autograd would ADD into it on the calling stream
    p.grad = None
    assert not K._adopting(p * 2)                 # non-leaf: the producing node consumes the gradient
    h = p.register_hook(lambda g: g)
    assert not K._adopting(p)
    h.remove()
    h = p.register_post_accumulate_grad_hook(lambda q: None)
    assert not K._adopting(p)
    p._pb_join_aware = True
    assert K._adopting(p)
    h.remove()
    ran = []
    K.branch_counts.update(side=0, inline=0)
    K.weight_grad_branch(torch.zeros(1), (p,), [lambda: ran.append(1)], ())
    assert ran == [1] and K.branch_counts == {"side": 0, "inline": 1}
    K.join_weight_grads()                         # nothing in flight: a no-op
