"""Conditioning screen for the train-mode gradient fixtures (test infrastructure; build container only).

A fixture can carry a FLAT tolerance only if the function it pins is well conditioned at that input: if the reference's own
fp32 arithmetic, re-associated in equally valid ways, moves a gradient by more than the tolerance, the fixture measures the
summation order of whoever is compared with it, not correctness (MoViNet-A2's train8 fixture sat one ulp from a
kink; MobileNetLarge3D's ended in 4x4 frames).  For a candidate (clip shape, input salt) this module runs the REAL reference

  * once in fp64                                  -> the golden values,
  * in fp32 under several arithmetic VARIANTS     -> per-key deviation from fp64 of every variant; `spread` = the largest,
      v0: 8 threads, channels-last-3d input (train.py:106)      v1: 1 thread
      v3: contiguous NCDHW input, 3 threads                       v4: 5 threads
      (a fifth, oneDNN switched off, aborted inside torch's native conv3d - "double free or corruption" - on the 224x224 clips)
  * under bf16 autocast for the same variants      -> `bf16_bound` = mean + 3 sigma of the per-key deviation (floor 1e-2),

and accepts the candidate when every gradient key's fp32 spread is <= LIMIT (2.5e-4: a quarter of the 1e-3 bar).
"""
import contextlib
import io

import numpy as np
import torch

from . import protocol

LIMIT = 2.5e-4
VARIANTS = ("v0", "v1", "v3", "v4")


@contextlib.contextmanager
def variant(name):
    old_threads = torch.get_num_threads()
    try:
        if name == "v1":
            torch.set_num_threads(1)
        elif name == "v3":
            torch.set_num_threads(3)
        elif name == "v4":
            torch.set_num_threads(5)
        else:
            torch.set_num_threads(8)
        if name == "v2":
            with torch.backends.mkldnn.flags(enabled=False):
                yield
        else:
            yield
    finally:
        torch.set_num_threads(old_threads)


def run_isolated(case, shape, salt, dtype, autocast=False, vname="v0", mode="train"):
    """run() in a child process (torch's CPU kernels aborted - "double free or corruption" - in one of the arithmetic
    variants on the 224x224 clips: a child's abort costs that variant, not the whole fixture run).  -> raw tensors or None."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "r.npz")
        code = ("import sys; sys.path.insert(0, %r); import numpy as np, torch\n"
                "from oracle import make_golden, condition_screen as cs\n"
                "ns = make_golden.load_reference_namespace()\n"
                "r = cs.run(ns, %r, %r, %r, getattr(torch, %r), autocast=%r, vname=%r, mode=%r)\n"
                "np.savez(%r, **{k: v.numpy() for k, v in r.items()})\n"
                % (root, case, tuple(shape), int(salt), str(dtype).split(".")[-1], bool(autocast), vname, mode, out))
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(out):
            print(f"[{case}/{mode}] variant {vname} (autocast={autocast}) died: rc {r.returncode} {r.stderr.strip().splitlines()[-1:] }", flush=True)
            return None
        with np.load(out) as z:
            return {k: torch.from_numpy(z[k]) for k in z.files}


def run(ns, case, shape, salt, dtype, autocast=False, vname="v0", mode="train"):
    quiet = io.StringIO()
    with contextlib.redirect_stdout(quiet), variant(vname):
        return protocol.collect_grad_case(ns, case, mode, dtype=dtype, autocast=autocast, shape=shape, salt=salt,
                                          contiguous_input=(vname == "v3"))


KINKS = {"ReLU": (0.0,), "LeakyReLU": (0.0,), "Hardswish": (-3.0, 3.0), "Hardsigmoid": (-3.0, 3.0), "ReLU6": (0.0, 6.0)}


def kink_census(ns, case, shape, salt, radius=1e-5):
    """fp64 train-mode forward of the REAL reference: how many activation inputs lie within `radius` of a point where the
    activation's derivative jumps (ReLU 0; hardswish / hardsigmoid +-3), and the smallest distance.  One such element flipping
    between two arithmetics changes every upstream gradient by ~1e-3..1e-2 (measured: MobileNetLarge3D 8x8x112x112, two
    single-element hardswish flips under a 1e-6 parameter perturbation moved all gradients by 1.4 %)."""
    import torch.nn as nn
    quiet = io.StringIO()
    with contextlib.redirect_stdout(quiet):
        model = protocol.build_model(ns, case, "cpu").double().train()
    stat = {"near": 0, "min": float("inf"), "elems": 0}

    def hook(mod, inp):                  # a PRE-hook: nn.ReLU(inplace=True) (mobilenet.py:18) overwrites its input
        ks = KINKS.get(type(mod).__name__)
        if ks is None:
            return
        x = inp[0].detach()
        for k in ks:
            d = (x - k).abs()
            stat["near"] += int((d < radius).sum())
            stat["min"] = min(stat["min"], float(d.min()))
        stat["elems"] += x.numel()
    hs = [m.register_forward_pre_hook(hook) for m in model.modules() if type(m).__name__ in KINKS]
    from . import formula
    with torch.no_grad(), contextlib.redirect_stdout(quiet):
        model(formula.clip_features(*shape, dtype=torch.float64, salt=salt))
    for h in hs:
        h.remove()
    return stat


def kappa(ns, case, shape, salt, eps=1e-8):
    """Smooth conditioning of the train-mode gradients: relative change of every parameter gradient per unit relative
    perturbation of the parameters, in fp64 (eps small enough that no activation changes sides of a kink)."""
    import torch.nn.functional as F
    from . import formula
    ncls = protocol.MODEL_CASES[case][1]["num_classes"]

    def grads(pert):
        quiet = io.StringIO()
        with contextlib.redirect_stdout(quiet):
            m = protocol.build_model(ns, case, "cpu").double()
        if pert:
            with torch.no_grad():
                for name, p in m.named_parameters():
                    u = torch.from_numpy(formula.unit("pert:" + name, p.numel(), lane=5).reshape(tuple(p.shape))).double()
                    p.mul_(1 + pert * u)
        m.train()
        x = formula.clip_features(*shape, dtype=torch.float64, salt=salt)
        y = formula.labels(shape[0], ncls)
        with contextlib.redirect_stdout(quiet):
            F.cross_entropy(m(x), y).backward()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    g0, g1 = grads(0.0), grads(eps)
    gmax = max(float(v.norm()) for v in g0.values())
    ks = np.array([float((g1[k] - g0[k]).norm() / g0[k].norm()) / eps for k in g0 if float(g0[k].norm()) > 1e-6 * gmax])
    return float(np.median(ks)), float(ks.max())


def screen(ns, case, shape, salt, with_bf16=True, verbose=True):
    """-> (accepted, truth, spread {key: float}, bf16_bound {key: float}, report str)."""
    truth = run(ns, case, shape, salt, torch.float64)
    devs = {v: protocol.deviations(run(ns, case, shape, salt, torch.float32, vname=v), truth) for v in VARIANTS}
    spread = {k: max(float(devs[v][k][0]) for v in VARIANTS) for k in devs["v0"]}
    gkeys = [k for k in spread if k.startswith("grad/")]
    gmax = max(float(torch.linalg.vector_norm(truth[k])) for k in gkeys)
    live = [k for k in gkeys if float(torch.linalg.vector_norm(truth[k])) >= 1e-6 * gmax]     # (structural zeros: conv bias before BN)
    worst = max(live, key=lambda k: spread[k])
    g = np.array([spread[k] for k in live])
    ok = bool(g.max() <= LIMIT)
    rep = (f"[{case} {shape} salt {salt}] fp32 spread over {len(VARIANTS)} variants, {len(live)} gradient keys: median {np.median(g):.2e} "
           f"p90 {np.percentile(g, 90):.2e} max {g.max():.2e} ({worst}) -> {'ACCEPT' if ok else 'reject'}")
    bound = {}
    if with_bf16 and ok:
        draws = {v: protocol.deviations(run(ns, case, shape, salt, torch.float32, autocast=True, vname=v), truth) for v in VARIANTS}
        for k in draws["v0"]:
            d = np.array([float(draws[v][k][0]) for v in VARIANTS])
            bound[k] = max(1e-2, float(d.mean() + 3.0 * d.std(ddof=1)))
        b = np.array([bound[k] for k in live])
        rep += f"; bf16 bound (mean+3sigma over {len(VARIANTS)} draws, floor 1e-2): median {np.median(b):.2e} max {b.max():.2e}"
    if verbose:
        print(rep, flush=True)
    return ok, truth, spread, bound, rep
