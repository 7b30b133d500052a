"""Wall time of the replayed MobileNetLarge3D training step split into its forward (captured alone, autograd graph built and
dropped) and the rest (backward + optimizer), with the weight-gradient side stream on and off.
    python tools/probes/fwd_bwd_split.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import picklebot_amd as pb  # noqa: E402
from picklebot_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = pb.MobileNetLarge3D(num_classes=13).to(dev).train()
x = torch.randn(32, 3, 16, 224, 224, device=dev).to(torch.bfloat16).to(memory_format=torch.channels_last_3d)
y = torch.randint(0, 13, (32,), device=dev)
opt = pb.FlatAdamW(model.parameters(), lr=3e-4)


def fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return torch.nn.functional.cross_entropy(model(x).float(), y)


def full():
    opt.zero_grad()
    loss = fwd()
    loss.backward()
    opt.step()
    return loss


PRIO = int(os.environ.get("PB_PROBE_MAIN_PRIORITY", "0"))


def replay_ms(fn):
    s = torch.cuda.Stream(priority=PRIO)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        K.reset_zero_arena()
        with torch.cuda.graph(g, stream=s):
            out = fn()
        K.reset_zero_arena()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            g.replay()
        torch.cuda.synchronize()
    del out
    return (time.perf_counter() - t0) / 30 * 1e3


for overlap in (True, False):
    K._WGRAD_OVERLAP = overlap
    f, t = replay_ms(fwd), replay_ms(full)
    print(f"weight-gradient side stream {'on ' if overlap else 'off'}: forward {f:.3f} ms, whole step {t:.3f} ms, backward + optimizer {t - f:.3f} ms")
