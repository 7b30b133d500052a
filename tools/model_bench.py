"""Per-kernel-family timing of the other BASELINE.json workloads (not the bench line; that is bench.py).

  python tools/model_bench.py mobilevit_s_train --batch 4 --steps 5 [--detail gpurun_out/x.txt]

Workloads: mobilevit_s_train / mobilevit_xxs_train (16x256x256), mobilevit_xxs_fwd (32x224x224, eval),
movinet_a2_fwd (50x172x172, eval), mobilenet_small_train (16x224x224).
Prints wall ms/step (eager launches, synchronised) and the HIP-event family table.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CFG = {  # reference config/*.json
    "s": dict(dims=[144, 192, 240], channels=[16, 32, 64, 64, 96, 96, 128, 128, 160, 160, 640]),
    "xs": dict(dims=[96, 120, 144], channels=[16, 32, 48, 48, 64, 64, 80, 80, 96, 96, 384]),
    "xxs": dict(dims=[64, 80, 96], channels=[16, 16, 24, 24, 48, 48, 64, 64, 80, 80, 320]),
}


def movinet_a2_stream(batch, chunk, steps):
    """BASELINE config 3, streaming variant: the 26 depthwise convs of MoViNet-A2 at their true shapes in a 50-frame
    172x172 clip, each as a causal stream-buffer layer (picklebot_amd.CausalConv3d.forward_stream, the (kt-1)-frame
    history carried in a caller-owned state), fed `chunk` frames at a time: us per frame and achieved HBM rate."""
    import picklebot_amd as pb
    from picklebot_amd import arch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    layers = arch.movinet_a2_depthwise_layers(172, 172)
    print(f"# MoViNet-A2 depthwise stack, streaming: batch {batch}, chunks of {chunk} frame(s) of a 50-frame 172x172 clip, bf16")
    print("%-42s %10s %10s %10s" % ("layer (E, kernel, stride, HxW in)", "us/frame", "GB/s", "full us/fr"))
    tot_stream = tot_full = tot_bytes = 0.0
    for E, kernel, stride, hw in layers:
        layer = pb.CausalConv3d(E, E, kernel, stride=stride, padding=(0, kernel[1] // 2, kernel[2] // 2), groups=E, bias=False).to(dev).eval()
        x = torch.randn(batch, E, 50, hw[0], hw[1], device=dev).to(torch.bfloat16).to(memory_format=torch.channels_last_3d)

        def run_stream():
            state = None
            for t0 in range(0, 50, chunk):
                y, state = layer.forward_stream(x[:, :, t0:t0 + chunk], state)
            return y

        def timed(fn):
            with torch.no_grad():
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(steps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
            return e0.elapsed_time(e1) / steps * 1e3          # us per clip batch
        with torch.no_grad():
            yfull = layer(x)
        us_s, us_f = timed(run_stream), timed(lambda: layer(x))
        nbytes = (x.numel() + yfull.numel()) * 2
        tot_stream += us_s; tot_full += us_f; tot_bytes += nbytes
        print("%-42s %10.2f %10.1f %10.2f" % (f"{E} {kernel} {stride[1:]} {hw[0]}x{hw[1]}", us_s / 50, nbytes / us_s / 1e3, us_f / 50))
    print("%-42s %10.2f %10.1f %10.2f" % ("all 26 layers", tot_stream / 50, tot_bytes / tot_stream / 1e3, tot_full / 50))
    # the same streaming step - one chunk through all 26 layers, each carrying its own history - replayed from ONE HIP graph:
    # what a deployed stream costs on the GPU once the ~30 us of Python + ctypes per launch are out of the way
    from picklebot_amd import kernels as K
    pairs = []
    for E, kernel, stride, hw in layers:
        layer = pb.CausalConv3d(E, E, kernel, stride=stride, padding=(0, kernel[1] // 2, kernel[2] // 2), groups=E, bias=False).to(dev).eval()
        xc = torch.randn(batch, E, chunk, hw[0], hw[1], device=dev).to(torch.bfloat16).to(memory_format=torch.channels_last_3d)
        nh = kernel[0] - 1
        st = torch.zeros(batch, nh, hw[0], hw[1], E, device=dev, dtype=torch.bfloat16) if nh else None
        pairs.append((layer, xc, st))

    def chunk_step():
        for layer, xc, st in pairs:
            layer.forward_stream(xc, st)
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chunk_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        K.reset_zero_arena()
        with torch.cuda.graph(g):
            chunk_step()
        K.reset_zero_arena()
        nrep = (50 + chunk - 1) // chunk
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps * nrep):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    us_clip = e0.elapsed_time(e1) / steps * 1e3
    print("%-42s %10.2f %10.1f   (one HIP graph per %d-frame chunk, %d replays per clip)" % (
        "all 26 layers, graph replay", us_clip / 50, tot_bytes / us_clip / 1e3, chunk, nrep))
    print(json.dumps({"workload": "movinet_a2_stream", "batch": batch, "chunk": chunk, "us_per_frame_stream": round(tot_stream / 50, 2),
                      "us_per_frame_stream_hipgraph": round(us_clip / 50, 2), "us_per_frame_full_clip": round(tot_full / 50, 2),
                      "GBps_stream": round(tot_bytes / tot_stream / 1e3, 1), "GBps_stream_hipgraph": round(tot_bytes / us_clip / 1e3, 1)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--detail", default=None)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--graph", action="store_true", help="also time the step replayed from one HIP graph (as bench.py does)")
    ap.add_argument("--fp8", action="store_true", help="inference: patch attention with fp8 MFMA operands (BASELINE config 5)")
    ap.add_argument("--chunk", type=int, default=1, help="movinet_a2_stream: frames per chunk")
    args = ap.parse_args()
    if args.workload == "movinet_a2_stream":
        return movinet_a2_stream(args.batch, args.chunk, args.steps)

    import picklebot_amd as pb
    from picklebot_amd import kernels as K
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    w = args.workload
    train = w.endswith("_train")
    if w.startswith("mobilevit_"):
        size = w.split("_")[1]
        model = pb.MobileViT(num_classes=13, **CFG[size])
        shape = (args.batch, 3, 16, 256, 256) if train else (args.batch, 3, 32, 224, 224)
    elif w.startswith("movinet_a2"):
        model = pb.MoViNetA2(num_classes=13)
        shape = (args.batch, 3, 50, 172, 172)
    elif w.startswith("mobilenet_small"):
        model = pb.MobileNetSmall3D(num_classes=13)
        shape = (args.batch, 3, 16, 224, 224)
    elif w.startswith("mobilenet_large"):
        model = pb.MobileNetLarge3D(num_classes=13)
        shape = (args.batch, 3, 16, 224, 224)
    else:
        raise SystemExit("unknown workload " + w)
    model = model.to(dev)
    model.train(train)
    x = torch.randn(shape, device=dev)
    if not args.fp32:
        x = x.to(torch.bfloat16)
    x = x.to(memory_format=torch.channels_last_3d)
    y = torch.randint(0, 13, (args.batch,), device=dev)
    opt = torch.optim.AdamW(model.parameters(), lr=3e-4, fused=True, capturable=args.graph) if train else None
    K.weight_grad_overlap().__enter__()      # this loop reads gradients only after backward() has returned (opt-in, round 4)

    def step():
        if train:
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.fp32):
                loss = torch.nn.functional.cross_entropy(model(x).float(), y)
            loss.backward()
            opt.step()
            return loss
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.fp32):
            if args.fp8:
                with pb.fp8_attention():
                    return model(x)
            return model(x)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3

    graph_ms = None
    if args.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if train:
            opt.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        K.reset_zero_arena()
        with torch.cuda.graph(g):
            step()
        K.reset_zero_arena()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4 * args.steps):
            g.replay()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / (4 * args.steps) * 1e3
        print(f"{w} batch {args.batch}: {graph_ms:.3f} ms/step replayed from a HIP graph ({args.batch / graph_ms * 1e3:.1f} clips/s)")

    overlap, K._WGRAD_OVERLAP = K._WGRAD_OVERLAP, False      # per-kernel numbers in line (one kernel at a time), as in bench.py
    K.start_profile()
    for _ in range(args.steps):
        step()
    prof = K.stop_profile()
    K._WGRAD_OVERLAP = overlap
    detail = {k: v for k, v in prof.items() if " | " in k}
    prof = {k: v for k, v in prof.items() if " | " not in k}
    tot = sum(d["ms"] for d in prof.values())
    print(f"{w} batch {args.batch} {shape}: {ms:.2f} ms/step eager ({args.batch / ms * 1e3:.1f} clips/s), "
          f"sum of kernel events {tot / args.steps:.2f} ms/step, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    for name, d in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        print("  %-34s n=%4d  %8.3f ms/step  %7.1f GB/s  %5.1f%%" % (
            name, d["launches"] // args.steps, d["ms"] / args.steps, d["bytes"] / 1e6 / max(d["ms"], 1e-9), 100 * d["ms"] / tot))
    # patch attention against the dense MFMA peak (MI355X_MICROARCH.md: 2.5 PFLOP/s bf16, 5 PFLOP/s fp8): algorithmic
    # flops = 4 * n^2 * d per (group, head) forward (QK^T and PV), 2.5x that backward
    import re
    for k, d in sorted(detail.items(), key=lambda kv: -kv[1]["ms"]):
        mm = re.match(r"(attention_fwd_fp8|attention_fwd|attention_bwd) \| G(\d+) n(\d+)", k)
        if mm:
            kind, G, n = mm.group(1), int(mm.group(2)), int(mm.group(3))
            flops = 4.0 * n * n * 16 * 8 * G * (2.5 if kind == "attention_bwd" else 1.0)
            us = d["ms"] / d["launches"] * 1e3
            peak = 5.0e15 if kind.endswith("fp8") else 2.5e15
            print("  %-18s G%-5d n%-4d %8.1f us/launch  %7.1f TFLOP/s  = %.3f of the dense %s MFMA peak" % (
                kind, G, n, us, flops / us / 1e6, flops / (us * 1e-6) / peak, "fp8" if kind.endswith("fp8") else "bf16"))
    if args.detail:
        with open(args.detail, "w") as f:
            for k, d in sorted(detail.items(), key=lambda kv: -kv[1]["ms"]):
                f.write("%-100s n=%3d  %8.1f us/launch  %7.1f GB/s\n" % (
                    k, d["launches"] // args.steps, d["ms"] / d["launches"] * 1e3, d["bytes"] / 1e6 / max(d["ms"], 1e-9)))
    print(json.dumps({"workload": w, "batch": args.batch, "ms_per_step_eager": round(ms, 3),
                      "ms_per_step_hipgraph": None if graph_ms is None else round(graph_ms, 3)}))


if __name__ == "__main__":
    main()
