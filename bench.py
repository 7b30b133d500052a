"""bench.py - clips/sec of the MobileNetLarge3D training step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic clips: uint8-valued clips already
resident in HBM as the bf16 channels-last tensor train.py:106 produces, forward under
autocast(bf16), CrossEntropyLoss, backward, gradient all-reduce over RCCL (N > 1), optimizer step
(torch fused AdamW - the reference's bitsandbytes AdamW8bit is outside the path).
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel family and `cpu_baseline` (the CPU oracle timed on this host, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable copy)
T, H, W, NCLS = 16, 224, 224, 13


def synthetic_batch(batch, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    u8 = torch.randint(0, 256, (batch, T, H, W, 3), dtype=torch.uint8, generator=g)
    labels = torch.randint(0, NCLS, (batch,), generator=g)
    x = (u8.to(device).permute(0, -1, 1, 2, 3).to(torch.bfloat16) / 255)      # train.py:102-108
    return x, labels.to(device)


def cpu_baseline(seconds_budget=20.0):
    """The CPU oracle (oracle/torch_ref.py, parity-locked to the reference goldens) on this host:
    fp32 train-mode fwd+bwd of MobileNetLarge3D, batch 4, same synthetic recipe."""
    from oracle import torch_ref
    torch.manual_seed(1234)
    # the GPU box gives one GPU a share of 16 host cores; more threads than that only oversubscribe
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cores = torch.get_num_threads()
    m = torch_ref.MobileNetLarge3D(NCLS).train()
    B = 4
    g = torch.Generator().manual_seed(1234)
    x = (torch.randint(0, 256, (B, T, H, W, 3), dtype=torch.uint8, generator=g).permute(0, -1, 1, 2, 3).float() / 255)
    y = torch.randint(0, NCLS, (B,), generator=g)
    best, iters, t_start = float("inf"), 0, time.time()
    while iters < 2 or (time.time() - t_start < seconds_budget and iters < 12):
        t0 = time.time()
        m.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(m(x), y).backward()
        dt = time.time() - t0
        if iters > 0:
            best = min(best, dt)
        iters += 1
    return {"value": round(B / best, 3), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"MobileNetLarge3D fp32 train fwd+bwd, batch {B} x 16x224x224, best of {iters - 1} iterations, "
                      f"torch {torch.__version__} CPU, os.cpu_count()={os.cpu_count()}"}


def log(*a):
    if os.environ.get("PB_BENCH_VERBOSE"):
        print("[bench %s]" % os.environ.get("RANK", "-"), *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU (BASELINE config: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="submit every kernel from Python instead of replaying one HIP graph per step")
    ap.add_argument("--torch-adamw", action="store_true", help="torch.optim.AdamW(fused=True) instead of picklebot_amd.FlatAdamW")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` by itself: one process per GPU (train.py:24-36 reads torchrun's environment) - start the
        # launcher as a CHILD process, relay its output (rank 0's one JSON line) and its exit code.  Nothing has touched the
        # GPU in this process, and nothing is exec'ed.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import picklebot_amd
    from picklebot_amd import kernels as K
    from picklebot_amd.parallel import GradReducer, collectives_retired, setup_distributed

    if os.environ.get("PB_BENCH_FORCE_DP") and "RANK" not in os.environ:
        # exercise the data-parallel code path (process group, bucketed reducer, graph + eager all-reduce) on one GPU
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    dp_path = "RANK" in os.environ and ("PB_BENCH_FORCE_DP" in os.environ or int(os.environ.get("WORLD_SIZE", "1")) > 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not os.environ.get("PB_BENCH_FORCE_DP"):
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    # Data parallel: by default the process group is created AFTER the HIP graph of forward + backward has been captured
    # (early_capture): ProcessGroupNCCL's watchdog thread polls its work objects' events every 100 ms, and on this stack
    # (ROCm 7 / torch 2.10) such a poll while ANY stream of the process is capturing can fail with hipErrorCapturedEvent,
    # invalidate the capture and take the process down in C++ `terminate` - seen in rounds 3 and 4 with and without
    # collectives inside the graph (gpurun_out/r4ev/dp_3.err: 1 run in 6).  No watchdog, no poll: the capture happens first.
    graph_collectives = bool(os.environ.get("PB_BENCH_GRAPH_COLLECTIVES"))       # opt-in: collectives INSIDE the graph
    early_capture = dp_path and not args.no_graph and not graph_collectives
    if dp_path and not early_capture:
        local_rank, rank, _ = setup_distributed("nccl")
    elif dp_path:
        local_rank, rank = int(os.environ.get("LOCAL_RANK", "0")), int(os.environ["RANK"])
    else:
        local_rank, rank = 0, 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    torch.manual_seed(1234)                                   # train.py:196
    model = picklebot_amd.MobileNetLarge3D(NCLS).to(device)
    model.initialize_weights()                                # train.py:177 (a no-op for this model)
    model.train()
    reducer = GradReducer(model, bind_later=early_capture) if dp_path else None
    use_graph = not dp_path and not args.no_graph
    flat_opt = not args.torch_adamw
    if flat_opt:
        # AdamW (train.py:208-212, plain fp32 state) as one fused launch per flat bucket; in data parallel it adopts
        # the reducer's gradient buckets
        opt = picklebot_amd.FlatAdamW(reducer if reducer is not None else model.parameters(), lr=3e-4, weight_decay=5e-4)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=3e-4, weight_decay=5e-4, fused=True, capturable=use_graph)
        if not dp_path:
            # this loop reads no gradient before backward() has returned: opt in to the weight-gradient side stream
            # (FlatAdamW / GradReducer opt their parameters in themselves; the library's default is in line)
            K.weight_grad_overlap().__enter__()
    x, y = synthetic_batch(args.batch, device, 1234 + rank)

    def opt_step_reduced():
        if flat_opt:
            opt.step(packed=True)         # the reducer's buckets already hold the (reduced) gradients
        else:
            opt.step()

    def step():
        if reducer is not None:
            reducer.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):    # train.py:264
            loss = torch.nn.functional.cross_entropy(model(x).float(), y)
        loss.backward()
        if reducer is not None:
            reducer.finish()
            opt_step_reduced()
        else:
            opt.step()
        return loss

    def sync():
        if dp_path:
            dist.barrier()
        torch.cuda.synchronize()

    run_stream = torch.cuda.Stream(device)      # warm-up, capture, replay and the profiling pass all run on this ONE stream:
    torch.cuda.set_stream(run_stream)           # autograd's AccumulateGrad nodes are created under the stream they later run on

    def capture(fn, has_collective=False):
        """Warm up, then capture fn() into a HIP graph on the same stream; returns (graph, fn's result)."""
        for _ in range(3):
            fn()
        K.launch_count[0] = 0                 # counts the C-ABI calls of the captured step
        torch.cuda.synchronize()
        if dp_path and has_collective:
            # RCCL's watchdog thread polls the end events of the eager warm-up collectives; on ROCm that query fails with
            # hipErrorCapturedEvent once RCCL's stream has joined a capture and the watchdog takes the process down (round 3,
            # 1 run in ~10).  Hand-off by ORDER (parallel.collectives_retired): the device is idle, so the watchdog's next poll
            # retires every work it holds; capture starts only after the flight recorder shows it HAS.  Otherwise: no
            # collective inside a graph (the caller falls back, all ranks together).
            if collectives_retired() is not True:
                raise RuntimeError("RCCL watchdog still holds eager work objects (or cannot be asked): not capturing a collective")
        g = torch.cuda.CUDAGraph()
        K.reset_zero_arena()              # the zero-filled scratch must be allocated AND filled inside the graph
        # (data parallel: thread-local capture mode - RCCL's watchdog thread queries events of eager collectives meanwhile)
        with torch.cuda.graph(g, stream=run_stream, capture_error_mode="thread_local" if dp_path else "global"):
            out = fn()
        K.reset_zero_arena()
        return g, out

    def all_ranks_ok(ok):
        """Launch-mode decisions are COLLECTIVE: every rank reports whether its capture attempt succeeded and all ranks take
        the fallback together when any one failed (ranks in different modes would issue collectives from different places)."""
        if not dp_path:
            return ok
        t = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def recover(what, e):
        print(f"[bench] rank {rank}: {what} failed ({type(e).__name__}: {e})", file=sys.stderr)
        try:
            torch.cuda.synchronize()
            K.reset_zero_arena()
        except Exception as e2:          # a rank that cannot get back to a clean state must not continue in a divergent mode
            print(f"[bench] rank {rank}: cannot recover after failed capture ({e2}); aborting", file=sys.stderr)
            os._exit(3)

    # One training step = ~340 kernel launches through ctypes (6-8 ms of host time on a quiet box, 20+ ms on a
    # busy one, against ~12 ms of GPU work): on one GPU the whole step (forward, loss, backward, fused AdamW) is
    # captured once into a HIP graph and replayed; every replay does the full work on the same buffers.
    launch_mode = "eager"
    eager_step = step
    abi_calls = None
    if use_graph:
        try:
            opt.zero_grad(set_to_none=True)
            graph, graph_loss = capture(step)
            abi_calls = K.launch_count[0]                      # (capture() = 3 eager warm-up steps + the captured one: all equal)

            def step():  # noqa: F811
                graph.replay()
                return graph_loss
            launch_mode = "hipgraph"
        except Exception as e:  # capture is an optimisation: fall back to eager submission
            recover("HIP graph capture", e)
            step = eager_step
    elif early_capture:
        # data parallel, default.  forward + backward (gradients handed over as they are, one multi-tensor copy per bucket at
        # the end) captured into a HIP graph BEFORE the process group exists; then the group is created, the replicas are
        # synchronised (in place), and a step = graph replay + bucket all-reduces over RCCL + optimizer.
        def fwd_bwd():
            reducer.detach_grads()
            with reducer.no_sync():
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = torch.nn.functional.cross_entropy(model(x).float(), y)
                loss.backward()
            reducer.pack_grads()
            return loss

        ok, graph = True, None
        try:
            graph, graph_loss = capture(fwd_bwd)
        except Exception as e:
            ok = False
            recover("HIP graph capture (fwd+bwd, before the process group)", e)
        setup_distributed("nccl")                         # the watchdog thread starts here - after the capture
        reducer.bind()                                    # world size + broadcast of rank 0's parameters and buffers
        reducer.attach_grads()
        reducer.zero_grad()
        step()                                            # one eager step: creates the RCCL communicator
        sync()
        log("process group up, first eager data-parallel step done")
        if all_ranks_ok(ok):                              # collective: one rank without a graph -> nobody replays
            def step():  # noqa: F811
                graph.replay()
                reducer.reduce_now()
                opt_step_reduced()
                return graph_loss
            launch_mode = "hipgraph(fwd+bwd, captured before the process group exists) + eager RCCL all-reduce per bucket + optimizer"
        else:
            graph = None
            step = eager_step
    elif dp_path and not args.no_graph:
        # data parallel, opt-in (PB_BENCH_GRAPH_COLLECTIVES=1).  First choice: the WHOLE step in one HIP graph, the exchange included - backward hands its gradients
        # over as they are, the hook of a bucket's last gradient packs the bucket and issues its RCCL all-reduce on RCCL's
        # stream (a fork inside the capture), so bucket k travels over xGMI while the rest of backward replays; the
        # optimizer joins behind the last reduction.  Second choice (capture of the collective refused): forward +
        # backward from a graph, then the bucket all-reduces and the optimizer eagerly (no overlap).  Third: eager.
        # After every attempt the ranks agree (all-reduce MIN of a success flag) and fall back TOGETHER.
        def overlapped_step():
            reducer.adopt_grads()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = torch.nn.functional.cross_entropy(model(x).float(), y)
            loss.backward()
            reducer.finish()
            opt_step_reduced()
            return loss

        def fwd_bwd():
            reducer.detach_grads()                        # backward hands its gradient tensors over as they are ...
            with reducer.no_sync():
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = torch.nn.functional.cross_entropy(model(x).float(), y)
                loss.backward()
            reducer.pack_grads()                          # ... and one concatenation per bucket gathers them
            return loss

        step()                                            # creates the RCCL communicator outside any capture
        sync()
        log("first eager data-parallel step done")
        captured = False
        # The whole-step graph WITH the collectives inside is OPT-IN (PB_BENCH_GRAPH_COLLECTIVES=1).  Round 4 saw RCCL's watchdog
        # thread abort the process twice more (hipErrorCapturedEvent from WorkNCCL::isCompleted while RCCL's stream was part of
        # the capture), once WITH the retired-work hand-off below in place - so the hand-off is necessary but not sufficient on
        # this stack, and a bench that dies in C++ `terminate` on the 8-GPU node measures nothing.  Default: forward + backward
        # replayed from a graph that contains NO collective (RCCL's stream never joins a capture), then the bucket
        # all-reduces and the optimizer eagerly.
        if flat_opt and os.environ.get("PB_BENCH_GRAPH_COLLECTIVES"):
            ok, graph = True, None
            try:
                graph, graph_loss = capture(overlapped_step, has_collective=True)
            except Exception as e:
                ok = False
                recover("capture with in-graph all-reduce", e)
            if all_ranks_ok(ok):
                def step():  # noqa: F811
                    graph.replay()
                    return graph_loss
                launch_mode = "hipgraph(whole step; RCCL all-reduce per bucket on a forked stream, overlapped with backward)"
                captured = True
                eager_step = overlapped_step
            else:
                graph = None
                reducer.reset()
                reducer.adopt_grads(False)
                reducer.attach_grads()
                reducer.zero_grad()
        if not captured:
            ok, graph = True, None
            try:
                graph, graph_loss = capture(fwd_bwd)
            except Exception as e:
                ok = False
                recover("HIP graph capture (fwd+bwd)", e)
            if all_ranks_ok(ok):
                reducer.attach_grads()                    # the optimiser reads the (reduced) buckets from now on

                def step():  # noqa: F811
                    graph.replay()
                    reducer.reduce_now()
                    opt_step_reduced()
                    return graph_loss
                launch_mode = "hipgraph(fwd+bwd) + eager all-reduce/optimizer"
            else:
                graph = None
                reducer.reset()
                reducer.attach_grads()
                reducer.zero_grad()
                step = eager_step
    log("launch mode:", launch_mode)
    for _ in range(args.warmup):
        step()
    sync()
    log("warm-up done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3     # host time to submit a step (GPU runs behind)
    sync()
    elapsed = time.perf_counter() - t0
    if dp_path:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = args.batch * world * args.steps / elapsed
    side_branches = K.branch_counts["side"]       # > 0: the captured / submitted steps used the second stream

    # ---- per-kernel roofline: the same K steps again with HIP events around every launch --------
    roofline, families = None, {}
    if rank == 0:
        prof_step = eager_step                                              # HIP events need eager launches
        if dp_path:
            reducer.adopt_grads(False)
            reducer.attach_grads()

            def prof_step():      # rank 0 profiles alone: no collective may be issued here (the other ranks are waiting)
                reducer.zero_grad()
                with reducer.no_sync():
                    with torch.autocast("cuda", dtype=torch.bfloat16):
                        loss = torch.nn.functional.cross_entropy(model(x).float(), y)
                    loss.backward()
                opt_step_reduced()
                return loss
        # (in line: with the weight gradients on their side stream - kernels.weight_grad_branch, how the timed steps ran -
        # two kernels share the chip and an event pair brackets both; the per-kernel numbers are about each kernel alone)
        overlap, K._WGRAD_OVERLAP = K._WGRAD_OVERLAP, False
        K.start_profile()
        for _ in range(args.steps):
            prof_step()
        prof = K.stop_profile()
        K._WGRAD_OVERLAP = overlap
        detail = {k: v for k, v in prof.items() if " | " in k}
        prof = {k: v for k, v in prof.items() if " | " not in k}
        if os.environ.get("PB_BENCH_DETAIL"):
            with open(os.environ["PB_BENCH_DETAIL"], "w") as f:
                for k, d in sorted(detail.items(), key=lambda kv: -kv[1]["ms"]):
                    f.write("%-90s n=%3d  %8.1f us/launch  %7.1f GB/s\n" % (
                        k, d["launches"] // args.steps, d["ms"] / d["launches"] * 1e3, d["bytes"] / 1e6 / max(d["ms"], 1e-9)))
        tot_ms = sum(d["ms"] for d in prof.values())
        for name, d in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
            gbs = d["bytes"] / 1e9 / (d["ms"] / 1e3) if d["ms"] > 0 else 0.0
            families[name] = {"launches_per_step": d["launches"] // args.steps, "ms_per_step": round(d["ms"] / args.steps, 4),
                              "GBps": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4),
                              "share_of_kernel_time": round(d["ms"] / tot_ms, 4)}
        # The dominant KERNEL, by name (the per-mode split of `ew_kernel` hid that the BatchNorm / activation passes
        # are the largest consumer): families that are template modes of one kernel (`ew_kernel<...>`) are added up.
        by_name = {}
        for name, d in prof.items():
            b = by_name.setdefault(name.split("<")[0], dict(launches=0, ms=0.0, bytes=0))
            for k in ("launches", "ms", "bytes"):
                b[k] += d[k]
        dom = max(by_name, key=lambda k: by_name[k]["ms"])
        d = by_name[dom]
        ach = d["bytes"] / 1e9 / (d["ms"] / 1e3)
        # HBM bytes per launch of the dominant kernel from the PMC summary (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this
        # same command, corrected as MI355X_MICROARCH.md prescribes; tools/profile_round.sh + tools/pmc_summary.py).  Only a
        # summary collected from THIS kernel code counts (its csrc digest must match): otherwise null, never a stale number.
        traffic = None
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from pmc_summary import csrc_digest
            pmc = json.load(open(os.environ.get("PB_PMC_TRAFFIC", os.path.join(ROOT, "profiles", "r04_pmc_traffic.json"))))
            if pmc.get("meta", {}).get("csrc_digest") == csrc_digest():
                if dom in pmc["kernels"]:
                    traffic = pmc["kernels"][dom]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "bytes_per_launch": int(d["bytes"] / d["launches"]), "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2),
                    "launches": d["launches"],
                    "share_of_kernel_time": round(d["ms"] / tot_ms, 4),
                    "measured_over": f"{args.steps} steps re-run in line (weight-gradient side stream off) with HIP events around every launch; "
                                     "the kernel with the largest total time BY NAME (template modes added up)"}
        # the depthwise-3D path as a whole (north_star target: >= 0.50 of the HBM roofline)
        dw = [v for k, v in prof.items() if k.startswith("dw")]
        if dw:
            dwb, dwms = sum(v["bytes"] for v in dw), sum(v["ms"] for v in dw)
            roofline["depthwise_path"] = {"GBps": round(dwb / 1e9 / (dwms / 1e3), 1), "frac": round(dwb / 1e9 / (dwms / 1e3) / HBM_PEAK_GBS, 4),
                                          "ms_per_step": round(dwms / args.steps, 4), "launches_per_step": sum(v["launches"] for v in dw) // args.steps}
    if dp_path:
        dist.barrier()
    log("profiling pass done")

    if rank == 0:
        out = {
            "metric": "clips/sec fwd+bwd, MobileNetLarge3D 16x224x224 bf16", "value": round(value, 2), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "MobileNetLarge3D (mobilenet_large.json) training step, synthetic 16x224x224 clips",
                       "clips_per_gpu": args.batch, "global_batch": args.batch * world, "num_classes": NCLS,
                       "precision": "fp32 params, bf16 activations (autocast)", "parallelism": f"dp{world}",
                       "step": "fwd + CE loss + bwd + RCCL grad all-reduce (n>1) + fused AdamW (%s)" % ("picklebot_amd.FlatAdamW" if flat_opt else "torch fused"),
                       "launch": launch_mode,
                       "c_abi_calls_per_step": abi_calls,      # entry points called per replayed step (a few launch 2-3 kernels; torch adds ~15 launches of its own)
                       "streams": ("each block's weight gradients on a second HIP stream beside the next block's data-gradient chain, "
                                   "joined when backward ends (kernels.weight_grad_branch; opted in by FlatAdamW / GradReducer)") if side_branches else "one",

                       "weights": "random init (seed 1234)", "loss": round(float(loss.item()), 4)},
            "roofline": roofline,
            "host_enqueue_ms_per_step": round(host_enqueue_ms, 3),
            "kernel_families": families,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        line = json.dumps(out)
    else:
        line = None
    log("result ready")
    if line is not None:
        sys.stdout.flush()
        print(line, flush=True)           # the ONE JSON line - BEFORE the process group is torn down (a teardown that
        line = None                       # misbehaves must not cost the measurement)
    if dp_path:
        os.dup2(2, 1)                     # whatever RCCL prints while it shuts down goes to stderr: the JSON line stays last on stdout
        dist.destroy_process_group()
    log("process group destroyed")
    if line is not None:
        sys.stdout.flush()
        print(line, flush=True)           # the ONE JSON line, after anything RCCL may have written


if __name__ == "__main__":
    main()
