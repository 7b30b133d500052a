"""CPU (gloo, world_size 2): the gradient reducer against a manual average of per-rank gradients,
the no_sync() accumulation path, and replica broadcast at construction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(12, 32), nn.ReLU(), nn.Linear(32, 32), nn.ReLU(), nn.Linear(32, 5))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from picklebot_amd.parallel import GradReducer, setup_distributed
    lr, r, w = setup_distributed("gloo")
    assert (lr, r, w) == (rank, rank, world)
    model = _model(seed=100 + rank)                      # replicas start DIFFERENT; ctor must broadcast rank 0
    red = GradReducer(model, bucket_bytes=2048)          # several buckets
    assert len(red.buckets) > 1
    ref = _model(seed=100)
    for a, b in zip(model.parameters(), ref.parameters()):
        assert torch.equal(a, b)
    torch.manual_seed(7)
    data = torch.randn(world, 3, 6, 12)                  # [rank][micro-step]
    target = torch.randn(world, 3, 6, 5)

    def local_grads(steps):
        m = _model(seed=100)
        for s in steps:
            nn.functional.mse_loss(m(data[rank, s]), target[rank, s]).backward()
        return [p.grad.clone() for p in m.parameters()]

    # 1) one step, reduced
    red.zero_grad()
    nn.functional.mse_loss(model(data[rank, 0]), target[rank, 0]).backward()
    red.finish()
    mine = [p.grad.clone() for p in model.parameters()]
    want = local_grads([0])
    gathered = [[torch.zeros_like(g) for _ in range(world)] for g in want]
    for g, outs in zip(want, gathered):
        dist.all_gather(outs, g)
    for got, outs in zip(mine, gathered):
        assert torch.allclose(got, sum(outs) / world, atol=1e-6), "averaged gradient mismatch"
    # 2) accumulation: two micro-steps under no_sync, third reduces the sum
    red.zero_grad()
    with red.no_sync():
        for s in (0, 1):
            nn.functional.mse_loss(model(data[rank, s]), target[rank, s]).backward()
    nn.functional.mse_loss(model(data[rank, 2]), target[rank, 2]).backward()
    red.finish()
    mine = [p.grad.clone() for p in model.parameters()]
    want = local_grads([0, 1, 2])
    gathered = [[torch.zeros_like(g) for _ in range(world)] for g in want]
    for g, outs in zip(want, gathered):
        dist.all_gather(outs, g)
    for got, outs in zip(mine, gathered):
        assert torch.allclose(got, sum(outs) / world, atol=1e-6), "accumulated gradient mismatch"
    # 3) the HIP-graph form of a step: backward with the hooks off hands its gradients over as they are
    #    (detach_grads), one concatenation per bucket packs them, reduce_now() exchanges, attach_grads() points the
    #    optimiser at the buckets
    red.detach_grads()
    with red.no_sync():
        nn.functional.mse_loss(model(data[rank, 1]), target[rank, 1]).backward()
    assert all(p.grad is not None and p.grad.untyped_storage().data_ptr() != red.flat[0].untyped_storage().data_ptr()
               for p in red.buckets[0])
    red.pack_grads()
    red.reduce_now()
    red.attach_grads()
    mine = [p.grad.clone() for p in model.parameters()]
    want = local_grads([1])
    gathered = [[torch.zeros_like(g) for _ in range(world)] for g in want]
    for g, outs in zip(want, gathered):
        dist.all_gather(outs, g)
    for got, outs in zip(mine, gathered):
        assert torch.allclose(got, sum(outs) / world, atol=1e-6), "packed gradient mismatch"
    # .grad tensors are views of the flat buckets again
    spans = [(f.data_ptr(), f.data_ptr() + f.numel() * f.element_size()) for f in red.flat]
    assert all(any(lo <= p.grad.data_ptr() < hi for lo, hi in spans) for p in model.parameters())
    # 3b) adopt mode: .grad = None before backward, every bucket packed + reduced from the hook of its last gradient
    red.adopt_grads()
    nn.functional.mse_loss(model(data[rank, 2]), target[rank, 2]).backward()
    red.finish()
    want = local_grads([2])
    gathered = [[torch.zeros_like(g) for _ in range(world)] for g in want]
    for g, outs in zip(want, gathered):
        dist.all_gather(outs, g)
    flat_views = [v for views in red.views for v in views]
    params_in_order = [p for plist in red.buckets for p in plist]
    by_param = {id(p): v for p, v in zip(params_in_order, flat_views)}
    for p, outs in zip(model.parameters(), gathered):
        assert torch.allclose(by_param[id(p)], sum(outs) / world, atol=1e-6), "adopt-mode bucket mismatch"
    red.adopt_grads(False)
    red.attach_grads()
    # 4) a BatchNorm-bearing replica: BatchNorm is per rank (no SyncBN), so
    #    train mode: the reduced gradient == the average of the per-shard gradients, each with its OWN batch statistics;
    #    eval mode : the reduced gradient == the single-process gradient on the concatenated batch
    def bn_model():
        torch.manual_seed(5)
        return nn.Sequential(nn.Conv3d(3, 6, 1), nn.BatchNorm3d(6), nn.Hardswish(), nn.AdaptiveAvgPool3d(1), nn.Flatten(),
                             nn.Linear(6, 4))
    torch.manual_seed(11)
    clips = torch.randn(world, 4, 3, 2, 5, 5)
    labels = torch.randint(0, 4, (world, 4))
    for train in (True, False):
        m = bn_model().train(train)
        rd = GradReducer(m, bucket_bytes=64)
        rd.zero_grad()
        # mean over the LOCAL shard; the reducer averages over ranks = mean over the global batch (equal shard sizes)
        nn.functional.cross_entropy(m(clips[rank]), labels[rank]).backward()
        rd.finish()
        got = [p.grad.clone() for p in m.parameters()]
        if train:
            shard = []
            for r in range(world):
                ref_m = bn_model().train()
                nn.functional.cross_entropy(ref_m(clips[r]), labels[r]).backward()
                shard.append([p.grad.clone() for p in ref_m.parameters()])
            want = [sum(gs) / world for gs in zip(*shard)]
        else:
            ref_m = bn_model().eval()
            nn.functional.cross_entropy(ref_m(clips.flatten(0, 1)), labels.flatten()).backward()
            want = [p.grad.clone() for p in ref_m.parameters()]
        for a, b in zip(got, want):
            assert torch.allclose(a, b, atol=1e-6, rtol=1e-5), ("BatchNorm replica", train, (a - b).abs().max())
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]


def _worker_bind_later(rank, world, port, q):
    """bench.py's data-parallel order (round 4): reducer and `recorded step` first, process group afterwards."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from picklebot_amd.parallel import GradReducer, setup_distributed
    assert not dist.is_initialized()
    model = _model(seed=200 + rank)                      # replicas start DIFFERENT
    red = GradReducer(model, bucket_bytes=2048, bind_later=True)      # no process group yet: no collective may be issued
    torch.manual_seed(9)
    data, target = torch.randn(world, 6, 12), torch.randn(world, 6, 5)
    # the "captured" part of a step works without a process group: backward with the hooks off + one copy per bucket
    red.detach_grads()
    with red.no_sync():
        nn.functional.mse_loss(model(data[rank]), target[rank]).backward()
    red.pack_grads()
    try:
        red.reduce_now()
        raise AssertionError("reduce_now() before bind() must refuse")
    except RuntimeError as e:
        assert "bind()" in str(e)
    setup_distributed("gloo")                            # the process group (and, with RCCL, its watchdog thread) starts HERE
    red.bind()                                           # world size + broadcast of rank 0's parameters
    ref = _model(seed=200)
    assert red.world == world and all(torch.equal(a, b) for a, b in zip(model.parameters(), ref.parameters()))
    # a full step in the replay form after binding
    red.detach_grads()
    with red.no_sync():
        nn.functional.mse_loss(model(data[rank]), target[rank]).backward()
    red.pack_grads()
    red.reduce_now()
    red.attach_grads()
    mine = [p.grad.clone() for p in model.parameters()]
    m = _model(seed=200)
    nn.functional.mse_loss(m(data[rank]), target[rank]).backward()
    want = [p.grad.clone() for p in m.parameters()]
    gathered = [[torch.zeros_like(g) for _ in range(world)] for g in want]
    for g, outs in zip(want, gathered):
        dist.all_gather(outs, g)
    for got, outs in zip(mine, gathered):
        assert torch.allclose(got, sum(outs) / world, atol=1e-6), "bind_later: averaged gradient mismatch"
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_grad_reducer_bound_after_the_step_was_recorded_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bind_later, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]
