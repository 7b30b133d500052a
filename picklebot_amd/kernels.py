"""Tensor-level wrappers over the C-ABI (one function per entry point of include/picklebot_amd.h).

Each wrapper validates shapes/dtypes/contiguity on the host (a wrong shape must never reach a
hand-written kernel), allocates outputs with torch (plumbing) and launches on the current HIP
stream of the tensor's device.  Activations are 2-D row matrices [M, C] or 5-D NDHWC tensors.
"""
import ctypes
import os
import threading
import weakref

import torch

from . import _C

REPL = _C.STAT_SLOTS        # stats buffers are [REPL, 2, C]: one slot per producing workgroup (include/picklebot_amd.h)
STEM_REPL = 2048            # the stem's statistics: its launch wants 2048 workgroups (16 channels: 256 KB)

ACT = {"none": _C.ACT_NONE, "relu": _C.ACT_RELU, "leaky_relu": _C.ACT_LEAKY_RELU, "hardswish": _C.ACT_HSWISH,
       "hardsigmoid": _C.ACT_HSIGMOID, "silu": _C.ACT_SILU}


# ---- zero-initialised fp32 scratch (statistics, pools, gradient accumulators) -------------------
class _ZeroArena:
    """Hands out zero-filled fp32 buffers carved from 16 MB chunks that are zeroed by ONE fill kernel.
    A training step needs ~90 small zero buffers (BN statistics slots, weight-gradient accumulators);
    as separate torch.zeros calls they were 91 fill launches = 0.44 ms of a 13.4 ms step on MI355X.  Every carved range is handed out once (the offset only advances), so it is
    zero when handed out; a chunk is freed by the allocator when its last view dies."""
    CHUNK = 16 << 20                 # bytes

    def __init__(self):
        self._lock = threading.Lock()
        self._buf = {}               # (device, stream) -> [flat tensor, offset in elements]

    def take(self, numel, device):
        n = (int(numel) + 63) // 64 * 64                      # 256-byte granules keep every view 16 B aligned
        if n * 4 > self.CHUNK // 2:
            return torch.zeros(int(numel), dtype=torch.float32, device=device)
        # one chunk per STREAM: the fill kernel runs on the stream that is current when the chunk is made, and only work
        # enqueued on that stream afterwards is ordered behind it (weight_grad_branch launches from a second stream)
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        with self._lock:
            ent = self._buf.get(key)
            if ent is None or ent[1] + n > ent[0].numel():
                ent = [torch.zeros(self.CHUNK // 4, dtype=torch.float32, device=device), 0]
                self._buf[key] = ent
            off = ent[1]
            ent[1] = off + n
            return ent[0][off:off + int(numel)]


_arena = _ZeroArena()


def reset_zero_arena():
    """Forget the current chunk (call after capturing a HIP graph: the chunk belongs to the graph's memory pool)."""
    with _arena._lock:
        _arena._buf.clear()
_USE_ARENA = os.environ.get("PB_ZERO_ARENA", "1") != "0"


def zeros(shape, device):
    """Zero-filled fp32 tensor of `shape` on `device` (from the arena above)."""
    shape = tuple(shape) if not isinstance(shape, int) else (shape,)
    n = 1
    for d in shape:
        n *= int(d)
    if torch.device(device).type != "cuda" or not _USE_ARENA:
        return torch.zeros(shape, dtype=torch.float32, device=device)
    return _arena.take(n, torch.device(device)).view(shape)


# ---- optional per-launch profiling (bench.py): HIP events recorded on the launch stream -------
_spans = None


def start_profile():
    """Record (family, algorithmic bytes, start event, end event) for every kernel launch."""
    global _spans
    _spans = []


def stop_profile():
    """-> {family: dict(launches, ms, bytes)}; synchronises the device once."""
    global _spans
    spans, _spans = _spans, None
    torch.cuda.synchronize()
    out = {}
    for name, nbytes, e0, e1, detail in spans or []:
        ms = e0.elapsed_time(e1)
        for key in (name, name + " | " + detail):
            d = out.setdefault(key, dict(launches=0, ms=0.0, bytes=0))
            d["launches"] += 1
            d["ms"] += ms
            d["bytes"] += nbytes
    return out


def _nb(*ts):
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


_SYNC_DEBUG = os.environ.get("PB_SYNC_DEBUG", "0") != "0"    # print every launch and synchronise after it (fault / hang hunting)


launch_count = [0]          # C-ABI calls so far (bench.py reports how many one captured step makes)


def _launch(family, nbytes, fn, *args, detail=""):
    """Algorithmic bytes = every external input/output tensor once at its stored dtype."""
    launch_count[0] += 1
    if _SYNC_DEBUG:
        print(f"[pb] {fn} {family} {detail}", flush=True)
        _C.call(fn, *args)
        torch.cuda.synchronize()
        return
    if _spans is None:
        _C.call(fn, *args)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _C.call(fn, *args)
    e1.record()
    _spans.append((family, nbytes, e0, e1, detail))


def _dt(t):
    if t.dtype == torch.float32:
        return _C.PB_F32
    if t.dtype == torch.bfloat16:
        return _C.PB_BF16
    raise TypeError(f"picklebot_amd: unsupported activation dtype {t.dtype} (float32 or bfloat16)")


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _dev_stream(t):
    if not t.is_cuda:
        raise RuntimeError("picklebot_amd kernels need tensors on a HIP device (no CPU fallback exists)")
    idx = t.device.index if t.device.index is not None else torch.cuda.current_device()
    return idx, ctypes.c_void_p(torch.cuda.current_stream(idx).cuda_stream)


def _f32(t, shape=None, name="tensor"):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise TypeError(f"{name}: expected contiguous float32, got {t.dtype} contiguous={t.is_contiguous()}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def _rows(t, name="activation"):
    if t.dim() != 2 or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous [M, C] matrix, got shape {tuple(t.shape)} strides {t.stride()}")
    return t


# ------------------------------------------------------------------------------ fixed-order reduction of partials
def _scratch(nbytes, device):
    """Uninitialised fp32 workspace of at least `nbytes` (16-byte aligned: allocator blocks are 512-byte aligned)."""
    pad = int(os.environ.get("PB_SCRATCH_PAD", "0"))
    if pad:      # diagnostic: sentinel-filled slack behind the workspace (tools/probe_block.py checks it)
        t = torch.full((max(int(nbytes) // 4, 4) + pad // 4,), 12345.0, dtype=torch.float32, device=device)
        _pads.append(t[max(int(nbytes) // 4, 4):])
        return t[:max(int(nbytes) // 4, 4)]
    return torch.empty(max(int(nbytes) // 4, 4), dtype=torch.float32, device=device)


_pads = []


def partials_reduce(jobs, device):
    """jobs: list of (part, out, n, stride, split, accumulate) - out[e] (+)= sum_s part[s*stride + e], slots in a fixed
    order, ALL jobs in one launch (pb_partials_reduce)."""
    arr = (_C.ReduceJob * len(jobs))()
    keep = []
    for i, (part, out, n, stride, split, acc) in enumerate(jobs):
        arr[i].part, arr[i].out = part.data_ptr(), out.data_ptr()
        arr[i].n, arr[i].stride, arr[i].split, arr[i].accumulate = int(n), int(stride), int(split), int(acc)
        keep.append((part, out))
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = ctypes.c_void_p(torch.cuda.current_stream(idx).cuda_stream)
    _launch("partials_reduce", sum(j[2] * 4 * (j[4] + 1) for j in jobs), "pb_partials_reduce", arr, len(jobs), idx, st)


# ------------------------------------------------------------------------------ side stream for the weight gradients
# A block's backward is a chain (BatchNorm backward -> data gradient of the projection -> squeeze-excite -> depthwise
# data gradient -> data gradient of the expansion) with the weight gradients hanging off it: nothing in the chain, and
# nothing in the blocks before it, reads them.  `weight_grad_branch` launches a block's weight-gradient kernels and
# their fixed-order reduction as ONE branch on a second HIP stream, ordered behind everything the chain has enqueued
# (one event fork per block), and does NOT join it: the branch runs beside the NEXT block's chain, whose many small
# launches (finalize, squeeze-excite, slot sums: single-digit microseconds with the chip mostly idle) it fills.
# Measured on the bench step: 7.16 -> 6.78 ms.  (Forking per kernel and joining inside the block was 3 % SLOWER than no
# overlap at all: every cross-stream edge of the replayed graph costs a few microseconds.)
# Same kernels, same order within the branch, no atomics: the gradients are bit-identical.
#
# What makes the missing join safe:
#   * memory: `keep` holds every tensor the branch reads or writes until the calling stream has been ordered behind the
#     branch - one block later (normally long finished, so the wait is free), or at the join;
#   * consumers: the join is queued as an autograd-engine callback, i.e. it runs when the backward pass ends, before
#     `loss.backward()` returns; FlatAdamW / GradReducer join before they read a gradient (bucket packing inside hooks);
#   * the caller returns FRESH VIEWS of the gradient buffers, not the tensor objects it lists in `keep`: autograd adopts a
#     gradient only when nobody else holds the object, otherwise it clones it - on the calling stream, at once;
#   * a block only defers when autograd will merely ADOPT its gradient tensors - `.grad is None`, no Python hooks - AND
#     the consumer of the gradients has OPTED IN (round 4).  Whoever reads `.grad` DURING backward must know about the
#     branch, and the hooks that do so cannot all be seen from here: stock DistributedDataParallel (train.py:203-204) and
#     anything else that hooks the AccumulateGrad NODE (`grad_acc.register_hook`) lives in C++ and leaves `p._backward_hooks`
#     / `p._post_accumulate_grad_hooks` empty - DDP's `mark_variable_ready` would copy `.grad` into its bucket on the
#     calling stream before the side stream has written it.  So the default is IN LINE, and deferral is switched on
#       - per parameter, by a consumer that joins before it reads: `FlatAdamW` and `GradReducer` mark their parameters
#         `_pb_join_aware` (their pack / hooks call `join_weight_grads()` first), or
#       - for a region of code, by `with picklebot_amd.weight_grad_overlap():` - the caller's promise that nothing reads a
#         gradient before `backward()` has returned (plain `loss.backward(); optimizer.step()` loops qualify; a model
#         wrapped in stock DDP does NOT);
#   * a parameter that reaches a second weight-gradient node in the same backward pass (a module called twice, tied
#     weights): autograd's input buffer ADDS the two gradients on the calling stream when the second arrives, so the
#     second block joins every branch in flight first and runs in line.
# PB_WGRAD_OVERLAP=0 switches it off altogether.  Works under HIP-graph capture (event fork / join inside the captured
# region).
_WGRAD_OVERLAP = os.environ.get("PB_WGRAD_OVERLAP", "1") != "0"
_side = {}
_lag = {}        # device index -> (done event, tensors kept alive) of the branch not yet joined
_pending = {}    # device index -> reduction jobs (with the tensors they keep alive) the deferred branches have not launched yet
_flushed = []    # job lists already launched on the side stream: their workspaces live until the next join
_REDUCE_BATCH = int(os.environ.get("PB_REDUCE_BATCH", "24"))     # jobs per reduction launch of the deferred branches (<= 32)
_group = {}      # device index -> [(thunks, keep, held)] of blocks whose branch has not been forked yet
# blocks per branch: every fork / join of the two streams costs the chain ~10 us inside a replayed HIP graph (the kernel in front
# of it carries a completion signal, the one behind it a cross-queue wait)
_WGRAD_GROUP = int(os.environ.get("PB_WGRAD_GROUP", "4"))
_seen = {}       # graph-task id -> ids of the parameters whose gradient a branch of this pass has deferred
branch_counts = {"side": 0, "inline": 0}      # how the blocks' weight gradients were issued (tests read this)


class weight_grad_overlap:
    """`with picklebot_amd.weight_grad_overlap():` around a training loop (or just around `loss.backward()`): the caller
    promises that nothing reads a parameter's `.grad` before `backward()` has returned - no stock DistributedDataParallel,
    no hooks on AccumulateGrad nodes - so every block may run its weight gradients on the second stream.  `FlatAdamW` and
    `GradReducer` opt their own parameters in and need no such region."""
    _depth = 0

    def __enter__(self):
        weight_grad_overlap._depth += 1
        return self

    def __exit__(self, *exc):
        weight_grad_overlap._depth -= 1
        return False


def _side_stream(idx):
    st = _side.get(idx)
    if st is None:
        st = _side[idx] = torch.cuda.Stream(device=idx)
    return st


def _adopting(p, region):
    """autograd will adopt this parameter's gradient tensor as it is AND whoever reads it later joins first"""
    if p is None:
        return True
    if not p.is_leaf or p.grad is not None or getattr(p, "_backward_hooks", None):
        return False          # (a non-leaf weight: the node that produced it consumes the gradient on the calling stream)
    aware = getattr(p, "_pb_join_aware", False)
    if getattr(p, "_post_accumulate_grad_hooks", None) and not aware:
        return False
    return aware or region


_WGRAD_TAGS = set(filter(None, os.environ.get("PB_WGRAD_TAGS", "").split(",")))     # debugging: only these block kinds defer


def _run_inline(thunks, held=()):
    branch_counts["inline"] += 1
    with deferred_reductions(held):
        for t in thunks:
            t()


def weight_grad_branch(ref, params, thunks, keep, tag="", held=(), now=False):
    """Run `thunks` (the weight-gradient launches of one block; their partial-tile reductions are issued as one launch at
    the end) - on the side stream without joining when that is safe (see above), in line otherwise.  held: reduction jobs
    of kernels the CHAIN already launched (held_reductions) - they join the same single reduction launch.  The thunks of
    _WGRAD_GROUP consecutive blocks are issued together, i.e. possibly AFTER this call has returned: they must write into
    tensors that exist already - or pass now=True (the branch, with everything collected before it, is forked at once)."""
    region = weight_grad_overlap._depth > 0          # (process-wide: backward runs on autograd's device thread)
    defer = _WGRAD_OVERLAP and ref.is_cuda and all(_adopting(p, region) for p in params) and (not _WGRAD_TAGS or tag in _WGRAD_TAGS)
    if not defer:
        return _run_inline(thunks, held)
    task = torch._C._current_graph_task_id()
    seen = _seen.setdefault(task, set())
    ids = [id(p) for p in params if p is not None]
    if any(i in seen for i in ids):
        # second gradient for a parameter in this pass: autograd's input buffer is about to ADD it to the first one on the
        # calling stream - order that stream behind the branch that is still writing the first
        join_weight_grads()
        return _run_inline(thunks, held)
    idx = ref.device.index if ref.device.index is not None else torch.cuda.current_device()
    main, side = torch.cuda.current_stream(idx), _side_stream(idx)
    try:
        # joins when this backward pass ends, before backward() returns.  Queued by EVERY branch (the later ones find nothing
        # left to join): a pass that died half-way leaves a stale entry behind, and "first branch of the pass" cannot be told
        # from it
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
    except RuntimeError:                  # not inside a backward pass of the engine: nobody would join
        return _run_inline(thunks, held)
    seen.update(ids)
    grp = _group.setdefault(idx, [])
    grp.append((thunks, keep, held))
    if now or len(grp) >= _WGRAD_GROUP:
        _launch_group(idx, main, side)


def _launch_group(idx, main, side):
    """Fork: the collected blocks' weight-gradient launches as ONE branch of the side stream behind the calling stream."""
    grp = _group.pop(idx, None)
    if not grp:
        return
    prev = _lag.pop(idx, None)
    if prev is not None:
        main.wait_event(prev[0])          # the previous branch (normally long finished); its tensors may go now
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    pend = _pending.setdefault(idx, [])
    with torch.cuda.stream(side):
        # the partial-tile reductions of SEVERAL blocks go out as one launch (whoever reads a gradient joins first, and the
        # join flushes what is left): 23 -> 6 reduction launches per MobileNetLarge3D step
        for thunks, _, held in grp:
            with deferred_reductions(held, pending=pend):
                for t in thunks:
                    t()
        if len(pend) >= _REDUCE_BATCH:
            _flush_pending(idx)
    done = torch.cuda.Event()
    done.record(side)
    _lag[idx] = (done, [(keep, held) for _, keep, held in grp])      # (held: the chain's workspaces the branch's reduction reads)
    branch_counts["side"] += len(grp)


def _flush_pending(idx):
    """One pb_partials_reduce launch (on the current = side stream) for every reduction job the branches of device idx left."""
    jobs = _pending.get(idx)
    if not jobs:
        return
    _pending[idx] = []
    arr = (_C.ReduceJob * len(jobs))(*[j for j, _ in jobs])
    st = ctypes.c_void_p(torch.cuda.current_stream(idx).cuda_stream)
    _launch("partials_reduce", sum(j.n * 4 * (j.split + 1) for j, _ in jobs), "pb_partials_reduce", arr, len(jobs), idx, st)
    _flushed.append(jobs)      # the workspaces stay alive until the calling stream has joined (join_weight_grads)


def join_weight_grads():
    """Order the current stream(s) behind every weight-gradient branch still in flight (cheap when there is none); reduction
    jobs the branches left pending are launched first (behind the last branch, on the side stream)."""
    for idx in list(_group):
        _launch_group(idx, torch.cuda.current_stream(idx), _side_stream(idx))
    for idx in [i for i, jobs in _pending.items() if jobs]:
        side = _side_stream(idx)
        with torch.cuda.stream(side):
            _flush_pending(idx)
        done = torch.cuda.Event()
        done.record(side)
        prev = _lag.pop(idx, None)
        _lag[idx] = (done, prev[1] if prev else None)
    for idx in list(_lag):
        done, _ = _lag.pop(idx)
        torch.cuda.current_stream(idx).wait_event(done)
    del _flushed[:]


def _end_of_backward():
    join_weight_grads()
    _seen.clear()


# A block's backward computes several weight gradients; each leaves per-workgroup partial tiles in a workspace.
# Inside `deferred_reductions()` their fixed-order reductions are collected and issued as ONE launch at exit
# (one launch per bottleneck instead of three).
_deferred = threading.local()


class deferred_reductions:
    def __init__(self, held=(), pending=None):
        self.held = list(held)
        self.pending = pending          # a list: the jobs are appended to it instead of being launched at exit

    def __enter__(self):
        self.prev = getattr(_deferred, "jobs", None)
        _deferred.jobs = list(self.held)
        return self

    def __exit__(self, *exc):
        jobs, _deferred.jobs = _deferred.jobs, self.prev
        if exc[0] is None and jobs and self.pending is not None:
            self.pending.extend(jobs)
            return False
        if exc[0] is None and jobs:
            flush = [(j.part, j.out, j.n, j.stride, j.split, j.accumulate) for j, _ in jobs]
            arr = (_C.ReduceJob * len(jobs))(*[j for j, _ in jobs])
            dev = jobs[0][1][0].device
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            st = ctypes.c_void_p(torch.cuda.current_stream(idx).cuda_stream)
            _launch("partials_reduce", sum(f[2] * 4 * (f[4] + 1) for f in flush), "pb_partials_reduce", arr, len(jobs), idx, st)
        return False


class held_reductions(deferred_reductions):
    """Like deferred_reductions, but nothing is launched at exit: the collected jobs (`.jobs`, with the workspaces they
    keep alive) are handed to a later weight_grad_branch(..., held=jobs), whose single reduction launch adds them up -
    for a kernel of the data-gradient CHAIN that also leaves weight-gradient partials (dw_bwd_fused)."""

    def __exit__(self, *exc):
        self.jobs, _deferred.jobs = _deferred.jobs, self.prev
        return False


def _defer_job():
    """-> (ctypes job or None, byref argument): None when reductions are not being deferred."""
    if getattr(_deferred, "jobs", None) is None or os.environ.get("PB_DEFER", "1") == "0":
        return None, None
    job = _C.ReduceJob()
    return job, ctypes.byref(job)


def _keep_job(job, *tensors):
    if job is not None and job.split > 0:
        _deferred.jobs.append((job, tensors))      # the workspace and the output stay alive until the flush


# ------------------------------------------------------------------------------ pointwise conv
_GEMM_DMA = os.environ.get("PB_GEMM_DMA", "1") != "0"
_GEMM_DMA_ALL = os.environ.get("PB_GEMM_DMA_ALL", "1") != "0"
_K16_FOLD = os.environ.get("PB_K16_FOLD", "1") != "0"


def _dma_gemm_pays(M, K, N):
    """bf16 GEMMs of more than a few rows go to pb_pwconv_fwd_bf16w: the weight-stationary streaming kernel (K <= 160,
    pw_stream) or the LDS-DMA ring kernel.  (Round 1 kept the K = 16 layers on the register-staged kernel, 3.0-3.4 TB/s on
    the 4M-row tensors; the streaming kernel measured 4.2 TB/s there.)  PB_GEMM_DMA_ALL=0 restores that split."""
    if M <= 64:
        return False
    return _GEMM_DMA_ALL or K >= 24


# bf16 weight copies live for one "weight scope" = one top-level module forward and its backward.  Version
# counters cannot be trusted to invalidate them: torch's fused optimisers update parameters without bumping
# `_version` (checked on this stack: AdamW(fused=True) leaves it at 0), so every top-level forward starts a new
# generation and re-derives the copies (one batched launch per model forward, see prepare_weights).
_wcache = {}
_wgen = 0
_wscope_depth = threading.local()


class weight_scope:
    """Entered by every module forward; the outermost one starts a new weight generation."""

    def __enter__(self):
        global _wgen
        d = getattr(_wscope_depth, "d", 0)
        if d == 0:
            _wgen += 1
        _wscope_depth.d = d + 1
        return self

    def __exit__(self, *exc):
        _wscope_depth.d -= 1
        return False


def _prepped_weight(w, transposed, fold4=False):
    """bf16 GEMM weights [N,K] of the fp32 parameter view `w` ([N,K], or [K,N] when `transposed`) for the current
    weight generation; fold4: blockdiag(W,W,W,W) [4N,4K] for the 4-rows-per-line form of K = 16 layers."""
    if fold4:
        key4 = (w.data_ptr(), w.numel(), bool(transposed), "fold4")
        ent = _wcache.get(key4)
        scoped = getattr(_wscope_depth, "d", 0) > 0 or torch._C._current_graph_task_id() != -1
        if scoped and ent is not None and ent[0] == _wgen:
            return ent[1]
        wb = _prepped_weight(w, transposed)
        out = torch.block_diag(wb, wb, wb, wb).contiguous()
        if scoped:
            _wcache[key4] = (_wgen, out)
        return out
    key = (w.data_ptr(), w.numel(), bool(transposed))
    ent = _wcache.get(key)
    # a cached copy is trusted only inside a weight scope or in the backward pass of one (autograd graph task):
    # a bare kernels.pw_fwd call has no generation to tie it to (and the allocator recycles addresses)
    scoped = getattr(_wscope_depth, "d", 0) > 0 or torch._C._current_graph_task_id() != -1
    if scoped and ent is not None and ent[0] == _wgen and ent[1].device == w.device:
        return ent[1]
    R, C = w.shape
    shape = (C, R) if transposed else (R, C)
    out = ent[1] if (scoped and ent is not None and tuple(ent[1].shape) == shape and ent[1].device == w.device) else \
        torch.empty(shape, dtype=torch.bfloat16, device=w.device)
    dev, st = _dev_stream(w)
    _launch("pw_weight_prep_kernel", _nb(w, out), "pb_pw_weight_prep", _p(w), R, C, int(bool(transposed)), None, 1, _p(out), dev, st)
    if scoped:
        _wcache[key] = (_wgen, out)
    return out


class WeightSet:
    """All pointwise weights of a model, converted (both orientations) by ONE launch per forward."""

    def __init__(self):
        self.ptrs = None
        self.desc = None
        self.outs = None

    def prepare(self, weights):
        """weights: list of fp32 [N,K] parameter views (N*K % 8 == 0).  Must be called inside a weight_scope."""
        weights = [w for w in weights if w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.is_contiguous()
                   and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0]
        if not weights:
            return
        ptrs = [w.data_ptr() for w in weights]
        if ptrs != self.ptrs:
            rows, outs = [], []
            for w in weights:
                N, K = w.shape
                o0 = torch.empty((N, K), dtype=torch.bfloat16, device=w.device)
                o1 = torch.empty((K, N), dtype=torch.bfloat16, device=w.device)
                rows.append([w.data_ptr(), o0.data_ptr(), (N << 32) | K, 0])
                rows.append([w.data_ptr(), o1.data_ptr(), (N << 32) | K, 1])
                f0 = f1 = None
                # the K = 16 GEMMs run as K = 64 GEMMs on blockdiag(W,W,W,W) (pw_fwd): forward when this layer reads 16
                # channels (K == 16, 4N <= 96), backward-data when it writes 16 (N == 16, 4K <= 96)
                if _K16_FOLD and K == 16 and 4 * N <= 96:
                    f0 = torch.empty((4 * N, 4 * K), dtype=torch.bfloat16, device=w.device)
                    rows.append([w.data_ptr(), f0.data_ptr(), (N << 32) | K, 2])
                if _K16_FOLD and N == 16 and 4 * K <= 96:
                    f1 = torch.empty((4 * K, 4 * N), dtype=torch.bfloat16, device=w.device)
                    rows.append([w.data_ptr(), f1.data_ptr(), (N << 32) | K, 3])
                outs.append((o0, o1, f0, f1))
            self.desc = torch.tensor(rows, dtype=torch.int64, device=weights[0].device)
            self.ptrs, self.outs = ptrs, outs
        dev, st = _dev_stream(weights[0])
        _launch("pw_weight_prep_batch_kernel", sum(w.numel() * 8 for w in weights), "pb_pw_weight_prep_batch", _p(self.desc),
                self.desc.shape[0], dev, st)
        for w, (o0, o1, f0, f1) in zip(weights, self.outs):
            _wcache[(w.data_ptr(), w.numel(), False)] = (_wgen, o0)      # forward: GEMM weights [N,K] = w
            _wcache[(w.data_ptr(), w.numel(), True)] = (_wgen, o1)       # backward-data: GEMM weights [K,N] = w^T
            if f0 is not None:
                _wcache[(w.data_ptr(), w.numel(), False, "fold4")] = (_wgen, f0)
            if f1 is not None:
                _wcache[(w.data_ptr(), w.numel(), True, "fold4")] = (_wgen, f1)



# ------------------------------------------------------------------------------ inference BatchNorm folds
# (scale, shift, mean, rstd) of every eval-mode BatchNorm of a model, computed by ONE launch per top-level forward and valid
# for that forward only (`_wgen`): a training step replayed from a HIP graph updates running statistics and affine parameters
# through raw pointers - no Python, no version bump - so nothing derived from them may outlive the forward that derived it.
_foldcache = {}


class BnFoldSet:
    def __init__(self):
        self.key = None
        self.desc = None
        self.out = None
        self.views = None

    def prepare(self, bns):
        """bns: list of (gamma|None, beta|None, running_mean, running_var, eps).  Must be called inside a weight_scope."""
        import struct
        bns = [b for b in bns if b[2] is not None and b[2].is_cuda]
        if not bns:
            return
        key = tuple((0 if g is None else g.data_ptr(), 0 if b is None else b.data_ptr(), m.data_ptr(), v.data_ptr(), m.numel(), float(e))
                    for g, b, m, v, e in bns)
        dev_t = bns[0][2].device
        if key != self.key:
            total = sum(4 * k[4] for k in key)
            self.out = torch.empty(total, dtype=torch.float32, device=dev_t)
            rows, views, off = [], [], 0
            for k in key:
                C = k[4]
                rows.append([k[0], k[1], k[2], k[3], self.out.data_ptr() + 4 * off, C,
                             struct.unpack("<q", struct.pack("<d", k[5]))[0], 0])
                views.append(tuple(self.out[off + i * C: off + (i + 1) * C] for i in range(4)))
                off += 4 * C
            self.desc = torch.tensor(rows, dtype=torch.int64, device=dev_t)
            self.key, self.views = key, views
        dev, st = _dev_stream(self.out)
        _launch("bn_fold_batch", self.out.numel() * 4, "pb_bn_fold_batch", _p(self.desc), len(key), dev, st)
        for k, v in zip(key, self.views):
            _foldcache[k[2]] = (_wgen, k, v)


def bn_fold_lookup(gamma, beta, running_mean, running_var, eps):
    """The fold BnFoldSet.prepare made for this BatchNorm in the CURRENT top-level forward, or None."""
    ent = _foldcache.get(running_mean.data_ptr())
    if ent is None or ent[0] != _wgen or getattr(_wscope_depth, "d", 0) <= 0:
        return None
    k = (0 if gamma is None else gamma.data_ptr(), 0 if beta is None else beta.data_ptr(), running_mean.data_ptr(),
         running_var.data_ptr(), running_mean.numel(), float(eps))
    return ent[2] if ent[1] == k else None


def pw_fwd(x, w, *, w_transposed=False, in_scale=None, in_shift=None, in_act="none", in_slope=0.0, row_scale=None,
           rows_per_group=0, out_scale=None, out_shift=None, out_act="none", out_slope=0.0, stats=None, residual=None,
           scaled_weights=None):
    """y[M,N] = P(x)[M,K] @ Wl[N,K]^T with fused prologue/epilogue (see pb_pwconv_fwd).
    w: fp32 [N,K], or [K,N] with w_transposed (backward-data).  scaled_weights: bf16 [groups,N,K] = w * row_scale per
    clip, already made by se_fused_fwd (saves the weight-preparation launch of the row_scale path)."""
    _rows(x, "x")
    M, K = x.shape
    _f32(w, name="w")
    if w.dim() != 2:
        raise ValueError("w must be 2-D")
    N = w.shape[1] if w_transposed else w.shape[0]
    if (w.shape[0] if w_transposed else w.shape[1]) != K:
        raise ValueError(f"pw_fwd: K mismatch x {tuple(x.shape)} w {tuple(w.shape)} transposed={w_transposed}")
    _f32(in_scale, (K,), "in_scale"); _f32(in_shift, (K,), "in_shift")
    _f32(out_scale, (N,), "out_scale"); _f32(out_shift, (N,), "out_shift")
    _f32(stats, (REPL, 2, N), "stats")
    if row_scale is not None:
        if rows_per_group <= 0 or M % rows_per_group:
            raise ValueError("pw_fwd: rows_per_group must divide M")
        _f32(row_scale, (M // rows_per_group, K), "row_scale")
    if residual is not None:
        _rows(residual, "residual")
        if tuple(residual.shape) != (M, N) or residual.dtype != x.dtype or stats is not None:
            raise ValueError("pw_fwd: residual must be [M,N] of x's dtype (and excludes stats)")
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    dev, st = _dev_stream(x)
    if (_GEMM_DMA and _K16_FOLD and x.dtype == torch.bfloat16 and K == 16 and N % 8 == 0 and 4 * N <= 96 and M % 4 == 0 and M >= 4096
            and in_scale is None and in_shift is None and in_act == "none" and row_scale is None
            and out_scale is None and out_shift is None and w.is_contiguous() and residual is None):
        # K = 16 (the first stage): a row is 32 bytes, a quarter of an LDS stage.  [M,16] is the same memory as
        # [M/4,64], and Y[M,N] the same as [M/4,4N] = X' * blockdiag(W,W,W,W)^T: a full-width K = 64 GEMM on a quarter
        # of the rows (4x the MFMA work, on zeros - the kernel is HBM-bound): 3.2 -> 4.6 TB/s on the 4M-row tensors when
        # the folded output stays one column tile (N = 16); wider outputs (N = 64 -> 256 columns) measured slower.
        # BatchNorm statistics: the four replicas of a channel are folded inside the workgroup.
        wb4 = _prepped_weight(w, w_transposed, fold4=True)
        _launch("pw_stream_kernel", _nb(x, y) + 4 * N * 64 * 2, "pb_pwconv_fwd_fold4", _p(x), _p(wb4), _p(y), M, N, ACT[out_act], float(out_slope),
                _p(stats), dev, st, detail=f"M{M} K{K} N{N} (4-row fold{', stats' if stats is not None else ''})")
        return y
    if (_GEMM_DMA and x.dtype == torch.bfloat16 and _dma_gemm_pays(M, K, N) and K % 8 == 0 and N % 8 == 0
            and in_scale is None and in_shift is None and in_act == "none" and w.is_contiguous()
            and (row_scale is None or rows_per_group >= 128)):
        # both operands by LDS-DMA: bf16 weights (cached per optimiser step; the squeeze-excite scale is
        # folded into per-clip weights, so the activation stream is read untouched)
        wb = _prepped_weight(w, w_transposed)
        groups = 1
        if row_scale is not None:
            groups = M // rows_per_group
            if scaled_weights is not None and tuple(scaled_weights.shape) == (groups, N, K) and not w_transposed:
                wbs = scaled_weights
            else:
                wbs = torch.empty((groups, N, K), dtype=torch.bfloat16, device=x.device)
                _launch("pw_weight_prep_kernel", _nb(w, row_scale, wbs), "pb_pw_weight_prep", _p(w), N, K, 0, _p(row_scale), groups,
                        _p(wbs), dev, st)
            wb = wbs
        _launch(_bf16w_family(M, K, N, groups, stats is not None), _nb(x, y) + N * K * 2, "pb_pwconv_fwd_bf16w", _p(x), _p(wb), _p(y), M, K, N, groups,
                int(rows_per_group) if groups > 1 else 0, _p(out_scale), _p(out_shift), ACT[out_act], float(out_slope),
                _p(stats), _p(residual), dev, st,
                detail=f"M{M} K{K} N{N}" + (" stats" if stats is not None else "") + (" res" if residual is not None else "") + (" se" if groups > 1 else ""))
        return y
    skinny = (M <= 64 and in_scale is None and in_shift is None and in_act == "none" and row_scale is None
              and stats is None)
    family = "pw_skinny_kernel" if skinny else "pw_gemm_kernel"
    if w_transposed and N * K >= 4096 and not skinny:     # (the skinny kernel reads W in place)
        # backward-data reads W transposed; staging a transposed tile costs scattered 2-byte LDS writes in
        # every workgroup and K stage (2x on the wide layers), transposing the (<= 0.6 MB) matrix once is ~free
        wt = torch.empty((N, K), dtype=torch.float32, device=w.device)
        _launch("transpose_w", _nb(w, wt), "pb_transpose_f32", _p(w), _p(wt), K, N, dev, st)
        w, w_transposed = wt, False
    _launch(family, _nb(x, y, w), "pb_pwconv_fwd", _p(x), _p(w), _p(y), M, K, N, w.shape[1], int(w_transposed), _dt(x), _p(in_scale),
            _p(in_shift), ACT[in_act], float(in_slope), _p(row_scale), int(rows_per_group), _p(out_scale),
            _p(out_shift), ACT[out_act], float(out_slope), _p(stats), dev, st, detail=f"M{M} K{K} N{N}")
    if residual is not None:                 # (only the LDS-DMA kernel adds it in its store pass)
        copy_cols(residual, 0, y, 0, N, accumulate=True)
    return y


_ws_bytes = {}
_family_memo = {}


def _bf16w_family(M, K, N, groups, has_stats):
    """The kernel pb_pwconv_fwd_bf16w runs for this shape (the profile labels follow the library's own dispatch)."""
    key = (M, K, N, groups, has_stats)
    fam = _family_memo.get(key)
    if fam is None:
        path = _C.load().pb_pwconv_fwd_bf16w_path(M, K, N, groups, M // max(groups, 1), int(has_stats))
        fam = _family_memo[key] = ("pw_gemm_dma_kernel", "pw_stream_kernel", "pw_stream_wide_kernel")[path]
    return fam


def pw_bwd_weight(dy, x, dw, *, in_scale=None, in_shift=None, in_act="none", in_slope=0.0, row_scale=None,
                  rows_per_group=0):
    """dw[N,K] += dy[M,N]^T @ P(x)[M,K]."""
    _rows(dy, "dy"); _rows(x, "x")
    M, N = dy.shape
    if x.shape[0] != M or x.dtype != dy.dtype:
        raise ValueError("pw_bwd_weight: dy/x mismatch")
    K = x.shape[1]
    _f32(dw, (N, K), "dw")
    _f32(in_scale, (K,), "in_scale"); _f32(in_shift, (K,), "in_shift")
    if row_scale is not None:
        if rows_per_group <= 0 or M % rows_per_group:
            raise ValueError("pw_bwd_weight: rows_per_group must divide M")
        _f32(row_scale, (M // rows_per_group, K), "row_scale")
    dev, st = _dev_stream(x)
    big = x.dtype == torch.bfloat16 and (N > 64 or K > 64) and N % 8 == 0 and K % 8 == 0
    affine = in_scale is not None or in_shift is not None or in_act != "none"
    se_inloop = row_scale is not None and (rows_per_group < 64 or M % max(rows_per_group, 1))
    family = "pw_wgrad_kernel" if not big else ("pw_wgrad128_kernel" if (affine or se_inloop) else "pw_wgrad128_dma_kernel")
    # workspace for the row ranges' partial tiles (plain stores + a fixed-order reduction: no atomics anywhere)
    key = (M, K, N, x.dtype, affine, row_scale is not None, int(rows_per_group), dev)
    nws = _ws_bytes.get(key)
    if nws is None:
        nws = _ws_bytes[key] = int(_C.load().pb_pwconv_bwd_weight_ws_bytes(M, K, N, _dt(x), int(affine), int(row_scale is not None),
                                                                           int(rows_per_group), dev))
    ws = _scratch(nws, x.device)
    job, jref = _defer_job()
    _launch(family, _nb(dy, x, dw), "pb_pwconv_bwd_weight_ws", _p(dy), _p(x), _p(dw), M, K, N, _dt(x), _p(in_scale), _p(in_shift),
            ACT[in_act], float(in_slope), _p(row_scale), int(rows_per_group), _p(ws), ws.numel() * 4, jref, dev, st,
            detail=f"M{M} K{K} N{N}")
    _keep_job(job, ws, dw)
    return dw


# ------------------------------------------------------------------------------ depthwise conv
def _dw_kernel_name(which, s, dt):
    """HIP kernel pb_dwconv3d_* will dispatch to (profiling labels only); which: 0 fwd, 1 bwd-data, 2 bwd-weight."""
    if _spans is None:
        return "dw"
    return _C.load().pb_dwconv3d_kernel_name(ctypes.byref(s), dt, which).decode()


def dw_out_shape(in_thw, kernel, stride, pad_front, pad_back=None):
    pad_back = pad_front if pad_back is None else pad_back
    return tuple((i + pf + pb - k) // s + 1 for i, k, s, pf, pb in zip(in_thw, kernel, stride, pad_front, pad_back))


def _dw_shape(N, C, in_thw, out_thw, kernel, stride, pad):
    s = _C.DwShape()
    s.N, s.C = N, C
    s.Ti, s.Hi, s.Wi = in_thw
    s.To, s.Ho, s.Wo = out_thw
    s.kt, s.kh, s.kw = kernel
    s.st, s.sh, s.sw = stride
    s.pt, s.ph, s.pw = pad
    return s


def _ndhwc(t, name):
    if t.dim() != 5 or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous [N,T,H,W,C], got {tuple(t.shape)} strides {t.stride()}")
    return t


def dw_fwd(x, w, kernel, stride, pad_front, pad_back=None, pool=None):
    """x [N,T,H,W,C] -> y [N,To,Ho,Wo,C]; w fp32 [C, kt*kh*kw]; pool fp32 [N,C] = sum_THW(y) (overwritten).
    The kernels write one partial sum per work unit (no atomics); the units are added up in a fixed order."""
    _ndhwc(x, "x")
    N, T, H, W, C = x.shape
    taps = kernel[0] * kernel[1] * kernel[2]
    _f32(w, (C, taps), "w")
    out = dw_out_shape((T, H, W), kernel, stride, pad_front, pad_back)
    if min(out) <= 0:
        raise ValueError(f"dw_fwd: empty output {out}")
    _f32(pool, (N, C), "pool")
    y = torch.empty((N,) + out + (C,), dtype=x.dtype, device=x.device)
    s = _dw_shape(N, C, (T, H, W), out, kernel, stride, pad_front)
    dev, st = _dev_stream(x)
    part, slots = None, 0
    if pool is not None:
        slots = int(_C.load().pb_dwconv3d_pool_slots(ctypes.byref(s), _dt(x)))
        part = pool.view(1, N, C) if slots == 1 else torch.empty((slots, N, C), dtype=torch.float32, device=x.device)
    _launch(_dw_kernel_name(0, s, _dt(x)), _nb(x, y, w), "pb_dwconv3d_fwd", _p(x), _p(w), _p(y), ctypes.byref(s), _dt(x), _p(part), slots,
            dev, st, detail=f"C{C} {T}x{H}x{W}->{out[0]}x{out[1]}x{out[2]} k{kernel} s{stride}")
    if pool is not None and slots > 1:
        partials_reduce([(part, pool, N * C, N * C, slots, 0)], x.device)
    return y


def dw_fwd_pool(x, w, kernel, stride, pad_front, pad_back=None):
    """Depthwise conv with the fused squeeze: -> (y, pool_part fp32 [slots, N, C]); the per-work-unit partial sums of y over
    (To,Ho,Wo) are consumed by se_fused_fwd (which adds the slots in a fixed order)."""
    _ndhwc(x, "x")
    N, T, H, W, C = x.shape
    taps = kernel[0] * kernel[1] * kernel[2]
    _f32(w, (C, taps), "w")
    out = dw_out_shape((T, H, W), kernel, stride, pad_front, pad_back)
    if min(out) <= 0:
        raise ValueError(f"dw_fwd_pool: empty output {out}")
    y = torch.empty((N,) + out + (C,), dtype=x.dtype, device=x.device)
    s = _dw_shape(N, C, (T, H, W), out, kernel, stride, pad_front)
    dev, st = _dev_stream(x)
    slots = int(_C.load().pb_dwconv3d_pool_slots(ctypes.byref(s), _dt(x)))
    part = torch.empty((slots, N, C), dtype=torch.float32, device=x.device)
    _launch(_dw_kernel_name(0, s, _dt(x)), _nb(x, y, w), "pb_dwconv3d_fwd", _p(x), _p(w), _p(y), ctypes.byref(s), _dt(x), _p(part), slots,
            dev, st, detail=f"C{C} {T}x{H}x{W}->{out[0]}x{out[1]}x{out[2]} k{kernel} s{stride}")
    return y, part


def dw_fwd_hist(x, hist, w, kernel, stride, pad_front, pad_back=None):
    """Causal depthwise conv of the chunk x [N,T,H,W,C] whose preceding frames are hist [N,nhist,H,W,C] (a CausalConv3d
    stream buffer): pad_front[0] = the unchunked call's total front padding on T (>= nhist; the rest is zeros).
    Returns y, or None when no history-aware kernel takes the shape (the caller concatenates)."""
    _ndhwc(x, "x"); _ndhwc(hist, "hist")
    N, T, H, W, C = x.shape
    nhist = hist.shape[1]
    if tuple(hist.shape) != (N, nhist, H, W, C) or hist.dtype != x.dtype or nhist < 1 or nhist > pad_front[0]:
        raise ValueError(f"dw_fwd_hist: history {tuple(hist.shape)} does not fit x {tuple(x.shape)} / pad_front {pad_front}")
    taps = kernel[0] * kernel[1] * kernel[2]
    _f32(w, (C, taps), "w")
    out = dw_out_shape((T, H, W), kernel, stride, pad_front, pad_back)
    if min(out) <= 0:
        raise ValueError(f"dw_fwd_hist: empty output {out}")
    y = torch.empty((N,) + out + (C,), dtype=x.dtype, device=x.device)
    s = _dw_shape(N, C, (T, H, W), out, kernel, stride, pad_front)
    dev, st = _dev_stream(x)
    rc = _C.call("pb_dwconv3d_fwd_hist", _p(x), _p(hist), nhist, _p(w), _p(y), ctypes.byref(s), _dt(x), dev, st, allow=(3,))
    return y if rc is None or rc == 0 else None


def dw_bwd_data(dy, w, in_thw, kernel, stride, pad_front, g_scale=None, g_shift=None):
    _ndhwc(dy, "dy")
    N, To, Ho, Wo, C = dy.shape
    _f32(w, (C, kernel[0] * kernel[1] * kernel[2]), "w")
    _f32(g_scale, (N, C), "g_scale"); _f32(g_shift, (N, C), "g_shift")
    dx = torch.empty((N,) + tuple(in_thw) + (C,), dtype=dy.dtype, device=dy.device)
    s = _dw_shape(N, C, in_thw, (To, Ho, Wo), kernel, stride, pad_front)
    dev, st = _dev_stream(dy)
    _launch(_dw_kernel_name(1, s, _dt(dy)), _nb(dy, dx, w), "pb_dwconv3d_bwd_data", _p(dy), _p(w), _p(dx), ctypes.byref(s), _dt(dy), _p(g_scale), _p(g_shift), dev, st,
            detail=f"C{C} {To}x{Ho}x{Wo}->{tuple(in_thw)} k{kernel} s{stride}")
    return dx


def dw_bwd_weight(dy, x, dw, kernel, stride, pad_front, g_scale=None, g_shift=None):
    _ndhwc(dy, "dy"); _ndhwc(x, "x")
    N, To, Ho, Wo, C = dy.shape
    if x.shape[0] != N or x.shape[4] != C or x.dtype != dy.dtype:
        raise ValueError("dw_bwd_weight: dy/x mismatch")
    _f32(dw, (C, kernel[0] * kernel[1] * kernel[2]), "dw")
    _f32(g_scale, (N, C), "g_scale"); _f32(g_shift, (N, C), "g_shift")
    s = _dw_shape(N, C, tuple(x.shape[1:4]), (To, Ho, Wo), kernel, stride, pad_front)
    dev, st = _dev_stream(dy)
    ws = _scratch(_C.load().pb_dwconv3d_bwd_weight_ws_bytes(ctypes.byref(s), _dt(dy)), dy.device)
    job, jref = _defer_job()
    _launch(_dw_kernel_name(2, s, _dt(dy)), _nb(dy, x, dw), "pb_dwconv3d_bwd_weight", _p(dy), _p(x), _p(dw), ctypes.byref(s), _dt(dy), _p(g_scale), _p(g_shift),
            _p(ws), ws.numel() * 4, jref, dev, st, detail=f"C{C} {tuple(x.shape[1:4])}->{To}x{Ho}x{Wo} k{kernel} s{stride}")
    _keep_job(job, ws, dw)
    return dw


def dw_bwd_fused_ws_bytes(dy, in_thw, kernel, stride, pad_front):
    """Workspace of the one-launch depthwise backward for this shape; 0 = not taken (use dw_bwd_data + dw_bwd_weight)."""
    if not dy.is_cuda or dy.dim() != 5:
        return 0
    N, To, Ho, Wo, C = dy.shape
    s = _dw_shape(N, C, tuple(in_thw), (To, Ho, Wo), kernel, stride, pad_front)
    return int(_C.load().pb_dwconv3d_bwd_fused_ws_bytes(ctypes.byref(s), _dt(dy)))


def dw_bwd_fused(dy, x, w, dw, kernel, stride, pad_front, g_scale=None, g_shift=None):
    """-> dx; dw[C, taps] += weight gradient.  Both gradients of a (1,k,k) stride-1 depthwise conv in ONE walk over dy
    (g = dy * g_scale + g_shift read once): pb_dwconv3d_bwd_fused.  Check dw_bwd_fused_ws_bytes(...) > 0 first."""
    _ndhwc(dy, "dy"); _ndhwc(x, "x")
    N, To, Ho, Wo, C = dy.shape
    if x.shape[0] != N or x.shape[4] != C or x.dtype != dy.dtype:
        raise ValueError("dw_bwd_fused: dy/x mismatch")
    taps = kernel[0] * kernel[1] * kernel[2]
    _f32(w, (C, taps), "w"); _f32(dw, (C, taps), "dw")
    _f32(g_scale, (N, C), "g_scale"); _f32(g_shift, (N, C), "g_shift")
    s = _dw_shape(N, C, tuple(x.shape[1:4]), (To, Ho, Wo), kernel, stride, pad_front)
    nws = int(_C.load().pb_dwconv3d_bwd_fused_ws_bytes(ctypes.byref(s), _dt(dy)))
    if nws <= 0:
        raise ValueError(f"dw_bwd_fused: shape {tuple(dy.shape)} k{kernel} s{stride} {dy.dtype} is not taken by the fused kernel")
    dx = torch.empty_like(x)
    ws = _scratch(nws, dy.device)
    dev, st = _dev_stream(dy)
    job, jref = _defer_job()
    _launch("dw_rows_bwd_fused_kernel", _nb(dy, x, dx, dw), "pb_dwconv3d_bwd_fused", _p(dy), _p(x), _p(w), _p(dx), _p(dw), ctypes.byref(s), _dt(dy),
            _p(g_scale), _p(g_shift), _p(ws), ws.numel() * 4, jref, dev, st, detail=f"C{C} {tuple(x.shape[1:4])}<->{To}x{Ho}x{Wo} k{kernel} s{stride}")
    _keep_job(job, ws, dw)
    return dx


# ------------------------------------------------------------------------------ frame gather / scatter
def frames_gather(x, t0, tstep, count):
    """x [N,T,H,W,C] -> [N,count,H,W,C]: frames t0, t0+tstep, ... (contiguous copy)."""
    _ndhwc(x, "x")
    N, T = x.shape[:2]
    fb = x[0, 0].numel() * x.element_size()
    if fb % 16 or t0 < 0 or t0 + (count - 1) * tstep >= T:
        raise ValueError(f"frames_gather: frames {t0}+{tstep}*[0,{count}) of {tuple(x.shape)}")
    out = torch.empty((N, count) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
    dev, st = _dev_stream(x)
    _launch("frames_kernel", _nb(out) * 2, "pb_frames_gather", _p(x), _p(out), N, T, count, fb, int(t0), int(tstep), 0, None, 0, dev, st,
            detail=f"gather {tuple(x.shape)} -> T{count}")
    return out


def frames_scatter(xs, T, t0, tstep, fill=None):
    """xs [N,Tu,H,W,C] -> [N,T,H,W,C] with xs[:, j] at frame t0 + j*tstep; every other frame holds exact zeros, or the
    per-channel row `fill` [C] (same dtype as xs) repeated over its pixels."""
    _ndhwc(xs, "xs")
    N, Tu = xs.shape[:2]
    C = xs.shape[4]
    fb = xs[0, 0].numel() * xs.element_size()
    if fb % 16:
        raise ValueError("frames_scatter: frame size must be a multiple of 16 bytes")
    if fill is not None and (fill.dtype != xs.dtype or tuple(fill.shape) != (C,) or not fill.is_contiguous() or (C * xs.element_size()) % 16):
        raise ValueError("frames_scatter: fill must be a contiguous [C] row of the activations' dtype (C * itemsize a multiple of 16)")
    out = torch.empty((N, T) + tuple(xs.shape[2:]), dtype=xs.dtype, device=xs.device)
    dev, st = _dev_stream(xs)
    _launch("frames_kernel", _nb(out, xs), "pb_frames_gather", _p(xs), _p(out), N, Tu, T, fb, int(t0), int(tstep), 1, _p(fill),
            C * xs.element_size() if fill is not None else 0, dev, st, detail=f"scatter T{Tu} -> {tuple(out.shape)}")
    return out


# ------------------------------------------------------------------------------ stem conv
_XDT = {torch.float32: 0, torch.bfloat16: 1, torch.uint8: 2}


def _stem_shape(x, cout, out_thw, kernel, stride, pad):
    """x: logical (N,Cin,T,H,W) with arbitrary strides."""
    s = _C.StemShape()
    s.N, s.Cin, s.Ti, s.Hi, s.Wi = x.shape
    s.Cout = cout
    s.To, s.Ho, s.Wo = out_thw
    s.kt, s.kh, s.kw = kernel
    s.st, s.sh, s.sw = stride
    s.pt, s.ph, s.pw = pad
    s.xs_n, s.xs_c, s.xs_t, s.xs_h, s.xs_w = x.stride()
    return s


def _stem_kernel_name(what, x, cout, kernel, stride, pad, dtype, x_scale):
    """Mirror of stem3_ok() in csrc/stem.hip (profiling labels only)."""
    fast = (x.dtype == torch.bfloat16 and dtype == torch.bfloat16 and x.shape[1] == 3 and kernel[2] == 3 and stride[2] == 2
            and pad[2] == 1 and x.stride(1) == 1 and x.stride(4) == 3 and kernel[0] * kernel[1] <= 9 and cout == 16
            and x_scale == 1.0)
    if what == "fwd":
        return "stem3_fwd_kernel" if fast else "stem_fwd_mfma_kernel"
    return "stem3_wgrad_kernel" if fast else "stem_wgrad_kernel"


def stem_fwd(x, w, bias, kernel, stride, pad, out_dtype, x_scale=1.0, stats=None):
    """x logical (N,Cin,T,H,W), any strides, float32/bfloat16/uint8 -> y [N,To,Ho,Wo,Cout] in out_dtype."""
    if x.dim() != 5 or x.dtype not in _XDT:
        raise ValueError(f"stem_fwd: bad input {tuple(x.shape)} {x.dtype}")
    cout = w.shape[0]
    _f32(w, (cout, x.shape[1], *kernel), "w"); _f32(bias, (cout,), "bias")
    if stats is not None:
        _f32(stats, (stats.shape[0], 2, cout), "stats")
    out = dw_out_shape(tuple(x.shape[2:]), kernel, stride, pad)
    y = torch.empty((x.shape[0],) + out + (cout,), dtype=out_dtype, device=x.device)
    s = _stem_shape(x, cout, out, kernel, stride, pad)
    dev, st = _dev_stream(x)
    _launch(_stem_kernel_name("fwd", x, w.shape[0], kernel, stride, pad, out_dtype, x_scale), _nb(x, y, w), "pb_stem_conv3d_fwd", _p(x), _XDT[x.dtype], float(x_scale), _p(w), _p(bias), _p(y), ctypes.byref(s),
            _dt(y), _p(stats), stats.shape[0] if stats is not None else 0, dev, st)
    return y


def stem_bwd_weight(dy, x, dw, dbias, kernel, stride, pad, x_scale=1.0):
    _ndhwc(dy, "dy")
    cout = dy.shape[4]
    _f32(dw, (cout, x.shape[1], *kernel), "dw"); _f32(dbias, (cout,), "dbias")
    s = _stem_shape(x, cout, tuple(dy.shape[1:4]), kernel, stride, pad)
    dev, st = _dev_stream(dy)
    ws = _scratch(_C.load().pb_stem_conv3d_bwd_weight_ws_bytes(ctypes.byref(s), _XDT[x.dtype], float(x_scale), _dt(dy)), dy.device)
    _launch(_stem_kernel_name("wgrad", x, dw.shape[0], kernel, stride, pad, dy.dtype, x_scale), _nb(dy, x, dw), "pb_stem_conv3d_bwd_weight", _p(dy), _p(x), _XDT[x.dtype], float(x_scale), _p(dw), _p(dbias),
            ctypes.byref(s), _dt(dy), _p(ws), ws.numel() * 4, dev, st)
    return dw


# ------------------------------------------------------------------------------ statistics / norm / activation
def chan_stats(x, stats):
    _rows(x, "x")
    _f32(stats, (REPL, 2, x.shape[1]), "stats")
    dev, st = _dev_stream(x)
    _launch("ew_kernel<stats>", _nb(x), "pb_chan_stats", _p(x), x.shape[0], x.shape[1], _dt(x), _p(stats), dev, st)
    return stats


def bn_finalize(stats, count, gamma, beta, eps, momentum, running_mean, running_var, use_running, C, device,
                num_batches_tracked=None):
    """-> (scale, shift, mean, rstd), each fp32 [C]; running stats (and nn.BatchNorm's int64 step counter) updated in
    place when training."""
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or num_batches_tracked.numel() != 1):
        raise TypeError("bn_finalize: num_batches_tracked must be a 1-element int64 tensor")
    if stats is not None:
        _f32(stats, (stats.shape[0], 2, C), "stats")
    _f32(gamma, (C,), "gamma"); _f32(beta, (C,), "beta")
    _f32(running_mean, (C,), "running_mean"); _f32(running_var, (C,), "running_var")
    out = torch.empty((4, C), dtype=torch.float32, device=device)
    ref = stats if stats is not None else running_mean
    dev, st = _dev_stream(ref)
    _launch("bn_finalize", _nb(out), "pb_bn_finalize", _p(stats), stats.shape[0] if stats is not None else 0, float(count), _p(gamma), _p(beta), float(eps), float(momentum),
            _p(running_mean), _p(running_var), _p(None if use_running else num_batches_tracked), int(use_running), _p(out[0]),
            _p(out[1]), _p(out[2]), _p(out[3]), C, dev, st)
    return out[0], out[1], out[2], out[3]


def _frame_map(frame_map, rows_compact, who):
    """frame_map = (rows per clip of the full tensor, data rows per clip, first data row) -> (clips, full M, map args)."""
    if frame_map is None:
        return None
    full, data, lead = (int(v) for v in frame_map)
    if full <= 0 or data <= 0 or lead < 0 or lead + data > full or rows_compact % data:
        raise ValueError(f"{who}: bad frame map {frame_map} for {rows_compact} compact rows")
    return rows_compact // data * full, full, data, lead


def affine_act_fwd(x, scale=None, shift=None, act="none", slope=0.0, row_scale=None, rows_per_group=0, stats=None,
                   row_shift=None, mean=None, frame_map=None):
    """y = act((x-mean)*scale+shift) * row_scale[g] + row_shift[g]; stats[REPL,2,C] += (sum y, sum y^2).
    frame_map (rows per clip, data rows per clip, first data row): x is COMPACT (the data rows only, exact zeros elsewhere:
    the frames a (1,k,k) depthwise conv pads in time), y is the full tensor; rows_per_group counts full rows."""
    _rows(x, "x")
    M, C = x.shape
    fm = _frame_map(frame_map, M, "affine_act_fwd")
    if fm is not None:
        M = fm[0]
    _f32(scale, (C,), "scale"); _f32(shift, (C,), "shift"); _f32(stats, (REPL, 2, C), "stats"); _f32(mean, (C,), "mean")
    if row_scale is not None or row_shift is not None:
        if rows_per_group <= 0 or M % rows_per_group:
            raise ValueError("affine_act_fwd: rows_per_group must divide M")
        _f32(row_scale, (M // rows_per_group, C), "row_scale")
        _f32(row_shift, (M // rows_per_group, C), "row_shift")
    y = torch.empty((M, C), dtype=x.dtype, device=x.device)
    dev, st = _dev_stream(x)
    f1, f2, f3 = fm[1:] if fm is not None else (0, 0, 0)
    _launch("ew_kernel<affine_act>", _nb(x, y), "pb_affine_act_fwd", _p(x), _p(y), M, C, _dt(x), _p(scale), _p(shift), _p(mean), ACT[act], float(slope),
            _p(row_scale), _p(row_shift), int(rows_per_group), _p(stats), f1, f2, f3, dev, st, detail=f"M{M} C{C}")
    return y


def act_row(shift, act, slope, dtype):
    """act(shift) per channel as a [C] row of `dtype`: what a folded inference BatchNorm + activation makes of an exact-zero
    pre-activation (the frames a (1,k,k) depthwise conv pads in time); the fill row of frames_scatter."""
    _f32(shift, name="shift")
    C = shift.shape[0]
    return affine_act_fwd(shift.view(1, C), act=act, slope=slope).to(dtype).view(C)      # (the cast of C values: host-side glue)


def act_bwd_reduce(dy, z, *, scale=None, shift=None, act="none", slope=0.0, row_scale=None, rows_per_group=0,
                   dy_rows_per_group=0, dy_mul=1.0, mean=None, rstd=None, sums=None, want_g=True, frame_map=None):
    """g = dy*row_scale*act'(z*scale+shift) (returned when want_g); sums[2,C] += (sum g, sum g*xhat).
    frame_map (see affine_act_fwd): z and the returned g are COMPACT, dy is the full tensor, the sums run over all full rows
    (z = 0 on the padded ones)."""
    _rows(z, "z"); _rows(dy, "dy")
    M, C = z.shape
    fm = _frame_map(frame_map, M, "act_bwd_reduce")
    if fm is not None:
        M = fm[0]
    if dy_rows_per_group > 0:
        if M % dy_rows_per_group or tuple(dy.shape) != (M // dy_rows_per_group, C):
            raise ValueError("act_bwd_reduce: broadcast dy shape mismatch")
    elif tuple(dy.shape) != (M, C):
        raise ValueError("act_bwd_reduce: dy/z shape mismatch")
    if dy.dtype != z.dtype:
        raise TypeError("act_bwd_reduce: dy/z dtype mismatch")
    for t, n in ((scale, "scale"), (shift, "shift"), (mean, "mean"), (rstd, "rstd")):
        _f32(t, (C,), n)
    _f32(sums, (REPL, 2, C), "sums")
    if row_scale is not None:
        if rows_per_group <= 0 or M % rows_per_group:
            raise ValueError("act_bwd_reduce: rows_per_group must divide M")
        _f32(row_scale, (M // rows_per_group, C), "row_scale")
    g = torch.empty_like(z) if want_g else None
    dev, st = _dev_stream(z)
    f1, f2, f3 = fm[1:] if fm is not None else (0, 0, 0)
    _launch("ew_kernel<act_bwd>", _nb(dy, z, g), "pb_act_bwd_reduce", _p(dy), _p(z), _p(g), M, C, _dt(z), _p(scale), _p(shift), ACT[act], float(slope),
            _p(row_scale), int(rows_per_group), int(dy_rows_per_group), float(dy_mul), _p(mean), _p(rstd), _p(sums),
            f1, f2, f3, dev, st, detail=f"M{M} C{C}")
    return g


def bn_bwd_finalize(sums, count, gamma, mean, rstd, training, want_affine_grads):
    """-> (A, B, Cc, dgamma, dbeta); dz = A*g + B*z + Cc."""
    C = mean.shape[0]
    _f32(sums, (sums.shape[0], 2, C), "sums"); _f32(gamma, (C,), "gamma"); _f32(mean, (C,), "mean"); _f32(rstd, (C,), "rstd")
    out = torch.empty((5, C), dtype=torch.float32, device=mean.device)
    dev, st = _dev_stream(mean)
    _launch("bn_bwd_finalize", _nb(out), "pb_bn_bwd_finalize", _p(sums), sums.shape[0], float(count), _p(gamma), _p(mean), _p(rstd), int(training), _p(out[0]),
            _p(out[1]), _p(out[2]), _p(out[3]) if want_affine_grads else None,
            _p(out[4]) if want_affine_grads else None, C, dev, st)
    return out[0], out[1], out[2], (out[3] if want_affine_grads else None), (out[4] if want_affine_grads else None)


def affine2(g, z, A, B=None, Cc=None, mean=None, row_scale=None, rows_per_group=0, act_scale=None, act_shift=None,
            act="none", slope=0.0):
    """out = A*(g*row_scale*act'((z-mean)*act_scale+act_shift)) + B*(z - mean) + Cc."""
    _rows(g, "g")
    M, C = g.shape
    if row_scale is not None:
        if rows_per_group <= 0 or M % rows_per_group:
            raise ValueError("affine2: rows_per_group must divide M")
        _f32(row_scale, (M // rows_per_group, C), "row_scale")
    if z is not None:
        _rows(z, "z")
        if z.shape != g.shape or z.dtype != g.dtype:
            raise ValueError("affine2: g/z mismatch")
    _f32(A, (C,), "A"); _f32(B, (C,), "B"); _f32(Cc, (C,), "Cc"); _f32(mean, (C,), "mean")
    _f32(act_scale, (C,), "act_scale"); _f32(act_shift, (C,), "act_shift")
    if act != "none" and z is None:
        raise ValueError("affine2: an activation needs z")
    out = torch.empty_like(g)
    dev, st = _dev_stream(g)
    _launch("ew_kernel<affine2>", _nb(g, z, out), "pb_affine2", _p(g), _p(z), _p(out), M, C, _dt(g), _p(A), _p(B), _p(Cc), _p(mean), _p(row_scale),
            int(rows_per_group), _p(act_scale), _p(act_shift), ACT[act], float(slope), dev, st, detail=f"M{M} C{C}")
    return out


def rowgroup_reduce(a, b, G):
    """out[G,C] = per-group sum over rows of a (b None) or a*b; a is [G*R, C].  Per-workgroup partial rows in a
    workspace, added in a fixed order (no atomics)."""
    _rows(a, "a")
    M, C = a.shape
    if M % G:
        raise ValueError("rowgroup_reduce: G must divide M")
    if b is not None:
        _rows(b, "b")
        if b.shape != a.shape or b.dtype != a.dtype:
            raise ValueError("rowgroup_reduce: a/b mismatch")
    out = torch.empty((G, C), dtype=torch.float32, device=a.device)
    dev, st = _dev_stream(a)
    nws = int(_C.load().pb_rowgroup_reduce_ws_bytes(G, M // G, C, _dt(a)))
    ws = _scratch(nws, a.device) if nws else None
    _launch("rowgroup_kernel", _nb(a, b, out), "pb_rowgroup_reduce", _p(a), _p(b), _p(out), G, M // G, C, _dt(a), _p(ws), nws, dev, st,
            detail=f"M{M} C{C} G{G}")
    return out


# ------------------------------------------------------------------------------ squeeze-excite
def se_fc_fwd(pool, inv_count, w1, b1, w2, b2):
    """pool fp32 [G,C] (sums) -> (scale [G,C], hidden [G,Cm])."""
    G, C = pool.shape
    Cm = w1.shape[0]
    _f32(pool, name="pool"); _f32(w1, (Cm, C), "w1"); _f32(w2, (C, Cm), "w2"); _f32(b1, (Cm,), "b1"); _f32(b2, (C,), "b2")
    scale = torch.empty((G, C), dtype=torch.float32, device=pool.device)
    hidden = torch.empty((G, Cm), dtype=torch.float32, device=pool.device)
    dev, st = _dev_stream(pool)
    _launch("se_fc_fwd", _nb(pool, w1, w2, scale), "pb_se_fc_fwd", _p(pool), float(inv_count), _p(w1), _p(b1), _p(w2), _p(b2), _p(scale), _p(hidden), G, C,
            Cm, dev, st)
    return scale, hidden


def se_fc_bwd(dscale, pool, inv_count, scale, hidden, w1, w2, want_bias=True):
    """-> (g_shift [G,C], dw1 [Cm,C], db1 [Cm], dw2 [C,Cm], db2 [C])."""
    G, C = pool.shape
    Cm = w1.shape[0]
    for t, shp, n in ((dscale, (G, C), "dscale"), (pool, (G, C), "pool"), (scale, (G, C), "scale"),
                      (hidden, (G, Cm), "hidden"), (w1, (Cm, C), "w1"), (w2, (C, Cm), "w2")):
        _f32(t, shp, n)
    dev_t = pool.device
    g_shift = torch.empty((G, C), dtype=torch.float32, device=dev_t)
    flat = zeros(2 * C * Cm + C + Cm, dev_t)
    dw1, dw2 = flat[:C * Cm].view(Cm, C), flat[C * Cm:2 * C * Cm].view(C, Cm)
    db1, db2 = flat[2 * C * Cm:2 * C * Cm + Cm], flat[2 * C * Cm + Cm:]
    work = torch.empty(G * (C + Cm), dtype=torch.float32, device=dev_t)
    dev, st = _dev_stream(pool)
    _launch("se_fc_bwd", _nb(pool, w1, w2, dw1, dw2), "pb_se_fc_bwd", _p(dscale), _p(pool), float(inv_count), _p(scale), _p(hidden), _p(w1), _p(w2),
            _p(g_shift), _p(dw1), _p(db1) if want_bias else None, _p(dw2), _p(db2) if want_bias else None, _p(work),
            G, C, Cm, dev, st)
    return g_shift, dw1, (db1 if want_bias else None), dw2, (db2 if want_bias else None)


def _few_slots(part, limit=64):
    """The fused squeeze-excite kernels re-add the slots in several workgroups per clip: beyond `limit` slots (MoViNet's
    large frames) one fixed-order reduction launch folds them first."""
    if part.shape[0] <= limit:
        return part
    out = torch.empty((1,) + tuple(part.shape[1:]), dtype=torch.float32, device=part.device)
    n = out.numel()
    partials_reduce([(part, out, n, n, part.shape[0], 0)], part.device)
    return out


def se_fused_fwd(pool_part, inv_count, w1, b1, w2, b2, wproj=None, want_scaled=False):
    """pool_part fp32 [slots,G,C] -> (pool [G,C], scale [G,C], hidden [G,Cm], wbs): ONE launch.  wbs (want_scaled and
    wproj fp32 [N,C] given, C and N multiples of 8): bf16 [G,N,C] = wproj * scale per clip for pw_fwd(scaled_weights=)."""
    if pool_part.dim() != 3:
        raise ValueError("se_fused_fwd: pool_part must be [slots, G, C]")
    pool_part = _few_slots(pool_part)
    slots, G, C = pool_part.shape
    Cm = w1.shape[0]
    _f32(pool_part, name="pool_part"); _f32(w1, (Cm, C), "w1"); _f32(w2, (C, Cm), "w2"); _f32(b1, (Cm,), "b1"); _f32(b2, (C,), "b2")
    dev_t = pool_part.device
    pool = torch.empty((G, C), dtype=torch.float32, device=dev_t)
    scale = torch.empty((G, C), dtype=torch.float32, device=dev_t)
    hidden = torch.empty((G, Cm), dtype=torch.float32, device=dev_t)
    wbs, N = None, 0
    if want_scaled and wproj is not None and C % 8 == 0 and wproj.shape[0] % 8 == 0 and wproj.data_ptr() % 16 == 0:
        _f32(wproj, (wproj.shape[0], C), "wproj")
        N = wproj.shape[0]
        wbs = torch.empty((G, N, C), dtype=torch.bfloat16, device=dev_t)
    dev, st = _dev_stream(pool_part)
    _launch("se_fused_fwd", _nb(pool_part, w1, w2, scale, wbs), "pb_se_fused_fwd", _p(pool_part), slots, float(inv_count), _p(w1), _p(b1),
            _p(w2), _p(b2), _p(pool), _p(scale), _p(hidden), _p(wproj if wbs is not None else None), _p(wbs), N, G, C, Cm, dev, st)
    return pool, scale, hidden, wbs


def pool_head_fwd(pool_part, inv_count, w1, b1, w2, b2, act, out_dtype):
    """The pooled classifier head on the squeeze-excite kernels: pool_part fp32 [slots,G,C] -> (pool [G,C] sums, pre1 [G,Cm]
    fp32: the hidden layer's PRE-activation for hardswish, its output for ReLU - what the backward kernels need, logits [G,Co]
    in out_dtype); act = "hardswish" | "relu" between the two FCs."""
    pool_part = _few_slots(pool_part)
    slots, G, C = pool_part.shape
    Cm, Co = w1.shape[0], w2.shape[0]
    _f32(pool_part, name="pool_part"); _f32(w1, (Cm, C), "w1"); _f32(w2, (Co, Cm), "w2"); _f32(b1, (Cm,), "b1"); _f32(b2, (Co,), "b2")
    dev_t = pool_part.device
    pool = torch.empty((G, C), dtype=torch.float32, device=dev_t)
    pre1 = torch.empty((G, Cm), dtype=torch.float32, device=dev_t)
    logits = torch.empty((G, Co), dtype=torch.float32, device=dev_t)
    lb = torch.empty((G, Co), dtype=torch.bfloat16, device=dev_t) if out_dtype == torch.bfloat16 else None
    dev, st = _dev_stream(pool_part)
    _launch("se_fused_fwd", _nb(pool_part, w1, w2, logits), "pb_pool_head_fwd", _p(pool_part), slots, float(inv_count), _p(w1), _p(b1), _p(w2),
            _p(b2), ACT[act], _p(pool), _p(logits), _p(lb), _p(pre1), G, C, Cm, Co, dev, st)
    return pool, pre1, (lb if lb is not None else logits)


def pool_head_bwd(dlogits, inv_count, pre1, w1, w2, act, row_dtype):
    """dlogits fp32 [G,Co] -> (dpool_row [G,C] in row_dtype: the gradient of every pooled row of clip g, work [G,Co+Cm] for
    pool_head_bwd_w)."""
    G, Co = dlogits.shape
    Cm, C = w1.shape
    _f32(dlogits, (G, Co), "dlogits"); _f32(pre1, (G, Cm), "pre1"); _f32(w1, (Cm, C), "w1"); _f32(w2, (Co, Cm), "w2")
    dev_t = dlogits.device
    row = torch.empty((G, C), dtype=torch.float32, device=dev_t)
    rb = torch.empty((G, C), dtype=torch.bfloat16, device=dev_t) if row_dtype == torch.bfloat16 else None
    work = torch.empty((G, Co + Cm), dtype=torch.float32, device=dev_t)
    dev, st = _dev_stream(dlogits)
    _launch("se_fused_bwd", _nb(dlogits, w1, w2, row), "pb_pool_head_bwd", _p(dlogits), float(inv_count), _p(pre1), _p(w1), _p(w2), ACT[act],
            _p(row), _p(rb), _p(work), G, C, Cm, Co, dev, st)
    return (rb if rb is not None else row), work


def pool_head_bwd_w(work, pool, inv_count, pre1, act, dw1, db1, dw2, db2):
    """dw1 [Cm,C], db1, dw2 [Co,Cm], db2 += the head's weight / bias gradients (off the data-gradient chain)."""
    G, C = pool.shape
    Cm = pre1.shape[1]
    Co = work.shape[1] - Cm
    _f32(work, (G, Co + Cm), "work"); _f32(pool, (G, C), "pool"); _f32(pre1, (G, Cm), "pre1")
    _f32(dw1, (Cm, C), "dw1"); _f32(dw2, (Co, Cm), "dw2"); _f32(db1, (Cm,), "db1"); _f32(db2, (Co,), "db2")
    dev, st = _dev_stream(work)
    _launch("se_fc_bwd_w", _nb(dw1, dw2), "pb_pool_head_bwd_w", _p(work), _p(pool), float(inv_count), _p(pre1), ACT[act], _p(dw1), _p(db1), _p(dw2),
            _p(db2), G, C, Cm, Co, dev, st)


def rowgroup_partials(a, b, G):
    """Per-workgroup partial sums [slots,G,C] of rowgroup_reduce (for se_fused_bwd, which adds the slots itself)."""
    _rows(a, "a")
    M, C = a.shape
    if M % G:
        raise ValueError("rowgroup_partials: G must divide M")
    if b is not None:
        _rows(b, "b")
        if b.shape != a.shape or b.dtype != a.dtype:
            raise ValueError("rowgroup_partials: a/b mismatch")
    slots = int(_C.load().pb_rowgroup_slots(G, M // G, C, _dt(a)))
    part = torch.empty((slots, G, C), dtype=torch.float32, device=a.device)
    dev, st = _dev_stream(a)
    _launch("rowgroup_kernel", _nb(a, b, part), "pb_rowgroup_partials", _p(a), _p(b), _p(part), G, M // G, C, _dt(a), dev, st,
            detail=f"M{M} C{C} G{G}")
    return part


def se_fused_bwd(dscale_part, pool, inv_count, scale, hidden, w1, w2, want_bias=True, defer_weights=False):
    """dscale_part fp32 [slots,G,C] -> (g_shift [G,C], dw1, db1, dw2, db2): the per-clip vectors (two launches) and the FC weight
    gradients (one).  defer_weights: the weight-gradient launch is NOT issued; a sixth return value is the thunk that issues it
    (pb_se_fc_bwd_w) - for the block's weight-gradient branch: nothing on the data-gradient chain reads these gradients."""
    dscale_part = _few_slots(dscale_part)
    slots, G, C = dscale_part.shape
    Cm = w1.shape[0]
    for t, shp, n in ((pool, (G, C), "pool"), (scale, (G, C), "scale"), (hidden, (G, Cm), "hidden"), (w1, (Cm, C), "w1"),
                      (w2, (C, Cm), "w2")):
        _f32(t, shp, n)
    _f32(dscale_part, name="dscale_part")
    dev_t = pool.device
    g_shift = torch.empty((G, C), dtype=torch.float32, device=dev_t)
    flat = zeros(2 * C * Cm + C + Cm, dev_t)
    dw1, dw2 = flat[:C * Cm].view(Cm, C), flat[C * Cm:2 * C * Cm].view(C, Cm)
    db1, db2 = flat[2 * C * Cm:2 * C * Cm + Cm], flat[2 * C * Cm + Cm:]
    work = torch.empty(G * (C + Cm), dtype=torch.float32, device=dev_t)
    dev, st = _dev_stream(pool)
    if defer_weights:
        _launch("se_fused_bwd", _nb(dscale_part, w1, w2), "pb_se_fused_bwd", _p(dscale_part), slots, _p(pool), float(inv_count),
                _p(scale), _p(hidden), _p(w1), _p(w2), _p(g_shift), None, None, None, None, _p(work), G, C, Cm, dev, st)

        def weights():
            d2, s2 = _dev_stream(pool)
            _launch("se_fc_bwd_w", _nb(dw1, dw2), "pb_se_fc_bwd_w", _p(work), _p(pool), float(inv_count), _p(hidden), _p(dw1),
                    _p(db1) if want_bias else None, _p(dw2), _p(db2) if want_bias else None, G, C, Cm, d2, s2)
        weights.keep = (work, pool, hidden, flat)
        return g_shift, dw1, (db1 if want_bias else None), dw2, (db2 if want_bias else None), weights
    _launch("se_fused_bwd", _nb(dscale_part, w1, w2, dw1, dw2), "pb_se_fused_bwd", _p(dscale_part), slots, _p(pool), float(inv_count),
            _p(scale), _p(hidden), _p(w1), _p(w2), _p(g_shift), _p(dw1), _p(db1) if want_bias else None, _p(dw2),
            _p(db2) if want_bias else None, _p(work), G, C, Cm, dev, st)
    return g_shift, dw1, (db1 if want_bias else None), dw2, (db2 if want_bias else None)


# ------------------------------------------------------------------------------ dense conv (implicit GEMM)
_CONV_DMA = os.environ.get("PB_CONV_DMA", "1") != "0"


def _conv_shape(N, cin, cout, in_thw, out_thw, kernel, stride, pad):
    s = _C.ConvShape()
    s.N, s.Cin, s.Cout = N, cin, cout
    s.Ti, s.Hi, s.Wi = in_thw
    s.To, s.Ho, s.Wo = out_thw
    s.kt, s.kh, s.kw = kernel
    s.st, s.sh, s.sw = stride
    s.pt, s.ph, s.pw = pad
    return s


def conv_weight_prep(w, mode, cout, cin, taps):
    """mode 0: [Cout,Cin,taps] -> [Cout,taps,Cin]; 1: -> [Cin,taps reversed,Cout]; 2: [Cout,taps,Cin] -> [Cout,Cin,taps]."""
    _f32(w, name="w")
    if w.numel() != cout * cin * taps:
        raise ValueError("conv_weight_prep: size mismatch")
    out = torch.empty(cout * cin * taps, dtype=torch.float32, device=w.device)
    dev, st = _dev_stream(w)
    _launch("conv_weight_prep", _nb(w, out), "pb_conv_weight_prep", _p(w), _p(out), cout, cin, taps, int(mode), dev, st)
    return out


def conv3d_fwd(x, wp, cout, kernel, stride, pad_front, out_thw, *, out_scale=None, out_shift=None, out_act="none",
               out_slope=0.0, stats=None):
    """x [N,T,H,W,Cin], wp fp32 [Cout*taps*Cin] (prep mode 0) -> y [N,*out_thw,Cout]."""
    _ndhwc(x, "x")
    N, T, H, W, cin = x.shape
    taps = kernel[0] * kernel[1] * kernel[2]
    _f32(wp, (cout * taps * cin,), "wp")
    _f32(out_scale, (cout,), "out_scale"); _f32(out_shift, (cout,), "out_shift"); _f32(stats, (REPL, 2, cout), "stats")
    if min(out_thw) <= 0:
        raise ValueError(f"conv3d_fwd: empty output {out_thw}")
    y = torch.empty((N,) + tuple(out_thw) + (cout,), dtype=x.dtype, device=x.device)
    s = _conv_shape(N, cin, cout, (T, H, W), out_thw, kernel, stride, pad_front)
    dev, st = _dev_stream(x)
    if (_CONV_DMA and x.dtype == torch.bfloat16 and tuple(stride) == (1, 1, 1) and cin % 8 == 0 and cout % 8 == 0 and taps <= 64
            and taps * cin < 65536 and y.numel() // cout < 2 ** 31 and x.numel() // cin < 2 ** 30):
        # implicit GEMM on the LDS-DMA kernel (bf16 filter copy: one tiny launch)
        wb = torch.empty(cout * taps * cin, dtype=torch.bfloat16, device=x.device)
        _launch("pw_weight_prep_kernel", _nb(wp, wb), "pb_pw_weight_prep", _p(wp), cout, taps * cin, 0, None, 1, _p(wb), dev, st)
        _launch("conv3d_dma_kernel", _nb(x, y, wb), "pb_conv3d_fwd_bf16w", _p(x), _p(wb), _p(y), ctypes.byref(s), _p(zeros(8, x.device)),
                _p(out_scale), _p(out_shift), ACT[out_act], float(out_slope), _p(stats), dev, st,
                detail=f"Cin{cin} Cout{cout} {T}x{H}x{W} k{kernel}")
        return y
    _launch("pw_gemm_kernel<conv>", _nb(x, y, wp), "pb_conv3d_fwd", _p(x), _p(wp), _p(y), ctypes.byref(s), _dt(x), _p(out_scale),
            _p(out_shift), ACT[out_act], float(out_slope), _p(stats), dev, st,
            detail=f"Cin{cin} Cout{cout} {T}x{H}x{W} k{kernel} s{stride}")
    return y


def conv3d_bwd_weight(dy, x, kernel, stride, pad_front):
    """-> dwp fp32 [Cout, taps, Cin] (convert with conv_weight_prep mode 2)."""
    _ndhwc(dy, "dy"); _ndhwc(x, "x")
    N, To, Ho, Wo, cout = dy.shape
    cin = x.shape[4]
    taps = kernel[0] * kernel[1] * kernel[2]
    if x.shape[0] != N or x.dtype != dy.dtype:
        raise ValueError("conv3d_bwd_weight: dy/x mismatch")
    dwp = zeros(cout * taps * cin, x.device)
    s = _conv_shape(N, cin, cout, tuple(x.shape[1:4]), (To, Ho, Wo), kernel, stride, pad_front)
    dev, st = _dev_stream(x)
    nws = int(_C.load().pb_conv3d_bwd_weight_ws_bytes(ctypes.byref(s), _dt(x), dev))
    ws = _scratch(nws, x.device)
    family = "conv3d_wgrad_dma_kernel" if (x.dtype == torch.bfloat16 and cin % 8 == 0 and cout % 8 == 0) else "pw_wgrad_kernel<conv>"
    _launch(family, _nb(dy, x, dwp), "pb_conv3d_bwd_weight_ws", _p(dy), _p(x), _p(dwp), ctypes.byref(s), _dt(x),
            _p(ws), ws.numel() * 4, _p(zeros(8, x.device)), dev, st, detail=f"Cin{cin} Cout{cout} {tuple(x.shape[1:4])} k{kernel}")
    return dwp


# ------------------------------------------------------------------------------ MobileViT token ops
def unfold(x5, patch):
    """voxels [B,T,H,W,D] -> tokens [B, pt*ph*pw, t*h*w, D] (zero padded to patch multiples)."""
    _ndhwc(x5, "x5")
    B, T, H, W, D = x5.shape
    pt, ph, pw = patch
    t, h, w = -(-T // pt), -(-H // ph), -(-W // pw)
    tok = torch.empty((B, pt * ph * pw, t * h * w, D), dtype=x5.dtype, device=x5.device)
    dev, st = _dev_stream(x5)
    _launch("unfold", _nb(x5, tok), "pb_unfold_fold", _p(x5), _p(tok), B, D, T, H, W, pt, ph, pw, 1, _dt(x5), dev, st)
    return tok


def fold(tok, thw, patch):
    """tokens [B,P,n,D] -> voxels [B,T,H,W,D] (cropped)."""
    if tok.dim() != 4 or not tok.is_contiguous():
        raise ValueError("fold: expected contiguous [B,P,n,D] tokens")
    B, P_, n, D = tok.shape
    T, H, W = thw
    pt, ph, pw = patch
    if P_ != pt * ph * pw or n != (-(-T // pt)) * (-(-H // ph)) * (-(-W // pw)):
        raise ValueError("fold: token shape does not match the voxel grid")
    vox = torch.empty((B, T, H, W, D), dtype=tok.dtype, device=tok.device)
    dev, st = _dev_stream(tok)
    _launch("fold", _nb(tok, vox), "pb_unfold_fold", _p(tok), _p(vox), B, D, T, H, W, pt, ph, pw, 0, _dt(tok), dev, st)
    return vox


def layernorm_fwd(x, eps):
    _rows(x, "x")
    M, D = x.shape
    y = torch.empty_like(x)
    stat = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    dev, st = _dev_stream(x)
    _launch("layernorm_fwd", _nb(x, y), "pb_layernorm", _p(x), None, _p(y), _p(stat), M, D, float(eps), 0, _dt(x), dev, st)
    return y, stat


def layernorm_bwd(x, dy, stat, eps):
    _rows(x, "x"); _rows(dy, "dy")
    M, D = x.shape
    _f32(stat, (M, 2), "stat")
    if dy.shape != x.shape or dy.dtype != x.dtype:
        raise ValueError("layernorm_bwd: x/dy mismatch")
    dx = torch.empty_like(x)
    dev, st = _dev_stream(x)
    _launch("layernorm_bwd", _nb(x, dy, dx), "pb_layernorm", _p(x), _p(dy), _p(dx), _p(stat), M, D, float(eps), 1, _dt(x), dev, st)
    return dx


def layernorm_wide_fwd(x32, eps):
    """bf16 LayerNorm(x) of the fp32 token stream -> (xn bf16 [M,D], stat fp32 [M,2])."""
    _rows(x32, "x"); _f32(x32, None, "x")
    M, D = x32.shape
    xn = torch.empty((M, D), dtype=torch.bfloat16, device=x32.device)
    stat = torch.empty((M, 2), dtype=torch.float32, device=x32.device)
    dev, st = _dev_stream(x32)
    _launch("layernorm_fwd", _nb(x32, xn), "pb_layernorm_wide", _p(x32), None, None, _p(xn), None, _p(stat), M, D, float(eps), 0, dev, st)
    return xn, stat


def layernorm_wide_bwd(x32, dxn, stat, eps, res32, want_bf16=True):
    """-> (g32 = res32 + dLayerNorm(x32; dxn bf16), bf16(g32) or None): LayerNorm backward + the residual branch of the
    gradient stream in one pass."""
    _rows(x32, "x"); _f32(x32, None, "x"); _rows(dxn, "dxn"); _f32(res32, tuple(x32.shape), "res")
    M, D = x32.shape
    _f32(stat, (M, 2), "stat")
    if dxn.shape != x32.shape or dxn.dtype != torch.bfloat16:
        raise ValueError("layernorm_wide_bwd: dxn must be bf16 of x's shape")
    g32 = torch.empty_like(x32)
    gbf = torch.empty((M, D), dtype=torch.bfloat16, device=x32.device) if want_bf16 else None
    dev, st = _dev_stream(x32)
    _launch("layernorm_bwd", _nb(x32, dxn, res32, g32, gbf), "pb_layernorm_wide", _p(x32), _p(dxn), _p(res32), _p(gbf), _p(g32), _p(stat),
            M, D, float(eps), 1, dev, st)
    return g32, gbf


def stream_add(a, b, want32=True, want_bf16=True):
    """(a + b in fp32, its bf16 rounding): a fp32 or bf16, b bf16, same shape - the residual add of the wide token stream."""
    _rows(a, "a"); _rows(b, "b")
    if a.shape != b.shape or b.dtype != torch.bfloat16 or a.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("stream_add: a (fp32 / bf16) and b (bf16) must have the same shape")
    if a.numel() % 8:
        raise ValueError("stream_add: element count must be a multiple of 8")
    o32 = torch.empty(a.shape, dtype=torch.float32, device=a.device) if want32 else None
    obf = torch.empty(a.shape, dtype=torch.bfloat16, device=a.device) if want_bf16 else None
    dev, st = _dev_stream(a)
    _launch("stream_add", _nb(a, b, o32, obf), "pb_stream_add", _p(a), _dt(a), _p(b), _p(o32), _p(obf), a.numel(), dev, st)
    return o32, obf


def copy_cols(src, src_off, dst, dst_off, C, accumulate=False):
    """dst[:, dst_off:dst_off+C] (+)= src[:, src_off:src_off+C] for contiguous 2-D src/dst with equal row counts."""
    _rows(src, "src"); _rows(dst, "dst")
    if src.shape[0] != dst.shape[0] or src.dtype != dst.dtype:
        raise ValueError("copy_cols: src/dst mismatch")
    if src_off + C > src.shape[1] or dst_off + C > dst.shape[1]:
        raise ValueError("copy_cols: column range out of bounds")
    dev, st = _dev_stream(src)
    _launch("copy_cols", 2 * src.shape[0] * C * src.element_size(), "pb_copy_cols", _p(src), src.shape[1], int(src_off), _p(dst),
            dst.shape[1], int(dst_off), src.shape[0], int(C), int(accumulate), _dt(src), dev, st)
    return dst


def attention_fwd(qkv, G, n, scale, fp8=False):
    """qkv [G*n, 384] -> (out [G*n, 128], lse fp32 [G*n, 8]); 8 heads of 16 (mobilevit.py:116).
    fp8: bf16 tensors, fp8 (e4m3) MFMA operands in both products (inference; BASELINE config 5)."""
    _rows(qkv, "qkv")
    if qkv.shape != (G * n, 384):
        raise ValueError(f"attention_fwd: qkv shape {tuple(qkv.shape)} != ({G * n}, 384)")
    out = torch.empty((G * n, 128), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((G * n, 8), dtype=torch.float32, device=qkv.device)
    dev, st = _dev_stream(qkv)
    if fp8:
        if qkv.dtype != torch.bfloat16:
            raise TypeError("attention_fwd(fp8=True): bfloat16 activations only")
        _launch("attention_fwd_fp8", _nb(qkv, out), "pb_attention_fwd_fp8", _p(qkv), _p(out), _p(lse), G, n, float(scale), dev, st,
                detail=f"G{G} n{n}")
        return out, lse
    _launch("attention_fwd", _nb(qkv, out), "pb_attention_fwd", _p(qkv), _p(out), _p(lse), G, n, float(scale), _dt(qkv), dev, st,
            detail=f"G{G} n{n}")
    return out, lse


def attention_bwd(qkv, out, dout, lse, G, n, scale):
    _rows(qkv, "qkv"); _rows(out, "out"); _rows(dout, "dout")
    if qkv.shape != (G * n, 384) or out.shape != (G * n, 128) or dout.shape != out.shape or dout.dtype != qkv.dtype:
        raise ValueError("attention_bwd: shape/dtype mismatch")
    _f32(lse, (G * n, 8), "lse")
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((G * n, 8), dtype=torch.float32, device=qkv.device)
    dev, st = _dev_stream(qkv)
    _launch("attention_bwd", _nb(qkv, out, dout, dqkv), "pb_attention_bwd", _p(qkv), _p(out), _p(dout), _p(lse), _p(delta), _p(dqkv),
            G, n, float(scale), _dt(qkv), dev, st, detail=f"G{G} n{n}")
    return dqkv


def linattn_fwd(qkv, G, n):
    """MobileViT-V2 separable attention core on 3-D patch tokens (mobilevitv2.py:37-52): qkv [G*n, 1+2D] in qkv_proj's
    (q | k | v) column order -> (out [G*n, D] = relu(v) * ctx, scores fp32 [G*n], ctx fp32 [G, D])."""
    _rows(qkv, "qkv")
    L = qkv.shape[1]
    D = (L - 1) // 2
    if qkv.shape[0] != G * n or L != 1 + 2 * D or D < 1:
        raise ValueError(f"linattn_fwd: qkv shape {tuple(qkv.shape)} is not ({G * n}, 1 + 2*D)")
    out = torch.empty((G * n, D), dtype=qkv.dtype, device=qkv.device)
    scores = torch.empty((G * n,), dtype=torch.float32, device=qkv.device)
    ctx = torch.empty((G, D), dtype=torch.float32, device=qkv.device)
    dev, st = _dev_stream(qkv)
    _launch("linattn_fwd", _nb(qkv, out), "pb_linattn_fwd", _p(qkv), _p(scores), _p(ctx), _p(out), G, n, D, _dt(qkv), dev, st,
            detail=f"G{G} n{n} D{D}")
    return out, scores, ctx


def linattn_bwd(qkv, scores, ctx, dout, G, n):
    _rows(qkv, "qkv"); _rows(dout, "dout")
    D = (qkv.shape[1] - 1) // 2
    if qkv.shape != (G * n, 1 + 2 * D) or dout.shape != (G * n, D) or dout.dtype != qkv.dtype:
        raise ValueError("linattn_bwd: shape/dtype mismatch")
    _f32(scores, (G * n,), "scores"); _f32(ctx, (G, D), "ctx")
    dqkv = torch.empty_like(qkv)
    scratch = torch.empty((G * D + G * n,), dtype=torch.float32, device=qkv.device)
    dev, st = _dev_stream(qkv)
    _launch("linattn_bwd", _nb(qkv, dout, dqkv), "pb_linattn_bwd", _p(qkv), _p(scores), _p(ctx), _p(dout), _p(scratch[:G * D]),
            _p(scratch[G * D:]), _p(dqkv), G, n, D, _dt(qkv), dev, st, detail=f"G{G} n{n} D{D}")
    return dqkv


# ------------------------------------------------------------------------------ optimizer step
def adamw_flat(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay, inv_scale=None, found_inf=None, lr_wd=None):
    """In-place fused AdamW over flat fp32 buffers; `step`: device float tensor holding this update's step number;
    lr_wd: optional device float[2] = (lr, weight_decay) read by the kernel instead of the host values."""
    n = p.numel()
    for t, name in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _f32(t, (n,), name)
    _f32(step, (), "step"); _f32(inv_scale, (), "inv_scale"); _f32(found_inf, (), "found_inf"); _f32(lr_wd, (2,), "lr_wd")
    dev, st = _dev_stream(p)
    _launch("adamw_flat", 7 * n * 4, "pb_adamw_flat", _p(p), _p(g), _p(m), _p(v), n, float(lr), float(beta1), float(beta2), float(eps),
            float(weight_decay), _p(step), _p(inv_scale), _p(found_inf), _p(lr_wd), dev, st)


def grad_check(g, found_inf):
    """found_inf (device float scalar, zeroed by the caller) = 1 when any element of g is inf/nan."""
    _f32(g, name="g"); _f32(found_inf, (), "found_inf")
    dev, st = _dev_stream(g)
    _launch("grad_check", g.numel() * 4, "pb_grad_check", _p(g), g.numel(), _p(found_inf), dev, st)
