"""Fused AdamW over flat parameter buckets (the step right after the forward/backward path).

The reference steps `bitsandbytes.optim.AdamW8bit` (train.py:208-212) through a `GradScaler` (train.py:283-289);
torch's own fused AdamW walks ~170 small tensors in four multi-tensor launches (0.14 ms of a 9 ms step).  Here the
parameters of a bucket live in ONE flat fp32 buffer (`p.data` become views of it, exactly like DistributedDataParallel's
gradient buckets), so does the gradient, and one `pb_adamw_flat` launch per bucket applies torch.optim.AdamW's update
rule.  State (exp_avg, exp_avg_sq) is fp32.

    opt = FlatAdamW(model.parameters(), lr=3e-4, weight_decay=5e-4)        # or FlatAdamW(reducer, ...) for data parallel
    opt.zero_grad()                 # = detach_grads(): backward hands its gradient tensors over as they are
    loss.backward()
    opt.step()                      # gathers the gradients into the flat bucket (one multi-tensor copy) and updates

`overlap_weight_grads` (default: on, except in a process group without a `GradReducer`): `pack_grads()` joins the
weight-gradient side stream before it reads a gradient, so the blocks of these parameters may run their weight gradients
there (`kernels.weight_grad_branch`).  Pass False when anything else reads `.grad` while backward is still running -
stock DistributedDataParallel does (train.py:203-204).

Construct it after the model is on its device (parameters are re-pointed into the flat buffer; `state_dict` keys and values are
unchanged).  With a `GradReducer` the reducer's buckets are adopted (its flat gradient buffers are what the all-reduce fills).
Everything is stream-ordered and allocation-free after construction, so a whole step can be captured into a HIP graph:
the step count AND the (lr, weight_decay) pair live on the device.  It IS a torch.optim.Optimizer (one param group), so
the reference's `CosineAnnealingLR(optimizer, ...)` (train.py:215) drives it; `step()` uploads a changed lr before it
launches, and a loop that replays a captured step calls `refresh_hyper()` before each replay for the same effect.
"""
import torch

from . import kernels as K
from .parallel import bucket_layout


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, overlap_weight_grads=None):
        reducer = params if hasattr(params, "buckets") and hasattr(params, "flat") else None
        self._reducer = reducer
        if overlap_weight_grads is None:
            # a process group without a GradReducer: the gradients are most likely exchanged by stock DistributedDataParallel,
            # whose C++ hooks read `.grad` DURING backward on the calling stream - keep every weight gradient in line then
            overlap_weight_grads = reducer is not None or not (torch.distributed.is_available() and torch.distributed.is_initialized())
        if reducer is not None:
            buckets, gflat, gviews = reducer.buckets, reducer.flat, reducer.views
        else:
            plist = [p for p in params if p.requires_grad]
            if not plist:
                raise ValueError("FlatAdamW: no parameters")
            offs, n = bucket_layout(plist)
            buckets = [plist]
            gflat = [torch.zeros(n, dtype=torch.float32, device=plist[0].device)]
            gviews = [[gflat[0][o:o + p.numel()].view_as(p) for p, o in zip(plist, offs)]]
        # ONE param group (lr schedulers edit param_groups[0]["lr"]); torch.optim.Optimizer keeps the bookkeeping
        super().__init__([p for b in buckets for p in b], dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.buckets, self.gflat, self.gviews = buckets, gflat, gviews
        self.pflat, self.exp_avg, self.exp_avg_sq = [], [], []
        for plist, g in zip(self.buckets, self.gflat):
            flat = torch.zeros_like(g)
            offs, _ = bucket_layout(plist)                   # (the same 64-byte-aligned layout as the gradient bucket)
            for p, off in zip(plist, offs):
                if p.dtype != torch.float32:
                    raise TypeError("FlatAdamW: parameters must be float32 masters")
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat[off:off + n].view_as(p)        # the parameter now lives in the bucket
                # this optimiser joins the weight-gradient side stream before it reads a gradient (pack_grads), so its
                # parameters' blocks may defer their weight gradients (kernels.weight_grad_branch: opt-in per consumer)
                if reducer is None:                           # (a GradReducer has marked its parameters itself)
                    p._pb_join_aware = bool(overlap_weight_grads)
            self.pflat.append(flat)
            self.exp_avg.append(torch.zeros_like(flat))
            self.exp_avg_sq.append(torch.zeros_like(flat))
        dev = self.pflat[0].device
        self.step_t = torch.zeros((), dtype=torch.float32, device=dev)
        self.found_inf = torch.zeros((), dtype=torch.float32, device=dev)
        self.hyper = torch.tensor([lr, weight_decay], dtype=torch.float32, device=dev)   # what the kernel reads
        self._hyper_host = (float(lr), float(weight_decay))
        self._packed = False

    def refresh_hyper(self):
        """Upload (lr, weight_decay) of param_groups[0] when a scheduler changed them.  step() calls this itself; a loop
        that REPLAYS a captured step calls it before each replay (the kernel reads the pair from device memory)."""
        g0 = self.param_groups[0]
        cur = (float(g0["lr"]), float(g0["weight_decay"]))
        if cur != self._hyper_host:
            self.hyper.copy_(torch.tensor(cur, dtype=torch.float32), non_blocking=True)
            self._hyper_host = cur

    # -- gradients ------------------------------------------------------------------------
    def detach_grads(self):
        """.grad = None everywhere: the next backward leaves its own gradient tensors (no accumulate kernels)."""
        for plist in self.buckets:
            for p in plist:
                p.grad = None
        self._packed = False

    def zero_grad(self, set_to_none=True):
        self.detach_grads()

    def pack_grads(self):
        """One multi-tensor copy per bucket of whatever backward left in .grad (missing gradients count as zeros)."""
        K.join_weight_grads()              # (a no-op unless a weight-gradient branch is still in flight)
        if self._reducer is not None:
            self._reducer.pack_grads()
        else:
            for views, plist in zip(self.gviews, self.buckets):
                srcs = [p.grad if p.grad is not None else torch.zeros_like(p) for p in plist]
                torch._foreach_copy_(views, srcs)
        self._packed = True

    # -- update ---------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, inv_scale=None, check_inf=False, packed=None):
        """packed: the flat gradient buffers already hold this step's gradients (a reducer filled / reduced them);
        default: gather them now unless pack_grads() was called since the last zero_grad().
        inv_scale (device float tensor): GradScaler's 1/scale; check_inf: skip the update when a gradient is inf/nan."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if packed is None:
            packed = self._packed
        if not packed:
            self.pack_grads()
        if not (self.hyper.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.refresh_hyper()                             # (during capture the values are whatever was uploaded last)
        g0 = self.param_groups[0]
        lr, (b1, b2), eps, wd = g0["lr"], g0["betas"], g0["eps"], g0["weight_decay"]
        found = None
        if check_inf:
            self.found_inf.zero_()
            for g in self.gflat:
                K.grad_check(g, self.found_inf)
            found = self.found_inf
            self.step_t.add_(1.0 - found)                   # a skipped update does not count (GradScaler semantics)
        else:
            self.step_t += 1
        for p, g, m, v in zip(self.pflat, self.gflat, self.exp_avg, self.exp_avg_sq):
            K.adamw_flat(p, g, m, v, self.step_t, lr, b1, b2, eps, wd, inv_scale=inv_scale, found_inf=found, lr_wd=self.hyper)
        self._packed = False
        return loss

    # -- state ----------------------------------------------------------------------------
    def state_dict(self):
        return {"step": self.step_t.clone(), "exp_avg": [t.clone() for t in self.exp_avg],
                "exp_avg_sq": [t.clone() for t in self.exp_avg_sq], "param_groups": [
                    {k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.step_t.copy_(sd["step"])
        for dst, src in zip(self.exp_avg, sd["exp_avg"]):
            dst.copy_(src)
        for dst, src in zip(self.exp_avg_sq, sd["exp_avg_sq"]):
            dst.copy_(src)
        self.param_groups[0].update(sd["param_groups"][0])
        self.refresh_hyper()
