"""Data-parallel gradient reduction for the WGAN phases: bucketed all-reduce over RCCL/xGMI.

The reference ships no gradient synchronisation on this path (its multi-GPU mode runs N
diverging replicas); the only statement of intent is the dormant
``all_reduce_and_rescale_tensors`` (onmt/utils/distributed.py:35-86): flat buckets,
all_reduce(SUM), then divide by the number of ranks.  This module implements exactly those
semantics -- reduced grad = (sum over ranks) / world -- once per optimizer step:

* ``arm(module)`` before the LAST backward that touches ``module``: per-parameter
  post-accumulate hooks count finished gradients; when a bucket's last gradient lands the bucket
  is gathered into its flat buffer (one multi-tensor copy) and its all-reduce launched on a side
  stream, overlapping WHAT IS LEFT of that backward.  What is left is short for the main model: the
  decode engine is one autograd node whose parameter gradients (151 MB of the model's 331 MB: the 12
  decoder layers, embedding and generator) are batched products at the END of its backward sweep, so
  their buckets go out when the sweep -- 17 of the generator phase's ~24 ms -- is already over; the
  overlap window is the ~4 ms of encoder / bond-layer / style-generator backward that follows (their own
  gradients land during it and are flushed by the final sync).  The window is not the decode sweep;
* ``__call__(modules)`` after the last backward: reduces whatever was not armed / not
  flushed, waits for the side stream, divides by world and points every ``.grad`` at its slice of the
  reduced flat buffer (no scatter copy: the optimizer reads the slices in place).

One process per GPU; backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU tests.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets default to 32 MiB so that each
collective is large enough to be bandwidth- rather than latency-bound; the ~10 buckets of the main
model leave in a burst after the decode sweep (see above).  Measured single-rank cost of the path
(identity reduction): tools/dp_overhead.py, profiles/r03_dp_single_rank.txt.
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, world_size=None, bucket_bytes=32 << 20, group=None, always=False, hook_min_numel=4096):
        """``hook_min_numel``: matrices of at least this many elements are reduced from gradient hooks while the
        backward runs; everything smaller waits for the final sync (see _plan).
        ``always``: issue the collectives even for a single rank (an identity reduction), which lets one GPU
        exercise the RCCL path end to end -- hooks, side stream, async handles -- as the multi-GPU job will run it."""
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group)
                                                                 if dist.is_initialized() else 1)
        self.active = self.world > 1 or (always and dist.is_initialized())
        if self.active and torch.cuda.is_available():
            # the reduction runs on a side stream, and replayed decode sweeps slow down (~2 ms each) in a process that
            # drives a second stream (include/mssr_hip.h, mssr_decode_graphs_enable; profiles/r03_dp_single_rank.txt):
            # data-parallel ranks launch their sweeps eagerly
            from . import _lib as L
            L.load().mssr_decode_graphs_enable(0)
        self.bucket_bytes = int(bucket_bytes)
        self.hook_min_numel = int(hook_min_numel)
        self._plans = {}        # id(module) -> list of buckets
        self._pending = []      # (bucket, work handle)
        self._hooks = []
        self._stream = None
        self._in_hook = False   # True while a gradient hook (autograd thread) runs: nothing there may block the host
        import os
        self._check = os.environ.get("MSSR_DP_CHECK", "0") not in ("", "0")

    # ---------------------------------------------------------------- bucket plan
    def _plan(self, module):
        plan = self._plans.get(id(module))
        if plan is not None:
            return plan
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()                      # gradients become ready roughly in reverse order
        # Matrices go into hooked buckets (reduced while the backward is still running); vectors -- biases, LayerNorm
        # gains: two thirds of the parameters, 0.1 % of the bytes -- go into buckets of their own that are reduced
        # at the final sync, so that only a third of the gradients cost a Python hook call (~6 us each from the
        # autograd thread).
        big = [p for p in params if p.dim() >= 2 and p.numel() >= self.hook_min_numel]
        small = [p for p in params if not (p.dim() >= 2 and p.numel() >= self.hook_min_numel)]
        buckets = []
        for group, hooked in ((big, True), (small, False)):
            cur, cur_bytes = [], 0
            for p in group:
                nb = p.numel() * 4
                if cur and cur_bytes + nb > self.bucket_bytes:
                    buckets.append((cur, hooked))
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nb
            if cur:
                buckets.append((cur, hooked))
        plan = []
        for ps, hooked in buckets:
            # every slice starts on a 16-byte boundary: the multi-tensor kernels (gather below, clip + Adam in optim.py)
            # read the slices with 16-byte accesses
            offs, n = [], 0
            for p in ps:
                offs.append(n)
                n += (p.numel() + 3) // 4 * 4
            flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
            views = [flat[off:off + p.numel()].view_as(p) for off, p in zip(offs, ps)]
            plan.append(dict(params=ps, flat=flat, views=views, ready=0, launched=False, armed=False, hooked=hooked,
                             scaled=False, tables=None, absent=[False] * len(ps)))
        self._plans[id(module)] = plan
        # gradient hooks are registered ONCE (registering and removing ~400 of them per phase cost ~2 ms of host
        # time per step); they only count while their bucket is armed
        for b in plan:
            if not b["hooked"]:
                continue
            for p in b["params"]:
                def hook(param, b=b):
                    if not b["armed"]:
                        return
                    b["ready"] += 1
                    if b["ready"] == len(b["params"]) and not b["launched"]:
                        self._in_hook = True
                        try:
                            self._fill(b)
                            self._launch(b)
                        finally:
                            self._in_hook = False
                self._hooks.append(p.register_post_accumulate_grad_hook(hook))
        return plan

    def _side_stream(self, device):
        if device.type != "cuda":
            return None
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def _fill(self, b):
        """Gradients of the bucket's parameters -> its flat buffer; parameters without a gradient contribute zeros.
        On the GPU: ONE multi-tensor launch per 96 tensors (mssr_multi_copy, pointers in the kernel arguments) that
        also pre-divides by the world size, so the reduced sum IS the mean; elsewhere (CPU tensors under gloo) one
        torch multi-tensor copy."""
        grads = [p.grad for p in b["params"]]
        # Parameters without a gradient in this phase stay without one after the reduction (their zero slice is
        # reduced, not handed out): torch.optim.Adam / optim.Adam skip grad-None parameters -- no moment decay, no
        # step count -- and data-parallel training must do the same as a single process.  The autograd graph of a
        # phase is the same on every rank (same code, same phase), so the set is too; MSSR_DP_CHECK=1 verifies it
        # with one small collective per sync.
        b["absent"] = [g is None for g in grads]
        if b["flat"].is_cuda and all(g is None or (g.dtype == torch.float32 and g.is_contiguous() and g.is_cuda
                                                   and g.data_ptr() % 16 == 0) for g in grads):
            import ctypes
            from . import _lib as L
            n = len(grads)
            if b["tables"] is None:
                b["tables"] = ((ctypes.c_void_p * n)(*[v.data_ptr() for v in b["views"]]), (ctypes.c_void_p * n)(),
                               (ctypes.c_int64 * n)(*[p.numel() for p in b["params"]]))
            dst, src, numels = b["tables"]
            for i, g in enumerate(grads):
                src[i] = None if g is None else g.data_ptr()     # a gradient that IS its slice is scaled in place
            L.check(L.load().mssr_multi_copy(n, dst, src, numels, 1.0 / self.world, L.stream()), "multi_copy")
            b["scaled"] = True
            return
        b["scaled"] = False
        live = [(v, g) for v, g in zip(b["views"], grads) if g is not None]
        if len(live) < len(b["params"]):
            # (a gradient that still IS its slice -- accumulation without zero_grad -- must survive the zero fill)
            keep = [(v, g.clone()) for v, g in live if g.data_ptr() == v.data_ptr()]
            b["flat"].zero_()
            live = [(v, g) for v, g in live if g.data_ptr() != v.data_ptr()] + keep
        else:
            live = [(v, g) for v, g in live if g.data_ptr() != v.data_ptr()]
        if live:
            torch._foreach_copy_([v for v, _ in live], [g for _, g in live])

    def _launch(self, b):
        if not self.active:
            b["launched"] = True
            return
        flat = b["flat"]
        if flat.is_cuda and dist.get_backend(self.group) == "gloo":
            # INVARIANT: nothing reachable from a gradient hook may block the host.  This branch does (a device ->
            # host copy waits for every kernel the host has queued, then a blocking collective): it is reached from
            # __call__ only, after the phase's last backward has returned (arm() refuses gloo-on-device: _overlap_ok).
            assert not self._in_hook, "blocking gloo reduction reached from a gradient hook"
            host = flat.cpu()                        # blocking copy: the bucket is final
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            flat.copy_(host)
            b["launched"] = True
            self._pending.append((b, None))
            return
        # RCCL (and gloo on host tensors): pure enqueues, safe from a hook -- an event record + stream wait, then an
        # async collective on the side stream; the handle is waited for in __call__ only (for RCCL that wait is
        # itself a stream-side event wait, not a host block)
        side = self._side_stream(flat.device)
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(side):
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        b["launched"] = True
        self._pending.append((b, work))

    # ---------------------------------------------------------------- overlap path
    def _overlap_ok(self, module):
        """Collectives are launched from gradient hooks only where the launch is a pure enqueue (RCCL on device
        tensors, gloo on host tensors).  gloo on DEVICE tensors -- the transport of the one-GPU tests -- runs
        worker threads that block in stream-synchronize and stalled against a host that is thousands of
        launches ahead of the GPU (ranks stuck in the KFD event wait); that combination is reduced at the
        phase's final sync, staged through host memory (_launch)."""
        p = next(module.parameters(), None)
        return not (p is not None and p.is_cuda and dist.get_backend(self.group) == "gloo")

    def _check_presence(self):
        """MSSR_DP_CHECK=1: every rank must have left the same parameters without a gradient (blocking; debug only)."""
        mine = torch.tensor([float(a) for b, _ in self._pending for a in b["absent"]])
        dev = self._pending[0][0]["flat"].device if self._pending else torch.device("cpu")
        if dist.get_backend(self.group) != "gloo":
            mine = mine.to(dev)
        tot = mine.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        bad = ((tot != 0) & (tot != self.world)).nonzero().flatten().tolist()
        if bad:
            raise RuntimeError("GradSync: ranks disagree on which parameters have no gradient (flat indices %s)" % bad[:8])

    def arm(self, module):
        """Hook the next backward through ``module``: buckets are reduced as they fill."""
        if not self.active or not self._overlap_ok(module):
            return
        for b in self._plan(module):
            b["ready"], b["launched"], b["armed"] = 0, False, True

    # ---------------------------------------------------------------- finish
    def __call__(self, modules):
        if not self.active:
            return
        assert not self._in_hook, "GradSync.__call__ (the only place that waits) entered from a gradient hook"
        for m in modules:
            for b in self._plan(m):
                b["armed"] = False
        for m in modules:
            for b in self._plan(m):
                if not b["launched"]:
                    self._fill(b)            # parameters whose hook never fired keep / get zeros
                    self._launch(b)
        for b, work in self._pending:
            if work is not None:
                work.wait()
        if self._check and self.world > 1:
            self._check_presence()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        inv = 1.0 / self.world
        for b, _ in self._pending:
            if self.world > 1 and not b["scaled"]:
                b["flat"].mul_(inv)
            # No scatter back: every parameter's .grad BECOMES its slice of the reduced flat buffer (a pointer swap
            # on the host; the optimizer reads the slices in place).  The slices stay valid until this bucket is
            # filled again, i.e. until the next backward of these parameters has finished -- by then zero_grad() has
            # dropped them (or, if the caller accumulates without zero_grad, autograd has added into them in place and
            # _fill's copy is a copy onto itself).
            for v, p, absent in zip(b["views"], b["params"], b["absent"]):
                if absent:
                    p.grad = None
                elif p.dtype == torch.float32:
                    p.grad = v
                else:                                   # buckets are fp32; a parameter of another dtype gets a copy
                    p.grad = v.to(p.dtype)
            b["ready"], b["launched"] = 0, False
        self._pending = []


def init_distributed(backend=None):
    """Join the job torch.distributed.run started (env://, MASTER_ADDR 127.0.0.1 on one node)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world
