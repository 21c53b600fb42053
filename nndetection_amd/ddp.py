"""Patch-level data parallelism: one process per GPU, gradient all-reduce over RCCL/xGMI.

The reference has no DDP code of its own (multi-GPU is whatever pl.Trainer(gpus=N,
accelerator='ddp') does, scripts/train.py:265-289, documented as unsupported in README.md:572-575).
This is the MI355X-native replacement for that wiring.

Design for 8 x MI355X (fully connected xGMI, 7 links x ~153 GB/s per GPU), 18.9 M parameters = 75.6 MB fp32:
  * STATIC flat buckets: the parameter -> bucket layout is fixed at construction (reverse registration order =
    the order gradients become ready in backward: segmenter/heads -> decoder -> encoder), so every rank issues
    the same collectives in the same order with no per-step graph traversal (stock DDP needs
    find_unused_parameters=True here, see below).
  * A SMALL first bucket (default 4 MB) so the first all-reduce starts as soon as the heads' gradients exist,
    then ~24 MB buckets: large enough to run near link bandwidth, small enough that 3-4 of them pipeline
    behind the decoder/encoder backward. Collectives are issued asynchronously from autograd hooks the moment
    the last gradient of a bucket is accumulated; `finish()` waits and writes the averaged gradients back.
  * Unused parameters are ZERO-FILLED instead of discovered: `decoder.out.P1.*` never receives a gradient and a
    rank without positive anchors has no gradient for the 12 regressor tensors (nndet/arch/heads/comb.py:397-401);
    a missing gradient contributes zeros so the bucket layout stays identical on all ranks.
  * Norm layers are per-sample (InstanceNorm / GroupNorm): no activation collectives. Batch-level hard-negative
    mining and batch-dice stay per rank, exactly what the reference under Lightning-DDP would compute.
"""
import contextlib
import os
from typing import List, Optional

import torch
import torch.distributed as dist


_ALIGN = 64                             # floats: every parameter's region starts on a 256-byte boundary (as _lib._GradPool.take does)


def _padded(n: int) -> int:
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


class GradBucket:
    """`flat`: the bucket's slice of the reducer's ONE flat buffer (so that one fill zeroes all buckets); the parameters' regions are
    256-byte aligned inside it, the padding stays zero and is reduced along (<= 252 bytes per parameter)."""

    def __init__(self, params: List[torch.nn.Parameter], flat: torch.Tensor):
        self.params = params
        self.flat = flat
        self.numel = flat.numel()
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += _padded(p.numel())
        assert off == self.numel
        self.pending = len(params)
        self.work = None
        self.streams = set()            # raw handles of the streams that accumulated a gradient of this bucket in this backward
        self.from_hook = False          # launched by a gradient hook (under the backward pass) or by finish()?


class GradAllReducer:
    """Usage:  ddp = GradAllReducer(model);  loss.backward();  ddp.finish();  optimizer.step();  optimizer.zero_grad(set_to_none=True)
    After finish() every p.grad is a VIEW of its bucket: zero_grad(set_to_none=True) (not in-place zeroing or accumulation over
    several backward passes) is required before the next backward."""

    def __init__(self, model: torch.nn.Module, first_bucket_mb: Optional[float] = None, bucket_mb: Optional[float] = None,
                 process_group=None, overlap: bool = True, static_unused=None, force_overlap: bool = False,
                 bucket_dtype: Optional[torch.dtype] = None, profile: bool = False):
        """first_bucket_mb / bucket_mb (defaults 4 / 24, env NNDET_DDP_FIRST_MB / NNDET_DDP_BUCKET_MB): bucket sizes in MB of fp32
        gradients -- the sweep switch for tuning against the 7 xGMI links of a node (bench.py prints per-bucket launch offsets and
        the exposed all-reduce time for each setting). bucket_dtype (env NNDET_DDP_BF16=1 -> bfloat16): all-reduce the buckets in a
        16-bit type (half the wire bytes; the averaged gradient is then rounded to that type: off by default).
        profile: record events around the bucket launches and finish() so that `profile_summary()` can report how much of the
        communication was NOT hidden under the backward pass.
        static_unused: parameters that NEVER receive a gradient on any rank (default: `model.never_used_parameters()` if the
        model has it -- for RetinaUNet the `decoder.out.P<l>` convs of levels nobody reads). They are zero-filled and do not
        count towards bucket readiness; otherwise their bucket -- and, because collectives are issued in order, every later
        one -- could only be launched from finish(), i.e. without overlapping the backward pass.
        force_overlap: run the hook -> bucket copy -> (all-reduce) -> gradient-view path even at world size 1, so the
        overlapped path (incl. its stream synchronisation against the multi-stream detection head) is testable on ONE GPU."""
        if first_bucket_mb is None:
            first_bucket_mb = float(os.environ.get("NNDET_DDP_FIRST_MB", "4"))
        if bucket_mb is None:
            bucket_mb = float(os.environ.get("NNDET_DDP_BUCKET_MB", "24"))
        if bucket_dtype is None:
            bucket_dtype = torch.bfloat16 if os.environ.get("NNDET_DDP_BF16", "0") == "1" else torch.float32
        self.first_bucket_mb, self.bucket_mb, self.bucket_dtype = first_bucket_mb, bucket_mb, bucket_dtype
        self.pg = process_group
        self.world = dist.get_world_size(self.pg) if dist.is_initialized() else 1
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("model has no trainable parameters")
        device = params[0].device
        self.buckets: List[GradBucket] = []
        groups, cur, cur_bytes, limit = [], [], 0, first_bucket_mb * 2 ** 20
        for p in reversed(params):                       # gradients arrive roughly in reverse registration order
            cur.append(p); cur_bytes += p.numel() * 4
            if cur_bytes >= limit:
                groups.append(cur); cur, cur_bytes, limit = [], 0, bucket_mb * 2 ** 20
        if cur:
            groups.append(cur)
        sizes = [sum(_padded(p.numel()) for p in g) for g in groups]
        self._flat_all = torch.zeros(sum(sizes), dtype=bucket_dtype, device=device)
        off = 0
        for g, n in zip(groups, sizes):
            self.buckets.append(GradBucket(g, self._flat_all[off:off + n]))
            off += n
        # In place (round 5; fp32 buckets, NNDET_DDP_INPLACE=0 disables): the convolution nodes write their parameter gradients
        # straight into the buckets (_lib.grad_pool static layout), the all-reduce runs on that memory and the optimizer reads it:
        # no per-step copy of the 75 MB of gradients (2.3 % of a step at world size 1, profiles/round4_scale_sweep_n1_forced_dist.txt).
        self.inplace = bucket_dtype == torch.float32 and os.environ.get("NNDET_DDP_INPLACE", "1") != "0"
        self.copied_last = 0            # parameters whose gradient had to be copied into its bucket in the last backward pass
        self._copied = 0
        if static_unused is None and hasattr(model, "never_used_parameters"):
            static_unused = model.never_used_parameters()
        self._static_unused = {id(p) for p in (static_unused or [])}
        self._marked = set()            # parameters declared gradient-free for the CURRENT step (mark_no_grad)
        self._names = {id(p): n for n, p in model.named_parameters()}
        self._where = {}
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b.params):
                self._where[p] = (bi, pi)
            b.expected = sum(1 for p in b.params if id(p) not in self._static_unused)
            b.pending = b.expected
        self.force = bool(force_overlap)
        self.overlap = overlap and (self.world > 1 or self.force)
        self._cuda = device.type == "cuda"
        self._next = 0
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._dev_index = device.index if device.index is not None else (torch.cuda.current_device() if self._cuda else 0)
        self._stream_objs = {}          # raw stream handle -> torch.cuda.Stream
        self._comm = None               # communication stream (bucket copies + collectives), created on first use
        self._event_pool, self._used_events = [], []
        self._hooks = []
        # the mean is taken by the collective itself where the backend can (RCCL / NCCL: ReduceOp.AVG), else by one scaling pass
        backend = dist.get_backend(self.pg) if dist.is_initialized() else ""
        self._avg_in_collective = self.world > 1 and backend == "nccl" and hasattr(dist.ReduceOp, "AVG") and \
            os.environ.get("NNDET_DDP_AVG", "1") != "0"
        self.profile = bool(profile) and self._cuda
        self._prof = {"steps": 0, "exposed_ms": 0.0, "span_ms": 0.0, "offsets_ms": [0.0] * len(self.buckets)}
        self._prof_ev = None
        if self.overlap:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        if self.inplace and (self.world > 1 or self.force):
            from . import _lib as L
            L.grad_pool.install_static({p: v.reshape(-1) for b in self.buckets for p, v in zip(b.params, b.views)}, self._flat_all,
                                       owner=model)
            self._no_grad_cb = self.mark_no_grad
            L.no_grad_listeners.append(self._no_grad_cb)
            self._begin_cb = self.reset_step_marks
            L.step_begin_listeners.append(self._begin_cb)
        self.check_layout_across_ranks()
        self.broadcast_parameters(model)

    def close(self):
        """Detach from the model: hooks, the static gradient layout and the no-gradient listener."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        from . import _lib as L
        L.grad_pool.remove_static(self._flat_all)
        cb = getattr(self, "_no_grad_cb", None)
        if cb is not None and cb in L.no_grad_listeners:
            L.no_grad_listeners.remove(cb)
        cb = getattr(self, "_begin_cb", None)
        if cb is not None and cb in L.step_begin_listeners:
            L.step_begin_listeners.remove(cb)

    def layout_signature(self) -> str:
        """What every rank must agree on before the first collective: bucket boundaries (sizes in elements), the parameters of each
        bucket by NAME, shape and order, the never-used set, the bucket dtype and whether the collective averages."""
        import hashlib
        h = hashlib.sha256()
        h.update(("dtype=%s avg=%d world=%d\n" % (self.bucket_dtype, int(self._avg_in_collective), self.world)).encode())
        for bi, b in enumerate(self.buckets):
            h.update(("bucket %d numel=%d expected=%d\n" % (bi, b.numel, b.expected)).encode())
            for p in b.params:
                h.update(("  %s %s unused=%d\n" % (self._names.get(id(p), "?"), tuple(p.shape), int(id(p) in self._static_unused))).encode())
        return h.hexdigest()

    def check_layout_across_ranks(self) -> None:
        """All ranks must have built the SAME bucket layout and never-used set: a mismatch (a different NNDET_DDP_* environment on one
        rank, a model built from another plan, a parameter frozen on one rank only) makes the ranks issue collectives of different
        sizes -- on RCCL that is a hang of the whole 8-GPU job, not an error. One all-gather of a 32-byte digest at construction."""
        self.layout_digest = self.layout_signature()
        if self.world <= 1 or not dist.is_initialized():
            return
        mine = torch.tensor(list(bytes.fromhex(self.layout_digest)), dtype=torch.uint8)
        backend = dist.get_backend(self.pg)
        if backend == "nccl":
            mine = mine.to(self.buckets[0].flat.device)
        got = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(got, mine, group=self.pg)
        digests = [bytes(t.cpu().tolist()).hex() for t in got]
        bad = [r for r, d in enumerate(digests) if d != digests[0]]
        if bad:
            rank = dist.get_rank(self.pg)
            raise RuntimeError(
                "GradAllReducer: the gradient bucket layout differs between ranks (rank 0: %s..., ranks %s differ; this is rank %d with "
                "%d buckets of %s elements, %d never-used parameters, first/bucket MB %s/%s). Every rank must build the same model "
                "and use the same NNDET_DDP_FIRST_MB / NNDET_DDP_BUCKET_MB / NNDET_DDP_BF16 / NNDET_DDP_AVG settings."
                % (digests[0][:12], bad, rank, len(self.buckets), [b.numel for b in self.buckets], len(self._static_unused),
                   self.first_bucket_mb, self.bucket_mb))

    def reset_step_marks(self) -> None:
        """A new top-level training forward pass begins (called by _lib.grad_pool.begin): declarations of a forward pass that was never
        followed by backward + finish() (skipped step, exception, NaN guard) must not leak into this step."""
        if self._marked and self._next == 0:
            for b in self.buckets:
                b.pending = b.expected
            self._marked.clear()

    def mark_no_grad(self, params):
        """These parameters will get NO gradient in the coming backward pass (told during the forward pass: the regressor on a rank
        whose batch has no positive anchor, nndet/arch/heads/comb.py:397-401). They contribute zeros and no longer hold their bucket
        back: without this, that bucket AND every later one (collectives are issued in order) could only be launched from finish(),
        i.e. this rank would overlap nothing and the other ranks would wait for it."""
        if not self.overlap:
            return
        for p in params:
            w = self._where.get(p)
            if w is None or id(p) in self._static_unused or id(p) in self._marked:
                continue
            self._marked.add(id(p))
            self.buckets[w[0]].pending -= 1
        # (buckets are launched by the next gradient hook: nothing is issued from inside the forward pass)

    def broadcast_parameters(self, model):
        """Rank 0's parameters / buffers to everybody. Written through the tensors themselves under no_grad (NOT `.data`), so
        `_version` is bumped and the packed-weight caches of the conv blocks (keyed on version + data_ptr) are refreshed."""
        if self.world > 1:
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t, src=0, group=self.pg)
            for m in model.modules():
                if hasattr(m, "_pack_cache"):
                    m._pack_cache.clear()

    def _launch(self, b: GradBucket):
        # Gradients of one bucket are accumulated on DIFFERENT streams (the detection-head branches run on side streams,
        # arch/heads.py; autograd only joins them at the end of backward). The bucket copy below runs on the stream of whichever
        # parameter completed the bucket: it first waits for every other stream that contributed. All contributions are already
        # enqueued (every hook of the bucket has fired), so ONE event per contributing stream, recorded now, covers them -- the hooks
        # themselves only note a raw stream handle (an event per parameter cost ~15 us x 92 hooks inside the autograd thread).
        # The copy and the collective are issued on a COMMUNICATION stream that waits for the contributing streams -- among them the
        # weight-gradient stream (_lib.wgrad_streams), which lags behind the data-gradient chain by design. Waiting for it on the
        # stream of the completing parameter (as the first version did) stalled the critical chain at every bucket: +1.3 ms per step.
        comm = None
        if self._cuda:
            cur = torch.cuda.current_stream()
            if self._comm is None:
                from . import _lib as _L
                self._comm = _L.new_stream("comm", cur.device)
            comm = self._comm
            self._stream_objs.setdefault(cur.cuda_stream, cur)
            b.streams.add(cur.cuda_stream)
            for sid in b.streams:
                ev = self._event_pool.pop() if self._event_pool else torch.cuda.Event()
                ev.record(self._stream_objs[sid])
                comm.wait_event(ev)
                self._used_events.append(ev)
            b.streams.clear()
        ctx = torch.cuda.stream(comm) if comm is not None else contextlib.nullcontext()
        with ctx:
            if self.profile and self._prof_ev is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()                              # on the communication stream: the bucket's copy starts here
                self._prof_ev["launch"].append(ev)
            src, dst = [], []
            zeroed = self._pool_zeroed()                 # this step's buckets were zero-filled by the gradient pool (static layout)
            for p, v in zip(b.params, b.views):
                g = p.grad
                if g is None:
                    if not zeroed:
                        v.zero_()                        # unused on this rank: contributes zeros
                elif g.data_ptr() == v.data_ptr() and g.dtype == v.dtype and g.is_contiguous():
                    pass                                 # written in place by its autograd node
                else:
                    src.append(g); dst.append(v)
            if dst:
                torch._foreach_copy_(dst, src)           # one multi-tensor kernel per bucket instead of one copy per parameter
            self._copied += len(dst)
            if self.world > 1 or (self.force and dist.is_initialized()):
                op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
                b.work = dist.all_reduce(b.flat, op=op, group=self.pg, async_op=True)
            elif self._cuda:
                b.work = torch.cuda.Event()
                b.work.record()                          # world 1 (force_overlap): finish() still orders against the copy stream

    def _pool_zeroed(self) -> bool:
        from . import _lib as L
        return bool(L.grad_pool.armed and L.grad_pool.static_flat is self._flat_all)

    def begin_step(self):
        """profile=True only: call right before backward(); marks t = 0 of the per-bucket launch offsets."""
        if self.profile:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self._prof_ev = {"t0": e0, "launch": []}

    def profile_summary(self) -> dict:
        """Averages over the profiled steps: `exposed_allreduce_ms` = GPU time the caller's stream spent in finish() waiting for
        buckets (communication that the backward pass did not hide), `backward_to_ready_ms` = backward start -> gradients ready,
        `bucket_launch_offset_ms[i]` = backward start -> bucket i's copy + all-reduce issued."""
        n = max(1, self._prof["steps"])
        return {"steps": self._prof["steps"], "world": self.world, "first_bucket_mb": self.first_bucket_mb, "bucket_mb": self.bucket_mb,
                "bucket_dtype": str(self.bucket_dtype).replace("torch.", ""), "avg_in_collective": bool(self._avg_in_collective),
                "bucket_mbytes": [round(b.numel * b.flat.element_size() / 2 ** 20, 2) for b in self.buckets],
                "exposed_allreduce_ms": round(self._prof["exposed_ms"] / n, 4),
                "backward_to_ready_ms": round(self._prof["span_ms"] / n, 4),
                "bucket_launch_offset_ms": [round(v / n, 4) for v in self._prof["offsets_ms"]]}

    def _on_grad(self, p):
        bi, _ = self._where[p]
        if id(p) in self._static_unused:
            raise RuntimeError("a parameter declared as never used received a gradient")
        if id(p) in self._marked:
            if p.grad is None:
                return                                   # (the engine runs the accumulation node of an input whose gradient a Function
                                                         #  returned as None, with an undefined gradient: nothing arrived)
            raise RuntimeError("parameter %s was declared gradient-free for this step (mark_no_grad) but received a gradient"
                               % self._names.get(id(p), "?"))
        b = self.buckets[bi]
        b.pending -= 1
        if self._cuda:
            sid = self._raw_stream(self._dev_index) if self._raw_stream is not None else torch.cuda.current_stream().cuda_stream
            if sid not in self._stream_objs:
                self._stream_objs[sid] = torch.cuda.current_stream()     # the stream this gradient was accumulated on
            b.streams.add(sid)
            from . import _lib as L                      # the weight-gradient kernels themselves run on their own stream (_lib.wgrad_streams)
            for ws in L.wgrad_streams.active.values():
                self._stream_objs.setdefault(ws.cuda_stream, ws)
                b.streams.add(ws.cuda_stream)
        # collectives must be issued in the SAME order on every rank: a bucket is only launched once all
        # earlier buckets are (a bucket holding a parameter that is unused on this rank is launched by finish())
        while self._next < len(self.buckets) and self.buckets[self._next].pending <= 0:
            self.buckets[self._next].from_hook = True
            self._launch(self.buckets[self._next])
            self._next += 1

    def finish(self):
        """Wait for all buckets (launching those whose parameters never produced a gradient) and write the mean back."""
        if self.world == 1 and not self.force:
            return
        while self._next < len(self.buckets):
            self.buckets[self._next].from_hook = False
            self._launch(self.buckets[self._next])
            self._next += 1
        self.launched_from_hooks = [b.from_hook for b in self.buckets]      # (tests / diagnostics: which buckets overlapped the backward pass)
        self.copied_last, self._copied = self._copied, 0
        self._marked.clear()
        if self.inplace:
            from . import _lib as L
            if L.grad_pool.static_flat is self._flat_all:
                L.grad_pool.armed = False                # the buckets now hold REDUCED gradients: regions are handed out again after begin()
        inv = 1.0 / self.world
        prof = self._prof_ev if self.profile else None
        if prof is not None:
            prof["f0"] = torch.cuda.Event(enable_timing=True)
            prof["f0"].record()                          # caller's stream: the backward pass is done here
        for b in self.buckets:
            if self._cuda and isinstance(b.work, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(b.work)
            elif b.work is not None:
                b.work.wait()
            if self.world > 1 and not self._avg_in_collective:
                b.flat.mul_(inv)                         # (gloo has no ReduceOp.AVG; RCCL averages inside the collective)
            # hand the averaged gradients to the optimizer as views of the bucket (no copy back). They stay valid until the
            # bucket is refilled by the next backward; the training loop drops them with zero_grad(set_to_none=True).
            if b.flat.dtype == torch.float32:
                for p, v in zip(b.params, b.views):
                    p.grad = v
            else:                                        # 16-bit buckets: the optimizer wants fp32 gradients
                for p, v in zip(b.params, b.views):
                    p.grad = v.float()
            b.work = None
            b.pending = b.expected
        self._next = 0
        self._event_pool.extend(self._used_events)       # every wait on them has been enqueued: safe to re-record next step
        self._used_events.clear()
        if prof is not None:
            f1 = torch.cuda.Event(enable_timing=True)
            f1.record()
            self._prof_pending = getattr(self, "_prof_pending", [])
            self._prof_pending.append((prof["t0"], prof["f0"], f1, prof["launch"]))
            self._prof_ev = None

    def profile_reset(self):
        """Drop everything recorded so far (bench.py: after the warm-up steps, whose first one includes one-time set-up work)."""
        self.profile_collect()
        self._prof = {"steps": 0, "exposed_ms": 0.0, "span_ms": 0.0, "offsets_ms": [0.0] * len(self.buckets)}

    def profile_collect(self):
        """Fold the event pairs of the finished steps into the running sums (synchronises: call outside the timed region)."""
        for t0, f0, f1, launches in getattr(self, "_prof_pending", []):
            f1.synchronize()
            self._prof["steps"] += 1
            self._prof["exposed_ms"] += f0.elapsed_time(f1)
            self._prof["span_ms"] += t0.elapsed_time(f1)
            for i, ev in enumerate(launches[:len(self.buckets)]):
                self._prof["offsets_ms"][i] += t0.elapsed_time(ev)
        self._prof_pending = []
