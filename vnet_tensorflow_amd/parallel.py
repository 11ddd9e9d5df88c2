"""Data-parallel training over the GPUs of one node: one process per GPU, patches sharded across
ranks, ONE exchange step per iteration -- a sum all-reduce (RCCL over xGMI; backend "nccl" is RCCL
on ROCm) of the flat fp32 gradient buffer, cut into contiguous buckets in backward-production
order and launched from autograd hooks on a side stream so it overlaps the remaining backward
convolutions.  The reference has no multi-GPU code; semantics are those of
SURVEY.md 8(e): per-replica batch-norm statistics, gradient = mean of per-rank gradients.

The class is device-agnostic (gloo on CPU tensors exercises exactly the same bucket/hook logic)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group torchrun / torch.distributed.run prepared (RANK, WORLD_SIZE, MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("VNET_DIST_BACKEND")          # test hook: gloo with GPU tensors on a 1-GPU box
        if backend == "gloo" and 0 < torch.cuda.device_count() <= local:
            local %= torch.cuda.device_count()                      # test hook only: several ranks share one device
        if backend is None:
            backend = "nccl" if torch.cuda.device_count() > 0 else "gloo"   # device_count() does not initialise the GPU
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def broadcast_parameters(flat_data, src=0, group=None):
    """All replicas start from rank 0's weights (the reference initialises from an unseeded NumPy RNG)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_data, src=src, group=group)


class BucketedGradAllReduce(object):
    """All-reduces `flat.grad` bucket by bucket as soon as every variable of a bucket has its gradient."""

    def __init__(self, flat, bucket_bytes=32 << 20, group=None, overlap=True, force=False, hold_fraction=0.0, phase1_last=None,
                 comm_dtype="fp32"):
        """`force`: run the collective path even in a group of one (lets a 1-GPU box exercise RCCL itself).
        `comm_dtype`: "fp32" (default) -- a sum all-reduce of the fp32 bucket; "bf16" (opt-in, TrainingSetting.GradCommDtype with
        ComputeDtype "bf16": everything else of that mode is bf16 already and at a 5 ms step the fp32 all-reduce of 175.8 MB is the
        largest exposed cost) -- HALF the bytes on the links, fp32 ACCUMULATION on receipt: every rank rounds
        its bucket to bf16 (RNE), an all-to-all hands rank r the r-th 1/N slice of every rank's bucket, rank r adds the N slices
        in fp32 in rank order (deterministic, independent of the link topology), rounds the sum to bf16 once and an all-gather
        returns the slices; the fp32 master gradient (flat.grad) receives the bf16 sums.  Two roundings per element (2^-8 relative each)
        against none for "fp32"; same traffic as a bf16 ring all-reduce, without its N - 1 bf16 additions along the ring.
        `hold_fraction`: ready buckets are held back until this fraction of the gradient bytes is ready, then all of
        them go out back to back (0 = every bucket as soon as it is ready).  V-Net produces 98 % of its gradient bytes
        (decoder, bottom level, encoder level 4/3) while the backward pass is in the 32^3...8^3 levels, whose kernels are
        planned for exactly 256 CUs and take ~1.9x while a collective holds some of them (measured);
        the 64^3 / 128^3 kernels that follow lose 1.16-1.25x and last longer than the whole all-reduce."""
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.hold_fraction = float(hold_fraction)
        if comm_dtype not in ("fp32", "bf16"):
            raise ValueError("comm_dtype %r (fp32 | bf16)" % (comm_dtype,))
        self.comm_dtype = comm_dtype
        self.comm_bytes = 0              # bytes this rank SENDS over the links in the current step (ring all-reduce resp. all-to-all + all-gather)
        self._exposed = None             # (start, end) of the last reduce_all(): the all-reduce nothing overlaps (StepModeAutotune)
        # a bucket boundary exactly where the cumulative bytes cross hold_fraction, so the launch point does not depend
        # on where the size-driven cuts happen to fall (without it the 0.99 threshold of the V-Net layout is only crossed
        # by the LAST bucket and nothing overlaps backward)
        cut = [flat.first_index_reaching(self.hold_fraction)] if 0.0 < self.hold_fraction < 1.0 else []
        # phase1_last: index (gradient-production order) of the last variable whose gradient the FIRST backward pass writes
        # (output layer, decoder, bottom level); a bucket ends there, so reduce_prefix() can send exactly those buckets
        self.phase1_last = phase1_last
        if phase1_last is not None:
            cut = cut + [int(phase1_last)]
        self.buckets = flat.buckets(bucket_bytes, cut_after=cut)
        self.hold_all = False            # graph mode "segmented": hooks only count, reduce_all() launches everything
        self.launch_log = []             # (bucket index, gradient events seen so far) per launch of the current step
        self.is_cuda = flat.grad.is_cuda
        self.overlap = overlap and self.is_cuda and self.active
        self.comm_stream = torch.cuda.Stream(device=flat.grad.device) if self.overlap else None
        self._bucket_of = {}
        for bi, (_, _, first, last) in enumerate(self.buckets):
            for pi in range(first, last):
                self._bucket_of[pi] = bi
        self._bytes = [4 * (e - s_) for s_, e, _, _ in self.buckets]
        self._total_bytes = float(sum(self._bytes)) or 1.0
        self._ready_bytes = 0
        self._held = []
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self._hooks = []
        self._expected = None
        self._events = []
        if self.active:
            for pi, p in enumerate(flat.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(pi)))
                sink = getattr(p, "_vnet_sink", None)
                if sink is not None:         # gradients written in place by the HIP backward kernels
                    sink.ready = (lambda pi=pi: self._count(pi))
        self.begin_step()

    def begin_step(self):
        """Per step.  A bucket is launched when every gradient EVENT of its variables has arrived.  A variable can
        receive more than one event per backward pass (e.g. the input batch-norm's gamma/beta get a direct kernel
        write from their own layer AND an autograd accumulation from the fused input conv), so the event counts are
        calibrated on the first step (no early launches) and checked on every later one."""
        n = len(self.flat.params)
        self._events = [0] * n
        for bi, (_, _, first, last) in enumerate(self.buckets):
            self._pending[bi] = (sum(self._expected[first:last]) if self._expected is not None else -1)
            self._launched[bi] = False
        self._handles = []
        self._ready_bytes = 0
        self._held = []
        self.launch_log = []
        self.comm_bytes = 0

    def exposed_seconds(self):
        """Duration of the last reduce_all() -- every bucket's all-reduce after backward with nothing to hide under ('serial' step
        mode): HIP events on the compute stream around launch + wait (synchronises), wall clock on CPU.  None if there was none."""
        if self._exposed is None:
            return None
        a, b = self._exposed
        if isinstance(a, float):
            return b - a
        b.synchronize()
        return a.elapsed_time(b) * 1e-3

    def reduce_prefix(self):
        """Two-pass backward (model.train_step, segmented graphs): all-reduce the buckets the first pass completed, on the
        communication stream, WITHOUT waiting -- the second pass (encoder backward) runs meanwhile."""
        if not self.active:
            return
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self.launch_log = []
        for bi, (_, _, _, last) in enumerate(self.buckets):
            if self.phase1_last is not None and last <= self.phase1_last + 1:
                self._launch(bi)

    def reduce_rest(self):
        """The remaining buckets, then make the current stream wait for every all-reduce of this step."""
        if not self.active:
            return
        for bi in range(len(self.buckets)):
            self._launch(bi)                 # (skips the ones reduce_prefix launched)
        for h in self._handles:
            h.wait()
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._handles = []

    def reduce_all(self):
        """All-reduce every bucket now (the gradients are complete on the current stream) and make the current stream
        wait for the result: the exchange step between the gradients graph and the optimiser graph (model.train_step,
        'segmented' mode).  Leaves the event calibration untouched."""
        if not self.active:
            return
        import time
        if self.is_cuda:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
        else:
            t0 = time.perf_counter()
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self.launch_log = []
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for h in self._handles:
            h.wait()
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._handles = []
        if self.is_cuda:
            t1.record()
        else:
            t1 = time.perf_counter()
        self._exposed = (t0, t1)

    def _make_hook(self, pi):
        def hook(param):
            # autograd may have swapped .grad for a fresh tensor: fold it back into the flat buffer
            o = self.flat.offsets[pi]
            g = param.grad
            if g is not None and g.data_ptr() != self.flat.grad.data_ptr() + 4 * o:
                view = self.flat.grad[o:o + param.numel()].view(param.shape)
                view.add_(g)
                param.grad = view
            self._count(pi)
        return hook

    def _count(self, pi):
        self._events[pi] += 1
        if self._expected is None or self.hold_all:
            return                      # calibration step / segmented graph mode: everything is reduced after backward
        bi = self._bucket_of[pi]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._ready_bytes += self._bytes[bi]
            if self._ready_bytes < self.hold_fraction * self._total_bytes:
                self._held.append(bi)                  # goes out with the bucket that crosses the threshold (or in finish())
                return
            for hb in self._held:
                self._launch(hb)
            self._held = []
            self._launch(bi)

    def _launch(self, bi):
        if self._launched[bi] or not self.active:
            return
        self._launched[bi] = True
        self.launch_log.append((bi, sum(self._events)))
        s, e, _, _ = self.buckets[bi]
        view = self.flat.grad[s:e]
        pg = None
        if self.is_cuda:
            from . import ops
            pg = ops.param_grad_stream(view.device, create=False)      # filter/bias gradients are written on this stream
        if self.overlap:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            if pg is not None:
                self.comm_stream.wait_stream(pg)
            with torch.cuda.stream(self.comm_stream):
                self._all_reduce(view)
        else:
            if pg is not None:
                torch.cuda.current_stream().wait_stream(pg)
            self._all_reduce(view)

    def _all_reduce(self, view):
        """Sum `view` (a slice of the flat fp32 gradient) over the ranks, in place, on the current stream."""
        if self.comm_dtype == "fp32":
            self.comm_bytes += 2 * (self.world - 1) * 4 * view.numel() // max(self.world, 1)       # reduce-scatter + all-gather halves of a ring
            self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        # bf16 on the links, fp32 accumulation on receipt (see __init__).  Stream-ordered: each collective is enqueued behind the
        # previous one on this stream (async_op=False does not block the host with RCCL; gloo on CPU tensors is synchronous anyway).
        W, n = self.world, view.numel()
        chunk = -(-n // W)
        dev = view.device
        if view.is_cuda and dist.get_backend(self.group) == "gloo":
            dev = torch.device("cpu")                               # test hook (several ranks on one GPU over gloo): gloo has no device all-to-all
        send = torch.zeros(W * chunk, dtype=torch.bfloat16, device=view.device)
        send[:n].copy_(view)                                        # RNE fp32 -> bf16
        if W == 1:
            view.copy_(send[:n])
            return
        send = send.to(dev)
        recv = torch.empty(W * chunk, dtype=torch.bfloat16, device=dev)
        dist.all_to_all_single(recv, send, group=self.group)         # slice r of every rank's bucket -> rank r
        part = recv.view(W, chunk)
        acc = part[0].to(torch.float32)
        for r in range(1, W):                                       # fixed rank order: bit-identical on every run and every rank
            acc += part[r]
        mine = acc.to(torch.bfloat16)
        full = torch.empty(W * chunk, dtype=torch.bfloat16, device=dev)
        dist.all_gather_into_tensor(full, mine, group=self.group)
        view.copy_(full[:n])
        self.comm_bytes += 2 * (W - 1) * 2 * chunk                  # (W - 1) slices out in the all-to-all, the own slice to W - 1 peers in the all-gather
        for t in (send, recv, full, mine):
            if t.is_cuda:
                t.record_stream(torch.cuda.current_stream())

    def finish(self):
        """Flush buckets that are still pending (calibration step; variables that never receive a gradient such as
        the dead batch-norms), then make the compute stream wait for every all-reduce.  The optimiser divides by
        world (gscale)."""
        if not self.active:
            return
        if self._expected is not None and self._events != self._expected:
            early = [bi for bi, (_, _, f, l) in enumerate(self.buckets)
                     if self._launched[bi] and any(self._events[pi] > self._expected[pi] for pi in range(f, l))]
            if early:
                raise RuntimeError("gradient events changed between steps (dynamic graph?): buckets %s were all-reduced "
                                   "before their last gradient arrived" % early)
        for bi in range(len(self.buckets)):
            self._launch(bi)
        for h in self._handles:
            h.wait()
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._handles = []
        self._expected = list(self._events)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


class StepModeAutotune(object):
    """Start-up choice between ways of running the data-parallel step (model.image2label: 'segmented' hipGraph replay with the
    all-reduce between the two graphs vs the same graphs with every all-reduce after backward ('serial') vs the eager
    kernel-by-kernel enqueue whose bucket all-reduces overlap backward).  Which one wins depends on the machine: the eager
    enqueue costs 6-20 ms of host time per 128^3 step (8 ranks share the host's cores), the serial replay costs no host time but
    exposes the all-reduce (~1-2.5 ms), the overlapped replay hides it but shares the CUs with a resident collective.

    Round 3 (one noisy 5-step block used to decide the whole job): the candidates are measured in
    `blocks` >= 3 interleaved rounds (c0 c1 c2 c0 c1 c2 ...) of `steps` REAL training steps each; a block's wall time is
    max-reduced over the ranks (so every rank sees the same numbers and picks the same winner); a candidate's score is the MEDIAN
    of its blocks; and `prefer` (the serial replay: no collective ever shares the chip with an MFMA kernel sized for all 256 CUs,
    the mode whose time is the easiest to predict) is kept unless another candidate's median is more than `margin` (2 %) faster.
    Usage per step:  mode = tuner.mode();  tuner.before();  <run the step in that mode>;  tuner.after()."""

    def __init__(self, candidates, steps=5, group=None, sync=None, clock=None, blocks=3, prefer="serial", margin=0.02,
                 exposed=None, exposed_threshold=0.05, overlapped=("segmented", "off")):
        """`exposed`: callable -> seconds the last 'serial' step spent in its all-reduce with nothing overlapping it (or None);
        sampled at the end of every 'serial' block.  Round 6: the preference is STEP-LENGTH-AWARE -- 'serial'
        is the default only while that exposed time stays below `exposed_threshold` (5 %) of its step (fp32 128^3: <= 2.5 ms of
        25 ms can be borderline, fp32_split3 16 ms and bf16 5.3 ms are not: there 1-2.5 ms is 20-45 % of the step); above it
        the preferred mode is the fastest OVERLAPPED candidate (`overlapped`, in that order of preference when they tie within the
        margin), and 'serial' has to beat it by more than `margin` to be chosen."""
        import time
        self.candidates, self.steps, self.group = list(candidates), int(steps), group
        self.blocks = max(1, int(blocks))
        self.prefer, self.margin = prefer, float(margin)
        self.sync = sync if sync is not None else (lambda: None)          # device synchronisation
        self.clock = clock if clock is not None else time.perf_counter
        self.exposed, self.exposed_threshold, self.overlapped = exposed, float(exposed_threshold), tuple(overlapped)
        self.exposed_samples = []                                         # seconds per 'serial' step in the exposed all-reduce
        self.exposed_fraction = None
        self.preferred = prefer
        self.samples = [[] for _ in self.candidates]                      # per candidate: seconds per step of each block
        self.times, self._i, self._n, self._t0 = [], 0, 0, None
        self.choice = self.candidates[0] if len(self.candidates) == 1 else None

    def total_steps(self):
        """Training steps the measurement takes (0 for a single candidate)."""
        return 0 if len(self.candidates) == 1 else len(self.candidates) * self.blocks * self.steps

    def mode(self):
        return self.choice if self.choice is not None else self.candidates[self._i % len(self.candidates)]

    def before(self):
        if self.choice is None and self._n == 0:
            self.sync()
            if dist.is_initialized():
                dist.barrier(group=self.group)
            self._t0 = self.clock()

    @staticmethod
    def _median(v):
        v = sorted(v)
        n = len(v)
        return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])

    def after(self):
        if self.choice is not None:
            return
        self._n += 1
        if self._n < self.steps:
            return
        self.sync()
        t = torch.tensor([self.clock() - self._t0], dtype=torch.float64)
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
            t = t.to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        cur = self.candidates[self._i % len(self.candidates)]
        self.samples[self._i % len(self.candidates)].append(float(t.item()) / self.steps)
        if cur == "serial" and self.exposed is not None:
            e = self.exposed()
            e = torch.tensor([float(e) if e is not None else 0.0], dtype=torch.float64)
            if dist.is_initialized() and dist.get_world_size(self.group) > 1:
                e = e.to(t.device)
                dist.all_reduce(e, op=dist.ReduceOp.MAX, group=self.group)     # every rank decides on the same number
            self.exposed_samples.append(float(e.item()))
        self._i, self._n = self._i + 1, 0
        if self._i == len(self.candidates) * self.blocks:
            self.times = [self._median(v) for v in self.samples]
            best = min(range(len(self.times)), key=lambda k: (self.times[k], k))
            prefer = self.prefer
            if prefer == "serial" and "serial" in self.candidates and self.exposed_samples:
                ks = self.candidates.index("serial")
                self.exposed_fraction = self._median(self.exposed_samples) / max(self.times[ks], 1e-12)
                if self.exposed_fraction > self.exposed_threshold:
                    over = [c for c in self.overlapped if c in self.candidates]
                    if over:                     # the exposed all-reduce is a real share of this step: prefer hiding it
                        prefer = min(over, key=lambda c: (self.times[self.candidates.index(c)], over.index(c)))
            self.preferred = prefer
            if prefer in self.candidates:
                k = self.candidates.index(prefer)
                if self.times[best] >= (1.0 - self.margin) * self.times[k]:
                    best = k                     # nobody beats the preferred mode by more than the margin
            self.choice = self.candidates[best]
