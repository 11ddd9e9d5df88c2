"""Flat parameter/gradient buffers + the TF1 optimisers the reference selects (model.py:641-666).

All trainable variables live in ONE contiguous fp32 buffer (and their gradients in another), laid
out in reverse creation order = the order autograd produces gradients (output layer, decoder
level 1 .. 4, bottom, encoder level 4 .. 1).  That makes (a) the optimiser one fused HIP kernel
over 44 M parameters and (b) data-parallel gradient buckets contiguous slices that RCCL can
all-reduce in place while backward is still running (parallel.py)."""
import math

import torch

from . import ops


class FlatParams(object):
    def __init__(self, named_params):
        """named_params: list of (tf_name, Parameter) in creation (= forward) order."""
        self.names = [n for n, _ in reversed(named_params)]
        params = [p for _, p in reversed(named_params)]
        self.params = params
        dev = params[0].device
        # 4-float (16 B) alignment for every variable so vector loads on slices stay aligned
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                n = p.numel()
                self.data[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.data[o:o + n].view(p.shape)
                p.grad = self.grad[o:o + n].view(p.shape)
                p._vnet_sink = ops.GradSink(p.grad)      # backward kernels write here directly
        ops.invalidate_packed()
        self.needs_zero = False     # set when autograd was seen accumulating into the buffer (see begin_step / check_accumulation)
        self._ver = None
        self._steps = 0

    def begin_step(self):
        """Start of a training step WITHOUT re-zeroing the gradient buffer: every gradient the networks produce is written
        (not accumulated) by its backward kernel straight into its slice, a variable that gets a second contribution from
        autograd receives it as `+=` AFTER that write (the AccumulateGrad node runs when all its edges have reported), and
        the slices nobody writes -- conv biases in front of batch-norms (closed form, exact 0), the dead batch-norms,
        alignment padding -- still hold the zeros of construction.  Saves a 176 MB memset per step.

        That is only sound while NO gradient reaches a variable purely through autograd's AccumulateGrad (`p.grad += g` onto the
        previous step's -- under data parallelism already all-reduced -- value): a parameter used outside the fused ops, a
        data-parallel hook folding a temporary back, a skipped backward cut.  Such an in-place add bumps the
        version counter the .grad views share with the flat buffer, kernel writes through raw pointers do not; so the first
        step of a FlatParams clears the buffer like zero_grad(), every pass is checked (check_accumulation), and once an
        accumulation has been seen every later step clears first.  (A replayed step graph repeats the launches of the eager
        steps it was captured from, which were checked.)"""
        self._steps += 1
        if self.needs_zero or self._steps == 1:
            self.zero_grad()
            self._ver = self.grad._version
            return
        self._ver = self.grad._version
        for p, o in zip(self.params, self.offsets):
            p._vnet_sink.written = False
            g = p.grad
            if g is None or g.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
                p._vnet_sink.view = p.grad

    def check_accumulation(self):
        """After a backward pass that began with begin_step(): did autograd add into the buffer in place?"""
        if self._ver is None or self.grad._version == self._ver:
            return
        if not self.needs_zero and self._steps > 1:
            self.needs_zero = True
            raise ops.VnetHipError(
                "a gradient was accumulated by autograd (p.grad += g) onto the previous step's value: the gradient buffer was not "
                "cleared for this step (FlatParams.begin_step).  This step's gradients are invalid; later steps clear the buffer "
                "first.  (A parameter used outside the fused ops / a changed network between steps; call flat.zero_grad() "
                "yourself or set flat.needs_zero = True before training.)")
        self.needs_zero = True

    def zero_grad(self):
        self.grad.zero_()
        for p in self.params:
            p._vnet_sink.written = False
        # autograd may have replaced .grad objects; re-point them at the flat buffer
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None or g.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
                p._vnet_sink.view = p.grad

    def buckets(self, bucket_bytes=32 << 20, cut_after=()):
        """Contiguous [start, end) slices of the flat buffer, cut at variable boundaries, plus the
        index range of the variables each one holds.  `cut_after`: variable indices that must END a bucket."""
        out, start, first = [], 0, 0
        cut_after = set(cut_after)
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            end = o + (p.numel() + 3) // 4 * 4
            if (end - start) * 4 >= bucket_bytes or i == len(self.params) - 1 or i in cut_after:
                out.append((start, end, first, i + 1))
                start, first = end, i + 1
        return out

    def first_index_reaching(self, fraction):
        """Index of the variable (gradient-production order) with which the cumulative gradient bytes reach
        `fraction` of the total."""
        total = float(self.numel)
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            if (o + (p.numel() + 3) // 4 * 4) >= fraction * total:
                return i
        return len(self.params) - 1


def exponential_decay(lr0, global_step, decay_steps, decay_rate):
    """tf.train.exponential_decay(..., staircase=False) (reference model.py:642-643)."""
    return lr0 * decay_rate ** (global_step / float(decay_steps))


class _Base(object):
    """apply(lr) = schedule(lr) + launch(...).  `schedule` is the host-side scalar bookkeeping of one step (returns the
    learning-rate scalar the kernel uses); `launch` enqueues the fused update + the batched filter repack.  With
    `state` (ops.step_state) the kernel reads its scalar from device memory, so the launch can sit inside a captured
    hipGraph while `schedule` + ops.set_step_state run on the host before each replay (model.image2label)."""

    def __init__(self, flat):
        self.flat = flat
        self.gscale = 1.0       # 1/world_size under data parallelism (mean of per-rank gradients)

    def schedule(self, lr):
        return lr

    def apply(self, lr):
        self.launch(self.schedule(lr))

    def _after(self):
        ops.invalidate_packed()
        ops.repack_registered()

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass


class GradientDescentOptimizer(_Base):
    """tf.train.GradientDescentOptimizer"""

    def launch(self, lr, state=None):
        ops.sgd_apply(self.flat.data, self.flat.grad, lr, self.gscale, state=state)
        self._after()


class AdamOptimizer(_Base):
    """tf.train.AdamOptimizer (epsilon-hat form): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    p -= lr_t * m / (sqrt(v) + eps)."""

    def __init__(self, flat, beta1=0.9, beta2=0.999, epsilon=1e-8):
        super().__init__(flat)
        self.b1, self.b2, self.eps, self.t = beta1, beta2, epsilon, 0
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)

    def schedule(self, lr):
        self.t += 1
        return lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)

    def launch(self, lr_t, state=None):
        ops.adam_apply(self.flat.data, self.flat.grad, self.m, self.v, lr_t, self.b1, self.b2, self.eps, self.gscale, state=state)
        self._after()

    def state_dict(self):
        return {"t": self.t, "m": self.m, "v": self.v}

    def load_state_dict(self, sd):
        self.t = int(sd["t"])
        self.m.copy_(sd["m"].to(self.m.device))
        self.v.copy_(sd["v"].to(self.v.device))


class MomentumOptimizer(_Base):
    """tf.train.MomentumOptimizer(use_nesterov=...)"""

    def __init__(self, flat, momentum=0.9, use_nesterov=False):
        super().__init__(flat)
        self.momentum, self.nesterov = momentum, use_nesterov
        self.acc = torch.zeros_like(flat.data)

    def launch(self, lr, state=None):
        ops.momentum_apply(self.flat.data, self.flat.grad, self.acc, lr, self.momentum, self.nesterov, self.gscale, state=state)
        self._after()

    def state_dict(self):
        return {"acc": self.acc}

    def load_state_dict(self, sd):
        self.acc.copy_(sd["acc"].to(self.acc.device))


def make_optimizer(name, flat, momentum=0.9):
    """Optimizer.Name switch of the reference (model.py:649-658)."""
    if name == "SGD":
        return GradientDescentOptimizer(flat)
    if name == "Adam":
        return AdamOptimizer(flat)
    if name == "Momentum":
        return MomentumOptimizer(flat, momentum)
    if name == "NesterovMomentum":
        return MomentumOptimizer(flat, momentum, use_nesterov=True)
    raise SystemExit("Invalid optimizer")
