"""FlatSGD: torch.optim.SGD's update as ONE launch of the engine (csrc/optim.hip, cl3d_sgd_step) over flat buffers.

The reference trains with torch.optim.SGD (function/train_modelnet_dist.py:137-141; momentum and weight decay from the
YAML).  On the device that optimizer is already a single multi-tensor kernel; what an eagerly launched step pays for it is
the HOST: `optimizer.step()` + `optimizer.zero_grad()` are ~0.11 ms of Python next to a 0.29 ms local-aggregation step
(profiles/r05/eager_host.txt).  (Measured, profiles/r05/eager_optimizer_ab.txt: no gain on this repository's benches -- the
in-place accumulation into the flat gradient buffer costs a small kernel per parameter, which outweighs the saved Python
where the step is already device-bound, and the host-bound eager steps vary too much from process to process to show a
difference.  Offered for loops with many parameters; `bench.py` keeps torch.optim.SGD.)  FlatSGD moves every parameter of a group into one flat fp32 buffer (the parameter
tensors become views of it, as `dp.FlatGradients` does for the gradients, which it also owns), so a step is one C-ABI
call per parameter group, and the same call zeroes the gradients for the next accumulation:

    opt = FlatSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-3)
    loss.backward(); opt.step()            # no zero_grad() needed (it is a no-op kept for loop compatibility)
    opt.flat_grads                         # one buffer per group: what a data-parallel step all-reduces

Same arithmetic per element as torch.optim.SGD (tests/test_optim_gpu.py holds it to the library optimizer over several
steps, with momentum, dampening, Nesterov and weight decay).  Only fp32 parameters on one device; a parameter that received
no gradient in a step still holds a zero gradient here (torch.optim.SGD would skip it: with weight decay or momentum the two
differ for such parameters -- the networks of this repository give every parameter a gradient in every step).
"""
import ctypes

import torch

from . import _lib


class FlatSGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                      nesterov=nesterov))
        self.flat_params, self.flat_grads, self._bufs, self._steps = [], [], [], []
        self._recheck = False
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self.flat_params.append(None); self.flat_grads.append(None); self._bufs.append(None); self._steps.append(0)
                continue
            dev = ps[0].device
            for p in ps:
                if p.dtype != torch.float32 or p.device != dev:
                    raise TypeError("FlatSGD: fp32 parameters on one device per group")
            total = sum(p.numel() for p in ps)
            flat_p = torch.empty(total, dtype=torch.float32, device=dev)
            flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
            off = 0
            with torch.no_grad():
                for p in ps:
                    n = p.numel()
                    flat_p[off:off + n].copy_(p.detach().reshape(-1))
                    if p.grad is not None:
                        flat_g[off:off + n].copy_(p.grad.detach().reshape(-1))
                    p.data = flat_p[off:off + n].view(p.shape)   # the parameter IS a slice of the flat buffer from now on
                    p.grad = flat_g[off:off + n].view(p.shape)   # ... and autograd accumulates into a slice of the other
                    off += n
            self.flat_params.append(flat_p)
            self.flat_grads.append(flat_g)
            self._bufs.append(torch.zeros(total, dtype=torch.float32, device=dev) if group["momentum"] != 0 else None)
            self._steps.append(0)
            self._expose_state(len(self._bufs) - 1)

    def _expose_state(self, gi):
        """torch.optim.SGD's per-parameter state, `state[p]['momentum_buffer']`, as VIEWS of the flat momentum buffer: what
        `state_dict()` saves and the reference's checkpoints hold (function/train_modelnet_dist.py:145,160 saves and
        restores `optimizer.state_dict()`), so a resumed run continues with its momentum."""
        buf = self._bufs[gi]
        if buf is None:
            return
        off = 0
        for p in self.param_groups[gi]["params"]:
            if not p.requires_grad:
                continue
            n = p.numel()
            self.state[p]["momentum_buffer"] = buf[off:off + n].view(p.shape)
            off += n

    def load_state_dict(self, state_dict):
        """Accepts its own state dict and torch.optim.SGD's (same layout: one `momentum_buffer` per parameter).  The base
        class replaces the state tensors by copies; they are copied INTO the flat buffer and the state re-pointed at its
        views.  A group that comes back with momentum has taken its first step (torch sets buf = g there)."""
        super().load_state_dict(state_dict)
        self._recheck = True
        for gi, group in enumerate(self.param_groups):
            buf = self._bufs[gi]
            if group["momentum"] != 0 and buf is None and self.flat_params[gi] is not None:
                buf = self._bufs[gi] = torch.zeros_like(self.flat_params[gi])
            if buf is None:
                continue
            off, loaded = 0, False
            with torch.no_grad():
                for p in group["params"]:
                    if not p.requires_grad:
                        continue
                    n = p.numel()
                    mb = self.state.get(p, {}).get("momentum_buffer")
                    if mb is not None and mb.data_ptr() != buf[off:off + n].data_ptr():
                        buf[off:off + n].copy_(mb.detach().reshape(-1).to(buf.dtype))
                        loaded = True
                    off += n
            if loaded:
                self._steps[gi] = max(self._steps[gi], 1)
            self._expose_state(gi)

    def _check_views(self, gi):
        """A `module.zero_grad(set_to_none=True)`, a `.to()` or a `p.grad = ...` after construction detaches a parameter
        from the flat buffers; the flat update would then silently apply zeros.  Fail instead."""
        flat_p, flat_g = self.flat_params[gi], self.flat_grads[gi]
        p0, g0 = flat_p.data_ptr(), flat_g.data_ptr()
        off = 0
        for p in self.param_groups[gi]["params"]:
            if not p.requires_grad:
                continue
            if p.data_ptr() != p0 + 4 * off or p.grad is None or p.grad.data_ptr() != g0 + 4 * off:
                raise RuntimeError("FlatSGD: a parameter or its .grad no longer lives in the flat buffers (zero_grad("
                                   "set_to_none=True) on the module, .to() or a re-assigned .grad after the optimizer was "
                                   "built); use optimizer.zero_grad() and build the optimizer after moving the model")
            off += p.numel()

    def zero_grad(self, set_to_none=False):
        """The step itself leaves the gradients zeroed; kept so that the reference's loop runs unchanged.  (Never sets
        the gradients to None: they are views of the flat buffer.)"""
        return None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            flat_p = self.flat_params[gi]
            if flat_p is None:
                continue
            flat_g, buf = self.flat_grads[gi], self._bufs[gi]
            if self._steps[gi] % 64 == 0 or self._recheck:  # (a loop that detaches them does so in its first iteration)
                self._check_views(gi)
            # torch's first step sets buf = g; from a zeroed buffer that is what momentum * buf + (1 - dampening) * g gives
            # unless dampening != 0 -- only then does the kernel need to be told, and only then may the first step not be
            # frozen into a captured graph (the flag would be replayed)
            first = 1 if (self._steps[gi] == 0 and group["dampening"] != 0 and buf is not None) else 0
            if first and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FlatSGD with dampening: run one step eagerly before capturing it in a graph")
            with _lib.on_device(flat_p.device):
                _lib.check(lib.cl3d_sgd_step(
                    ctypes.c_void_p(flat_p.data_ptr()), ctypes.c_void_p(flat_g.data_ptr()),
                    ctypes.c_void_p(buf.data_ptr()) if buf is not None else None, flat_p.numel(), float(group["lr"]),
                    float(group["momentum"]), float(group["dampening"]), float(group["weight_decay"]),
                    1 if group["nesterov"] else 0, first, 1, _lib.stream_ptr(flat_p.device)))
            self._steps[gi] += 1
        self._recheck = False
        return loss
