"""Adam + global-norm clipping on the gfx950 kernels (csrc/optim.hip, include/mssr_hip.h).

Counterpart of what the reference trainer does per optimizer step (onmt/trainer_wgan_wh.py:192-211: one
``torch.optim.Adam(lr, betas=(adam_beta1, adam_beta2))`` per network; :623-628, :973-977:
``clip_grad_norm_(params, max_grad_norm)`` then ``step()``): here both are a handful of launches over ALL tensors of
the network (norm: ~5 + 1, update: ~5 for the 400-tensor main model), the clip coefficient never leaves the device, and
the weight caches of ``ops`` / the engines are told about the update (``ops.params_changed``) by the step itself.
"""
import ctypes

import torch

from . import _lib as L
from . import ops


class Adam:
    """``Adam(module_or_params, lr, betas)``; ``zero_grad()``; ``step(max_grad_norm=0)``.

    Same update as ``torch.optim.Adam`` (bias-corrected, no weight decay, no amsgrad); parameters whose ``.grad`` is
    None are skipped; ``max_grad_norm > 0`` clips by the global L2 norm of this optimizer's gradients first, scaling
    ``.grad`` in place like ``clip_grad_norm_``.  ``last_norm`` holds the (device) norm of the last step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.modules = [params] if isinstance(params, torch.nn.Module) else []
        plist = list(params.parameters()) if isinstance(params, torch.nn.Module) else list(params)
        self.params = [p for p in plist if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        for p in self.params:
            L.require_cuda(p)
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("parameters must be contiguous float32 tensors")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.steps = 0
        dev = self.params[0].device
        n = len(self.params)
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        lib = L.load()
        self._partial = torch.empty(int(lib.mssr_grad_norm_scratch_floats(n)), dtype=torch.float32, device=dev)
        self.last_norm = torch.zeros(2, dtype=torch.float32, device=dev)
        self._n = n
        self._pp = (ctypes.c_void_p * n)(*[p.data_ptr() for p in self.params])
        self._ptr_key = tuple(p.data_ptr() for p in self.params)
        self._gp = (ctypes.c_void_p * n)()
        self._numel = (ctypes.c_int64 * n)(*[p.numel() for p in self.params])
        self._off = (ctypes.c_int64 * n)(*offs)
        self._steps = (ctypes.c_int32 * n)()          # per parameter, as torch.optim.Adam's state["step"]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _collect(self):
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                self._gp[i] = None
                continue
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device or g.data_ptr() % 16:
                g = p.grad = g.to(device=p.device, dtype=torch.float32).clone(memory_format=torch.contiguous_format)
            self._gp[i] = g.data_ptr()
        key = tuple(p.data_ptr() for p in self.params)
        if key != self._ptr_key:                      # module.to() / load into new storage
            self._pp = (ctypes.c_void_p * self._n)(*key)
            self._ptr_key = key

    @torch.no_grad()
    def grad_norm(self, max_grad_norm=0.0):
        """Global L2 norm of the gradients -> self.last_norm[0] (clip coefficient in [1]); no host sync."""
        self._collect()
        L.check(L.load().mssr_grad_norm(self._n, self._gp, self._numel, float(max_grad_norm), self._partial.data_ptr(),
                                        self.last_norm.data_ptr(), L.stream()), "grad_norm")
        return self.last_norm[0]

    @torch.no_grad()
    def step(self, max_grad_norm=0.0):
        lib = L.load()
        self._collect()
        coef = None
        if max_grad_norm and max_grad_norm > 0:
            L.check(lib.mssr_grad_norm(self._n, self._gp, self._numel, float(max_grad_norm), self._partial.data_ptr(),
                                       self.last_norm.data_ptr(), L.stream()), "grad_norm")
            coef = self.last_norm.data_ptr()
        self.steps += 1
        for i in range(self._n):
            if self._gp[i]:
                self._steps[i] += 1
        L.check(lib.mssr_adam_step(self._n, self._pp, self._gp, self._numel, self._off, self.exp_avg.data_ptr(),
                                   self.exp_avg_sq.data_ptr(), coef, self.lr, self.betas[0], self.betas[1], self.eps,
                                   self._steps, 1 if coef else 0, L.stream()), "adam_step")
        # the kernels wrote the parameters without bumping Parameter._version: tell the weight caches
        ops.params_changed_list(self.params)
