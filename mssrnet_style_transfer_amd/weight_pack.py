"""Job tables of the weight-pack kernel (csrc/pack.hip, ``mssr_pack_weights``).

An engine's pack (decode_engine._Pack, encoder_engine._Pack) allocates its derived buffers ONCE and describes how
each one follows from the fp32 master parameters as a list of jobs; the table is uploaded to the GPU once, and
``refresh()`` after every optimizer step is one kernel launch instead of ~50 torch ops per network.
"""
import ctypes
import weakref

import torch

from . import _lib as L


class _Pin:
    """Held by an autograd node between forward and backward: while one is alive the pack's buffers still belong to
    a saved graph and must not be refreshed in place."""
    __slots__ = ("__weakref__",)


class JobTable:
    def __init__(self, cd, device):
        self.cd, self.device = cd, device
        self.jobs = []
        self.keep = []              # tensors the jobs point into
        self.table = None
        self.n_blocks = 0
        self._pins = weakref.WeakSet()

    @staticmethod
    def _src(p):
        t = p.detach()
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise TypeError("weight pack: parameters must be contiguous float32 tensors on the GPU (got %s, %s)" % (
                t.dtype, t.device))
        return t

    def add(self, src, dst=None, dst_t=None, dst_g=None, gamma=None, beta=None, bias=None, colsum=None, bias_f=None,
            dst_f=None):
        """src: parameter [rows, cols] or [cols] (one row).  dst / dst_g: [rows, cols] views (row stride free);
        dst_t: [cols, rows] view; colsum / bias_f: [rows] views; dst_f: contiguous fp32 view of src's size;
        gamma / beta / bias: fp32 master vectors read in place."""
        s = self._src(src)
        rows, cols = (1, s.shape[0]) if s.dim() == 1 else (s.shape[0], s.shape[1])
        j = L.PackJob()
        j.src, j.rows, j.cols = s.data_ptr(), rows, cols

        def mat(t, r, c, name):
            if t is None:
                return None, 0
            if t.dtype != self.cd or tuple(t.shape) != (r, c) or (c > 1 and t.stride(1) != 1):
                raise ValueError("weight pack: %s is %s %s, expected [%d, %d] %s" % (name, tuple(t.shape), t.dtype, r, c,
                                                                                     self.cd))
            return t.data_ptr(), t.stride(0)

        def vec(t, n, name):
            if t is None:
                return None
            if t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous():
                raise ValueError("weight pack: %s must be a contiguous float32 vector of %d" % (name, n))
            return t.data_ptr()

        j.dst, j.dst_ld = mat(dst, rows, cols, "dst")
        j.dst_t, j.dst_t_ld = mat(dst_t, cols, rows, "dst_t")
        j.dst_g, j.dst_g_ld = mat(dst_g, rows, cols, "dst_g")
        j.gamma = vec(None if gamma is None else self._src(gamma), cols, "gamma")
        j.beta = vec(None if beta is None else self._src(beta), cols, "beta")
        j.bias = vec(None if bias is None else self._src(bias), rows, "bias")
        j.colsum, j.bias_f = vec(colsum, rows, "colsum"), vec(bias_f, rows, "bias_f")
        j.dst_f = vec(dst_f, rows * cols, "dst_f")
        self.jobs.append(j)
        self.keep += [t for t in (s, dst, dst_t, dst_g, gamma, beta, bias, colsum, bias_f, dst_f) if t is not None]

    def finish(self):
        lib = L.load()
        b = 0
        arr = (L.PackJob * len(self.jobs))()
        for i, j in enumerate(self.jobs):
            j.block0 = b
            b += lib.mssr_pack_blocks(j.rows)
            arr[i] = j
        self.n_blocks = b
        raw = bytearray(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr)))
        self.table = torch.frombuffer(raw, dtype=torch.uint8).to(self.device)
        self.src_ptrs = tuple(j.src for j in self.jobs)

    def run(self):
        L.check(L.load().mssr_pack_weights(self.table.data_ptr(), len(self.jobs), self.n_blocks,
                                           L.F32 if self.cd == torch.float32 else L.BF16, L.stream()), "pack_weights")

    def pin(self):
        p = _Pin()
        self._pins.add(p)
        return p

    def pinned(self):
        return len(self._pins) > 0
