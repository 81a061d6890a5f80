"""Kernel-layout weight forms.

The GEMM kernels (sb_linear_fwd) stage dense row-major [N, K] weights; several layers need their parameter in another
arrangement (Conv1d / ConvTranspose1d taps folded into K or N, transposes for the data gradients, the 3x3 convolutions'
tap-major zero-padded rows, a bias repeated over the taps).  Instead of permuted copies made with torch ops in every
forward and backward call (~70 tiny launches per train step in round 1), all forms of a model live in ONE arena that a
single launch (sb_wview_gather) refreshes from the parameters -- once per optimiser step, or never in an inference loop:
the refresh is skipped while no parameter changed (torch version counters + the epoch the fused Adam kernel bumps).
Parameters keep the reference's names, shapes and layouts; gradients are written back through the same views by the
weight-gradient reductions (sb_wgrad_args.wv).
"""
import ctypes as C

import torch

from . import _lib as L

WEIGHT_EPOCH = 0        # bumped by every in-place parameter update torch's version counters cannot see (FusedAdam.step)


def bump_weight_epoch():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


class WeightForms:
    def __init__(self):
        self.specs = []         # (key, parameter, WView, N, K)
        self.arena = None
        self.views = {}
        self._table = None
        self._ptrs = None
        self._key = None
        self.max_elems = 0

    def add(self, key, param, view, N, K):
        self.specs.append((key, param, view, int(N), int(K)))

    def _build(self):
        dev = self.specs[0][1].device
        offs, off = [], 0
        for _, _, _, N, K in self.specs:
            offs.append(off)
            off += (N * K + 3) // 4 * 4                       # 16-byte aligned forms
        self.arena = torch.empty(off, device=dev, dtype=torch.float32)
        jobs = (L.WViewJob * len(self.specs))()
        self.views = {}
        for j, ((key, p, v, N, K), o) in enumerate(zip(self.specs, offs)):
            assert p.is_contiguous() and p.dtype == torch.float32 and p.device == dev
            jobs[j].src, jobs[j].dst = p.data_ptr(), self.arena.data_ptr() + 4 * o
            jobs[j].v, jobs[j].N, jobs[j].K = v, N, K
            self.views[key] = self.arena[o:o + N * K].view(N, K) if K > 1 else self.arena[o:o + N]
        raw = bytes(jobs)
        self._table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self._ptrs = tuple(p.data_ptr() for _, p, _, _, _ in self.specs)
        self.max_elems = max(N * K for _, _, _, N, K in self.specs)
        self._key = None

    def refresh(self, force=False):
        """-> dict key -> dense form.  Launches the gather only when a source parameter may have changed (always with
        `force`: a training forward refreshes unconditionally -- one 5 us launch -- so that an in-place update that
        bypasses torch's version counters, e.g. through `.data`, can never leave stale forms behind)."""
        if not self.specs:
            return self.views
        ptrs = tuple(p.data_ptr() for _, p, _, _, _ in self.specs)
        if self.arena is None or ptrs != self._ptrs:                 # first use / parameters moved (FlatBucket, .to())
            self._build()
        key = (WEIGHT_EPOCH, tuple(p._version for _, p, _, _, _ in self.specs))
        if force or key != self._key:
            if not (self.arena.is_cuda and self._table.is_cuda):
                raise L.SoundBubbleHipError("weight forms: parameters must live on the GPU")
            L.check(L.load().sb_wview_gather(C.c_void_p(self._table.data_ptr()), len(self.specs), self.max_elems,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "sb_wview_gather")
            self._key = key
        return self.views
