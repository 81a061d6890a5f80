"""Kernel-layout weight forms.

The GEMM kernels (sb_linear_fwd) stage dense row-major [N, K] weights; several layers need their parameter in another
arrangement (Conv1d / ConvTranspose1d taps folded into K or N, transposes for the data gradients, the 3x3 convolutions'
tap-major zero-padded rows, a bias repeated over the taps).  Instead of permuted copies made with torch ops in every
forward and backward call (~70 tiny launches per train step in round 1), all forms of a model live in ONE arena that a
single launch (sb_wview_gather) refreshes from the LIVE parameters at the start of every forward pass (one ~5 us launch;
round 2 skipped it while torch's version counters and an optimiser epoch were unchanged -- which an in-place write through
`.data` defeats).
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
        self.specs = []         # (key, getter -> the live parameter, WView, N, K)
        self.arena = None
        self.views = {}
        self._table = None
        self._src = None
        self.max_elems = 0

    def add(self, key, getter, view, N, K):
        """getter: a callable returning the CURRENT parameter tensor (re-resolved from the module at every refresh, so a
        parameter that was replaced -- load_state_dict(assign=True), module.weight = nn.Parameter(...) -- is picked up)"""
        self.specs.append((key, getter, view, int(N), int(K)))

    def _build(self, params):
        dev = params[0].device
        offs, off = [], 0
        for _, _, _, N, K in self.specs:
            offs.append(off)
            off += (N * K + 3) // 4 * 4                       # 16-byte aligned forms
        if self.arena is None or self.arena.device != dev or self.arena.numel() != off:
            self.arena = torch.empty(off, device=dev, dtype=torch.float32)
        jobs = (L.WViewJob * len(self.specs))()
        self.views = {}
        for j, ((key, _, v, N, K), p, o) in enumerate(zip(self.specs, params, offs)):
            assert p.is_contiguous() and p.dtype == torch.float32 and p.device == dev
            jobs[j].src, jobs[j].dst = p.data_ptr(), self.arena.data_ptr() + 4 * o
            jobs[j].v, jobs[j].N, jobs[j].K = v, N, K
            self.views[key] = self.arena[o:o + N * K].view(N, K) if K > 1 else self.arena[o:o + N]
        raw = bytes(jobs)
        self._table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.max_elems = max(N * K for _, _, _, N, K in self.specs)

    def source_key(self):
        """identity + address of every source parameter right now (StreamingSeparator re-captures its graph when it changes)"""
        return tuple((id(p), p.data_ptr()) for p in (g() for _, g, _, _, _ in self.specs))

    def refresh(self):
        """-> dict key -> dense form.  ALWAYS launches the gather (one ~5 us launch per forward pass, captured into the
        hipGraph of a streaming loop like every other launch): an in-place write that torch's version counters cannot see
        (`p.data.copy_`, an EMA swap, dist.broadcast(p.data), the fused Adam kernel) must never leave the front-end /
        back-end / conv-LSTM GEMMs on stale forms while the recurrent kernels read the live parameters.  The job table is
        rebuilt when a source parameter was replaced or moved."""
        if not self.specs:
            return self.views
        params = [g() for _, g, _, _, _ in self.specs]
        src = tuple((id(p), p.data_ptr()) for p in params)
        if self.arena is None or src != self._src:                   # first use / parameters replaced or moved
            if torch.cuda.is_current_stream_capturing():
                raise L.SoundBubbleHipError("weight forms: a source parameter changed identity under stream capture")
            self._build(params)
            self._src = src
        if not (self.arena.is_cuda and self._table.is_cuda):
            raise L.SoundBubbleHipError("weight forms: parameters must live on the GPU")
        L.check(L.load().sb_wview_gather(C.c_void_p(self._table.data_ptr()), len(self.specs), self.max_elems,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "sb_wview_gather")
        return self.views
