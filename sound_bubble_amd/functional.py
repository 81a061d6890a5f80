"""Stage-level autograd functions of the Sound-Bubble hot path on MI355X.

One torch.autograd.Function per fused stage; forward and backward are sequences
of HIP launches through the C ABI (sound_bubble_amd.ops).  Activations are
channels-last [B, T, F, C] fp32 throughout (the `optim` layout of the reference,
optim/tfgridnet_causal.py:398), so both the F-walk and the T-walk read contiguous
C-vectors.  Tiny weight re-layouts (a few KB) are done with torch tensor ops.

Reference lines each stage stands in for are cited in the docstrings.
"""
import torch

from . import _lib as L
from . import ops
from .ops import Geom, H, dense

GRAD_MODE = True     # set by the Net wrapper from torch.is_grad_enabled() (Function.forward runs with grad off)
DIRECT_GRADS = True  # accumulate parameter gradients straight into pre-allocated .grad buffers (see _gt)


def direct_grad_target(p):
    """p's flat-bucket gradient buffer when the HIP reductions may add into it directly, else None: only buffers
    train.FlatBucket set up (it tags the parameter) -- an ordinary .grad left over from autograd or a frozen parameter go
    through autograd as usual"""
    g = p.grad
    if (DIRECT_GRADS and g is not None and getattr(p, "_sb_flat_grad", False) and p.requires_grad
            and g.dtype == torch.float32 and g.shape == p.shape and g.is_contiguous() and g.device == p.device):
        return g
    return None


class _GradTargets:
    """Where the HIP reductions put d(loss)/d(param).  All weight-gradient kernels ACCUMULATE (atomicAdd) into their
    output, so when a parameter already owns a .grad buffer (train.FlatBucket points every .grad into one flat,
    once-per-step zeroed buffer) they add into it directly and autograd is handed None: no zero-fill and no
    AccumulateGrad add per parameter (~180 tiny launches per step).  Without a .grad buffer a fresh zero tensor is
    used and returned to autograd as usual."""

    def __init__(self):
        self.ret = {}

    def __call__(self, name, p):
        g = direct_grad_target(p)
        if g is not None:
            self.ret[name] = None
            return g
        z = torch.zeros_like(p, dtype=torch.float32)
        self.ret[name] = z
        return z

    def __getitem__(self, name):
        return self.ret[name]

    def all_direct(self):
        """every target handed out so far is a flat-bucket buffer (nothing goes back through autograd: nobody reads the sums
        before the end of the backward pass, so their reductions may ride on the side stream -- ops.defer_small_launches)"""
        return all(v is None for v in self.ret.values())
# Inference workspaces (set by the Net wrapper per forward, None in training): the zero-bordered staging tensors of the
# front end / back end (zp, yp, the spectrum rows) are kept per model and shape, zeroed ONCE -- their borders and padding
# columns are never written afterwards, the interiors are rewritten by every call -- so a forward launches no fill kernels
# for them (4 of the streaming chunk step's graph nodes).  Training allocates fresh tensors (they are saved for backward).
# One forward per model at a time: two streams running the SAME module concurrently would share these.
WORKSPACE = None
INFER_WORKSPACE = __import__("os").environ.get("SB_NO_INFER_WORKSPACE", "0") != "1"


class Workspaces:
    """per-model cache of the zero-bordered inference staging tensors, keyed by (tag, shape, device).  At most `cap` live
    entries, evicted ONE at a time in least-recently-used order -- never an entry a captured hipGraph refers to: a graph
    replays against the addresses it was captured with, so StreamingSeparator pins the keys its capture touched (`record` /
    `pin` / `unpin`) and a pinned tensor stays allocated, with its zero borders intact, for as long as that graph lives."""

    def __init__(self, cap=12):
        from collections import OrderedDict
        self.t = OrderedDict()
        self.pins = {}
        self.cap = cap
        self.touched = None                # a list while a capture records the keys it uses

    def get(self, key, make):
        t = self.t.get(key)
        if t is None:
            if len(self.t) >= self.cap:
                for k in list(self.t):                     # oldest first
                    if len(self.t) < self.cap:
                        break
                    if not self.pins.get(k):
                        del self.t[k]
            t = self.t[key] = make()
        else:
            self.t.move_to_end(key)
        if self.touched is not None:
            self.touched.append(key)
        return t

    def record(self):
        self.touched = []

    def pin_recorded(self):
        keys, self.touched = list(dict.fromkeys(self.touched or [])), None
        for k in keys:
            self.pins[k] = self.pins.get(k, 0) + 1
        return keys

    def unpin(self, keys):
        for k in keys or []:
            n = self.pins.get(k, 0) - 1
            if n > 0:
                self.pins[k] = n
            else:
                self.pins.pop(k, None)

    def __len__(self):
        return len(self.t)


def _ws_zeros(tag, shape, device):
    ws = WORKSPACE
    n = 1
    for d in shape:
        n *= d
    if ws is None or GRAD_MODE or n > (1 << 22):      # small shapes only (chunk / short-clip inference): there a fill launch
        return None                                   # costs as much as the kernel it feeds; big batches keep fresh tensors
    return ws.get((tag, tuple(shape), str(device)), lambda: torch.zeros(*shape, device=device, dtype=torch.float32))


ZC = 32          # padded channel count of the front-end feature tensor
NSPEC = 304      # 290 STFT bins (re/im) padded to a multiple of 16


class IntraPlainFn(torch.autograd.Function):
    """y = x + Linear_{2H->C}(biLSTM_F(LN_C(x)))   -- dis_embd3/tfgridnet_causal.py:795,818-827;
    optim/tfgridnet_causal.py:699-707."""

    @staticmethod
    def forward(ctx, x, ln_g, ln_b, wif, whf, bif, bhf, wir, whr, bir, bhr, lin_w, lin_b, defer_sum=False, ovl=None,
                next_ln_g=None, next_ln_b=None):
        """defer_sum: return the two per-direction partial products `part` [B, T, F, 2, C] instead of
        y = x + part[..., 0, :] + part[..., 1, :]; the InterFn that follows forms the sum in its loader (and owns the
        residual's forward); the gradient that comes back for `part` is the gradient of that sum, broadcast.
        ovl (ops.FwdOverlap): x is being produced by the preceding InterFn's kernel right now -- this pass starts behind it
        on the idle CUs (overlapped forward).
        next_ln_g / next_ln_b (with defer_sum): the LayerNorm parameters of the InterFn that follows -- not used in the forward;
        when the backward is overlapped across the two passes (ops.CrossBwd) this node's kernel runs that LayerNorm's backward
        and returns its parameter gradients."""
        B, T, F, Cc = x.shape
        ctx.next_ln = (next_ln_g, next_ln_b)
        ctx.film_of = ops.FILM_OF.pop(x.data_ptr(), None) if (ops.FILM_OF and GRAD_MODE) else None
        x = x.contiguous()
        P = B * T * F
        train = GRAD_MODE and any(ctx.needs_input_grad)
        geom = Geom.intra(B * T, F)
        dirs = [(wif, whf, bif, bhf), (wir, whr, bir, bhr)]
        ctx.bptt = ops.layer_mode("intra-plain", Cc)      # BPTT-state precision this node records (and backpropagates) under
        with ops.bptt_mode(ctx.bptt):
            return IntraPlainFn._forward(ctx, x, ln_g, ln_b, dirs, lin_w, lin_b, defer_sum, ovl, geom, train)

    @staticmethod
    def _forward(ctx, x, ln_g, ln_b, dirs, lin_w, lin_b, defer_sum, ovl, geom, train):
        B, T, F, Cc = x.shape
        P = B * T * F
        (wif, whf, bif, bhf), (wir, whr, bir, bhr) = dirs
        if ops.intra_lin_fusion_ok(train, Cc, B * T):
            # the Linear inside the recurrence: two per-direction partial products + one elementwise pass; hs is then
            # only a (fp16) side output for the backward kernels, and none at all in inference
            part = torch.empty(P, 2, Cc, device=x.device, dtype=torch.float32)
            hs, _, gates, u = ops.lstm_fwd(x.view(P, Cc), ln_g, ln_b, dirs, geom, save=train,
                                           lin=(lin_w.contiguous(), lin_b, part),
                                           want_hs=train and not ops.bi_hs_from_records(Cc),
                                           no_gates=train and ops.GATE_RECOMPUTE and ops.BPTT == "compact", consume=ovl)   # (compact-mode memory saver)
            y = part.view(B, T, F, 2, Cc) if defer_sum else ops.add3(x.view(P, Cc), part).view(B, T, F, Cc)
        else:
            assert not defer_sum
            hs, _, gates, u = ops.lstm_fwd(x.view(P, Cc), ln_g, ln_b, dirs, geom, save=train)
            y = torch.empty_like(x)
            g, s_in = dense(P, 2 * H)
            _, s_out = dense(P, Cc)
            ops.linear(hs, lin_w, lin_b, y, g, s_in, s_out, 2 * H, Cc, epi=L.EPI_RES, res=x)
        if train:
            ctx.save_for_backward(x, ln_g, wif, whf, wir, whr, lin_w, hs, u, ln_b, bif, bhf, bir, bhr, lin_b,
                                  *[t for t in gates if t is not None])
            ctx.dims = (B, T, F, Cc)
            ctx.no_gates = gates[0] is None          # records without gates: the fused backward recomputes them
            ctx.defer_sum = bool(defer_sum)
        return y

    @staticmethod
    def backward(ctx, dy):
        with ops.bptt_mode(ctx.bptt):
            return IntraPlainFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        x, ln_g, wif, whf, wir, whr, lin_w, hs, u, ln_b, bif, bhf, bir, bhr, lin_b, *g_ = ctx.saved_tensors
        gates = (None, g_[0]) if ctx.no_gates else (g_[0], g_[1] if len(g_) > 1 else None)
        gt = _GradTargets()
        B, T, F, Cc = ctx.dims
        P = B * T * F
        if ctx.defer_sum:                      # [B, T, F, 2, C], both halves equal (a stride-0 broadcast): take one
            dy = dy[..., 0, :]
        pend = ops.CROSS_PENDING.pop(dy.data_ptr(), None) if ops.CROSS_PENDING else None
        if pend is not None:                   # the gradients of the following InterFn's LayerNorm parameters are formed HERE
            nlg, nlb = ctx.next_ln
            pend.d_ln_g, pend.d_ln_b = gt("next_ln_g", nlg), gt("next_ln_b", nlb)
        else:
            gt.ret["next_ln_g"] = gt.ret["next_ln_b"] = None
        if pend is not None and not dy.is_contiguous():
            pend.materialize()
            pend = None
        dy = dy.contiguous()
        geom = Geom.intra(B * T, F)
        gP, sC = dense(P, Cc)
        _, s2H = dense(P, 2 * H)
        # Linear backward (its data gradient d(hs) = dy . W_lin is formed inside the recurrent kernel when possible)
        fuse = ops.can_fuse_linear_bwd()
        dhs = None
        if not fuse:
            dhs = torch.empty(P, 2 * H, device=dy.device, dtype=torch.float32)
            ops.linear(dy, lin_w.t().contiguous(), None, dhs, gP, sC, s2H, Cc, 2 * H)
        # BPTT
        tg = [(gt("wif", wif), gt("whf", whf), gt("bif", bif), gt("bhf", bhf)),
              (gt("wir", wir), gt("whr", whr), gt("bir", bir), gt("bhr", bhr))]
        if pend is not None and not (fuse and Cc == 32 and hs is None and gates[0] is not None and ops.can_fuse_stream_bi(u, hs)):
            pend.materialize()                 # (the overlapped consumer is not available for this node after all)
            pend = None
        if fuse and Cc == 32 and ops.can_fuse_stream_bi(u, hs):
            # recurrence + streaming part + the Linear's weight gradient in one launch (dgates stay in LDS)
            du = None
            if pend is not None:               # ... started next to the inter-frame backward that is producing its input (dy) right now
                lt = (gt("lin_w", lin_w), gt("lin_b", lin_b))
                du = ops.lstm_bwd_fused_bi([whf, whr], gates, geom, u, hs, [wif, wir], tg, dy=dy.view(P, Cc), w_lin=lin_w,
                                           lin_targets=lt, consume=pend, defer_ok=gt.all_direct())
                if du is None:
                    pend.materialize()
            if du is None:
                du = ops.lstm_bwd_fused_bi([whf, whr], gates, geom, u, hs, [wif, wir], tg, dy=dy.view(P, Cc), w_lin=lin_w,
                                           lin_targets=(gt("lin_w", lin_w), gt("lin_b", lin_b)),
                                           biases=[(bif, bhf), (bir, bhr)])
        else:
            ops.wgrad(dy, Cc, Cc, hs, s2H, gP, 2 * H, gt("lin_w", lin_w), dbias=gt("lin_b", lin_b))
            dg = ops.lstm_bwd_rec([whf, whr], gates, dhs, geom, dy=dy.view(P, Cc) if fuse else None,
                                  w_lin=lin_w if fuse else None)
            # one pass over dgates: weight/bias gradients + dU; then LayerNorm backward (+ residual)
            _, du = ops.lstm_bwd_stream(dg, u, hs, [wif, wir], 1, F, 1, targets=tg)
        fo = getattr(ctx, "film_of", None)
        if fo is not None and Cc == 32 and du.numel() == 2 * P * Cc and ops.LN_FILM_FUSION:
            # ... with the FiLM backward of the block in front (whose inter-frame epilogue applied it to our input) in the same
            # pass: dx never reaches memory; that block's backward finds the gradient marked and skips its own FiLM kernel
            f_w, y_pre, bank, k = fo
            if bank.get("G") is None:
                bank["G"] = torch.zeros(bank["n"], 2, *f_w.shape, device=f_w.device, dtype=torch.float32)
            lg, lb = gt("ln_g", ln_g), gt("ln_b", ln_b)
            dx = ops.ln_film_bwd(du, x.view(P, Cc), ln_g, dy.view(P, Cc), y_pre, f_w, bank["G"][k, 0], bank["G"][k, 1],
                                 lg, lb, (B, T, F, Cc), defer_ok=gt["ln_g"] is None and gt["ln_b"] is None)
            ops.FILM_DONE[dx.data_ptr()] = True           # (consumed by the previous block's InterFn.backward; leftovers raise at the end)
            ops.arm_handover_check()
        else:
            dx, _, _, _ = ops.ln_bwd(du, x.view(P, Cc), ln_g, res=dy.view(P, Cc), d_g=gt("ln_g", ln_g), d_b=gt("ln_b", ln_b),
                                     hint=True)
        dx = dx.view(B, T, F, Cc)
        return (dx, gt["ln_g"], gt["ln_b"], gt["wif"], gt["whf"], gt["bif"], gt["bhf"], gt["wir"], gt["whr"], gt["bir"],
                gt["bhr"], gt["lin_w"], gt["lin_b"], None, None, gt["next_ln_g"], gt["next_ln_b"])


class InterFn(torch.autograd.Function):
    """y = x + Linear_{H->C}(LSTM_T(LN_C(x)), carried (h0,c0))  -- dis_embd3/tfgridnet_causal.py:830-849;
    optim :709-728.  Returns (y, hN, cN); the state rows are b*F+f as in the reference."""

    @staticmethod
    def forward(ctx, x, ln_g, ln_b, wi, wh, bi, bh, lin_w, lin_b, h0, c0, part=None, film_w=None, film_b=None, bank=None,
                film_k=0, ovl=None):
        """part (optional, [B, T, F, 2, C]): the deferred halves of the preceding IntraPlainFn -- the block input is then
        x + part[..., 0, :] + part[..., 1, :], summed by the kernel's loader; x is the intra-frame block's own input and
        gets NO gradient from here (the residual's gradient is applied by IntraPlainFn.backward, as before).
        film_w / film_b (optional, [B, F, C]): the FiLM planes of the NEXT block, applied to y in the kernel's epilogue
        (the returned y is post-FiLM); bank / film_k as in FilmFn.
        ovl (ops.FwdOverlap): the kernel publishes y slab by slab for the next block's IntraPlainFn (overlapped forward)."""
        B, T, F, Cc = x.shape
        x = x.contiguous()
        P = B * T * F
        train = GRAD_MODE and any(ctx.needs_input_grad)
        geom = Geom.inter(B, T, F)
        ctx.bptt = ops.layer_mode("inter", Cc) if ops.can_fuse_linear_fwd() else ("legacy" if ops.BPTT == "wide" else ops.BPTT)
        with ops.bptt_mode(ctx.bptt):
            return InterFn._forward(ctx, x, ln_g, ln_b, wi, wh, bi, bh, lin_w, lin_b, h0, c0, part, film_w, film_b, bank,
                                    film_k, ovl, geom, train)

    @staticmethod
    def _forward(ctx, x, ln_g, ln_b, wi, wh, bi, bh, lin_w, lin_b, h0, c0, part, film_w, film_b, bank, film_k, ovl, geom,
                 train):
        B, T, F, Cc = x.shape
        P = B * T * F
        h0c = h0.reshape(B * F, H).contiguous() if h0 is not None else None
        c0c = c0.reshape(B * F, H).contiguous() if c0 is not None else None
        y = torch.empty_like(x)
        fuse = ops.can_fuse_linear_fwd()            # Linear + residual applied inside the recurrent kernel
        x_sum = None
        if part is not None:
            assert fuse and Cc == 32
            x_sum = torch.empty_like(x) if train else None
        film = None
        if film_w is not None:
            assert fuse
            film = (film_w.contiguous(), film_b.contiguous(), torch.empty_like(x) if train else None)
        # wide mode, C = 32, a geometry whose backward runs as the recurrence + stream-kernel pair: no gate records -- the
        # backward recurrence recomputes them from the u / hs pairs (ops.inter_gate_recompute_ok)
        no_gates = bool(train and fuse and ops.inter_gate_recompute_ok(geom, Cc, x.device))
        # ... and no hs where the backward will be the role-split fused kernel that recomputes h from the records (round 4: the
        # cross-pass schedule's producer): 256 of the 2 048 bytes this store-bound pass writes per position
        skip_hs = bool(train and fuse and part is not None and not no_gates and ops.inter_hs_from_records_ok(geom, Cc, x.device))
        hs, (hN, cN), gates, u = ops.lstm_fwd(x.view(P, Cc), ln_g, ln_b, [(wi, wh, bi, bh)], geom, h0=h0c, c0=c0c,
                                              save=train, want_state=True, no_gates=no_gates,
                                              lin=(lin_w.contiguous(), lin_b, y) if fuse else None,
                                              want_hs=(train and not skip_hs) or not fuse,
                                              x_part=part.contiguous() if part is not None else None, x_sum=x_sum,
                                              film=film, produce=ovl if fuse else None)
        if part is not None and train:
            x = x_sum                               # the block's real input: what the backward's LayerNorm needs
        if not fuse:
            g, s_in = dense(P, H)
            _, s_out = dense(P, Cc)
            ops.linear(hs, lin_w, lin_b, y, g, s_in, s_out, H, Cc, epi=L.EPI_RES, res=x)
        if train:
            ctx.save_for_backward(x, ln_g, wi, wh, lin_w, hs, u, ln_b, bi, bh, lin_b, *[t for t in gates if t is not None])
            ctx.dims = (B, T, F, Cc)
            ctx.no_gates = gates[0] is None       # (then the one record tensor saved is c_prev)
            ctx.h0 = h0c if ctx.no_gates else None
            ctx.deferred = part is not None
            ctx.film = (film[0], film[2], bank, film_k) if film is not None else None
        if train and film is not None and bank is not None and ops.LN_FILM_FUSION and Cc == 32:
            # the IntraPlainFn that takes y as its input may run this FiLM's backward together with its own LayerNorm backward
            ops.FILM_OF.clear()
            ops.FILM_OF[y.data_ptr()] = (film[0], film[2], bank, film_k)
        hN, cN = hN.view(1, B * F, H), cN.view(1, B * F, H)
        ctx.set_materialize_grads(False)     # no zero tensors for the state outputs' (absent) gradients
        ctx.mark_non_differentiable(hN, cN)
        return y, hN, cN

    @staticmethod
    def backward(ctx, dy, _dh, _dc):
        with ops.bptt_mode(ctx.bptt):
            return InterFn._backward(ctx, dy, _dh, _dc)

    @staticmethod
    def _backward(ctx, dy, _dh, _dc):
        x, ln_g, wi, wh, lin_w, hs, u, ln_b, bi, bh, lin_b, *g_ = ctx.saved_tensors
        gates = (None, g_[0]) if getattr(ctx, "no_gates", False) else (g_[0], g_[1] if len(g_) > 1 else None)
        gt = _GradTargets()
        B, T, F, Cc = ctx.dims
        P = B * T * F
        dy = dy.contiguous()
        geom = Geom.inter(B, T, F)
        d_fw = d_fb = None
        if ctx.film is not None:               # FiLM applied in the forward kernel's epilogue: its backward comes first
            f_w, y_pre, bank, k = ctx.film
            out = None
            if bank is not None:
                if bank.get("G") is None:
                    bank["G"] = torch.zeros(bank["n"], 2, *f_w.shape, device=f_w.device, dtype=torch.float32)
                out = (bank["G"][k, 0], bank["G"][k, 1])
            if ops.FILM_DONE.pop(dy.data_ptr(), None) is not None:
                d_fw, d_fb = out               # (dy has the FiLM factor in it: the intra-frame backward's fused pass applied it)
            else:
                dy, d_fw, d_fb = ops.film_bwd(y_pre, f_w, dy, out=out)

        def ret(dx):
            # deferred sum: the gradient of x + part0 + part1 goes to `part` (broadcast over the two halves, no copy) and
            # NOT to x -- IntraPlainFn.backward applies the residual's share itself
            if ctx.deferred:
                dpart = dx.view(B, T, F, 1, Cc).expand(B, T, F, 2, Cc)
                return (None, gt["ln_g"], gt["ln_b"], gt["wi"], gt["wh"], gt["bi"], gt["bh"], gt["lin_w"], gt["lin_b"],
                        None, None, dpart, d_fw, d_fb, None, None, None)
            return (dx, gt["ln_g"], gt["ln_b"], gt["wi"], gt["wh"], gt["bi"], gt["bh"], gt["lin_w"], gt["lin_b"],
                    None, None, None, d_fw, d_fb, None, None, None)

        gP, sC = dense(P, Cc)
        _, sH = dense(P, H)
        fuse = ops.can_fuse_linear_bwd()
        dhs = None
        if not fuse:
            dhs = torch.empty(P, H, device=dy.device, dtype=torch.float32)
            ops.linear(dy, lin_w.t().contiguous(), None, dhs, gP, sC, sH, Cc, H)
        # previous hidden state of (b,t,f) is hs[(b,t-1,f)] = position p - F; rows with t == 0 see h0 (zero in training)
        tg = [(gt("wi", wi), gt("wh", wh), gt("bi", bi), gt("bh", bh))]
        if gates[0] is None:
            # no gate records (wide gate recomputation): the recurrence + stream-kernel pair is the one backward that can run --
            # overlapped when the side stream is there, in plain order otherwise
            assert fuse and ops.BPTT == "wide"
            args = (wh, gates, geom, dy.view(P, Cc), lin_w, u, hs, wi, tg[0], (gt("lin_w", lin_w), gt("lin_b", lin_b)),
                    (x.view(P, Cc), ln_g, gt("ln_g", ln_g), gt("ln_b", ln_b)))
            dx = None
            if ops.can_overlap_inter_bwd(geom, u, hs):
                dx = ops.lstm_bwd_inter_overlapped(*args, recompute=(bi, bh, ctx.h0))
            if dx is None:
                dx = ops.lstm_bwd_inter_overlapped(*args, recompute=(bi, bh, ctx.h0), serial=True)
            return ret(dx.view(B, T, F, Cc))
        du = None
        if fuse and ops.BPTT == "wide" and hs is None and not ops.can_cross_overlap_bwd(geom, Cc, u, hs):
            # hs was not stored (the forward counted on the cross-pass schedule) and the side stream has been lost since: the same
            # fused role-split kernel in plain order (it recomputes h from the records), then the LayerNorm backward below
            du = ops.lstm_bwd_fused(wh, gates, geom, dy.view(P, Cc), lin_w, u, None, wi, tg[0],
                                    lin_targets=(gt("lin_w", lin_w), gt("lin_b", lin_b))).view(P, 1, Cc)
        elif (fuse and ops.BPTT == "wide" and ctx.deferred and gates[0] is not None
                and ops.can_cross_overlap_bwd(geom, Cc, u, hs)):
            # Backward overlapped across the two passes of the block: this pass as ONE fused role-split launch on its 145 CUs,
            # publishing du slab by slab; the intra-frame backward of the same block (the next autograd node: `deferred` says its
            # forward handed us the two halves) starts on the idle CUs and forms dx = LN-backward(du) + dy itself, tile by tile.
            # What goes back through autograd is the dx BUFFER, filled by that kernel (ops.CROSS_PENDING carries the rest).
            slab = ops.BWD_CROSS_SLAB
            # (4 + producer tiles + 16 item counters + three words per consumer tile of 16 frames: sb_lstm_bwd_cross_produce zeroes them)
            nfl = (geom.nseq + 15) // 16 + 4 + 16 + 3 * ((B * T + 15) // 16) + 24
            flags = ops.zeroed_flags(nfl, dy.device)   # (one fill per step for all blocks; None: the library zeroes them itself)
            prezeroed = flags is not None
            if flags is None:
                flags = ops.flag_words(nfl, dy.device)
            du, overlapped, keep = ops.lstm_bwd_fused(wh, gates, geom, dy.view(P, Cc), lin_w, u, hs, wi, tg[0],
                                                      lin_targets=(gt("lin_w", lin_w), gt("lin_b", lin_b)),
                                                      produce=(flags, slab, prezeroed))
            dx = torch.empty(P, Cc, device=dy.device, dtype=torch.float32)
            order, need = ops._cross_order(B, T, F, slab, dy.device)
            pend = ops.CrossBwd(flags, slab, (geom.nseq + 15) // 16, order, need, du, x.view(P, Cc), dy.view(P, Cc), ln_g,
                                None, None, dx, keep + [flags, du, x, dy, ln_g])
            if overlapped:                     # (this LayerNorm's parameter gradients come out of the consumer node: IntraPlainFn)
                ops.CROSS_PENDING[dx.data_ptr()] = pend
                ops.arm_handover_check()
                gt.ret["ln_g"] = gt.ret["ln_b"] = None
            else:
                pend.d_ln_g, pend.d_ln_b = gt("ln_g", ln_g), gt("ln_b", ln_b)
                pend.materialize()
            return ret(dx.view(B, T, F, Cc))
        elif fuse and ops.BPTT == "wide" and ops.can_overlap_inter_bwd(geom, u, hs):
            # wide form with fewer tiles than CUs: recurrence || stream kernel (two-term dgates through L2) instead of the fused
            # single launch, which would leave the idle CUs idle
            dx = ops.lstm_bwd_inter_overlapped(wh, gates, geom, dy.view(P, Cc), lin_w, u, hs, wi, tg[0],
                                               (gt("lin_w", lin_w), gt("lin_b", lin_b)),
                                               (x.view(P, Cc), ln_g, gt("ln_g", ln_g), gt("ln_b", ln_b)))
            if dx is not None:
                return ret(dx.view(B, T, F, Cc))
        if du is not None:                     # (hs-free fallback above)
            pass
        elif fuse and ops.can_fuse_stream(u, hs, geom):
            # recurrence + streaming part + the Linear's weight gradient in one launch (where it pays)
            ln = (x.view(P, Cc), ln_g, gt("ln_g", ln_g), gt("ln_b", ln_b)) if (Cc == 16 and ops.FUSED_LN_BWD) else None
            du = ops.lstm_bwd_fused(wh, gates, geom, dy.view(P, Cc), lin_w, u, hs, wi, tg[0],
                                    lin_targets=(gt("lin_w", lin_w), gt("lin_b", lin_b)), ln=ln).view(P, 1, Cc)
            if ln is not None:                         # ... and the LayerNorm backward + residual: du is dx already
                dx = du.view(B, T, F, Cc)
                return ret(dx)
        else:
            lin_t = (gt("lin_w", lin_w), gt("lin_b", lin_b))
            if fuse and ops.can_overlap_inter_bwd(geom, u, hs):
                # fewer tiles than CUs: the stream kernel (with its LayerNorm / Linear riders) starts on the idle CUs while
                # the recurrence runs
                dx = ops.lstm_bwd_inter_overlapped(wh, gates, geom, dy.view(P, Cc), lin_w, u, hs, wi, tg[0], lin_t,
                                                   (x.view(P, Cc), ln_g, gt("ln_g", ln_g), gt("ln_b", ln_b)))
                if dx is not None:             # None: the side stream was lost since the check -- plain order below
                    return ret(dx.view(B, T, F, Cc))
            dg = ops.lstm_bwd_rec([wh], gates, dhs, geom, dy=dy.view(P, Cc) if fuse else None,
                                  w_lin=lin_w if fuse else None)
            if ops.can_fuse_stream_ln(dg, u, hs):          # ... and the LayerNorm backward + residual in the same pass,
                ride = fuse and ops.STREAM_LIN_WGRAD       # ... and the Linear's weight gradient (its dy is that residual)
                if not ride:
                    ops.wgrad(dy, Cc, Cc, hs, sH, gP, H, lin_t[0], dbias=lin_t[1])
                _, dx = ops.lstm_bwd_stream(dg, u, hs, [wi], F, T * F, F, targets=tg,
                                            ln=(x.view(P, Cc), ln_g, dy.view(P, Cc), gt("ln_g", ln_g), gt("ln_b", ln_b)),
                                            lin_targets=lin_t if ride else None)
                return ret(dx.view(B, T, F, Cc))
            ops.wgrad(dy, Cc, Cc, hs, sH, gP, H, lin_t[0], dbias=lin_t[1])
            _, du = ops.lstm_bwd_stream(dg, u, hs, [wi], F, T * F, F, targets=tg)
        dx, _, _, _ = ops.ln_bwd(du, x.view(P, Cc), ln_g, res=dy.view(P, Cc), d_g=gt("ln_g", ln_g), d_b=gt("ln_b", ln_b),
                                 hint=True)
        dx = dx.view(B, T, F, Cc)
        return ret(dx)


class IntraConvFn(torch.autograd.Function):
    """Conv-LSTM intra path: Conv1d(C->C,k=s=down) -> PReLU -> LN -> biLSTM over F/down steps ->
    ConvTranspose1d(2H->C,k=s=down) -> + x.   optim/tfgridnet_causal.py:684-697,706-707;
    dis_embd3 :800-813 (bias_tail=False: frequencies beyond down*floor(F/down) get no deconv output)."""

    @staticmethod
    def forward(ctx, x, conv_w, conv_b, act_a, ln_g, ln_b, wif, whf, bif, bhf, wir, whr, bir, bhr, dec_w, dec_b,
                down, bias_tail, wc, wd, bd, wdT, wcT):
        B, T, F, Cc = x.shape
        x = x.contiguous()
        Kd = F // down
        Fm = Kd * down
        P2 = B * T * Kd
        train = GRAD_MODE and any(ctx.needs_input_grad)
        dev = x.device
        ctx.bptt = ops.layer_mode("intra-conv", Cc)
        with ops.bptt_mode(ctx.bptt):
            return IntraConvFn._forward(ctx, x, conv_w, conv_b, act_a, ln_g, ln_b, wif, whf, bif, bhf, wir, whr, bir, bhr,
                                        dec_w, dec_b, down, bias_tail, wc, wd, bd, wdT, wcT, train)

    @staticmethod
    def _forward(ctx, x, conv_w, conv_b, act_a, ln_g, ln_b, wif, whf, bif, bhf, wir, whr, bir, bhr, dec_w, dec_b,
                 down, bias_tail, wc, wd, bd, wdT, wcT, train):
        B, T, F, Cc = x.shape
        Kd = F // down
        Fm = Kd * down
        P2 = B * T * Kd
        dev = x.device
        # kernel-layout forms of the two conv weights (forms.WeightForms, refreshed once per optimiser step):
        # wc [co][j*C + ci], wd [j*C + c][h], bd = bias repeated over the taps, wdT / wcT their transposes (backward)
        v_pre = torch.empty(P2, Cc, device=dev, dtype=torch.float32) if train else None
        a = torch.empty(P2, Cc, device=dev, dtype=torch.float32)
        grid = (B * T, 1, Kd)
        s_x = (F * Cc, 0, down * Cc)
        s_a = (Kd * Cc, 0, Cc)
        ops.linear(x, wc, conv_b, a, grid, s_x, s_a, down * Cc, Cc, epi=L.EPI_PRELU, prelu_a=act_a, aux_out=v_pre)
        geom = Geom.intra(B * T, Kd)
        dirs = [(wif, whf, bif, bhf), (wir, whr, bir, bhr)]
        hs, _, gates, u = ops.lstm_fwd(a, ln_g, ln_b, dirs, geom, save=train)
        y = torch.empty_like(x)
        s_h = (Kd * 2 * H, 0, 2 * H)
        ops.linear(hs, wd, bd, y, grid, s_h, s_x, 2 * H, down * Cc, epi=L.EPI_RES, res=x)
        if Fm < F:      # tail frequencies: residual (+ bias when ConvTranspose1d has output_padding)
            ops.tail_rows(x, dec_b if bias_tail else None, y, B * T, F, Fm, Cc)
        if train:
            ctx.save_for_backward(x, act_a, ln_g, wif, whf, wir, whr, hs, u, v_pre, ln_b, bif, bhf, bir, bhr,
                                  conv_w, conv_b, dec_w, dec_b, wdT, wcT, *[t for t in gates if t is not None])
            ctx.dims = (B, T, F, Cc, down, Kd, bool(bias_tail))
        return y

    @staticmethod
    def backward(ctx, dy):
        with ops.bptt_mode(ctx.bptt):
            return IntraConvFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        (x, act_a, ln_g, wif, whf, wir, whr, hs, u, v_pre, ln_b, bif, bhf, bir, bhr, conv_w, conv_b, dec_w, dec_b,
         wdT, wcT, *g_) = ctx.saved_tensors
        gates = (g_[0], g_[1] if len(g_) > 1 else None)
        gt = _GradTargets()
        B, T, F, Cc, down, Kd, bias_tail = ctx.dims
        Fm = Kd * down
        P2 = B * T * Kd
        dev = dy.device
        dy = dy.contiguous()
        dym = dy if Fm == F else dy[:, :, :Fm, :].contiguous()        # dense rows [P2, down*C]
        NC = down * Cc
        grid = (B * T, 1, Kd)
        gP2, sNC = dense(P2, NC)
        _, s2H = dense(P2, 2 * H)
        _, sC = dense(P2, Cc)
        # ConvTranspose1d backward
        dhs = torch.empty(P2, 2 * H, device=dev, dtype=torch.float32)
        gm_dhs = ops.zero_scalar(dev) if ops.ABSMAX_HINTS else None   # max |dhs| on the fly
        ops.linear(dym, wdT, None, dhs, gP2, sNC, s2H, NC, 2 * H, absmax_out=gm_dhs)
        # dW[n = j*C + c][k = h] and its bias sums land in the parameters' native layout: dec_w [2H, C, down] (transposed,
        # rows n -> c*down + j), dec_b [C] (rows folded mod C) -- straight into the .grad buffers when they exist
        t_dec_w, t_dec_b = gt("dec_w", dec_w), gt("dec_b", dec_b)
        ops.wgrad(dym, NC, NC, hs, s2H, gP2, 2 * H, t_dec_w, dbias=t_dec_b, transpose_out=True, perm_n=Cc, bias_mod=Cc)
        if Fm < F and bias_tail:
            for f in range(Fm, F):
                ops.colsum(dy, B * T, F * Cc, Cc, t_dec_b, g_off=f * Cc)
        # BPTT
        geom = Geom.intra(B * T, Kd)
        tg = [(gt("wif", wif), gt("whf", whf), gt("bif", bif), gt("bhf", bhf)),
              (gt("wir", wir), gt("whr", whr), gt("bir", bir), gt("bhr", bhr))]
        if Cc == 16 and ops.can_fuse_stream_bi(u, hs):
            du = ops.lstm_bwd_fused_bi([whf, whr], gates, geom, u, hs, [wif, wir], tg, dhs=dhs, gmax=gm_dhs)
        else:
            dg = ops.lstm_bwd_rec([whf, whr], gates, dhs, geom, gmax=gm_dhs)
            # one pass over dgates (weight grads + dU), then LayerNorm + PReLU backward -> gradient of the Conv1d output
            _, du = ops.lstm_bwd_stream(dg, u, hs, [wif, wir], 1, Kd, 1, targets=tg)
        dv, _, _, _ = ops.ln_bwd(du, v_pre, ln_g, prelu_a=act_a, d_g=gt("ln_g", ln_g), d_b=gt("ln_b", ln_b),
                                 d_a=gt("act_a", act_a))
        # Conv1d backward: dx = dy + dv . Wc ; dWc = dv^T x_rows
        dx = torch.empty_like(x)
        s_x = (F * Cc, 0, NC)
        s_v = (Kd * Cc, 0, Cc)
        # dx is the dy of the inter-frame backward of the previous block: its max |.| is measured here
        gm_dx = ops.zero_scalar(dev) if (ops.ABSMAX_HINTS and Fm == F) else None
        ops.linear(dv, wcT, None, dx, grid, s_v, s_x, Cc, NC, epi=L.EPI_RES, res=dy, absmax_out=gm_dx)
        if Fm < F:
            ops.tail_rows(dy, None, dx, B * T, F, Fm, Cc)
        if gm_dx is not None:
            ops.absmax_hint_put(dx, gm_dx)
        # dW[co][k = j*C + ci] -> conv_w [co, ci, j]
        ops.wgrad(dv, Cc, Cc, x, s_x, grid, NC, gt("conv_w", conv_w), dbias=gt("conv_b", conv_b), perm_k=Cc)
        return (dx, gt["conv_w"], gt["conv_b"], gt["act_a"], gt["ln_g"], gt["ln_b"], gt["wif"], gt["whf"], gt["bif"],
                gt["bhf"], gt["wir"], gt["whr"], gt["bir"], gt["bhr"], gt["dec_w"], gt["dec_b"], None, None,
                None, None, None, None, None)


class AttentionFn(torch.autograd.Function):
    """y = x + LN_{F*C}(PReLU(Linear(local full-band self-attention(x))))  -- tfgridnet_causal.py:856-898
    (modules :639-684, causal window :722-744).  Returns (y, new K_buf, new V_buf).  The carried K/V buffers are
    state, not differentiated (the reference trains from zero-filled buffers)."""

    @staticmethod
    def forward(ctx, x, K_buf, V_buf, wq, bq, aq, gq, eq, wk, bk, ak, gk, ek, wv, bv, av, gv, ev, wp, bp, ap_, gp, ep,
                n_head, E, Lw):
        train = GRAD_MODE and any(ctx.needs_input_grad)
        B, T, F, Cc = x.shape
        x = x.contiguous()
        dev = x.device
        P = B * T * F
        Cv = Cc // n_head
        HE = n_head * E
        gP, sC = dense(P, Cc)

        # zero-padded copies of the two narrow projections' weights: ONE fill for both (the padded rows of a projection whose
        # width is a multiple of 16 -- V -- are the parameter itself)
        hpad = (HE + 15) // 16 * 16
        arena = torch.zeros(2, hpad * Cc + hpad, device=dev, dtype=torch.float32) if hpad > HE else None

        def proj(w, b, n_out, slot):
            """pre-activation of Linear(C -> n_out), stored with a 16-padded row stride (padding columns zero)"""
            npad = (n_out + 15) // 16 * 16
            if npad == n_out:
                wpad, bpad = w.contiguous(), b.contiguous()
            else:
                wpad, bpad = arena[slot, : npad * Cc].view(npad, Cc), arena[slot, npad * Cc:]
                wpad[:n_out] = w
                bpad[:n_out] = b
            out = torch.empty(P, npad, device=dev, dtype=torch.float32)       # (the GEMM writes columns < n_out of every row)
            if npad > n_out:
                out[:, n_out:].zero_()
            ops.linear(x, wpad, bpad, out, gP, sC, (0, 0, npad), Cc, npad, n_valid=n_out)
            return out, npad

        ldk = (F * E + 15) // 16 * 16
        ldv = (F * Cv + 15) // 16 * 16
        rows = Lw - 1 + T
        BH = B * n_head
        Qn = torch.empty(BH, T, ldk, device=dev, dtype=torch.float32)
        # (rows Lw - 1 .. are written whole -- padding columns included -- by sb_head_ln below: only the carried rows' padding
        #  needs zeroing, not 216 MB of V window per block)
        Kc = torch.empty(BH, rows, ldk, device=dev, dtype=torch.float32)
        Vc = torch.empty(BH, rows, ldv, device=dev, dtype=torch.float32)
        Kc[:, : Lw - 1, : F * E] = K_buf
        Vc[:, : Lw - 1, : F * Cv] = V_buf
        if ldk > F * E:
            Kc[:, : Lw - 1, F * E:].zero_()
        if ldv > F * Cv:
            Vc[:, : Lw - 1, F * Cv:].zero_()
        pq, ldq = proj(wq, bq, HE, 0)
        pk, _ = proj(wk, bk, HE, 1)
        pv, ldvp = proj(wv, bv, Cc, 0)
        ops.head_ln(pq, gq, eq, Qn, B, T, F, n_head, E, T, 0, ldk, ldi=ldq, prelu_a=aq)
        ops.head_ln(pk, gk, ek, Kc, B, T, F, n_head, E, rows, Lw - 1, ldk, ldi=ldq, prelu_a=ak)
        ops.head_ln(pv, gv, ev, Vc, B, T, F, n_head, Cv, rows, Lw - 1, ldv, ldi=ldvp, prelu_a=av)
        O = torch.empty(B, T, F, Cc, device=dev, dtype=torch.float32)
        lse = torch.empty(BH, T, device=dev, dtype=torch.float32) if train else None
        scale = 1.0 / float(F * E) ** 0.5
        ops.attn_core(Qn, Kc, Vc, O, BH, n_head, T, F, Cv, Lw, ldk, ldv, scale, lse=lse)
        # merge heads -> Linear -> PReLU -> LayerNorm(F*C) -> + x
        pp = torch.empty(P, Cc, device=dev, dtype=torch.float32)
        ops.linear(O, wp, bp, pp, gP, sC, sC, Cc, Cc)
        y = torch.empty_like(x)
        ops.head_ln(pp, gp, ep, y, B, T, F, 1, Cc, T, 0, F * Cc, res=x, prelu_a=ap_)
        nK = Kc[:, rows - (Lw - 1):, : F * E].contiguous()
        nV = Vc[:, rows - (Lw - 1):, : F * Cv].contiguous()
        ctx.set_materialize_grads(False)     # no zero tensors for the state outputs' (absent) gradients
        ctx.mark_non_differentiable(nK, nV)
        if train:
            ctx.save_for_backward(x, pq, pk, pv, Qn, Kc, Vc, lse, O, pp, wq, aq, gq, wk, ak, gk, wv, av, gv, wp, ap_, gp)
            ctx.cfg = (B, T, F, Cc, n_head, E, Lw, ldk, ldv, ldq, ldvp, scale)
        return y, nK, nV

    @staticmethod
    def backward(ctx, dy, _dK, _dV):
        x, pq, pk, pv, Qn, Kc, Vc, lse, O, pp, wq, aq, gq, wk, ak, gk, wv, av, gv, wp, ap_, gp = ctx.saved_tensors
        B, T, F, Cc, n_head, E, Lw, ldk, ldv, ldq, ldvp, scale = ctx.cfg
        dev = x.device
        P = B * T * F
        Cv = Cc // n_head
        HE = n_head * E
        BH = B * n_head
        gP, sC = dense(P, Cc)
        dy = dy.contiguous()
        # LayerNorm(F*C) + PReLU of the output projection (the residual passes dy through)
        dpp, d_gp, d_ep, d_ap = ops.head_ln_bwd(pp, gp, dy, B, T, F, 1, Cc, T, 0, F * Cc, Cc, prelu_a=ap_)
        # every zero-initialised gradient target / padded transposed weight of this backward out of ONE zero fill
        lds = (ldq, ldq, ldvp)
        zsize = Cc * Cc + Cc + sum(2 * ld * Cc + ld for ld in lds)
        zbuf = torch.zeros(zsize, device=dev, dtype=torch.float32)
        zoff = [0]

        def carve(*shape):
            n = 1
            for d_ in shape:
                n *= d_
            t = zbuf[zoff[0]: zoff[0] + n].view(*shape)
            zoff[0] += n
            return t

        d_wp = carve(Cc, Cc)
        d_bp = carve(Cc)
        ops.wgrad(dpp, Cc, Cc, O, sC, gP, Cc, d_wp, dbias=d_bp)
        dO = torch.empty(P, Cc, device=dev, dtype=torch.float32)
        ops.linear(dpp, wp.t().contiguous(), None, dO, gP, sC, sC, Cc, Cc)
        # head-major, zero-padded copy of dO (layout glue), then the attention core
        dOh = torch.empty(BH, T, ldv, device=dev, dtype=torch.float32)
        if ldv > F * Cv:
            dOh[:, :, F * Cv:].zero_()
        dOh[:, :, : F * Cv] = dO.view(B, T, F, n_head, Cv).permute(0, 3, 1, 2, 4).reshape(BH, T, F * Cv)
        dQn, dKn, dVn = ops.attn_core_bwd(Qn, Kc, Vc, dOh, lse, BH, n_head, T, F, Cv, Lw, ldk, ldv, scale)
        # per-head LayerNorms + PReLUs of the three projections
        dpq, d_gq, d_eq, d_aq = ops.head_ln_bwd(pq, gq, dQn, B, T, F, n_head, E, T, 0, ldk, ldq, prelu_a=aq)
        dpk, d_gk, d_ek, d_ak = ops.head_ln_bwd(pk, gk, dKn, B, T, F, n_head, E, T, 0, ldk, ldq, prelu_a=ak)
        dpv, d_gv, d_ev, d_av = ops.head_ln_bwd(pv, gv, dVn, B, T, F, n_head, Cv, T, 0, ldv, ldvp, prelu_a=av)
        # projections: dx = dy + sum_j dpre_j W_j ; dW_j = dpre_j^T x
        dx = torch.empty_like(x)
        outs = []
        first = True
        for dpre, w, n_out, ld in ((dpq, wq, HE, ldq), (dpk, wk, HE, ldq), (dpv, wv, Cc, ldvp)):
            wt = carve(Cc, ld)
            wt[:, :n_out] = w.t()
            if first:
                ops.linear(dpre, wt, None, dx, gP, (0, 0, ld), sC, ld, Cc, epi=L.EPI_RES, res=dy)
            else:
                ops.linear(dpre, wt, None, dx, gP, (0, 0, ld), sC, ld, Cc, accumulate=True)
            first = False
            dwp_ = carve(ld, Cc)
            dbp_ = carve(ld)
            ops.wgrad(dpre, ld, ld, x, sC, gP, Cc, dwp_, dbias=dbp_)
            outs.append((dwp_[:n_out], dbp_[:n_out]))      # (leading rows of a contiguous block: contiguous views)
        (d_wq, d_bq), (d_wk, d_bk), (d_wv, d_bv) = outs
        return (dx, None, None, d_wq, d_bq, d_aq, d_gq, d_eq, d_wk, d_bk, d_ak, d_gk, d_ek, d_wv, d_bv, d_av, d_gv,
                d_ev, d_wp, d_bp, d_ap, d_gp, d_ep, None, None, None)


class FilmFn(torch.autograd.Function):
    """y = x * w[b,f,c] + bias[b,f,c]  -- FilmLayer.forward, dis_embd3/tfgridnet_causal.py:59-68.
    bank (optional): the FilmBankFn bookkeeping dict and this layer's index k -- the gradients of w and bias are then
    accumulated into slices of ONE zeroed buffer shared by all layers (a single fill per backward pass)."""

    @staticmethod
    def forward(ctx, x, w, b, bank=None, k=0):
        x, w, b = x.contiguous(), w.contiguous(), b.contiguous()
        ctx.save_for_backward(x, w)
        ctx.bank, ctx.k = bank, k
        return ops.film_fwd(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        out = None
        if ctx.bank is not None:
            if ctx.bank.get("G") is None:
                ctx.bank["G"] = torch.zeros(ctx.bank["n"], 2, *w.shape, device=w.device, dtype=torch.float32)
            out = (ctx.bank["G"][ctx.k, 0], ctx.bank["G"][ctx.k, 1])
        dx, dw, db = ops.film_bwd(x, w, dy.contiguous(), out=out)
        return dx, dw, db, None, None


class FilmBankFn(torch.autograd.Function):
    """All FiLM scale / shift planes of a forward pass in one autograd node (dis_embd3/tfgridnet_causal.py:150-173,
    51-68, 509-513): e = LN_d(view(W_e . dis_embed, [B, F, d])), plane[k, which] = e . W[k, which]^T + b[k, which] for
    the n = n_layers - 1 FiLM layers (which = 0 scale, 1 shift).  ONE HIP launch forward (sb_film_bank_fwd), two backward
    (sb_film_bank_bwd: one workgroup per frequency bin, partial rows reduced in fixed order), the parameter gradients
    accumulated straight into their targets (the flat bucket under train.FlatBucket; fresh tensors handed to autograd
    otherwise).  Round 4 did this with rocBLAS / ATen (mm, layer_norm, baddbmm, bmm, native_layer_norm_backward: ~30 launches).
    forward(dis_embed [B, 3], W_e [dF, 3], ln_w [d], ln_b [d], bank, *[w.weight, w.bias, b.weight, b.bias] * n)
      -> 2n tensors [B, F, C]: (scale_0, shift_0, scale_1, ...), slices of one buffer."""

    @staticmethod
    def forward(ctx, dis, W_e, ln_w, ln_b, bank, *conv):
        n = len(conv) // 4
        dis = dis.float().contiguous()
        planes = ops.film_bank_fwd(dis, W_e, ln_w, ln_b, conv)
        ctx.save_for_backward(dis, W_e, ln_w, ln_b, *conv)
        ctx.bank = bank
        return tuple(planes[i] for i in range(2 * n))

    @staticmethod
    def backward(ctx, *gs):
        dis, W_e, ln_w, ln_b, *conv = ctx.saved_tensors
        n = len(conv) // 4
        B, F_, Cc = dis.shape[0], W_e.shape[0] // ln_w.shape[0], conv[0].shape[0]
        G = ctx.bank.get("G") if ctx.bank is not None else None
        nb = B * F_ * Cc * 4
        if G is None or any(g is None or g.data_ptr() != G.data_ptr() + i * nb for i, g in enumerate(gs)):
            G = torch.stack([g if g is not None else torch.zeros(B, F_, Cc, device=dis.device) for g in gs])
        if ctx.bank is not None:
            ctx.bank["G"] = None
        gt = _GradTargets()
        need = ctx.needs_input_grad
        def tgt(name, p, wanted):
            return gt(name, p) if wanted else torch.zeros_like(p)       # (a frozen parameter: the sums go nowhere)
        dW_e, dlw, dlb = tgt("W_e", W_e, need[1]), tgt("ln_w", ln_w, need[2]), tgt("ln_b", ln_b, need[3])
        d_conv = [tgt(f"c{i}", p, need[5 + i]) for i, p in enumerate(conv)]
        ops.film_bank_bwd(G.contiguous(), dis, W_e, ln_w, ln_b, conv, dW_e, dlw, dlb, d_conv)
        ret = lambda name, wanted: gt[name] if wanted else None
        return (None, ret("W_e", need[1]), ret("ln_w", need[2]), ret("ln_b", need[3]), None,
                *[ret(f"c{i}", need[5 + i]) for i in range(len(conv))])


# GEMM forms of the fixed STFT / iSTFT filter banks (module buffers, never trained): built once per (tensor, version)
_FILTER_CACHE = {}


def _cached_filter_form(filters, tag, build):
    if filters.requires_grad:
        return build(filters)
    key = (tag, filters.data_ptr(), filters._version, str(filters.device), tuple(filters.shape))
    ent = _FILTER_CACHE.get(key)
    if ent is None:
        if len(_FILTER_CACHE) > 16:
            _FILTER_CACHE.clear()
        ent = (filters, build(filters))          # the entry keeps `filters` alive, so the address stays its own
        _FILTER_CACHE[key] = ent
    return ent[1]


def _pad16(n):
    return (n + 15) // 16 * 16


def _stft_weight(filters):
    """[2F, 1, win] asteroid filter bank -> [304, win16] analysis GEMM weight (zero rows 2F..303; zero columns win..win16-1 when
    the window is not a multiple of the GEMM's 16-wide K chunks: the reference constructor's n_fft = 280)"""
    def build(flt):
        f = flt.reshape(flt.shape[0], -1)
        w = torch.zeros(NSPEC, _pad16(f.shape[1]), device=f.device, dtype=torch.float32)
        w[: f.shape[0], : f.shape[1]] = f
        return w
    return _cached_filter_form(filters, "stft", build)


class FrontEndFn(torch.autograd.Function):
    """STFT -> [re, im, ILD, IPD] features -> causal 3x3 Conv2d(27->C) -> LayerNorm(C).
    dis_embd3/tfgridnet_causal.py:475-507 (+ :72-93, :32-48, :219-231, :332-354).
    mix [B, M, Np] is already padded (net.py:70-74); conv_buf [B, 27, 2, F] is the carried
    2-frame context.  Returns x0 [B,T,F,C] channels-last and the new conv_buf."""

    @staticmethod
    def forward(ctx, mix, enc_filters, conv_w, conv_b, ln_g, ln_b, conv_buf, use_ln, hop, wk):
        B, M, Np = mix.shape
        win = enc_filters.shape[-1]
        F = enc_filters.shape[0] // 2
        Cc = conv_w.shape[0]
        nfeat = conv_w.shape[1]
        T = (Np - win) // hop + 1
        dev = mix.device
        mix = mix.contiguous()
        train = GRAD_MODE and any(ctx.needs_input_grad)
        # 1. STFT as a position-wise GEMM over overlapping rows (K = the window rounded up to whole 16-sample chunks against zero
        # filter columns: the rows are then padded so that the last frame's chunk stays inside the tensor)
        spec = torch.empty(B * M, T, NSPEC, device=dev, dtype=torch.float32)
        winp = _pad16(win)
        if winp != win:
            mix = torch.nn.functional.pad(mix, (0, winp - win))
        ops.linear(mix, _stft_weight(enc_filters), None, spec, (B * M, T, 1), (mix.shape[-1], hop, 0), (T * NSPEC, NSPEC, 0),
                   winp, NSPEC)
        # 2. features into the zero-bordered, channel-padded tensor zp [B, T+2, F+2, 32]
        zp = _ws_zeros("zp", (B, T + 2, F + 2, ZC), dev)
        if zp is None:
            zp = torch.empty(B, T + 2, F + 2, ZC, device=dev, dtype=torch.float32)
        # the two carried frames (conv_buf, channels-first) -> frame rows 0, 1 with their zero borders and padding channels; the
        # feature kernel writes rows 2 .. (borders included); the new conv_buf is the last two frame rows
        ops.stage_frames(conv_buf.contiguous(), None, zp, B, T + 2, F, nfeat, ZC)
        ops.features(spec, NSPEC, zp, B, M, T, F)
        new_buf = ops.frames_to_state(zp, B, T + 2, F, nfeat, ZC, T)
        # 3. 3x3 conv as 3 K-segments of 96 contiguous floats (+ fused LayerNorm)
        # wk: kernel-layout form of the Conv2d weight, [co][(a*3 + d)*32 + ci] with channels 27..31 zero (forms.WeightForms)
        x0 = torch.empty(B, T, F, Cc, device=dev, dtype=torch.float32)
        pre = torch.empty(B * T * F, Cc, device=dev, dtype=torch.float32) if (train and use_ln) else None
        s_in = ((T + 2) * (F + 2) * ZC, (F + 2) * ZC, ZC)
        s_out = (T * F * Cc, F * Cc, Cc)
        ops.linear(zp, wk, conv_b, x0, (B, T, F), s_in, s_out, 9 * ZC, Cc, kseg=3 * ZC, is_seg=(F + 2) * ZC,
                   epi=L.EPI_LN if use_ln else L.EPI_NONE, ln_g=ln_g if use_ln else None,
                   ln_b=ln_b if use_ln else None, aux_out=pre, f16x3=True)
        if train:
            ctx.save_for_backward(zp, pre, ln_g, conv_w, conv_b, ln_b)
            ctx.dims = (B, T, F, Cc, nfeat, bool(use_ln))
        ctx.set_materialize_grads(False)     # no zero tensors for the state outputs' (absent) gradients
        ctx.mark_non_differentiable(new_buf)
        return x0, new_buf

    @staticmethod
    def backward(ctx, dx0, _dbuf):
        zp, pre, ln_g, conv_w, conv_b, ln_b = ctx.saved_tensors
        B, T, F, Cc, nfeat, use_ln = ctx.dims
        P = B * T * F
        dx0 = dx0.contiguous()
        gt = _GradTargets()
        if use_ln:
            dpre, _, _, _ = ops.ln_bwd(dx0.view(P, 1, Cc), pre, ln_g, d_g=gt("ln_g", ln_g), d_b=gt("ln_b", ln_b),
                                       hint=True)          # max |dpre| measured on the way (the weight gradient's fp16 scale)
        else:
            dpre = dx0.view(P, Cc)
            gt.ret["ln_g"] = gt.ret["ln_b"] = None
        s_in = ((T + 2) * (F + 2) * ZC, (F + 2) * ZC, ZC)
        # dW[co][(a*3 + d)*32 + ci] lands in conv_w's own [co, ci, a, d] layout (and in the flat bucket when it exists)
        tw, tb = gt("conv_w", conv_w), gt("conv_b", conv_b)
        gm = ops.absmax_or_hint(dpre) if ops.LINEAR_F16X3 else None
        # nothing in the backward pass reads this weight gradient, and what follows on the main stream are the tiny kernels of
        # the FiLM bank's backward: with flat-bucket targets the launch goes to the side stream (joined at the end of the pass)
        side = ops.deferred_side() if (gt.all_direct() and ops.defer_small_launches((dpre, zp, tw, tb, gm))) else None
        with ops.on_stream(side):
            ops.wgrad(dpre, Cc, Cc, zp, s_in, (B, T, F), 9 * ZC, tw, kseg=3 * ZC, is_seg=(F + 2) * ZC,
                      dbias=tb, wview=L.WView.make(nfeat * 9, 9, kmod=ZC, sk_hi=1, kvalid=nfeat), f16=True, gmax=gm)
        return None, None, gt["conv_w"], gt["conv_b"], gt["ln_g"], gt["ln_b"], None, None, None, None


def _istft_weights(dec_filters):
    """asteroid synthesis bank [2F, 1, win] -> GEMM weights for interleaved (re,im) spectra rows:
    w_syn [win16, 304] (frames = spec_row . w_syn^T) and w_ana [304, win16] (its transpose, for the gradient); win16 = the
    window rounded up to 16 samples, the extra synthesis samples identically zero (n_fft = 280: a 288-sample frame whose
    last 8 samples add nothing in the overlap-add)."""
    def build(flt):
        f = flt.reshape(flt.shape[0], -1)                               # [2F, win], rows: re(0..F-1), im(F..2F-1)
        Fq = f.shape[0] // 2
        inter = torch.stack([f[:Fq], f[Fq:]], dim=1).reshape(2 * Fq, -1)    # row 2f+o
        w_ana = torch.zeros(NSPEC, _pad16(f.shape[1]), device=f.device, dtype=torch.float32)
        w_ana[: 2 * Fq, : f.shape[1]] = inter
        return w_ana.t().contiguous(), w_ana
    return _cached_filter_form(dec_filters, "istft", build)


class BackEndFn(torch.autograd.Function):
    """causal ConvTranspose2d(C->2,(3,3),padding(2,1)) -> iSTFT (conv_transpose1d overlap-add) -> crops.
    dis_embd3/tfgridnet_causal.py:517-542.  y [B,T,F,C]; returns wave [B,1,hop*T], new deconv_buf [B,C,2,F],
    new istft_buf [B,1,2F,1]."""

    @staticmethod
    def forward(ctx, y, dec_filters, dw, db, deconv_buf, istft_buf, hop, wk, bk):
        B, T, F, Cc = y.shape
        dev = y.device
        win = _pad16(dec_filters.shape[-1])      # (frames of whole 16-sample chunks: see _istft_weights)
        train = GRAD_MODE and any(ctx.needs_input_grad)
        assert dw.shape[1] == 2, "num_src=1 only (every shipped config)"
        yp = _ws_zeros("yp", (B, T + 2, F + 2, Cc), dev)
        if yp is None:
            yp = torch.empty(B, T + 2, F + 2, Cc, device=dev, dtype=torch.float32)
        # one launch: carried frames (deconv_buf) -> rows 0, 1; y -> rows 2 ..; zero frequency borders; then the new deconv_buf
        ops.stage_frames(deconv_buf.contiguous(), y.contiguous(), yp, B, T + 2, F, Cc, Cc)
        new_dbuf = ops.frames_to_state(yp, B, T + 2, F, Cc, Cc, T)
        # spectrum rows [B, T+1, 304], interleaved (re,im) per frequency; row 0 = carried frame; columns 2F .. 303 padding (zero:
        # they meet zero synthesis weights, and 0 x garbage could be NaN)
        rows = _ws_zeros("rows", (B, T + 1, NSPEC), dev)
        if rows is None:
            rows = torch.empty(B, T + 1, NSPEC, device=dev, dtype=torch.float32)
        ops.spec_rows(rows, istft_buf.reshape(B, 2, F).contiguous(), B, T, F, NSPEC, 0)
        # wk [16][(a*3 + d)*C + c] = dw[c, o, 2-a, 2-d] (rows 2..15 zero), bk [16]: kernel-layout forms (forms.WeightForms)
        s_in = ((T + 2) * (F + 2) * Cc, (F + 2) * Cc, Cc)
        ops.linear(yp, wk, bk, rows, (B, T, F), s_in, ((T + 1) * NSPEC, NSPEC, 2), 9 * Cc, 16, kseg=3 * Cc,
                   is_seg=(F + 2) * Cc, n_valid=2, out_off=NSPEC, f16x3=True)
        w_syn, w_ana = _istft_weights(dec_filters)
        frames = torch.empty(B, T + 1, win, device=dev, dtype=torch.float32)
        g, s_r = dense(B * (T + 1), NSPEC)
        _, s_f = dense(B * (T + 1), win)
        ops.linear(rows, w_syn, None, frames, g, s_r, s_f, NSPEC, win)
        wave = ops.overlap_add(frames, B, T, win, hop)
        new_ibuf = torch.empty(B, 1, 2 * F, 1, device=dev, dtype=torch.float32)
        ops.spec_rows(rows, new_ibuf, B, T, F, NSPEC, 1)                          # [re | im] of the last frame's row
        if train:
            ctx.save_for_backward(yp, dw, w_ana, db)
            ctx.dims = (B, T, F, Cc, win, hop)
        ctx.set_materialize_grads(False)     # no zero tensors for the state outputs' (absent) gradients
        ctx.mark_non_differentiable(new_dbuf, new_ibuf)
        return wave.view(B, 1, hop * T), new_dbuf, new_ibuf

    @staticmethod
    def backward(ctx, dwave, _d1, _d2):
        yp, dw, w_ana, db = ctx.saved_tensors
        B, T, F, Cc, win, hop = ctx.dims
        dev = dwave.device
        dframes = ops.overlap_add_bwd(dwave.contiguous().view(B, hop * T), B, T, win, hop)
        drows = torch.empty(B, T + 1, NSPEC, device=dev, dtype=torch.float32)
        g, s_f = dense(B * (T + 1), win)
        _, s_r = dense(B * (T + 1), NSPEC)
        ops.linear(dframes, w_ana, None, drows, g, s_f, s_r, win, NSPEC)
        dspec = drows[:, 1:, : 2 * F].contiguous()                                # [B,T,F,2]
        gt = _GradTargets()
        s_in = ((T + 2) * (F + 2) * Cc, (F + 2) * Cc, Cc)
        # dW[o][(a*3 + d)*C + c] lands in the parameter's own [c, o, 2-a, 2-d] layout
        tw, tb = gt("dw", dw), gt("db", db)
        gm = ops.absmax_or_hint(dspec) if ops.LINEAR_F16X3 else None
        # the first kernel of the backward chain (the last block's inter-frame pass) is waiting for dy, not for this weight
        # gradient: with flat-bucket targets it runs on the side stream, beside the data gradient below
        side = ops.deferred_side() if (gt.all_direct() and ops.defer_small_launches((dspec, yp, tw, tb, gm))) else None
        with ops.on_stream(side):
            ops.wgrad(dspec, 2, 2, yp, s_in, (B, T, F), 9 * Cc, tw, kseg=3 * Cc, is_seg=(F + 2) * Cc,
                      dbias=tb, wview=L.WView.make(9, 18, off=8, kmod=Cc, sk_hi=-1, nvalid=2), f16=True, gmax=gm)
        dy = ops.deconv_bwd_data(dspec, dw, B, T, F, Cc)
        return dy, None, gt["dw"], gt["db"], None, None, None, None, None


class SnrlpLossFn(torch.autograd.Function):
    """mean_b SNRLPLoss(est, gt)[b]  -- src/losses/SNRLP.py:17-42 + hl_module:321 (.mean()); mode = ops.SNR_LOSS_MODES[name]"""

    @staticmethod
    def forward(ctx, est, gt, neg_weight, mode=0):
        if est.shape != gt.shape:
            raise ValueError(f"SNRLP: estimate {tuple(est.shape)} and target {tuple(gt.shape)} differ in shape")
        B = est.shape[0]
        e = est.reshape(B, -1).contiguous()
        t = gt.reshape(B, -1).contiguous()
        # per-sample losses, their batch mean (formed by the final kernel: no reduction launch) and the moments the gradient
        # kernel needs; the gradient itself is formed in backward, scaled by the incoming gradient inside the kernel
        lv, mean, stats = ops.snrlp_loss_fwd(e, t, neg_weight, mode=mode)
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(e, t, stats)
            ctx.cfg = (est.shape, float(neg_weight), int(mode))
        ctx.set_materialize_grads(False)     # no zero tensors for the state outputs' (absent) gradients
        ctx.mark_non_differentiable(lv)
        return mean.reshape(()), lv

    @staticmethod
    def backward(ctx, gout, _glv):
        e, t, stats = ctx.saved_tensors
        shape, neg_weight, mode = ctx.cfg
        g = gout.reshape(1).to(torch.float32).contiguous()
        return ops.snrlp_loss_bwd(e, t, neg_weight, stats, g, mode=mode).view(shape), None, None, None


class MultiResoFuseLossFn(torch.autograd.Function):
    """auraloss MultiResolutionSTFTLoss (perceptual weighting, linear-magnitude L1) + l1_ratio * L1(est, gt)
    -- src/losses/MultiResoLoss.py:6-31.  cfg: the MultiResoFuseLoss module (sound_bubble_amd.losses), which holds the
    A-weighting taps and, per resolution, the windowed DFT basis restricted to the window's support as GEMM weights
    (w [Npad, K] interleaved (re, im) rows and its transpose).  The STFTs run as sb_linear_fwd GEMMs over overlapping rows
    of the reflect-padded signals; the gradient w.r.t. est is formed in the forward pass (as SnrlpLossFn does) and scaled
    by the incoming gradient in backward."""

    @staticmethod
    def forward(ctx, est, gt, cfg):
        if est.shape != gt.shape:
            raise ValueError(f"MultiResoFuseLoss: estimate {tuple(est.shape)} and target {tuple(gt.shape)} differ in shape")
        shape = est.shape
        T = shape[-1]
        e = est.reshape(-1, T).contiguous().float()
        g = gt.reshape(-1, T).contiguous().float()
        R = e.shape[0]
        dev = e.device
        want_grad = ctx.needs_input_grad[0]
        loss = torch.zeros(1, device=dev, dtype=torch.float32)
        both = torch.cat([e, g], 0)                                   # [2R, T]: one set of launches for both signals
        # log-magnitude term: its gradient weighs a bin by 1 / |X|^2, so the A-weighted signal travels as a (hi, lo) pair of fp32
        # planes and the spectrum is accumulated in double (ops.stft_f64acc); the other terms take the plain fp32 chain
        pair = bool(cfg.w_log_mag) and cfg.taps is not None
        if pair:
            xw = ops.fir_pair(both, cfg.taps).view(4 * R, T)              # rows [0, 2R) hi, [2R, 4R) lo
        else:
            xw = ops.fir(both, cfg.taps) if cfg.taps is not None else both
        dxw = torch.empty(R, T, device=dev, dtype=torch.float32) if want_grad else None
        nres = len(cfg.res)
        for i, r in enumerate(cfg.res):
            pad, K, Npad, nbins, hop, off = r["pad"], r["K"], r["Npad"], r["nbins"], r["hop"], r["off"]
            nfr = 1 + T // hop
            ldp = T + 2 * pad + 32
            xp = ops.reflect_pad(xw, pad, ldp)
            spec = torch.empty(2 * R * nfr, Npad, device=dev, dtype=torch.float32)
            if cfg.w_log_mag:      # the log-magnitude gradient weighs a bin by 1 / |X|^2: its spectrum is accumulated in double
                ops.stft_f64acc(xp, r["w"], spec, 2 * R, nfr, ldp, hop, off, K, Npad, lo_off=2 * R * ldp if pair else 0)
            else:
                ops.linear(xp, r["w"], None, spec, (2 * R, nfr, 1), (ldp, hop, 0), (nfr * Npad, Npad, 0), K, Npad, in_off=off)
            cnt = float(R * nfr * nbins)
            if cfg.w_sc or cfg.w_log_mag:     # auraloss's other two terms: two passes (the SC gradient needs the global norms)
                dsx = ops.stft_mag_terms(spec[: R * nfr], spec[R * nfr:], R * nfr, nbins, Npad, cfg.eps, cfg.w_lin_mag,
                                         cfg.w_log_mag, cfg.w_sc, 1.0 / nres, loss, want_grad)
            else:
                dsx = ops.stft_mag_l1(spec[: R * nfr], spec[R * nfr:], R * nfr, nbins, Npad, cfg.eps,
                                      cfg.w_lin_mag / (nres * cnt), loss, cfg.w_lin_mag / (nres * cnt), want_grad)
            if want_grad:
                dfr = torch.empty(R * nfr, K, device=dev, dtype=torch.float32)
                gP, s_in = dense(R * nfr, Npad)
                _, s_out = dense(R * nfr, K)
                ops.linear(dsx, r["wT"], None, dfr, gP, s_in, s_out, Npad, K)
                ops.frames_fold(dfr, dxw, nfr, K, K, hop, off, pad, accumulate=i > 0)
        dest = None
        if want_grad:
            dest = ops.fir(dxw, cfg.taps_rev) if cfg.taps is not None else dxw
        if cfg.l1_ratio > 0:
            ops.l1_grad(e, g, cfg.l1_ratio / e.numel(), dest, True, loss, cfg.l1_ratio / e.numel())
        ctx.save_for_backward(dest)
        ctx.shape = shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        (dest,) = ctx.saved_tensors
        return (dest * gout).view(ctx.shape), None, None
