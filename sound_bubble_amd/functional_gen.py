"""Stage-level autograd functions of the GridNet blocks at GENERIC layer widths -- everything the tuned kernels of
functional.py (D in {16, 32}, H = 64: every shipped experiment JSON) are not built for, first of all the reference
constructors' own defaults D = 64, H = 128 (src/models/tfgridnet_realtime_clean_dis_embd3/net.py:21-26,
src/models/tfgridnet_realtime_clean_optim/net.py:21-26).

Same stages, same channels-last layout, same C-ABI library; the recurrences run on the generic-shape kernels of
csrc/sb_lstm_gen.hip (exact fp32 products, position-major fp32 records) and every matrix product around them is a
position-wise GEMM (sb_linear_fwd / sb_wgrad's generic form).  None of the fusions and overlapped schedules of the tuned
path: this is the correct-first instantiation.  No ATen arithmetic -- torch allocates and transposes weights, nothing else.
"""
import torch

from . import _lib as L
from . import ops
from .ops import Geom, dense
from .functional import _GradTargets
from . import functional as Fn


def _train(ctx):
    return Fn.GRAD_MODE and any(ctx.needs_input_grad)


class GenIntraPlainFn(torch.autograd.Function):
    """y = x + Linear_{2H->C}(biLSTM_F(LN_C(x)))   -- dis_embd3/tfgridnet_causal.py:795,818-827; optim :699-707."""

    @staticmethod
    def forward(ctx, x, ln_g, ln_b, wif, whf, bif, bhf, wir, whr, bir, bhr, lin_w, lin_b):
        B, T, F, Cc = x.shape
        Hh = whf.shape[1]
        x = x.contiguous()
        P = B * T * F
        train = _train(ctx)
        geom = Geom.intra(B * T, F)
        dirs = [(wif, whf, bif, bhf), (wir, whr, bir, bhr)]
        hs, _, rec, u = ops.lstm_gen_fwd(x.view(P, Cc), ln_g, ln_b, dirs, geom, save=train)
        y = torch.empty_like(x)
        g, s_in = dense(P, 2 * Hh)
        _, s_out = dense(P, Cc)
        ops.linear(hs, lin_w.contiguous(), lin_b, y, g, s_in, s_out, 2 * Hh, Cc, epi=L.EPI_RES, res=x)
        if train:
            ctx.save_for_backward(x, ln_g, ln_b, wif, whf, bif, bhf, wir, whr, bir, bhr, lin_w, lin_b, hs, rec, u)
            ctx.dims = (B, T, F, Cc, Hh)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ln_g, ln_b, wif, whf, bif, bhf, wir, whr, bir, bhr, lin_w, lin_b, hs, rec, u = ctx.saved_tensors
        B, T, F, Cc, Hh = ctx.dims
        P = B * T * F
        gt = _GradTargets()
        dy = dy.contiguous()
        geom = Geom.intra(B * T, F)
        gP, sC = dense(P, Cc)
        _, s2H = dense(P, 2 * Hh)
        dhs = torch.empty(P, 2 * Hh, device=dy.device, dtype=torch.float32)
        ops.linear(dy, lin_w.t().contiguous(), None, dhs, gP, sC, s2H, Cc, 2 * Hh)
        ops.wgrad(dy, Cc, Cc, hs, s2H, gP, 2 * Hh, gt("lin_w", lin_w), dbias=gt("lin_b", lin_b))
        tg = [(gt("wif", wif), gt("whf", whf), gt("bif", bif), gt("bhf", bhf)),
              (gt("wir", wir), gt("whr", whr), gt("bir", bir), gt("bhr", bhr))]
        du = ops.lstm_gen_bwd([(wif, whf, bif, bhf), (wir, whr, bir, bhr)], rec, dhs, u, hs, geom, tg)
        dx, _, _, _ = ops.ln_bwd(du, x.view(P, Cc), ln_g, res=dy.view(P, Cc), d_g=gt("ln_g", ln_g), d_b=gt("ln_b", ln_b))
        return (dx.view(B, T, F, Cc), gt["ln_g"], gt["ln_b"], gt["wif"], gt["whf"], gt["bif"], gt["bhf"], gt["wir"], gt["whr"],
                gt["bir"], gt["bhr"], gt["lin_w"], gt["lin_b"])


class GenInterFn(torch.autograd.Function):
    """y = x + Linear_{H->C}(LSTM_T(LN_C(x)), carried (h0, c0))  -- dis_embd3/tfgridnet_causal.py:830-849; optim :709-728.
    Returns (y, hN, cN); the state rows are b*F + f as in the reference."""

    @staticmethod
    def forward(ctx, x, ln_g, ln_b, wi, wh, bi, bh, lin_w, lin_b, h0, c0):
        B, T, F, Cc = x.shape
        Hh = wh.shape[1]
        x = x.contiguous()
        P = B * T * F
        train = _train(ctx)
        geom = Geom.inter(B, T, F)
        h0c = h0.reshape(B * F, Hh).contiguous() if h0 is not None else None
        c0c = c0.reshape(B * F, Hh).contiguous() if c0 is not None else None
        hs, (hN, cN), rec, u = ops.lstm_gen_fwd(x.view(P, Cc), ln_g, ln_b, [(wi, wh, bi, bh)], geom, h0=h0c, c0=c0c,
                                                save=train, want_state=True)
        y = torch.empty_like(x)
        g, s_in = dense(P, Hh)
        _, s_out = dense(P, Cc)
        ops.linear(hs, lin_w.contiguous(), lin_b, y, g, s_in, s_out, Hh, Cc, epi=L.EPI_RES, res=x)
        if train:
            ctx.save_for_backward(x, ln_g, ln_b, wi, wh, bi, bh, lin_w, lin_b, hs, rec, u)
            ctx.dims = (B, T, F, Cc, Hh)
        hN, cN = hN.view(1, B * F, Hh), cN.view(1, B * F, Hh)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(hN, cN)
        return y, hN, cN

    @staticmethod
    def backward(ctx, dy, _dh, _dc):
        x, ln_g, ln_b, wi, wh, bi, bh, lin_w, lin_b, hs, rec, u = ctx.saved_tensors
        B, T, F, Cc, Hh = ctx.dims
        P = B * T * F
        gt = _GradTargets()
        dy = dy.contiguous()
        geom = Geom.inter(B, T, F)
        gP, sC = dense(P, Cc)
        _, sH = dense(P, Hh)
        dhs = torch.empty(P, Hh, device=dy.device, dtype=torch.float32)
        ops.linear(dy, lin_w.t().contiguous(), None, dhs, gP, sC, sH, Cc, Hh)
        ops.wgrad(dy, Cc, Cc, hs, sH, gP, Hh, gt("lin_w", lin_w), dbias=gt("lin_b", lin_b))
        # h_prev of (b, t, f) is hs at (b, t - 1, f); rows with t == 0 see h0, whose gradient is not produced (net.py:88-89:
        # training starts every utterance from zero state)
        tg = [(gt("wi", wi), gt("wh", wh), gt("bi", bi), gt("bh", bh))]
        du = ops.lstm_gen_bwd([(wi, wh, bi, bh)], rec, dhs, u, hs, geom, tg)
        dx, _, _, _ = ops.ln_bwd(du, x.view(P, Cc), ln_g, res=dy.view(P, Cc), d_g=gt("ln_g", ln_g), d_b=gt("ln_b", ln_b))
        return (dx.view(B, T, F, Cc), gt["ln_g"], gt["ln_b"], gt["wi"], gt["wh"], gt["bi"], gt["bh"], gt["lin_w"], gt["lin_b"],
                None, None)


class GenIntraConvFn(torch.autograd.Function):
    """Conv-LSTM intra path: Conv1d(C->C, k = s = down) -> PReLU -> LN -> biLSTM over F // down steps ->
    ConvTranspose1d(2H->C, k = s = down) -> + x.   optim/tfgridnet_causal.py:684-697,706-707; dis_embd3 :800-813
    (bias_tail = False there: frequencies beyond down * floor(F / down) get no deconv output).
    wc / wd / bd / wdT / wcT: kernel-layout forms of the two convolution weights (forms.WeightForms, as in functional.IntraConvFn)."""

    @staticmethod
    def forward(ctx, x, conv_w, conv_b, act_a, ln_g, ln_b, wif, whf, bif, bhf, wir, whr, bir, bhr, dec_w, dec_b,
                down, bias_tail, wc, wd, bd, wdT, wcT):
        B, T, F, Cc = x.shape
        Hh = whf.shape[1]
        x = x.contiguous()
        Kd = F // down
        Fm = Kd * down
        P2 = B * T * Kd
        train = _train(ctx)
        dev = x.device
        v_pre = torch.empty(P2, Cc, device=dev, dtype=torch.float32) if train else None
        a = torch.empty(P2, Cc, device=dev, dtype=torch.float32)
        grid = (B * T, 1, Kd)
        s_x = (F * Cc, 0, down * Cc)
        s_a = (Kd * Cc, 0, Cc)
        ops.linear(x, wc, conv_b, a, grid, s_x, s_a, down * Cc, Cc, epi=L.EPI_PRELU, prelu_a=act_a, aux_out=v_pre)
        geom = Geom.intra(B * T, Kd)
        dirs = [(wif, whf, bif, bhf), (wir, whr, bir, bhr)]
        hs, _, rec, u = ops.lstm_gen_fwd(a, ln_g, ln_b, dirs, geom, save=train)
        y = torch.empty_like(x)
        s_h = (Kd * 2 * Hh, 0, 2 * Hh)
        ops.linear(hs, wd, bd, y, grid, s_h, s_x, 2 * Hh, down * Cc, epi=L.EPI_RES, res=x)
        if Fm < F:      # tail frequencies: residual (+ bias when ConvTranspose1d has output_padding)
            ops.tail_rows(x, dec_b if bias_tail else None, y, B * T, F, Fm, Cc)
        if train:
            ctx.save_for_backward(x, act_a, ln_g, ln_b, wif, whf, bif, bhf, wir, whr, bir, bhr, hs, u, v_pre, rec,
                                  conv_w, conv_b, dec_w, dec_b, wdT, wcT)
            ctx.dims = (B, T, F, Cc, Hh, down, Kd, bool(bias_tail))
        return y

    @staticmethod
    def backward(ctx, dy):
        (x, act_a, ln_g, ln_b, wif, whf, bif, bhf, wir, whr, bir, bhr, hs, u, v_pre, rec, conv_w, conv_b, dec_w, dec_b,
         wdT, wcT) = ctx.saved_tensors
        B, T, F, Cc, Hh, down, Kd, bias_tail = ctx.dims
        gt = _GradTargets()
        Fm = Kd * down
        P2 = B * T * Kd
        dev = dy.device
        dy = dy.contiguous()
        dym = dy if Fm == F else dy[:, :, :Fm, :].contiguous()        # dense rows [P2, down*C]
        NC = down * Cc
        grid = (B * T, 1, Kd)
        gP2, sNC = dense(P2, NC)
        _, s2H = dense(P2, 2 * Hh)
        # ConvTranspose1d backward: data gradient, then dW[n = j*C + c][k = h] and its bias sums straight into the parameters'
        # native layout -- dec_w [2H, C, down] (transposed, rows n -> c*down + j), dec_b [C] (rows folded mod C)
        dhs = torch.empty(P2, 2 * Hh, device=dev, dtype=torch.float32)
        ops.linear(dym, wdT, None, dhs, gP2, sNC, s2H, NC, 2 * Hh)
        t_dec_w, t_dec_b = gt("dec_w", dec_w), gt("dec_b", dec_b)
        ops.wgrad(dym, NC, NC, hs, s2H, gP2, 2 * Hh, t_dec_w, dbias=t_dec_b, transpose_out=True, perm_n=Cc, bias_mod=Cc)
        if Fm < F and bias_tail:
            for f in range(Fm, F):
                ops.colsum(dy, B * T, F * Cc, Cc, t_dec_b, g_off=f * Cc)
        # BPTT
        geom = Geom.intra(B * T, Kd)
        tg = [(gt("wif", wif), gt("whf", whf), gt("bif", bif), gt("bhf", bhf)),
              (gt("wir", wir), gt("whr", whr), gt("bir", bir), gt("bhr", bhr))]
        du = ops.lstm_gen_bwd([(wif, whf, bif, bhf), (wir, whr, bir, bhr)], rec, dhs, u, hs, geom, tg)
        dv, _, _, _ = ops.ln_bwd(du, v_pre, ln_g, prelu_a=act_a, d_g=gt("ln_g", ln_g), d_b=gt("ln_b", ln_b),
                                 d_a=gt("act_a", act_a))
        # Conv1d backward: dx = dy + dv . Wc ; dWc = dv^T x_rows
        dx = torch.empty_like(x)
        s_x = (F * Cc, 0, NC)
        s_v = (Kd * Cc, 0, Cc)
        ops.linear(dv, wcT, None, dx, grid, s_v, s_x, Cc, NC, epi=L.EPI_RES, res=dy)
        if Fm < F:
            ops.tail_rows(dy, None, dx, B * T, F, Fm, Cc)
        # dW[co][k = j*C + ci] -> conv_w [co, ci, j]
        ops.wgrad(dv, Cc, Cc, x, s_x, grid, NC, gt("conv_w", conv_w), dbias=gt("conv_b", conv_b), perm_k=Cc)
        return (dx, gt["conv_w"], gt["conv_b"], gt["act_a"], gt["ln_g"], gt["ln_b"], gt["wif"], gt["whf"], gt["bif"],
                gt["bhf"], gt["wir"], gt["whr"], gt["bir"], gt["bhr"], gt["dec_w"], gt["dec_b"], None, None,
                None, None, None, None, None)
