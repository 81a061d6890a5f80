"""Batched on-device metrics from one pass over (est, gt, mix) and ONE D2H copy.

Replaces the per-sample `.item()` loop of PLModule._step (hl_module:326-373) and
src/metrics/metrics.py:44-55 (torchmetrics snr / si_snr / si_sdr, third-party: restated from their
published definitions -- eps = float32 eps inside the ratios; si_snr = zero-mean si_sdr) and
compute_decay (metrics.py:20-36)."""
import numpy as np

from . import ops

EPS = float(np.finfo(np.float32).eps)


def _ratio_db(num, den):
    return 10.0 * np.log10((num + EPS) / (den + EPS))


def _snr(ss_p, ss_t, s_pt):                 # sum (t - p)^2 = tt - 2pt + pp
    return _ratio_db(ss_t, ss_t - 2.0 * s_pt + ss_p)


def _si_sdr(ss_p, ss_t, s_pt):
    alpha = (s_pt + EPS) / (ss_t + EPS)
    ts = alpha * alpha * ss_t
    noise = ts - 2.0 * alpha * s_pt + ss_p
    return _ratio_db(ts, noise)


def batch_metrics(est, gt, mix_ref, names=("snr_i", "si_snr_i", "si_sdr_i")):
    """est, gt: [B, 1, N] on the GPU; mix_ref: [B, N] view of the reference mixture channel.
    Returns {name: np.ndarray [B]} (+ 'decay') computed in float64 on the host from the moments."""
    B = est.shape[0]
    N = est.shape[-1]
    m = ops.signal_stats(est.reshape(B, N).contiguous(), gt.reshape(B, N).contiguous(), mix_ref).cpu().numpy()
    m = m.astype(np.float64)
    se, st, sm, see, stt, smm, set_, smt = (m[:, k] for k in range(8))
    out = {}
    # zero-mean second moments
    cee, ctt, cmm = see - se * se / N, stt - st * st / N, smm - sm * sm / N
    cet, cmt = set_ - se * st / N, smt - sm * st / N
    vals = {
        "snr": _snr(see, stt, set_), "snr_mix": _snr(smm, stt, smt),
        "si_sdr": _si_sdr(see, stt, set_), "si_sdr_mix": _si_sdr(smm, stt, smt),
        "si_snr": _si_sdr(cee, ctt, cet), "si_snr_mix": _si_sdr(cmm, ctt, cmt),
    }
    for n in names:
        if n.endswith("_i"):
            out[n] = vals[n[:-2]] - vals[n[:-2] + "_mix"]
        else:
            out[n] = vals[n]
    with np.errstate(divide="ignore"):
        out["decay"] = 10.0 * np.log10(smm) - 10.0 * np.log10(see)
    return out
