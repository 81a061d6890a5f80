"""Loss modules with the reference's constructor surface (JSON: "loss": "src.losses.SNRLP.SNRLPLoss")."""
import torch.nn as nn

from .functional import SnrlpLossFn


class SNRLPLoss(nn.Module):
    """src/losses/SNRLP.py:9-42 with snr_loss_name='snr' (the only name the shipped configs use).
    forward(est, gt) -> per-sample loss vector [B] (the harness takes .mean(), hl_module:321).
    `.mean_loss(est, gt)` returns the differentiable batch mean computed by the fused HIP kernel."""

    def __init__(self, snr_loss_name="snr", neg_weight=1):
        super().__init__()
        if snr_loss_name != "snr":
            raise NotImplementedError("only snr_loss_name='snr' (every shipped pre-train config) is built")
        self.neg_weight = float(neg_weight)

    def mean_loss(self, est, gt):
        return SnrlpLossFn.apply(est, gt, self.neg_weight)        # (mean, per-sample vector)

    def forward(self, est, gt, **kwargs):
        return self.mean_loss(est, gt)[1]
