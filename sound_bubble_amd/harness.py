"""Training / evaluation harness with the reference's protocol.

Mirrors the object `train_pt.py` drives (src/hl_modules/distance_based_hl_module.py:21-441): same
constructor keywords (so `pl_module_args` of the shipped JSONs pass through unchanged) and the same
methods -- train(), eval(), training_step(), validation_step(), reset_grad(), backprop(),
on_epoch_start(), on_epoch_end(best_path, wandb_run), dump_state(), load_state(), get_current_lr().

MI355X-native differences:
  * one process per GPU; gradients live in one flat bucket; `backprop()` = ONE RCCL all-reduce +
    fused clip + Adam (HIP) instead of nn.DataParallel + per-tensor optimizer ops;
  * metrics come from one fused moment kernel and ONE D2H copy per step instead of
    O(batch x metrics) `.item()` syncs (hl_module:334-373);
  * wandb is optional (no network on the GPU box).
Checkpoints keep the reference layout {model, optimizer, current_epoch, metric_values, statistics
[, scheduler]} with the reference's state_dict key names, so a reference `best.pt`'s model weights load.
"""
import importlib

import numpy as np
import torch

from .metrics import batch_metrics
from .train import FlatBucket, FusedAdam, allreduce_grads, broadcast_replica

ALIASES = {
    "src.models.tfgridnet_realtime_clean_dis_embd3.net.Net": "sound_bubble_amd.net.NetDisEmbd3",
    "src.models.tfgridnet_realtime_clean_optim.net.Net": "sound_bubble_amd.net.NetOptim",
    "src.losses.SNRLP.SNRLPLoss": "sound_bubble_amd.losses.SNRLPLoss",
    "src.losses.MultiResoLoss.MultiResoFuseLoss": "sound_bubble_amd.losses.MultiResoFuseLoss",
    "src.hl_modules.distance_based_hl_module.PLModule": "sound_bubble_amd.harness.PLModule",
    "src.datasets.general_multisrc_dataset_dis_embed.Dataset": "sound_bubble_amd.data.BubbleFolderDataset",
}


def import_attr(path):
    """utils.import_attr (src/utils.py:10-12) with the drop-in alias table."""
    path = ALIASES.get(path, path)
    module, attr = path.rsplit(".", 1)
    return getattr(importlib.import_module(module), attr)


def load_model_weights(path):
    """state_dict of a reference checkpoint: `last.pt` / `best.pt` ({'model': ...}, hl_module:88-93) or a Lightning-era
    `.ckpt` ({'state_dict': ...} whose keys carry the `model.` prefix the reference strips through FakeModel,
    hl_module:74-86)."""
    state = torch.load(path, map_location="cpu", weights_only=False)
    if "model" in state:
        return state["model"]
    sd = state["state_dict"]
    return {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}


class _LrCarrier(torch.optim.Optimizer):
    """Minimal torch optimizer whose only job is to carry `lr` for torch.optim.lr_scheduler.*"""

    def __init__(self, lr):
        self._p = torch.nn.Parameter(torch.zeros(1))
        super().__init__([self._p], dict(lr=lr))

    def step(self, closure=None):
        return None


class PLModule(object):
    def __init__(self, model, model_params, sr, optimizer, optimizer_params, scheduler=None, scheduler_params=None,
                 loss=None, loss_params=None, metrics=[], init_ckpt=None, grad_clip=None, use_dp=True,
                 val_log_interval=10, samples_per_speaker_number=3, device="cuda"):
        self.model = import_attr(model)(**model_params).to(device)
        self.use_dp = use_dp                    # kept for signature compatibility; DP = one process per GPU here
        self.sr = sr
        self.samples_per_speaker_number = samples_per_speaker_number
        self.metric_names = [m for m in metrics if m not in ("PESQ", "STOI")]
        self.metric_values, self.statistics = {}, {}
        self.monitor, self.monitor_mode, self.mode = "val/loss", "min", None
        self.loss_fn = import_attr(loss)(**(loss_params or {}))
        if init_ckpt is not None:
            self.model.load_state_dict(load_model_weights(init_ckpt))
        if optimizer not in ("torch.optim.Adam",):
            raise NotImplementedError(f"optimizer {optimizer}: only torch.optim.Adam (every shipped config) is fused")
        self.optim_name, self.opt_params = optimizer, dict(optimizer_params)
        self.bucket = FlatBucket(self.model)
        self.optimizer = FusedAdam(self.bucket, **self.opt_params)
        self.grad_clip = grad_clip
        if self.grad_clip is None:
            print("NOT USING GRAD CLIP (pl_module_args.grad_clip is unset -- as in syn_experiments/pretrain_stage.json)")
        self.scheduler_name, self.scheduler_params = scheduler, scheduler_params
        self._lr_carrier = _LrCarrier(self.opt_params.get("lr", 1e-3))
        self.scheduler = self.init_scheduler(scheduler, scheduler_params)
        self._sync_lr()          # e.g. LinearLR applies its start_factor at construction
        self.epoch = 0
        self._loss = None
        broadcast_replica(self.bucket, self.optimizer)      # N > 1: every rank starts as rank 0's replica

    # ---- checkpoints (hl_module:115-156) ----
    def dump_state(self, path):
        state = dict(model=self.model.state_dict(), optimizer=self.optimizer.state_dict(), current_epoch=self.epoch,
                     metric_values=self.metric_values, statistics=self.statistics)
        if self.scheduler is not None:
            state["scheduler"] = self.scheduler.state_dict()
        torch.save(state, path)

    def load_state(self, path, map_location=None):
        """hl_module:115-139.  Reads the reference's own last.pt / best.pt: `optimizer` in torch.optim.Adam.state_dict()
        layout is converted into the flat moment buffers (FusedAdam.load_state_dict), so a resumed run continues with
        the saved moments, step count and learning rate."""
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if multi:
            # only rank 0 needs to see the file (node-local run directories): it reads, the bookkeeping travels as one
            # object, parameters / moments by broadcast_replica below -- called by EVERY rank, whatever its file system holds
            box = [None]
            if dist.get_rank() == 0:
                st0 = torch.load(path, map_location=map_location or "cpu", weights_only=False)
                box[0] = {k: st0[k] for k in ("scheduler", "current_epoch", "metric_values", "statistics") if k in st0}
            dist.broadcast_object_list(box, src=0)
            state = st0 if dist.get_rank() == 0 else box[0]
        else:
            state = torch.load(path, map_location=map_location or "cpu", weights_only=False)
        if "model" in state:
            self.model.load_state_dict(state["model"])        # in-place copy_: bucket views stay valid
        opt = state.get("optimizer")
        if isinstance(opt, dict):
            self.optimizer.load_state_dict(opt)
        elif opt is not None:
            raise ValueError(f"{path}: unrecognised optimizer state ({type(opt).__name__})")
        if self.scheduler is not None and "scheduler" in state:
            self.scheduler = self.init_scheduler(self.scheduler_name, self.scheduler_params)
            self.scheduler.load_state_dict(state["scheduler"])
        # scheduler state does not carry the optimizer's lr (the reference restores it through the
        # optimizer state_dict): push the saved lr back into the carrier the scheduler drives
        for g in self._lr_carrier.param_groups:
            g["lr"] = self.optimizer.param_groups[0]["lr"]
        self.epoch = state.get("current_epoch", 0)
        self.metric_values = state.get("metric_values", {})
        if "statistics" in state:
            self.statistics = state["statistics"]
        broadcast_replica(self.bucket, self.optimizer)
        for g in self._lr_carrier.param_groups:               # (ranks > 0 received the lr just now)
            g["lr"] = self.optimizer.param_groups[0]["lr"]

    def get_current_lr(self):
        return self.optimizer.param_groups[0]["lr"]

    def _sync_lr(self):
        self.optimizer.param_groups[0]["lr"] = self._lr_carrier.param_groups[0]["lr"]

    # ---- epoch bookkeeping (hl_module:162-264) ----
    def on_epoch_start(self):
        print("\n" + "=" * 25, "STARTING EPOCH", self.epoch, "=" * 25 + "\n")

    def get_avg_metric_at_epoch(self, metric, epoch=None):
        epoch = self.epoch if epoch is None else epoch
        m = self.metric_values[epoch][metric]
        return m["epoch"] / m["num_elements"]

    def sync_epoch_metrics(self):
        """N > 1: merge this epoch's (sum, count) pairs of every logged metric over the ranks, so that the plateau
        scheduler, the best-checkpoint decision and the printed / logged epoch means see the WHOLE train / val set on
        every rank (the reference has one process, hl_module:174-202).  Ranks may hold different metric names (e.g.
        `si_sdr_i_2spk` only where a 2-speaker scene fell) or an empty val shard: the union of names is kept."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        mine = {k: (v["epoch"], v["num_elements"]) for k, v in self.metric_values.get(self.epoch, {}).items()
                if v.get("epoch") is not None}
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, mine)
        merged = {}
        for d in gathered:
            for k, (s_, n_) in d.items():
                a = merged.setdefault(k, [0.0, 0])
                a[0] += s_
                a[1] += n_
        ep = self.metric_values.setdefault(self.epoch, {})
        for k, (s_, n_) in merged.items():
            ev = ep.setdefault(k, dict(step=None, epoch=None))
            ev["epoch"], ev["num_elements"] = s_, n_

    def _check_sched(self):
        """the guarded schedules' watchdog word (all ranks together).  A trip is FATAL for the run but harmless for the weights:
        from the tripped step on the optimiser's kernel sees the word and skips its update (train.FusedAdam.skipped), so what
        is raised here reports garbage gradients that were never applied -- last.pt / best.pt hold the last good state."""
        from . import ops, _lib as L
        try:
            ops.check_sched_status_all_ranks()
        except L.SoundBubbleHipError as e:
            n = int(self.optimizer.skipped.item())
            self.optimizer.skipped.zero_()
            raise L.SoundBubbleHipError(
                f"{e}  The optimiser turned {n} step(s) since the trip into no-ops on this rank: parameters and Adam moments hold "
                "their last good values.") from None

    def on_epoch_end(self, best_path, wandb_run=None):
        self.sync_epoch_metrics()
        from . import ops
        # one sync per epoch: did a time-segmented / overlapped launch bail out?  (verdict shared by all ranks: a lone
        # raising rank would leave the others hanging in the next all-reduce); and is the side stream still concurrent?
        self._check_sched()
        if ops._OVERLAP_OK:
            g = ops.read_giveups()
            if g != getattr(self, "_giveups_seen", 0):
                import warnings
                warnings.warn(f"{g - getattr(self, '_giveups_seen', 0)} workgroup(s) of the overlapped forward stopped waiting for their "
                              "producer this epoch (results unaffected: the launch behind the producer did their items; the step "
                              "that happened in took a few ms longer)")
                self._giveups_seen = g
            ops.overlap_reprobe()
            # which order the passes of this epoch really took (a lost side stream switches to the plain order: same results,
            # other throughput -- the number a slow epoch is explained by)
            seen = getattr(self, "_sched_seen", {})
            print(f"schedules this epoch: { {k: v - seen.get(k, 0) for k, v in ops.SCHED_COUNTS.items()} }"
                  + (" [side stream LOST: plain order]" if ops._OVERLAP_LOST else ""))
            self._sched_seen = dict(ops.SCHED_COUNTS)
        last = self.get_avg_metric_at_epoch(self.monitor)
        best = all(not (last > self.get_avg_metric_at_epoch(self.monitor, e)) for e in range(len(self.metric_values) - 1))
        if best:
            print("Current checkpoint is the best! Saving it...")
            self.dump_state(best_path)
        for k in sorted(self.metric_values[self.epoch]):
            if k.startswith("val/"):
                print(f"{k}: {self.get_avg_metric_at_epoch(k):.3f}")
        if wandb_run is not None:
            wandb_run.log({"lr-Adam": self.get_current_lr(), "epoch": self.epoch,
                           **{k: self.get_avg_metric_at_epoch(k) for k in self.metric_values[self.epoch]}},
                          step=self.epoch + 1)
        if self.scheduler is not None:
            if isinstance(self.scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau):
                self.scheduler.step(last)
            else:
                self.scheduler.step()
            self._sync_lr()
        self.epoch += 1

    def log_metric(self, name, value, batch_size=1, on_step=False, on_epoch=True, **_):
        ev = self.metric_values.setdefault(self.epoch, {}).setdefault(name, dict(step=None, epoch=None))
        if on_step:
            ev["step"] = (ev["step"] or []) + [float(value)]
        if on_epoch:
            if ev["epoch"] is None:
                ev["epoch"], ev["num_elements"] = 0.0, 0
            ev["epoch"] += float(value) * batch_size
            ev["num_elements"] += batch_size

    # ---- steps (hl_module:303-428) ----
    def _step(self, batch, batch_idx, step="train"):
        inputs, targets = batch
        B = inputs["mixture"].shape[0]
        outputs = self.model(inputs)
        est, gt = outputs["output"], targets["target"]
        loss, loss_vec = self.loss_fn.mean_loss(est, gt)
        with torch.no_grad():
            n_spk = np.asarray(targets["num_target_speakers"].cpu() if torch.is_tensor(targets["num_target_speakers"])
                               else targets["num_target_speakers"]).reshape(-1)
            mix_ref = inputs["mixture"][:, 0, : est.shape[-1]]
            mets = batch_metrics(est.detach(), gt, mix_ref, self.metric_names)       # ONE D2H copy
            self._loss_host = None
            for name in self.metric_names:
                for i in range(B):
                    if n_spk[i] > 0:
                        self.log_metric(f"{step}/{name}", mets[name][i], 1)
                        if name == "si_sdr_i":
                            self.log_metric(f"{step}/{name}_{int(n_spk[i])}spk", mets[name][i], 1)
            for i in range(B):
                if n_spk[i] == 0:
                    self.log_metric(f"{step}/decay", mets["decay"][i], 1)
        self._pending = (step, loss.detach(), B)
        return loss, B

    def _flush_loss(self):
        if getattr(self, "_pending", None) is not None:
            step, l, B = self._pending
            self.log_metric(f"{step}/loss", float(l), B, on_step=(step == "train"))
            self._pending = None

    def train(self):
        self.model.train()
        self.mode = "train"

    def eval(self):
        self.model.eval()
        self.mode = "val"

    def training_step(self, batch, batch_idx):
        return self._step(batch, batch_idx, "train")

    def validation_step(self, batch, batch_idx):
        out = self._step(batch, batch_idx, "val")
        self._flush_loss()
        return out

    def reset_grad(self):
        self.bucket.zero_grad()

    def backprop(self):
        """all-reduce (one flat bucket) -> clip_grad_norm_(grad_clip) -> Adam, hl_module:430-441."""
        world = allreduce_grads(self.bucket)
        self.optimizer.step(grad_clip=self.grad_clip, world_size=world)
        self._flush_loss()
        # the guarded schedules' watchdog word, every 50th step as well as at the epoch's end (an epoch of a real run is hours:
        # a launch that gave up must not train on garbage that long).  Same step on every rank: the verdict is all-reduced.
        self._opt_steps = getattr(self, "_opt_steps", 0) + 1
        if self._opt_steps % 50 == 0:
            self._check_sched()

    def init_scheduler(self, scheduler, scheduler_params):
        """hl_module:460-481 ('sequential' -> SequentialLR with cumulative milestones)."""
        if scheduler is None:
            return None
        opt = self._lr_carrier
        if scheduler == "sequential":
            scheds, miles = [], []
            for sp in scheduler_params:
                scheds.append(import_attr(sp["name"])(opt, **sp["params"]))
                miles.append(sp["epochs"])
            miles = list(np.cumsum(miles))[:-1]
            return torch.optim.lr_scheduler.SequentialLR(opt, scheds, [int(m) for m in miles])
        return import_attr(scheduler)(opt, **scheduler_params)
