"""Synthetic stand-in for the reference datasets (schema of
src/datasets/general_multisrc_dataset_dis_embed.py:204-216): items are
(inputs{mixture [6,N], dis_embed [3]}, targets{target [1,N], num_target_speakers, num_interfering_speakers,
num_noises}).  Scene synthesis is out of scope; SyntheticBubbleDataset feeds the harness and bench
with inputs of the right shape and statistics, BubbleFolderDataset reads rendered scene folders in the
reference's own layout (the bundled test_samples/ are such folders) so that `train_cli` can train on real scenes."""
import glob
import json
import os

import numpy as np
import torch

# folder name -> bubble radius in metres (general_multisrc_dataset_dis_embed.py:44-64)
RADIUS_OF = {"syn_1m": 1.0, "syn_1_5m": 1.5, "syn_2m": 2.0, "glasses_1m": 1.0, "glass_1_5m": 1.5, "glass_2m": 2.0,
             "hearing_1_5m": 1.5, "hearing2_1_5m": 1.5, "binural_1_5m": 1.5}
ONE_HOT = {1.0: [0.0, 0.0, 1.0], 1.5: [0.0, 1.0, 0.0], 2.0: [1.0, 0.0, 0.0]}      # :191-198


class BubbleFolderDataset(torch.utils.data.Dataset):
    """Scene-folder dataset with the constructor surface and item schema of
    src/datasets/general_multisrc_dataset_dis_embed.py:19-218 (the `train_dataset` / `val_dataset` of
    syn_experiments/*.json): every `dataset_dirs[i] = {"path", "max_samples"}` holds numbered scene folders
    (mixture.wav: all microphones, PCM16; <mic>_<voice>.wav: one speaker at one microphone; metadata.json: voiceNN.dis in
    metres, or centimetres when metadata['real']).  The bubble radius comes from the folder NAME (.../syn_1m/train -> 1 m,
    :44-64; here the last path component may be the radius folder itself, as in the bundled test_samples/syn_1m).
    Ground truth = sum over the speakers with dis <= radius of their recording at the reference microphone (:150-166);
    scenes longer than `sig_len` seconds (at the constructor's `sr`: the reference's default 48 000 makes that 9 s of
    24 kHz audio, so 5 s scenes pass whole) are cropped at a random offset (:173-177).
    Not built: the audio perturbations (host-side augmentation) -- a non-empty list raises.
    Extension: `repeat` (each scene appears that many times per epoch).
    Where this class is MORE LENIENT than the reference's (none of it matters for the bundled scenes): (1) the radius
    folder is accepted as the LAST path component as well as the second-to-last (`binural_1_5m` as third-to-last or in the last
    two), where the reference looks at split('/')[-2] only ([-3] for binural); (2) `targets_outside` is allocated at the CROPPED
    length, the reference allocates it before cropping, at the uncropped one; (3) a missing metadata['real'] / ['n_BG'] reads as
    False / 0 where the reference raises KeyError."""

    def __init__(self, dataset_dirs, n_mics=6, sr=48000, directional=True, fair_compare=False, prob_neg=0,
                 perturbations=[], downsample=1, mic_config=[], sig_len=4.5, reference_channels=None, split="val",
                 repeat=1):
        if perturbations:
            raise NotImplementedError("BubbleFolderDataset: audio perturbations are not built (host-side augmentation)")
        if downsample != 1:
            raise NotImplementedError("BubbleFolderDataset: downsample != 1 is not built")
        self.dirs, self.radii = [], []
        for d in dataset_dirs:
            parts = os.path.normpath(d["path"]).split(os.sep)
            name = next((p for p in (parts[-2:] if len(parts) > 1 else parts)[::-1] if p in RADIUS_OF), None)
            if name is None and len(parts) > 2 and parts[-3] == "binural_1_5m":
                name = parts[-3]
            if name is None:
                raise ValueError(f"Invalid distance dataset: {d['path']}")
            scenes = sorted(p for p in glob.glob(os.path.join(d["path"], "[0-9]*")) if os.path.isdir(p))[: d["max_samples"]]
            self.dirs += scenes
            self.radii += [RADIUS_OF[name]] * len(scenes)
        self.mics = list(mic_config) or [f"mic{m:02d}" for m in range(n_mics)]
        assert n_mics == len(self.mics)
        self.reference_mics = [0] if reference_channels is None else list(reference_channels)
        self.sig_len = int(sig_len * sr / downsample)
        self.split, self.repeat = split, int(repeat)

    def __len__(self):
        return len(self.dirs) * self.repeat

    def __getitem__(self, idx):
        from .eval_samples import read_wav
        i = idx % len(self.dirs)
        d, radius = self.dirs[i], self.radii[i]
        with open(os.path.join(d, "metadata.json"), "rb") as f:
            meta = json.load(f)
        voices = [k for k in meta if "voice" in k]
        mics_all = [k for k in meta if "mic" in k]
        mixture, _ = read_wav(os.path.join(d, "mixture.wav"))
        if len(self.mics) < mixture.shape[0]:
            mixture = mixture[[int(m[-2:]) for m in self.mics]]
        target = np.zeros((len(self.reference_mics), mixture.shape[-1]), np.float32)
        n_tgt = 0
        for v in voices:
            dis = int(meta[v]["dis"]) / 100 if meta.get("real") else meta[v]["dis"]
            if dis <= radius:
                for c, mic in enumerate(self.reference_mics):
                    solo, _ = read_wav(os.path.join(d, f"{mics_all[mic]}_{v}.wav"))
                    target[c] += solo[0, : target.shape[-1]]
                n_tgt += 1
        assert (np.abs(target).max() > 0) == (n_tgt > 0), "zero target <=> no speaker inside the bubble (:168-171)"
        if self.sig_len < mixture.shape[-1]:
            b = np.random.randint(1000, mixture.shape[-1] - self.sig_len - 1)
            mixture, target = mixture[..., b: b + self.sig_len], target[..., b: b + self.sig_len]
        inputs = {"mixture": torch.from_numpy(np.ascontiguousarray(mixture)), "reference_channels": self.reference_mics,
                  "dis_embed": torch.tensor(ONE_HOT[radius])}
        targets = {"target": torch.from_numpy(np.ascontiguousarray(target)),
                   "targets_outside": torch.zeros(1, mixture.shape[-1]), "num_target_speakers": n_tgt,
                   "num_interfering_speakers": len(voices) - n_tgt, "num_noises": meta.get("n_BG", 0)}
        return inputs, targets


class SyntheticBubbleDataset(torch.utils.data.Dataset):
    def __init__(self, n_items=64, n_samples=120000, num_ch=6, seed=1234, with_dis_embed=True, silent_every=8,
                 split="train", **_ignored):
        self.n_items, self.n, self.m = n_items, n_samples, num_ch
        self.seed = seed + (0 if split == "train" else 7919)
        self.with_dis, self.silent_every = with_dis_embed, silent_every

    def __len__(self):
        return self.n_items

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        base = 0.1 * torch.randn(1, self.n + 8, generator=g)
        mix = torch.cat([base[..., 4 - min(m, 4): 4 - min(m, 4) + self.n] for m in range(self.m)], 0)
        mix = (mix + 0.02 * torch.randn(self.m, self.n, generator=g)).clamp(-1, 1)
        tgt = 0.05 * torch.randn(1, self.n, generator=g)
        n_tgt = 1 + (i % 2)
        if self.silent_every and i % self.silent_every == self.silent_every - 1:
            tgt.zero_()
            n_tgt = 0
        inputs = {"mixture": mix}
        if self.with_dis:
            d = torch.zeros(3)
            d[i % 3] = 1.0
            inputs["dis_embed"] = d
        targets = {"target": tgt, "num_target_speakers": n_tgt, "num_interfering_speakers": i % 3,
                   "num_noises": 1}
        return inputs, targets
