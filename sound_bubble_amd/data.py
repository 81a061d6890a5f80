"""Synthetic stand-in for the reference datasets (schema of
src/datasets/general_multisrc_dataset_dis_embed.py:204-216): items are
(inputs{mixture [6,N], dis_embed [3]}, targets{target [1,N], num_target_speakers, num_interfering_speakers,
num_noises}).  Dataset IO / scene synthesis is out of scope; this feeds the harness and bench
with inputs of the right shape and statistics."""
import torch


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
