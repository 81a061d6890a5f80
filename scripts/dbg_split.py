import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden, rel_l2
import sound_bubble_amd as sb
from sound_bubble_amd import ops
from sound_bubble_amd.functional import SnrlpLossFn
from sound_bubble_amd.train import FlatBucket
rec, params, _ = load_golden("tiny_big")
for nb in (2, 6):
    m = sb.NetDisEmbd3(**dict(params, B=nb))
    torch.manual_seed(3)
    for p in m.parameters():
        torch.nn.init.uniform_(p, -0.2, 0.2) if p.dim() > 1 else None
    m = m.cuda().train()
    bucket = FlatBucket(m)
    g = torch.Generator().manual_seed(11)
    mix = 0.1 * torch.randn(4, 6, 192 * 150 + 96, generator=g)
    tgt = 0.05 * torch.randn(4, 1, 192 * 150, generator=g)
    dis = torch.eye(3)[torch.arange(4) % 3]
    def grad(sl):
        bucket.zero_grad()
        est = m({"mixture": mix[sl].cuda(), "dis_embed": dis[sl].cuda()}, pad=False)["output"]
        loss, lv = SnrlpLossFn.apply(est, tgt[sl].cuda(), 100.0)
        loss.backward()
        return bucket.grad.clone().cpu().numpy(), lv.cpu().numpy()
    ga, la = grad(slice(0, 4)); g0, l0 = grad(slice(0, 2)); g1, l1 = grad(slice(2, 4))
    print(nb, "loss vecs", la, l0, l1, "rel", rel_l2((g0 + g1) / 2, ga), "norms", np.linalg.norm(ga), np.linalg.norm(g0), np.linalg.norm(g1))
