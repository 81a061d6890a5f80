import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import torch, numpy as np
from conftest import load_golden, golden_state_dict, poison_free_memory
import sound_bubble_amd as sb
from sound_bubble_amd import ops
from sound_bubble_amd.functional import SnrlpLossFn
rec, params, _ = load_golden("tiny_big")
m = sb.NetDisEmbd3(**params); m.load_state_dict(golden_state_dict(rec, torch)); m = m.cuda().train()
ops.OVERLAP_MIN_FILL = 0.0
for (B_, T_) in [(3, 131), (2, 150)]:
    torch.manual_seed(7)
    x = (0.1 * torch.randn(B_, 6, 192 * T_ + 96)).cuda()
    dis = torch.eye(3)[torch.arange(B_) % 3].cuda()
    tgt = (0.05 * torch.randn(B_, 1, 192 * T_)).cuda()
    for fo in (True, False):
        ops.FWD_OVERLAP = fo
        outs = []
        for it in range(4):
            poison_free_memory(torch, 1)
            with torch.enable_grad():
                o = m({"mixture": x, "dis_embed": dis}, pad=False)["output"]
            if it % 2 == 1:
                loss, _ = SnrlpLossFn.apply(o, tgt, 100.0); loss.backward()
            outs.append(o.detach().clone())
        torch.cuda.synchronize()
        print(B_, T_, "fwd overlap", fo, [bool(torch.equal(outs[0], t)) for t in outs[1:]], [float((outs[0]-t).abs().max()) for t in outs[1:]], bool(torch.isfinite(outs[0]).all()))
