"""Developer check: overlapped forward vs plain launch order, bit-identical outputs at odd geometries (partial tiles, tiles that
straddle batch entries, short last slabs) on the big-family golden model."""
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import torch, numpy as np
from conftest import load_golden, golden_state_dict
import sound_bubble_amd as sb
from sound_bubble_amd import ops
rec, params, fl = load_golden("tiny_big")
m = sb.NetDisEmbd3(**params); m.load_state_dict(golden_state_dict(rec, torch), strict=True); m = m.cuda().eval()
ops.OVERLAP_MIN_FILL = 0.0
for B_, T_ in ((3, 200), (1, 129), (5, 137), (7, 333)):
    torch.manual_seed(B_ * 1000 + T_)
    x = (0.1 * torch.randn(B_, 6, 192 * T_ + 96)).cuda()
    dis = torch.from_numpy(rec["dis_embed"][:1]).cuda().expand(B_, -1).contiguous()
    outs = []
    for ov in (False, True):
        ops.FWD_OVERLAP = ov
        with torch.no_grad():
            outs.append(m({"mixture": x, "dis_embed": dis}, pad=False)["output"].clone())
    torch.cuda.synchronize(); ops.check_sched_status()
    print(B_, T_, "bit-identical:", bool(torch.equal(outs[0], outs[1])), "overlap eligible:", ops.can_overlap_fwd(B_, T_, 145, 32, False, x.device))
