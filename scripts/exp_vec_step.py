"""per-step latency of the few-sequence vector kernel vs the 16-sequence tile kernel (one sequence, long walk)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sound_bubble_amd import ops
torch.manual_seed(0)
for C in (16, 32):
    for nseq, S in ((1, 14500), (1, 145), (64, 1450), (128, 1450)):
        lstm = torch.nn.LSTM(C, 64, 1, batch_first=True, bidirectional=True)
        d = lambda t: t.detach().float().cuda().contiguous()
        dirs = [(d(lstm.weight_ih_l0), d(lstm.weight_hh_l0), d(lstm.bias_ih_l0), d(lstm.bias_hh_l0)),
                (d(lstm.weight_ih_l0_reverse), d(lstm.weight_hh_l0_reverse), d(lstm.bias_ih_l0_reverse), d(lstm.bias_hh_l0_reverse))]
        x = torch.randn(nseq * S, C).cuda()
        g, b = torch.ones(C).cuda(), torch.zeros(C).cuda()
        geom = ops.Geom.intra(nseq, S)
        res = {}
        for vec in (True, False):
            ops.VEC_LSTM = vec
            for _ in range(3):
                ops.lstm_fwd(x, g, b, dirs, geom)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20 if S < 1000 else 5
            e0.record()
            for _ in range(n):
                ops.lstm_fwd(x, g, b, dirs, geom)
            e1.record()
            torch.cuda.synchronize()
            res[vec] = e0.elapsed_time(e1) / n * 1e3
        ops.VEC_LSTM = True
        print(f"C={C} nseq={nseq} S={S}: vector {res[True]:9.1f} us ({res[True] / S:.3f} us/step)   tile {res[False]:9.1f} us ({res[False] / S:.3f} us/step)")
