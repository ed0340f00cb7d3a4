"""Developer probe (GPU box): PipelinedForward against the eager forward, step by step."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import test_gpu_graph as T

m = T._model(26)
d, n = T._inputs(1, 16, 24, 550)
d2, n2 = T._inputs(1, 16, 24, 551)
rest = (d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])
rest2 = (d2["mvs0"], d2["mvs1"], d2["pms"], d2["rms"], d2["ufs"])
xs = d["x"] * 2.0 ** 13
seq = [(d["x"], rest, n), (xs, rest, n), (d2["x"], rest2, n2), (d["x"], rest, n), (xs, rest, n)]
warnings.simplefilter("ignore")
with torch.no_grad():
    pipe = m.capture_pipelined(d["x"], *rest, gumbel_uniform=n)
    print("out ptrs", pipe.caps[0].out.data_ptr(), pipe.caps[1].out.data_ptr(), "L1 ptrs", pipe.caps[0].L1_fea.data_ptr(), pipe.caps[1].L1_fea.data_ptr())
    want = []
    for x, r, nn in seq:
        o, l1 = m(x, *r, gumbel_uniform=nn)
        m.finish_range_guard()
        want.append((o.clone(), l1.clone(), bool(m.last_range["fallback"])))
    print([w[2] for w in want])
    held = []
    for k, (x, r, nn) in enumerate(seq):
        held.append(pipe.submit(x, *r, gumbel_uniform=nn))
        if k:
            torch.cuda.synchronize()
            print(k - 1, "out diff", float((held[k - 1][0] - want[k - 1][0]).abs().max()), "L1 diff", float((held[k - 1][1] - want[k - 1][1]).abs().max()),
                  pipe.caps[(k - 1) & 1].last_range_seen)
    pipe.drain()
    torch.cuda.synchronize()
    k = len(seq)
    print(k - 1, "out diff", float((held[k - 1][0] - want[k - 1][0]).abs().max()), "L1 diff", float((held[k - 1][1] - want[k - 1][1]).abs().max()), pipe.last_range)
