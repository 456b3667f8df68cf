"""Block-band plan on C4: every kernel shape measured carefully (5 rounds x 50 products each, median and min) -- how much of the
spread between runs is the shape and how much is noise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
nbr, per = 1 << 18, 16
boff, bcols, bvals = G.uniform_bcsr(nbr, nbr, per)
xh = G.uniform_distribution_int(nbr * 4)
b = S.BCSR(4, 4, nbr * 4, nbr * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
x = torch.from_numpy(xh).cuda(); y = torch.empty(nbr * 4, device="cuda")
def rounds(fn, n=5, iters=50):
    out = []
    for _ in range(3): fn()
    for _ in range(n):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(e) / iters * 1e3)
    return out
t = rounds(lambda: S.bcsr_thread_mapped(b, x, y, mfma=1))
print("mfma            median %.1f min %.1f" % (np.median(t), min(t)))
for hb in [int(v) for v in os.environ.get("BB_HB", "0").split(",")]:
    plan = S.BCSRBandPlan(b, band_block_rows=hb)
    for w in (8, 16):
        for u in (1, 2, 4):
            for nt in (0, 1):
                plan.set_shape(w, u, nt)
                t = rounds(lambda: plan.spmv(x, y))
                print("HB %4d (%2d,%d,%d)  median %.1f min %.1f max %.1f" % (plan.HB, w, u, nt, np.median(t), min(t), max(t)))
    plan.close()
