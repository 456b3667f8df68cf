"""EXPERIMENT (round 5): phased x gathers in the 4x4 MFMA BCSR kernel on BASELINE C4 (LOOPS_BCSR_PHASED=parts,shift,ticks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from oracle import oracle as O

def batch_ms(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

nbr, per = 1 << 18, 16
boff, bcols, bvals = G.uniform_bcsr(nbr, nbr, per)
x = G.uniform_distribution_int(nbr * 4)
want = O.bcsr_spmv_f32(4, 4, nbr * 4, boff, bcols, bvals, x)
b = S.BCSR(4, 4, nbr * 4, nbr * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
xd = torch.from_numpy(x).cuda(); y = torch.empty(nbr * 4, device="cuda")
for tag, mode in (("automatic (h4 u2)", 1), ("h4 u4", 144)):
    os.environ.pop("LOOPS_BCSR_PHASED", None)
    t = batch_ms(lambda: S.bcsr_thread_mapped(b, xd, y, mfma=mode))
    print(f"{tag:28s} {t*1e3:7.1f} us exact={bool(np.array_equal(y.cpu().numpy(), want))}", flush=True)
for parts in (2, 4, 8):
    shift = 18 - {2: 1, 4: 2, 8: 3}[parts]
    for ticks in (50, 100, 200, 400, 800):
        os.environ["LOOPS_BCSR_PHASED"] = f"{parts},{shift},{ticks}"
        y.fill_(-1)
        t = batch_ms(lambda: S.bcsr_thread_mapped(b, xd, y, mfma=1))
        print(f"phased parts {parts} ticks {ticks:4d}   {t*1e3:7.1f} us exact={bool(np.array_equal(y.cpu().numpy(), want))}", flush=True)
