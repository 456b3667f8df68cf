"""Block-band plan on a BCSR with hub block-rows (a few block-rows hold most blocks): what the replicated accumulators buy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from oracle import oracle as O
nbr = nbc = 1 << 17
rng = np.random.default_rng(3)
lens = np.full(nbr, 8, np.int64)
hubs = rng.choice(nbr, size=64, replace=False)
lens[hubs] = 16384                                   # 64 hub block-rows x 16 384 blocks = half of all blocks
boff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
bcols = np.concatenate([np.sort(rng.choice(nbc, size=int(n), replace=False)) for n in lens]).astype(np.int32)
bvals = (rng.integers(1, 9, size=bcols.size * 16) / 8.0).astype(np.float32)
xh = G.uniform_distribution_int(nbc * 4)
want = O.bcsr_spmv_f32(4, 4, nbr * 4, boff, bcols, bvals, xh)
b = S.BCSR(4, 4, nbr * 4, nbc * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
x = torch.from_numpy(xh).cuda(); y = torch.empty(nbr * 4, device="cuda")
def ms(fn, iters=30):
    for _ in range(3): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3
print("blocks", bcols.size, "mfma kernel %.1f us" % ms(lambda: S.bcsr_thread_mapped(b, x, y, mfma=1)))
t = ms(lambda: S.bcsr_thread_mapped(b, x, y, mfma="merge_path"))
S.bcsr_thread_mapped(b, x, y, mfma="merge_path")
print("merge-path one-shot kernel %.1f us exact=%s" % (t, bool(np.array_equal(y.cpu().numpy(), want))))
S.bcsr_thread_mapped(b, x, y, mfma="tuned"); torch.cuda.synchronize()    # (first call: merge-path tiles + the probe of the block-row lengths)
t = ms(lambda: S.bcsr_thread_mapped(b, x, y, mfma="tuned"))
S.bcsr_thread_mapped(b, x, y, mfma="tuned")
print("mode 'tuned' (class %s) %.1f us exact=%s" % (S.bcsr_row_length_class(b), t, bool(np.array_equal(y.cpu().numpy(), want))))
for hb in (0, 4096):
    plan = S.BCSRBandPlan(b, band_block_rows=hb)
    t = ms(lambda: plan.spmv(x, y))
    plan.spmv(x, y)
    print("block-band HB %d bands %d chunks %d partials %d: %.1f us exact=%s hubs per band max %d" % (plan.HB, plan.num_bands, plan.num_chunks, plan.num_partials, t,
          bool(np.array_equal(y.cpu().numpy(), want)), int(plan.arrays()[5][:, 0].max())))
    plan.close()
