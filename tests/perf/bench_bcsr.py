"""BASELINE config C4: BCSR 4x4, 2^18 block-rows x 16 blocks, bcsr_thread_mapped register path vs the
MFMA path, against the B_bcsr roofline of SURVEY 8(d)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from oracle import oracle as O

def ev(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    return float(np.mean(ts)), float(np.median(ts))

log2_nbr = int(sys.argv[1]) if len(sys.argv) > 1 else 18
nbr, per = 1 << log2_nbr, 16
boff, bcols, bvals = G.uniform_bcsr(nbr, nbr, per)
x = G.uniform_distribution_int(nbr * 4)
want = O.bcsr_spmv_f32(4, 4, nbr * 4, boff, bcols, bvals, x)
b = S.BCSR(4, 4, nbr * 4, nbr * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
xd = torch.from_numpy(x).cuda(); y = torch.empty(nbr * 4, device="cuda")
nb = bcols.size
abytes = nb * (16 * 4 + 4) + (nbr + 1) * 4 + nbr * 4 * 4 + nbr * 4 * 4
flops = 2 * 16 * nb
out = {"workload": f"BCSR 4x4, {nbr} block-rows x {per} blocks ({nb} blocks), fp32", "algorithmic_bytes": abytes, "flops": flops, "rows": {}}
shapes = [(f"MFMA h={h} unroll={u} (groups per wave: automatic)", 100 + 10 * h + u) for h in (1, 2, 4, 8, 16) for u in (1, 2, 4, 8)]
# + 1000 g: a wavefront pipelines through g consecutive groups of 16 / h block-rows (g = 1: the round-1 kernel shape)
shapes += [(f"MFMA h={h} unroll={u} groups/wave={g}", 1000 * g + 100 + 10 * h + u)
           for (h, u) in ((4, 4), (4, 2), (2, 4), (8, 2), (2, 8)) for g in (1, 2, 4, 8, 16, 32)]
only_shapes = False
if os.environ.get("BCSR_SHAPES"):  # e.g. BCSR_SHAPES=128,144: only these tuning shapes (profiling runs)
    keep = {int(t) for t in os.environ["BCSR_SHAPES"].split(",")}
    shapes = [s for s in shapes if s[1] in keep]
    only_shapes = True
for name, mfma in ([] if only_shapes else [("bcsr_thread_mapped (registers)", 0)]) + [("bcsr_thread_mapped (MFMA 4x4x1, automatic shape)", 1)] + shapes:
    avg, med = ev(lambda: S.bcsr_thread_mapped(b, xd, y, mfma=mfma))
    ok = bool(np.array_equal(y.cpu().numpy(), want))
    out["rows"][name] = {"avg_ms": round(avg, 5), "median_ms": round(med, 5), "GFLOPs": round(flops / avg / 1e6, 1),
                         "GBps": round(abytes / avg / 1e6, 1), "frac_of_8TBps": round(abytes / avg / 1e6 / 8000, 4), "bit_exact": ok}
    print(f"{name:50s} {avg*1e3:8.1f} us  {flops/avg/1e6:8.1f} GFLOP/s  {abytes/avg/1e6:8.1f} GB/s  frac {abytes/avg/1e6/8000:.3f} exact={ok}", file=sys.stderr)
print(json.dumps(out))
