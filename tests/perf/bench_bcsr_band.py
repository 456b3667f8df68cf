"""BASELINE config C4 (BCSR 4x4, 2^18 block-rows x 16 blocks): the block-band plan (kernels/bcsr_band.hxx) next to the shipped
MFMA kernel -- plan build time, every kernel shape, band heights and cuts.  BB_HB=4096,2048 BB_CHUNKS=0,128 select the sweep."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from oracle import oracle as O


def per_launch_ms(fn, iters=30, warm=3, rounds=5):
    for _ in range(warm): fn()
    best = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / iters)
    return float(np.median(best))


log2_nbr = int(sys.argv[1]) if len(sys.argv) > 1 else 18
nbr, per = 1 << log2_nbr, 16
boff, bcols, bvals = G.uniform_bcsr(nbr, nbr, per)
xh = G.uniform_distribution_int(nbr * 4)
want = O.bcsr_spmv_f32(4, 4, nbr * 4, boff, bcols, bvals, xh)
b = S.BCSR(4, 4, nbr * 4, nbr * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
x = torch.from_numpy(xh).cuda(); y = torch.empty(nbr * 4, device="cuda")
nb = bcols.size
abytes = nb * 68 + (nbr + 1) * 4 + nbr * 4 * 4 + nbr * 4 * 4
out = {"workload": f"BCSR 4x4, {nbr} block-rows x {per} blocks", "algorithmic_bytes": abytes, "rows": []}
ms = per_launch_ms(lambda: S.bcsr_thread_mapped(b, x, y, mfma=1))
print(f"bcsr4x4_mfma_spmv (shipped)                     {ms*1e3:7.1f} us  frac {abytes/ms/1e6/8000:.3f}", file=sys.stderr)
out["mfma_us"] = round(ms * 1e3, 2)
ms = per_launch_ms(lambda: S.bcsr_thread_mapped(b, x, y, mfma="merge_path"))
ok = bool(np.array_equal(y.cpu().numpy(), want))
print(f"bcsr4x4_mfma_merge_path (one-shot, balanced)        {ms*1e3:7.1f} us  frac {abytes/ms/1e6/8000:.3f} exact={ok}", file=sys.stderr)
out["merge_path_us"] = round(ms * 1e3, 2)
hbs = [int(t) for t in os.environ.get("BB_HB", "0").split(",")]
cuts = [int(t) for t in os.environ.get("BB_CHUNKS", "0").split(",")]
for hb in hbs:
    for cut in cuts:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan = S.BCSRBandPlan(b, band_block_rows=hb, target_chunks=cut)
        torch.cuda.synchronize(); build_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter(); p2 = S.BCSRBandPlan(b, band_block_rows=hb, target_chunks=cut); torch.cuda.synchronize()
        build2_ms = (time.perf_counter() - t0) * 1e3; p2.close()
        if os.environ.get("BB_NO_TUNE"):   # (counter passes: the default shape only)
            times = {}
            best = (plan.waves, plan.unroll, plan.nt)
        else:
            times = plan.tune(20)
            best = min(times, key=times.get)
        ms = per_launch_ms(lambda: plan.spmv(x, y))
        ok = bool(np.array_equal(y.cpu().numpy(), want))
        a_ms = per_launch_ms(lambda: plan.spmv_stage(0, x, y))
        row = {"HB": plan.HB, "bands": plan.num_bands, "chunks": plan.num_chunks, "partials": plan.num_partials, "build_ms_first": round(build_ms, 2),
               "build_ms": round(build2_ms, 2), "best_shape": best, "us": round(ms * 1e3, 2), "accumulate_only_us": round(a_ms * 1e3, 2),
               "frac": round(abytes / ms / 1e6 / 8000, 4), "bit_exact": ok, "shapes_us": {str(k): round(v * 1e3, 1) for k, v in times.items()}}
        out["rows"].append(row)
        print(f"block-band HB={plan.HB:5d} bands={plan.num_bands:4d} chunks={plan.num_chunks:4d} best={best}  {ms*1e3:7.1f} us (A only {a_ms*1e3:6.1f})  "
              f"frac {row['frac']:.3f} exact={ok} build {build2_ms:.2f} ms", file=sys.stderr)
        print("    " + "  ".join(f"{k}:{v*1e3:.1f}" for k, v in times.items()), file=sys.stderr)
        plan.close()
print(json.dumps(out))
