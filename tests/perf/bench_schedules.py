"""Times every schedule's SpMV (tuned C-ABI path and the reference-shaped schedule-API kernels)
on the C2 workload and on a C3-sized scale-free stand-in; prints a table + JSON (stderr/stdout).
Also times the reference's own HIP kernels on the same GPU when oracle/_ref is present."""
import argparse, json, os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S, _lib

def ev(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))

ap = argparse.ArgumentParser()
ap.add_argument("--log2-rows", type=int, default=20)
ap.add_argument("--log2-nnz", type=int, default=24)
ap.add_argument("--window", type=int, default=0)
ap.add_argument("--ref-gpu", action="store_true")
ap.add_argument("--tag", default="c2")
ap.add_argument("--rows", type=int, default=0, help="exact row count (overrides --log2-rows)")
ap.add_argument("--nnz", type=int, default=0, help="exact nonzero count (overrides --log2-nnz)")
ap.add_argument("--cap", type=int, default=1 << 14)
ap.add_argument("--tuned-only", action="store_true", help="skip the schedule-API (atomic) kernels: profiling runs")
ap.add_argument("--mtx", default="", help="Matrix-Market coordinate file to run instead of a generated matrix (BASELINE C3: "
                "SuiteSparse LAW/indochina-2004, datasets/suitesparse.txt:2052 in the reference; not shipped). Loaded as the "
                "reference's loader does (container/market.hxx:100-289): pattern entries = 1, symmetric files mirrored, "
                "duplicates kept, rows sorted by (row, column)")
ap.add_argument("--rmat", default="", help="SCALE,EDGE_FACTOR,RELABEL (none|random|degree): an R-MAT graph (generate.rmat_csr, Graph500 "
                "a, b, c) instead of the hashed-column generator, e.g. 23,23,none = 8.4 M vertices / 193 M edges, the size of C3")
args = ap.parse_args()
if args.rmat:
    sc, ef, rl = args.rmat.split(",")
    off, idx, val = G.rmat_csr(int(sc), int(ef), relabel=rl)
    rows = cols = 1 << int(sc)
    nnz = int(off[-1])
    deg = np.diff(off.astype(np.int64))
    args.tag = f"rmat scale {sc} x {ef}, labels {rl}"
elif args.mtx:
    import scipy.io
    m = scipy.io.mmread(args.mtx).tocsr()  # mmread mirrors symmetric files; pattern -> 1
    m.sort_indices()
    rows, cols, nnz = m.shape[0], m.shape[1], int(m.nnz)
    off, idx, val = m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float32)
    deg = np.diff(m.indptr)
    args.tag = os.path.basename(args.mtx)
    del m
else:
    rows = cols = args.rows or (1 << args.log2_rows)
    nnz = args.nnz or (1 << args.log2_nnz)
    deg = G.powerlaw_degrees(rows, nnz, cap=args.cap)
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, args.window or None,
                                       hosts=G.host_blocks(cols) if args.window == G.HOST_BLOCKED else None)
xh = G.uniform_distribution_int(cols)
csr = S.CSR.from_numpy(rows, cols, off, idx, val)
x = torch.from_numpy(xh).cuda()
y = torch.empty(rows, device="cuda")
from oracle import oracle as O
ref = O.spmv_f32(off, idx, val, xh, omp=True)
exact_inputs = not args.mtx or bool(np.all(val == np.round(val * 8) / 8) and np.abs(val).max() * deg.max() * 10 < 2 ** 21)
abytes = nnz * 8 + (rows + 1) * 4 + rows * 4 + cols * 4
res = {"workload": f"{args.tag}: {rows} rows, {nnz} nnz, max degree {int(deg.max())}, window={args.window}", "rows": {}}
plan = S.MergePathPlan(csr)
def rec(name, fn, check=True):
    ms = ev(fn)
    got = y.cpu().numpy()
    ok = (bool(np.array_equal(got, ref)) if exact_inputs else bool(np.allclose(got, ref, rtol=1e-5, atol=1e-5))) if check else None
    res["rows"][name] = {"ms": round(ms, 4), "GFLOPs": round(2 * nnz / ms / 1e6, 1), "GBps": round(abytes / ms / 1e6, 1), "bit_exact": ok}
    print(f"{name:42s} {ms*1e3:9.1f} us {2*nnz/ms/1e6:8.1f} GFLOP/s {abytes/ms/1e6:8.1f} GB/s exact={ok}", file=sys.stderr, flush=True)
rec("merge_path_flat (planned, fused+fixup)", lambda: S.merge_path_flat(csr, x, y, plan=plan))
# round 4: tile shape AND kernel variant by measurement (the phased-gather twins), what the structural guess says, and the
# held SpMV plans with / without a re-ordered copy
vtile, variant, table = S.autotune_merge_path_variants(csr, x, 10)
res["variant_autotuner"] = {"pick": vtile + ("+phased" if variant else ""), "ms": {k: round(v, 4) for k, v in table.items()},
                            "structural_guess_scattered": S.columns_look_scattered(csr)}
print(f"variant autotuner: {res['variant_autotuner']}", file=sys.stderr, flush=True)
vplan = S.MergePathPlan(csr, vtile)
rec(f"merge_path_flat (planned, {vtile}{'+phased' if variant else ''}: the autotuner's pick)", lambda: S.merge_path_flat(csr, x, y, plan=vplan, variant=variant))
for allow in (False, True):
    sp = S.SpmvPlan(csr, allow_copy=allow, measure=True, repeats=10)
    rec(f"held SpMV plan, copy {'allowed' if allow else 'not allowed'}: {sp.info['layout']} {sp.info['tile']}", lambda: sp.spmv(x, y))
    sp.close()
for sched in ("merge_path_flat", "work_oriented", "group_mapped") + (() if args.tuned_only else ("thread_mapped", "original", "flat_partitioned")):
    rec(f"tuned {sched}", lambda: S.spmv(sched, csr, x, y))
for sched in () if args.tuned_only else ("merge_path_flat", "work_oriented", "group_mapped", "thread_mapped", "flat_partitioned"):
    rec(f"schedule-API {sched} (incl. y zero-fill)", lambda: S.spmv_schedule_api(sched, csr, x, y))
so = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "_ref", "libloops_ref_gpu.so")
if args.ref_gpu and os.path.exists(so):
    R = _lib.load_shared(so)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for kind, name in ((2, "merge_path_flat"), (0, "thread_mapped"), (1, "work_oriented")):
        yr = np.zeros(rows, np.float32); ms = C.c_float()
        rc = R.refgpu_spmv_f32(kind, C.c_long(rows), C.c_long(cols), C.c_long(nnz), p(off), p(idx), p(val), p(xh), p(yr), 10, C.byref(ms))
        ok = bool(np.allclose(yr, ref, rtol=1e-4, atol=1e-3))
        res["rows"][f"REFERENCE HIP backend {name}"] = {"ms": round(ms.value, 4), "GFLOPs": round(2 * nnz / ms.value / 1e6, 1), "GBps": round(abytes / ms.value / 1e6, 1), "ok": ok, "rc": rc}
        print(f"{'REFERENCE HIP backend ' + name:42s} {ms.value*1e3:9.1f} us {2*nnz/ms.value/1e6:8.1f} GFLOP/s {abytes/ms.value/1e6:8.1f} GB/s ok={ok}", file=sys.stderr, flush=True)
print(json.dumps(res))
