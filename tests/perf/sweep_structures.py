"""Load-balance robustness sweep (the reference's evaluation runs its schedules over 4 830 SuiteSparse
matrices, plots/data/*.csv; none of them is available here): the tuned schedules + the reference's own
HIP backend on synthetic matrices of very different row-length / column structure, all ~2^24 nnz."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S, _lib
from oracle import oracle as O

def ev(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))

def chunked(deg, cols, window):
    rows = deg.size; nnz = int(deg.sum())
    chunks = max(1, nnz >> 24); parts = []
    b = np.linspace(0, rows, chunks + 1).astype(np.int64)
    for a0, a1 in zip(b[:-1], b[1:]):
        parts.append(G.csr_from_degrees(deg[a0:a1], cols, 1, int(a0), True, window))
    off = np.concatenate([[0]] + [p[0][1:].astype(np.int64) + sum(int(q[0][-1]) for q in parts[:i]) for i, p in enumerate(parts)]).astype(np.int32)
    return off, np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])

def scale_free(deg, cols, skew=3.0, seed=11):
    """Graph-like adjacency: power-law row lengths AND power-law column popularity (hub vertices): column id =
    perm[floor(cols * u^skew)], u uniform -- half of all nonzeros fall on cols / 2^skew columns.  Columns sorted
    inside a row, repeats kept (a multigraph; SpMV semantics do not care)."""
    r = np.random.default_rng(seed)
    rows, nnz = deg.size, int(deg.sum())
    off = np.zeros(rows + 1, np.int64); np.cumsum(deg, out=off[1:])
    hub = r.permutation(cols).astype(np.int64)
    col = hub[np.minimum((cols * r.random(nnz) ** skew).astype(np.int64), cols - 1)]
    key = (np.repeat(np.arange(rows, dtype=np.int64), deg) << 32) | col
    key.sort()
    idx = (key & 0xFFFFFFFF).astype(np.int32)
    val = ((r.integers(1, 9, nnz)).astype(np.float32) / np.float32(8))
    return off.astype(np.int32), idx, val

N = 1 << 24
rng = np.random.default_rng(1)
cases = {}
cases["power-law rows, uniform cols (C2)"] = (G.powerlaw_degrees(1 << 20, N), 1 << 20, None)
cases["power-law rows, banded cols (w=8192)"] = (G.powerlaw_degrees(1 << 20, N), 1 << 20, 8192)
cases["power-law rows, consecutive cols (runs)"] = (G.powerlaw_degrees(1 << 20, N), 1 << 20, -1)
cases["power-law rows, power-law cols (scale-free graph)"] = (G.powerlaw_degrees(1 << 20, N), 1 << 20, "scale_free")
cases["uniform degree 16, uniform cols"] = (np.full(1 << 20, 16, np.int64), 1 << 20, None)
cases["uniform degree 16, banded (FEM-like, w=64)"] = (np.full(1 << 20, 16, np.int64), 1 << 20, 64)
cases["short rows: degree 1-3, 8M rows"] = (rng.integers(1, 4, 1 << 23).astype(np.int64), 1 << 23, None)
d = np.ones(1 << 18, np.int64); d[rng.choice(1 << 18, 64, replace=False)] = 1 << 18; d = (d * (N / d.sum())).astype(np.int64).clip(1, 1 << 18)
cases["extreme skew: 64 rows hold ~all nnz"] = (d, 1 << 18, None)
d = np.where(np.arange(1 << 21) % 4 == 0, 32, 0).astype(np.int64)
cases["75 % empty rows, degree 32 otherwise"] = (d, 1 << 21, None)

# round 5: closer proxies of real graphs -- R-MAT / Kronecker (Graph500 a, b, c = 0.57, 0.19, 0.19), 2^20 vertices, 16 edges per
# vertex: labels scattered (what Graph500 prescribes), the generator's own order, and relabelled by degree (locality restored)
for relabel in ("random", "none", "degree"):
    cases[f"R-MAT scale 20 x 16, labels: {relabel}"] = (None, 1 << 20, "rmat:" + relabel)

so = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "_ref", "libloops_ref_gpu.so")
R = _lib.load_shared(so) if os.path.exists(so) else None
only = [a for a in sys.argv[1:] if not a.startswith("-")]   # optional: substrings of the case names to run
out = {}
for name, (deg, cols, window) in cases.items():
    if only and not any(o in name for o in only):
        continue
    if isinstance(window, str) and window.startswith("rmat:"):
        off, idx, val = G.rmat_csr(20, 16, relabel=window[5:])
        deg = np.diff(off.astype(np.int64))
    else:
        off, idx, val = scale_free(deg, cols) if isinstance(window, str) else chunked(deg, cols, window)
    rows, nnz = deg.size, int(off[-1])
    xh = G.uniform_distribution_int(cols)
    ref = O.spmv_f32(off, idx, val, xh, omp=True)
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    x = torch.from_numpy(xh).cuda(); y = torch.empty(rows, device="cuda")
    abytes = nnz * 8 + (rows + 1) * 4 + rows * 4 + cols * 4
    row = {"rows": rows, "cols": cols, "nnz": nnz, "max_degree": int(deg.max())}
    tile, _ = S.autotune_merge_path(csr, x, 10)   # launch-box autotuner: the tile shape of the held plan
    row["tile"] = tile
    plan = S.MergePathPlan(csr, tile)
    # round 4: the autotuner over shapes AND kernel variants (the phased-gather twins of 512x8 / 256x16, where the plan is not
    # self-completing): what it picks on this structure and what the phased kernel costs / buys there
    vtile, variant, table = S.autotune_merge_path_variants(csr, x, 10)
    plain_best = min(v for k, v in table.items() if "+" not in k)
    phased_best = min((v for k, v in table.items() if "+" in k), default=None)
    row["structural_guess_scattered"] = S.columns_look_scattered(csr)   # what the plan-less C++ wrapper goes by
    row["variant_autotuner"] = {"pick": vtile + ("+phased" if variant else ""), "best_plain_us": round(plain_best * 1e3, 1),
                                "best_phased_us": None if phased_best is None else round(phased_best * 1e3, 1),
                                "phased_over_plain": None if phased_best is None else round(phased_best / plain_best, 3)}
    vplan = S.MergePathPlan(csr, vtile)
    ms = ev(lambda: S.merge_path_flat(csr, x, y, plan=vplan, variant=variant))
    row["merge_path_flat_autotuned_variant"] = {"us": round(ms * 1e3, 1), "GBps": round(abytes / ms / 1e6),
                                                "bit_exact": bool(np.array_equal(y.cpu().numpy(), ref))}
    vplan.close()
    for label, fn in (("merge_path_flat", lambda: S.merge_path_flat(csr, x, y, plan=plan)),
                      ("work_oriented", lambda: S.spmv("work_oriented", csr, x, y)),
                      ("group_mapped", lambda: S.spmv("group_mapped", csr, x, y)),
                      ("flat_partitioned", lambda: S.spmv("flat_partitioned", csr, x, y)),
                      ("thread_mapped", lambda: S.spmv("thread_mapped", csr, x, y))):
        ms = ev(fn, iters=10 if label == "thread_mapped" else 20)
        ok = bool(np.array_equal(y.cpu().numpy(), ref))
        row[label] = {"us": round(ms * 1e3, 1), "GBps": round(abytes / ms / 1e6), "bit_exact": ok}
    cb = S.RowBandPlan(csr)   # y accumulators in LDS, column-sorted gathers (round 5)
    cb.tune(5)
    ms = ev(lambda: cb.spmv(x, y))
    row["row_band"] = {"us": round(ms * 1e3, 1), "GBps": round(abytes / ms / 1e6), "bands": cb.num_bands, "band_rows": cb.H, "wavefronts": cb.waves,
                       "bit_exact": bool(np.array_equal(y.cpu().numpy(), ref))}
    cb.close()
    pb = S.PanelBinnedPlan(csr)     # x panels in LDS, no memory gather (round 3)
    ms = ev(lambda: pb.spmv(x, y))
    row["panel_binned"] = {"us": round(ms * 1e3, 1), "GBps": round(abytes / ms / 1e6), "subband_rows": pb.Hw,
                           "bit_exact": bool(np.array_equal(y.cpu().numpy(), ref))}
    pb.close()
    sp = S.SpmvPlan(csr, allow_copy=True, measure=True, repeats=10)   # what a caller holding ONE handle gets (round 3)
    ms = ev(lambda: sp.spmv(x, y))
    row["held_spmv_plan"] = {"us": round(ms * 1e3, 1), "GBps": round(abytes / ms / 1e6), "layout": sp.layout, "tile": sp.tile,
                             "bit_exact": bool(np.array_equal(y.cpu().numpy(), ref))}
    sp.close()
    sp = S.SpmvPlan(csr, allow_copy=False, measure=True, repeats=10)  # the same WITHOUT a copy: tile shape + kernel variant only (round 4)
    ms = ev(lambda: sp.spmv(x, y))
    row["held_spmv_plan_no_copy"] = {"us": round(ms * 1e3, 1), "GBps": round(abytes / ms / 1e6), "tile": sp.info["tile"],
                                     "bit_exact": bool(np.array_equal(y.cpu().numpy(), ref))}
    sp.close()
    if R is not None:
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        for kind, label in ((2, "ref_hip_merge_path"), (1, "ref_hip_work_oriented"), (0, "ref_hip_thread_mapped")):
            yr = np.zeros(rows, np.float32); ms = C.c_float()
            R.refgpu_spmv_f32(kind, C.c_long(rows), C.c_long(cols), C.c_long(nnz), p(off), p(idx), p(val), p(xh), p(yr), 3, C.byref(ms))
            row[label] = {"us": round(ms.value * 1e3, 1)}
    out[name] = row
    print(f"{name:46s} nnz {nnz:9d} maxdeg {int(deg.max()):7d} tile {tile} | " + " ".join(f"{k} {v['us']:8.1f}us" + ("" if v.get('bit_exact', True) else "(!)") for k, v in row.items() if isinstance(v, dict) and "us" in v)
          + f" | variants: {row['variant_autotuner']} guess_scattered={row['structural_guess_scattered']}", file=sys.stderr, flush=True)
    del csr, x, y, plan
    print(json.dumps({name: row}), flush=True)   # one JSON object per case: a run cut short keeps what it measured
