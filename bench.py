#!/usr/bin/env python
"""bench.py -- merge_path_flat CSR SpMV on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one SpMV y = A x of the workload with inputs resident in HBM and the merge-path
plan (per-workgroup coordinates) prebuilt -- the region the reference times
(algorithms/spmv/merge_path_flat.cuh:121-136: its timer starts after the coordinate pre-pass):
fused merge-tile kernel + carry-out fix-up, and for N > 1 the allgatherv of y.  The time of a
step WITH the coordinate pre-pass is reported next to it (config.ms_per_step_with_prepass).

Workload at N = 1: BASELINE config C2 -- synthetic power-law CSR, 2^20 rows, 2^24 nnz, max
degree 2^14, fp32 (SURVEY 8d generator).  At N > 1 (default, --scaling strong): BASELINE config
C5 -- the 2^24-row / 2^29-nnz matrix of the same generator, STRONG-scaled: contiguous row ranges
balanced by rows + nnz, one per GPU (2^26 nnz each at N = 8), x (64 MB) replicated, allgatherv(y)
over RCCL every step; `value` = 2 * 2^29 flop / step time.  Rank 0 additionally times the SAME
matrix on its one GPU outside the timed region (config.one_gpu_same_matrix), so the ">= 6x at 8 GPUs"
target of BASELINE.md is a ratio inside one record (config.speedup_vs_one_gpu_same_matrix), and the
gathered y is compared bit for bit with the one-GPU y.  --scaling weak keeps the round-1 mode
(N x C2: N * 2^20 rows / N * 2^24 nnz) for context.  At N > 1 a rank holds its shard in a
re-ordered copy (--layout): x is larger than the 4 MB per-XCD L2 there; panel-binned
(include/loops/kernels/panel_binned.hxx: x panels in LDS) or row-band (include/loops/kernels/rowband.hxx:
y accumulators in LDS, column-sorted gathers), whichever a start-up probe finds faster on the worst
rank.  The N = 1 headline runs on the unmodified CSR.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the roofline / cpu_baseline fields.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# dmabuf IPC (the only mode the host driver supports): needed by RCCL's intra-node transport and by the peer mappings of
# the fused-stores exchange; must be in the environment before the HIP runtime is loaded
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured float4 copy


def algorithmic_bytes(rows, cols, nnz, vbytes=4):
    # nnz * (4 + 4) + (rows + 1) * 4 + rows * 4 + cols * 4 for fp32
    return nnz * (4 + vbytes) + (rows + 1) * 4 + rows * vbytes + cols * vbytes


KERNEL_SOURCES = ("include/loops/kernels/merge_path_spmv.hxx", "include/loops/util/wave.hxx")
VARIANT_PHASED = 8  # include/loops_amd.h LOOPS_VARIANT_PHASED: the default kernel with phased x gathers (same CSR, same bits)


def headline_kernel(args):
    """Name prefix (incl. the tile shape's template arguments) of the dominant kernel of the N = 1 run in rocprofv3's tables."""
    tpb, ipt = args.tile.split("x")
    base = "merge_path_spmv_fused_phased" if args.variant == VARIANT_PHASED else "merge_path_spmv_fused"
    return f"{base}<{tpb}, {ipt},"


def kernel_sources_digest():
    """sha256 of the files the headline kernel is compiled from: what ties a committed counter summary to the code that runs."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()


def pmc_summary(args):
    """(summary dict, path, note) of the committed rocprofv3 PMC passes of THIS command (profiles/rNN_c2_pmc_summary_<tile>.json,
    scripts/pmc_c2.sh: separate --pmc runs) -- only when the configuration is the profiled one AND the summary was collected
    at the kernel sources as they are now (`_kernel_sources_sha256`, written by scripts/pmc_summarize.py); otherwise
    (None, None, why)."""
    if args.gpus != 1 or args.window or args.log2_rows != 20 or args.log2_nnz != 24 or args.variant not in (0, VARIANT_PHASED) \
            or args.layout in ("rowband", "panel") or args.scaling == "strong":
        return None, None, "configuration differs from the profiled one (C2, N = 1, unmodified CSR)"
    import glob
    tag = args.tile + ("_phased" if args.variant == VARIANT_PHASED else "")
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_c2_pmc_summary_{tag}.json")), reverse=True)
    if not paths:
        return None, None, f"no committed counter summary for tile {tag}"
    d = json.load(open(paths[0]))
    rel = os.path.relpath(paths[0], ROOT)
    if d.get("_kernel_sources_sha256") != kernel_sources_digest():
        return None, None, (f"{rel} was collected at other kernel sources than the ones loaded now (digest of {', '.join(KERNEL_SOURCES)} "
                            "differs or is absent): re-run scripts/pmc_c2.sh")
    return d, rel, None


def pmc_traffic(args):
    """HBM-side bytes per launch of the dominant kernel from the committed counters: TCC_EA0_RDREQ x 128 B (every fabric read
    of this kernel is a 128-B line: TCC_EA0_RDREQ_32B = 0; FETCH_SIZE tallies them at 64 B on gfx950, MI355X_MICROARCH.md HBM
    section) + WRITE_SIZE.  Returns (bytes, source file, note); bytes is None -- with the reason in note -- when no summary
    matches this configuration and these kernel sources."""
    d, rel, why = pmc_summary(args)
    if d is None:
        return None, None, why
    for k, v in d.items():
        if headline_kernel(args) not in k or not isinstance(v, dict):
            continue
        wr = v.get("WRITE_SIZE", {}).get("mean")
        if "TCC_EA0_RDREQ_sum" in v and wr is not None:
            return int(v["TCC_EA0_RDREQ_sum"]["mean"] * 128 + wr * 1024), rel, None
        if "FETCH_SIZE" in v and wr is not None:
            return int((2 * v["FETCH_SIZE"]["mean"] + wr) * 1024), rel, None
    return None, rel, "the summary holds no fabric-read counters for the headline kernel"


def pmc_bound(args):
    """What the committed counters of the dominant kernel say bounds it (same summary file as pmc_traffic):
    L2 requests per launch, L2 hit rate, average L1 -> L2 round trip, reads in flight per CU, share of its active time the
    vector L1 waits for data, L2 requests per clock and XCD.  None when the configuration differs from the profiled one."""
    d, src, _ = pmc_summary(args)
    if d is None:
        return None
    for k, v in d.items():
        if headline_kernel(args) in k and isinstance(v, dict) and "TCP_TCC_READ_REQ_LATENCY_sum" in v:
            m = {c: x["mean"] for c, x in v.items()}
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            return {"l2_requests_per_launch": int(m["TCC_REQ_sum"]), "l2_hit_rate": round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4),
                    "avg_l1_to_l2_round_trip_clks": round(m["TCP_TCC_READ_REQ_LATENCY_sum"] / m["TCP_TCC_READ_REQ_sum"], 1),
                    "reads_in_flight_per_cu": round(m["TCP_TCC_READ_REQ_LATENCY_sum"] / cyc / 256, 1),
                    "l1_waiting_for_data_frac": round(m["TCP_PENDING_STALL_CYCLES_sum"] / m["TCP_GATE_EN1_sum"], 3),
                    "l2_requests_per_clk_per_xcd": round(m["TCC_REQ_sum"] / 8 / cyc, 2),
                    "reading": "bound by the CU's outstanding-read capacity (~95 in flight) x round-trip latency, not by the L2 request "
                               "path (16 per clk per XCD) nor by HBM bandwidth; calibration: profiles/r02_inflight_calibration.json",
                    "collected_at": d.get("_kernel_build"),
                    # (pmc_summary hands the file out only when the digest of the kernel sources it was collected at equals the digest
                    #  of the sources this process runs: the counters describe THIS kernel even when `collected_at` is an older commit)
                    "digest_matches_head": True, "kernel_sources_sha256": d.get("_kernel_sources_sha256"),
                    "source": src}
    return None


def full_matrix_on_device(G, S, torch, degrees, cols, chunks=8):
    """The whole synthetic matrix as one device CSR, generated and uploaded in row chunks (host memory stays at one
    chunk: C5 is 4.3 GB of indices + values)."""
    rows = degrees.size
    off = np.zeros(rows + 1, np.int64)
    np.cumsum(degrees, out=off[1:])
    nnz = int(off[-1])
    idx_d = torch.empty(nnz, dtype=torch.int32, device="cuda")
    val_d = torch.empty(nnz, dtype=torch.float32, device="cuda")
    cut = np.linspace(0, rows, chunks + 1).astype(np.int64)
    for a, b in zip(cut[:-1], cut[1:]):
        _, i, v = G.csr_from_degrees(degrees[a:b], cols, seed=1, row_begin=int(a))
        idx_d[int(off[a]):int(off[b])].copy_(torch.from_numpy(i))
        val_d[int(off[a]):int(off[b])].copy_(torch.from_numpy(v))
    return S.CSR(rows, cols, torch.from_numpy(off.astype(np.int32)).cuda(), idx_d, val_d)


class Watchdog:
    """Host-side deadline around the parts of an N > 1 run that could hang (a candidate exchange of the start-up probe
    that never completes on some rank).  A hung collective cannot be cancelled from inside the process, so the run is built
    measure-first: the step is timed with the safe exchange BEFORE any other candidate is tried, and when a deadline
    passes every rank's own watchdog ends its process -- rank 0 after printing the record it already holds, with the
    reason in config.watchdog.  A hung candidate therefore costs that candidate, not the run."""

    def __init__(self, rank):
        import threading
        self.rank, self.what, self.deadline, self.fallback = rank, None, None, None
        self.record_printed = False  # the record is out: a deadline missed afterwards (teardown) must not fail the run
        self._lock = threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def arm(self, what, seconds):
        with self._lock:
            self.what, self.deadline = what, time.monotonic() + seconds

    def disarm(self):
        with self._lock:
            self.what, self.deadline = None, None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self._lock:
                expired = self.deadline is not None and time.monotonic() > self.deadline
                what = self.what
            if expired:
                print(f"[rank {self.rank}] watchdog: '{what}' did not finish in time", file=sys.stderr, flush=True)
                if self.rank == 0 and self.fallback is not None:
                    rec = self.fallback(f"'{what}' did not finish within its deadline; reporting the measurement taken before it")
                    if rec is not None:
                        print(json.dumps(rec), flush=True)
                        os._exit(0)
                os._exit(0 if self.rank != 0 or self.record_printed else 3)


def timed_ms(torch, fn, iters, warm=3):
    """ms per call: `iters` back-to-back calls between one pair of HIP events on the launch stream."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def one_gpu_same_matrix(G, S, torch, degrees, cols, x, y_gathered, iters=20):
    """BASELINE C5's denominator: the SAME matrix on ONE GPU (rank 0's, outside the timed region) -- the planned
    merge_path_flat SpMV on the unmodified CSR and the product of the held SpMV plan (layout by measurement),
    each y compared bit for bit with the vector the N ranks gathered (SURVEY 8e parity)."""
    t0 = time.time()
    csr = full_matrix_on_device(G, S, torch, degrees, cols)
    gen_s = time.time() - t0
    y = torch.empty(csr.rows, dtype=torch.float32, device="cuda")
    plan = S.MergePathPlan(csr, "512x8")
    ms_csr = timed_ms(torch, lambda: S.merge_path_flat(csr, x, y, plan=plan), iters)
    eq_csr = bool(torch.equal(y, y_gathered))
    out = {"workload": f"{csr.rows} rows / {csr.nnzs} nnz on rank 0's GPU alone (x {cols * 4 >> 20} MB)",
           "csr_ms_per_spmv": round(ms_csr, 5), "csr_equals_gathered_y_bit_for_bit": eq_csr, "generate_upload_seconds": round(gen_s, 1)}
    plan.close()
    try:  # what a caller gets by default from a held plan: loops_spmv_plan_* picks tile shape and layout by measurement
        sp = S.SpmvPlan(csr, allow_copy=True, measure=True, repeats=5)
        ms_p = timed_ms(torch, lambda: sp.spmv(x, y), iters)
        out.update({"planned_ms_per_spmv": round(ms_p, 5), "planned_choice": sp.info,
                    "planned_equals_gathered_y_bit_for_bit": bool(torch.equal(y, y_gathered))})
        sp.close()
    except Exception as e:  # noqa: BLE001
        out["planned_error"] = f"{type(e).__name__}: {e}"
    out["best_ms_per_spmv"] = min(v for k, v in out.items() if k.endswith("_ms_per_spmv"))
    out["GFLOPs"] = round(2.0 * csr.nnzs / out["best_ms_per_spmv"] / 1e6, 2)  # the N = 1 `value` of THIS matrix (bench.py --gpus 1 runs C2)
    return out


def context_c4_bcsr(G, S, O, torch, iters=50):
    """BASELINE config C4 at full size next to the headline (context line of the N = 1 record): BCSR 4x4, 2^18 block
    rows x 16 blocks, bcsr_thread_mapped with the MFMA block inner product, bit-exact against the oracle."""
    nbr, per = 1 << 18, 16
    boff, bcols, bvals = G.uniform_bcsr(nbr, nbr, per)
    xh = G.uniform_distribution_int(nbr * 4)
    b = S.BCSR(4, 4, nbr * 4, nbr * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
    x, y = torch.from_numpy(xh).cuda(), torch.empty(nbr * 4, device="cuda")
    nb = int(bcols.size)
    abytes = nb * (16 * 4 + 4) + (nbr + 1) * 4 + nbr * 4 * 4 + nbr * 4 * 4  # SURVEY 8d B_bcsr: 294 649 860
    out = {"workload": f"BCSR 4x4, {nbr} block-rows x {per} blocks, fp32 (BASELINE configs[3])", "algorithmic_bytes": abytes}
    for name, mode in (("mfma", 1), ("thread_per_block_row", 0)):
        ms = timed_ms(torch, lambda: S.bcsr_thread_mapped(b, x, y, mfma=mode), iters)
        out[name] = {"avg_launch_ms": round(ms, 5), "GFLOPs": round(2 * 16 * nb / ms / 1e6, 1), "achieved_GBps": round(abytes / ms / 1e6, 1),
                     "frac": round(abytes / ms / 1e6 / HBM_PEAK_GBPS, 4)}
    S.bcsr_thread_mapped(b, x, y, mfma=1)
    out["mfma"]["kernel"] = "loops::kernels::bcsr4x4_mfma_spmv"
    want = O.bcsr_spmv_f32(4, 4, nbr * 4, boff, bcols, bvals, xh)
    out["parity_vs_oracle_bit_exact"] = bool(np.array_equal(y.cpu().numpy(), want))
    # the held plan for this matrix: the block-band copy (kernels/bcsr_band.hxx) -- blocks sorted by block column inside bands whose
    # row sums live in LDS, MFMA block products; what it costs to build and after how many products it has paid for itself
    try:
        S.BCSRBandPlan(b).close()                                       # (first build in the process: allocator warm-up)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan = S.BCSRBandPlan(b)
        torch.cuda.synchronize()
        build_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        shapes = plan.tune(10)
        tune_ms = (time.perf_counter() - t0) * 1e3
        ms = timed_ms(torch, lambda: plan.spmv(x, y), iters)
        y.fill_(-1.0)
        plan.spmv(x, y)
        saved = out["mfma"]["avg_launch_ms"] - ms
        out["block_band_plan"] = {
            "kernel": "loops::kernels::bcsr_band::bcsr_band_accumulate", "avg_launch_ms": round(ms, 5), "GFLOPs": round(2 * 16 * nb / ms / 1e6, 1),
            "achieved_GBps": round(abytes / ms / 1e6, 1), "frac": round(abytes / ms / 1e6 / HBM_PEAK_GBPS, 4),
            "band_block_rows": plan.HB, "bands": plan.num_bands, "chunks": plan.num_chunks, "partial_vectors": plan.num_partials,
            "shape": {"waves": plan.waves, "steps_per_batch": plan.unroll, "non_temporal": plan.nt},
            "plan_build_ms": round(build_ms, 3), "plan_tune_ms": round(tune_ms, 3),
            "break_even_products": (int(np.ceil(build_ms / saved)) if saved > 0 else None),
            "break_even_products_incl_tune": (int(np.ceil((build_ms + tune_ms) / saved)) if saved > 0 else None),
            "parity_vs_oracle_bit_exact": bool(np.array_equal(y.cpu().numpy(), want)),
            "note": "held plan (a re-ordered copy of the blocks); the one-shot bcsr_thread_mapped<4, 4> wrapper launches the MFMA kernel above"}
        plan.close()
    except Exception as e:  # noqa: BLE001
        out["block_band_plan"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def context_c3_standins(G, S, O, torch, iters=10):
    """BASELINE config C3 next to the headline (context): `group_mapped` vs `work_oriented` (+ merge_path_flat) on generated
    stand-ins of indochina-2004's exact shape -- 7 414 866 rows / 194 109 311 nnz; the SuiteSparse file is not shipped
    (datasets/suitesparse.txt:2052 in the reference): scale-free degrees with uniformly random columns (no locality: a lower
    bound for a crawl-ordered web graph), with columns in a 65 536-wide band, and host-blocked (how LAW graphs are laid out).  Whole calls through loops_spmv_csr_f32,
    bit-exact against the oracle."""
    rows = cols = 7_414_866
    nnz = 194_109_311
    deg = G.powerlaw_degrees(rows, nnz)
    xh = G.uniform_distribution_int(cols)
    x = torch.from_numpy(xh).cuda()
    abytes = algorithmic_bytes(rows, cols, nnz)
    out = {"shape": f"{rows} rows / {nnz} nnz (LAW/indochina-2004's), fp32", "algorithmic_bytes": abytes,
           "note": "generated stand-ins: the SuiteSparse file is not available offline; tests/perf/bench_schedules.py --mtx PATH runs the real one"}
    shape_rows, shape_nnz = rows, nnz
    for tag, window in (("uniform_columns", None), ("band_65536", 65536), ("host_blocked", G.HOST_BLOCKED), ("rmat_2e23_x23_generator_order", "rmat")):
        # host_blocked: the locality class LAW graphs belong to -- consecutive ids form hosts of power-law size (generate.host_blocks:
        # >= 256 ids, Pareto 1.1, <= 2^17), 3 of 4 links stay inside the row's host, the rest go anywhere
        # rmat (round 5): a Graph500 R-MAT graph of the nearest power-of-two size (2^23 vertices x 23 edges = 192.9 M) in the generator's
        # own order -- hub vertices at the low ids, as a crawl leaves them: the stand-in on which `group_mapped` falls behind
        # `work_oriented` the way the reference's published C3 row does (11.87 against 2.33 ms on its GPU, plots/data/*.csv)
        if window == "rmat":
            off, idx, val = G.rmat_csr(23, 23, relabel="none")
            rows = cols = 1 << 23
            nnz = int(off[-1])
            xh = G.uniform_distribution_int(cols)
            x = torch.from_numpy(xh).cuda()
            abytes = algorithmic_bytes(rows, cols, nnz)
        else:
            off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window, hosts=G.host_blocks(cols) if window == G.HOST_BLOCKED else None)
        csr = S.CSR.from_numpy(rows, cols, off, idx, val)
        ref = O.spmv_f32(off, idx, val, xh, omp=True)
        y = torch.empty(rows, device="cuda")
        res = {}
        # merge_path_flat runs over a held 256 x 8 plan here: the headline's kernel symbol (merge_path_spmv_fused<512, 8>) must
        # stay exclusive to the C2 matrix in this process, so that rocprofv3's per-kernel average of this command is the headline's
        mplan = S.MergePathPlan(csr, "256x8")
        # (and the phased-gather twin over 256 x 16 tiles -- 32 parts at this |x| -- a measured choice only: it gains where the
        # columns are scattered and LOSES where they are local; another template instantiation than the headline's, 512 x 8 / 8 parts)
        pplan = S.MergePathPlan(csr, "256x16")
        runs = {"group_mapped": lambda: S.spmv("group_mapped", csr, x, y), "work_oriented": lambda: S.spmv("work_oriented", csr, x, y),
                "merge_path_flat": lambda: S.merge_path_flat(csr, x, y, plan=mplan),
                "merge_path_flat_phased_gathers": lambda: S.merge_path_flat(csr, x, y, plan=pplan, variant=VARIANT_PHASED)}
        for sched, fn in runs.items():
            ms = timed_ms(torch, fn, iters)
            res[sched] = {"ms_per_spmv": round(ms, 4), "GFLOPs": round(2.0 * nnz / ms / 1e6, 1), "achieved_GBps": round(abytes / ms / 1e6, 1),
                          "frac": round(abytes / ms / 1e6 / HBM_PEAK_GBPS, 4), "parity_vs_oracle_bit_exact": bool(np.array_equal(y.cpu().numpy(), ref))}
        mplan.close()
        pplan.close()
        # what a caller gets by default from a held plan (loops_spmv_plan_*: tile shape + layout picked by measurement)
        sp = S.SpmvPlan(csr, allow_copy=True, measure=True, repeats=5)
        ms = timed_ms(torch, lambda: sp.spmv(x, y), iters)
        res["held_spmv_plan"] = {"ms_per_spmv": round(ms, 4), "GFLOPs": round(2.0 * nnz / ms / 1e6, 1), "achieved_GBps": round(abytes / ms / 1e6, 1),
                                 "frac": round(abytes / ms / 1e6 / HBM_PEAK_GBPS, 4), "choice": sp.info,
                                 "parity_vs_oracle_bit_exact": bool(np.array_equal(y.cpu().numpy(), ref))}
        sp.close()
        if (rows, nnz) != (shape_rows, shape_nnz):
            res["shape"] = f"{rows} rows / {nnz} nnz, fp32"
            res["algorithmic_bytes"] = abytes
        out[tag] = res
        del csr, off, idx, val, y
    return out


def context_schedules(S, torch, csr, x, ref_y, abytes, iters=50):
    """The other tuned schedules on the headline matrix (work_oriented and group_mapped are BASELINE C3's pair):
    whole call through loops_spmv_csr_f32 (work_oriented includes its coordinate pre-pass), bit-exact vs the headline y."""
    y = torch.empty_like(ref_y)
    out = {}
    wplan = S.MergePathPlan(csr, "256x8")  # work_oriented with a held plan: the region the reference's timer brackets
    ms = timed_ms(torch, lambda: S.work_oriented(csr, x, y, plan=wplan), iters)
    out["work_oriented_held_plan"] = {"ms_per_spmv": round(ms, 5), "GFLOPs": round(2.0 * csr.nnzs / ms / 1e6, 1),
                                      "frac": round(abytes / ms / 1e6 / HBM_PEAK_GBPS, 4), "equals_merge_path_y": bool(torch.equal(y, ref_y))}
    for sched in ("work_oriented", "group_mapped"):
        ms = timed_ms(torch, lambda: S.spmv(sched, csr, x, y), iters)
        out[sched] = {"ms_per_spmv": round(ms, 5), "GFLOPs": round(2.0 * csr.nnzs / ms / 1e6, 1),
                      "frac": round(abytes / ms / 1e6 / HBM_PEAK_GBPS, 4), "equals_merge_path_y": bool(torch.equal(y, ref_y))}
    return out


def self_launch(n):
    """`python bench.py --gpus N ...` started WITHOUT a launcher (N > 1, no WORLD_SIZE in the environment): start the N ranks
    ourselves -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>
    bench.py <the same arguments>` -- and pass rank 0's JSON line (the children inherit stdout / stderr) and the job's exit
    code through.  A rank that fails makes torchrun stop the others and exit non-zero; its traceback is on stderr."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print("[bench] no launcher in the environment: " + " ".join(cmd), file=sys.stderr, flush=True)
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log2-rows", type=int, default=20, help="N = 1 and --scaling weak: rows per GPU = 2^this (C2: 20)")
    ap.add_argument("--log2-nnz", type=int, default=24, help="N = 1 and --scaling weak: nnz per GPU = 2^this (C2: 24)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "weak"],
                    help="strong = ONE matrix (--workload) cut into N row ranges -- the default at N > 1, and at N = 1 the like-for-like "
                         "first point of the 1 -> 8 GPU curve (`--gpus 1 --scaling strong` runs C5 on one GPU with the timed step of "
                         "N > 1 minus the exchange); weak = N x C2 (round-1 mode, context only); auto = C2 at N = 1, strong at N > 1")
    ap.add_argument("--workload", default="auto", choices=["auto", "c2", "c5"],
                    help="the matrix of a strong-scaling run: c5 = 2^--c5-log2-rows rows / 2^--c5-log2-nnz nnz (BASELINE configs[4]; "
                         "auto), c2 = 2^--log2-rows rows / 2^--log2-nnz nnz (BASELINE configs[1] 'reported at 1, 2, 4 and 8 GPUs': "
                         "expect the exchange of y to bound it)")
    ap.add_argument("--c5-log2-rows", type=int, default=24, help="strong scaling: TOTAL rows = 2^this (C5: 24)")
    ap.add_argument("--c5-log2-nnz", type=int, default=29, help="strong scaling: TOTAL nnz = 2^this (C5: 29)")
    ap.add_argument("--no-one-gpu-reference", action="store_true",
                    help="strong scaling: skip rank 0's one-GPU run of the same matrix (outside the timed region)")
    ap.add_argument("--no-context", action="store_true",
                    help="N = 1: skip the context measurements (C4 BCSR, the other schedules on C2, column-blocked, local columns)")
    ap.add_argument("--tile", default="auto",
                    help="merge-tile shape TPBxIPT of the held plan; auto = the launch-box autotuner picks it on this "
                         "matrix before the timed region (loops_autotune_merge_path_f32)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-local-context", action="store_true",
                    help="skip the N = 1 context measurement of the same kernel on a banded matrix of the same size")
    ap.add_argument("--window", type=int, default=0,
                    help="0: uniform hashed columns (SURVEY 8d, the headline); W > 0: columns in a band of W around "
                         "the diagonal (locality variant, reported for context)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for functional tests)")
    ap.add_argument("--single-device", action="store_true",
                    help="functional test: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--ref-gpu", dest="ref_gpu", action="store_true", default=True,
                    help="N = 1 (default: on): also time the reference's own HIP kernels on this GPU, outside the timed region "
                         "(oracle/_ref/libloops_ref_gpu.so, the reference compiled by oracle/Makefile: three launches)")
    ap.add_argument("--no-ref-gpu", dest="ref_gpu", action="store_false", help="skip the reference-HIP-backend leg")
    ap.add_argument("--sweep", action="store_true", help="also time every compiled tile/variant (stderr)")
    ap.add_argument("--overlap-chunks", default="2,4",
                    help="N > 1: also try the step with the SpMV cut into this many row chunks whose exchanges overlap "
                         "the next chunk's kernels (comma list, one candidate each: 'p2p-chunked' for the first, "
                         "'p2p-chunked-C' for the others; 0 = do not try); the fastest candidate of the start-up probe is used")
    ap.add_argument("--no-fused-stores", action="store_true",
                    help="N > 1: do not try the exchange fused into the SpMV epilogue (peer-mapped stores, SURVEY 8 f2)")
    ap.add_argument("--exchange", default="auto",
                    help="N > 1: allgatherv implementation; auto = the fastest of the start-up probe")
    ap.add_argument("--layout", default="auto", choices=["auto", "csr", "rowband", "panel"],
                    help="how a rank holds its row-range shard: 'csr' as sliced; 'rowband' = row-band (y accumulators in LDS, "
                         "column-sorted gathers: include/loops/kernels/rowband.hxx); 'panel' = "
                         "panel-binned (x panels in LDS, no gather: include/loops/kernels/panel_binned.hxx); "
                         "auto = csr at N = 1 (the headline is the unmodified CSR), at N > 1 whichever of rowband / panel has the "
                         "smaller worst-rank time in a probe before the timed region")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    from loops_amd import generate as G, partition as P, probes as PR, spmv as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    # ------------------------------------------------------------------ workload (synthetic)
    strong = args.scaling == "strong" or (world > 1 and args.scaling == "auto")
    workload = args.workload if args.workload != "auto" else ("c5" if strong else "c2")
    if strong and workload == "c5":
        rows, nnz = 1 << args.c5_log2_rows, 1 << args.c5_log2_nnz
    elif strong:
        rows, nnz = 1 << args.log2_rows, 1 << args.log2_nnz
    else:
        rows, nnz = world << args.log2_rows, world << args.log2_nnz
    cols = rows
    t0 = time.time()
    degrees = G.powerlaw_degrees(rows, nnz)
    bounds = P.row_ranges_from_degrees(degrees, world)
    shard = P.Shard(rank, world, int(bounds[rank]), int(bounds[rank + 1]), bounds)
    off, idx, val = G.csr_from_degrees(degrees[shard.row_begin:shard.row_end], cols, seed=1, row_begin=shard.row_begin,
                                       window=args.window or None)
    x_h = G.uniform_distribution_int(cols)
    csr = S.CSR.from_numpy(shard.row_end - shard.row_begin, cols, off, idx, val)
    x = torch.from_numpy(x_h).cuda()
    y_full = torch.zeros(rows, dtype=torch.float32, device="cuda")
    y_loc = y_full[shard.row_begin:shard.row_end]
    gen_s = time.time() - t0
    tile_probe = None
    # (the default N = 1 run is the headline: the unmodified CSR.  A strong-scaling run holds its shard -- at N = 1 the whole
    # matrix -- as N > 1 does: in the re-ordered copy a start-up probe finds fastest)
    layout = args.layout if args.layout != "auto" else ("csr" if world == 1 and not strong else "auto")
    if args.tile == "auto" and layout != "csr":
        args.tile = "512x8"  # the shard is held in a re-ordered copy: nothing to tune on the CSR shard
    if "+" in args.tile:  # "512x8+phased": the phased-gather twin of that shape
        args.tile, tag = args.tile.split("+", 1)
        assert tag == "phased", f"--tile {args.tile}+{tag}: unknown kernel variant"
        args.variant = VARIANT_PHASED
    if args.tile == "auto":  # measured launch box: every compiled tile shape AND kernel variant (the phased-gather twins:
        # same CSR, same bits, another order of the x gathers) timed on this shard, outside the timed region
        best, _, tile_probe = S.autotune_merge_path_variants(csr, x, repeats=30)
        if world > 1:  # one kernel for the whole job: the one with the smallest worst-rank time (a candidate some rank did
            names = sorted(tile_probe)  # not time -- a self-completing shard has no phased twin -- is out for everybody)
            every = [None] * world
            dist.all_gather_object(every, names)
            names = sorted(set.intersection(*map(set, every)))
            t = torch.tensor([tile_probe[n] for n in names], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tile_probe = {n: float(v) for n, v in zip(names, t.tolist())}
        best = min(tile_probe, key=tile_probe.get)
        # the library's default kernel unless another one is measurably (> 1 %) faster: candidates within the run-to-run
        # noise of the probe must not flip the kernel (and its profile) between runs
        for preferred in ("512x8", "512x8+phased"):  # (the plain default first; then ONE phased shape, so that a tie between the
            if preferred in tile_probe and tile_probe[preferred] <= 1.01 * tile_probe[best]:  # two phased twins -- the rule on C2 --
                best = preferred                                                               # cannot flip the headline kernel)
                break
        args.tile = best.split("+")[0]
        if best.endswith("+phased"):
            args.variant = VARIANT_PHASED
        tile_probe = {k: round(v, 5) for k, v in tile_probe.items()}
    plan = S.MergePathPlan(csr, args.tile)
    blocked = None      # the shard's re-ordered copy, if any: a RowBandPlan or a PanelBinnedPlan
    shard_kind = "csr"  # "csr" | "rowband" | "panel"
    layout_probe = None
    csr_same_shards = None

    def make_shard_plan(kind, sub=None):
        """The re-ordered copy of a CSR (this rank's shard, or a row chunk of it) in layout `kind`."""
        m = csr if sub is None else sub
        if kind == "rowband":
            rb = S.RowBandPlan(m)
            rb.tune(5)
            return rb
        return S.PanelBinnedPlan(m)

    if layout != "csr":
        # Which copy the shards are held in: both candidates are built and timed on every rank (10 products each, outside the
        # timed region) and the job adopts the one with the smaller WORST-rank time -- every rank the same layout.  A layout
        # that cannot be built on some rank (no memory for the copy, index range) is out for everybody.
        cands = ["rowband", "panel"] if layout == "auto" else [layout]
        built, times = {}, {}
        for kind in cands:
            ok, ms = 1.0, float("inf")
            try:
                built[kind] = make_shard_plan(kind)
            except Exception as e:  # noqa: BLE001
                ok = 0.0
                print(f"[rank {rank}] shard layout {kind} unavailable ({type(e).__name__}: {e})", file=sys.stderr)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()  # every rank times the same candidate at the same time (ranks sharing a GPU in the functional test compete evenly)
            if ok:
                try:
                    ms = timed_ms(torch, lambda: built[kind].spmv(x, y_loc), 10)
                except Exception as e:  # noqa: BLE001
                    ok = 0.0
                    print(f"[rank {rank}] shard layout {kind} failed ({type(e).__name__}: {e})", file=sys.stderr)
            if world > 1:
                t = torch.tensor([ms if ok else 1e30, 1.0 - ok], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms, ok = float(t[0]), 1.0 - float(t[1])
            if ok >= 1.0:
                times[kind] = ms
            elif args.layout == kind:
                raise RuntimeError(f"--layout {kind} cannot be built on every rank")
        layout_probe = {k: round(v, 5) for k, v in times.items()}
        # the metric BASELINE.json names -- merge_path_flat on the unmodified CSR -- on the same shards, same probe (10 products,
        # worst rank): the record carries it beside whatever layout the timed region runs
        ms = timed_ms(torch, lambda: S.merge_path_flat(csr, x, y_loc, plan=plan, variant=args.variant), 10)
        if world > 1:
            t = torch.tensor([ms], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t[0])
        csr_same_shards = {"ms_per_spmv_worst_rank": round(ms, 5), "GFLOPs": round(2.0 * nnz / (ms * 1e-3) / 1e9, 2),
                           "kernel": "loops::kernels::merge_path_spmv_fused" + ("_phased" if args.variant == VARIANT_PHASED else ""),
                           "tile": args.tile, "note": "SpMV only (no exchange), start-up probe outside the timed region"}
        if times:
            shard_kind = min(times, key=times.get)
            blocked = built.pop(shard_kind)
        for other in built.values():
            other.close()
        built.clear()
    torch.cuda.synchronize()

    gather_mode = {"mode": "p2p"}
    exchanges = {}
    wd = Watchdog(rank) if world > 1 else None

    def spmv_local():
        if blocked is not None:
            blocked.spmv(x, y_loc)
        else:
            S.merge_path_flat(csr, x, y_loc, plan=plan, variant=args.variant)

    chunked = {}  # exchange mode name -> {"plans": [(run, y_sub)], "exchange": ChunkedAllgatherv, "keep": [...], "chunks": C}
    chunk_counts = [int(t) for t in str(args.overlap_chunks).split(",") if t.strip() and int(t) >= 2]

    def chunk_mode(c):
        return "p2p-chunked" if c == chunk_counts[0] else f"p2p-chunked-{c}"

    fused = {"fan": None, "run": None}

    def step():
        if gather_mode["mode"] == "fused-stores":
            fused["fan"].run(fused["run"])
            fused["fan"].finish()
            return
        if gather_mode["mode"] in chunked:
            ch = chunked[gather_mode["mode"]]
            for c, (sub_run, y_sub) in enumerate(ch["plans"]):
                sub_run(y_sub)
                ch["exchange"].post(c)
            ch["exchange"].finish()
            return
        spmv_local()
        if world > 1:
            exchanges[gather_mode["mode"]].run()

    def build_chunked(chunks):
        """The shard cut into `chunks` row pieces (balanced by rows + nnz), each with its own plan in the shard's
        layout, and the chunked exchange over the matching pieces of every rank's slice."""
        cb = P.chunk_bounds_from_degrees(degrees, bounds, chunks)
        mine = cb[rank] - shard.row_begin
        plans = []
        keep = []  # device CSRs / plans must outlive the closures
        for c in range(chunks):
            a, b = int(mine[c]), int(mine[c + 1])
            so, si, sv = P.slice_csr(off, idx, val, a, b)
            sub = S.CSR.from_numpy(b - a, cols, so, si, sv)
            y_sub = y_loc[a:b]
            if blocked is not None:
                pl = make_shard_plan(shard_kind, sub)
                run = (lambda pl: (lambda y_sub: pl.spmv(x, y_sub)))(pl)
            else:
                pl = S.MergePathPlan(sub, args.tile)
                run = (lambda sub, pl: (lambda y_sub: S.merge_path_flat(sub, x, y_sub, plan=pl, variant=args.variant)))(sub, pl)
            keep.append((sub, pl))
            plans.append((run, y_sub))
        chunked[chunk_mode(chunks)] = {"plans": plans, "keep": keep, "exchange": P.ChunkedAllgatherv(y_full, shard, cb), "chunks": chunks}

    comm_dev = "cuda" if args.backend == "nccl" else "cpu"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if world == 1:
            return float(v)
        t = torch.tensor([v], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    def probe_ms(warm=3, timed=20):
        """ms per step of the current gather_mode: max over ranks, connections / caches warmed first."""
        for _ in range(warm):
            step()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(timed):
            step()
        torch.cuda.synchronize()
        return round(max_over_ranks((time.perf_counter() - t0) / timed * 1e3), 5)

    regions = {"ms_per_step": None}

    def one_region():
        """Exactly K steps between barrier + synchronize on both sides; max over ranks (ms per step)."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        return max_over_ranks(time.perf_counter() - t0) / args.steps * 1e3

    def timed_region():
        """THE measurement: W untimed steps, then the K-step region -- bracketed as the contract says -- R times back to back,
        R = max(5, ceil(50 ms / region)) capped at 50, and the MEDIAN region is reported: a `--steps 20` run of the 0.1 ms C2
        step is a 2 ms sample, and single regions of that length differ by 2-3 % between runs (clock ramp, first-touch of the
        launch queue); the median of >= 5 agrees with a 200-step run to 1 %.  Every region's figure is in
        config.timed_regions_ms_per_step.  R derives from the first region's max-over-ranks time, so every rank runs the same count."""
        for _ in range(args.warmup):
            step()
        first = one_region()
        r = int(min(50, max(5, -(-50.0 // max(first * args.steps, 1e-6)))))
        all_ms = [first] + [one_region() for _ in range(r - 1)]
        regions["ms_per_step"] = [round(v, 5) for v in all_ms]
        return float(np.median(all_ms))

    def agreed(ok, why=None):
        """(every rank succeeded?, the reasons of those that did not).  Collective: every rank reaches it whatever failed
        locally -- a failure is agreed on, never skipped around."""
        reasons = [None] * world
        dist.all_gather_object(reasons, None if ok else (why or "failed"))
        bad = {f"rank {r}": w for r, w in enumerate(reasons) if w is not None}
        return not bad, bad

    # ------------------------------------------------------------------ parity of the local product (outside timing)
    oracle_y = None

    def check_local_parity():
        nonlocal oracle_y
        from oracle import oracle as O  # checker only
        if oracle_y is None:
            oracle_y = O.spmv_f32(off, idx, val, x_h, omp=True)
        return bool(np.array_equal(y_loc.cpu().numpy(), oracle_y))

    def check_gathered_parity():
        """The gathered vector on EVERY rank against the checksum of all ranks' oracle results (the element-wise check of the
        whole vector is the bit-for-bit comparison with rank 0's one-GPU product of the same matrix further down)."""
        ok = check_local_parity()
        if world > 1:
            s = torch.tensor([float(oracle_y.astype(np.float64).sum())], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(s)
            ok = ok and abs(float(y_full.double().sum()) - float(s)) == 0.0
            ok, _ = agreed(ok, "gathered y differs from the oracle")
        return ok

    # ------------------------------------------------------------------ per-kernel durations (HIP events
    # on the launch stream = torch's current stream) for the dominant kernel's roofline
    def event_time(fn, iters):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return float(np.mean(ts)), float(ts[len(ts) // 2])

    def batch_event_time(fn, iters):
        """Average duration of `iters` back-to-back launches between ONE pair of events: the kernel's duration plus
        the in-queue gap to the next launch (an event pair around every launch adds its own ~2.5 us to each)."""
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    iters = max(20, min(args.steps, 200))
    K_ = {"main_avg": None, "main_med": None, "main_single": None, "fix_avg": None, "reduce_avg": None, "spmv_only_ms": None}

    def measure_kernels():
        """Local (no collective inside except the final max): kernel durations of this rank's shard + its SpMV without exchange."""
        if shard_kind == "panel":  # two streaming kernels: products (x panels in LDS), sub-band reduce
            K_["main_single"], K_["main_med"] = event_time(lambda: blocked.spmv_stage(0, x, y_loc), iters)
            K_["main_avg"] = batch_event_time(lambda: blocked.spmv_stage(0, x, y_loc), iters)
            K_["fix_avg"] = 0.0
            K_["reduce_avg"] = batch_event_time(lambda: blocked.spmv_stage(1, x, y_loc), iters)
        elif blocked is not None:  # row-band: accumulate (band sums in LDS), then the combine of bands cut into chunks (if any)
            K_["main_single"], K_["main_med"] = event_time(lambda: blocked.spmv_stage(0, x, y_loc), iters)
            K_["main_avg"] = batch_event_time(lambda: blocked.spmv_stage(0, x, y_loc), iters)
            K_["fix_avg"] = 0.0
            K_["reduce_avg"] = batch_event_time(lambda: blocked.spmv_stage(1, x, y_loc), iters) if blocked.num_partials else 0.0
        else:
            K_["main_single"], K_["main_med"] = event_time(lambda: S.merge_path_flat_stage(csr, x, y_loc, plan, 0, args.variant), iters)
            K_["main_avg"] = batch_event_time(lambda: S.merge_path_flat_stage(csr, x, y_loc, plan, 0, args.variant), iters)
            K_["fix_avg"], _ = event_time(lambda: S.merge_path_flat_stage(csr, x, y_loc, plan, 1, args.variant), iters)
        # the SpMV of a step without the exchange (BASELINE C5: "kernel-only and kernel + allgatherv")
        for _ in range(5):
            spmv_local()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            spmv_local()
        torch.cuda.synchronize()
        K_["spmv_only_ms"] = max_over_ranks((time.perf_counter() - t0) / iters * 1e3)  # the slowest rank's kernels: what the exchange waits for

    exchange_probe = None
    exchange_dropped = {}
    safe = {"mode": None, "ms_per_step": None, "parity": None}
    R_ = {"one_gpu": None, "ms_with_prepass": None, "rowband_info": None, "local_info": None, "schedules_info": None, "c4_info": None,
          "c3_info": None, "copy_gbps": None, "gather_gps": None, "ref_gpu": None, "cpu": None, "fused_note": None, "l2_gather_gps": None, "panel_info": None}

    def record(ms_per_step, parity, watchdog=None):
        """The one JSON line, from whatever has been measured so far (the watchdog calls it with the safe exchange's figures)."""
        gflops = 2.0 * nnz / (ms_per_step * 1e-3) / 1e9
        loc_rows, loc_nnz = csr.rows, csr.nnzs
        abytes = algorithmic_bytes(loc_rows, cols, loc_nnz)
        k_main = K_["main_avg"]
        if k_main and shard_kind in ("panel", "rowband"):  # the product is two kernels: the roofline is quoted on their sum
            k_main = K_["main_avg"] + K_["reduce_avg"]
        roofline = None
        if k_main:
            achieved = abytes / (k_main * 1e-3) / 1e9
            traffic, traffic_src, traffic_note = pmc_traffic(args)
            counters = pmc_bound(args)
            roofline = {"bound": "hbm", "kernel": {"csr": "loops::kernels::merge_path_spmv_fused" + ("_phased" if args.variant == VARIANT_PHASED else ""),
                                                    "rowband": "loops::kernels::rowband::rowband_accumulate + rowband_combine",
                                                    "panel": "loops::kernels::panel::panel_products + panel_reduce"}[shard_kind],
                        "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                        "traffic_source": traffic_src, "traffic_note": traffic_note, "counters": counters,
                        "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": round(k_main, 5),
                        "median_launch_ms": round(K_["main_med"], 5), "avg_launch_ms_event_pair_per_launch": round(K_["main_single"], 5),
                        "panel_products_avg_launch_ms": round(K_["main_avg"], 5) if shard_kind == "panel" else None,
                        "fixup_avg_launch_ms": round(K_["fix_avg"], 5)}
            if R_["copy_gbps"]:
                roofline.update({"measured_copy_GBps": round(R_["copy_gbps"], 1), "frac_of_measured_copy": round(achieved / R_["copy_gbps"], 4)})
            if R_["gather_gps"]:
                g = R_["gather_gps"]
                roofline.update({"measured_gather_Gelem_per_s": round(g, 2),
                                 # time the x gathers of this shard alone need at the measured random-gather rate of this box
                                 # (same indices, same x, no streams) over the kernel's time: how much of the kernel is the gather
                                 "gather_only_ms": round(loc_nnz / g / 1e6, 5),
                                 "gather_only_over_kernel": round(loc_nnz / g / 1e6 / k_main, 4)})
            if R_["l2_gather_gps"]:
                # what bounds `frac` on this input: CSR needs one 4-byte gather of x per nonzero, and the chip serves
                # scattered 4-byte loads that all HIT an L2 at this measured rate (hashed loads over a table of x's size, no streams:
                # 16 tag lookups per clk per XCD).  No one-gather-per-nonzero kernel can be faster than nnz / rate on this matrix
                g2 = R_["l2_gather_gps"]
                roofline.update({"pure_l2_hit_gather_Gelem_per_s": round(g2, 1), "request_rate_floor_ms": round(loc_nnz / g2 / 1e6, 5),
                                 "request_rate_floor_frac": round(abytes / (loc_nnz / g2 / 1e6 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                 "kernel_over_request_rate_floor": round(k_main / (loc_nnz / g2 / 1e6), 3)})
            if K_["reduce_avg"] is not None:
                roofline["panel_reduce_avg_launch_ms" if shard_kind == "panel" else "rowband_combine_avg_launch_ms"] = round(K_["reduce_avg"], 5)
        mode = gather_mode["mode"] if watchdog is None else safe["mode"]
        one_gpu, spmv_only_ms = R_["one_gpu"], K_["spmv_only_ms"]
        step_includes = {"csr": "fused merge-tile kernel" + (f" (phased x gathers: {8 if cols * 4 <= (6 << 20) else 16 if cols * 4 <= (24 << 20) else 32} passes by column range, clock-aligned across workgroups)"
                                                            if args.variant == VARIANT_PHASED else "") + " + carry-out fix-up", "rowband": "row-band accumulate (band sums in LDS) + combine of the cut bands",
                         "panel": "panel products (x panels in LDS) + sub-band reduce"}[shard_kind]
        if world > 1:
            step_includes += f" + allgatherv(y) [{mode}"
            if mode == "fused-stores":
                step_includes += ": finished rows stored to the peers from the kernels' epilogue + one barrier"
            if mode in chunked:
                step_includes += f", {chunked[mode]['chunks']} chunks overlapping the SpMV"
            step_includes += "]"
        return {
            # what the timed region runs: merge_path_flat on the unmodified CSR, or a held plan over a re-ordered copy of the shards (then
            # config.merge_path_flat_csr_same_shards holds the merge_path_flat figure of the same shards)
            "metric": {"csr": "CSR SpMV GFLOP/s, merge_path_flat", "rowband": "CSR SpMV GFLOP/s, row-band held plan",
                       "panel": "CSR SpMV GFLOP/s, panel-binned held plan"}[shard_kind], "value": round(gflops, 2), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "strong" if strong else (None if world == 1 else "weak"), "vs_baseline": None,
            # the scaling curve's own figures at the top level: which matrix, what the step holds, the ratio to the SAME matrix on
            # one GPU (N > 1: measured on rank 0 outside the timed region; N = 1 strong: this record is that denominator)
            "scaling_detail": None if not strong else {
                "workload": workload, "rows": rows, "nnz": nnz, "shard_layout": shard_kind,
                "exchange": None if world == 1 else (gather_mode["mode"] if watchdog is None else safe["mode"]),
                "spmv_only_ms_per_step": None if K_["spmv_only_ms"] is None else round(K_["spmv_only_ms"], 5),
                "speedup_vs_one_gpu_same_matrix": None if not R_["one_gpu"] or world == 1 else
                                                  round(R_["one_gpu"]["best_ms_per_spmv"] / ms_per_step, 3),
                "like_for_like_first_point": "python bench.py --gpus 1 --scaling strong" + ("" if workload == "c5" else " --workload c2")},
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic power-law CSR, {rows} rows / {nnz} nnz total "
                                   + (f"(ONE matrix cut into {world} row ranges balanced by rows + nnz: {loc_nnz} nnz on rank 0)" if strong else
                                      f"({world} x 2^{args.log2_rows} rows / 2^{args.log2_nnz} nnz per GPU)") + ", max degree 2^14, "
                                   "fp32, merge_path_flat" + (f", columns banded (window {args.window})" if args.window else ", columns uniform")
                                   + (f", row-range sharded + allgatherv(y) over {'RCCL' if args.backend == 'nccl' else 'gloo (functional test)'}" if world > 1 else ""),
                       "baseline_config": ("BASELINE.json configs[4] (C5), strong scaling" if workload == "c5" else
                                           "BASELINE.json configs[1] (C2) as ONE matrix cut into N row ranges, strong scaling") if strong else
                                          ("BASELINE.json configs[1]" if world == 1 else "configs[1] per GPU (weak scaling; context mode)"),
                       "tile": args.tile, "tile_autotune_ms": tile_probe, "variant": args.variant,
                       "merge_tiles_per_gpu": plan.num_tiles,
                       "shard_layout": "csr" if blocked is None else
                                       (f"row-band, {blocked.num_bands} bands of {blocked.H} rows in {blocked.num_chunks} chunks, {blocked.waves} wavefronts "
                                        f"(x per GPU {cols * 4 >> 20} MB)" if shard_kind == "rowband" else
                                        f"panel-binned{' (compact)' if blocked.compact else ''}, {blocked.num_panels} panels of {blocked.W} columns x "
                                        f"{blocked.num_subbands} sub-bands of {blocked.Hw} rows (x per GPU {cols * 4 >> 20} MB)"),
                       "shard_layout_probe_ms": layout_probe,
                       "merge_path_flat_csr_same_shards": csr_same_shards,
                       "step_includes": step_includes,
                       "ms_per_step_with_prepass": None if R_["ms_with_prepass"] is None else round(R_["ms_with_prepass"], 5),
                       "timed_regions_ms_per_step": regions["ms_per_step"],
                       "timed_region_statistic": "median of the K-step regions listed in timed_regions_ms_per_step",
                       # the single contract-bracketed region (W warm-up steps, then exactly K steps between barrier + synchronize):
                       "first_region_ms_per_step": regions["ms_per_step"][0] if regions["ms_per_step"] else None,
                       "achieved_GBps_whole_step": round(algorithmic_bytes(rows, cols, nnz) / world / (ms_per_step * 1e-3) / 1e9, 1),
                       "spmv_only_ms_per_step": None if spmv_only_ms is None else round(spmv_only_ms, 5),
                       "spmv_only_GFLOPs": None if spmv_only_ms is None else round(2.0 * nnz / (spmv_only_ms * 1e-3) / 1e9, 2),
                       "spmv_plus_allgatherv_ms_per_step": round(ms_per_step, 5) if world > 1 else None,
                       "one_gpu_same_matrix": one_gpu,
                       "speedup_vs_one_gpu_same_matrix": None if not one_gpu else
                                                         {"spmv_plus_allgatherv": round(one_gpu["best_ms_per_spmv"] / ms_per_step, 3),
                                                          "spmv_only": round(one_gpu["best_ms_per_spmv"] / spmv_only_ms, 3),
                                                          "target": ">= 6 at 8 GPUs (BASELINE.md section 2, C5)"},
                       "allgatherv_probe_ms_per_step": exchange_probe,
                       "allgatherv_candidates_dropped": exchange_dropped if world > 1 else None,
                       "allgatherv_safe_mode_ms_per_step": None if world == 1 else {"mode": safe["mode"], "ms_per_step": safe["ms_per_step"]},
                       "fused_stores_note": R_["fused_note"],
                       "watchdog": watchdog,
                       "parity_vs_oracle_bit_exact": parity, "generate_seconds": round(gen_s, 1),
                       "row_band_layout_same_matrix": R_["rowband_info"],
                       "panel_binned_layout_same_matrix": R_["panel_info"],
                       "same_kernel_local_columns": R_["local_info"],
                       "schedules_c2": R_["schedules_info"],
                       "c4_bcsr_mfma": R_["c4_info"],
                       "c3_standin_schedules": R_["c3_info"],
                       "reference_hip_backend_on_this_gpu": R_["ref_gpu"]},
            "roofline": roofline, "cpu_baseline": R_["cpu"],
        }

    parity = None
    if world > 1:
        # The exchange is an allgatherv(y).  Implementations (loops_amd/partition.py): "p2p" = one grouped batch of direct
        # sends / receives (every xGMI link at once), "padded" = the library all_gather on max-count slots + local
        # compaction, "p2p-chunked[-C]" = p2p posted per row chunk so that it overlaps the next chunk's kernels,
        # "fused-stores" = no exchange step, the kernels' epilogue stores to the peers.  Which one is fastest depends on
        # the RCCL build, the host cost of a grouped launch and the fabric, so all are timed here and every rank adopts the
        # same winner.  MEASURE FIRST: as soon as ONE library exchange works, the whole timed region is run with it and kept
        # as the watchdog's fallback record; only then are the other candidates tried, each under a deadline.
        exchange_probe = {}
        for mode in ("p2p", "padded"):
            wd.arm(f"exchange candidate {mode}", 300)
            why = None
            try:
                exchanges[mode] = P.Allgatherv(y_full, shard, mode)
                gather_mode["mode"] = mode
                step()
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                why = f"{type(e).__name__}: {e}"
                print(f"[rank {rank}] allgatherv mode {mode} unavailable ({why})", file=sys.stderr)
            ok, bad = agreed(why is None, why)
            if not ok:
                exchanges.pop(mode, None)
                exchange_dropped[mode] = bad
                continue
            exchange_probe[mode] = probe_ms()
            if safe["mode"] is None:
                safe["mode"] = mode
                if not args.no_check:
                    step()
                    torch.cuda.synchronize()
                    safe["parity"] = check_gathered_parity()
                    assert safe["parity"], "GPU result differs from the oracle"
                safe["ms_per_step"] = round(timed_region(), 5)
                measure_kernels()
                wd.fallback = lambda reason: record(safe["ms_per_step"], safe["parity"], watchdog=reason)
        wd.disarm()
        assert exchange_probe, f"no allgatherv implementation works on this backend: {exchange_dropped}"
        gather_mode["mode"] = safe["mode"]
        step()
        barrier()
        y_exchanged = y_full.clone()  # what every other candidate must reproduce, element for element, on every rank

        def reproduces(name):
            """One step of the current gather_mode into a zeroed y_full equals the exchanged vector on every rank."""
            y_full.zero_()
            barrier()
            why = None
            try:
                step()
            except Exception as e:  # noqa: BLE001
                why = f"{type(e).__name__}: {e}"
            barrier()
            if why is None and not torch.equal(y_full, y_exchanged):
                why = "result differs from the exchanged vector"
            if why is None:
                # ... and the NEXT product must arrive too: with x doubled every element of y doubles exactly, so a rank that still
                # sees last step's values somewhere (a line of its y_full cached from before a peer's store) is caught here
                try:
                    x.mul_(2.0)
                    barrier()
                    step()
                    barrier()
                    if not torch.equal(y_full, 2.0 * y_exchanged):
                        why = "a second product (x doubled) did not arrive everywhere"
                    x.mul_(0.5)
                    barrier()
                    step()
                    barrier()
                    if why is None and not torch.equal(y_full, y_exchanged):
                        why = "a third product (x restored) did not arrive everywhere"
                except Exception as e:  # noqa: BLE001
                    why = f"{type(e).__name__}: {e}"
            ok, bad = agreed(why is None, why)
            if not ok:
                exchange_dropped[name] = bad
                if rank == 0:
                    print(f"[rank 0] exchange candidate {name} dropped: {bad}", file=sys.stderr)
            return ok

        for chunks in (chunk_counts if "p2p" in exchange_probe else []):
            # (at N = 8 a shard's y slice needs ~110 us of xGMI link time against ~570 us of kernels: with C chunks only
            # the last chunk's 1 / C of it stays exposed, at the price of C smaller launches and C grouped p2p calls)
            name = chunk_mode(chunks)
            wd.arm(f"exchange candidate {name}", 300)
            why = None
            try:
                build_chunked(chunks)
            except Exception as e:  # noqa: BLE001
                why = f"{type(e).__name__}: {e}"
            ok, bad = agreed(why is None, why)
            if ok:
                gather_mode["mode"] = name
                if reproduces(name):
                    exchange_probe[name] = probe_ms()
                    continue
            else:
                exchange_dropped[name] = bad
            chunked.pop(name, None)
        wd.disarm()
        if args.backend == "nccl":
            # the same grouped point-to-point exchange issued by the library itself (loops_allgatherv_f32 on a communicator of
            # its own, include/loops/multi_gpu/allgatherv.hxx): no Python op list, no work objects per call.  RCCL only (two
            # ranks of the gloo functional test share one device, which RCCL refuses).
            wd.arm("exchange candidate native-p2p", 300)
            why = None
            try:
                exchanges["native-p2p"] = P.NativeAllgatherv(y_full, shard)
            except Exception as e:  # noqa: BLE001
                why = f"{type(e).__name__}: {e}"
            ok, bad = agreed(why is None, why)
            if ok:
                gather_mode["mode"] = "native-p2p"
                if reproduces("native-p2p"):
                    exchange_probe["native-p2p"] = probe_ms()
            else:
                exchange_dropped["native-p2p"] = bad
                exchanges.pop("native-p2p", None)
                if rank == 0:
                    print(f"[rank 0] exchange candidate native-p2p dropped: {bad}", file=sys.stderr)
            wd.disarm()
        if args.no_fused_stores:
            exchange_dropped["fused-stores"] = {"all ranks": "--no-fused-stores"}
        else:
            # no exchange step at all -- the kernels that finish rows of y also store them into every peer's vector through
            # peer-mapped memory (loops_spmv_*_fanout_f32); one tiny barrier ends the step.  Adopted only if it maps on every
            # rank, reproduces the exchanged vector on every rank and is faster.
            wd.arm("exchange candidate fused-stores", 300)
            handles = [None] * world
            why = None
            try:
                mine = P.FusedFanout.export(y_full)
            except Exception as e:  # noqa: BLE001
                why = f"cannot export y for peer mapping ({type(e).__name__}: {e})"
                mine = None
            dist.all_gather_object(handles, mine)
            if why is None and not all(h is not None for h in handles):
                why = "a peer could not export its vector"
            if why is None:
                try:
                    fused["fan"] = P.FusedFanout(y_full, shard, P.FusedFanout.open_peers(handles, rank))
                    if blocked is not None:
                        fused["run"] = lambda y, peers: blocked.spmv_fanout(x, y, peers)
                    else:
                        fan_plan = plan if args.tile == "512x8" else S.MergePathPlan(csr, "512x8")
                        fused["plan"] = fan_plan
                        fused["run"] = lambda y, peers: S.merge_path_flat_fanout(csr, x, y, fan_plan, peers)
                except Exception as e:  # noqa: BLE001
                    why = f"peer mapping unavailable ({type(e).__name__}: {e}; HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')})"
            ok, bad = agreed(why is None, why)
            if ok:
                gather_mode["mode"] = "fused-stores"
                if reproduces("fused-stores"):
                    exchange_probe["fused-stores"] = probe_ms()
                    R_["fused_note"] = ("the timed loop never reads y_full between steps; an iterative consumer that does must alternate two "
                                        "y_full buffers or add a barrier before the next product (partition.FusedFanout), which this step time excludes")
            else:
                exchange_dropped["fused-stores"] = bad
                if rank == 0:
                    print(f"[rank 0] exchange candidate fused-stores dropped: {bad}", file=sys.stderr)
            wd.disarm()
        gather_mode["mode"] = min(exchange_probe, key=exchange_probe.get)
        if args.exchange != "auto":
            assert args.exchange in exchange_probe, f"--exchange {args.exchange} is not available here: {exchange_probe} (dropped: {exchange_dropped})"
            gather_mode["mode"] = args.exchange
        wd.arm(f"timed region with {gather_mode['mode']}", 600)

    # ------------------------------------------------------------------ parity (outside timing)
    if not args.no_check:
        # the adopted exchange must reproduce the oracle; one that does not is dropped (reason recorded) and the next fastest
        # takes its place -- a wrong candidate costs that candidate, not the run
        order = [gather_mode["mode"]] if (world == 1 or args.exchange != "auto") else sorted(exchange_probe, key=exchange_probe.get)
        for mode in order:
            gather_mode["mode"] = mode
            step()
            torch.cuda.synchronize()
            parity = check_gathered_parity()
            if parity:
                break
            if world > 1:
                exchange_dropped[mode] = {"all ranks": "the gathered y differed from the oracle after adoption"}
                exchange_probe.pop(mode, None)
        assert parity, "GPU result differs from the oracle"

    # ------------------------------------------------------------------ timed region
    ms_per_step = timed_region()
    measure_kernels()
    k_main_avg = K_["main_avg"]
    if wd is not None:
        wd.fallback = lambda reason: record(ms_per_step, parity, watchdog=reason + " (the timed region with the adopted exchange had completed)")
        wd.arm("one-GPU run of the same matrix", 1200)

    # BASELINE C5's denominator: the same matrix on one GPU (rank 0, outside the timed region; the others wait)
    if strong and world > 1 and not args.no_one_gpu_reference:
        step()  # (collective: every rank) y_full = the gathered vector the one-GPU result is compared with
        barrier()
        if rank == 0:
            try:
                R_["one_gpu"] = one_gpu_same_matrix(G, S, torch, degrees, cols, x, y_full)
            except Exception as e:  # noqa: BLE001 -- the N-GPU measurement stands without its denominator
                R_["one_gpu"] = None
                print(f"[rank 0] one-GPU run of the same matrix failed ({type(e).__name__}: {e})", file=sys.stderr)
        barrier()
    if wd is not None:
        wd.disarm()

    if blocked is None:  # the plan-less entry point: coordinates rebuilt every call, as the reference wrapper does
        def with_prepass():
            S.spmv("merge_path_flat", csr, x, y_loc)  # loops_spmv_csr_f32: coordinates + tile kernel + fix-up, no held plan

        for _ in range(5):
            with_prepass()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            with_prepass()
        torch.cuda.synchronize()
        R_["ms_with_prepass"] = (time.perf_counter() - t0) / iters * 1e3

    def build_price(make, layout_ms, csr_ms):
        """(plan, wall ms of one creation incl. the device work, products after which the copy has paid for itself against the CSR)."""
        make().close()  # (the first creation pays allocator warm-up)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan_ = make()
        torch.cuda.synchronize()
        b_ms = (time.perf_counter() - t0) * 1e3
        return plan_, b_ms

    def break_even(b_ms, layout_ms, csr_ms):
        return round(b_ms / (csr_ms - layout_ms), 1) if layout_ms < csr_ms else None

    # for context at N = 1: the same SpMV from the row-band copy (y accumulators in LDS, column-sorted gathers; never `value`)
    if world == 1 and blocked is None and rank == 0 and not args.no_context:
        rb, rb_build = build_price(lambda: S.RowBandPlan(csr), None, None)
        tune8, tune16 = rb.tune(10)
        yb = torch.empty_like(y_loc)
        ms_b = batch_event_time(lambda: rb.spmv(x, yb), iters)
        ms_ba = batch_event_time(lambda: rb.spmv_stage(0, x, yb), iters)
        ms_bb = batch_event_time(lambda: rb.spmv_stage(1, x, yb), iters) if rb.num_partials else 0.0
        rb.spmv(x, yb)
        torch.cuda.synchronize()
        ab = algorithmic_bytes(csr.rows, cols, csr.nnzs)
        csr_ms = K_["main_avg"] + (K_["fix_avg"] or 0.0) if K_["main_avg"] else None
        R_["rowband_info"] = {"bands": rb.num_bands, "band_rows": rb.H, "chunks": rb.num_chunks, "partial_vectors": rb.num_partials,
                              "wavefronts": rb.waves, "tune_ms": {"8": round(tune8, 5), "16": round(tune16, 5)},
                              "padding_items_over_nnz": round((rb.padded - csr.nnzs) / max(csr.nnzs, 1), 4),
                              "ms_per_step": round(ms_b, 5), "accumulate_ms": round(ms_ba, 5), "combine_ms": round(ms_bb, 5),
                              "GFLOPs": round(2.0 * nnz / (ms_b * 1e-3) / 1e9, 2), "frac": round(ab / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                              "plan_build_ms": round(rb_build, 2),
                              "break_even_products": None if csr_ms is None else break_even(rb_build, ms_b, csr_ms),
                              "equal_to_csr_result": bool(torch.equal(yb, y_loc)),
                              "note": "plan-time re-ordered copy of the matrix (include/loops/kernels/rowband.hxx): 7 B per nonzero streamed, the "
                                      "band's y sums in LDS (fp64), x gathered through column-sorted (coalescing) loads; not the headline"}
        rb.close()

    # for context at N = 1: the same SpMV from the panel-binned copy (x panels in LDS, no memory gather; never `value`)
    if world == 1 and blocked is None and rank == 0 and not args.no_context:
        pb, pb_build = build_price(lambda: S.PanelBinnedPlan(csr), None, None)
        yp = torch.empty_like(y_loc)
        ms_p = batch_event_time(lambda: pb.spmv(x, yp), iters)
        ms_pa = batch_event_time(lambda: pb.spmv_stage(0, x, yp), iters)
        ms_pb = batch_event_time(lambda: pb.spmv_stage(1, x, yp), iters)
        pb.spmv(x, yp)
        torch.cuda.synchronize()
        ab = algorithmic_bytes(csr.rows, cols, csr.nnzs)
        R_["panel_info"] = {"panels": pb.num_panels, "panel_columns": pb.W, "subbands": pb.num_subbands, "subband_rows": pb.Hw,
                            "compact": pb.compact, "runs_over_nnz": round(pb.runs / max(csr.nnzs, 1), 4),
                            "ms_per_step": round(ms_p, 5), "products_ms": round(ms_pa, 5), "reduce_ms": round(ms_pb, 5),
                            "GFLOPs": round(2.0 * nnz / (ms_p * 1e-3) / 1e9, 2), "frac": round(ab / (ms_p * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                            "plan_build_ms": round(pb_build, 2),
                            "break_even_products": break_even(pb_build, ms_p, K_["main_avg"] + (K_["fix_avg"] or 0.0)) if K_["main_avg"] else None,
                            "equal_to_csr_result": bool(torch.equal(yp, y_loc)),
                            "note": "plan-time re-ordered copy of the matrix (include/loops/kernels/panel_binned.hxx): 7 B read per nonzero + "
                                    "10 B per run of equal (row, panel) -- 17 B per nonzero when nothing is pre-summed -- instead of 8 B + a gather; "
                                    "fp64 accumulators in LDS; not the headline"}
        pb.close()

    # for context at N = 1: the SAME kernel on a matrix of the same size whose columns are local (16 per row inside a
    # 64-column band): what the kernel does when the x gather is served by L1 -- its roofline fraction as a kernel,
    # next to the headline's, which is set by the random gather (never `value`)
    if world == 1 and not strong and rank == 0 and not args.window and not args.no_local_context and not args.no_context:
        l_off, l_idx, l_val = G.csr_from_degrees(np.full(csr.rows, csr.nnzs // csr.rows, np.int64), cols, seed=1, window=64)
        l_csr = S.CSR.from_numpy(csr.rows, cols, l_off, l_idx, l_val)
        l_plan = S.MergePathPlan(l_csr, "256x8")
        yl = torch.empty_like(y_loc)
        for _ in range(5):
            S.merge_path_flat(l_csr, x, yl, plan=l_plan)
        l_avg = batch_event_time(lambda: S.merge_path_flat(l_csr, x, yl, plan=l_plan), 200)
        y_tm = torch.empty_like(yl)  # cross-check against the row-sequential thread_mapped kernel (exact inputs: equal)
        S.spmv("thread_mapped", l_csr, x, y_tm)
        l_ok = bool(torch.equal(yl, y_tm))
        lb = algorithmic_bytes(l_csr.rows, cols, l_csr.nnzs)
        R_["local_info"] = {"workload": f"{l_csr.rows} rows x {l_csr.nnzs // l_csr.rows} nnz, columns in a 64-wide band, fp32, held plan 256x8"
                                        + (" (self-completing: one kernel)" if l_plan.self_complete else ""),
                            "avg_launch_ms": round(l_avg, 5), "GFLOPs": round(2.0 * l_csr.nnzs / (l_avg * 1e-3) / 1e9, 1),
                            "roofline": {"bound": "hbm", "achieved": round(lb / (l_avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                                         "unit": "GB/s", "frac": round(lb / (l_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
                            "equal_to_thread_mapped_result": l_ok, "note": "context, not the headline workload"}
        del l_csr, l_plan, yl, y_tm

    # for context at N = 1: the other tuned schedules on the headline matrix, and BASELINE C4 (BCSR 4x4 + MFMA) at full size
    if world == 1 and not strong and rank == 0 and not args.no_context and not args.window:
        step()
        torch.cuda.synchronize()
        R_["schedules_info"] = context_schedules(S, torch, csr, x, y_loc.clone(), algorithmic_bytes(csr.rows, cols, csr.nnzs))
        if not args.no_check:
            from oracle import oracle as O  # checker only
            R_["c4_info"] = context_c4_bcsr(G, S, O, torch)
            R_["c3_info"] = context_c3_standins(G, S, O, torch)

    # calibration probes: achievable streaming rate and gather rate on this box
    n_copy = 1 << 28  # 1 GiB in + 1 GiB out: beyond the 256 MiB Infinity Cache
    src = torch.empty(n_copy, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    # (4 vectors per lane in flight, non-temporal loads and stores, 32 workgroups per CU: the fastest of scripts/probe_copy_rate.py
    #  -- 5.9 TB/s where the plain one-vector loop gets 4.7)
    copy_avg = batch_event_time(lambda: PR.stream_copy_tuned(src, dst, 4, 3, 256 * 32), 20)
    R_["copy_gbps"] = 2 * n_copy * 4 / (copy_avg * 1e-3) / 1e9
    del src, dst
    gidx = torch.from_numpy(idx[: 1 << 24]).cuda() if idx.size >= 1 << 24 else csr.indices
    gout = torch.empty(gidx.numel(), dtype=torch.float32, device="cuda")
    gat_avg = batch_event_time(lambda: PR.gather(x, gidx, gout), 20)
    R_["gather_gps"] = gidx.numel() / (gat_avg * 1e-3) / 1e9
    # scattered 4-byte loads that all hit L2: hashed addresses over a table of x's size (power of two), nothing else in flight
    words = 1 << max(10, int(cols - 1).bit_length())
    table = torch.rand(words, device="cuda") if words * 4 <= (64 << 20) else None
    if table is not None and words * 4 <= (4 << 20):  # (the floor is about an x that fits ONE L2; larger x is bound by the fabric instead)
        pblocks, preps = 2048, 2048
        pout = torch.zeros(pblocks * 256, device="cuda")
        l2_avg = batch_event_time(lambda: PR.address_rate(table, preps, 1, pblocks, pout), 5)
        R_["l2_gather_gps"] = pblocks * 256 * preps / (l2_avg * 1e-3) / 1e9
    del table

    loc_rows, loc_nnz = csr.rows, csr.nnzs
    abytes = algorithmic_bytes(loc_rows, cols, loc_nnz)

    if args.sweep and rank == 0:
        for tile in ("256x8", "256x7", "128x7", "512x8", "256x16"):
            p2 = S.MergePathPlan(csr, tile)
            for variant in (0, 1, 2, 3):
                avg, med = event_time(lambda: S.merge_path_flat_stage(csr, x, y_loc, p2, 0, variant), 50)
                print(f"[sweep] tile={tile} variant={variant} main kernel avg {avg*1e3:.1f} us med {med*1e3:.1f} us "
                      f"-> {abytes/avg/1e6:.0f} GB/s", file=sys.stderr)

    # ------------------------------------------------------------------ the reference's own HIP path on this GPU
    so = os.path.join(ROOT, "oracle", "_ref", "libloops_ref_gpu.so")
    if args.ref_gpu and rank == 0 and world == 1 and not strong and not args.window and os.path.exists(so):
        import ctypes as C
        from loops_amd import _lib
        R = _lib.load_shared(so)
        ref_gpu = {}
        for kind, name in ((2, "merge_path_flat"), (0, "thread_mapped"), (1, "work_oriented")):
            yr = np.zeros(csr.rows, np.float32)
            ms = C.c_float()
            p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
            rc = R.refgpu_spmv_f32(kind, C.c_long(csr.rows), C.c_long(cols), C.c_long(csr.nnzs), p(off), p(idx), p(val),
                                   p(x_h), p(yr), 10, C.byref(ms))
            ref_gpu[name] = {"rc": rc, "best_kernel_ms": round(ms.value, 5),
                             "GFLOPs": round(2.0 * csr.nnzs / (ms.value * 1e-3) / 1e9, 2) if ms.value > 0 else None}
        ref_gpu["what"] = ("the reference's own kernels (include/loops/algorithms/spmv/{merge_path_flat,thread_mapped,work_oriented}.cuh, its HIP "
                           "backend) compiled from /root/reference by oracle/Makefile and run on this GPU on the same matrix: kernel-only time of "
                           "the wrapper's util::timer_t (merge_path_flat.cuh:111-136), best of 10, y zero-filled outside")
        R_["ref_gpu"] = ref_gpu
    elif args.ref_gpu and rank == 0 and world == 1 and not strong and not args.window:
        R_["ref_gpu"] = {"error": "oracle/_ref/libloops_ref_gpu.so is not built (python -c 'import __graft_entry__ as g; g.build_checker()' where "
                                  "/root/reference is mounted)"}

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1)
    if rank == 0 and world == 1 and not strong and not args.no_cpu_baseline:
        from oracle import oracle as O
        # bounded sample: ~10 s of single-core work (what --validate executes) + ~3 s of the OpenMP variant
        t0 = time.perf_counter()
        O.spmv_f32(off, idx, val, x_h)
        first = time.perf_counter() - t0
        reps1 = int(min(2000, max(8, 10.0 / max(first, 1e-6))))
        t0 = time.perf_counter()
        for _ in range(reps1):
            O.spmv_f32(off, idx, val, x_h)
        t1 = (time.perf_counter() - t0) / reps1
        # OpenMP leg: the oracle sizes its team to the CPUs this process may really use (affinity mask capped by
        # the cgroup quota, oracle.usable_cpus): a box that shows 256 logical CPUs but grants 16 CPUs' worth of
        # time throttles a 128-thread team to a crawl
        threads = O.lib().oracle_num_threads()
        t0 = time.perf_counter()
        O.spmv_f32(off, idx, val, x_h, omp=True)
        O.spmv_f32(off, idx, val, x_h, omp=True)
        firstn = (time.perf_counter() - t0) / 2
        repsn = int(min(5000, max(16, 3.0 / max(firstn, 1e-6))))
        t0 = time.perf_counter()
        for _ in range(repsn):
            O.spmv_f32(off, idx, val, x_h, omp=True)
        tn = (time.perf_counter() - t0) / repsn
        R_["cpu"] = {"value": round(2.0 * loc_nnz / t1 / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
                     "sample": f"{reps1} full passes of the same C2 matrix ({loc_rows} rows, {loc_nnz} nnz), "
                               "oracle/loops_oracle.c oracle_spmv_f32 (restatement of reference::spmv, "
                               f"util/reference.hxx:57-76), gcc -O3 -march=x86-64-v3; {reps1 * t1:.1f} s of CPU work",
                     "all_cores": {"value": round(2.0 * loc_nnz / tn / 1e9, 3), "cores": int(threads),
                                   "note": f"same loop, OpenMP row-parallel schedule(dynamic,1024), {repsn} passes"}}

    if rank == 0:
        print(json.dumps(record(ms_per_step, parity)), flush=True)
    if world > 1:
        wd.fallback, wd.record_printed = None, True
        wd.arm("teardown", 120)  # (a process that cannot leave its last barrier must not outlive the record it printed)
        fused.clear()  # peer mappings go before the processes that own the memory do
        handles = None
        import gc
        gc.collect()
        torch.cuda.ipc_collect()
        barrier()
        dist.destroy_process_group()
        wd.disarm()


if __name__ == "__main__":
    main()
