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

# (re-exported: scripts/pmc_summarize.py, tests/test_bench_watchdog.py and tests/perf use these names off `bench`)
from benchlib.counters import (HBM_PEAK_GBPS, KERNEL_SOURCES, VARIANT_PHASED, algorithmic_bytes, headline_kernel, kernel_sources_digest,  # noqa: E402,F401
                               pmc_bound, pmc_summary, pmc_traffic)
from benchlib.context import (context_c3_standins, context_c4_bcsr, context_schedules, full_matrix_on_device, one_gpu_same_matrix,  # noqa: E402,F401
                              timed_ms)
from benchlib.launch import Watchdog, self_launch  # noqa: E402,F401


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
        # (both gather orders of the kernel where the tile shape has a phased twin: the faster one on the WORST rank is the figure)
        variants = [0] + ([VARIANT_PHASED] if args.tile in ("512x8", "256x16") else [])
        ms_by = []
        for v in variants:
            ms = timed_ms(torch, lambda: S.merge_path_flat(csr, x, y_loc, plan=plan, variant=v), 10)
            if world > 1:
                t = torch.tensor([ms], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms = float(t[0])
            ms_by.append(ms)
        ms = min(ms_by)
        phased_won = len(ms_by) > 1 and ms_by[1] < ms_by[0]
        csr_same_shards = {"ms_per_spmv_worst_rank": round(ms, 5), "GFLOPs": round(2.0 * nnz / (ms * 1e-3) / 1e9, 2),
                           "kernel": "loops::kernels::merge_path_spmv_fused" + ("_phased" if phased_won else ""),
                           "tile": args.tile, "ms_plain_gathers": round(ms_by[0], 5), "ms_phased_gathers": round(ms_by[1], 5) if len(ms_by) > 1 else None,
                           "note": "SpMV only (no exchange), start-up probe outside the timed region"}
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
