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
degree 2^14, fp32 (SURVEY 8d generator).  At N > 1 (weak scaling): N * 2^20 rows,
N * 2^24 nnz of the same generator, contiguous row ranges balanced by rows + nnz, one range
per GPU, x replicated, allgatherv(y) over RCCL every step.  At N > 1 a rank holds its shard
COLUMN-BLOCKED by owner (--layout, include/loops/kernels/column_blocked.hxx): x is N * 4 MB there and no
longer fits the 4 MB per-XCD L2; the blocked layout keeps each XCD inside one x block (same fused
kernel + a K-way row reduce).  The N = 1 headline runs on the unmodified CSR.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the roofline / cpu_baseline fields.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s measured float4 copy


def algorithmic_bytes(rows, cols, nnz, vbytes=4):
    # nnz * (4 + 4) + (rows + 1) * 4 + rows * 4 + cols * 4 for fp32
    return nnz * (4 + vbytes) + (rows + 1) * 4 + rows * vbytes + cols * vbytes


def pmc_traffic(args):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of
    THIS command (profiles/r01_c2_pmc_summary.json; separate --pmc FETCH_SIZE / WRITE_SIZE runs):
    (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE counts the 128-B requests of this kernel at
    64 B on gfx950 (MI355X_MICROARCH.md, HBM section; checked here against TCC_MISS * 128 B).
    None when the configuration differs from the profiled one."""
    path = os.path.join(ROOT, "profiles", "r01_c2_pmc_summary.json" if args.tile == "256x8"
                        else f"r01_c2_pmc_summary_{args.tile}.json")
    if not os.path.exists(path) or args.gpus != 1 or args.window or args.log2_rows != 20 or args.log2_nnz != 24 \
            or args.variant != 0 or args.layout == "blocked":
        return None
    d = json.load(open(path))
    for k, v in d.items():
        if "merge_path_spmv_fused" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            return int((2 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * 1024)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log2-rows", type=int, default=20, help="rows per GPU = 2^this (C2: 20)")
    ap.add_argument("--log2-nnz", type=int, default=24, help="nnz per GPU = 2^this (C2: 24)")
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
    ap.add_argument("--ref-gpu", action="store_true",
                    help="also time the reference's own HIP kernels on this GPU (oracle/_ref/libloops_ref_gpu.so)")
    ap.add_argument("--sweep", action="store_true", help="also time every compiled tile/variant (stderr)")
    ap.add_argument("--overlap-chunks", type=int, default=2,
                    help="N > 1: also try the step with the SpMV cut into this many row chunks whose exchanges overlap "
                         "the next chunk's kernel (0 = do not try); the fastest candidate of the start-up probe is used")
    ap.add_argument("--exchange", default="auto", choices=["auto", "p2p", "padded", "p2p-chunked"],
                    help="N > 1: allgatherv implementation; auto = the fastest of the start-up probe")
    ap.add_argument("--layout", default="auto", choices=["auto", "csr", "blocked"],
                    help="how a rank holds its row-range shard: 'csr' as sliced; 'blocked' = column-blocked by owner "
                         "(x of N x 4 MB does not fit the per-XCD L2: include/loops/kernels/column_blocked.hxx); "
                         "auto = csr at N = 1 (the headline is the unmodified CSR), blocked at N > 1")
    args = ap.parse_args()

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
    rows = world << args.log2_rows
    cols = rows
    nnz = world << args.log2_nnz
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
    if args.tile == "auto":  # measured launch box: every compiled tile shape timed on this shard, outside the timed region
        best, tile_probe = S.autotune_merge_path(csr, x, repeats=30)
        if world > 1:  # one shape for the whole job: the one with the smallest worst-rank time
            names = sorted(tile_probe)
            t = torch.tensor([tile_probe[n] for n in names], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            best = names[int(torch.argmin(t))]
            tile_probe = {n: float(v) for n, v in zip(names, t.tolist())}
        # the library's default shape unless another one is measurably (> 1 %) faster: shapes within the run-to-run
        # noise of the probe must not flip the kernel (and its profile) between runs
        if "512x8" in tile_probe and tile_probe["512x8"] <= 1.01 * tile_probe[best]:
            best = "512x8"
        args.tile = best
        tile_probe = {k: round(v, 5) for k, v in tile_probe.items()}
    plan = S.MergePathPlan(csr, args.tile)
    layout = args.layout if args.layout != "auto" else ("csr" if world == 1 else "blocked")
    blocked = None
    if layout == "blocked":
        try:
            blocked = S.ColumnBlockedPlan(csr, block_bounds=P.column_block_bounds(bounds))
        except Exception as e:  # noqa: BLE001 -- a rank that cannot build the blocked copy keeps its CSR shard
            if args.layout == "blocked":
                raise
            print(f"[rank {rank}] column-blocked plan unavailable ({type(e).__name__}: {e}); using the CSR shard",
                  file=sys.stderr)
    torch.cuda.synchronize()

    gather_mode = {"mode": "p2p"}
    exchanges = {}

    def spmv_local():
        if blocked is not None:
            blocked.spmv(x, y_loc)
        else:
            S.merge_path_flat(csr, x, y_loc, plan=plan, variant=args.variant)

    chunked = {"plans": None, "exchange": None}

    def step():
        if gather_mode["mode"] == "p2p-chunked":
            for c, (sub_run, y_sub) in enumerate(chunked["plans"]):
                sub_run(y_sub)
                chunked["exchange"].post(c)
            chunked["exchange"].finish()
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
                pl = S.ColumnBlockedPlan(sub, block_bounds=P.column_block_bounds(bounds))
                run = (lambda pl: (lambda y_sub: pl.spmv(x, y_sub)))(pl)
            else:
                pl = S.MergePathPlan(sub, args.tile)
                run = (lambda sub, pl: (lambda y_sub: S.merge_path_flat(sub, x, y_sub, plan=pl, variant=args.variant)))(sub, pl)
            keep.append((sub, pl))
            plans.append((run, y_sub))
        chunked["plans"], chunked["keep"] = plans, keep
        chunked["exchange"] = P.ChunkedAllgatherv(y_full, shard, cb)

    comm_dev = "cuda" if args.backend == "nccl" else "cpu"

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
        t = torch.tensor([(time.perf_counter() - t0) / timed * 1e3], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return round(float(t), 5)

    exchange_probe = None
    if world > 1:
        # The exchange is an allgatherv(y).  Two implementations (loops_amd/partition.py): "p2p" = one grouped
        # batch of direct sends / receives (every xGMI link at once), "padded" = the library all_gather on
        # max-count slots + local compaction.  Which one is faster depends on the RCCL build and on how much
        # host time a grouped p2p launch costs, so both are timed here, outside the timed region, and every
        # rank adopts the same winner (a mode that raises on any rank is excluded everywhere).
        exchange_probe = {}
        for mode in ("p2p", "padded"):
            ok = 1.0
            try:
                exchanges[mode] = P.Allgatherv(y_full, shard, mode)
                gather_mode["mode"] = mode
                step()
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] allgatherv mode {mode} unavailable ({type(e).__name__}: {e})", file=sys.stderr)
                ok = 0.0
            flag = torch.tensor([ok], device=comm_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag) < 1.0:
                exchanges.pop(mode, None)
                continue
            exchange_probe[mode] = probe_ms()
        assert exchange_probe, "no allgatherv implementation works on this backend"
        if args.overlap_chunks >= 2 and "p2p" in exchange_probe:
            # third candidate: the same p2p exchange, posted per row chunk so that it overlaps the next chunk's kernel
            ok = 1.0
            try:
                build_chunked(args.overlap_chunks)
                gather_mode["mode"] = "p2p-chunked"
                step()
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] chunked overlap unavailable ({type(e).__name__}: {e})", file=sys.stderr)
                ok = 0.0
            flag = torch.tensor([ok], device=comm_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag) >= 1.0:
                exchange_probe["p2p-chunked"] = probe_ms()
        gather_mode["mode"] = min(exchange_probe, key=exchange_probe.get)
        if args.exchange != "auto":
            assert args.exchange in exchange_probe, f"--exchange {args.exchange} is not available here: {exchange_probe}"
            gather_mode["mode"] = args.exchange

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ parity (outside timing)
    parity = None
    if not args.no_check:
        from oracle import oracle as O  # checker only
        step()
        torch.cuda.synchronize()
        ref = O.spmv_f32(off, idx, val, x_h, omp=True)
        got = y_loc.cpu().numpy()
        parity = bool(np.array_equal(got, ref))
        if world > 1:  # the gathered vector: checksum of all ranks' oracle results
            s = torch.tensor([float(ref.astype(np.float64).sum())], dtype=torch.float64,
                             device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(s)
            parity = parity and abs(float(y_full.double().sum()) - float(s)) == 0.0
        assert parity, "GPU result differs from the oracle"

    # ------------------------------------------------------------------ timed region
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ms_per_step = elapsed / args.steps * 1e3
    gflops = 2.0 * nnz / (ms_per_step * 1e-3) / 1e9

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
    k_reduce_avg = None
    if blocked is not None:
        k_main_avg, k_main_med = event_time(lambda: blocked.spmv_stage(0, x, y_loc), iters)
        k_main_single = k_main_avg
        k_main_avg = batch_event_time(lambda: blocked.spmv_stage(0, x, y_loc), iters)
        k_fix_avg, _ = event_time(lambda: blocked.spmv_stage(1, x, y_loc), iters)
        k_reduce_avg, _ = event_time(lambda: blocked.spmv_stage(2, x, y_loc), iters)
    else:
        k_main_avg, k_main_med = event_time(lambda: S.merge_path_flat_stage(csr, x, y_loc, plan, 0, args.variant), iters)
        k_main_single = k_main_avg  # an event pair per launch (kept in the output for comparison)
        k_main_avg = batch_event_time(lambda: S.merge_path_flat_stage(csr, x, y_loc, plan, 0, args.variant), iters)
        k_fix_avg, _ = event_time(lambda: S.merge_path_flat_stage(csr, x, y_loc, plan, 1, args.variant), iters)

    # the SpMV of a step without the exchange (BASELINE C5: "kernel-only and kernel + allgatherv")
    for _ in range(5):
        spmv_local()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        spmv_local()
    torch.cuda.synchronize()
    spmv_only_ms = (time.perf_counter() - t0) / iters * 1e3

    ms_with_prepass = None
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
        ms_with_prepass = (time.perf_counter() - t0) / iters * 1e3

    # for context at N = 1: the same SpMV with the matrix held column-blocked (never `value`)
    blocked_info = None
    if world == 1 and blocked is None and rank == 0:
        cb = S.ColumnBlockedPlan(csr, block_bounds=P.column_block_bounds(bounds))
        yb = torch.empty_like(y_loc)
        for _ in range(5):
            cb.spmv(x, yb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            cb.spmv(x, yb)
        torch.cuda.synchronize()
        ms_b = (time.perf_counter() - t0) / iters * 1e3
        kb_avg, _ = event_time(lambda: cb.spmv_stage(0, x, yb), iters)
        cb.spmv(x, yb)
        torch.cuda.synchronize()
        traffic_b = None
        pb = os.path.join(ROOT, "profiles", "r01_c2_blocked_pmc_summary.json")
        if os.path.exists(pb) and pmc_traffic(args) is not None:  # same configuration as the profiled one
            for k, v in json.load(open(pb)).items():
                if "merge_path_spmv_fused_stacked" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                    traffic_b = int((2 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * 1024)
        ab = algorithmic_bytes(csr.rows, cols, csr.nnzs)
        blocked_info = {"blocks": cb.num_blocks, "ms_per_step": round(ms_b, 5),
                        "roofline": {"kernel": "loops::kernels::merge_path_spmv_fused_stacked", "avg_launch_ms": round(kb_avg, 5),
                                     "achieved": round(ab / (kb_avg * 1e-3) / 1e9, 1), "unit": "GB/s",
                                     "frac": round(ab / (kb_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": traffic_b},
                        "GFLOPs": round(2.0 * nnz / (ms_b * 1e-3) / 1e9, 2), "equal_to_csr_result": bool(torch.equal(yb, y_loc)),
                        "note": "plan-time re-ordered copy of the matrix (include/loops/kernels/column_blocked.hxx); "
                                "same fused kernel + K-way row reduce; not the headline"}
        cb.close()

    # for context at N = 1: the SAME kernel on a matrix of the same size whose columns are local (16 per row inside a
    # 64-column band): what the kernel does when the x gather is served by L1 -- its roofline fraction as a kernel,
    # next to the headline's, which is set by the random gather (never `value`)
    local_info = None
    if world == 1 and rank == 0 and not args.window and not args.no_local_context:
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
        local_info = {"workload": f"{l_csr.rows} rows x {l_csr.nnzs // l_csr.rows} nnz, columns in a 64-wide band, fp32, held plan 256x8"
                                  + (" (self-completing: one kernel)" if l_plan.self_complete else ""),
                      "avg_launch_ms": round(l_avg, 5), "GFLOPs": round(2.0 * l_csr.nnzs / (l_avg * 1e-3) / 1e9, 1),
                      "roofline": {"bound": "hbm", "achieved": round(lb / (l_avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                                   "unit": "GB/s", "frac": round(lb / (l_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
                      "equal_to_thread_mapped_result": l_ok, "note": "context, not the headline workload"}
        del l_csr, l_plan, yl, y_tm

    # calibration probes: achievable streaming rate and gather rate on this box
    n_copy = 1 << 28  # 1 GiB in + 1 GiB out: beyond the 256 MiB Infinity Cache
    src = torch.empty(n_copy, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    copy_avg = batch_event_time(lambda: PR.stream_copy(src, dst), 20)
    copy_gbps = 2 * n_copy * 4 / (copy_avg * 1e-3) / 1e9
    del src, dst
    gidx = torch.from_numpy(idx[: 1 << 24]).cuda() if idx.size >= 1 << 24 else csr.indices
    gout = torch.empty(gidx.numel(), dtype=torch.float32, device="cuda")
    gat_avg = batch_event_time(lambda: PR.gather(x, gidx, gout), 20)
    gather_gps = gidx.numel() / (gat_avg * 1e-3) / 1e9

    loc_rows, loc_nnz = csr.rows, csr.nnzs
    abytes = algorithmic_bytes(loc_rows, cols, loc_nnz)
    achieved = abytes / (k_main_avg * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": "loops::kernels::merge_path_spmv_fused", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": pmc_traffic(args),
                "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": round(k_main_avg, 5),
                "median_launch_ms": round(k_main_med, 5), "avg_launch_ms_event_pair_per_launch": round(k_main_single, 5), "fixup_avg_launch_ms": round(k_fix_avg, 5),
                "measured_copy_GBps": round(copy_gbps, 1), "frac_of_measured_copy": round(achieved / copy_gbps, 4),
                "measured_gather_Gelem_per_s": round(gather_gps, 2),
                # time the x gathers of this shard alone need at the measured random-gather rate of this box
                # (same indices, same x, no streams) over the kernel's time: how much of the kernel is the gather
                "gather_only_ms": round(loc_nnz / gather_gps / 1e6, 5),
                "gather_only_over_kernel": round(loc_nnz / gather_gps / 1e6 / k_main_avg, 4)}
    if k_reduce_avg is not None:
        roofline["block_reduce_avg_launch_ms"] = round(k_reduce_avg, 5)

    if args.sweep and rank == 0:
        for tile in ("256x8", "256x7", "128x7", "512x8", "256x16"):
            p2 = S.MergePathPlan(csr, tile)
            for variant in (0, 1, 2, 3):
                avg, med = event_time(lambda: S.merge_path_flat_stage(csr, x, y_loc, p2, 0, variant), 50)
                print(f"[sweep] tile={tile} variant={variant} main kernel avg {avg*1e3:.1f} us med {med*1e3:.1f} us "
                      f"-> {abytes/avg/1e6:.0f} GB/s", file=sys.stderr)

    # ------------------------------------------------------------------ the reference's own HIP path on this GPU
    ref_gpu = None
    so = os.path.join(ROOT, "oracle", "_ref", "libloops_ref_gpu.so")
    if args.ref_gpu and rank == 0 and world == 1 and os.path.exists(so):
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

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
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
        cpu = {"value": round(2.0 * loc_nnz / t1 / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
               "sample": f"{reps1} full passes of the same C2 matrix ({loc_rows} rows, {loc_nnz} nnz), "
                         "oracle/loops_oracle.c oracle_spmv_f32 (restatement of reference::spmv, "
                         f"util/reference.hxx:57-76), gcc -O3 -march=x86-64-v3; {reps1 * t1:.1f} s of CPU work",
               "all_cores": {"value": round(2.0 * loc_nnz / tn / 1e9, 3), "cores": int(threads),
                             "note": f"same loop, OpenMP row-parallel schedule(dynamic,1024), {repsn} passes"}}

    if rank == 0:
        out = {
            "metric": "CSR SpMV GFLOP/s, merge_path_flat", "value": round(gflops, 2), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic power-law CSR, {rows} rows / {nnz} nnz total "
                                   f"({world} x 2^{args.log2_rows} rows / 2^{args.log2_nnz} nnz per GPU), max degree 2^14, "
                                   "fp32, merge_path_flat" + (f", columns banded (window {args.window})" if args.window else ", columns uniform")
                                   + (f", row-range sharded + allgatherv(y) over {'RCCL' if args.backend == 'nccl' else 'gloo (functional test)'}" if world > 1 else ""),
                       "baseline_config": "BASELINE.json configs[1]" if world == 1 else "configs[1] per GPU (weak scaling)",
                       "tile": args.tile, "tile_autotune_ms": tile_probe, "variant": args.variant,
                       "merge_tiles_per_gpu": plan.num_tiles,
                       "shard_layout": "csr" if blocked is None else
                                       f"column-blocked by owner, {blocked.num_blocks} blocks (x per GPU {cols * 4 >> 20} MB)",
                       "step_includes": "fused merge-tile kernel + carry-out fix-up" + (" + block reduce" if blocked is not None else "") + (f" + allgatherv(y) [{gather_mode['mode']}" + (f", {args.overlap_chunks} chunks overlapping the SpMV" if gather_mode['mode'] == 'p2p-chunked' else "") + "]" if world > 1 else ""),
                       "ms_per_step_with_prepass": None if ms_with_prepass is None else round(ms_with_prepass, 5),
                       "achieved_GBps_whole_step": round(algorithmic_bytes(rows, cols, nnz) / world / (ms_per_step * 1e-3) / 1e9, 1),
                       "spmv_only_ms_per_step": round(spmv_only_ms, 5),
                       "allgatherv_probe_ms_per_step": exchange_probe,
                       "parity_vs_oracle_bit_exact": parity, "generate_seconds": round(gen_s, 1),
                       "column_blocked_layout_same_matrix": blocked_info,
                       "same_kernel_local_columns": local_info,
                       "reference_hip_backend_on_this_gpu": ref_gpu},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
