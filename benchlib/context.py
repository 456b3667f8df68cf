"""bench.py, workload side: the whole matrix on one device, launch timing, and the context measurements of the N = 1 record
(the same matrix on one GPU for N > 1, BASELINE C4, the C3 stand-ins, the other schedules on C2)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .counters import HBM_PEAK_GBPS, VARIANT_PHASED, algorithmic_bytes


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
    for name, mode in (("mfma", 1), ("thread_per_block_row", 0), ("merge_path_one_shot", 4)):
        ms = timed_ms(torch, lambda: S.bcsr_thread_mapped(b, x, y, mfma=mode), iters)
        out[name] = {"avg_launch_ms": round(ms, 5), "GFLOPs": round(2 * 16 * nb / ms / 1e6, 1), "achieved_GBps": round(abytes / ms / 1e6, 1),
                     "frac": round(abytes / ms / 1e6 / HBM_PEAK_GBPS, 4)}
    out["merge_path_one_shot"]["note"] = ("loops::kernels::bcsr4x4_mfma_merge_path + fix-up: the load-balanced one-shot form (equal tiles "
                                          "of block-row ends + blocks) -- nothing to balance on C4's uniform block-rows; 47 x the MFMA kernel on skewed lengths "
                                          "(profiles/r06_bcsr_band_c4_experiments.txt, section 7)")
    # mode "tuned" (what the drop-in wrapper bcsr_thread_mapped<4, 4> launches): the kernel is chosen by the class of the block-row lengths,
    # found by a probe on the first call and remembered -- C4's are even, so after the first call this IS the MFMA kernel
    S.bcsr_thread_mapped(b, x, y, mfma="tuned")
    torch.cuda.synchronize()
    ms = timed_ms(torch, lambda: S.bcsr_thread_mapped(b, x, y, mfma="tuned"), iters)
    out["tuned_mode"] = {"avg_launch_ms": round(ms, 5), "frac": round(abytes / ms / 1e6 / HBM_PEAK_GBPS, 4),
                         "block_row_lengths": S.bcsr_row_length_class(b),
                         "note": "even lengths -> bcsr4x4_mfma_spmv, skewed -> bcsr4x4_mfma_merge_path (kernels::bcsr_row_length_class)"}
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
        shapes = plan.tune(30)
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
    # the reference's own bcsr_thread_mapped<4, 4> (algorithms/spmv/bcsr_thread_mapped.cuh:36-123, its HIP backend) on this GPU and matrix:
    # oracle/_ref/libloops_ref_gpu.so, built by oracle/Makefile where /root/reference is mounted; kernel-only time, best of 10
    so = os.path.join(ROOT, "oracle", "_ref", "libloops_ref_gpu.so")
    if os.path.exists(so):
        try:
            import ctypes as C
            from loops_amd import _lib
            R = _lib.load_shared(so)
            yr = np.zeros(nbr * 4, np.float32)
            ms = C.c_float()
            p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
            rc = R.refgpu_bcsr_spmv(4, 0, C.c_long(nbr * 4), C.c_long(nbr * 4), C.c_long(nbr), C.c_long(nbr), C.c_long(nb), p(boff), p(bcols), p(bvals),
                                    p(xh), p(yr), 10, C.byref(ms))
            out["reference_hip_backend_on_this_gpu"] = {
                "rc": rc, "best_kernel_ms": round(ms.value, 5), "frac": round(abytes / ms.value / 1e6 / HBM_PEAK_GBPS, 4) if ms.value > 0 else None,
                "equals_oracle": bool(np.array_equal(yr, want)), "kernel": "the reference's __bcsr_thread_mapped<4, 4> (thread per block-row)"}
        except Exception as e:  # noqa: BLE001
            out["reference_hip_backend_on_this_gpu"] = {"error": f"{type(e).__name__}: {e}"}
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
                "merge_path_flat_plan_less": lambda: S.spmv("merge_path_flat", csr, x, y),
                "merge_path_flat": lambda: S.merge_path_flat(csr, x, y, plan=mplan),
                "merge_path_flat_phased_gathers": lambda: S.merge_path_flat(csr, x, y, plan=pplan, variant=VARIANT_PHASED)}
        # which kernels an entry launches on a matrix of this size (the one-shot entries decide by size: abi_csr.inc)
        launches = {"group_mapped": "group_mapped_spmv_publish + group_mapped_spmv_claims + group_mapped_fixup (one-kernel group_mapped_spmv_fused once "
                                    "the entry's memo says nothing was published)",
                    "work_oriented": "shares of ONE tile through merge_path_flat's one-shot launch (sampled columns, merge_path_spmv_fused_auto: plain or "
                                     "phased gathers decided on the device) -- the persistent work_oriented_spmv_fused only below the sampling threshold (x < 3 MB)",
                    "merge_path_flat_plan_less": "what algorithms::spmv::merge_path_flat(csr, x, y) / loops_spmv_csr_f32(MERGE_PATH_FLAT) give: coordinate pre-pass + "
                                                 "merge_path_spmv_fused_auto<256, 16, 32> (gather order decided on the device from the remembered column sample) + fix-up",
                    "merge_path_flat": "merge_path_spmv_fused_planned<256, 8> (held plan, PLAIN gathers: variant 0 as asked for) + fix-up",
                    "merge_path_flat_phased_gathers": "merge_path_spmv_fused_phased_planned<256, 16, 32> (held plan) + fix-up"}
        for sched, fn in runs.items():
            ms = timed_ms(torch, fn, iters)
            res[sched] = {"ms_per_spmv": round(ms, 4), "GFLOPs": round(2.0 * nnz / ms / 1e6, 1), "achieved_GBps": round(abytes / ms / 1e6, 1),
                          "frac": round(abytes / ms / 1e6 / HBM_PEAK_GBPS, 4), "parity_vs_oracle_bit_exact": bool(np.array_equal(y.cpu().numpy(), ref)),
                          "launches": launches[sched]}
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
    # the plan-less entries, whole calls (coordinates rebuilt per call as the reference's wrappers do; the column sample behind the choice
    # of gather order is remembered per matrix) -- "merge_path_flat_one_shot" is what a caller of the reference's API gets without a plan
    for key, sched in (("merge_path_flat_one_shot", "merge_path_flat"), ("work_oriented", "work_oriented"), ("group_mapped", "group_mapped"),
                       ("thread_mapped", "thread_mapped"),   # (a thread owns whole rows, sums in the row's order -- the reference loop's bits)
                       ("flat_partitioned", "flat_partitioned")):  # (runs stitched per wavefront, one atomic per row and wavefront; y zero-filled by the entry)
        ms = timed_ms(torch, lambda: S.spmv(sched, csr, x, y), iters)
        out[key] = {"ms_per_spmv": round(ms, 5), "GFLOPs": round(2.0 * csr.nnzs / ms / 1e6, 1),
                    "frac": round(abytes / ms / 1e6 / HBM_PEAK_GBPS, 4), "equals_merge_path_y": bool(torch.equal(y, ref_y))}
    return out


