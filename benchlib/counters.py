"""bench.py, record side: algorithmic bytes, the HBM peak, and the committed rocprofv3 counter summaries of the headline kernel
(which file, whether it was collected at the kernel sources that run now, HBM-side traffic, what the counters say bounds it)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


