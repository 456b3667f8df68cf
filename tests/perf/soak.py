"""Soak: the tuned kernels launched back to back for a fixed wall time on C2 and on a self-completing band matrix; every
result compared ON THE GPU with the first one (bit-equal) -- races / ordering bugs show up as a mismatch count > 0.
usage: python tests/perf/soak.py [seconds per case, default 20] [r2|r3|r4|r5|r6]     (rN = only the kernels added in that round)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from oracle import oracle as O

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
only_r2 = len(sys.argv) > 2 and sys.argv[2] == "r2"
only_r3 = len(sys.argv) > 2 and sys.argv[2] == "r3"
rows = cols = 1 << 20


def soak(name, label, fn, ref, n_out):
    y = torch.empty(n_out, device="cuda")
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    rounds, t0 = 0, time.time()
    while time.time() - t0 < secs:
        for _ in range(200):
            y.fill_(float("nan"))
            extra = fn(y)
            bad += (y != ref).any()
            for e in extra or ():
                bad += (e != ref).any()
        rounds += 200
        torch.cuda.synchronize()
    print(f"{name:7s} {label:46s} rounds {rounds:7d} mismatching rounds {int(bad.item())}", flush=True)


# ---- round 3: panel-binned (plain and fan-out; reproducibility of its LDS accumulation is the point), the stitched
# flat_partitioned (atomics: exactly summable inputs make every order give the same bits), coalesced BCSR, the measured SpMV plan
def round3():
    off, idx, val = G.csr_from_degrees(G.powerlaw_degrees(rows, 1 << 24), cols, 1)
    xh = G.uniform_distribution_int(cols)
    x = torch.from_numpy(xh).cuda()
    ref = torch.from_numpy(O.spmv_f32(off, idx, val, xh, omp=True)).cuda()
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    pb = S.PanelBinnedPlan(csr)
    peers = [torch.empty(rows, device="cuda") for _ in range(2)]
    def fan_panel(y):
        for p in peers: p.fill_(float("nan"))
        pb.spmv_fanout(x, y, peers)
        return peers
    soak("c2", "panel_binned (products + sub-band reduce)", lambda y: (pb.spmv(x, y), None)[1], ref, rows)
    soak("c2", "panel_binned + fan-out (2 peers)", fan_panel, ref, rows)
    soak("c2", "flat_partitioned (wavefront-stitched)", lambda y: (S.spmv("flat_partitioned", csr, x, y), None)[1], ref, rows)
    sp = S.SpmvPlan(csr, allow_copy=True, measure=True, repeats=5)
    soak("c2", f"held SpMV plan ({sp.layout}, {sp.tile})", lambda y: (sp.spmv(x, y), None)[1], ref, rows)
    # realistic values: the panel-binned result must be the same bits every time (no order depends on timing)
    off2, idx2, val2 = G.csr_from_degrees(G.powerlaw_degrees(rows, 1 << 24), cols, 1, 0, False)
    csr2 = S.CSR.from_numpy(rows, cols, off2, idx2, val2)
    xr = torch.from_numpy(G.realistic_x(cols)).cuda()
    pb2 = S.PanelBinnedPlan(csr2)
    first = pb2.spmv(xr).clone()
    soak("c2real", "panel_binned, real values: run-to-run bit identity", lambda y: (pb2.spmv(xr, y), None)[1], first, rows)
    del pb, pb2, sp, csr, csr2
    for R, dt in ((2, np.float32), (3, np.float32), (8, np.float32)):
        nbr = 1 << 16
        boff, bcols, _ = G.uniform_bcsr(nbr, nbr, 16, R, R)
        bvals = (np.random.default_rng(R).integers(1, 9, size=bcols.size * R * R) / 8.0).astype(dt)
        xb = torch.from_numpy(G.uniform_distribution_int(nbr * R)).cuda()
        b = S.BCSR(R, R, nbr * R, nbr * R, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
        want = S.bcsr_thread_mapped(b, xb, mfma="thread").clone()
        soak("bcsr", f"coalesced BCSR {R}x{R}, 2^16 block-rows x 16", lambda y, b=b, xb=xb: (S.bcsr_thread_mapped(b, xb, y, mfma="tuned"), None)[1], want, nbr * R)


# ---- round 4: phased x gathers (clock-aligned passes with a workgroup barrier each: the product must equal the plain kernel's
# bit for bit on every launch, exactly summable AND real values), 8 / 16 / 32 parts, both tile shapes, fp32 and fp64
def round4():
    from loops_amd import _lib
    only_self = len(sys.argv) > 3 and sys.argv[3] == "self"
    for lr, ln in ((20, 24), (21, 25), (23, 26), (-20, 24), (-23, 25)):  # (negative: uniform degrees -> self-completing plans)
        if only_self and lr > 0:
            continue
        uniform = lr < 0
        lr = abs(lr)
        r = c = 1 << lr
        deg = np.full(r, (1 << ln) >> lr, np.int64) if uniform else G.powerlaw_degrees(r, 1 << ln)
        for exact in (True, False):
            off, idx, val = G.csr_from_degrees(deg, c, 1, 0, exact)
            xh = G.uniform_distribution_int(c) if exact else G.realistic_x(c)
            for dt in (np.float32, np.float64):
                if dt is np.float64 and lr != 20:
                    continue
                csr = S.CSR.from_numpy(r, c, off, idx, val.astype(dt))
                x = torch.from_numpy(xh.astype(dt)).cuda()
                for tile in ("512x8", "256x16"):
                    plan = S.MergePathPlan(csr, tile)
                    ref = S.merge_path_flat(csr, x, plan=plan, variant=0).clone()
                    def run(y, csr=csr, x=x, plan=plan):
                        S.merge_path_flat(csr, x, y, plan=plan, variant=_lib.VARIANT_PHASED)
                    y_dtype = ref.dtype
                    def soak_t(name, label, fn, ref, n_out):  # (soak() with the value type of this case)
                        y = torch.empty(n_out, device="cuda", dtype=y_dtype)
                        bad = torch.zeros((), dtype=torch.int64, device="cuda")
                        rounds, t0 = 0, time.time()
                        while time.time() - t0 < secs:
                            for _ in range(100):
                                y.fill_(float("nan"))
                                fn(y)
                                bad += (y != ref).any()
                            rounds += 100
                            torch.cuda.synchronize()
                        print(f"{name:9s} {label:60s} rounds {rounds:7d} mismatching rounds {int(bad.item())}", flush=True)
                    soak_t(f"2^{lr}" + ("u" if uniform else ""), f"phased gathers {tile} {np.dtype(dt).name} {'exact' if exact else 'real'} values vs the plain kernel", run, ref, r)
                    plan.close()
                del csr


def round4_planless():
    """The asynchronous plan-less entry at |x| = 8 MB (the device samples the columns, merge_path_spmv_fused_auto takes the
    plain or the phased path): every launch against the held plain plan's result, scattered and banded columns."""
    r = c = 1 << 21
    deg = G.powerlaw_degrees(r, 1 << 25)
    for tag, window in (("scattered", None), ("band8192", 8192)):
        off, idx, val = G.csr_from_degrees(deg, c, 1, 0, False, window)
        csr = S.CSR.from_numpy(r, c, off, idx, val)
        x = torch.from_numpy(G.realistic_x(c)).cuda()
        plan = S.MergePathPlan(csr, "512x8")
        ref = S.merge_path_flat(csr, x, plan=plan, variant=0).clone()
        soak("2^21", f"plan-less merge_path_flat (device-decided), {tag}, real values", lambda y: (S.spmv("merge_path_flat", csr, x, y), None)[1], ref, r)
        del csr, plan


# ---- round 5: the row-band layout -- LDS fp64 atomics from 8 / 16 wavefronts, hub replicas, partial vectors + combine, DPP prefix
# sums of the column deltas; exactly summable inputs (any order gives the same bits) and real values (run-to-run identity)
def round6():
    """Round 6: the block-band BCSR plan (uncut and cut bands, several kernel shapes), group_mapped with shared-out groups (R-MAT in
    generator order, hub rows on group boundaries; the one-shot entry's memo flips between the two forms on the way), the row-band
    layout with 8-byte values and its dense placement path."""
    nbr = 1 << 18
    boff, bcols, bvals = G.uniform_bcsr(nbr, nbr, 16)
    xh4 = G.uniform_distribution_int(nbr * 4)
    ref4 = torch.from_numpy(O.bcsr_spmv_f32(4, 4, nbr * 4, boff, bcols, bvals, xh4)).cuda()
    b = S.BCSR(4, 4, nbr * 4, nbr * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
    x4 = torch.from_numpy(xh4).cuda()
    for label, hb, chunks, shape in (("automatic (HB 1024 uncut), 16 x 1", 0, 0, (16, 1, 0)), ("HB 4096 cut in 4, 16 x 1 nt", 4096, 0, (16, 1, 1)),
                                     ("HB 1024, 8 x 4", 1024, 0, (8, 4, 0)), ("HB 2048, 600 chunks, 16 x 2", 2048, 600, (16, 2, 1))):
        plan = S.BCSRBandPlan(b, hb, chunks)
        plan.set_shape(*shape)
        soak("c4", "block_band " + label, lambda y: (plan.spmv(x4, y), None)[1], ref4, nbr * 4)
        plan.close()
    soak("c4", "bcsr merge-path one-shot (mode 4)", lambda y: (S.bcsr_thread_mapped(b, x4, y, mfma="merge_path"), None)[1], ref4, nbr * 4)
    soak("c4", "bcsr mode 3 (probe -> MFMA kernel)", lambda y: (S.bcsr_thread_mapped(b, x4, y, mfma="tuned"), None)[1], ref4, nbr * 4)
    del b
    rng = np.random.default_rng(3)                                       # 64 hub block-rows of 16 384 blocks among 2^17 of 8
    nbh = 1 << 17
    lens = np.full(nbh, 8, np.int64)
    lens[rng.choice(nbh, size=64, replace=False)] = 16384
    boffh = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    bcolsh = np.concatenate([np.sort(rng.choice(nbh, size=int(n), replace=False)) for n in lens]).astype(np.int32)
    bvalsh = (rng.integers(1, 9, size=bcolsh.size * 16) / 8.0).astype(np.float32)
    xhh = G.uniform_distribution_int(nbh * 4)
    refh = torch.from_numpy(O.bcsr_spmv_f32(4, 4, nbh * 4, boffh, bcolsh, bvalsh, xhh)).cuda()
    bh = S.BCSR(4, 4, nbh * 4, nbh * 4, torch.from_numpy(boffh).cuda(), torch.from_numpy(bcolsh).cuda(), torch.from_numpy(bvalsh).cuda())
    xh4 = torch.from_numpy(xhh).cuda()
    soak("hubs", "bcsr merge-path one-shot, hub block-rows", lambda y: (S.bcsr_thread_mapped(bh, xh4, y, mfma="merge_path"), None)[1], refh, nbh * 4)
    soak("hubs", "bcsr mode 3 (probe -> merge-path tiles)", lambda y: (S.bcsr_thread_mapped(bh, xh4, y, mfma="tuned"), None)[1], refh, nbh * 4)
    planh = S.BCSRBandPlan(bh)
    soak("hubs", f"block_band automatic (HB {planh.HB}, {planh.num_chunks} chunks, replicas)", lambda y: (planh.spmv(xh4, y), None)[1], refh, nbh * 4)
    planh.close()
    del bh
    off, idx, val = G.rmat_csr(20, 16, relabel="none")
    xh = G.uniform_distribution_int(cols)
    x = torch.from_numpy(xh).cuda()
    ref = torch.from_numpy(O.spmv_f32(off, idx, val, xh, omp=True)).cuda()
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    soak("rmat20", "group_mapped, heavy groups shared out", lambda y: (S.spmv("group_mapped", csr, x, y), None)[1], ref, rows)
    deg = G.powerlaw_degrees(rows, 1 << 24)
    deg[255] = deg[256] = deg[1 << 19] = 70_000                       # hub rows on both sides of a group boundary and mid-matrix
    off2, idx2, val2 = G.csr_from_degrees(deg, cols, 1)
    ref2 = torch.from_numpy(O.spmv_f32(off2, idx2, val2, xh, omp=True)).cuda()
    csr2 = S.CSR.from_numpy(rows, cols, off2, idx2, val2)
    soak("c2hubs", "group_mapped, three heavy groups among 4 096", lambda y: (S.spmv("group_mapped", csr2, x, y), None)[1], ref2, rows)
    del csr, csr2
    # the plan-less entries with the gather order decided on the device from a remembered sample: C2 (x = 4 MB: merge_path_flat /
    # work_oriented ask, group_mapped does not) and C2's rows over x = 8 MB with a hub row (group_mapped asks too, and shares out)
    offc, idxc, valc = G.powerlaw_csr(rows, cols, 1 << 24)
    refc = torch.from_numpy(O.spmv_f32(offc, idxc, valc, xh, omp=True)).cuda()
    csrc = S.CSR.from_numpy(rows, cols, offc, idxc, valc)
    for sched in ("merge_path_flat", "work_oriented"):
        soak("c2", sched + " plan-less (sample remembered)", lambda y, sched=sched: (S.spmv(sched, csrc, x, y), None)[1], refc, rows)
    del csrc
    deg8 = G.powerlaw_degrees(rows, 1 << 24)
    deg8[1000] = 90_000
    off8, idx8, val8 = G.csr_from_degrees(deg8, 2 * cols, 1)
    xh8 = G.uniform_distribution_int(2 * cols)
    x8 = torch.from_numpy(xh8).cuda()
    ref8 = torch.from_numpy(O.spmv_f32(off8, idx8, val8, xh8, omp=True)).cuda()
    csr8 = S.CSR.from_numpy(rows, 2 * cols, off8, idx8, val8)
    for sched in ("group_mapped", "merge_path_flat"):
        soak("c2x8", sched + " plan-less, x = 8 MB, one hub row", lambda y, sched=sched: (S.spmv(sched, csr8, x8, y), None)[1], ref8, rows)
    del csr8
    # one-shot CSC: the binned product (count / scan / scatter into bins of 4 096 rows, fp64 LDS sums) on C2's matrix and on R-MAT (shared bins)
    import scipy.sparse as sp
    for tag, (o_, i_, v_) in (("c2csc", G.powerlaw_csr(rows, cols, 1 << 24)), ("rmatcsc", G.rmat_csr(20, 16, relabel="none"))):
        m = sp.csr_matrix((v_.astype(np.float64), i_, o_), shape=(rows, cols))
        refc = torch.from_numpy((m @ xh.astype(np.float64)).astype(np.float32)).cuda()
        c = m.tocsc(); c.sort_indices()
        dc = [torch.from_numpy(a).cuda() for a in (c.indptr.astype(np.int32), c.indices.astype(np.int32), c.data.astype(np.float32))]
        soak(tag, "csc one-shot, binned products", lambda y: (S.csc_spmv(rows, cols, dc[0], dc[1], dc[2], x, y, tuned=True), None)[1], refc, rows)
    off3, idx3, val3 = G.csr_from_degrees(G.powerlaw_degrees(rows, 1 << 24), cols, 1)
    csr3 = S.CSR.from_numpy(rows, cols, off3, idx3, val3.astype(np.float64))
    x64 = torch.from_numpy(xh.astype(np.float64)).cuda()
    ref3 = torch.from_numpy(O.spmv_f64(off3, idx3, val3.astype(np.float64), xh.astype(np.float64))).cuda()
    for waves in (8, 16):
        rb = S.RowBandPlan(csr3)
        rb.set_waves(waves)
        y64 = torch.empty(rows, dtype=torch.float64, device="cuda")
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        n, t0 = 0, time.time()
        while time.time() - t0 < secs:
            for _ in range(200):
                y64.fill_(float("nan"))
                rb.spmv(x64, y64)
                bad += (y64 != ref3).any()
            n += 200
            torch.cuda.synchronize()
        print(f"{'c2f64':7s} {'row_band, 8-byte values, %d wavefronts' % waves:46s} rounds {n:7d} mismatching rounds {int(bad.item())}", flush=True)
        rb.close()


def round5():
    off, idx, val = G.csr_from_degrees(G.powerlaw_degrees(rows, 1 << 24), cols, 1)
    xh = G.uniform_distribution_int(cols)
    x = torch.from_numpy(xh).cuda()
    ref = torch.from_numpy(O.spmv_f32(off, idx, val, xh, omp=True)).cuda()
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    peers = [torch.empty(rows, device="cuda") for _ in range(2)]
    for label, band_rows, target, waves in (("automatic (H 16384, 256 chunks)", 0, 0, 8), ("H 8192, 256 chunks, 16 wavefronts", 8192, 256, 16),
                                            ("H 4096, uncut", 4096, 256, 8), ("H 16384, 1000 chunks", 16384, 1000, 8)):
        rb = S.RowBandPlan(csr, band_rows, target)
        rb.set_waves(waves)
        soak("c2", "row_band " + label, lambda y: (rb.spmv(x, y), None)[1], ref, rows)
        rb.close()
    rb = S.RowBandPlan(csr)
    def fan(y):
        for p in peers: p.fill_(float("nan"))
        rb.spmv_fanout(x, y, peers)
        return peers
    soak("c2", "row_band + fan-out (2 peers)", fan, ref, rows)
    sp = S.SpmvPlan(csr, allow_copy=True, measure=True, repeats=5)
    soak("c2", f"held SpMV plan ({sp.layout})", lambda y: (sp.spmv(x, y), None)[1], ref, rows)
    rb.close(); sp.close()
    off2, idx2, val2 = G.csr_from_degrees(G.powerlaw_degrees(rows, 1 << 24), cols, 1, 0, False)
    csr2 = S.CSR.from_numpy(rows, cols, off2, idx2, val2)
    xr = torch.from_numpy(G.realistic_x(cols)).cuda()
    for target in (0, 1000):
        rb2 = S.RowBandPlan(csr2, 0, target)
        first = rb2.spmv(xr).clone()
        soak("c2real", f"row_band, real values, {rb2.num_chunks} chunks: run-to-run bit identity", lambda y: (rb2.spmv(xr, y), None)[1], first, rows)
        rb2.close()
    del csr, csr2
    off3, idx3, val3 = G.csr_from_degrees(G.powerlaw_degrees(rows, 1 << 24), cols, 1, 0, True, 8192)
    ref3 = torch.from_numpy(O.spmv_f32(off3, idx3, val3, xh, omp=True)).cuda()
    csr3 = S.CSR.from_numpy(rows, cols, off3, idx3, val3)
    rb3 = S.RowBandPlan(csr3)
    rb3.set_waves(16)
    soak("band", "row_band, columns in an 8192-wide band, 16 wavefronts", lambda y: (rb3.spmv(x, y), None)[1], ref3, rows)
    rb3.close()
    del csr3
    # a graph whose first band holds most nonzeros (R-MAT relabelled by degree): that band is cut into 100+ chunks and its partial
    # vectors are added by 16 threads per group of rows (rowband_combine_wide)
    off4, idx4, val4 = G.rmat_csr(20, 16, relabel="degree")
    ref4 = torch.from_numpy(O.spmv_f32(off4, idx4, val4, xh, omp=True)).cuda()
    csr4 = S.CSR.from_numpy(rows, cols, off4, idx4, val4)
    for target in (0, 90):
        rb4 = S.RowBandPlan(csr4, 0, target)
        pieces = int(rb4.arrays()[6][:, 2].max())
        soak("rmat", f"row_band, R-MAT by degree, {rb4.num_chunks} chunks, up to {pieces} per band", lambda y: (rb4.spmv(x, y), None)[1], ref4, rows)
        rb4.close()


if len(sys.argv) > 2 and sys.argv[2] == "r6":
    round6()
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "r5":
    round5()
    sys.exit(0)

if len(sys.argv) > 2 and sys.argv[2] == "r4":
    if len(sys.argv) > 3 and sys.argv[3] == "planless":
        round4_planless()
        sys.exit(0)
    round4()
    sys.exit(0)
if only_r3:
    round3()
    sys.exit(0)

# ---- round 2: ELL merge-path on the fused engine, DIA, epilogue fan-out (two stand-in peers on the same device)
off, idx, val = G.csr_from_degrees(np.minimum(G.powerlaw_degrees(1 << 18, 1 << 22, cap=256), 256), cols, 1)
r2 = off.size - 1
deg = np.diff(off)
pitch = int(deg.max())
ind = np.full((r2, pitch), -1, np.int32); ev = np.zeros((r2, pitch), np.float32)
m = np.arange(pitch)[None, :] < deg[:, None]
ind[m] = idx; ev[m] = val
xh = G.uniform_distribution_int(cols); x = torch.from_numpy(xh).cuda()
ref = torch.from_numpy(O.spmv_f32(off, idx, val, xh, omp=True)).cuda()
di, dv = torch.from_numpy(ind).cuda(), torch.from_numpy(ev).cuda()
soak("ell", "ell_merge_path (fused engine), 2^18 x pitch 256", lambda y: (S.ell_spmv(r2, cols, pitch, di, dv, x, y, tuned="merge_path"), None)[1], ref, r2)
del di, dv
offs = np.arange(-16, 17, dtype=np.int32)
n = 1 << 20
cells = ((np.arange(offs.size * n, dtype=np.int64).reshape(offs.size, n) * 7 % 8 + 1) / 8.0).astype(np.float32)
wantd = np.zeros(n, np.float64)
for k, o in enumerate(offs):
    r = np.arange(max(0, -o), min(n, n - o))
    wantd[r] += cells[k, r].astype(np.float64) * xh[r + o]
refd = torch.from_numpy(wantd.astype(np.float32)).cuda()
dd, dc = torch.from_numpy(offs).cuda(), torch.from_numpy(cells).cuda()
soak("dia", "dia_row4_spmv, 2^20 rows x 33 diagonals", lambda y: (S.dia_spmv(n, n, dd, dc, x, y, tuned=True), None)[1], refd, n)
del dd, dc
off, idx, val = G.csr_from_degrees(G.powerlaw_degrees(rows, 1 << 24), cols, 1)
ref = torch.from_numpy(O.spmv_f32(off, idx, val, xh, omp=True)).cuda()
csr = S.CSR.from_numpy(rows, cols, off, idx, val)
plan = S.MergePathPlan(csr, "512x8")
cb = S.RowBandPlan(csr)
peers = [torch.empty(rows, device="cuda") for _ in range(2)]
def fan_csr(y):
    for p in peers: p.fill_(float("nan"))
    S.merge_path_flat_fanout(csr, x, y, plan, peers)
    return peers
def fan_blocked(y):
    for p in peers: p.fill_(float("nan"))
    cb.spmv_fanout(x, y, peers)
    return peers
soak("c2", "merge_path_flat + epilogue fan-out (2 peers)", fan_csr, ref, rows)
soak("c2", "row-band + fan-out (2 peers)", fan_blocked, ref, rows)
del csr, plan, cb, peers
if only_r2:
    sys.exit(0)
round3()

cases = {"c2": (G.powerlaw_degrees(rows, 1 << 24), None), "band64": (np.full(rows, 16, np.int64), 64)}
for name, (deg, window) in cases.items():
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window)
    xh = G.uniform_distribution_int(cols)
    ref = torch.from_numpy(O.spmv_f32(off, idx, val, xh, omp=True)).cuda()
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    x = torch.from_numpy(xh).cuda()
    plans = {t: S.MergePathPlan(csr, t) for t in ("256x8", "512x8")}
    cb = S.RowBandPlan(csr)
    cb16 = S.RowBandPlan(csr, 0, 700)
    cb16.set_waves(16)
    kernels = {f"merge_path_flat {t} (self={int(p.self_complete)})": (lambda y, p=p: S.merge_path_flat(csr, x, y, plan=p)) for t, p in plans.items()}
    kernels["work_oriented"] = lambda y: S.spmv("work_oriented", csr, x, y)
    kernels["group_mapped"] = lambda y: S.spmv("group_mapped", csr, x, y)
    kernels["thread_mapped (assisted long rows)"] = lambda y: S.spmv("thread_mapped", csr, x, y)
    kernels["work_oriented, held plan"] = lambda y, p=plans["256x8"]: S.work_oriented(csr, x, y, plan=p)
    kernels["row_band"] = lambda y: cb.spmv(x, y)
    kernels["row_band, cut bands, 16 wavefronts"] = lambda y: cb16.spmv(x, y)
    for label, fn in kernels.items():
        y = torch.empty(rows, device="cuda")
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        rounds, t0 = 0, time.time()
        while time.time() - t0 < secs:
            for _ in range(200):
                y.fill_(float("nan"))
                fn(y)
                bad += (y != ref).any()
            rounds += 200
            torch.cuda.synchronize()
        print(f"{name:7s} {label:40s} rounds {rounds:7d} mismatching rounds {int(bad.item())}", flush=True)
