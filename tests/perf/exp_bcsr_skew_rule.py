"""Where the thread_mapped MFMA kernel (mode 1) loses to the merge-path tiles (mode 4) as block-row lengths spread: the data behind
kernels::bcsr_row_length_class (bcsr_merge_path.hxx).  2^18 block-rows, ~16 blocks per block-row on average, 4 x 4 fp32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S

nbr = nbc = 1 << 18
rng = np.random.default_rng(5)


def lengths(kind):
    if kind == "even16": return np.full(nbr, 16)
    if kind.startswith("uniform_"):                      # uniform in [16 - w, 16 + w]
        w = int(kind.split("_")[1]); return rng.integers(16 - w, 16 + w + 1, size=nbr)
    if kind.startswith("geometric"): return np.minimum(rng.geometric(1 / 16.0, size=nbr), 4096)
    if kind.startswith("pareto_"):                       # heavy tail, mean ~16, capped
        cap = int(kind.split("_")[1]); l = np.minimum((rng.pareto(1.5, size=nbr) * 5 + 1).astype(np.int64), cap); return l
    if kind.startswith("one_row_"):                      # even lengths + ONE block-row of L blocks (in the middle)
        l = np.full(nbr, 16); l[nbr // 2] = int(kind.split("_")[2]); return l
    if kind.startswith("last_row_"):                     # ... at the very end (nothing left to overlap its chain)
        l = np.full(nbr, 16); l[-1] = int(kind.split("_")[2]); return l
    raise KeyError(kind)


def ms(fn, iters=30):
    for _ in range(3): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3


kinds = sys.argv[1:] or ["even16", "uniform_4", "uniform_8", "uniform_12", "uniform_16", "geometric", "pareto_64", "pareto_256", "pareto_4096",
                         "one_row_64", "one_row_128", "one_row_256", "one_row_512", "one_row_2048", "last_row_128", "last_row_256", "last_row_512"]
for kind in kinds:
    lens = np.asarray(lengths(kind), np.int64)
    boff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nb = int(boff[-1])
    # (block columns: a random start per block-row + consecutive offsets scattered by a stride -- cheap to generate, distinct, sorted)
    start = rng.integers(0, nbc, size=nbr)
    within = np.arange(nb) - np.repeat(boff[:-1].astype(np.int64), lens)
    bcols = np.sort(((np.repeat(start, lens) + within * 40503) % nbc).reshape(-1)).astype(np.int32) if False else ((np.repeat(start, lens) + within * 40503) % nbc).astype(np.int32)
    bvals = (rng.integers(1, 9, size=nb * 16) / 8.0).astype(np.float32)
    xh = G.uniform_distribution_int(nbc * 4)
    b = S.BCSR(4, 4, nbr * 4, nbc * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
    x = torch.from_numpy(xh).cuda(); y = torch.empty(nbr * 4, device="cuda"); y2 = torch.empty(nbr * 4, device="cuda")
    t1 = ms(lambda: S.bcsr_thread_mapped(b, x, y, mfma=1))
    t4 = ms(lambda: S.bcsr_thread_mapped(b, x, y2, mfma="merge_path"))
    pad = lens.reshape(-1, 4)
    lock = 4 * pad.max(axis=1).sum() / nb
    print("%-14s blocks %8d longest %5d lockstep x%.2f  mfma %7.1f us  merge-path %7.1f us  ratio %.2f  class %-6s same=%s" % (
        kind, nb, lens.max(), lock, t1, t4, t1 / t4, S.bcsr_row_length_class(b), bool(torch.equal(y, y2))), flush=True)
