"""How much of a merge tile's x gathers an LDS window of x could serve (round-4 review, item 3: "an LDS window of x for tiles whose
columns are local").  Host-only analysis on the first 2^20 rows of the C3 stand-ins: for every 4096-nonzero tile of the 512 x 8
merge-path kernel the best-placed window of W columns (aligned to W / 4) and the share of the tile's nonzeros inside it; then the L2
requests a tile would issue with the window (window lines + the gathers outside it) against one per nonzero.
usage: python scripts/analyze_lds_window.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from loops_amd import generate as G

rows_all = cols = 7_414_866
nnz_all = 194_109_311
deg = G.powerlaw_degrees(rows_all, nnz_all)
R = 1 << 20
TILE = 4096
for tag, window in (("host-blocked", G.HOST_BLOCKED), ("band 65536", 65536), ("uniform", None)):
    off, idx, val = G.csr_from_degrees(deg[:R], cols, 1, 0, True, window, hosts=G.host_blocks(cols) if window == G.HOST_BLOCKED else None)
    n = (idx.size // TILE) * TILE
    c = idx[:n].astype(np.int64).reshape(-1, TILE)
    print(f"{tag}: {c.shape[0]} tiles of {TILE} nonzeros (rows 0 .. 2^20 of the stand-in)")
    for W in (4096, 16384):
        g = W // 4                                              # window start granularity
        b = c // g
        # per tile: counts per granule via sorting; best run of 4 consecutive granules
        bs = np.sort(b, axis=1)
        best = np.zeros(c.shape[0], np.int64)
        for t0 in range(0, c.shape[0], 2048):                   # chunks to bound memory
            blk = bs[t0:t0 + 2048]
            lo = blk                                            # window starting at granule lo covers [lo, lo + 4)
            # number of elements in [lo_i, lo_i + 4) for every candidate start lo_i (each element's granule as a start)
            hi_idx = np.empty_like(blk)
            for r in range(blk.shape[0]):
                hi_idx[r] = np.searchsorted(blk[r], blk[r] + 4, side="left")
            cnt = hi_idx - np.arange(TILE)[None, :]
            best[t0:t0 + 2048] = cnt.max(axis=1)
        share = best / TILE
        lines_window = W * 4 // 128
        req = lines_window + (TILE - best)                      # window load + one request per gather outside it
        print(f"  W = {W:6d} columns ({W * 4 >> 10} KB of LDS): nonzeros inside the best window: mean {share.mean():.3f}, median {np.median(share):.3f}, "
              f"tiles with >= 50 %: {(share >= 0.5).mean():.3f}; L2 requests per tile {req.mean():.0f} against {TILE} "
              f"(x {TILE / req.mean():.2f} fewer if EVERY tile used the window, x {TILE / np.minimum(req, TILE).mean():.2f} with a per-tile choice)")
