"""Numpy specification of the row-band layout (include/loops/kernels/rowband.hxx): what the device builder must produce,
array for array.  Test infrastructure (also used by tests/perf/)."""
import heapq

import numpy as np

STEP = 256
COLBLOCK_BITS = 16
MAX_HUBS = 32
HUB_REPLICAS = 16


def hub_table(off, rows, H):
    """-> (hubidx [rows] = hub number of the row inside its band or -1, hubs [bands, 33] = count, then rows inside the band).
    A row is a hub when it holds >= max(64, band items // 128) nonzeros; the first 32 per band in row order."""
    B = -(-rows // H) if rows else 0
    deg = np.diff(off.astype(np.int64))
    hubidx = np.full(rows, -1, np.int16)
    hubs = np.zeros((B, MAX_HUBS + 1), np.uint16)
    for b in range(B):
        r0, r1 = b * H, min((b + 1) * H, rows)
        items = int(off[r1]) - int(off[r0])
        thr = max(64, items // 128)
        cand = np.flatnonzero(deg[r0:r1] >= thr)[:MAX_HUBS]
        hubidx[r0 + cand] = np.arange(cand.size, dtype=np.int16)
        hubs[b, 0] = cand.size
        hubs[b, 1:1 + cand.size] = cand
    return hubidx, hubs


def layout(off, idx, val, rows, cols, H):
    """-> (val [steps * 256], rc, perm, stepcol [steps], band_step [bands + 1], hubs [bands, 33]).
    Items sorted by (band = row // H, column, CSR position); segments (band, column >> 16) padded to whole steps of 256; inside a
    step sorted position q sits at lane q % 64, element q // 64; rc = (row code) << 16 | (column & 0xFFFF), row code = row - band * H,
    or H + 1 + hub * 16 + q % 16 for the items of a hub row; padding: value 0, row code H, offset 0, perm -1."""
    nnz = idx.size
    B = -(-rows // H) if rows else 0
    CB = max(1, -(-cols // (1 << COLBLOCK_BITS)))
    hubidx, hubs = hub_table(off, rows, H)
    row_of = np.repeat(np.arange(rows, dtype=np.int64), np.diff(off.astype(np.int64)))
    band = row_of // H
    col = idx.astype(np.int64)
    order = np.lexsort((np.arange(nnz), col, band))
    g = band[order] * CB + (col[order] >> COLBLOCK_BITS)
    counts = np.bincount(g, minlength=B * CB).astype(np.int64)
    seg_steps = -(-counts // STEP)
    seg_step = np.r_[0, np.cumsum(seg_steps)]
    seg_start = np.r_[0, np.cumsum(counts)]
    steps = int(seg_step[-1])
    within = np.arange(nnz, dtype=np.int64) - seg_start[g]
    step = seg_step[g] + within // STEP
    q = within % STEP
    at = step * STEP + (q % 64) * 4 + q // 64
    v = np.zeros(steps * STEP, val.dtype)
    rc = np.full(steps * STEP, H << 16, np.uint32)
    perm = np.full(steps * STEP, -1, np.int32)
    stepcol = np.zeros(steps, np.int32)
    v[at] = val[order]
    h = hubidx[row_of[order]].astype(np.int64)
    code = np.where(h >= 0, H + 1 + h * HUB_REPLICAS + q % HUB_REPLICAS, row_of[order] - band[order] * H)
    rc[at] = ((code << 16) | (col[order] & 0xFFFF)).astype(np.uint32)
    perm[at] = order.astype(np.int32)
    stepcol[step] = ((col[order] >> COLBLOCK_BITS) << COLBLOCK_BITS).astype(np.int32)
    band_step = seg_step[np.arange(B + 1) * CB].astype(np.int32)
    return v, rc, perm, stepcol, band_step, hubs


def chunk_list(band_step, target_chunks):
    """-> (chunks [n, 4] = {band, first step, end step, partial slot or -1}, multi [m, 3] = {band, first slot, chunks}).
    Bands are cut in proportion to their steps; surplus cuts go to the band whose chunks are the longest (ties: the lower band)."""
    B = len(band_step) - 1
    n = np.diff(np.asarray(band_step, np.int64))
    total = int(n.sum())
    pieces = np.ones(B, np.int64)
    if total > 0:
        pieces = np.where(n > 0, np.clip(n * target_chunks // total, 1, np.maximum(n, 1)), 1)
    s = int(pieces.sum())
    if s < target_chunks:
        heap = [(-float(n[b]) / pieces[b], b) for b in range(B) if n[b] > pieces[b]]
        heapq.heapify(heap)
        while s < target_chunks and heap:
            _, b = heapq.heappop(heap)
            pieces[b] += 1
            s += 1
            if n[b] > pieces[b]:
                heapq.heappush(heap, (-float(n[b]) / pieces[b], b))
    chunks, multi, partials = [], [], 0
    for b in range(B):
        s0, nb = int(band_step[b]), int(n[b])
        if nb <= 0:
            chunks.append((b, s0, s0, -1))
            continue
        size = -(-nb // int(pieces[b]))
        count = -(-nb // size)
        if count > 1:
            multi.append((b, partials, count))
        for k in range(count):
            begin = s0 + k * size
            chunks.append((b, begin, min(begin + size, s0 + nb), partials + k if count > 1 else -1))
        if count > 1:
            partials += count
    # the work list's order: by piece number first (uncut bands are piece 0), bands ascending inside
    piece, last = [], None
    for c in chunks:
        piece.append(piece[-1] + 1 if c[0] == last else 0)
        last = c[0]
    order = np.argsort(np.asarray(piece, np.int64), kind="stable") if chunks else []
    chunks = [chunks[i] for i in order]
    return np.array(chunks, np.int32).reshape(-1, 4), np.array(multi, np.int32).reshape(-1, 3)


def product(v, rc, stepcol, band_step, hubs, H, rows, x):
    """What the kernels compute, in numpy (fp32 products, fp64 sums): the specification of y for a layout."""
    y = np.zeros(rows)
    if v.size == 0:
        return y
    step = np.arange(v.size) // STEP
    band = np.searchsorted(band_step, step, side="right") - 1
    code = (rc >> 16).astype(np.int64)
    c = stepcol[step].astype(np.int64) + (rc & 0xFFFF)
    prod = (v * x[c]).astype(np.float64)
    plain = code < H
    np.add.at(y, (band * H + code)[plain], prod[plain])
    hub = code > H
    hrow = hubs[band[hub], 1 + (code[hub] - H - 1) // HUB_REPLICAS].astype(np.int64)
    np.add.at(y, band[hub] * H + hrow, prod[hub])
    return y
