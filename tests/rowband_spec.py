"""Numpy specification of the row-band layout (include/loops/kernels/rowband.hxx): what the device builder must produce,
array for array.  Test infrastructure (also used by tests/perf/)."""
import heapq

import numpy as np

STEP = 256
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
    """-> (val [steps * 256], row16, delta8, perm, stepbase [steps, 4], band_step [bands + 1], hubs [bands, 33]).
    Items sorted by (band = row // H, column, CSR position).  Inside a band a gap of more than 255 columns between neighbours is
    bridged by (gap - 1) // 255 padding slots of delta 255 in front of the item; every band is padded to whole steps of 256 slots.
    Inside a step slot q sits at lane q % 64, element q // 64.  Per slot: the value, the row code (row - band * H, or
    H + 1 + hub * 16 + q % 16 for the items of a hub row; H = padding), the column delta to the previous slot (0 .. 255; 0 for trailing
    padding and for the first slot of every group of 64) and perm (CSR position, -1 = padding); per step and group of 64 slots the absolute column of the group's first slot."""
    nnz = idx.size
    B = -(-rows // H) if rows else 0
    hubidx, hubs = hub_table(off, rows, H)
    row_of = np.repeat(np.arange(rows, dtype=np.int64), np.diff(off.astype(np.int64)))
    band = row_of // H
    col = idx.astype(np.int64)
    order = np.lexsort((np.arange(nnz), col, band))
    sb, sc, sr = band[order], col[order], row_of[order]
    first = np.r_[True, sb[1:] != sb[:-1]] if nnz else np.zeros(0, bool)
    gap = np.where(first, 0, sc - np.r_[0, sc[:-1]])
    pads = np.where(gap > 255, (gap - 1) // 255, 0)
    cnt = 1 + pads
    pos = np.r_[0, np.cumsum(cnt)]                                    # running slot count, band after band (not yet step-aligned)
    band_start = np.searchsorted(sb, np.arange(B + 1), side="left")
    tot = pos[band_start[1:]] - pos[band_start[:-1]]
    band_step = np.r_[0, np.cumsum(-(-tot // STEP))].astype(np.int64)
    steps = int(band_step[-1])
    n = steps * STEP
    v = np.zeros(n, val.dtype)
    row16 = np.full(n, H, np.uint16)
    delta8 = np.zeros(n, np.uint8)
    perm = np.full(n, -1, np.int32)
    colabs = np.zeros(n, np.int64)                                    # absolute column of every slot (for the step bases)
    slot_in_band = pos[:-1] - pos[band_start[sb]] + pads              # of the item itself; its pads sit right in front
    s_abs = band_step[sb] * STEP + slot_in_band                       # slot number counted in sorted order
    def mem(s):                                                       # sorted slot number -> position in the arrays
        q = s % STEP
        return (s // STEP) * STEP + (q % 64) * 4 + q // 64
    at = mem(s_abs)
    v[at] = val[order]
    h = hubidx[sr].astype(np.int64)
    q = s_abs % STEP
    code = np.where(h >= 0, H + 1 + h * HUB_REPLICAS + q % HUB_REPLICAS, sr - sb * H)
    row16[at] = code.astype(np.uint16)
    delta8[at] = (gap - 255 * pads).astype(np.uint8)
    perm[at] = order.astype(np.int32)
    colabs[at] = sc
    # the padding slots that bridge gaps: pad t (0-based) of item j sits pads_j - t slots in front of it, at column prev + 255 (t + 1)
    jj = np.flatnonzero(pads > 0)
    if jj.size:
        rep = np.repeat(jj, pads[jj])
        t = np.arange(rep.size) - np.repeat(np.cumsum(pads[jj]) - pads[jj], pads[jj])
        ps = s_abs[rep] - pads[rep] + t
        pat = mem(ps)
        delta8[pat] = 255
        colabs[pat] = (sc[rep] - gap[rep]) + 255 * (t + 1)
    # step bases: the absolute column of the first slot of every group of 64 sorted slots (an item's own column, a bridging
    # pad's column); 0 for groups that are trailing padding altogether (their deltas are 0: they gather x[0] into the dump word)
    used = np.zeros(n, bool)
    used[at] = True
    if jj.size:
        used[pat] = True
    first_of_group = mem(np.arange(0, n, 64, dtype=np.int64))
    delta8[first_of_group] = 0                                        # (the group's base IS the column of its first slot)
    stepbase = np.where(used[first_of_group], colabs[first_of_group], 0).reshape(steps, 4).astype(np.int32) if steps else np.zeros((0, 4), np.int32)
    return v, row16, delta8, perm, stepbase, band_step.astype(np.int32), hubs


def chunk_list(band_step, target_chunks):
    """-> (chunks [n, 4] = {band, first step, end step, partial slot or -1}, multi [m, 3] = {band, first slot, chunks}).
    Every band one chunk, then cut by cut: the next cut goes to the band whose chunks are the longest (ties: the lower band),
    until the list has target_chunks entries."""
    B = len(band_step) - 1
    n = np.diff(np.asarray(band_step, np.int64))
    pieces = np.ones(B, np.int64)
    s = B
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


def product(v, row16, delta8, stepbase, band_step, hubs, H, rows, x):
    """What the kernels compute, in numpy (fp32 products, fp64 sums): the specification of y for a layout."""
    y = np.zeros(rows)
    if v.size == 0:
        return y
    steps = v.size // STEP
    # memory position (step, lane, e) holds sorted slot 64 e + lane: columns = base of the group + running sum of the deltas
    d = delta8.reshape(steps, 64, 4).transpose(0, 2, 1).astype(np.int64)          # [step, e, lane]
    c = (stepbase.astype(np.int64)[:, :, None] + np.cumsum(d, axis=2)).transpose(0, 2, 1).reshape(-1)   # back to memory order
    step = np.arange(v.size) // STEP
    band = np.searchsorted(band_step, step, side="right") - 1
    code = row16.astype(np.int64)
    prod = (v * x[c]).astype(np.float64)
    plain = code < H
    np.add.at(y, (band * H + code)[plain], prod[plain])
    hub = code > H
    hrow = hubs[band[hub], 1 + (code[hub] - H - 1) // HUB_REPLICAS].astype(np.int64)
    np.add.at(y, band[hub] * H + hrow, prod[hub])
    return y
