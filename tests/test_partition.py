"""Row-range sharding + allgatherv(y): host logic on CPU, N > 1 path on gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest

from conftest import battery


def test_row_ranges_balance_and_cover():
    from loops_amd import generate as G, partition as P
    deg = G.powerlaw_degrees(1 << 14, 1 << 18, cap=1 << 12)
    off = np.zeros(deg.size + 1, np.int64)
    np.cumsum(deg, out=off[1:])
    for parts in (1, 2, 3, 8):
        b = P.row_ranges(off, parts)
        assert b[0] == 0 and b[-1] == deg.size and (np.diff(b) >= 0).all()
        work = np.array([(b[i + 1] - b[i]) + (off[b[i + 1]] - off[b[i]]) for i in range(parts)])
        assert work.sum() == deg.size + off[-1]
        assert work.max() <= work.sum() / parts + (1 << 12) + 1  # within one max-degree row of even
    # the split is the kernels' merge-path split
    from oracle import oracle as O
    for d in (0, 1, 77, 5000, int(off[-1]) + deg.size):
        assert P.merge_path_split(off, d) == O.diag_search(d, off[1:].astype(np.int32), 0, deg.size, int(off[-1]))


def test_slices_reassemble_spmv():
    from loops_amd import partition as P
    from oracle import oracle as O
    for name, (r, c, off, idx, val) in battery().items():
        if r < 2:
            continue
        x = O.xgen_int(c)
        want = O.spmv_f32(off, idx, val, x)
        b = P.row_ranges(off.astype(np.int64), 3)
        got = np.concatenate([O.spmv_f32(*P.slice_csr(off, idx, val, int(b[p]), int(b[p + 1])), x) for p in range(3)])
        assert np.array_equal(got, want), name


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from loops_amd import partition as P
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bounds = {2: np.array([0, 5, 12], np.int64), 3: np.array([0, 4, 4, 12], np.int64),
                  8: np.array([0, 3, 3, 10, 11, 20, 25, 31, 40], np.int64)}[world]   # uneven, one empty shard
        total = int(bounds[-1])
        shard = P.Shard(rank, world, int(bounds[rank]), int(bounds[rank + 1]), bounds)
        want = torch.cat([torch.arange(int(bounds[r]), int(bounds[r + 1]), dtype=torch.float32) * (r + 1) for r in range(world)])
        ok = True
        for mode in ("p2p", "padded"):
            y = torch.full((total,), -1.0)
            y[shard.row_begin:shard.row_end] = torch.arange(shard.row_begin, shard.row_end, dtype=torch.float32) * (rank + 1)
            P.allgatherv_(y, shard, mode=mode)
            ok = ok and bool(torch.equal(y, want))
            # the prebuilt-exchange object, run twice (op lists / scratch are reused between steps)
            y2 = torch.full((total,), -1.0)
            ex = P.Allgatherv(y2, shard, mode)
            for rep in (1.0, 3.0):
                y2.fill_(-1.0)
                y2[shard.row_begin:shard.row_end] = torch.arange(shard.row_begin, shard.row_end, dtype=torch.float32) * (rank + 1) * rep
                ex.run()
                ok = ok and bool(torch.equal(y2, want * rep))
        # chunked exchange: every rank's slice in 2 pieces, posted one after the other, waited at the end
        cb = [np.array([bounds[r], (bounds[r] + bounds[r + 1]) // 2, bounds[r + 1]], np.int64) for r in range(world)]
        y3 = torch.full((total,), -1.0)
        y3[shard.row_begin:shard.row_end] = torch.arange(shard.row_begin, shard.row_end, dtype=torch.float32) * (rank + 1)
        cx = P.ChunkedAllgatherv(y3, shard, cb)
        for rep in range(2):
            for c in range(cx.chunks):
                cx.post(c)
            cx.finish()
            ok = ok and bool(torch.equal(y3, want))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_allgatherv_gloo(world):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res) and len(res) == world


def test_native_row_ranges_match_the_numpy_specification():
    """loops_row_ranges (include/loops/multi_gpu/partition.hxx through the C ABI: the kernels' own search on N - 1 diagonals,
    a host function -- no GPU needed) against the numpy / Python-integer specification: battery matrices, power-law degrees,
    empty rows at both ends, more parts than rows, one part; the ranges tile the rows and are balanced by rows + nnz."""
    import ctypes as C
    from loops_amd import generate as G, partition as P, _lib as L
    from conftest import battery
    cases = [np.asarray(off, np.int64) for (_, _, off, _, _) in battery().values()]
    for rows, nnz, cap in ((1 << 13, 1 << 17, 1 << 11), (100_003, 1 << 20, 1 << 14), (5, 40, 10)):
        deg = G.powerlaw_degrees(rows, nnz, cap=cap)
        cases.append(np.concatenate([[0], np.cumsum(deg)]))
    z = np.zeros(1000, np.int64)
    z[400:600] = 7
    cases.append(np.concatenate([[0], np.cumsum(z)]))                    # empty rows at both ends
    for off in cases:
        rows = off.size - 1
        for parts in (1, 2, 3, 8, 17):
            want = P.row_ranges_numpy(off, parts)
            got = P.row_ranges(off, parts)
            assert np.array_equal(got, want), (rows, parts)
            assert got[0] == 0 and got[-1] == rows and np.all(np.diff(got) >= 0)
            if rows and parts > 1:                                       # balanced: no range exceeds its share by more than one row's items
                items = np.diff(got) + np.diff(off[got])
                longest = int(np.diff(off).max(initial=0)) + 1
                assert items.max() <= -(-(rows + int(off[-1])) // parts) + longest
    # argument errors come back as codes, not crashes
    off32 = np.array([0, 1, 2], np.int32)
    b = np.zeros(3, np.int64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert L.lib().loops_row_ranges(2, p(off32), 0, p(b)) == -1
    assert L.lib().loops_row_ranges(-1, p(off32), 2, p(b)) == -1
    assert L.lib().loops_row_ranges(2, None, 2, p(b)) == -1
    big = np.array([0, (1 << 31) - 2], np.int32)
    assert L.lib().loops_row_ranges(1, p(big), 2, p(b)) == 0 and L.lib().loops_row_ranges(2, p(np.array([0, 1 << 30, (1 << 31) - 1], np.int32)), 2, p(b)) == -2


@pytest.mark.gpu
def test_native_communicator_world_size_one():
    """loops_comm_* / loops_allgatherv_* on the one GPU a test box has: the RCCL entry points resolve inside this process (the
    instance PyTorch loaded), a communicator of one rank initialises, the allgatherv of a one-rank partition leaves y alone and
    returns success on torch's stream.  Two ranks cannot share a device under RCCL: the N > 1 flow is covered with gloo
    (test_allgatherv_gloo) and by bench.py's probe, which lists "native-p2p" only where it worked on every rank."""
    import torch
    from loops_amd import partition as P
    torch.cuda.set_device(0)
    comm = P.NativeComm(0, 1)
    y = torch.arange(1000, dtype=torch.float32, device="cuda")
    want = y.clone()
    comm.allgatherv(y, np.array([0, 1000], np.int64))
    torch.cuda.synchronize()
    assert torch.equal(y, want)
    yd = torch.arange(10, dtype=torch.float64, device="cuda")
    comm.allgatherv(yd, np.array([0, 10], np.int64))
    comm.close()
    comm.close()   # idempotent
