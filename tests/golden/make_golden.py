"""Regenerates the committed golden fixtures from the REAL reference (dev container only).

    python tests/golden/make_golden.py

Needs oracle/_ref/libloops_ref.so (built by `make -C oracle ref` from /root/reference, in
place).  Everything written here is data: inputs and the outputs the reference's own host
code (matrix_market_t::load, csr_t(coo), generate::random::uniform_distribution,
reference::spmv / spmv_f64 / row_l1_products, bcsr_t(csr), layout views) produced for them.
The merge-path / work_oriented / group_mapped tables are device-only in the reference; their
goldens come from the reference's device code run on an MI355X through
oracle/_ref/libloops_ref_gpu.so at test time (tests/test_ref_gpu_pin.py, on the GPU box) -- they are compared live,
not stored.  The reference's own SpMV battery (mt19937 factories) is a separate fixture: tests/golden/make_ref_battery.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
from conftest import battery  # noqa: E402


def main():
    assert O.ref() is not None, "build oracle/_ref first (make -C oracle ref)"
    # ---- C1: chesapeake through the reference loader + generator + CPU SpMV ----------------
    rows, cols, off, idx, val = O.ref_load_mtx(os.path.join(HERE, "chesapeake.mtx"))
    x = O.ref_xgen_int(cols, 1, 10, 42)
    np.savez(os.path.join(HERE, "chesapeake.npz"), rows=rows, cols=cols, offsets=off, indices=idx, values=val, x=x,
             y=O.ref_spmv_f32(off, idx, val, x), y_f64acc=O.ref_spmv_f32(off, idx, val, x, kind="f64acc"))
    # ---- x generator known answers ------------------------------------------------------------
    np.savez(os.path.join(HERE, "xgen.npz"), x_1_10_42=O.ref_xgen_int(4096, 1, 10, 42),
             x_0_1_7=O.ref_xgen_int(4096, 0, 1, 7), x_m5_5_12345=O.ref_xgen_int(4096, -5, 5, 12345),
             hash=np.array([O.ref().ref_hash(a) for a in (0, 1, 2, 41, 12345, 2**32 - 1)], np.uint32))
    # ---- battery: reference::spmv, spmv_f64, row_l1_products ------------------------------------
    out = {}
    for name, (r, c, off, idx, val) in battery().items():
        xi = O.ref_xgen_int(c, 1, 10, 42)
        xr = (0.5 + np.random.default_rng(23).random(c)).astype(np.float32)
        out[name + ".shape"] = np.array([r, c], np.int64)
        out[name + ".offsets"], out[name + ".indices"], out[name + ".values"] = off, idx, val
        out[name + ".x_int"], out[name + ".x_real"] = xi, xr
        for tag, xv in (("int", xi), ("real", xr)):
            out[f"{name}.y_{tag}"] = O.ref_spmv_f32(off, idx, val, xv, cols=c)
            out[f"{name}.y64_{tag}"] = O.ref_spmv_f32(off, idx, val, xv, cols=c, kind="f64acc")
            out[f"{name}.l1_{tag}"] = O.ref_spmv_f32(off, idx, val, xv, cols=c, kind="l1")
        for R in (2, 3, 4):
            if r and idx.size:
                bo, bc, bv = O.csr_to_bcsr_f32(R, R, r, c, off, idx, val, use_ref=True)
                out[f"{name}.bcsr{R}.offsets"], out[f"{name}.bcsr{R}.cols"], out[f"{name}.bcsr{R}.values"] = bo, bc, bv
    np.savez_compressed(os.path.join(HERE, "battery.npz"), **out)
    # ---- layout views: every accessor of every view, evaluated by the reference's PODs ----------
    lay = {}
    offs = {"csr4": np.array([0, 2, 2, 5, 7], np.int32), "bcsr5": np.array([0, 3, 5, 5, 9, 12], np.int32),
            "one_row": np.array([0, 6], np.int32), "all_empty": np.array([0, 0, 0, 0], np.int32)}
    import ctypes as C
    for name, o in offs.items():
        nt, na = o.size - 1, int(o[-1])
        p = o.ctypes.data_as(C.c_void_p)
        lay[name + ".offsets"] = o
        lay[name + ".csr"] = np.array([[O.ref().ref_layout_csr(p, nt, na, w, a) for a in range(nt)] for w in range(4)]
                                      + [[O.ref().ref_layout_csr(p, nt, na, 4, a) if a < na else -1 for a in range(nt)]],
                                      np.int64)
        lay[name + ".tile_of"] = np.array([O.ref().ref_layout_csr(p, nt, na, 4, a) for a in range(na)], np.int64)
        for K in (2, 4, 8, 16):
            T = O.ref().ref_layout_flat(K, p, nt, na, 5, 0)
            lay[f"{name}.flat{K}"] = np.array([[O.ref().ref_layout_flat(K, p, nt, na, w, t) for t in range(T)]
                                               for w in range(4)], np.int64).reshape(4, T)
            lay[f"{name}.flat{K}.tile_of"] = np.array([O.ref().ref_layout_flat(K, p, nt, na, 4, a) for a in range(na)], np.int64)
            lay[f"{name}.flat{K}.base_tile_of"] = np.array([O.ref().ref_layout_flat(K, p, nt, na, 7, a) for a in range(na)], np.int64)
    for nt, pitch in ((5, 3), (1, 7), (4, 1)):
        lay[f"ell{nt}x{pitch}"] = np.array([[O.ref().ref_layout_ell(nt, pitch, w, t) for t in range(nt)] for w in range(4)], np.int64)
        lay[f"ell{nt}x{pitch}.tile_of"] = np.array([O.ref().ref_layout_ell(nt, pitch, 4, a) for a in range(nt * pitch)], np.int64)
    lay["coo9"] = np.array([[O.ref().ref_layout_coo(9, w, t) for t in range(9)] for w in range(5)], np.int64)
    np.savez(os.path.join(HERE, "layouts.npz"), **lay)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
