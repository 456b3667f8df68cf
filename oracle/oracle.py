"""ctypes front-end for the CPU oracle (oracle/loops_oracle.c) and, when present, the real
reference build (oracle/_ref/libloops_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (loops_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

c_int_p = C.POINTER(C.c_int)
c_float_p = C.POINTER(C.c_float)


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "loops_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return so


def build_ref() -> str | None:
    """Compile oracle/_ref/*.so from /root/reference when the tree is present (dev container).
    On the GPU box the prebuilt files that travelled with the snapshot are used."""
    so = os.path.join(_HERE, "_ref", "libloops_ref.so")
    if os.path.isdir("/root/reference/include/loops"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref", "ref_gpu"])
    return so if os.path.exists(so) else None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.oracle_ceil_div.restype = C.c_longlong
        _LIB.oracle_ceil_div.argtypes = [C.c_longlong, C.c_longlong]
        _LIB.oracle_merge_path_num_tiles.restype = C.c_long
        _LIB.oracle_merge_path_num_tiles.argtypes = [C.c_long] * 4
        _LIB.oracle_count_errors_f32.restype = C.c_long
        _LIB.oracle_csr_to_bcsr_f32.restype = C.c_long
        _LIB.oracle_hash.restype = C.c_uint
        _LIB.oracle_hash.argtypes = [C.c_uint]
        _LIB.oracle_default_ne_f32.argtypes = [C.c_float, C.c_float]
        _LIB.oracle_set_num_threads(usable_cpus())  # OpenMP variants: never more threads than CPUs granted
    return _LIB


def usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    # one process per GPU (torchrun): the ranks of a node share the CPUs
    try:
        n = max(1, n // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))
    except ValueError:
        pass
    return n


def ref() -> C.CDLL | None:
    """The real reference host code (None when oracle/_ref was never built)."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libloops_ref.so")
        if not os.path.exists(so):
            return None
        # libloops_ref.so is hipcc-built and DT_NEEDs libamdhip64.so.7: import torch first so the
        # process keeps ONE HIP runtime (torch bundles its own copy; a second copy loaded before
        # it leaves torch with "No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _REF = C.CDLL(so)
        _REF.ref_hash.restype = C.c_uint
        _REF.ref_hash.argtypes = [C.c_uint]
        _REF.ref_ceil_div.restype = C.c_long
        _REF.ref_ceil_div.argtypes = [C.c_long, C.c_long]
        _REF.ref_default_ne_f32.argtypes = [C.c_float, C.c_float]
        _REF.ref_csr_to_bcsr_f32.restype = C.c_long
    return _REF


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------- SpMV
def spmv_f32(offsets, indices, values, x, omp: bool = False):
    offsets, indices, values, x = _i32(offsets), _i32(indices), _f32(values), _f32(x)
    rows = offsets.size - 1
    y = np.zeros(rows, np.float32)
    fn = lib().oracle_spmv_f32_omp if omp else lib().oracle_spmv_f32
    fn(C.c_long(rows), _p(offsets), _p(indices), _p(values), _p(x), _p(y))
    return y


def spmv_f64(offsets, indices, values, x):
    offsets, indices = _i32(offsets), _i32(indices)
    values = np.ascontiguousarray(values, np.float64)
    x = np.ascontiguousarray(x, np.float64)
    rows = offsets.size - 1
    y = np.zeros(rows, np.float64)
    lib().oracle_spmv_f64(C.c_long(rows), _p(offsets), _p(indices), _p(values), _p(x), _p(y))
    return y


def spmv_f64acc_f32(offsets, indices, values, x):
    offsets, indices, values, x = _i32(offsets), _i32(indices), _f32(values), _f32(x)
    rows = offsets.size - 1
    y = np.zeros(rows, np.float32)
    lib().oracle_spmv_f64acc_f32(C.c_long(rows), _p(offsets), _p(indices), _p(values), _p(x), _p(y))
    return y


def row_l1_f32(offsets, indices, values, x):
    offsets, indices, values, x = _i32(offsets), _i32(indices), _f32(values), _f32(x)
    rows = offsets.size - 1
    y = np.zeros(rows, np.float32)
    lib().oracle_row_l1_f32(C.c_long(rows), _p(offsets), _p(indices), _p(values), _p(x), _p(y))
    return y


def count_errors_f32(y, ref_y):
    y, ref_y = _f32(y), _f32(ref_y)
    return int(lib().oracle_count_errors_f32(_p(y), _p(ref_y), C.c_long(y.size)))


class RigorousReport(C.Structure):
    _fields_ = [("total_rows", C.c_long), ("naive_mismatches", C.c_long),
                ("f32_baseline_overruns", C.c_long), ("gpu_overruns", C.c_long),
                ("max_gpu_abs_error", C.c_double), ("max_gpu_rel_error", C.c_double),
                ("wilkinson_k", C.c_double)]


def rigorous_validate_f32(offsets, indices, values, x, y_gpu, k=8.0, floor=1e-3):
    offsets, indices, values, x, y_gpu = _i32(offsets), _i32(indices), _f32(values), _f32(x), _f32(y_gpu)
    rep = RigorousReport()
    lib().oracle_rigorous_validate_f32(C.c_long(offsets.size - 1), _p(offsets), _p(indices), _p(values),
                                       _p(x), _p(y_gpu), C.c_double(k), C.c_double(floor), C.byref(rep))
    return rep


# --------------------------------------------------------------------------- generator
def xgen_int(n, lo=1, hi=10, seed=42, dtype=np.float32):
    out = np.zeros(n, dtype)
    fn = lib().oracle_xgen_int_f32 if dtype == np.float32 else lib().oracle_xgen_int_f64
    fn(C.c_long(n), C.c_int(lo), C.c_int(hi), C.c_uint(seed), _p(out))
    return out


# --------------------------------------------------------------------------- schedules
def diag_search(diagonal, tile_end, b0, a_len, b_len):
    tile_end = _i32(tile_end)
    out = np.zeros(2, np.uint32)
    lib().oracle_diag_search(C.c_longlong(diagonal), _p(tile_end), C.c_longlong(b0), C.c_longlong(a_len),
                             C.c_longlong(b_len), _p(out))
    return int(out[0]), int(out[1])


def merge_path_num_tiles(rows, nnz, tpb, ipt):
    return int(lib().oracle_merge_path_num_tiles(rows, nnz, tpb, ipt))


def merge_path_coords(offsets, tpb, ipt):
    offsets = _i32(offsets)
    rows, nnz = offsets.size - 1, int(offsets[-1])
    M = merge_path_num_tiles(rows, nnz, tpb, ipt)
    coords = np.zeros((M + 1, 2), np.uint32)
    lib().oracle_merge_path_coords(_p(offsets), C.c_long(rows), C.c_long(nnz), C.c_long(tpb), C.c_long(ipt),
                                   _p(coords))
    return coords


def merge_path_assign(offsets, tpb, ipt):
    """-> thread_start[M*tpb, 2] (local coords), atom_owner[nnz], atom_row[nnz], atom_visits[nnz]."""
    offsets = _i32(offsets)
    rows, nnz = offsets.size - 1, int(offsets[-1])
    M = merge_path_num_tiles(rows, nnz, tpb, ipt)
    ts = np.zeros((M * tpb, 2), np.uint32)
    owner = np.full(max(nnz, 1), -1, np.int32)
    row = np.full(max(nnz, 1), -1, np.int32)
    visits = np.zeros(max(nnz, 1), np.int32)
    lib().oracle_merge_path_assign(_p(offsets), C.c_long(rows), C.c_long(nnz), C.c_long(tpb), C.c_long(ipt),
                                   _p(ts), _p(owner), _p(row), _p(visits))
    return ts, owner[:nnz], row[:nnz], visits[:nnz]


def merge_path_spmv_f32(offsets, indices, values, x, tpb, ipt):
    offsets, indices, values, x = _i32(offsets), _i32(indices), _f32(values), _f32(x)
    rows, nnz = offsets.size - 1, int(offsets[-1])
    y = np.zeros(rows, np.float32)
    lib().oracle_merge_path_spmv_f32(_p(offsets), _p(indices), _p(values), _p(x), C.c_long(rows), C.c_long(nnz),
                                     C.c_long(tpb), C.c_long(ipt), _p(y))
    return y


def work_oriented_assign(offsets, num_threads):
    offsets = _i32(offsets)
    rows, nnz = offsets.size - 1, int(offsets[-1])
    tm = np.zeros((num_threads, 4), np.int32)
    owner = np.full(max(nnz, 1), -1, np.int32)
    row = np.full(max(nnz, 1), -1, np.int32)
    visits = np.zeros(max(nnz, 1), np.int32)
    lib().oracle_work_oriented_assign(_p(offsets), C.c_long(rows), C.c_long(nnz), C.c_long(num_threads),
                                      _p(tm), _p(owner), _p(row), _p(visits))
    return tm, owner[:nnz], row[:nnz], visits[:nnz]


def group_mapped_assign(offsets, group):
    offsets = _i32(offsets)
    rows, nnz = offsets.size - 1, int(offsets[-1])
    owner = np.full(max(nnz, 1), -1, np.int32)
    row = np.full(max(nnz, 1), -1, np.int32)
    visits = np.zeros(max(nnz, 1), np.int32)
    lib().oracle_group_mapped_assign(_p(offsets), C.c_long(rows), C.c_long(nnz), C.c_long(group),
                                     _p(owner), _p(row), _p(visits))
    return owner[:nnz], row[:nnz], visits[:nnz]


# --------------------------------------------------------------------------- BCSR
def csr_to_bcsr_f32(R, Cc, rows, cols, offsets, indices, values, use_ref=False):
    offsets, indices, values = _i32(offsets), _i32(indices), _f32(values)
    nbr = (rows + R - 1) // R
    boff = np.zeros(nbr + 1, np.int32)
    if use_ref:
        fn = lambda bc, bv: ref().ref_csr_to_bcsr_f32(  # noqa: E731
            C.c_int(R), C.c_int(Cc), C.c_long(rows), C.c_long(cols), C.c_long(indices.size), _p(offsets),
            _p(indices), _p(values), _p(boff), _p(bc), _p(bv))
    else:
        fn = lambda bc, bv: lib().oracle_csr_to_bcsr_f32(  # noqa: E731
            C.c_int(R), C.c_int(Cc), C.c_long(rows), C.c_long(cols), _p(offsets), _p(indices), _p(values),
            _p(boff), _p(bc), _p(bv))
    nb = int(fn(None, None))
    bcols = np.zeros(max(nb, 1), np.int32)
    bvals = np.zeros(max(nb, 1) * R * Cc, np.float32)
    fn(bcols, bvals)
    return boff, bcols[:nb], bvals[: nb * R * Cc]


def bcsr_spmv_f32(R, Cc, rows, block_offsets, block_cols, block_values, x_padded):
    block_offsets, block_cols = _i32(block_offsets), _i32(block_cols)
    block_values, x_padded = _f32(block_values), _f32(x_padded)
    y = np.zeros(rows, np.float32)
    lib().oracle_bcsr_spmv_f32(C.c_int(R), C.c_int(Cc), C.c_long(rows), C.c_long(block_offsets.size - 1),
                               _p(block_offsets), _p(block_cols), _p(block_values), _p(x_padded), _p(y))
    return y


def spmm(offsets, indices, values, B):
    """C = A * B, the reference's SpMM semantics (spmm/thread_mapped.cuh:38-51); f32 or f64 by values.dtype."""
    offsets, indices = _i32(offsets), _i32(indices)
    f64 = np.asarray(values).dtype == np.float64
    dt = np.float64 if f64 else np.float32
    values = np.ascontiguousarray(values, dt)
    B = np.ascontiguousarray(B, dt)
    rows, n = offsets.size - 1, B.shape[1]
    out = np.zeros((rows, n), dt)
    fn = lib().oracle_spmm_f64 if f64 else lib().oracle_spmm_f32
    fn(C.c_long(rows), _p(offsets), _p(indices), _p(values), _p(B), C.c_long(n), _p(out))
    return out


# --------------------------------------------------------------------------- reference (real)
def ref_load_mtx(path):
    r = ref()
    rows, cols, nnz = C.c_long(), C.c_long(), C.c_long()
    po, pi, pv = c_int_p(), c_int_p(), c_float_p()
    rc = r.ref_mtx_load_csr_f32(path.encode(), C.byref(rows), C.byref(cols), C.byref(nnz), C.byref(po),
                                C.byref(pi), C.byref(pv))
    if rc != 0:
        raise RuntimeError("reference loader rejected " + path)
    off = np.ctypeslib.as_array(po, (rows.value + 1,)).copy()
    idx = np.ctypeslib.as_array(pi, (max(nnz.value, 1),)).copy()[: nnz.value]
    val = np.ctypeslib.as_array(pv, (max(nnz.value, 1),)).copy()[: nnz.value]
    for p in (po, pi, pv):
        r.ref_free(p)
    return rows.value, cols.value, off, idx, val


def ref_xgen_int(n, lo=1, hi=10, seed=42):
    out = np.zeros(n, np.float32)
    ref().ref_xgen_int_f32(C.c_long(n), C.c_int(lo), C.c_int(hi), C.c_uint(seed), _p(out))
    return out


def ref_spmv_f32(offsets, indices, values, x, cols=None, kind="f32"):
    offsets, indices, values, x = _i32(offsets), _i32(indices), _f32(values), _f32(x)
    rows = offsets.size - 1
    cols = x.size if cols is None else cols
    y = np.zeros(rows, np.float32)
    fn = {"f32": ref().ref_spmv_f32, "f64acc": ref().ref_spmv_f64acc_f32, "l1": ref().ref_row_l1_f32}[kind]
    fn(C.c_long(rows), C.c_long(cols), C.c_long(indices.size), _p(offsets), _p(indices), _p(values), _p(x), _p(y))
    return y
