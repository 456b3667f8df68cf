"""ctypes binding of libloops_probes.so -- MEASUREMENT code only (loops_amd/csrc/loops_probes.h): calibration
kernels (streaming copy, random gather, address rate, row gather) and experimental instantiations of the product
kernels (cache-policy bits).  Used by bench.py, scripts/ and tests/perf/; the product (loops_amd.spmv,
libloops_amd.so) never imports or loads it."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L

_probes = None


def lib() -> C.CDLL:
    global _probes
    if _probes is None:
        if not os.path.exists(L.PROBES_LIB_PATH):
            raise L.LoopsError(f"{L.PROBES_LIB_PATH} not found: build it with loops_amd._lib.build_probes() "
                               "(__graft_entry__.build() does)")
        P = L.load_shared(L.PROBES_LIB_PATH)
        vp, ci = C.c_void_p, C.c_int
        P.loops_stream_copy_f32.argtypes = [vp, vp, C.c_size_t, vp]
        P.loops_gather_f32.argtypes = [vp, vp, vp, C.c_size_t, ci, vp]
        P.loops_stream_copy_tuned_f32.argtypes = [vp, vp, C.c_size_t, ci, ci, ci, vp]
        P.loops_stream_read_prefetch_f32.argtypes = [vp, vp, C.c_size_t, ci, ci, ci, vp]
        P.loops_mixed_gather_f32.argtypes = [vp, vp, vp, C.c_size_t, ci, vp]
        P.loops_address_rate_f32.argtypes = [vp, ci, ci, ci, ci, vp, vp]
        P.loops_row_gather_f32.argtypes = [vp, vp, C.c_size_t, ci, ci, vp, vp]
        P.loops_lds_update_rate_f32.argtypes = [ci, ci, ci, ci, vp, vp]
        P.loops_probe_merge_path_scratch_bytes.argtypes = [ci, ci]
        P.loops_probe_merge_path_scratch_bytes.restype = C.c_size_t
        P.loops_probe_policy_name.argtypes = [ci]
        P.loops_probe_policy_name.restype = C.c_char_p
        P.loops_probe_merge_path_f32.argtypes = [ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
        P.loops_probe_merge_path_shape_f32.argtypes = [ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
        P.loops_probe_persistent_f32.argtypes = [ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
        _probes = P
    return _probes


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def stream_copy(src, dst):
    L.check(lib().loops_stream_copy_f32(_ptr(src), _ptr(dst), src.numel(), _stream()), "loops_stream_copy_f32")


def stream_copy_tuned(src, dst, unroll: int, flags: int, blocks: int):
    L.check(lib().loops_stream_copy_tuned_f32(_ptr(src), _ptr(dst), src.numel(), unroll, flags, blocks, _stream()), "loops_stream_copy_tuned_f32")


def stream_read_prefetch(src, sink, distance: int, line_words: int = 32, waves_per_cu: int = 32):
    L.check(lib().loops_stream_read_prefetch_f32(_ptr(src), _ptr(sink), src.numel(), distance, line_words, waves_per_cu, _stream()),
            "loops_stream_read_prefetch_f32")


def mixed_gather(table, idx, out, scalar_per_64: int):
    L.check(lib().loops_mixed_gather_f32(_ptr(table), _ptr(idx), _ptr(out), idx.numel(), scalar_per_64, _stream()), "loops_mixed_gather_f32")


def gather(table, idx, out, mode: int = 0):
    L.check(lib().loops_gather_f32(_ptr(table), _ptr(idx), _ptr(out), idx.numel(), mode, _stream()), "loops_gather_f32")


def address_rate(table, reps: int, pattern: int, blocks: int, out):
    L.check(lib().loops_address_rate_f32(_ptr(table), table.numel(), reps, pattern, blocks, _ptr(out), _stream()),
            "loops_address_rate_f32")


def lds_update_rate(mode: int, pattern: int, reps: int, blocks: int, out):
    L.check(lib().loops_lds_update_rate_f32(mode, pattern, reps, blocks, _ptr(out), _stream()), "loops_lds_update_rate_f32")


def row_gather(table, idx, row_floats: int, blocks: int, out):
    L.check(lib().loops_row_gather_f32(_ptr(table), _ptr(idx), idx.numel(), row_floats, blocks, _ptr(out), _stream()),
            "loops_row_gather_f32")


def policies():
    """Names of the compiled cache-policy variants of the fused merge_path_flat kernel, by policy id."""
    return [lib().loops_probe_policy_name(i).decode() for i in range(lib().loops_probe_policy_count())]


class PolicyRunner:
    """merge_path_spmv_fused<512, 8> of a CSR with a chosen cache policy (loops_probe_merge_path_f32)."""

    def __init__(self, csr):
        self.csr = csr
        n = lib().loops_probe_merge_path_scratch_bytes(csr.rows, csr.nnzs)
        self.scratch = torch.empty(n, dtype=torch.uint8, device=csr.values.device)
        self.built = False

    def run(self, policy: int, x, y, stages: int = 3):
        c = self.csr
        if not self.built:
            stages |= 4
            self.built = True
        L.check(lib().loops_probe_merge_path_f32(policy, stages, c.rows, c.cols, c.nnzs, _ptr(c.offsets), _ptr(c.indices),
                                                 _ptr(c.values), _ptr(x), _ptr(y), _ptr(self.scratch), _stream()),
                "loops_probe_merge_path_f32")
        return y


class PersistentRunner:
    """The merge tiles walked by `groups` persistent workgroups (work_oriented_spmv_fused<512, 8>), plain or with each tile's
    gathers issued before the stream loads of the next tile (loops_probe_persistent_f32)."""

    def __init__(self, csr):
        self.csr = csr
        n = lib().loops_probe_merge_path_scratch_bytes(csr.rows, csr.nnzs)
        self.scratch = torch.empty(n, dtype=torch.uint8, device=csr.values.device)
        self.built = False

    def run(self, pipelined: int, groups: int, x, y, stages: int = 3):
        c = self.csr
        if not self.built:
            stages |= 4
            self.built = True
        L.check(lib().loops_probe_persistent_f32(int(pipelined), groups, stages, c.rows, c.cols, c.nnzs, _ptr(c.offsets),
                                                 _ptr(c.indices), _ptr(c.values), _ptr(x), _ptr(y), _ptr(self.scratch), _stream()),
                "loops_probe_persistent_f32")
        return y


SHAPES = ["512x8", "512x16", "1024x8", "1024x4", "256x8"]


class ShapeRunner:
    """merge_path_spmv_fused with tile shapes the product does not ship (loops_probe_merge_path_shape_f32)."""

    def __init__(self, csr):
        self.csr = csr
        n = 4 * lib().loops_probe_merge_path_scratch_bytes(csr.rows, csr.nnzs)
        self.scratch = {i: torch.empty(n, dtype=torch.uint8, device=csr.values.device) for i in range(len(SHAPES))}
        self.built = set()

    def run(self, shape: int, x, y, stages: int = 3):
        c = self.csr
        if shape not in self.built:
            stages |= 4
            self.built.add(shape)
        L.check(lib().loops_probe_merge_path_shape_f32(shape, stages, c.rows, c.cols, c.nnzs, _ptr(c.offsets), _ptr(c.indices),
                                                       _ptr(c.values), _ptr(x), _ptr(y), _ptr(self.scratch[shape]), _stream()),
                "loops_probe_merge_path_shape_f32")
        return y
