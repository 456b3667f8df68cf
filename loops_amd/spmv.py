"""Host-side mirror of the reference's SpMV operator interface, over the C ABI.

Names follow ``loops::algorithms::spmv::*`` (reference include/loops/algorithms/spmv/*.cuh):
``merge_path_flat(csr, x, y)``, ``work_oriented``, ``thread_mapped``, ``group_mapped``,
``original``, ``flat_partitioned``, ``bcsr_thread_mapped``; ``MergePathPlan`` mirrors
``schedule::merge_path::preprocess_t``.  torch is used for device memory and streams only;
every array crosses the boundary as a raw device pointer + size.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(t.data_ptr() if t is not None else 0)


@dataclass
class CSR:
    """csr_t<int, int, T> on the device (reference include/loops/container/csr.hxx:36-95)."""
    rows: int
    cols: int
    offsets: torch.Tensor   # int32 [rows + 1]
    indices: torch.Tensor   # int32 [nnz]
    values: torch.Tensor    # float32 | float64 [nnz]

    @property
    def nnzs(self) -> int:
        return int(self.indices.numel())

    @staticmethod
    def from_numpy(rows, cols, offsets, indices, values, device="cuda"):
        return CSR(int(rows), int(cols),
                   torch.from_numpy(np.ascontiguousarray(offsets, np.int32)).to(device),
                   torch.from_numpy(np.ascontiguousarray(indices, np.int32)).to(device),
                   torch.from_numpy(np.ascontiguousarray(values)).to(device))

    def check(self, x, y):
        assert self.offsets.dtype == torch.int32 and self.indices.dtype == torch.int32
        assert self.offsets.is_cuda and self.offsets.is_contiguous() and self.offsets.numel() == self.rows + 1
        assert self.indices.is_contiguous() and self.values.is_contiguous()
        assert x.dtype == self.values.dtype == y.dtype and x.is_contiguous() and y.is_contiguous()
        assert x.numel() >= self.cols and y.numel() >= self.rows


class MergePathPlan:
    """Per-workgroup merge-path start coordinates for (csr.offsets, tile shape); reusable across
    SpMVs on the same sparsity structure (mirrors schedule::merge_path::preprocess_t)."""

    def __init__(self, csr: CSR, tile: str = "256x8"):
        """``tile``: a compiled shape ("256x8", "512x8", ...) or "auto" = LOOPS_TILE_AUTO (256x8 when the plan is
        self-completing with it, 512x8 otherwise; ``self.tile`` then names the shape that was picked)."""
        self.rows, self.nnz = csr.rows, csr.nnzs
        self._h = C.c_void_p()
        cfg = L.TILE_AUTO if tile == "auto" else L.TILES[tile][0]
        L.check(L.lib().loops_merge_plan_create(csr.rows, csr.nnzs, _ptr(csr.offsets), cfg, _stream(),
                                                C.byref(self._h)), "loops_merge_plan_create")
        if tile == "auto":  # which shape was picked: the two candidates differ in their tile count
            total = csr.rows + csr.nnzs
            tile = "256x8" if self.num_tiles == (total + 2047) // 2048 else "512x8"
        self.tile = tile
        self.cfg, self.tpb, self.ipt = L.TILES[tile]

    @property
    def handle(self):
        return self._h

    @property
    def num_tiles(self) -> int:
        return int(L.lib().loops_merge_plan_num_tiles(self._h))

    @property
    def self_complete(self) -> bool:
        """True when the planned SpMV runs as one kernel (no row needs a carry-out; loops_merge_plan_self_complete)."""
        return bool(L.lib().loops_merge_plan_self_complete(self._h))

    def coords(self) -> np.ndarray:
        out = np.zeros((self.num_tiles + 1, 2), np.uint32)
        L.check(L.lib().loops_merge_plan_coords(self._h, out.ctypes.data_as(C.c_void_p)), "loops_merge_plan_coords")
        return out

    def refresh(self, csr: CSR):
        L.check(L.lib().loops_merge_plan_refresh(self._h, _ptr(csr.offsets), _stream()), "loops_merge_plan_refresh")

    def close(self):
        if self._h:
            L.lib().loops_merge_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _suffix(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError("values must be float32 or float64")


def spmv(schedule: str, csr: CSR, x: torch.Tensor, y: torch.Tensor | None = None) -> torch.Tensor:
    """y = A x with the tuned path of the named schedule (asynchronous on the current stream)."""
    if y is None:
        y = torch.empty(csr.rows, dtype=csr.values.dtype, device=csr.values.device)
    csr.check(x, y)
    fn = getattr(L.lib(), "loops_spmv_csr_" + _suffix(csr.values))
    L.check(fn(L.SCHEDULES[schedule], csr.rows, csr.cols, csr.nnzs, _ptr(csr.offsets), _ptr(csr.indices),
               _ptr(csr.values), _ptr(x), _ptr(y), _stream()), "loops_spmv_csr(" + schedule + ")")
    return y


def merge_path_flat(csr: CSR, x, y=None, plan: MergePathPlan | None = None, variant: int = 0):
    """algorithms::spmv::merge_path_flat.  With ``plan`` only the fused kernel + fix-up run (the
    region the reference times); without, the coordinates are rebuilt first, as the reference
    wrapper does on every call."""
    if plan is None:
        return spmv("merge_path_flat", csr, x, y)
    if y is None:
        y = torch.empty(csr.rows, dtype=csr.values.dtype, device=csr.values.device)
    csr.check(x, y)
    fn = getattr(L.lib(), "loops_spmv_merge_path_" + _suffix(csr.values))
    L.check(fn(plan.handle, variant, csr.rows, csr.cols, csr.nnzs, _ptr(csr.offsets), _ptr(csr.indices),
               _ptr(csr.values), _ptr(x), _ptr(y), _stream()), "loops_spmv_merge_path")
    return y


def _peer_array(peers):
    """HOST array of device pointers (float* const*) from tensors / integers: where this shard's y[0] lives in each peer."""
    ptrs = [int(t.data_ptr()) if hasattr(t, "data_ptr") else int(t) for t in peers]
    assert len(ptrs) <= 7, "at most 7 peers (8 GPUs per node)"
    return (C.c_void_p * max(len(ptrs), 1))(*ptrs), len(ptrs)


def merge_path_flat_fanout(csr: CSR, x, y, plan: MergePathPlan, peers):
    """merge_path_flat whose finished rows also go to ``peers`` (loops_spmv_merge_path_fanout_f32): tensors or raw
    device pointers, one per peer GPU, each addressing where THIS shard's y[0] lives in that peer's full-length y
    (peer-mapped memory).  The allgatherv(y) of a row-range sharded SpMV issued from the kernels' epilogue;
    ``plan`` must use tile 512x8, f32 only."""
    csr.check(x, y)
    arr, n = _peer_array(peers)
    L.check(L.lib().loops_spmv_merge_path_fanout_f32(plan.handle, csr.rows, csr.cols, csr.nnzs, _ptr(csr.offsets),
                                                     _ptr(csr.indices), _ptr(csr.values), _ptr(x), _ptr(y), n, arr, _stream()),
            "loops_spmv_merge_path_fanout_f32")
    return y


def enable_peer_access(peer_device: int):
    L.check(L.lib().loops_enable_peer_access(int(peer_device)), "loops_enable_peer_access")


def merge_path_flat_stage(csr: CSR, x, y, plan: MergePathPlan, stage: int, variant: int = 0):
    """One kernel of the planned merge_path_flat SpMV (0: fused tile kernel, 1: fix-up)."""
    L.check(L.lib().loops_spmv_merge_path_stage_f32(plan.handle, variant, stage, csr.rows, csr.cols, csr.nnzs,
                                                    _ptr(csr.offsets), _ptr(csr.indices), _ptr(csr.values), _ptr(x),
                                                    _ptr(y), _stream()), "loops_spmv_merge_path_stage_f32")
    return y


def work_oriented(csr, x, y=None, plan: MergePathPlan | None = None):
    """algorithms::spmv::work_oriented; with a held ``plan`` (tile 256x8) the coordinate pre-pass is skipped."""
    if plan is None:
        return spmv("work_oriented", csr, x, y)
    if y is None:
        y = torch.empty(csr.rows, dtype=csr.values.dtype, device=csr.values.device)
    csr.check(x, y)
    fn = getattr(L.lib(), "loops_spmv_work_oriented_" + _suffix(csr.values))
    L.check(fn(plan.handle, csr.rows, csr.cols, csr.nnzs, _ptr(csr.offsets), _ptr(csr.indices), _ptr(csr.values), _ptr(x),
               _ptr(y), _stream()), "loops_spmv_work_oriented")
    return y


def thread_mapped(csr, x, y=None):
    return spmv("thread_mapped", csr, x, y)


def group_mapped(csr, x, y=None):
    return spmv("group_mapped", csr, x, y)


def original(csr, x, y=None):
    return spmv("original", csr, x, y)


def flat_partitioned(csr, x, y=None):
    return spmv("flat_partitioned", csr, x, y)


def spmv_schedule_api(schedule: str, csr: CSR, x, y=None, tile: str = "256x8"):
    """The reference-shaped kernels written against schedule::setup<> (atomics; y is zero-filled
    here, as the reference's callers do)."""
    assert csr.values.dtype == torch.float32
    if y is None:
        y = torch.zeros(csr.rows, dtype=torch.float32, device=csr.values.device)
    else:
        y.zero_()
    csr.check(x, y)
    L.check(L.lib().loops_spmv_csr_schedule_api_f32(L.SCHEDULES[schedule], L.TILES[tile][0], csr.rows, csr.cols,
                                                    csr.nnzs, _ptr(csr.offsets), _ptr(csr.indices), _ptr(csr.values),
                                                    _ptr(x), _ptr(y), _stream()), "loops_spmv_csr_schedule_api")
    return y


class PanelBinnedPlan:
    """Panel-binned copy of a CSR (loops_panel_plan_*; include/loops/kernels/panel_binned.hxx): SpMV without a memory gather
    -- x panels in LDS, products streamed, one wavefront per sub-band of rows adds them up in LDS.  For x far larger than
    the per-XCD L2."""

    def __init__(self, csr: CSR, subband_rows: int = 0, panel_columns: int = 0, compact: int | None = None):
        """``compact``: None = automatic, False = one B-order slot per nonzero, True = kernel A pre-sums runs of equal
        (row, panel) and the B order holds one slot per run (loops_panel_plan_create_layout_*)."""
        assert csr.values.dtype in (torch.float32, torch.float64)
        self.dtype = csr.values.dtype
        self._sfx = _suffix(csr.values)
        self.rows, self.cols, self.nnz = csr.rows, csr.cols, csr.nnzs
        self._h = C.c_void_p()
        create = getattr(L.lib(), "loops_panel_plan_create_layout_" + self._sfx)
        L.check(create(csr.rows, csr.cols, csr.nnzs, _ptr(csr.offsets), _ptr(csr.indices), _ptr(csr.values), int(panel_columns),
                       int(subband_rows), -1 if compact is None else int(bool(compact)), _stream(), C.byref(self._h)),
                "loops_panel_plan_create_layout")
        info = (C.c_int * 7)()
        L.check(L.lib().loops_panel_plan_info(self._h, info), "loops_panel_plan_info")
        self.W, self.Hw, self.num_panels, self.num_subbands, self.padded, self.num_chunks, _ = list(info)
        lay = (C.c_longlong * 4)()
        L.check(L.lib().loops_panel_plan_layout(self._h, lay), "loops_panel_plan_layout")
        self.compact, self.runs, self.padded_b, self.a_window = bool(lay[0]), int(lay[1]), int(lay[2]), int(lay[3])
        # row blocks (matrices of 2^28 nonzeros or more, automatic parameters): independent copies run back to back
        n = C.c_int(0)
        L.check(L.lib().loops_panel_plan_row_blocks(self._h, C.byref(n), None), "loops_panel_plan_row_blocks")
        bounds = (C.c_int * (n.value + 1))()
        L.check(L.lib().loops_panel_plan_row_blocks(self._h, C.byref(n), bounds), "loops_panel_plan_row_blocks")
        self.row_blocks, self.row_block_bounds = n.value, list(bounds)

    @property
    def handle(self):
        return self._h

    def arrays(self):
        """(values, col16, dst4, row16, perm, subband_start) copied to the host: values / col16 / perm [padded] and dst4
        [padded / 4] in (panel, sub-band) order, row16 [padded_b] in (sub-band, panel) order (compact: one slot per run, col16
        bit 15 = run end, dst4 bit 31 = the group holds padding)."""
        val = np.zeros(self.padded, np.float32 if self.dtype == torch.float32 else np.float64)
        col16, row16 = np.zeros(self.padded, np.uint16), np.zeros(self.padded_b, np.uint16)
        perm, dst4 = np.zeros(self.padded, np.int32), np.zeros(self.padded // 4, np.int32)
        bstart = np.zeros(self.num_subbands + 1, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        L.check(L.lib().loops_panel_plan_arrays(self._h, p(val), p(col16), p(dst4), p(row16), p(perm), p(bstart)), "loops_panel_plan_arrays")
        return val, col16, dst4, row16, perm, bstart

    def windows(self):
        """(window_start [subbands + 1], windows [n, 2] = {first item, items | packed << 16}, segment_start [subbands * panels
        + 1]) copied to the host: kernel B's work list (loops_panel_plan_windows)."""
        ws = np.zeros(self.num_subbands + 1, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        L.check(L.lib().loops_panel_plan_windows(self._h, p(ws), None, None), "loops_panel_plan_windows")
        wins = np.zeros((int(ws[-1]), 2), np.int32)
        segb = np.zeros(self.num_subbands * self.num_panels + 1, np.int32)
        L.check(L.lib().loops_panel_plan_windows(self._h, p(ws), p(wins) if wins.size else None, p(segb)), "loops_panel_plan_windows")
        return ws, wins, segb

    def refresh_values(self, values: torch.Tensor):
        assert values.dtype == self.dtype and values.numel() == self.nnz
        L.check(getattr(L.lib(), "loops_panel_plan_refresh_values_" + self._sfx)(self._h, _ptr(values), _stream()),
                "loops_panel_plan_refresh_values")

    def _check(self, x, y):
        assert x.dtype == self.dtype and y.dtype == self.dtype and x.numel() >= self.cols and y.numel() >= self.rows
        assert x.is_contiguous() and y.is_contiguous()

    def spmv(self, x: torch.Tensor, y: torch.Tensor | None = None) -> torch.Tensor:
        if y is None:
            y = torch.empty(self.rows, dtype=self.dtype, device=x.device)
        self._check(x, y)
        L.check(getattr(L.lib(), "loops_spmv_panel_" + self._sfx)(self._h, _ptr(x), _ptr(y), _stream()), "loops_spmv_panel")
        return y

    def spmv_stage(self, stage: int, x, y):
        L.check(getattr(L.lib(), "loops_spmv_panel_stage_" + self._sfx)(self._h, stage, _ptr(x), _ptr(y), _stream()), "loops_spmv_panel_stage")
        return y

    def spmv_fanout(self, x, y, peers):
        """``spmv`` whose sub-band store also goes to ``peers`` (loops_spmv_panel_fanout_*; see merge_path_flat_fanout)."""
        self._check(x, y)
        arr, n = _peer_array(peers)
        L.check(getattr(L.lib(), "loops_spmv_panel_fanout_" + self._sfx)(self._h, _ptr(x), _ptr(y), n, arr, _stream()),
                "loops_spmv_panel_fanout")
        return y

    def close(self):
        if self._h:
            L.lib().loops_panel_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RowBandPlan:
    """Row-band copy of a CSR (loops_rowband_plan_*; include/loops/kernels/rowband.hxx): the y accumulators of a band of H rows
    live in LDS as fp64 words, the band's nonzeros are sorted by column so that a wavefront's 64 x gathers fall on a few
    neighbouring lines; 3 bytes of row code + column delta per nonzero next to the value.  For x of a few MB, or column locality
    at band scale.  fp32 and fp64 values."""

    STEP = 256

    def __init__(self, csr: CSR, band_rows: int = 0, target_chunks: int = 0):
        assert csr.values.dtype in (torch.float32, torch.float64)
        self.dtype = csr.values.dtype
        self._sfx = _suffix(csr.values)
        self.rows, self.cols, self.nnz = csr.rows, csr.cols, csr.nnzs
        self._h = C.c_void_p()
        L.check(getattr(L.lib(), "loops_rowband_plan_create_" + self._sfx)(csr.rows, csr.cols, csr.nnzs, _ptr(csr.offsets), _ptr(csr.indices),
                                                                            _ptr(csr.values), int(band_rows), int(target_chunks), _stream(),
                                                                            C.byref(self._h)), "loops_rowband_plan_create")
        self._read_info()

    def _read_info(self):
        info = (C.c_int * 8)()
        L.check(L.lib().loops_rowband_plan_info(self._h, info), "loops_rowband_plan_info")
        (self.H, self.num_bands, self.gap_pads, self.steps, self.num_chunks, self.num_partials, self.num_multi,
         self.waves) = list(info)
        self.padded = self.steps * self.STEP

    @property
    def handle(self):
        return self._h

    def set_chunks(self, target_chunks: int = 0):
        """Re-cut the bands into about ``target_chunks`` chunks (0 = automatic) without rebuilding the layout."""
        L.check(L.lib().loops_rowband_plan_set_chunks(self._h, int(target_chunks)), "loops_rowband_plan_set_chunks")
        self._read_info()

    def tune(self, repeats: int = 10):
        """Time the product with 8 and 16 wavefronts per workgroup and keep the faster (loops_rowband_plan_tune) -> (ms8, ms16)."""
        ms = (C.c_float * 2)()
        L.check(L.lib().loops_rowband_plan_tune(self._h, int(repeats), ms, _stream()), "loops_rowband_plan_tune")
        self._read_info()
        return float(ms[0]), float(ms[1])

    def set_waves(self, waves: int):
        L.check(L.lib().loops_rowband_plan_set_waves(self._h, int(waves)), "loops_rowband_plan_set_waves")
        self._read_info()

    def arrays(self):
        """(values, row16, delta8, perm, stepbase [steps, 4], chunks [n, 4], multi [m, 3], hubs [bands, 33]) copied to the host."""
        val = np.zeros(self.padded, np.float32 if self.dtype == torch.float32 else np.float64)
        row16, delta8 = np.zeros(self.padded, np.uint16), np.zeros(self.padded, np.uint8)
        perm, stepbase = np.zeros(self.padded, np.int32), np.zeros((self.steps, 4), np.int32)
        chunks, multi = np.zeros((self.num_chunks, 4), np.int32), np.zeros((self.num_multi, 3), np.int32)
        hubs = np.zeros((self.num_bands, 33), np.uint16)
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None  # noqa: E731
        L.check(L.lib().loops_rowband_plan_arrays(self._h, p(val), p(row16), p(delta8), p(perm), p(stepbase), p(chunks), p(multi), p(hubs)),
                "loops_rowband_plan_arrays")
        return val, row16, delta8, perm, stepbase, chunks, multi, hubs

    def refresh_values(self, values: torch.Tensor):
        assert values.dtype == self.dtype and values.numel() == self.nnz
        L.check(getattr(L.lib(), "loops_rowband_plan_refresh_values_" + self._sfx)(self._h, _ptr(values), _stream()), "loops_rowband_plan_refresh_values")

    def _check(self, x, y):
        assert x.dtype == self.dtype and y.dtype == self.dtype and x.numel() >= self.cols and y.numel() >= self.rows
        assert x.is_contiguous() and y.is_contiguous()

    def spmv(self, x: torch.Tensor, y: torch.Tensor | None = None) -> torch.Tensor:
        if y is None:
            y = torch.empty(self.rows, dtype=self.dtype, device=x.device)
        self._check(x, y)
        L.check(getattr(L.lib(), "loops_spmv_rowband_" + self._sfx)(self._h, _ptr(x), _ptr(y), _stream()), "loops_spmv_rowband")
        return y

    def spmv_stage(self, stage: int, x, y):
        L.check(getattr(L.lib(), "loops_spmv_rowband_stage_" + self._sfx)(self._h, stage, _ptr(x), _ptr(y), _stream()), "loops_spmv_rowband_stage")
        return y

    def spmv_fanout(self, x, y, peers):
        """``spmv`` whose row stores also go to ``peers`` (loops_spmv_rowband_fanout_f32; see merge_path_flat_fanout)."""
        self._check(x, y)
        assert self.dtype == torch.float32, "the fan-out form is compiled for fp32"
        arr, n = _peer_array(peers)
        L.check(L.lib().loops_spmv_rowband_fanout_f32(self._h, _ptr(x), _ptr(y), n, arr, _stream()), "loops_spmv_rowband_fanout_f32")
        return y

    def close(self):
        if self._h:
            L.lib().loops_rowband_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SpmvPlan:
    """loops_spmv_plan_*: tile shape AND layout of one matrix chosen at plan time.  ``measure``: time the candidates on the
    device; ``allow_copy``: the plan may hold a re-ordered copy of the matrix (row-band, panel-binned) when that is faster;
    ``deterministic``: only layouts whose summation order is fixed (no row-band copy).  ``spmv(x, y)`` runs whatever was chosen;
    ``info`` says what that is."""

    LAYOUTS = {0: "csr", 2: "panel_binned", 3: "row_band"}

    def __init__(self, csr: CSR, allow_copy: bool = True, measure: bool = True, repeats: int = 10, deterministic: bool = False):
        self.csr = csr
        self._sfx = _suffix(csr.values)
        self._h = C.c_void_p()
        flags = (1 if measure else 0) | (2 if allow_copy else 0) | (4 if deterministic else 0)  # LOOPS_PLAN_MEASURE | _ALLOW_COPY | _DETERMINISTIC
        create = getattr(L.lib(), "loops_spmv_plan_create_" + self._sfx)
        L.check(create(csr.rows, csr.cols, csr.nnzs, _ptr(csr.offsets), _ptr(csr.indices), _ptr(csr.values), flags, repeats,
                       _stream(), C.byref(self._h)), "loops_spmv_plan_create")
        layout, tile, blocks = C.c_int(), C.c_int(), C.c_int()
        ms = (C.c_float * 4)()
        L.check(L.lib().loops_spmv_plan_info(self._h, C.byref(layout), C.byref(tile), C.byref(blocks), ms), "loops_spmv_plan_info")
        names = {cfg: name for name, (cfg, _, _) in L.TILES.items()}
        self.layout, self.tile, self.num_blocks = self.LAYOUTS[layout.value], names[tile.value], blocks.value
        self.measured_ms = {k: (round(float(v), 5) if v >= 0 else None) for k, v in zip(("csr_256x8", "csr_512x8", "row_band", "panel_binned"), ms)}
        variant, ms_phased = C.c_int(), C.c_float()
        L.check(L.lib().loops_spmv_plan_variant(self._h, C.byref(variant), C.byref(ms_phased)), "loops_spmv_plan_variant")
        self.variant = variant.value  # CSR layout: 0 or L.VARIANT_PHASED (phased x gathers)
        self.measured_ms["csr_phased"] = round(float(ms_phased.value), 5) if ms_phased.value >= 0 else None

    @property
    def info(self):
        return {"layout": self.layout, "tile": self.tile + ("+phased" if self.variant == L.VARIANT_PHASED else ""),
                "bands_or_panels": self.num_blocks, "measured_ms": self.measured_ms}

    def spmv(self, x: torch.Tensor, y: torch.Tensor | None = None) -> torch.Tensor:
        c = self.csr
        if y is None:
            y = torch.empty(c.rows, dtype=c.values.dtype, device=c.values.device)
        c.check(x, y)
        fn = getattr(L.lib(), "loops_spmv_planned_" + self._sfx)
        L.check(fn(self._h, _ptr(c.offsets), _ptr(c.indices), _ptr(c.values), _ptr(x), _ptr(y), _stream()), "loops_spmv_planned")
        return y

    def refresh_values(self):
        """The matrix's values changed in place (same structure): bring a held copy up to date."""
        fn = getattr(L.lib(), "loops_spmv_plan_refresh_values_" + self._sfx)
        L.check(fn(self._h, _ptr(self.csr.values), _stream()), "loops_spmv_plan_refresh_values")

    def close(self):
        if self._h:
            L.lib().loops_spmv_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------ SpMM
def spmm(csr: CSR, B: torch.Tensor, Cm: torch.Tensor | None = None, schedule: str = "merge_path_flat",
         plan: MergePathPlan | None = None) -> torch.Tensor:
    """C = A B with dense row-major B [cols, n] and C [rows, n] (algorithms::spmm).  ``schedule``:
    "merge_path_flat" (tuned) or "thread_mapped" (reference-shaped).  With ``plan`` the coordinate
    pre-pass is skipped (merge_path_flat, f32)."""
    assert B.dim() == 2 and B.is_contiguous() and B.shape[0] == csr.cols and B.dtype == csr.values.dtype
    n = B.shape[1]
    if Cm is None:
        Cm = torch.empty((csr.rows, n), dtype=B.dtype, device=B.device)
    assert Cm.is_contiguous() and tuple(Cm.shape) == (csr.rows, n) and Cm.dtype == B.dtype
    if plan is not None:
        assert schedule == "merge_path_flat"
        fn = getattr(L.lib(), "loops_spmm_merge_path_" + _suffix(csr.values))
        L.check(fn(plan.handle, csr.rows, csr.cols, csr.nnzs, _ptr(csr.offsets), _ptr(csr.indices), _ptr(csr.values),
                   _ptr(B), n, _ptr(Cm), _stream()), "loops_spmm_merge_path")
        return Cm
    fn = getattr(L.lib(), "loops_spmm_csr_" + _suffix(csr.values))
    L.check(fn(L.SCHEDULES[schedule], csr.rows, csr.cols, csr.nnzs, _ptr(csr.offsets), _ptr(csr.indices),
               _ptr(csr.values), _ptr(B), n, _ptr(Cm), _stream()), "loops_spmm_csr(" + schedule + ")")
    return Cm


# ------------------------------------------------------------------------------- schedule dumps
def dump_merge_path(csr: CSR, tile: str = "256x8", use_plan: bool = True):
    cfg, tpb, ipt = L.TILES[tile]
    total = csr.rows + csr.nnzs
    m = total // (tpb * ipt) + (1 if total % (tpb * ipt) else 0)
    dev = csr.offsets.device
    ts = torch.zeros(max(2 * m * tpb, 1), dtype=torch.int32, device=dev)
    owner = torch.full((max(csr.nnzs, 1),), -1, dtype=torch.int32, device=dev)
    row = torch.full((max(csr.nnzs, 1),), -1, dtype=torch.int32, device=dev)
    visits = torch.zeros(max(csr.nnzs, 1), dtype=torch.int32, device=dev)
    L.check(L.lib().loops_schedule_dump_merge_path(cfg, int(use_plan), csr.rows, csr.nnzs, _ptr(csr.offsets), _ptr(ts),
                                                   _ptr(owner), _ptr(row), _ptr(visits), _stream()),
            "loops_schedule_dump_merge_path")
    torch.cuda.synchronize()
    n = csr.nnzs
    return (ts.cpu().numpy().view(np.uint32).reshape(-1, 2)[: m * tpb], owner.cpu().numpy()[:n],
            row.cpu().numpy()[:n], visits.cpu().numpy()[:n])


def dump_work_oriented(csr: CSR, grid_blocks: int):
    dev = csr.offsets.device
    tm = torch.zeros(4 * grid_blocks * 256, dtype=torch.int32, device=dev)
    owner = torch.full((max(csr.nnzs, 1),), -1, dtype=torch.int32, device=dev)
    row = torch.full((max(csr.nnzs, 1),), -1, dtype=torch.int32, device=dev)
    visits = torch.zeros(max(csr.nnzs, 1), dtype=torch.int32, device=dev)
    L.check(L.lib().loops_schedule_dump_work_oriented(grid_blocks, csr.rows, csr.nnzs, _ptr(csr.offsets), _ptr(tm),
                                                      _ptr(owner), _ptr(row), _ptr(visits), _stream()),
            "loops_schedule_dump_work_oriented")
    torch.cuda.synchronize()
    n = csr.nnzs
    return tm.cpu().numpy().reshape(-1, 4), owner.cpu().numpy()[:n], row.cpu().numpy()[:n], visits.cpu().numpy()[:n]


def dump_group_mapped(csr: CSR, group_size: int = 256):
    dev = csr.offsets.device
    owner = torch.full((max(csr.nnzs, 1),), -1, dtype=torch.int32, device=dev)
    row = torch.full((max(csr.nnzs, 1),), -1, dtype=torch.int32, device=dev)
    visits = torch.zeros(max(csr.nnzs, 1), dtype=torch.int32, device=dev)
    L.check(L.lib().loops_schedule_dump_group_mapped(group_size, csr.rows, csr.nnzs, _ptr(csr.offsets), _ptr(owner),
                                                     _ptr(row), _ptr(visits), _stream()),
            "loops_schedule_dump_group_mapped")
    torch.cuda.synchronize()
    n = csr.nnzs
    return owner.cpu().numpy()[:n], row.cpu().numpy()[:n], visits.cpu().numpy()[:n]


def work_oriented_grid() -> int:
    out = C.c_int()
    L.check(L.lib().loops_work_oriented_grid(C.byref(out)), "loops_work_oriented_grid")
    return out.value


# ------------------------------------------------------------------------------- BCSR
@dataclass
class BCSR:
    """bcsr_t<R, C, int, int, float> on the device (reference include/loops/container/bcsr.hxx:60)."""
    R: int
    C: int
    rows: int
    cols: int
    block_offsets: torch.Tensor
    block_cols: torch.Tensor
    values: torch.Tensor

    @property
    def num_block_rows(self):
        return int(self.block_offsets.numel()) - 1

    @property
    def num_blocks(self):
        return int(self.block_cols.numel())

    @property
    def num_block_cols(self):
        return (self.cols + self.C - 1) // self.C


BCSR_MODES = {"thread": 0, "mfma": 1, "coalesced": 2, "tuned": 3, "merge_path": 4}


def bcsr_row_length_class(b: BCSR) -> str:
    """loops_bcsr_row_length_class: "even" / "skewed" block-row lengths by the rule of loops_spmv_bcsr_f32's mode "tuned" (synchronises)."""
    out = C.c_int(0)
    L.check(L.lib().loops_bcsr_row_length_class(b.num_block_rows, b.num_blocks, _ptr(b.block_offsets), C.byref(out), _stream()),
            "loops_bcsr_row_length_class")
    return {1: "even", 2: "skewed"}[out.value]


def bcsr_thread_mapped(b: BCSR, x_padded: torch.Tensor, y: torch.Tensor | None = None, mfma: bool | int | str = False):
    """algorithms::spmv::bcsr_thread_mapped<R, C>.  ``mfma`` is the `mode` of loops_spmv_bcsr_*: False / "thread" =
    thread per block-row (the reference's kernel shape), True / "mfma" = the 4x4 fp32 MFMA kernel, "coalesced" = the
    lane-group kernel of any shape / precision, "tuned" = what the C++ wrapper launches (4x4 fp32: MFMA on even block-row lengths, merge-path tiles on
    skewed ones and until the matrix's class is known; coalesced otherwise), "merge_path" = those tiles; integers pass through (tuning shapes, include/loops_amd.h)."""
    if isinstance(mfma, str):
        mfma = BCSR_MODES[mfma]
    if y is None:
        y = torch.empty(b.rows, dtype=b.values.dtype, device=b.values.device)
    assert x_padded.numel() >= b.num_block_cols * b.C and x_padded.dtype == b.values.dtype == y.dtype
    fn = getattr(L.lib(), "loops_spmv_bcsr_" + _suffix(b.values))
    L.check(fn(b.R, b.C, int(mfma), b.rows, b.num_block_rows, b.num_blocks, _ptr(b.block_offsets), _ptr(b.block_cols),
               _ptr(b.values), _ptr(x_padded), _ptr(y), _stream()), "loops_spmv_bcsr")
    return y


class BCSRBandPlan:
    """Block-band copy of a 4 x 4 fp32 BCSR (loops_bcsr_band_plan_*; include/loops/kernels/bcsr_band.hxx): bands of HB block-rows
    whose 4 HB row sums live in LDS as fp64 words, the band's blocks sorted by block column so that the x gathers of a
    wavefront share lines; block inner products on v_mfma_f32_4x4x1.  A held plan for bcsr_thread_mapped<4, 4> products."""

    STEP = 16

    def __init__(self, b: BCSR, band_block_rows: int = 0, target_chunks: int = 0):
        assert b.R == 4 and b.C == 4 and b.values.dtype == torch.float32
        self.rows, self.num_block_rows, self.num_block_cols, self.num_blocks = b.rows, b.num_block_rows, b.num_block_cols, b.num_blocks
        self._h = C.c_void_p()
        L.check(L.lib().loops_bcsr_band_plan_create_f32(b.rows, b.num_block_rows, b.num_block_cols, b.num_blocks, _ptr(b.block_offsets),
                                                        _ptr(b.block_cols), _ptr(b.values), int(band_block_rows), int(target_chunks),
                                                        _stream(), C.byref(self._h)), "loops_bcsr_band_plan_create_f32")
        self._read_info()

    def _read_info(self):
        info = (C.c_int * 10)()
        L.check(L.lib().loops_bcsr_band_plan_info(self._h, info), "loops_bcsr_band_plan_info")
        (self.HB, self.num_bands, self.cbits, self.steps, self.num_chunks, self.num_partials, self.num_multi, self.waves, self.unroll,
         self.nt) = list(info)
        self.slots = self.steps * self.STEP

    def set_chunks(self, target_chunks: int = 0):
        L.check(L.lib().loops_bcsr_band_plan_set_chunks(self._h, int(target_chunks)), "loops_bcsr_band_plan_set_chunks")
        self._read_info()

    def tune(self, repeats: int = 10):
        """Time every compiled kernel shape and keep the fastest -> {(waves, unroll, nt): ms}."""
        ms = (C.c_float * 12)()
        L.check(L.lib().loops_bcsr_band_plan_tune(self._h, int(repeats), ms, _stream()), "loops_bcsr_band_plan_tune")
        self._read_info()
        keys = [(w, u, nt) for w in (8, 16) for u in (1, 2, 4) for nt in (0, 1)]
        return {k: float(v) for k, v in zip(keys, ms)}

    def set_shape(self, waves: int, unroll: int, nt: int):
        L.check(L.lib().loops_bcsr_band_plan_set_shape(self._h, int(waves), int(unroll), int(nt)), "loops_bcsr_band_plan_set_shape")
        self._read_info()

    def arrays(self):
        """(values [slots, 4, 4], words, perm, chunks [n, 4], multi [m, 3], hubs [bands, 33]) copied to the host."""
        val, words, perm = np.zeros((self.slots, 4, 4), np.float32), np.zeros(self.slots, np.uint32), np.zeros(self.slots, np.int32)
        chunks, multi = np.zeros((self.num_chunks, 4), np.int32), np.zeros((self.num_multi, 3), np.int32)
        hubs = np.zeros((self.num_bands, 33), np.uint16)
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None  # noqa: E731
        L.check(L.lib().loops_bcsr_band_plan_arrays(self._h, p(val), p(words), p(perm), p(chunks), p(multi), p(hubs)), "loops_bcsr_band_plan_arrays")
        return val, words, perm, chunks, multi, hubs

    def refresh_values(self, values: torch.Tensor):
        assert values.dtype == torch.float32 and values.numel() == self.num_blocks * 16
        L.check(L.lib().loops_bcsr_band_plan_refresh_values_f32(self._h, _ptr(values), _stream()), "loops_bcsr_band_plan_refresh_values_f32")

    def spmv(self, x_padded: torch.Tensor, y: torch.Tensor | None = None) -> torch.Tensor:
        if y is None:
            y = torch.empty(self.rows, dtype=torch.float32, device=x_padded.device)
        assert x_padded.dtype == torch.float32 and y.dtype == torch.float32 and x_padded.numel() >= 4 * self.num_block_cols and y.numel() >= self.rows
        assert x_padded.is_contiguous() and y.is_contiguous()
        L.check(L.lib().loops_spmv_bcsr_band_f32(self._h, _ptr(x_padded), _ptr(y), _stream()), "loops_spmv_bcsr_band_f32")
        return y

    def spmv_stage(self, stage: int, x_padded, y):
        L.check(L.lib().loops_spmv_bcsr_band_stage_f32(self._h, stage, _ptr(x_padded), _ptr(y), _stream()), "loops_spmv_bcsr_band_stage_f32")
        return y

    def close(self):
        if self._h:
            L.lib().loops_bcsr_band_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def coo_spmv(rows: int, cols: int, row_indices, col_indices, values, x, y=None, tuned: bool = True):
    """COO SpMV (loops_spmv_coo_f32 / _f64): ``tuned`` = one atomic per run of equal row indices (y zero-filled
    inside); otherwise the reference shape, one atomic per nonzero into a y zero-filled here."""
    if y is None:
        y = torch.empty(rows, dtype=values.dtype, device=x.device)
    if not tuned:
        y.zero_()
    fn = getattr(L.lib(), "loops_spmv_coo_" + _suffix(values))
    L.check(fn(int(tuned), rows, cols, values.numel(), _ptr(row_indices), _ptr(col_indices), _ptr(values), _ptr(x), _ptr(y),
               _stream()), "loops_spmv_coo")
    return y


ELL_MODES = {"thread": 0, "row": 1, "merge_path": 2}


def ell_spmv(rows: int, cols: int, pitch: int, indices, values, x, y=None, tuned: bool | str = True):
    """ELL SpMV (loops_spmv_ell_f32 / _f64) over the reference's row-major rows x pitch arrays (padding: column -1).
    ``tuned``: False / "thread" = lane per row (reference shape), True / "row" = G lanes per row with 16-byte loads,
    "merge_path" = the merge_path_flat schedule over the ELL cells on the fused engine (algorithms::spmv::ell_merge_path)."""
    if y is None:
        y = torch.empty(rows, dtype=values.dtype, device=x.device)
    mode = ELL_MODES[tuned] if isinstance(tuned, str) else int(bool(tuned))
    fn = getattr(L.lib(), "loops_spmv_ell_" + _suffix(values))
    L.check(fn(mode, rows, cols, pitch, _ptr(indices), _ptr(values), _ptr(x), _ptr(y), _stream()), "loops_spmv_ell")
    return y


def dia_spmv(rows: int, cols: int, diag_offsets, values, x, y=None, stride: int | None = None, tuned: bool = True):
    """DIA SpMV (loops_spmv_dia_f32 / _f64): values column-major [num_diagonals x stride], diag_offsets[d] = col - row.
    ``tuned``: four rows per lane, 16-byte loads; otherwise lane per row (algorithms::spmv::dia_thread_mapped)."""
    if y is None:
        y = torch.empty(rows, dtype=values.dtype, device=x.device)
    nd = int(diag_offsets.numel())
    stride = rows if stride is None else stride
    fn = getattr(L.lib(), "loops_spmv_dia_" + _suffix(values))
    L.check(fn(int(tuned), rows, cols, nd, stride, _ptr(diag_offsets), _ptr(values), _ptr(x), _ptr(y), _stream()),
            "loops_spmv_dia")
    return y


def autotune_merge_path(csr: CSR, x, repeats: int = 5):
    """Times the planned merge_path_flat SpMV with every compiled tile shape on this matrix
    (loops_autotune_merge_path_f32).  Returns (best tile name, {tile name: ms})."""
    y = torch.empty(csr.rows, dtype=torch.float32, device=x.device)
    best = C.c_int()
    ms = (C.c_float * 6)()
    L.check(L.lib().loops_autotune_merge_path_f32(csr.rows, csr.cols, csr.nnzs, _ptr(csr.offsets), _ptr(csr.indices),
                                                  _ptr(csr.values), _ptr(x), _ptr(y), repeats, _stream(), C.byref(best), ms),
            "loops_autotune_merge_path_f32")
    names = {cfg: name for name, (cfg, _, _) in L.TILES.items()}
    return names[best.value], {names[i]: ms[i] for i in range(6) if ms[i] >= 0 and i in names}


def columns_look_scattered(csr: CSR) -> bool:
    """loops_columns_look_scattered: the structural guess at whether the phased-gather kernel pays on this matrix (what the
    plan-less C++ wrapper consults); measuring -- autotune_merge_path_variants, SpmvPlan(measure=True) -- is the reliable way."""
    out = C.c_int()
    L.check(L.lib().loops_columns_look_scattered(csr.cols, csr.nnzs, _ptr(csr.indices), csr.values.element_size(), _stream(),
                                                 C.byref(out)), "loops_columns_look_scattered")
    return bool(out.value)


def autotune_merge_path_variants(csr: CSR, x, repeats: int = 5):
    """The same over tile shapes AND kernel variants (loops_autotune_merge_path_variants_f32): the phased-gather twin of the
    shapes that have one is timed too.  Returns (best tile name, best variant, {name: ms}) with phased entries named
    "<tile>+phased"."""
    y = torch.empty(csr.rows, dtype=torch.float32, device=x.device)
    best, variant = C.c_int(), C.c_int()
    ms = (C.c_float * 12)()
    L.check(L.lib().loops_autotune_merge_path_variants_f32(csr.rows, csr.cols, csr.nnzs, _ptr(csr.offsets), _ptr(csr.indices),
                                                           _ptr(csr.values), _ptr(x), _ptr(y), repeats, _stream(), C.byref(best),
                                                           C.byref(variant), ms), "loops_autotune_merge_path_variants_f32")
    names = {cfg: name for name, (cfg, _, _) in L.TILES.items()}
    table = {names[i]: ms[i] for i in range(6) if ms[i] >= 0 and i in names}
    table.update({names[i] + "+phased": ms[6 + i] for i in range(6) if ms[6 + i] >= 0 and i in names})
    return names[best.value], variant.value, table


class CSCPlan:
    """loops_csc_plan_*: a CSC matrix held for repeated products -- transposed to CSR on the device once (the CSC product
    itself is one global atomic per nonzero), then a SpMV plan over that copy (``allow_copy`` / ``measure`` as in SpmvPlan)."""
    _coo = False

    def __init__(self, rows: int, cols: int, col_offsets, row_indices, values, allow_copy: bool = True, measure: bool = True,
                 repeats: int = 10):
        assert values.dtype in (torch.float32, torch.float64)
        self.rows, self.cols, self.nnz, self.dtype = rows, cols, values.numel(), values.dtype
        self._sfx = _suffix(values)
        self._h = C.c_void_p()
        flags = (1 if measure else 0) | (2 if allow_copy else 0)
        create = getattr(L.lib(), ("loops_coo_plan_create_" if self._coo else "loops_csc_plan_create_") + self._sfx)
        L.check(create(rows, cols, self.nnz, _ptr(col_offsets), _ptr(row_indices), _ptr(values), flags, repeats, _stream(),
                       C.byref(self._h)), "loops_csc_plan_create")
        layout, tile, blocks = C.c_int(), C.c_int(), C.c_int()
        L.check(L.lib().loops_csc_plan_info(self._h, C.byref(layout), C.byref(tile), C.byref(blocks), None), "loops_csc_plan_info")
        self.layout = SpmvPlan.LAYOUTS[layout.value]

    def spmv(self, x: torch.Tensor, y: torch.Tensor | None = None) -> torch.Tensor:
        if y is None:
            y = torch.empty(self.rows, dtype=self.dtype, device=x.device)
        assert x.dtype == self.dtype and y.dtype == self.dtype and x.numel() >= self.cols and y.numel() >= self.rows
        assert x.is_contiguous() and y.is_contiguous()
        L.check(getattr(L.lib(), "loops_spmv_csc_planned_" + self._sfx)(self._h, _ptr(x), _ptr(y), _stream()), "loops_spmv_csc_planned")
        return y

    def refresh_values(self, values: torch.Tensor):
        assert values.dtype == self.dtype and values.numel() == self.nnz
        L.check(getattr(L.lib(), "loops_csc_plan_refresh_values_" + self._sfx)(self._h, _ptr(values), _stream()),
                "loops_csc_plan_refresh_values")

    def close(self):
        if self._h:
            L.lib().loops_csc_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class COOPlan(CSCPlan):
    """loops_coo_plan_*: COO triplets in any order, sorted into a CSR copy on the device once; then as CSCPlan."""
    _coo = True

    def __init__(self, rows: int, cols: int, row_indices, col_indices, values, allow_copy: bool = True, measure: bool = True,
                 repeats: int = 10):
        # (the C entry takes (row_indices, col_indices): the base class passes its 4th and 5th argument in that order)
        super().__init__(rows, cols, row_indices, col_indices, values, allow_copy, measure, repeats)


def csc_spmv(rows: int, cols: int, col_offsets, row_indices, values, x, y=None, tuned: bool = True):
    """CSC SpMV (loops_spmv_csc_f32 / _f64): ``tuned`` = nonzero-split kernel (y zero-filled inside); otherwise the
    reference shape, lane per column, into a y zero-filled here."""
    if y is None:
        y = torch.empty(rows, dtype=values.dtype, device=x.device)
    if not tuned:
        y.zero_()
    fn = getattr(L.lib(), "loops_spmv_csc_" + _suffix(values))
    L.check(fn(int(tuned), rows, cols, values.numel(), _ptr(col_offsets), _ptr(row_indices), _ptr(values), _ptr(x), _ptr(y),
               _stream()), "loops_spmv_csc")
    return y
