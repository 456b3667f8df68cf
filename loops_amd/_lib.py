"""ctypes binding of libloops_amd.so -- the C ABI declared in include/loops_amd.h.

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no CPU
fallback anywhere in this package: if the shared library is missing, or a kernel cannot be
launched, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("LOOPS_AMD_LIB", os.path.join(_HERE, "libloops_amd.so"))
SRC_PATH = os.path.join(_HERE, "csrc", "loops_c_abi.hip")
PROBES_LIB_PATH = os.environ.get("LOOPS_PROBES_LIB", os.path.join(_HERE, "libloops_probes.so"))
PROBES_SRC_PATH = os.path.join(_HERE, "csrc", "loops_probes.hip")
INCLUDE_DIR = os.path.join(_ROOT, "include")

# enum loops_schedule (include/loops_amd.h)
MERGE_PATH_FLAT, WORK_ORIENTED, THREAD_MAPPED, GROUP_MAPPED, ORIGINAL, FLAT_PARTITIONED = range(6)
SCHEDULES = {
    "merge_path_flat": MERGE_PATH_FLAT, "work_oriented": WORK_ORIENTED, "thread_mapped": THREAD_MAPPED,
    "group_mapped": GROUP_MAPPED, "original": ORIGINAL, "flat_partitioned": FLAT_PARTITIONED,
}
# enum loops_tile_config: name -> (id, threads per block, items per thread)
TILE_AUTO = -1  # LOOPS_TILE_AUTO (loops_merge_plan_create): 256x8 when self-completing with it, 512x8 otherwise
TILES = {"256x8": (0, 256, 8), "128x7": (1, 128, 7), "4x2": (2, 4, 2), "256x7": (3, 256, 7), "512x8": (4, 512, 8), "256x16": (5, 256, 16)}

# every symbol include/loops_amd.h declares (tests/test_c_abi.py checks the export table)
SYMBOLS = [
    "loops_version", "loops_device_compute_units", "loops_release_scratch",
    "loops_merge_plan_create", "loops_merge_plan_destroy", "loops_merge_plan_refresh",
    "loops_merge_plan_num_tiles", "loops_merge_plan_coords", "loops_merge_plan_self_complete",
    "loops_spmv_csr_f32", "loops_spmv_csr_f64", "loops_spmv_merge_path_f32", "loops_spmv_merge_path_f64",
    "loops_spmv_merge_path_stage_f32", "loops_spmv_csr_schedule_api_f32",
    "loops_schedule_dump_merge_path", "loops_schedule_dump_work_oriented", "loops_schedule_dump_group_mapped",
    "loops_work_oriented_grid", "loops_spmv_bcsr_f32", "loops_bcsr_row_length_class",
    "loops_spmm_csr_f32", "loops_spmm_csr_f64", "loops_spmm_merge_path_f32", "loops_spmv_coo_f32", "loops_spmv_ell_f32", "loops_spmv_csc_f32", "loops_autotune_merge_path_f32",
        "loops_spmv_bcsr_f64", "loops_spmm_merge_path_f64", "loops_spmv_coo_f64", "loops_spmv_ell_f64", "loops_spmv_csc_f64",
    "loops_spmv_dia_f32", "loops_spmv_dia_f64",
    "loops_spmv_work_oriented_f32", "loops_spmv_work_oriented_f64", "loops_enable_peer_access", "loops_spmv_merge_path_fanout_f32",
    "loops_spmv_plan_create_f32", "loops_spmv_plan_create_f64", "loops_spmv_plan_destroy", "loops_spmv_plan_info",
    "loops_spmv_plan_refresh_values_f32", "loops_spmv_plan_refresh_values_f64", "loops_spmv_planned_f32", "loops_spmv_planned_f64",
    "loops_panel_plan_create_f32", "loops_panel_plan_create_f64", "loops_panel_plan_destroy", "loops_panel_plan_info",
    "loops_panel_plan_arrays", "loops_panel_plan_windows", "loops_panel_plan_refresh_values_f32", "loops_panel_plan_refresh_values_f64",
    "loops_csc_plan_create_f32", "loops_csc_plan_create_f64", "loops_coo_plan_create_f32", "loops_coo_plan_create_f64", "loops_csc_plan_destroy", "loops_csc_plan_info",
    "loops_csc_plan_refresh_values_f32", "loops_csc_plan_refresh_values_f64", "loops_spmv_csc_planned_f32", "loops_spmv_csc_planned_f64",
    "loops_panel_plan_create_layout_f32", "loops_panel_plan_create_layout_f64", "loops_panel_plan_layout", "loops_panel_plan_row_blocks",
    "loops_row_ranges", "loops_comm_unique_id", "loops_comm_init", "loops_comm_destroy", "loops_comm_error_string",
    "loops_allgatherv_f32", "loops_allgatherv_f64",
    "loops_autotune_merge_path_variants_f32", "loops_spmv_plan_variant", "loops_columns_look_scattered",
    "loops_spmv_panel_f32", "loops_spmv_panel_f64", "loops_spmv_panel_stage_f32", "loops_spmv_panel_stage_f64", "loops_spmv_panel_fanout_f32", "loops_spmv_panel_fanout_f64",
    "loops_rowband_plan_create_f32", "loops_rowband_plan_destroy", "loops_rowband_plan_info", "loops_rowband_plan_arrays",
    "loops_rowband_plan_set_chunks", "loops_rowband_plan_tune", "loops_rowband_plan_set_waves", "loops_rowband_plan_refresh_values_f32", "loops_spmv_rowband_f32", "loops_spmv_rowband_stage_f32",
    "loops_spmv_rowband_fanout_f32",
    "loops_rowband_plan_create_f64", "loops_rowband_plan_refresh_values_f64", "loops_spmv_rowband_f64", "loops_spmv_rowband_stage_f64",
    "loops_bcsr_band_plan_create_f32", "loops_bcsr_band_plan_destroy", "loops_bcsr_band_plan_info", "loops_bcsr_band_plan_arrays",
    "loops_bcsr_band_plan_set_chunks", "loops_bcsr_band_plan_tune", "loops_bcsr_band_plan_set_shape", "loops_bcsr_band_plan_refresh_values_f32",
    "loops_spmv_bcsr_band_f32", "loops_spmv_bcsr_band_stage_f32",
]


VARIANT_PHASED = 8  # include/loops_amd.h LOOPS_VARIANT_PHASED: the default merge-path kernel with phased x gathers


class LoopsError(RuntimeError):
    pass


def source_digest(deps) -> str:
    """sha256 over the CONTENTS of every file a library is compiled from (paths relative to the repository, sorted).
    What decides whether a built library is current: modification times do not survive a snapshot copy to another
    machine, and a `.so` newer than its sources says nothing about which sources it was built from."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(set(deps)):
        h.update(os.path.relpath(path, _ROOT).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def _compile(src: str, out: str, extra_deps=(), force: bool = False, verbose: bool = False, defines=()) -> str:
    """hipcc `src` -> `out` unless `out` was built from exactly these sources (digest recorded next to it in
    `<out>.srcdigest`).  `force` -- or LOOPS_FORCE_BUILD=1 in the environment -- always compiles."""
    deps = [src, *extra_deps]
    for base, _, files in os.walk(INCLUDE_DIR):
        deps += [os.path.join(base, f) for f in files]
    digest = source_digest(deps) + " " + " ".join(sorted(defines))
    stamp = out + ".srcdigest"
    force = force or os.environ.get("LOOPS_FORCE_BUILD", "0") not in ("", "0")
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == digest.strip():
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DLOOPS_TARGET_GFX=0x950",
           *["-D" + d for d in defines], "-I" + INCLUDE_DIR, src, "-o", out]
    if verbose:
        print(" ".join(cmd))
    if os.path.exists(stamp):
        os.remove(stamp)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(digest + "\n")
    return out


def built_from_current_sources(out: str = None, src: str = None, extra_deps=()) -> bool:
    """True when `out` (default: the product library) carries the digest of the sources as they are now."""
    out, src = out or LIB_PATH, src or SRC_PATH
    deps = [src, *extra_deps, *(_abi_parts() if src == SRC_PATH else [])]
    for base, _, files in os.walk(INCLUDE_DIR):
        deps += [os.path.join(base, f) for f in files]
    stamp = out + ".srcdigest"
    return os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().split()[0] == source_digest(deps)


def _abi_parts():
    """The per-family parts of the C ABI that loops_c_abi.hip #includes (loops_amd/csrc/abi_*.inc)."""
    csrc = os.path.dirname(SRC_PATH)
    return sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.startswith("abi_") and f.endswith(".inc"))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libloops_amd.so -- the product -- for gfx950 with hipcc (cross-compiles without a GPU)."""
    return _compile(SRC_PATH, LIB_PATH, _abi_parts(), force=force, verbose=verbose)


def build_probes(force: bool = False, verbose: bool = False) -> str:
    """Compile libloops_probes.so: calibration kernels and experimental instantiations, measurement only
    (loops_amd/csrc/loops_probes.h).  The product library neither links nor loads it."""
    csrc = os.path.dirname(PROBES_SRC_PATH)
    return _compile(PROBES_SRC_PATH, PROBES_LIB_PATH, [os.path.join(csrc, "probes.hxx"), os.path.join(csrc, "loops_probes.h")],
                    force=force, verbose=verbose)


_lib = None


def load_shared(path: str) -> C.CDLL:
    """dlopen a HIP shared library so that it shares ONE HIP runtime with PyTorch.

    PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so; a second copy of
    the runtime in the same process (e.g. /opt/rocm's, pulled in by an extension that is loaded
    before torch) cannot open the device again (hipErrorNoDevice / "No HIP GPUs are available").
    Importing torch first makes the loader bind our DT_NEEDED libamdhip64.so.7 to the copy that
    is already resident."""
    import torch  # noqa: F401  (side effect: torch's HIP runtime is loaded first)
    return C.CDLL(path)


def lib() -> C.CDLL:
    """The loaded library; raises LoopsError (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LoopsError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = load_shared(LIB_PATH)
        L.loops_version.restype = C.c_char_p
        vp, ci = C.c_void_p, C.c_int
        L.loops_merge_plan_create.argtypes = [ci, ci, vp, ci, vp, C.POINTER(vp)]
        L.loops_merge_plan_destroy.argtypes = [vp]
        L.loops_merge_plan_refresh.argtypes = [vp, vp, vp]
        L.loops_merge_plan_num_tiles.argtypes = [vp]
        L.loops_merge_plan_self_complete.argtypes = [vp]
        L.loops_merge_plan_coords.argtypes = [vp, vp]
        for name in ("loops_spmv_csr_f32", "loops_spmv_csr_f64"):
            getattr(L, name).argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]
        for name in ("loops_spmv_merge_path_f32", "loops_spmv_merge_path_f64"):
            getattr(L, name).argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]
        for name in ("loops_spmv_work_oriented_f32", "loops_spmv_work_oriented_f64"):
            getattr(L, name).argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, vp, vp]
        L.loops_enable_peer_access.argtypes = [ci]
        L.loops_spmv_merge_path_fanout_f32.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, vp]
        L.loops_spmv_merge_path_stage_f32.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]
        L.loops_spmv_csr_schedule_api_f32.argtypes = [ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]
        L.loops_schedule_dump_merge_path.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]
        L.loops_schedule_dump_work_oriented.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, vp]
        L.loops_schedule_dump_group_mapped.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp]
        L.loops_work_oriented_grid.argtypes = [C.POINTER(ci)]
        L.loops_device_compute_units.argtypes = [C.POINTER(ci)]
        for name in ("loops_spmv_bcsr_f32", "loops_spmv_bcsr_f64"):
            getattr(L, name).argtypes = [ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]
        L.loops_bcsr_row_length_class.argtypes = [ci, ci, vp, C.POINTER(ci), vp]
        L.loops_spmm_csr_f32.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp]
        L.loops_spmm_csr_f64.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp]
        L.loops_spmm_merge_path_f32.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp]
        L.loops_spmm_merge_path_f64.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp]
        for sfx in ("f32", "f64"):
            getattr(L, "loops_spmv_coo_" + sfx).argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]
            getattr(L, "loops_spmv_ell_" + sfx).argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp]
            getattr(L, "loops_spmv_csc_" + sfx).argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]
            getattr(L, "loops_spmv_dia_" + sfx).argtypes = [ci, ci, ci, ci, C.c_size_t, vp, vp, vp, vp, vp]
        for sfx in ("f32", "f64"):
            getattr(L, "loops_spmv_plan_create_" + sfx).argtypes = [ci, ci, ci, vp, vp, vp, ci, ci, vp, C.POINTER(vp)]
            getattr(L, "loops_spmv_plan_refresh_values_" + sfx).argtypes = [vp, vp, vp]
            getattr(L, "loops_spmv_planned_" + sfx).argtypes = [vp, vp, vp, vp, vp, vp, vp]
        for sfx in ("f32", "f64"):
            getattr(L, "loops_panel_plan_create_" + sfx).argtypes = [ci, ci, ci, vp, vp, vp, ci, ci, vp, C.POINTER(vp)]
            getattr(L, "loops_panel_plan_create_layout_" + sfx).argtypes = [ci, ci, ci, vp, vp, vp, ci, ci, ci, vp, C.POINTER(vp)]
            getattr(L, "loops_panel_plan_refresh_values_" + sfx).argtypes = [vp, vp, vp]
            getattr(L, "loops_spmv_panel_" + sfx).argtypes = [vp, vp, vp, vp]
            getattr(L, "loops_spmv_panel_fanout_" + sfx).argtypes = [vp, vp, vp, ci, vp, vp]
        L.loops_panel_plan_destroy.argtypes = [vp]
        L.loops_panel_plan_destroy.restype = None
        L.loops_panel_plan_info.argtypes = [vp, vp]
        L.loops_panel_plan_layout.argtypes = [vp, vp]
        L.loops_panel_plan_row_blocks.argtypes = [vp, vp, vp]
        L.loops_rowband_plan_create_f32.argtypes = [ci, ci, ci, vp, vp, vp, ci, ci, vp, C.POINTER(vp)]
        L.loops_rowband_plan_create_f64.argtypes = [ci, ci, ci, vp, vp, vp, ci, ci, vp, C.POINTER(vp)]
        L.loops_rowband_plan_refresh_values_f64.argtypes = [vp, vp, vp]
        L.loops_spmv_rowband_f64.argtypes = [vp, vp, vp, vp]
        L.loops_spmv_rowband_stage_f64.argtypes = [vp, ci, vp, vp, vp]
        L.loops_rowband_plan_destroy.argtypes = [vp]
        L.loops_rowband_plan_destroy.restype = None
        L.loops_rowband_plan_info.argtypes = [vp, vp]
        L.loops_rowband_plan_arrays.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.loops_rowband_plan_set_chunks.argtypes = [vp, ci]
        L.loops_rowband_plan_tune.argtypes = [vp, ci, vp, vp]
        L.loops_rowband_plan_set_waves.argtypes = [vp, ci]
        L.loops_rowband_plan_refresh_values_f32.argtypes = [vp, vp, vp]
        L.loops_spmv_rowband_f32.argtypes = [vp, vp, vp, vp]
        L.loops_spmv_rowband_stage_f32.argtypes = [vp, ci, vp, vp, vp]
        L.loops_spmv_rowband_fanout_f32.argtypes = [vp, vp, vp, ci, vp, vp]
        L.loops_bcsr_band_plan_create_f32.argtypes = [ci, ci, ci, ci, vp, vp, vp, ci, ci, vp, C.POINTER(vp)]
        L.loops_bcsr_band_plan_destroy.argtypes = [vp]
        L.loops_bcsr_band_plan_destroy.restype = None
        L.loops_bcsr_band_plan_info.argtypes = [vp, vp]
        L.loops_bcsr_band_plan_arrays.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.loops_bcsr_band_plan_set_chunks.argtypes = [vp, ci]
        L.loops_bcsr_band_plan_tune.argtypes = [vp, ci, vp, vp]
        L.loops_bcsr_band_plan_set_shape.argtypes = [vp, ci, ci, ci]
        L.loops_bcsr_band_plan_refresh_values_f32.argtypes = [vp, vp, vp]
        L.loops_spmv_bcsr_band_f32.argtypes = [vp, vp, vp, vp]
        L.loops_spmv_bcsr_band_stage_f32.argtypes = [vp, ci, vp, vp, vp]
        L.loops_row_ranges.argtypes = [ci, vp, ci, vp]
        L.loops_comm_unique_id.argtypes = [vp]
        L.loops_comm_init.argtypes = [ci, ci, vp, C.POINTER(vp)]
        L.loops_comm_destroy.argtypes = [vp]
        L.loops_comm_error_string.argtypes = [ci]
        L.loops_comm_error_string.restype = C.c_char_p
        L.loops_allgatherv_f32.argtypes = [vp, ci, ci, vp, vp, vp]
        L.loops_allgatherv_f64.argtypes = [vp, ci, ci, vp, vp, vp]
        L.loops_panel_plan_arrays.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.loops_panel_plan_windows.argtypes = [vp, vp, vp, vp]
        L.loops_spmv_panel_stage_f32.argtypes = [vp, ci, vp, vp, vp]
        L.loops_spmv_panel_stage_f64.argtypes = [vp, ci, vp, vp, vp]
        L.loops_spmv_plan_destroy.argtypes = [vp]
        L.loops_csc_plan_destroy.argtypes = [vp]
        L.loops_csc_plan_destroy.restype = None
        L.loops_csc_plan_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), vp]
        L.loops_spmv_plan_destroy.restype = None
        L.loops_spmv_plan_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), vp]
        L.loops_autotune_merge_path_f32.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, C.POINTER(ci), vp]
        L.loops_autotune_merge_path_variants_f32.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, ci, vp, C.POINTER(ci), C.POINTER(ci), vp]
        L.loops_spmv_plan_variant.argtypes = [vp, C.POINTER(ci), C.POINTER(C.c_float)]
        L.loops_columns_look_scattered.argtypes = [ci, ci, vp, ci, vp, C.POINTER(ci)]
        _lib = L
    return _lib


_ERRORS = {-1: "LOOPS_E_BADARG", -2: "LOOPS_E_RANGE (rows + nnz must stay below 2^31)", -3: "LOOPS_E_CONFIG"}


def check(code: int, what: str) -> None:
    if code != 0:
        raise LoopsError(f"{what} failed: {_ERRORS.get(code, 'hipError_t ' + str(code))}")
