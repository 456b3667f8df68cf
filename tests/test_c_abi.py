"""The C-ABI library builds, loads, and exports exactly what include/loops_amd.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

from conftest import ROOT
from loops_amd import _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "loops_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(loops_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    so = _lib.build()
    assert os.path.exists(so)
    L = _lib.load_shared(so)
    declared = _declared()
    assert declared, "no declarations parsed"
    assert sorted(_lib.SYMBOLS) == declared
    for name in declared:
        assert hasattr(L, name), name
    L.loops_version.restype = ctypes.c_char_p
    assert L.loops_version().decode().startswith("0.2.0")


def test_argument_errors_do_not_need_a_gpu():
    L = _lib.lib()
    # null pointers / bad sizes are rejected before any runtime call
    assert L.loops_spmv_csr_f32(0, -1, 1, 1, None, None, None, None, None, None) == -1
    assert L.loops_spmv_csr_f32(0, 4, 4, 4, None, None, None, None, None, None) == -1
    assert L.loops_merge_plan_create(4, 4, None, 0, None, None) == -1
    assert L.loops_merge_plan_num_tiles(None) == -1
    assert L.loops_spmv_merge_path_f32(None, 0, 1, 1, 1, None, None, None, None, None, None) == -1
    assert L.loops_spmv_bcsr_f32(4, 4, 0, 4, 1, 1, None, None, None, None, None, None) == -1
    # round 4: the variant autotuner and the plan's variant accessor reject missing outputs / handles the same way
    assert L.loops_autotune_merge_path_variants_f32(4, 4, 4, None, None, None, None, None, 1, None, None, None, None) == -1
    assert L.loops_spmv_plan_variant(None, None, None) == -1
    assert L.loops_columns_look_scattered(4, 4, None, 4, None, None) == -1
    assert L.loops_spmv_merge_path_f32(None, _lib.VARIANT_PHASED, 1, 1, 1, None, None, None, None, None, None) == -1


def test_product_never_imports_the_oracle():
    """The product path must not route through oracle/ (or any CPU fallback)."""
    for base, _, files in os.walk(os.path.join(ROOT, "loops_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hxx", ".h")):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r"(import\s+oracle|from\s+oracle|liboracle|loops_oracle|oracle/|oracle\.)", src), f
    for base, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            src = open(os.path.join(base, f)).read()
            assert "liboracle" not in src and "loops_oracle" not in src, f
    # nor do the probes / drivers outside tests/ (only tests/, smoke() and bench.py may touch oracle/)
    for top in ("scripts", "examples"):
        for base, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".cu", ".sh")) and f != "build_reference_examples.sh":
                    src = open(os.path.join(base, f)).read()
                    assert not re.search(r"(import\s+oracle|from\s+oracle|liboracle|oracle/_ref)", src), f


def test_missing_extension_fails_loudly(tmp_path):
    """No CPU fallback: with the shared library absent every product call raises LoopsError."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from loops_amd import _lib\n"
        "try:\n"
        "    _lib.lib()\n"
        "except _lib.LoopsError as e:\n"
        "    assert 'no CPU fallback' in str(e), str(e); print('LOUD')\n"
        "else:\n"
        "    print('SILENT')\n" % ROOT)
    env = dict(os.environ, LOOPS_AMD_LIB=str(tmp_path / "libloops_amd_missing.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("LOUD"), r.stdout + r.stderr


def test_argument_errors_of_the_newer_entry_points():
    L = _lib.lib()
    assert L.loops_spmm_csr_f32(0, 4, 4, 4, None, None, None, None, 8, None, None) == -1
    assert L.loops_spmm_csr_f32(0, 4, 4, 4, None, None, None, None, -1, None, None) == -1
    assert L.loops_spmm_merge_path_f32(None, 4, 4, 4, None, None, None, None, 8, None, None) == -1
    assert L.loops_rowband_plan_create_f32(4, 4, 4, None, None, None, 0, 0, None, None) == -1
    assert L.loops_rowband_plan_info(None, None) == -1
    assert L.loops_spmv_rowband_f32(None, None, None, None) == -1
    assert L.loops_spmv_coo_f32(1, 4, 4, 4, None, None, None, None, None, None) == -1
    assert L.loops_spmv_coo_f32(7, 4, 4, 0, None, None, None, None, 1, None) == -1       # unknown mode
    assert L.loops_spmv_ell_f32(1, 4, 4, 2, None, None, None, None, None) == -1
    assert L.loops_spmv_csc_f32(1, 4, 4, 4, None, None, None, None, None, None) == -1
    assert L.loops_autotune_merge_path_f32(4, 4, 4, None, None, None, None, None, 1, None, None, None) == -1


def test_measurement_code_lives_outside_the_product_library():
    """Calibration kernels / experimental instantiations are libloops_probes.so (loops_amd/csrc/loops_probes.h),
    loaded only by loops_amd/probes.py for bench.py, scripts/ and tests/perf/: the product library exports none of
    them, the product header declares none, the product modules never import the probes module, and no
    measurement macro is left in the shipped kernels."""
    L = _lib.load_shared(_lib.build())
    P = _lib.load_shared(_lib.build_probes())
    for name in ("loops_stream_copy_f32", "loops_gather_f32", "loops_address_rate_f32", "loops_row_gather_f32",
                 "loops_probe_merge_path_f32", "loops_probe_persistent_f32"):
        assert hasattr(P, name), name
        assert not hasattr(L, name), name
        assert name not in open(os.path.join(ROOT, "include", "loops_amd.h")).read()
    for mod in ("spmv.py", "partition.py", "generate.py", "__init__.py"):
        src = open(os.path.join(ROOT, "loops_amd", mod)).read()
        assert not re.search(r"import\s+probes|probes\s+as|from\s+\.probes|libloops_probes", src), mod
    for base, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "LOOPS_PROBE" not in open(os.path.join(base, f)).read(), f


def test_argument_errors_of_the_round_2_entry_points():
    L = _lib.lib()
    # format twins / DIA / ELL engine mode: null pointers and unknown modes are rejected before any runtime call
    assert L.loops_spmv_dia_f32(1, 4, 4, 2, 4, None, None, None, None, None) == -1
    assert L.loops_spmv_dia_f64(1, 4, 4, 2, 2, None, None, None, 1, None) == -1          # stride < rows
    assert L.loops_spmv_ell_f64(2, 4, 4, 2, None, None, None, None, None) == -1
    assert L.loops_spmv_coo_f64(1, 4, 4, 4, None, None, None, None, None, None) == -1
    assert L.loops_spmv_csc_f64(1, 4, 4, 4, None, None, None, None, None, None) == -1
    assert L.loops_spmv_bcsr_f64(4, 4, 0, 4, 1, 1, None, None, None, None, None, None) == -1
    assert L.loops_spmm_merge_path_f64(None, 4, 4, 4, None, None, None, None, 8, None, None) == -1
    # fused allgatherv: a plan is required, at most 7 peers, peer pointers must be given
    assert L.loops_spmv_merge_path_fanout_f32(None, 4, 4, 4, None, None, None, None, None, 0, None, None) == -1
    assert L.loops_spmv_rowband_fanout_f32(None, None, None, 0, None, None) == -1
