"""Drop-in acceptance: the REFERENCE's own example drivers (examples/spmv/*.cu etc.), compiled
UNCHANGED against include/loops of this repository by scripts/build_reference_examples.sh (dev
container only -- the sources are read in place from /root/reference, never copied), run on the
bundled chesapeake matrix with --validate.  BASELINE config C1: `Errors: 0`, y bit-exact."""
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "build", "examples")
MTX = os.path.join(GOLDEN, "chesapeake.mtx")

SPMV = ["merge_path", "thread_mapped", "work_oriented", "group_mapped", "original", "flat_partitioned",
        "bcsr_thread_mapped", "coo_thread_mapped", "csc_thread_mapped", "dia_thread_mapped", "ell_thread_mapped",
        "ell_merge_path", "custom_layout"]


def _run(exe, *args):
    # On a GPU box a missing or stale binary is a FAILURE, not a skip: a snapshot without the prebuilt artefacts
    # must not go green with the drop-in boundary untested.
    path = os.path.join(BIN, exe)
    assert os.path.exists(path), f"{exe} not built: run scripts/build_reference_examples.sh in the dev container (needs /root/reference)"
    return subprocess.run([path, *args], capture_output=True, text=True, timeout=300)


def test_example_binaries_were_built_from_these_headers():
    """The binaries travel prebuilt (their sources live in /root/reference): they must come from the CURRENT
    include/ tree, otherwise this suite would exercise stale kernels (scripts/build_reference_examples.sh records
    the digest of the headers it compiled against)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from headers_digest import digest
    manifest = os.path.join(BIN, "HEADERS.sha256")
    assert os.path.exists(manifest), "build/examples/HEADERS.sha256 missing: rebuild with scripts/build_reference_examples.sh"
    assert open(manifest).read().strip() == digest(), "example binaries are stale: include/ changed since scripts/build_reference_examples.sh ran"


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("name", SPMV)
def test_reference_spmv_example_validates(name, precision):
    r = _run(f"loops.spmv.{name}.{precision}", "-m", MTX, "--validate", "--rigorous")
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    # CSV line: <kernel>,<dataset>,<rows>,<cols>,<nnz>[,extra],<ms>
    assert re.search(r"chesapeake,39,39,340", out), out
    assert re.search(r"Errors:\s+0\b", out), out
    assert re.search(r"Dimensions:\s+39 x 39 \(340\)", out), out
    assert "Verdict:\tNOT_A_BUG" in out, out
    assert re.search(r"GPUOverruns:\s+0\b", out) and re.search(r"MaxAbsError:\s+0\b", out), out


def test_reference_cli_contract():
    r = _run("loops.spmv.merge_path.f32", "--help")
    assert r.returncode == 0 and "--market" in r.stdout and "--validate" in r.stdout
    r = _run("loops.spmv.merge_path.f32")  # no matrix: prints help, exits 0 (helpers.hxx:64-67)
    assert r.returncode == 0 and "--market" in r.stdout


def test_reference_spmm_saxpy_range_examples_run():
    r = _run("loops.spmm.thread_mapped", "-m", MTX)
    assert r.returncode == 0, r.stderr[-2000:]
    r = _run("loops.saxpy")
    assert r.returncode == 0, r.stderr[-2000:]
    r = _run("loops.range")
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.parametrize("exe,arg", [("loops.spmm.merge_path_flat", "10"), ("loops.spmm.merge_path_flat", "64"),
                                     ("loops.spmv.rowband", "0"), ("loops.spmv.rowband", "64"),
                                     ("loops.spmv.spmv_plan", "5")])
def test_own_examples_for_the_new_paths(exe, arg):
    """This repository's drivers for the paths the reference does not have (tuned SpMM, row-band
    SpMV, the SpMV plan): each checks itself against the reference-shaped / plain-CSR kernel and prints Errors: 0."""
    r = _run(exe, MTX, arg)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert re.search(r"Errors:\s+0\b", r.stdout), r.stdout
