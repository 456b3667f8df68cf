"""The REFERENCE's own unit tests (unittests/test_*.cu, SURVEY 8b: they are callers of include/loops like the examples), compiled
UNCHANGED and in place against this repository's headers by scripts/build_reference_unittests.sh -- Catch2 and
<cuda_runtime.h> replaced by the tests-only stand-ins of tests/stubs/ -- and run here, one binary per test file: layout
contracts of the six layout views + the flat partitioner, container constructors / conversions / cross-space copies, format
round trips, the Matrix-Market loader's accept / reject cases, SpMV per format over the reference's own matrix battery, the
schedule-coverage visit counts, the rigorous validator, math and range utilities.  What the restated tests/cpp/*.cpp cannot
catch: drift of the API the reference's callers actually use."""
import glob
import os
import re
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "build", "unittests")
NAMES = ["test_container_bcsr", "test_container_coo", "test_container_csc", "test_container_csr", "test_container_dia",
         "test_container_ell", "test_format_round_trip", "test_layout_bcsr", "test_layout_coo", "test_layout_csc",
         "test_layout_csr", "test_layout_dia", "test_layout_ell", "test_layout_flat_partitioner", "test_market_loader",
         "test_rigorous_validator", "test_schedule_coverage", "test_spmv_bcsr", "test_spmv_coo", "test_spmv_csc", "test_spmv_csr",
         "test_spmv_dia", "test_spmv_ell", "test_spmv_partitioned", "test_util_math", "test_util_range"]


def test_unit_test_binaries_were_built_from_these_headers():
    """Prebuilt in the dev container (the sources live in /root/reference): they must come from the CURRENT include/ tree."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from headers_digest import digest
    manifest = os.path.join(BIN, "HEADERS.sha256")
    assert os.path.exists(manifest), "build/unittests/HEADERS.sha256 missing: run scripts/build_reference_unittests.sh (needs /root/reference)"
    assert open(manifest).read().strip() == digest(), "unit-test binaries are stale: include/ changed since scripts/build_reference_unittests.sh ran"
    built = sorted(os.path.basename(p) for p in glob.glob(os.path.join(BIN, "test_*")) if os.access(p, os.X_OK) and "." not in os.path.basename(p))
    assert built == NAMES, built


@pytest.mark.parametrize("name", NAMES)
def test_reference_unit_test_passes(name):
    path = os.path.join(BIN, name)
    assert os.path.exists(path), f"{name} not built: run scripts/build_reference_unittests.sh in the dev container"
    r = subprocess.run([path], capture_output=True, text=True, timeout=600, cwd=ROOT)
    tail = r.stdout[-3000:] + r.stderr[-1500:]
    m = re.search(r"test cases: (\d+) \| (\d+) failed \| assertions: (\d+)", r.stdout)
    assert m, tail
    assert r.returncode == 0 and int(m.group(2)) == 0, tail
    assert int(m.group(1)) >= 1 and int(m.group(3)) >= 1, tail
