#!/bin/bash
# Evidence for the drop-in boundary (SURVEY 8b lists unittests/*.cu among the callers): compile the REFERENCE's own unit-test
# files, read in place from /root/reference/unittests (never copied), UNCHANGED, against include/loops of this repository.
# They need Catch2 and <cuda_runtime.h>, neither of which this image has: tests/stubs/ holds tests-only stand-ins for exactly
# the macros / names they use -- on the include path of THIS build only.  Binaries land in build/unittests/ and travel to
# the GPU box, where tests/test_reference_unittests_gpu.py runs them.  Only possible where the reference tree is mounted.
set -u
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/unittests
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -std=c++17 -O2 -x hip -DLOOPS_TARGET_GFX=0x950 -Wno-unused-result -I$ROOT/include -I$ROOT/tests/stubs -I$REF/unittests"
fail=0
JOBS=${JOBS:-$(nproc)}
n=0
for src in $REF/unittests/test_*.cu; do
  name=$(basename "$src" .cu)
  ( if /opt/rocm/bin/hipcc $FLAGS "$src" -o "$OUT/$name" > "$OUT/$name.log" 2>&1; then echo "ok   $name"; else echo "FAIL $name (see $OUT/$name.log)"; fi ) &
  n=$((n+1)); if [ $n -ge $JOBS ]; then wait; n=0; fi
done
wait
for src in $REF/unittests/test_*.cu; do [ -x "$OUT/$(basename "$src" .cu)" ] || fail=1; done
if [ $fail -eq 0 ]; then python3 "$ROOT/scripts/headers_digest.py" > "$OUT/HEADERS.sha256"; fi
exit $fail
