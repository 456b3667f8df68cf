#!/bin/bash
# Acceptance check of the drop-in boundary: compile the REFERENCE's own example drivers, read in
# place from /root/reference/examples (never copied), UNCHANGED, against include/loops of this
# repository.  Only possible where the reference tree is mounted (the dev container); the
# binaries land in build/examples/ and travel to the GPU box with the snapshot, where
# tests/test_examples_gpu.py runs them on tests/golden/chesapeake.mtx with --validate.
set -u
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/examples
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -std=c++17 -O3 -x hip -DLOOPS_TARGET_GFX=0x950 -Wno-unused-result -I$ROOT/include -I$ROOT/third_party/cxxopts"
fail=0
build() {  # src out extra-flags
  if /opt/rocm/bin/hipcc $FLAGS $3 "$1" -o "$2" > "$2.log" 2>&1; then echo "ok   $(basename $2)"; else echo "FAIL $(basename $2) (see $2.log)"; fail=1; fi
}
JOBS=${JOBS:-$(nproc)}
n=0
for src in $REF/examples/spmv/*.cu; do
  name=$(basename "$src" .cu)
  build "$src" "$OUT/loops.spmv.$name.f32" "-DLOOPS_VALUE_T=float -I$REF/examples/spmv" &
  build "$src" "$OUT/loops.spmv.$name.f64" "-DLOOPS_VALUE_T=double -I$REF/examples/spmv" &
  n=$((n+2)); if [ $n -ge $JOBS ]; then wait; n=0; fi
done
wait
build "$REF/examples/spmm/thread_mapped.cu" "$OUT/loops.spmm.thread_mapped" "-I$REF/examples/spmm" &
build "$REF/examples/saxpy/saxpy.cu" "$OUT/loops.saxpy" "" &
build "$REF/examples/range/range.cu" "$OUT/loops.range" "" &
# this repository's own example drivers for the paths the reference does not have
build "$ROOT/examples/spmm/merge_path_flat.cu" "$OUT/loops.spmm.merge_path_flat" "" &
build "$ROOT/examples/spmv/rowband.cu" "$OUT/loops.spmv.rowband" "" &
build "$ROOT/examples/spmv/spmv_plan.cu" "$OUT/loops.spmv.spmv_plan" "" &
wait
# which headers these binaries were built from: tests/test_examples_gpu.py refuses stale binaries
if [ $fail -eq 0 ]; then python3 "$ROOT/scripts/headers_digest.py" > "$OUT/HEADERS.sha256"; fi
exit $fail
