#!/usr/bin/env bash
# build_ref.sh -- compile the LITERAL reference hot-path functions where they
# lie under /root/reference into oracle/_ref/libntt_ref.so (git-ignored).
#
# TEST INFRASTRUCTURE ONLY.  Runs only where the reference tree exists (the
# authoring container); the GPU box uses the prebuilt .so that travels with the
# snapshot, or skips the tests that need it.  No reference text is written
# inside the repository: the extracted line ranges live in a mktemp directory
# that is removed on exit.
set -euo pipefail
REF="${NTT_REFERENCE_DIR:-/root/reference}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -f "$REF/src/test.cpp" ] || [ ! -f "$REF/src/aie_core.cc" ]; then
    echo "build_ref: reference tree not present at $REF -- skipped" >&2
    exit 0
fi
TMP="$(mktemp -d /tmp/ntt_ref_build.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
sed -n '15,60p'   "$REF/src/test.cpp"    > "$TMP/test_15_60.inc"
sed -n '69,71p'   "$REF/src/test.cpp"    > "$TMP/test_69_71.inc"
sed -n '212,219p' "$REF/src/test.cpp"    > "$TMP/test_212_219.inc"
sed -n '11,39p'   "$REF/src/aie_core.cc" > "$TMP/core_11_39.inc"
mkdir -p "$OUT"
g++ -O2 -std=c++17 -shared -fPIC \
    -DREF_TEST_15_60="\"$TMP/test_15_60.inc\"" \
    -DREF_TEST_69_71="\"$TMP/test_69_71.inc\"" \
    -DREF_TEST_212_219="\"$TMP/test_212_219.inc\"" \
    -DREF_CORE_11_39="\"$TMP/core_11_39.inc\"" \
    "$HERE/ref_shim.cpp" -o "$OUT/libntt_ref.so"
echo "build_ref: wrote $OUT/libntt_ref.so"
