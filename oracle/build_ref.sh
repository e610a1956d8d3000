#!/usr/bin/env bash
# Build oracle/_ref/libzignal_ref.so from the reference's Boost-free hand-written filters,
# compiled from the reference sources where they lie (default /root/reference).
# TEST INFRASTRUCTURE.  Outputs go only to oracle/_ref/ (git-ignored, travels with gpurun).
# The reference's own flags are `-O3 --std=c++1y`, no -march (CMakeLists.txt:18); x86-64
# baseline has no FMA, and -ffp-contract=off is passed to make that explicit.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
REF="${ZIGNAL_REFERENCE:-/root/reference}"
out="$here/_ref"
if [ ! -f "$REF/test/benchmark.cpp" ]; then
   echo "build_ref: reference not present at $REF -- keeping prebuilt $out (if any)" >&2
   exit 0
fi
B="$REF/test/benchmark.cpp"
M="$REF/experimental_steps/multi_wires_feedback.cpp"
# guard against a reference that differs from the surveyed revision
sed -n '18p'  "$B" | grep -q 'const float b0'            || { echo "build_ref: unexpected $B:18"  >&2; exit 1; }
sed -n '35p'  "$B" | grep -q 'make_custom'               || { echo "build_ref: unexpected $B:35"  >&2; exit 1; }
sed -n '49p'  "$B" | grep -q 'make_custom2'              || { echo "build_ref: unexpected $B:49"  >&2; exit 1; }
sed -n '65p'  "$B" | grep -q 'make_custom'               || { echo "build_ref: unexpected $B:65"  >&2; exit 1; }
sed -n '90p'  "$B" | grep -q 'make_custom'               || { echo "build_ref: unexpected $B:90"  >&2; exit 1; }
sed -n '116p' "$B" | grep -q 'make_custom'               || { echo "build_ref: unexpected $B:116" >&2; exit 1; }
sed -n '768p' "$M" | grep -q 'x_wire'                    || { echo "build_ref: unexpected $M:768" >&2; exit 1; }
mkdir -p "$out"
tu="$out/ref_tu.cpp"
{
   echo '#include <tuple>'
   echo '#include <cstddef>'
   echo 'namespace biquad {'
   sed -n '18,23p' "$B"
   echo 'namespace direct_form_1 {';            sed -n '35,55p'   "$B"; echo '}'
   echo 'namespace direct_form_2 {';            sed -n '65,76p'   "$B"; echo '}'
   echo 'namespace direct_form_1_transposed {'; sed -n '90,105p'  "$B"; echo '}'
   echo 'namespace direct_form_2_transposed {'; sed -n '116,126p' "$B"; echo '}'
   echo '}'
   sed -n '768,775p' "$M"
   cat "$here/ref_harness.inc"
} > "$tu"
g++ -std=c++14 -O3 -ffp-contract=off -fPIC -shared -o "$out/libzignal_ref.so" "$tu"
rm -f "$tu"     # keep only the binary: no reference text outside /root/reference
echo "build_ref: built $out/libzignal_ref.so"
