#!/bin/bash
# ONE COMMAND for a maintainer who has OpenCV:   tools/opencv_pin/run.sh [/path/to/ORB_SLAM3_modified] [--orbx]
#   1. builds the oracle + tools/validate_opencv.cpp against the OpenCV cmake finds (OpenCV_DIR / CMAKE_PREFIX_PATH as usual), with the flags
#      ORB_SLAM3 is built with; with the reference's path also its own src/ORBextractor.cc; with --orbx also liborbx.so (needs an MI355X)
#   2. writes the validation set (natural crops + synthetic frames; python3 + numpy — without them the built-in synthetic set is used and the
#      profile table is skipped, it belongs to the full set)
#   3. runs it: every primitive bit for bit against the oracle under the variant the adapter's calibration detects (first differing byte per
#      primitive), and which NAMED profile (orbx_set_cpu_profile) this OpenCV is, from the digest table of expected_digests.txt
# OPENCV_PIN_CMAKE_ARGS: further cmake arguments (-DOpenCV_DIR=..., -DWITH_EXTRAS=OFF for an OpenCV without calib3d).
# Exit code 0 = this OpenCV is one liborbx reproduces; the last lines say which.  Nothing is installed, everything lives in a temp directory.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"; ROOT="$(cd "$HERE/../.." && pwd)"
REF=""; ORBX=""
for a in "$@"; do if [ "$a" = "--orbx" ]; then ORBX=1; else REF="$a"; fi; done
B="$(mktemp -d)"; trap 'rm -rf "$B"' EXIT
cmake -S "$HERE" -B "$B" ${REF:+-DORB_SLAM3_ROOT="$REF"} ${ORBX:+-DWITH_ORBX=ON} $OPENCV_PIN_CMAKE_ARGS > "$B/cmake.log" 2>&1 || { cat "$B/cmake.log"; exit 2; }
cmake --build "$B" -j 4 > "$B/build.log" 2>&1 || { tail -40 "$B/build.log"; exit 2; }
ARGS=()
if python3 -c "import numpy" 2>/dev/null && (cd "$ROOT" && python3 tools/make_validate_set.py "$B/validate_set.bin" --synthetic 1 > /dev/null); then
  ARGS+=(--set "$B/validate_set.bin" --expect "$HERE/expected_digests.txt")
else
  echo "[opencv_pin] python3 + numpy not found: built-in synthetic images only, no profile table"
fi
[ -n "$ORBX" ] && ARGS+=(--orbx)
LD_LIBRARY_PATH="$B:$ROOT/orb_slam3_modified_amd:$LD_LIBRARY_PATH" "$B/opencv_pin" "${ARGS[@]}"
