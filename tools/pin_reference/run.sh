#!/bin/bash
# Pins the CPU oracle of this repository against the REAL reference.  Run on a host that has what halajun/VDO_SLAM pins
# (its Dockerfile:40-63: Ubuntu 16.04, OpenCV 3.4.0 + opencv_contrib 3.4.0, libeigen3-dev, libsuitesparse-dev) - e.g. inside
# the image that Dockerfile builds:
#     tools/pin_reference/run.sh /path/to/VDO_SLAM
# It copies the checkout to a scratch directory (the g2o build writes into its source tree), builds g2o + the reference + pin_dump,
# runs every case of tests/golden/inputs/ and writes the golden outputs into tests/golden/.  Then: python -m pytest tests/test_golden.py
set -euo pipefail
REF=${1:?usage: run.sh <halajun/VDO_SLAM checkout>}
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
WORK=${WORK:-$(mktemp -d)}
cp -r "$REF" "$WORK/ref"
cmake -S "$HERE" -B "$WORK/build" -DVDO_REF="$WORK/ref"
cmake --build "$WORK/build" -j"$(nproc)"
BIN="$WORK/build/pin_dump"; IN="$ROOT/tests/golden/inputs"; OUT="$ROOT/tests/golden"
"$BIN" orb "$IN/orb_gray_640x200.u8" 640 200 "$OUT"
for f in "$IN"/pnp_case*.bin; do "$BIN" pnp "$f" "$OUT/$(basename "${f%.bin}").out"; done
for f in "$IN"/flow2_case*.bin; do "$BIN" flow2 "$f" "$OUT/$(basename "${f%.bin}").out"; done
mkdir -p "$OUT/batch_case0"; "$BIN" batch "$IN/batch_map_case0.bin" "$OUT/batch_case0"
( cd "$OUT" && { echo "reference: $(cd "$REF" && git rev-parse HEAD 2>/dev/null || echo unknown)"; pkg-config --modversion opencv 2>/dev/null | sed 's/^/opencv: /' || true; date -u +"generated: %Y-%m-%dT%H:%MZ"; } > PINNED_BY.txt )
echo "golden vectors written to $OUT; now run: python -m pytest tests/test_golden.py -q"
