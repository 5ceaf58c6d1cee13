#!/bin/bash
# builds the tile-kernel probes into tools/probe/bin (git-ignored)
set -e
cd "$(dirname "$0")"
mkdir -p bin
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../include -Wno-unused-result"
for v in BASE TRACE NOMFMA; do
  D=""; for x in ${v//,/ }; do D="$D -DY3D_PROBE_$x"; done
  hipcc -x hip $F $D tile_probe.cpp ../../yolov10-3d_amd/csrc/y3d_api.cpp -o bin/tile_$v &
done
wait
ls -la bin
