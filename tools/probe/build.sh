#!/bin/bash
# builds the tile-kernel probes into tools/probe/bin (git-ignored)
set -e
cd "$(dirname "$0")"
mkdir -p bin
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I../../include -Wno-unused-result"
for v in ${VARIANTS:-BASE}; do
  D=""; for x in ${v//,/ }; do D="$D -DY3D_PROBE_$x"; [ $x = STAGGER ] && D="$D -DY3D_STAGGER"; [ $x = SPREAD ] && D="$D -DY3D_HALO_SPREAD"; [ $x = PRIO ] && D="$D -DY3D_SETPRIO"; done
  hipcc -x hip $F $D tile_probe.cpp ../../yolov10-3d_amd/csrc/y3d_api.cpp -o bin/tile_$v &
done
wait
ls -la bin
