#!/bin/bash
# The last library that was byte-green on an MI355X (178 `-m gpu` tests, round 2, commit 23272ea), built from ITS sources
# beside HEAD's: cwi-pcl-codec_amd/libpcc_hip_r02.so.  For the first GPU session after rounds without one: if HEAD is red on
# the chip, tools/ab_probe.sh and tools/r04_round.sh time and digest-check this library, HEAD with its optional forms off, and
# HEAD, in one session.  Needs the git history (this container); the .so travels to the GPU box with the snapshot.
#   bash tools/known_good/build.sh [commit]
set -e
REV=${1:-23272ea}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
git rev-parse --verify -q "$REV^{commit}" > /dev/null || { echo "known_good: commit $REV is not in this checkout (no history?): skipped" >&2; exit 3; }
D=tools/known_good/_src_$REV
STAMP=$D/.built_$(git rev-parse --short "$REV")
if [ -f "$STAMP" ] && [ -f cwi-pcl-codec_amd/libpcc_hip_r02.so ] && [ cwi-pcl-codec_amd/libpcc_hip_r02.so -nt tools/known_good/compat.cpp ]; then exit 0; fi
rm -rf "$D"; mkdir -p "$D"
git archive "$REV" cwi-pcl-codec_amd/csrc include | tar -x -C "$D"
make -s -C "$D/cwi-pcl-codec_amd/csrc"
hipcc -std=c++17 -O2 -fPIC -c tools/known_good/compat.cpp -o "$D/compat.o"
hipcc --offload-arch=gfx950 -shared -pthread -o cwi-pcl-codec_amd/libpcc_hip_r02.so "$D"/cwi-pcl-codec_amd/csrc/*.o "$D/compat.o"
touch "$STAMP"
echo "built cwi-pcl-codec_amd/libpcc_hip_r02.so from $REV"
