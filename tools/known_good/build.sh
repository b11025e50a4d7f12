#!/bin/bash
# A library of an OLDER commit built from ITS sources beside HEAD's, for the A/B columns and the bisect ladder of a GPU session
# (tools/r05_round.sh, tools/ab_probe.sh): HEAD's binding loads it through PCC_LIB (compat.cpp supplies the entry points the
# C ABI has gained since).  Needs the git history (this container); the .so travels to the GPU box with the snapshot.
#   bash tools/known_good/build.sh                  # both of the following
#   bash tools/known_good/build.sh 23272ea r02      # cwi-pcl-codec_amd/libpcc_hip_r02.so: the last commit whose `-m gpu` suite ran
#                                                   # byte-green on an MI355X (178 tests, round 2)
#   bash tools/known_good/build.sh 0314de2 r04x     # cwi-pcl-codec_amd/libpcc_hip_r04x.so: round 4's HEAD = branch experiments/r04-optin-forms,
#                                                   # the forms that left the product in round 5 (PCC_FUSED_KEYS, PCC_SORT_LOCAL, PCC_SORT_BARE,
#                                                   # PCC_SORT_XCD; read from the environment by that build), never run on a GPU
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
if [ $# -eq 0 ]; then bash "$0" 23272ea r02; bash "$0" 0314de2 r04x; exit 0; fi
REV=$1; TAG=$2
OUT=cwi-pcl-codec_amd/libpcc_hip_$TAG.so
git rev-parse --verify -q "$REV^{commit}" > /dev/null || { echo "known_good: commit $REV is not in this checkout (no history?): skipped" >&2; exit 3; }
D=tools/known_good/_src_$REV
STAMP=$D/.built_$(git rev-parse --short "$REV")
if [ -f "$STAMP" ] && [ -f $OUT ] && [ $OUT -nt tools/known_good/compat.cpp ]; then exit 0; fi
rm -rf "$D"; mkdir -p "$D"
git archive "$REV" cwi-pcl-codec_amd/csrc include | tar -x -C "$D"
make -s -j8 -C "$D/cwi-pcl-codec_amd/csrc"
hipcc -std=c++17 -O2 -fPIC -c tools/known_good/compat.cpp -o "$D/compat.o"
hipcc --offload-arch=gfx950 -shared -pthread -o $OUT "$D"/cwi-pcl-codec_amd/csrc/*.o "$D/compat.o"
touch "$STAMP"
echo "built $OUT from $REV"
