#!/bin/bash
# pin_with_pcl.sh -- the day a PCL installation is at hand: build the REAL reference codec into oracle/_ref/ with one
# plain compiler line (not the reference's CMake), run the golden clouds through it and diff every byte against the
# CPU oracle.  Until this has run and passed, everything PCL-inherited in the oracle stays "parity unpinned"
# (DESIGN.md (c)); the result file it writes (oracle/_ref/pin_report.json) is what lifts that label.
#
#   PCL_ROOT=/usr            # prefix of a PCL 1.8.1 .. 1.10.x installation (include/pcl-1.x, lib/libpcl_*.so)
#   EIGEN_INC=/usr/include/eigen3   BOOST_INC=/usr/include   JPEG_LIB="-lturbojpeg -ljpeg"   (defaults below)
#   REF=/root/reference
#   bash oracle/pin_with_pcl.sh
#   bash oracle/pin_with_pcl.sh --dry-run    # no PCL needed: checks that every file the build line names is where it is
#                                            # expected, probes the usual places for REAL PCL / Eigen / Boost headers and,
#                                            # only if all of them are found, type-checks the driver against them
#                                            # (-fsyntax-only); never writes or uses stand-in headers
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${REF:-/root/reference}
if [ "${1:-}" = "--dry-run" ]; then
  rc=0
  for f in "$HERE/ref_codec_driver.cpp" "$HERE/pin_check.py" "$REF/jpeg_io/src/jpeg_io.cpp" \
           "$REF/cloud_codec_v2/include/pcl/cloud_codec_v2/point_cloud_codec_v2.h" \
           "$REF/cloud_codec_v2/include/pcl/cloud_codec_v2/impl/point_cloud_codec_v2_impl.hpp" \
           "$REF/jpeg_io/include/pcl/io/jpeg_io.h"; do
    if [ -f "$f" ]; then echo "found    $f"; else echo "MISSING  $f"; rc=1; fi
  done
  PCL_INC=""; EIG=""; BST=""
  for root in "${PCL_ROOT:-}" /usr /usr/local /opt/pcl /opt/conda; do
    [ -n "$root" ] || continue
    c=$(ls -d "$root"/include/pcl-1.* 2>/dev/null | sort -V | tail -1 || true)
    [ -n "$c" ] && [ -f "$c/pcl/compression/octree_pointcloud_compression.h" ] && { PCL_INC=$c; break; }
  done
  for d in "${EIGEN_INC:-}" /usr/include/eigen3 /usr/local/include/eigen3 /opt/conda/include/eigen3; do
    [ -n "$d" ] && [ -f "$d/Eigen/Core" ] && { EIG=$d; break; }
  done
  for d in "${BOOST_INC:-}" /usr/include /usr/local/include /opt/conda/include; do
    [ -n "$d" ] && [ -f "$d/boost/shared_ptr.hpp" ] && { BST=$d; break; }
  done
  echo "PCL headers:   ${PCL_INC:-not installed}"
  echo "Eigen headers: ${EIG:-not installed}"
  echo "Boost headers: ${BST:-not installed}"
  if [ -n "$PCL_INC" ] && [ -n "$EIG" ] && [ -n "$BST" ] && [ $rc -eq 0 ]; then
    "${CXX:-g++}" -std=c++14 -fsyntax-only -w -I"$PCL_INC" -I"$EIG" -I"$BST" -I"$REF/cloud_codec_v2/include" -I"$REF/jpeg_io/include" \
      "$HERE/ref_codec_driver.cpp" && echo "type check of ref_codec_driver.cpp against the real headers: ok" || rc=1
  else
    echo "type check skipped: it needs the real PCL, Eigen and Boost headers (no stand-ins are written); parity stays UNPINNED"
  fi
  exit $rc
fi
PCL_ROOT=${PCL_ROOT:?set PCL_ROOT to the prefix of a PCL 1.8.1-1.10 installation}
EIGEN_INC=${EIGEN_INC:-/usr/include/eigen3}
BOOST_INC=${BOOST_INC:-/usr/include}
JPEG_LIB=${JPEG_LIB:--ljpeg}
CXX=${CXX:-g++}
PCL_INC=$(ls -d "$PCL_ROOT"/include/pcl-1.* 2>/dev/null | sort -V | tail -1)
[ -n "$PCL_INC" ] || { echo "no include/pcl-1.x under $PCL_ROOT" >&2; exit 2; }
[ -f "$REF/cloud_codec_v2/include/pcl/cloud_codec_v2/point_cloud_codec_v2.h" ] || { echo "reference checkout not found at $REF" >&2; exit 2; }
mkdir -p "$HERE/_ref"
# the reference's own build uses -g -O0 (CMakeLists.txt:85-87); -O1 keeps the arithmetic (no fast-math, no contraction)
"$CXX" -std=c++14 -O1 -fPIC -shared -ffp-contract=off -fno-fast-math -w \
  -I"$PCL_INC" -I"$EIGEN_INC" -I"$BOOST_INC" \
  -I"$REF/cloud_codec_v2/include" -I"$REF/jpeg_io/include" \
  "$HERE/ref_codec_driver.cpp" "$REF/jpeg_io/src/jpeg_io.cpp" \
  -L"$PCL_ROOT/lib" -Wl,-rpath,"$PCL_ROOT/lib" \
  -lpcl_common -lpcl_octree -lpcl_io -lpcl_kdtree -lpcl_search -lpcl_filters -lpcl_registration -lpcl_features -lpcl_sample_consensus \
  -lboost_system -lboost_filesystem -lboost_thread $JPEG_LIB -fopenmp \
  -o "$HERE/_ref/libcodec_ref.so"
echo "built oracle/_ref/libcodec_ref.so against $PCL_INC"
python3 "$HERE/pin_check.py" "$HERE/_ref/libcodec_ref.so"
