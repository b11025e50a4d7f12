#!/bin/bash
# Host stages under ASan + UBSan (g++, no GPU needed).
set -e
cd "$(dirname "$0")"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -march=x86-64-v3 \
    -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ host_fuzz.cpp ../../cwi-pcl-codec_amd/csrc/pcc_host_codec.cpp ../../cwi-pcl-codec_amd/csrc/pcc_delta_host.cpp \
    -o /tmp/pcc_host_fuzz
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 /tmp/pcc_host_fuzz
# the host decoder hands the walk over the tree to a second thread: the same program under ThreadSanitizer
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -march=x86-64-v3 -pthread \
    -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ host_fuzz.cpp ../../cwi-pcl-codec_amd/csrc/pcc_host_codec.cpp ../../cwi-pcl-codec_amd/csrc/pcc_delta_host.cpp \
    -o /tmp/pcc_host_fuzz_tsan
TSAN_OPTIONS=halt_on_error=1 /tmp/pcc_host_fuzz_tsan
