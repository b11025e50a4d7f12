#!/bin/bash
# Registers, scratch, LDS and the occupancy bound of every kernel, from the compiler (no GPU needed):
#   bash tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
cd "$(dirname "$0")/../cwi-pcl-codec_amd/csrc"
for f in pcc_kernels pcc_delta pcc_quality pcc_decode pcc_rc_device; do
  hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/_kr.o 2>&1 |
    grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed 's/.*remark: *//; s/ *\[-Rpass[^]]*\]//' |
    awk '/Function Name/ {if (line) print line; line=$0; next} {line=line " | " $0} END {print line}' |
    sed 's/Function Name: //' | while read -r l; do n=$(echo "$l" | cut -d' ' -f1 | c++filt | sed 's/(.*//'); echo "$f.hip  $n | $(echo "$l" | cut -d'|' -f2-)"; done
done
rm -f /tmp/_kr.o
