#!/bin/bash
# registers, spills, scratch, LDS and occupancy of every kernel (CPU only: hipcc cross-compiles gfx950); usage: tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
cd "$(dirname "$0")/.."
echo "# hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Rpass-analysis=kernel-resource-usage (osmt_kernels.hip, osmt_labels.hip, osmt_pngenc.hip)"
for f in osmt_kernels.hip osmt_labels.hip osmt_pngenc.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -c -o /dev/null -Rpass-analysis=kernel-resource-usage osm_renderer_amd/csrc/$f 2>&1 \
    | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill|LDS Size" | sed -E 's/^.*remark: //; s/ \[-Rpass-analysis=kernel-resource-usage\]//'
done
