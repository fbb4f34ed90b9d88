#!/bin/bash
# Emits the gfx950 ISA of icp.hip and cuts out the fused dense search kernel (gpurun_out/isa/fused0.s).
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -S --cuda-device-only -o gpurun_out/isa/icp.s realsense-pointcloud_amd/csrc/icp.hip 2>/dev/null
cd gpurun_out/isa
n=$(grep -n "^_ZN5rsreg17k_icp_fused_denseILi0ELi1.*:" icp.s | head -1 | cut -d: -f1)
awk -v n=$n 'NR>=n' icp.s | awk '/s_endpgm/{print; exit} {print}' > fused0.s
grep -A12 "\.name: *_ZN5rsreg17k_icp_fused_denseILi0ELi1" icp.s | grep "sgpr_count\|sgpr_spill\|vgpr_count\|vgpr_spill"
wc -l fused0.s
