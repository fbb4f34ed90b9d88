#!/bin/bash
# Emits the gfx950 ISA of icp.hip, cuts out the fused dense search kernel (gpurun_out/isa/fused0.s) and prints its register /
# spill / scratch line with the place of every scratch access (what the round's profiles/rNN_isa.txt keeps):
#   tools/isa.sh > profiles/r06_isa.txt
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/isa
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -S --cuda-device-only -o gpurun_out/isa/icp.s realsense-pointcloud_amd/csrc/icp.hip 2>/dev/null
cd gpurun_out/isa
n=$(grep -n "^_ZN5rsreg17k_icp_fused_denseILi0ELi1.*:" icp.s | head -1 | cut -d: -f1)
awk -v n=$n 'NR>=n' icp.s | awk '/s_endpgm/{print; exit} {print}' > fused0.s
echo "k_icp_fused_dense<0, 1> (the product's search launch), hipcc --offload-arch=gfx950 -O3 -ffp-contract=off, $(hipcc --version | grep -m1 "HIP version")"
grep -A25 "\.name: *_ZN5rsreg17k_icp_fused_denseILi0ELi1" icp.s | grep "sgpr_count\|sgpr_spill\|vgpr_count\|vgpr_spill\|private_segment_fixed_size\|group_segment_fixed_size" | sed 's/^ */  /'
echo "  lines of ISA: $(wc -l < fused0.s); buffer_load_dwordx4: $(grep -c buffer_load_dwordx4 fused0.s); v_pk_*: $(grep -c 'v_pk_' fused0.s); v_readlane/v_writelane: $(grep -c 'v_readlane\|v_writelane' fused0.s); MFMA: $(grep -c mfma fused0.s)"
echo "  scratch accesses (line: instruction; the innermost loop header in front of it):"
grep -n "scratch_" fused0.s | while IFS=: read ln rest; do
    hdr=$(awk -v n=$ln 'NR<n && /Loop Header|Inner Loop Header/ {h=$0} END{print h}' fused0.s | sed 's/^ *//')
    echo "    $ln: $(echo $rest | sed 's/^ *//')   [after: ${hdr:-no loop}]"
done
