#!/bin/bash
# ISA of one translation unit of the library: registers, spills and instruction mix per kernel.
#   tools/isa_stats.sh qc_heavy            (writes ldpc_amd/csrc/build/isa/<unit>*.s)
set -e
unit=${1:-qc_heavy}
cd "$(dirname "$0")/../ldpc_amd/csrc"
mkdir -p build/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-pass-failed \
    -c -o build/isa/$unit.o $unit.hip --save-temps=obj
S=build/isa/$unit-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.(name|vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):" $S
