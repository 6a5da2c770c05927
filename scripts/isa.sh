#!/bin/bash
# Dump the gfx950 ISA of one k_fused_temporal instantiation and print its instruction mix.
#   scripts/isa.sh NAME "float, 0, 2, 2, 0, 2, 8, 5"
# -> aggfly_amd/csrc/_build/asm/NAME.s
set -e
cd "$(dirname "$0")/../aggfly_amd/csrc"
mkdir -p _build/asm
cat > _build/asm/$1.hip <<EOT
#include "afhip_kernels.h"
namespace afhip { template __global__ void k_fused_temporal<$2>(const FusedArgs); }
EOT
/opt/rocm/bin/hipcc -I. -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math \
    -S --offload-device-only _build/asm/$1.hip -o _build/asm/$1.s
grep -E "^\s+(v|s|ds|global|buffer)_[a-z0-9_]+" -o _build/asm/$1.s | sort | uniq -c | sort -rn | head -${3:-45}
grep -E "\.(vgpr_count|sgpr_count|private_segment_fixed_size):" _build/asm/$1.s
