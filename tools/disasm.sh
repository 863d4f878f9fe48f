#!/bin/bash
# Disassemble one translation unit of the HIP library for gfx950 and print per-kernel resource lines.
#   tools/disasm.sh orbx_describe.hip [extra hipcc flags]   -> tools/_build/dis/<name>.s
set -e
cd "$(dirname "$0")/.."
src=$1; shift
out=tools/_build/dis/$(basename "$src" .hip).s
mkdir -p tools/_build/dis
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Iinclude -Iorb_slam2v2-1_amd/csrc \
    --cuda-device-only -S -o "$out" "orb_slam2v2-1_amd/csrc/$src" "$@"
grep -E "^\s+\.(sgpr_count|vgpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill_count):|^\s+\.name:" "$out" | paste - - - - - - | sed 's/  */ /g'
echo "$out"
