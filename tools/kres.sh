#!/bin/bash
# usage: kres.sh file.hip [flags]  -> prints per-kernel vgpr/sgpr/lds/scratch
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" --cuda-device-only -S -o /tmp/kres.s $src 2>&1 | grep -v warning | head -5
grep -E "^\s+\.(name|vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|agpr_count):" /tmp/kres.s | sed 's/ \+/ /g' | paste - - - - - - | cut -c1-300
