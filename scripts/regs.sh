#!/bin/bash
# register / scratch / LDS use of every kernel in ptx.hip (device-only compile to assembly)
# usage: scripts/regs.sh [extra hipcc flags]
OUT=${TMPDIR:-/tmp}/ptx_regs.s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Iinclude -Ipath-tracer_amd/csrc "$@" \
  --cuda-device-only -S -o $OUT path-tracer_amd/csrc/ptx.hip 2>/dev/null
awk '$1==".name:" && $2 ~ /^_Z/ {name=$2} $1==".vgpr_count:" {v=$2; printf "%-44s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d\n", substr(name,1,44), v, a, s, p, l}
     $1==".agpr_count:" {a=$2} $1==".sgpr_count:" {s=$2} $1==".private_segment_fixed_size:" {p=$2} $1==".group_segment_fixed_size:" {l=$2}' $OUT | sed 's/_ZN12_GLOBAL__N_1[0-9]*//'
