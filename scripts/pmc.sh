#!/bin/bash
# PMC passes for the bench (one counter group per run; never combined with tracing).
# usage: [PASSES="sq1 sq2 ..."] scripts/pmc.sh <outdir> <bench args...>
set -u
export TMPDIR=/tmp
OUT=$PWD/$1; shift
mkdir -p $OUT
BENCH_ARGS=("$@")
PASSES=${PASSES:-"sq1 sq2 sq3 fetch write"}
declare -A G
G[sq1]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
G[sq2]="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32"
G[sq3]="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64"
G[sq4]="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH"
G[fetch]="FETCH_SIZE GRBM_GUI_ACTIVE"
G[write]="WRITE_SIZE TCC_EA0_ATOMIC_sum"
G[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"
for name in $PASSES; do
  timeout -s KILL 240 rocprofv3 --pmc ${G[$name]} --output-format csv -d $OUT/$name -- python3 bench.py "${BENCH_ARGS[@]}" > $OUT/$name.json 2> $OUT/$name.err || { echo "pass $name failed"; tail -3 $OUT/$name.err; exit 1; }
  echo "pass $name ok"
done
