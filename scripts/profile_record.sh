#!/bin/bash
# The measurement record kept under profiles/ (run on the GPU box from the repo root):
#   1. bench.py with its default arguments            -> $OUT/bench.json
#   2. rocprofv3 --kernel-trace --stats of the bench  -> $OUT/kernel_stats.csv (+ bench line); the same on example.sdf -> kernel_stats_example.csv
#   3. counter passes, one group per run              -> $OUT/pmc/, summary + HBM traffic per launch
# usage: scripts/profile_record.sh gpurun_out/rec
set -u
export TMPDIR=/tmp
OUT=$PWD/$1
mkdir -p $OUT
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "bench ok"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --cpu-seconds 0 --no-also --no-census > $OUT/bench_traced.json 2> $OUT/trace.err || { echo "trace failed"; tail -5 $OUT/trace.err; exit 1; }
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
echo "trace ok"
# the same trace for BASELINE configs[1] (example.sdf), the scene the first record was taken on
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_example -- python3 bench.py --scene example.sdf --cpu-seconds 0 --no-also --no-census > $OUT/bench_traced_example.json 2> $OUT/trace_example.err || { echo "trace (example.sdf) failed"; tail -5 $OUT/trace_example.err; exit 1; }
cp $(ls $OUT/trace_example/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_example.csv
echo "trace example ok"
PASSES="sq1 sq2 sq4 tcp fetch write" scripts/pmc.sh $1/pmc --steps 1 --warmup 0 --cpu-seconds 0 --no-profile --no-also --no-census || exit 1
python3 scripts/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt
QG=$(python3 -c "import json,sys; print(json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])['config']['queue_gib'])")
python3 scripts/pmc_summary.py --traffic $OUT/pmc cornell_buster.sdf 1920 1080 32 5 $QG > $OUT/hbm_traffic.json
python3 scripts/record_table.py $OUT > $OUT/record_table.md
echo "pmc ok"
