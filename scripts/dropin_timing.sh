#!/bin/bash
# Throughput of the drop-in path itself: the reference's demo main against the GPU Renderer
# (pt_example: load_scene, Renderer::render with the library's default queue memory, ASCII PPM
# written), on BASELINE configs[1] and configs[2], each twice (the first run of a process pays
# the one-time start of the HIP runtime).  usage: scripts/dropin_timing.sh <outdir>
set -u
OUT=$1
mkdir -p $OUT
SDF=fixtures/sdf
for scene in example cornell_buster; do
  for rep in 1 2; do
    timeout -k 10 300 path-tracer_amd/pt_example $SDF/$scene.sdf /tmp/$scene.ppm 1920 1080 32 5 $SDF/ > $OUT/dropin_$scene.$rep.log 2>&1 || { echo "pt_example $scene failed"; cat $OUT/dropin_$scene.$rep.log; exit 1; }
    echo "== $scene run $rep"; cat $OUT/dropin_$scene.$rep.log
  done
done
# the same program under the kernel tracer: which kernels the drop-in path runs (the program itself after --, no wrapper)
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dropin_trace -- path-tracer_amd/pt_example $SDF/cornell_buster.sdf /tmp/traced.ppm 1920 1080 32 5 $SDF/ > $OUT/dropin_traced.log 2>&1 || { echo "traced pt_example failed"; tail -5 $OUT/dropin_traced.log; exit 1; }
cp $(ls $OUT/dropin_trace/*/*kernel_stats.csv | head -1) $OUT/dropin_kernel_stats.csv
echo "== kernels of the drop-in run"; head -7 $OUT/dropin_kernel_stats.csv | cut -c1-150
python3 - <<'PY'
import importlib, sys, time
sys.path.insert(0, '.')
import torch
pt = importlib.import_module('path-tracer_amd')
for name in ('cornell', 'cornell_buster'):
    hs = pt.HostScene(f'{pt.SDF_DIR}/{name}.sdf')
    t = time.perf_counter(); ds = pt.DeviceScene(hs, 0, 0); t_create = time.perf_counter() - t
    for (w, h, aa) in ((96, 64, 2), (320, 200, 2), (640, 400, 4)):
        row = []
        for pipeline in ('staged', 'fused'):
            best = 1e9
            for _ in range(5):
                t = time.perf_counter(); ds.render_image(w, h, aa, 5, pipeline=pipeline); best = min(best, time.perf_counter() - t)
            row.append('%s %.2f ms' % (pipeline, best * 1e3))
        print(name, 'scene create %.3f s;' % t_create, f'{w}x{h} aa {aa}:', ', '.join(row))
    ds.close()
PY
