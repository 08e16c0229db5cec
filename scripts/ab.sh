#!/bin/bash
# A/B of library builds on the bench workload at reduced spp: [SCENE=x.sdf] scripts/ab.sh <aa> <lib>...
AA=$1; shift
SCENE=${SCENE:-cornell_buster.sdf}
for lib in "$@"; do
  for rep in 1 2; do
    timeout -k 10 200 python3 bench.py --lib $lib --no-census --no-also --steps 2 --warmup 1 --cpu-seconds 0 --aa $AA --scene $SCENE --pipeline ${PIPE:-default} --queue-gib ${QG:-32} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$lib $SCENE', 'Msamples/s %.0f' % d['value'], 'ms/step %.1f' % d['ms_per_step'], 'rays/sample %.2f' % d['work']['rays_per_sample'], d['kernel_time']['pipeline'], 'k0', d['kernel_time']['closest_hit_or_fused_ms_per_depth'], 'shade', d['kernel_time']['shade_ms_per_depth'], 'shadow', d['kernel_time']['shadow_ms_per_depth'])"
  done
done
