#!/bin/bash
# Test counts per ray of a library build on the bench workload at reduced spp: [SCENE=x.sdf] scripts/census.sh <aa> <lib>...
AA=$1; shift
SCENE=${SCENE:-cornell_buster.sdf}
for lib in "$@"; do
  timeout -k 10 200 python3 bench.py --lib $lib --no-also --steps 1 --warmup 1 --cpu-seconds 0 --aa $AA --scene $SCENE --queue-gib ${QG:-32} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$lib $SCENE', 'Msamples/s %.0f' % d['value'], 'trace', {k: round(v, 2) for k, v in d['work']['tests_per_trace_ray'].items()}, 'shadow', {k: round(v, 2) for k, v in d['work']['tests_per_shadow_ray'].items()})"
done
