#!/bin/bash
# GPU-side check of a tree (run through gpurun from the repo root): the GPU tests, then an
# A/B of library builds at reduced spp on the two bench scenes.
# usage: scripts/gpu_check.sh <outdir> [lib ...]
set -u
OUT=$1; shift
mkdir -p $OUT
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest.log 2>&1
rc=$?
tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit $rc
for scene in cornell_buster.sdf example.sdf; do
  SCENE=$scene scripts/ab.sh ${AA:-16} "$@" 2>&1 | tee -a $OUT/ab.log
done
