#!/bin/bash
# A/B of two builds of the library on the same box: bash tools/ab.sh <a.so|-> <b.so> [rounds]   (- = the in-tree library)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in $(seq 1 ${3:-3}); do
  for so in $1 $2; do
    if [ "$so" = "-" ]; then unset BVH_AMD_SO; else export BVH_AMD_SO=$R/$so; fi
    python bench.py --steps 300 --warmup 20 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so', d['value'], d['ms_per_step'], d['phases_ms'])"
  done
done
