#!/bin/bash
# A/B of library builds on the 12 M-triangle scene (bench.py --workload cubes12m, parity on): bash tools/ab_12m.sh <a.so|-> <b.so> ... ; rounds in $ROUNDS
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in $(seq 1 ${ROUNDS:-2}); do
  for so in "$@"; do
    if [ "$so" = "-" ]; then unset BVH_AMD_SO; else export BVH_AMD_SO=$R/$so; fi
    python bench.py --workload cubes12m --steps 10 --warmup 2 --settle-steps 2 --regions 3 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-excluded 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d=json.load(open(d['detail'])) if 'detail' in d and 'phases_ms' not in d else d
print('$so', d['value'], d['ms_per_step'], d.get('phases_ms'), (d.get('parity') or {}).get('equal'))"
  done
done
