#!/bin/bash
# bench phases for several builds of the library: bash tools/ab_multi.sh a.so b.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do
  for so in "$@"; do
    BVH_AMD_SO=$R/$so python bench.py --steps 300 --warmup 20 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so', d['value'], d['ms_per_step'], d['phases_ms'])"
  done
done
