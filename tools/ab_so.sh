#!/bin/bash
# one workload, several library builds side by side: bash tools/ab_so.sh "<bench args>" bvh_amd/libbvh_mi355x.so tools/<variant>.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
wargs=$1; shift
for i in $(seq 1 ${ROUNDS:-2}); do
  for so in "$@"; do
    BVH_AMD_SO=$R/$so python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity $wargs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so', d['value'], d['ms_per_step'], d['phases_ms'])"
  done
done
