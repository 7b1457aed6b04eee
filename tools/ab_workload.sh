#!/bin/bash
# one workload, several settings of the environment side by side on ONE box:
#   bash tools/ab_workload.sh "<bench.py args>" "" "BVH_TUNE_12=0" "BVH_TUNE_12=4" ...   (each further argument: space-separated VAR=value list, "" = defaults)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
wargs=$1; shift
for i in $(seq 1 ${ROUNDS:-2}); do
  for envs in "$@"; do
    env $envs python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity $wargs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; print('[$envs]', d['value'], d['ms_per_step'], p, 'csr_ms', round(p['traverse_total_ms']-p['traverse_kernel_ms'],4))"
  done
done
