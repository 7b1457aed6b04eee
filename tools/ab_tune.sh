#!/bin/bash
# A/B of one tuning knob on the same box: bash tools/ab_tune.sh <ENVVAR=value ...> -- alternates the default run and the run with the environment set
# (bench.py reads BVH_TUNE_<knob number>=<value> overrides for developer A/B runs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in $(seq 1 ${ROUNDS:-3}); do
  for mode in base alt; do
    if [ $mode = alt ]; then export "$@"; fi
    python bench.py --steps 300 --warmup 20 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', d['value'], d['ms_per_step'], d['phases_ms'])"
    if [ $mode = alt ]; then for kv in "$@"; do unset "${kv%%=*}"; done; fi
  done
done
