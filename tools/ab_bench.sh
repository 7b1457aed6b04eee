#!/bin/bash
# A/B of library builds on the GPU box through bench.py (parity on):  bash tools/ab_bench.sh <out_log> "<bench args>" <so_A|-> <so_B|-> [reps]
# "-" = the in-tree library.  Prints value / ms_per_step / phases / parity per run, alternating A and B.
out=$1; args=$2; A=$3; B=$4; reps=${5:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p $(dirname $out); : > $out
for rep in $(seq $reps); do
  for so in "$A" "$B"; do
    if [ "$so" = "-" ]; then unset BVH_AMD_SO; else export BVH_AMD_SO=$R/$so; fi
    python bench.py --no-cpu-baseline --pipeline-streams 0 --no-extra $args 2>/dev/null | python -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$so', j['value'], j['ms_per_step'], j['phases_ms'], 'parity', j.get('parity', {}).get('equal'))" >> $out
  done
done
cat $out
