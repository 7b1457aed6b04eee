#!/bin/bash
# per-kernel averages of one workload under rocprofv3: bash tools/kstats_workload.sh <tag> "<bench.py args>" [ENV=...]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; wargs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/ksw_$tag
env "$@" rocprofv3 --kernel-trace --stats -d $out -o out -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity $wargs > $out.json 2> $out.err
db=$(ls $out/*.db $out/*/*.db 2>/dev/null | head -1)
python $R/tools/prof_summary.py $db $out.md "$tag: $wargs" > /dev/null
head -24 $out.md | cut -c1-120
