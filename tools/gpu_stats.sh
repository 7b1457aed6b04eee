#!/bin/bash
# quick per-kernel timing of the default bench loop: bash tools/gpu_stats.sh <tag> [bench args]
tag=${1:-s}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_trace -o out -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity $* > $R/gpurun_out/${tag}_bench.json 2> /dev/null
cd $R; db=$(ls gpurun_out/${tag}_trace/*.db gpurun_out/${tag}_trace/*/*.db 2>/dev/null | head -1); python tools/prof_summary.py $db gpurun_out/${tag}_kernel_stats.md $tag > /dev/null; rm -rf gpurun_out/${tag}_trace
head -24 gpurun_out/${tag}_kernel_stats.md
