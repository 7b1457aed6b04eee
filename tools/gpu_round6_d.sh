#!/bin/bash
# round 6, call D: the two-phase host batch (upload before the build, small last chunk): parity, rate, timeline
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_d; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_host.py -x -q 2>&1 | tail -15
timeout 900 python tools/host_step_bench.py > $O/host_step_bench.log 2>&1; cat $O/host_step_bench.log
cd /tmp && export TMPDIR=/tmp
for ch in 0; do
HOST_CHUNKS=$ch rocprofv3 --kernel-trace --memory-copy-trace -d $O/tl$ch -o out -- python $R/tools/host_step_one.py > $O/tl$ch.log 2>&1
db=$(ls $O/tl$ch/*.db $O/tl$ch/*/*.db 2>/dev/null | head -1)
python $R/tools/host_timeline.py $db 12 > $O/timeline_chunks$ch.txt 2>&1
cat $O/timeline_chunks$ch.txt
rm -rf $O/tl$ch
done
