#!/bin/bash
# round 6, call C: timeline of the host-resident step (kernels + copies)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ch in 4 1; do
HOST_CHUNKS=$ch rocprofv3 --kernel-trace --memory-copy-trace -d $O/tl$ch -o out -- python $R/tools/host_step_one.py > $O/tl$ch.log 2>&1
db=$(ls $O/tl$ch/*.db $O/tl$ch/*/*.db 2>/dev/null | head -1)
python $R/tools/host_timeline.py $db 12 > $O/timeline_chunks$ch.txt 2>&1
cat $O/timeline_chunks$ch.txt
rm -rf $O/tl$ch
done
