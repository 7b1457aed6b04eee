#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/tl
rocprofv3 --kernel-trace -d $out -o out -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity > $out.json 2> $out.err
db=$(ls $out/*.db $out/*/*.db 2>/dev/null | head -1)
python $R/tools/timeline.py $db 12
python $R/tools/timeline.py $db 20 | tail -3
python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity 2>/dev/null | head -c 700
