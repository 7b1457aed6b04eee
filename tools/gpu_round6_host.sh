#!/bin/bash
# the host-resident batch (ABI 7) on the GPU box: parity, the rate by chunk count / zero-copy setting / ray layout, and a timeline of one frame
# (kernels + transfers on one time axis)        bash tools/gpu_round6_host.sh        → gpurun_out/r6_host/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_host; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_host.py -x -q 2>&1 | tail -5
timeout 300 python tools/pcie_probe.py > $O/pcie_probe.log 2>&1; cat $O/pcie_probe.log
timeout 900 python tools/host_step_bench.py > $O/host_step_bench.log 2>&1; cat $O/host_step_bench.log
cd /tmp && export TMPDIR=/tmp
HOST_CHUNKS=0 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tl -o out -- python $R/tools/host_step_one.py > $O/tl.log 2>&1
db=$(ls $O/tl/*.db $O/tl/*/*.db 2>/dev/null | head -1)
python $R/tools/host_timeline.py $db 12 > $O/timeline.txt 2>&1; cat $O/timeline.txt
rm -rf $O/tl
