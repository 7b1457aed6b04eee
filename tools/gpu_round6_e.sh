#!/bin/bash
# round 6, call E: spread small item batches over all workgroup slots; host batch without the main-stream guard; then the whole GPU suite
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_e; mkdir -p $O
timeout 900 python tools/host_step_bench.py > $O/host_step_bench.log 2>&1; cat $O/host_step_bench.log
timeout 600 python tools/walk_size_sweep.py > $O/walk_size_sweep.log 2>&1; cat $O/walk_size_sweep.log
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
