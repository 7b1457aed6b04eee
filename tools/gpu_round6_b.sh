#!/bin/bash
# round 6, call B: the closest-hit anomaly under the eager flatten, the PCIe link, the ABI 7 host batch (parity + rate)
set -x
O=gpurun_out/r6_b; mkdir -p $O
timeout 600 python tools/closest_diag.py > $O/closest_diag.log 2>&1; tail -25 $O/closest_diag.log
timeout 300 python tools/pcie_probe.py > $O/pcie_probe.log 2>&1; cat $O/pcie_probe.log
timeout 1200 python -m pytest tests/test_gpu_host.py -x -q 2>&1 | tail -15
timeout 600 python tools/host_step_bench.py > $O/host_step_bench.log 2>&1; cat $O/host_step_bench.log
