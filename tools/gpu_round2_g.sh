#!/bin/bash
# iteration check: parity + contract tests, bench line, slot sweep
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_contract.py -x -q 2>&1 | tail -15 ) > gpurun_out/g_tests.log 2>&1
( timeout 600 python bench.py --no-cpu-baseline --no-extra > gpurun_out/g2_bench.json 2> gpurun_out/g2_bench.err )
( timeout 300 python tools/slots_diag.py f32 > gpurun_out/g_slots_f32.log 2>&1 )
( timeout 300 python tools/slots_diag.py f64 > gpurun_out/g_slots_f64.log 2>&1 )
tail -n 5 gpurun_out/g_tests.log; head -c 600 gpurun_out/g2_bench.json; echo; cat gpurun_out/g_slots_f32.log gpurun_out/g_slots_f64.log
