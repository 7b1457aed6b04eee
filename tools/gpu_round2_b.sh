#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide or variants or config1 or config4 or config0 or full_size" 2>&1 | tail -30 ) > gpurun_out/b_tests.log 2>&1
( timeout 300 python tools/wide_sweep.py f32 1000000 > gpurun_out/b_sweep_f32.log 2>&1 )
( timeout 200 python tools/wide_sweep.py f64 1000000 > gpurun_out/b_sweep_f64.log 2>&1 )
( timeout 200 python tools/wide_sweep.py f32 2000000 > gpurun_out/b_sweep_f32_2m.log 2>&1 )
( timeout 600 python bench.py --no-extra > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err )
tail -n 5 gpurun_out/b_tests.log; head -c 1200 gpurun_out/b_bench.json
