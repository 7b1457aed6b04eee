#!/bin/bash
# round 6, call A: the refactored bench.py (compact line + side file) as the driver runs it, then the bench / dist GPU tests
set -x
O=gpurun_out/r6_a; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_detail.json ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.err
wc -c $O/bench_default.json
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_bench.py -x -q -k "not default_line" 2>&1 | tail -15
