#!/bin/bash
# the default bench command as the driver runs it, N times on one box: how far the compact line's figures move from run to run
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r6_repeat
for i in $(seq 1 ${1:-3}); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out gpurun_out/r6_repeat/detail_$i.json 2>/dev/null | tee gpurun_out/r6_repeat/line_$i.json | python -c "
import json,sys; t=sys.stdin.read().strip().splitlines()[-1]; j=json.loads(t); x=j['step_excludes']; c=j['cpu_baseline']
print('run $i: bytes', len(t), 'value', j['value'], 'regions', j['regions_ms_per_step'], 'lazy', x.get('lazy_flat_array'), 'host_io', x.get('host_io'), 'pipelined', j['pipelined']['value'],
      'extras', [e.get('value') for e in j['extra_configs']], 'parity', all(e['parity']['equal'] for e in j['extra_configs']) and j['parity']['equal'],
      'cpu', c['value'], c.get('legs'), c.get('host_load_1m'))"
done 2>&1 | tee gpurun_out/r6_repeat/summary.log
