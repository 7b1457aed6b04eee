#!/bin/bash
# round 6, call H: VERDICT r5 #3 — the level tier's hand-over size (768 -> 1024 / 1536 shapes) with a larger workgroup-tier workgroup; byte-equal trees are checked by bench.py's parity leg
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_h; mkdir -p $O
cd $R
for i in 1 2 3; do
  for so in bvh_amd/libbvh_mi355x.so tools/libbvh_mid1024_512.so tools/libbvh_mid1536_512.so tools/libbvh_mid1024_384.so; do
    BVH_AMD_SO=$R/$so python bench.py --steps 200 --warmup 10 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-excluded 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']; print('$so', d['value'], d['ms_per_step'], 'build+flatten', round(p['build_ms']+p['flatten_ms'],4), 'levels', d.get('build_levels'), 'parity', d['parity']['equal'], d['parity'].get('bvh_nodes_equal'))"
  done
done 2>&1 | tee $O/mid_threshold_ab.log
