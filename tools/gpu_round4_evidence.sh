#!/bin/bash
# Round-4 evidence run on the GPU box: rocprofv3 kernel trace + separate PMC passes for EVERY BASELINE config, plus the pure-f64 walk:
#   <tag>_c1     configs[1]  120 k triangles, 1 M rays, f32        (the headline)
#   <tag>_c2     configs[2]  stand-in scene, 10 M primary rays
#   <tag>_c3     configs[3]  stand-in scene, one 12.5 M-ray incoherent shard
#   <tag>_c4     configs[4]  120 k triangles, 1 M rays, f64: the guide walk (default)
#   <tag>_c4f64  configs[4]  the same with every slab test of the walk in double precision (BVHGPU_TUNE_WIDE_F64_GUIDE = 0)
# → gpurun_out/profiles_<tag>_cK/ ; copy into profiles/.   usage: bash tools/gpu_round4_evidence.sh <tag> [which: "1 2 3 4 4f64"]
tag=${1:-r4_v1}
which=${2:-"1 2 3 4 4f64"}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for c in $which; do
  unset BVH_TUNE_14
  case $c in
    1) args="" ;;
    2) args="--workload standin-primary" ;;
    3) args="--workload standin-incoherent --scaling weak --rays 12500000" ;;
    4) args="--dtype f64" ;;
    4f64) args="--dtype f64"; export BVH_TUNE_14=0 ;;
  esac
  ( timeout 900 bash tools/profile_round.sh ${tag}_c$c $args > gpurun_out/${tag}_c$c.log 2>&1 )
  tail -n 12 gpurun_out/${tag}_c$c.log
done
