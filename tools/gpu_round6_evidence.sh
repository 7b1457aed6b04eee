#!/bin/bash
# Round-6 evidence run on the GPU box: rocprofv3 kernel trace + separate PMC passes (tools/profile_round.sh) for every BASELINE config, the
# reference's whole harness loop and the scene beyond the caches:
#   <tag>_c1         configs[1]  120 k triangles, 1 M rays, f32                                   (the headline)
#   <tag>_c1closest  configs[1]  the harness step: ray generation + build + flatten + walk + closest hit   (--harness closest)
#   <tag>_c1tris     configs[1]  the harness step with every Intersection kept                              (--harness triangles)
#   <tag>_c2         configs[2]  stand-in scene, 10 M primary rays
#   <tag>_c2closest  configs[2]  the harness step, closest hit
#   <tag>_c3         configs[3]  stand-in scene, one 12.5 M-ray incoherent shard
#   <tag>_c4         configs[4]  120 k triangles, 1 M rays, f64: the guide walk (default)
#   <tag>_c4f64      configs[4]  every slab test of the walk in double precision (BVHGPU_TUNE_WIDE_F64_GUIDE = 0)
#   <tag>_c4closest  configs[4] geometry and rays, the harness step in f64: closest hit over 16 items per ray ((ray, item) slots)  (--dtype f64 --harness closest)
#   <tag>_c12m       beyond BASELINE: create_n_cubes(1 000 000) = 12 M triangles, 10 M rays — the regime where SURVEY §8d's HBM roofline applies
# → gpurun_out/profiles_<tag>_<which>/ ; copy into profiles/.   usage: bash tools/gpu_round6_evidence.sh <tag> ["c1 c1closest …"]
tag=${1:-r6_v1}
which=${2:-"c1 c1closest c1tris c2 c2closest c3 c4 c4f64 c4closest c12m"}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for c in $which; do
  unset BVH_TUNE_14
  case $c in
    c1) args="" ;;
    c1closest) args="--harness closest" ;;
    c1tris) args="--harness triangles" ;;
    c2) args="--workload standin-primary" ;;
    c2closest) args="--workload standin-primary --harness closest" ;;
    c3) args="--workload standin-incoherent --scaling weak --rays 12500000" ;;
    c4) args="--dtype f64" ;;
    c4f64) args="--dtype f64"; export BVH_TUNE_14=0 ;;
    c4closest) args="--dtype f64 --harness closest" ;;
    c12m) args="--workload cubes12m --settle-steps 5" ;;
    *) echo "unknown config $c"; continue ;;
  esac
  ( timeout 900 bash tools/profile_round.sh ${tag}_$c $args > gpurun_out/${tag}_$c.log 2>&1 )
  tail -n 12 gpurun_out/${tag}_$c.log
done
