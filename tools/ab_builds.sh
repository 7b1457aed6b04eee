#!/bin/bash
# build time by scene size for several builds of the library on ONE box: bash tools/ab_builds.sh <a.so|-> <b.so> ...   (rounds in $ROUNDS, sizes in $SIZES)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in $(seq 1 ${ROUNDS:-2}); do
  for so in "$@"; do
    if [ "$so" = "-" ]; then unset BVH_AMD_SO; else export BVH_AMD_SO=$R/$so; fi
    python tools/build_time.py $SIZES 2>/dev/null | tail -1
  done
done
