#!/bin/bash
# per-kernel averages of a short bench run under rocprofv3 for several library builds: bash tools/kstats.sh <pattern> a.so b.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  out=$R/gpurun_out/ks_$(basename $so .so)
  BVH_AMD_SO=$R/$so rocprofv3 --kernel-trace --stats -d $out -o out -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity > $out.json 2> $out.err
  db=$(ls $out/*.db $out/*/*.db 2>/dev/null | head -1)
  python $R/tools/prof_summary.py $db $out.md "$so" > /dev/null
  echo "== $so"; grep -E "$pat" $out.md | cut -c1-110
done
