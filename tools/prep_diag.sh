#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  out=$R/gpurun_out/prep$v
  BVH_AMD_SO=$R/tools/libbvh_prep$v.so rocprofv3 --kernel-trace --stats -d $out -o out -- python $R/tools/prep_diag.py > $out.log 2>&1
  db=$(ls $out/*.db $out/*/*.db 2>/dev/null | head -1)
  python $R/tools/prof_summary.py $db $out.md "prep variant $v" > /dev/null
  echo "== variant $v: $(grep build_ms $out.log)"; grep "k_prep\|k_bin\|k_split\|k_mid\|k_small\|k_publish" $out.md
done
