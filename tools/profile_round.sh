#!/bin/bash
# Round profile on the GPU box:  bash tools/profile_round.sh <tag> [extra bench.py args]
#   1. rocprofv3 --kernel-trace --stats of the default bench command  → gpurun_out/<tag>/kernel_stats.md
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, L2 hit/miss, L1 accesses, SQ) of the same command, as
#      MI355X_MICROARCH.md prescribes (never together with trace domains other than --kernel-trace)
#   3. gpurun_out/<tag>/bound.json: per kernel the counters per launch and the fraction of every resource's peak
# Copy the results you want judged into profiles/ (tools/profile_round.sh does it: profiles/<tag>_*).
tag=${1:-round}
shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --pipeline-streams 0 --no-extra --no-parity --no-excluded $*"   # (--no-excluded: the host-resident variants walk the batch in chunks — the same kernel at other sizes would dilute the per-launch means)
rocprofv3 --kernel-trace --stats -d $out/trace -o out -- $CMD > $out/bench_under_rocprof.json 2> $out/trace.err
db=$(ls $out/trace/*.db $out/trace/*/*.db 2>/dev/null | head -1)
python $R/tools/prof_summary.py $db $out/kernel_stats.md "$tag: $CMD" > /dev/null
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; do
  n=$(echo $set | tr " " "_" | cut -c1-32)
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $out/pmc_$n -o out --output-format csv -- $CMD > $out/pmc_$n.log 2>&1
done
python $R/tools/pmc_summary.py $out $db > $out/pmc_summary.md
cat $out/pmc_summary.md
mkdir -p $R/gpurun_out/profiles_$tag
for f in kernel_stats.md pmc_summary.md bound.json bench_under_rocprof.json; do cp $out/$f $R/gpurun_out/profiles_$tag/${tag}_$f 2>/dev/null; done
# keep the merge-back small (gpurun copies at most 64 MiB of gpurun_out): the rocpd database and the counter CSVs have been summarised
if [ -z "$KEEP_RAW" ]; then rm -rf $out/trace $out/pmc_*; fi
