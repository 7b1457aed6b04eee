# Developer diagnostic: per-kernel times of the bench loop.  Usage on the GPU box: bash tools/build_prof.sh [tag]
cd /tmp && export TMPDIR=/tmp; R=/root/repo; tag=${1:-cur}; mkdir -p $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o out -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --pipeline-streams 0 > $R/gpurun_out/prof_$tag/bench.json 2> $R/gpurun_out/prof_$tag/bench.err
db=$(ls $R/gpurun_out/prof_$tag/*.db $R/gpurun_out/prof_$tag/*/*.db 2>/dev/null | head -1)
python $R/tools/prof_summary.py $db $R/gpurun_out/prof_$tag/kernel_stats.md "$tag" | head -30
cat $R/gpurun_out/prof_$tag/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['phases_ms'])"
