#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats of tools/gpu_nlp_solve_bench.py (the batched interior-point solver).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_nlp_solve
W=/tmp/prof_nlp_solve
rm -rf $W; mkdir -p $OUT $W
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -- python3 $ROOT/tools/gpu_nlp_solve_bench.py > $OUT/bench_stats.json 2> $OUT/stats.err
echo "stats rc=$?"
find $W/stats -name "*kernel_stats.csv" | while read f; do cp "$f" $OUT/kernel_stats.csv; done
du -sh $OUT
