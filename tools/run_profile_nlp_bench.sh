#!/bin/bash
# BASELINE config 5 on one GPU: bench.py --workload nlp --batch <routes>, plain and under rocprofv3 --kernel-trace --stats.
#   tools/run_profile_nlp_bench.sh <routes> <tag>
R=$1; TAG=$2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
W=/tmp/prof_$TAG
rm -rf $W; mkdir -p $OUT $W
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --workload nlp --batch $R > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -- python3 $ROOT/bench.py --workload nlp --batch $R > $OUT/bench_stats.json 2> $OUT/stats.err
echo "stats rc=$?"
find $W/stats -name "*kernel_stats.csv" | while read f; do cp "$f" $OUT/stats_$(basename "$f"); done
head -c 1500 $OUT/bench.json
