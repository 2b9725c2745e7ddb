#!/bin/bash
# Runs on the GPU box: kernel-trace/stats pass and two PMC passes of tools/gpu_riccati_bench.py (k_riccati).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_riccati
W=/tmp/prof_riccati
rm -rf $W; mkdir -p $OUT $W
cd /tmp && export TMPDIR=/tmp
run() {
    local name=$1; shift
    rocprofv3 "$@" --output-format csv -d $W/$name -- python3 $ROOT/tools/gpu_riccati_bench.py > $OUT/bench_$name.json 2> $OUT/$name.err
    echo "$name rc=$?"
    find $W/$name -name "*.csv" | while read f; do cp "$f" $OUT/${name}_$(basename "$f" | sed 's/^[0-9]*_//'); done
}
run stats --kernel-trace --stats
run pmc1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64
run pmc2 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
du -sh $OUT
