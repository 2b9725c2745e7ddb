#!/bin/bash
# Runs on the GPU box: kernel-trace/stats pass and separate PMC passes of tools/gpu_nlp_bench.py (batch 1024).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_nlp
W=/tmp/prof_nlp
rm -rf $W; mkdir -p $OUT $W
cd /tmp && export TMPDIR=/tmp
B="$ROOT/tools/gpu_nlp_bench.py --batches 1024 --steps 20 --no-cpu"
run() {
    local name=$1; shift
    rocprofv3 "$@" --output-format csv -d $W/$name -- python3 $B > $OUT/bench_$name.json 2> $OUT/$name.err
    echo "$name rc=$?"
    find $W/$name -name "*.csv" | while read f; do cp "$f" $OUT/${name}_$(basename "$f"); done
}
run stats --kernel-trace --stats
run pmc1 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VALU_TRANS_F64
run pmc2 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
run pmc3 --pmc FETCH_SIZE
run pmc4 --pmc WRITE_SIZE
du -sh $OUT
