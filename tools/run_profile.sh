#!/bin/bash
# Runs on the GPU box: kernel-trace/stats pass and separate PMC passes of the default bench command for one workload.
#   tools/run_profile.sh <tag> <abmpc|fbmpc> [extra bench args]
# rocprofv3 writes under /tmp; only the CSV summaries are copied to gpurun_out/prof_<tag>/ (condensed afterwards with
# tools/profile_summary.py).
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
W=/tmp/prof_$TAG
rm -rf $W; mkdir -p $OUT $W
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --workload $WL --no-cpu-baseline --no-secondary $*"
run() {   # name, rocprof args...
    local name=$1; shift
    rocprofv3 "$@" --output-format csv -d $W/$name -- python3 $B > $OUT/bench_$name.json 2> $OUT/$name.err
    echo "$name rc=$?"
    find $W/$name -name "*.csv" | while read f; do cp "$f" $OUT/${name}_$(basename "$f"); done
}
run stats --kernel-trace --stats
if [ -z "$EEPACC_PROFILE_NO_DRIVER" ]; then B="$B --steps 20 --warmup 5" run stats_driver --kernel-trace --stats; fi
run pmc1 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU_TRANS_F64
run pmc2 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT
run pmc3 --pmc FETCH_SIZE
run pmc4 --pmc WRITE_SIZE
# stall attribution (round 3): which unit the waves' instructions occupy, memory instructions by kind, and the mean number of
# outstanding LDS / vector-memory / scalar-memory instructions (LEVEL / INSTS = mean latency in cycles)
run pmc5 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM
run pmc6 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
du -sh $OUT
