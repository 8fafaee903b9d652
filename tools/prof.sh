#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run from the repo root):
#   tools/prof.sh <tag> [bench args...]
# kernel-trace/stats pass, then separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a
# pass: TCC has 4 slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").
TAG=${1:-r01}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-extras $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_tcc -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_tcc.err
find $OUT -name "*.csv" | head -50
