#!/bin/bash
# rocprofv3 evidence for one SWT plan's level kernels (developer tool): tools/swtprof.sh <tag> wname rows cols levels
set -uo pipefail
TAG=$1; shift
OUT=gpurun_out/swtprof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/ktimes.py $1 $2 $3 $4 1 1 > $OUT/ktimes.txt 2> $OUT/stats.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 tools/ktimes.py $1 $2 $3 $4 1 1 > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 tools/ktimes.py $1 $2 $3 $4 1 1 > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/ktimes.py $1 $2 $3 $4 1 1 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/ktimes.py $1 $2 $3 $4 1 1 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_tcc -- python3 tools/ktimes.py $1 $2 $3 $4 1 1 > /dev/null 2> $OUT/pmc_tcc.err
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/ktimes.txt
grep -v "^  __amd\|fill_hash" $OUT/summary.txt | cut -c1-400
