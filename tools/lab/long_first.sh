#!/bin/bash
bash tools/planprof.sh r06_db20_b1 db20 4096 4096 3 1 > gpurun_out/planprof_r06_db20_b1.log 2>&1
bash tools/planprof.sh r06_db20_b4 db20 4096 4096 3 4 > gpurun_out/planprof_r06_db20_b4.log 2>&1
cat gpurun_out/planprof_r06_db20_b1/summary.txt | cut -c1-330
cat gpurun_out/planprof_r06_db20_b4/summary.txt | cut -c1-330
rm -rf gpurun_out/planprof_r06_db20_b1/pmc_* gpurun_out/planprof_r06_db20_b1/stats gpurun_out/planprof_r06_db20_b4/pmc_* gpurun_out/planprof_r06_db20_b4/stats
