#!/bin/bash
# End-of-round record on a GPU box (from the repo root): PMC passes + kernel statistics (tools/profile_r04_pmc.sh,
# tools/prof_stats.sh), the driver's bench command and the 1.25M-read line (tools/run_final_bench.sh), the GPU suite.
# Outputs under gpurun_out/; copy what is to be kept into profiles/rNN/.
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/profile_r04_pmc.sh > gpurun_out/pmc_r04.log 2>&1
tail -3 gpurun_out/pmc_r04.log
cp gpurun_out/prof_r04_pmc/pmc_traffic.json profiles/r04/pmc_traffic.json   # (so that the bench lines below carry `traffic`)
bash tools/prof_stats.sh r04p1 --steps 6 --warmup 2 --no-cpu-baseline --no-real-reads --no-host-to-host --pipelines 1 2>&1 | tail -2
bash tools/run_final_bench.sh
bash tools/run_gpu_suite.sh
