#!/bin/bash
# rocprofv3 kernel trace of the long-read workload (config 4 at 8,000 reads); summary left in gpurun_out/prof_lr/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_lr
rm -rf $O && mkdir -p $O
cd $R
LR="bench.py --scaling weak --reads-per-gpu 8000 --read-len 10000 --steps 2 --warmup 1 --no-cpu-baseline --pipelines 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $LR > $O/run.json 2> $O/run.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/kt -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/kt
head -6 $O/kernel_stats.csv | cut -c1-150
grep "k_align_reads" $O/kernel_trace.csv | awk -F, '{print $(NF-1)-$(NF-2), $0}' | cut -c1-60 | head -12
