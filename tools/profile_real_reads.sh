#!/bin/bash
# rocprofv3 kernel trace of the real-reads leg (tools/real_reads.py); summary left in gpurun_out/prof_rr/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_rr
rm -rf $O && mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/real_reads.py --steps 4 > $O/run.json 2> $O/run.err
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
head -12 $O/kernel_stats.csv | cut -c1-150
tail -1 $O/run.json | cut -c1-200
