#!/bin/bash
# One step of bench.py as a timeline (rocprofv3 kernel + memory-copy trace); outputs under gpurun_out/prof_tl.
# Read it with tools/timeline_r02.py (prints every kernel / copy of the last step with the idle gap before it).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_tl
rm -rf $O && mkdir -p $O
cd $R
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-real-reads --no-host-to-host --no-overlap > $O/t.log 2>&1
find $O/t -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
find $O/t -name "*memory_copy_trace.csv" -exec cp {} $O/memcpy_trace.csv \;
rm -rf $O/t
