#!/bin/bash
# GPU timeline of one resident step: tools/timeline.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tl_$1; shift
rm -rf $O && mkdir -p $O
cd $R
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -- python3 bench.py --steps 6 --warmup 2 --pipelines 1 --no-cpu-baseline --no-real-reads "$@" > $O/bench.json 2> $O/bench.err
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
g=$(find $O/t -name "*memory_copy_trace.csv" | head -1)
python3 tools/timeline.py "$f" "$g" 2>&1 | tail -${TL_TAIL:-120}
rm -rf $O/t
