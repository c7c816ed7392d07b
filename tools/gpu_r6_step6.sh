#!/bin/bash
# round 6: trace of the streamed build from files with the rebalanced fill
set -o pipefail
O=gpurun_out/r6s11
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 tools/streamed_files_trace.py run /dev/shm > $O/trace_run.txt 2>&1 || { tail -5 $O/trace_run.txt; exit 1; }
grep -v "^[EW]2026" $O/trace_run.txt | tail -4
find $O/trace -name "*_agent_info.csv" -delete 2>/dev/null
