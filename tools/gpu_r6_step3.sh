#!/bin/bash
# round 6: what binds the streamed build from many small slot files?  CPU share of the box, a kernel + copy trace, a sweep of the host knobs
set -o pipefail
O=gpurun_out/r6s3
mkdir -p $O
g++ -O2 -pthread -o /tmp/cpu_share_probe tools/cpu_share_probe.cpp && /tmp/cpu_share_probe > $O/cpu_share.txt 2>&1
cat $O/cpu_share.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 tools/streamed_files_trace.py run /dev/shm > $O/trace_run.txt 2>&1 || { tail -5 $O/trace_run.txt; exit 1; }
tail -4 $O/trace_run.txt
python3 tools/streamed_files_trace.py summarize $O/trace $O/trace_summary.txt > /dev/null 2>&1 || echo "summarize failed (will be done locally)"
# keep only what the summary needs (the merge back is capped)
find $O/trace -name "*_agent_info.csv" -delete 2>/dev/null
for ft in 4 8 12; do for jt in 8 12 16; do
  CP2_INGEST_THREADS=$ft SFAB_THREADS=$jt timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/sweep_fill${ft}_json${jt}.txt 2>&1 || { tail -5 $O/sweep_fill${ft}_json${jt}.txt; exit 1; }
  echo "fill $ft json $jt: $(grep 'file/fake' $O/sweep_fill${ft}_json${jt}.txt)"
done; done
CP2_INGEST_RING=5 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/sweep_ring5.txt 2>&1 || exit 1
echo "ring 5: $(grep 'file/fake' $O/sweep_ring5.txt)"
