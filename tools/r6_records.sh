#!/bin/bash
# Round 6's committed records of the streamed build from slot files, collected in ONE gpurun call on ONE box:
#   bash tools/r6_records.sh            (needs build/libcodex_p2_r05.so: round 5's library, built from commit c7434d0, for the A/B)
# -> gpurun_out/r6rec/{ab_*.txt, trace/, trace_run.txt}; then locally:
#   python3 tools/streamed_files_trace.py summarize gpurun_out/r6rec/trace profiles/r06_streamed_files_trace.txt
#   python3 tools/r6_ab_record.py gpurun_out/r6rec > profiles/r06_streamed_files_ab.txt
set -o pipefail
O=gpurun_out/r6rec
mkdir -p $O
{ echo "host $(hostname)"; date -u +%FT%TZ; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; df -h /tmp /dev/shm | tail -2; } > $O/box.txt 2>&1
R5=$PWD/build/libcodex_p2_r05.so
CODEX_P2_LIB=$R5 timeout -k 10 400 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/ab_small_shm_r05.txt 2>&1 || { tail -5 $O/ab_small_shm_r05.txt; exit 1; }
echo "small, /dev/shm, round 5's library: $(grep 'file/fake' $O/ab_small_shm_r05.txt)"
timeout -k 10 400 python tools/streamed_files_ab.py /dev/shm small - 4 > $O/ab_small_shm.txt 2>&1 || { tail -5 $O/ab_small_shm.txt; exit 1; }
echo "small, /dev/shm: $(grep 'file/fake' $O/ab_small_shm.txt)"
timeout -k 10 400 python tools/streamed_files_ab.py /tmp small - 4 > $O/ab_small_tmp.txt 2>&1 || { tail -5 $O/ab_small_tmp.txt; exit 1; }
echo "small, /tmp: $(grep 'file/fake' $O/ab_small_tmp.txt)"
for mb in 256 384 1024; do
  CP2_INGEST_CHUNK_MB=$mb timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/ab_small_shm_chunk$mb.txt 2>&1 || exit 1
  echo "small, /dev/shm, $mb MiB turns: $(grep 'file/fake' $O/ab_small_shm_chunk$mb.txt)"
done
CP2_INGEST_COPY_STREAM=1 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/ab_small_shm_copystream.txt 2>&1 || exit 1
echo "small, /dev/shm, separate copy stream: $(grep 'file/fake' $O/ab_small_shm_copystream.txt)"
CP2_INGEST_MAPPED=1 timeout -k 10 300 python tools/streamed_files_ab.py /tmp small - 2 > $O/ab_small_tmp_mapped.txt 2>&1 || exit 1
echo "small, /tmp, mapped: $(grep 'file/fake' $O/ab_small_tmp_mapped.txt)"
CODEX_P2_LIB=$R5 timeout -k 10 500 python tools/streamed_files_ab.py /dev/shm big 16 2 > $O/ab_big_shm_r05.txt 2>&1 || exit 1
echo "big, round 5's library: $(grep 'file/fake' $O/ab_big_shm_r05.txt)"
timeout -k 10 500 python tools/streamed_files_ab.py /dev/shm big 16 3 > $O/ab_big_shm.txt 2>&1 || exit 1
echo "big: $(grep 'file/fake' $O/ab_big_shm.txt)"
CP2_INGEST_COPY_STREAM=1 timeout -k 10 500 python tools/streamed_files_ab.py /dev/shm big 16 2 > $O/ab_big_shm_copystream.txt 2>&1 || exit 1
echo "big, separate copy stream: $(grep 'file/fake' $O/ab_big_shm_copystream.txt)"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/trace
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 tools/streamed_files_trace.py run /tmp > $O/trace_run.txt 2>&1 || { tail -5 $O/trace_run.txt; exit 1; }
grep -v "^[EW]2026" $O/trace_run.txt | tail -4
find $O/trace -name "*_agent_info.csv" -delete 2>/dev/null
