#!/bin/bash
set -o pipefail
O=gpurun_out/r6s12
mkdir -p $O
CP2_TRACE=1 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/small_trace.txt 2>&1 || { tail -5 $O/small_trace.txt; exit 1; }
grep -E "building thread|sampling hook|file/fake" $O/small_trace.txt
CP2_TRACE=1 CP2_INGEST_CHUNK_MB=1024 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/small_trace_1024.txt 2>&1 || exit 1
grep -E "building thread|sampling hook|file/fake" $O/small_trace_1024.txt
