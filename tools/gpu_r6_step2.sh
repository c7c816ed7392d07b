#!/bin/bash
# round 6: streamed-from-files A/B, the round-5 library against this one, same box, slot files in /dev/shm (tmpfs = page cache)
set -o pipefail
mkdir -p gpurun_out/r6s2
python -c "import os; print('affinity', len(os.sched_getaffinity(0)))" > gpurun_out/r6s2/box.txt 2>&1
CODEX_P2_LIB=$PWD/build/libcodex_p2_r05.so timeout -k 10 400 python tools/streamed_files_ab.py /dev/shm small - 2 > gpurun_out/r6s2/small_r05.txt 2>&1 || { tail -5 gpurun_out/r6s2/small_r05.txt; exit 1; }
tail -2 gpurun_out/r6s2/small_r05.txt
timeout -k 10 400 python tools/streamed_files_ab.py /dev/shm small - 3 > gpurun_out/r6s2/small_new.txt 2>&1 || { tail -5 gpurun_out/r6s2/small_new.txt; exit 1; }
tail -2 gpurun_out/r6s2/small_new.txt
CODEX_P2_LIB=$PWD/build/libcodex_p2_r05.so timeout -k 10 500 python tools/streamed_files_ab.py /dev/shm big 16 2 > gpurun_out/r6s2/big_r05.txt 2>&1 || { tail -5 gpurun_out/r6s2/big_r05.txt; exit 1; }
tail -2 gpurun_out/r6s2/big_r05.txt
timeout -k 10 500 python tools/streamed_files_ab.py /dev/shm big 16 3 > gpurun_out/r6s2/big_new.txt 2>&1 || { tail -5 gpurun_out/r6s2/big_new.txt; exit 1; }
tail -2 gpurun_out/r6s2/big_new.txt
