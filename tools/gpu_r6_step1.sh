#!/bin/bash
# round 6, first GPU step: the new GPU tests, then the streamed-from-files A/B with the round-5 library and with this one
set -o pipefail
mkdir -p gpurun_out/r6s1
(df -h /tmp /dev/shm .; nproc; free -g) > gpurun_out/r6s1/box.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > gpurun_out/r6s1/pytest_round6.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6s1/pytest_round6.txt
tail -3 gpurun_out/r6s1/pytest_round6.txt
CODEX_P2_LIB=$PWD/build/libcodex_p2_r05.so timeout -k 10 300 python tools/streamed_files_ab.py - small - 2 > gpurun_out/r6s1/small_r05.txt 2>&1; tail -2 gpurun_out/r6s1/small_r05.txt
timeout -k 10 300 python tools/streamed_files_ab.py - small - 3 > gpurun_out/r6s1/small_new.txt 2>&1; tail -2 gpurun_out/r6s1/small_new.txt
