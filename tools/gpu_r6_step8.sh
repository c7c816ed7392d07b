#!/bin/bash
# round 6: two-deep fill pipeline across turns
set -o pipefail
O=gpurun_out/r6s8
mkdir -p $O
(numactl -H; lscpu | grep -i -E "numa|socket|model name") > $O/numa.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_round6.py -q -m gpu -x > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 3 > $O/small_default.txt 2>&1 || { tail -5 $O/small_default.txt; exit 1; }
echo "default (512 MiB with room): $(grep 'file/fake' $O/small_default.txt)"
CP2_INGEST_CHUNK_MB=256 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/small_256.txt 2>&1 || exit 1
echo "256: $(grep 'file/fake' $O/small_256.txt)"
CP2_INGEST_CHUNK_MB=1024 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/small_1024.txt 2>&1 || exit 1
echo "1024: $(grep 'file/fake' $O/small_1024.txt)"
CP2_INGEST_THREADS=6 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/small_fill6.txt 2>&1 || exit 1
echo "fill 6: $(grep 'file/fake' $O/small_fill6.txt)"
CP2_INGEST_THREADS=12 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/small_fill12.txt 2>&1 || exit 1
echo "fill 12: $(grep 'file/fake' $O/small_fill12.txt)"
timeout -k 10 500 python tools/streamed_files_ab.py /dev/shm big 16 2 > $O/big_default.txt 2>&1 || exit 1
echo "big: $(grep 'file/fake' $O/big_default.txt)"
