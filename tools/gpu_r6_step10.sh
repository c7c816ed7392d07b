#!/bin/bash
# round 6: uploads on the turn's own hashing stream (no fifth stream sharing a hardware queue) against the separate copy stream
set -o pipefail
O=gpurun_out/r6s10
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -m gpu -x > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 3 > $O/small_default.txt 2>&1 || { tail -5 $O/small_default.txt; exit 1; }
echo "default: $(grep 'file/fake' $O/small_default.txt)"
CP2_INGEST_COPY_STREAM=1 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 3 > $O/small_copystream.txt 2>&1 || exit 1
echo "separate copy stream: $(grep 'file/fake' $O/small_copystream.txt)"
CP2_INGEST_CHUNK_MB=256 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/small_256.txt 2>&1 || exit 1
echo "256: $(grep 'file/fake' $O/small_256.txt)"
CP2_INGEST_CHUNK_MB=1024 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/small_1024.txt 2>&1 || exit 1
echo "1024: $(grep 'file/fake' $O/small_1024.txt)"
timeout -k 10 500 python tools/streamed_files_ab.py /dev/shm big 16 2 > $O/big_default.txt 2>&1 || exit 1
echo "big: $(grep 'file/fake' $O/big_default.txt)"
CP2_INGEST_COPY_STREAM=1 timeout -k 10 500 python tools/streamed_files_ab.py /dev/shm big 16 2 > $O/big_copystream.txt 2>&1 || exit 1
echo "big, separate copy stream: $(grep 'file/fake' $O/big_copystream.txt)"
