#!/bin/bash
# round 6: the slot-file tests again (mapped turns of many files are new), then the small shape: ring, mapped, chunk of two residencies
set -o pipefail
O=gpurun_out/r6s4
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -q -m gpu -k "round6 or rccl_by_name or mapping" > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 3 > $O/small_ring.txt 2>&1 || { tail -5 $O/small_ring.txt; exit 1; }
echo "ring: $(grep 'file/fake' $O/small_ring.txt)"
CP2_INGEST_MAPPED=1 CP2_TRACE=1 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 3 > $O/small_mapped.txt 2>&1 || { tail -5 $O/small_mapped.txt; exit 1; }
echo "mapped: $(grep 'file/fake' $O/small_mapped.txt)"
CP2_INGEST_CHUNK_MB=512 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/small_ring_512.txt 2>&1 || exit 1
echo "ring, 512 MiB chunks: $(grep 'file/fake' $O/small_ring_512.txt)"
CP2_INGEST_MAPPED=1 timeout -k 10 500 python tools/streamed_files_ab.py /dev/shm big 16 2 > $O/big_mapped.txt 2>&1 || exit 1
echo "big mapped: $(grep 'file/fake' $O/big_mapped.txt)"
timeout -k 10 500 python tools/streamed_files_ab.py /dev/shm big 16 2 > $O/big_ring.txt 2>&1 || exit 1
echo "big ring: $(grep 'file/fake' $O/big_ring.txt)"
