#!/bin/bash
# round 6: does the room cost the slot-file path hash throughput?  the same A/B with the launches refused their room (test hook), and serial order
set -o pipefail
O=gpurun_out/r6s9
mkdir -p $O
CODEX_P2_TEST_LDS_LIMIT=65536 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 3 > $O/small_noroom.txt 2>&1 || { tail -5 $O/small_noroom.txt; exit 1; }
echo "no room (384 MiB chunks): $(grep 'file/fake' $O/small_noroom.txt)"
CODEX_P2_TEST_LDS_LIMIT=65536 CP2_INGEST_CHUNK_MB=768 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/small_noroom_768.txt 2>&1 || exit 1
echo "no room, 768 MiB chunks: $(grep 'file/fake' $O/small_noroom_768.txt)"
CP2_STREAM_SERIAL=1 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/small_serial.txt 2>&1 || exit 1
echo "serial: $(grep 'file/fake' $O/small_serial.txt)"
