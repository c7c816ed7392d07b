#!/bin/bash
# round 6: the box gives the process 16 CPUs' worth of quota (cgroup cpu.max): fill threads + formatting threads beyond it get the whole process throttled
set -o pipefail
O=gpurun_out/r6s13
mkdir -p $O
for cfg in "8 7" "6 9" "6 8" "4 10" "5 8" "8 4"; do
  set -- $cfg
  CP2_TRACE=1 CP2_INGEST_THREADS=$1 SFAB_THREADS=$2 timeout -k 10 300 python tools/streamed_files_ab.py /dev/shm small - 2 > $O/small_fill$1_json$2.txt 2>&1 || { tail -5 $O/small_fill$1_json$2.txt; exit 1; }
  echo "fill $1 json $2: $(grep 'file/fake' $O/small_fill$1_json$2.txt | cut -c1-50) | $(grep 'file run 2' $O/small_fill$1_json$2.txt | sed 's/.*total/total/' | cut -c1-40) | fake $(grep 'fake run 2' $O/small_fill$1_json$2.txt | sed 's/.*total/total/' | cut -c1-16)"
  grep -E "building thread" $O/small_fill$1_json$2.txt | tail -1 | cut -c1-260
  grep -E "sampling hook over 6" $O/small_fill$1_json$2.txt | tail -1 | cut -c1-200
done
