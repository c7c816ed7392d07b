# round 6: the default turn size of builds whose launches leave room -- 512 (2 waves of 512 workgroups), 768 (3), 1024 (4) MiB -- three rounds, alternating, one box
# (the last sections of profiles/r06_streamed_files_ab.txt)
set -o pipefail
O=gpurun_out/r6exp2
mkdir -p $O
for round in 1 2 3; do for mb in 512 768 1024; do
  CP2_INGEST_CHUNK_MB=$mb timeout -k 10 300 python tools/streamed_files_ab.py /tmp small - 2 > $O/small_${mb}_$round.txt 2>&1 || { tail -5 $O/small_${mb}_$round.txt; exit 1; }
  echo "round $round, $mb MiB: $(grep '^small file/fake' $O/small_${mb}_$round.txt | cut -c1-48)"
done; done
for mb in 512 768; do
  CP2_INGEST_CHUNK_MB=$mb timeout -k 10 500 python tools/streamed_files_ab.py /dev/shm big 16 2 > $O/big_$mb.txt 2>&1 || exit 1
  echo "big, $mb MiB: $(grep '^big   file/fake' $O/big_$mb.txt | cut -c1-48)"
done
