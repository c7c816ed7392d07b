#!/bin/bash
# round 6, at scale on the final library: transient (compact / roots-only) streamed builds FROM FILES, long soaks, the nominal end-to-end run
set -o pipefail
O=gpurun_out/r6scale
mkdir -p $O
for keep in 2 0; do
  SFAB_KEEP=$keep timeout -k 10 500 python tools/streamed_files_ab.py /dev/shm big 16 2 > $O/big_keep$keep.txt 2>&1 || { tail -5 $O/big_keep$keep.txt; exit 1; }
  echo "16 x 8 GiB from files, keep=$keep: $(grep 'file/fake' $O/big_keep$keep.txt) | $(grep 'file run 2' $O/big_keep$keep.txt | cut -c1-120)"
  SFAB_KEEP=$keep timeout -k 10 300 python tools/streamed_files_ab.py /tmp small - 2 > $O/small_keep$keep.txt 2>&1 || { tail -5 $O/small_keep$keep.txt; exit 1; }
  echo "4096 x 8 MiB from files, keep=$keep: $(grep 'file/fake' $O/small_keep$keep.txt)"
done
timeout -k 10 1000 python tools/soak_multi.py 600 611 > $O/soak_multi_long.txt 2>&1 || { tail -20 $O/soak_multi_long.txt; exit 1; }
tail -1 $O/soak_multi_long.txt
timeout -k 10 700 python tools/soak_pipeline.py 420 612 > $O/soak_pipeline_long.txt 2>&1 || { tail -20 $O/soak_pipeline_long.txt; exit 1; }
tail -1 $O/soak_pipeline_long.txt
