# experiment (profiles/r06_lean_tail_kernels_experiment.txt): tail kernels (k_compress_layer, k_sample_paths) capped at 80 VGPRs so that they fit beside
# THREE hash waves per SIMD -- is the room still needed?  build/libcodex_p2_lean.so = this library built from a copy of the package whose csrc/kernels.hip has
# `__attribute__((amdgpu_waves_per_eu(6, 6)))` added to those two kernels (hipcc -Rpass-analysis=kernel-resource-usage: 72 VGPRs / 80 with 14 spilled).
set -o pipefail
O=gpurun_out/r6exp3
mkdir -p $O
LEAN=$PWD/build/libcodex_p2_lean.so
for round in 1 2; do
  timeout -k 10 300 python tools/streamed_files_ab.py /tmp small - 2 > $O/default_$round.txt 2>&1 || exit 1
  echo "default library, room: $(python3 -c "
import json
d=json.loads([l for l in open('$O/default_$round.txt') if l.startswith('{')][-1])['small']; print(d['best_total_s'], d['plain_tree_build_s'])")"
  CODEX_P2_LIB=$LEAN timeout -k 10 300 python tools/streamed_files_ab.py /tmp small - 2 > $O/lean_room_$round.txt 2>&1 || exit 1
  echo "lean tail kernels, room: $(python3 -c "
import json
d=json.loads([l for l in open('$O/lean_room_$round.txt') if l.startswith('{')][-1])['small']; print(d['best_total_s'], d['plain_tree_build_s'])")"
  CODEX_P2_LIB=$LEAN CODEX_P2_TEST_LDS_LIMIT=65536 timeout -k 10 300 python tools/streamed_files_ab.py /tmp small - 2 > $O/lean_noroom_$round.txt 2>&1 || exit 1
  echo "lean tail kernels, NO room: $(python3 -c "
import json
d=json.loads([l for l in open('$O/lean_noroom_$round.txt') if l.startswith('{')][-1])['small']; print(d['best_total_s'], d['plain_tree_build_s'])")"
  CODEX_P2_TEST_LDS_LIMIT=65536 timeout -k 10 300 python tools/streamed_files_ab.py /tmp small - 2 > $O/default_noroom_$round.txt 2>&1 || exit 1
  echo "default library, NO room: $(python3 -c "
import json
d=json.loads([l for l in open('$O/default_noroom_$round.txt') if l.startswith('{')][-1])['small']; print(d['best_total_s'], d['plain_tree_build_s'])")"
done
