#!/bin/bash
# Rehearses bench.py's N > 1 path on ONE GPU (what this pipeline's boxes have): 2 ranks share GPU 0, gloo carries the collectives
# (RCCL refuses two ranks per device), once clean and once per injected fault.  Run ON THE GPU BOX from the repo root:
#   bash tools/rehearse_bench.sh r5a [case ...]      cases: n1 clean rank_exit rank_hang gather_error child_hang torchrun_rank_exit clean4 torchrun_clean4 (4 ranks on the one GPU: 5 processes on the card with a child leg, the box allows 6)
# Every case runs under its own `timeout -k`; a case that had to be killed ends the script (no further GPU step after a kill).
# Output: gpurun_out/<tag>/<case>.json (stdout: the one JSON line), .err (stderr), times.txt (wall seconds, exit code per case).
TAG=${1:-r5a}; shift
CASES=${@:-n1 clean rank_exit rank_hang gather_error child_hang torchrun_rank_exit}
O=gpurun_out/$TAG
mkdir -p $O
export BENCH_BACKEND=gloo BENCH_SHARE_GPU=1
ARGS="--steps 5 --warmup 2"
for c in $CASES; do
  t0=$(date +%s.%N)
  case $c in
    n1)        env -u BENCH_BACKEND -u BENCH_SHARE_GPU timeout -k 10 420 python3 bench.py $ARGS > $O/$c.json 2> $O/$c.err ;;
    clean)     timeout -k 10 420 python3 bench.py --gpus 2 $ARGS > $O/$c.json 2> $O/$c.err ;;
    clean4)    timeout -k 10 420 python3 bench.py --gpus 4 $ARGS > $O/$c.json 2> $O/$c.err ;;
    torchrun_clean4)
               timeout -k 10 420 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29534 \
                 bench.py --gpus 4 $ARGS > $O/$c.json 2> $O/$c.err ;;
    torchrun_rank_exit)
               BENCH_INJECT=rank_exit timeout -k 10 420 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
                 bench.py --gpus 2 $ARGS > $O/$c.json 2> $O/$c.err ;;
    *)         BENCH_INJECT=$c timeout -k 10 420 python3 bench.py --gpus 2 $ARGS > $O/$c.json 2> $O/$c.err ;;
  esac
  rc=$?
  t1=$(date +%s.%N)
  echo "$c rc=$rc wall_s=$(python3 -c "print(round($t1-$t0,1))") line_bytes=$(grep -c '^{' $O/$c.json)" | tee -a $O/times.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$c had to be killed: stopping" | tee -a $O/times.txt; exit 1; fi
done
exit 0
