#!/bin/bash
# Runs the given steps one after the other ON THE GPU BOX, each under its own `timeout -k`, logging to gpurun_out/<tag>/<name>.log;
# a step that had to be killed (124 / 137) ends the script: no further GPU step after a kill.
#   bash tools/gpu_steps.sh <tag> "<name>|<seconds>|<command>" ...
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
for step in "$@"; do
  name=${step%%|*}; rest=${step#*|}; secs=${rest%%|*}; cmd=${rest#*|}
  t0=$(date +%s)
  timeout -k 10 $secs bash -c "$cmd" > $O/$name.log 2>&1
  rc=$?
  echo "$name rc=$rc seconds=$(( $(date +%s) - t0 ))" | tee -a $O/status.txt
  tail -4 $O/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name had to be killed: stopping" | tee -a $O/status.txt; exit 1; fi
done
exit 0
