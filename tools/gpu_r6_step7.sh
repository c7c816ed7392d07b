#!/bin/bash
# round 6: which runtime call holds the building thread?  HIP API trace beside the kernel + copy trace
set -o pipefail
O=gpurun_out/r6s7
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 tools/streamed_files_trace.py run /dev/shm > $O/trace_run.txt 2>&1 || { tail -5 $O/trace_run.txt; exit 1; }
grep -v "^[EW]2026" $O/trace_run.txt | tail -4
find $O/trace -name "*_agent_info.csv" -delete 2>/dev/null
ls -la $O/trace/*/
# the API trace of the whole process is large: keep the calls of the measured builds only (after the warm-ups), and only the long ones + the copies
python3 - <<'PY'
import csv, glob, os
O = "gpurun_out/r6s7"
f = glob.glob(O + "/trace/*/*_hip_api_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), "api rows", rows[0].keys())
keep = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000 or "Memcpy" in r["Function"] or "hipLaunchKernel" == r["Function"]]
w = csv.DictWriter(open(O + "/api_long.csv", "w"), fieldnames=list(rows[0].keys()))
w.writeheader()
w.writerows(keep)
os.remove(f)
print(len(keep), "kept")
PY
