#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/rehearse_bench.sh) -> profiles/<round>_bench_2rank_<case>.json: one record per rehearsal of bench.py's N > 1
path on ONE GPU (2 ranks share GPU 0, gloo collectives) -- clean and with each injected fault: the command, its wall time and exit
code, and the one JSON line it printed (the committed-record blocks the line quotes from profiles/ are dropped).
Usage: rehearsal_records.py <round> <tag> [bench.py commit]"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, TAG = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
O = os.path.join(ROOT, "gpurun_out", TAG)
what = {"clean": "no fault", "rank_exit": "BENCH_INJECT=rank_exit: rank 1 dies (os._exit(3)) at the start of the dataset leg; self-spawned ranks (python bench.py --gpus 2)",
        "rank_hang": "BENCH_INJECT=rank_hang: rank 1 hangs at the start of the dataset leg",
        "gather_error": "BENCH_INJECT=gather_error: rank 1 raises right where the others enter the all-gather of slot roots",
        "child_hang": "BENCH_INJECT=child_hang: the first child process of the in-process leg (RCCL by name) never returns",
        "clean4": "no fault, 4 ranks on the one GPU (python bench.py --gpus 4)",
        "torchrun_clean4": "no fault, 4 ranks on the one GPU under `python -m torch.distributed.run --nproc-per-node 4` (the driver's launcher)",
        "torchrun_rank_exit": "BENCH_INJECT=rank_exit under `python -m torch.distributed.run --nproc-per-node 2` (the driver's launcher): torchrun ends rank 0 with SIGTERM"}
times = {}
for l in open(os.path.join(O, "times.txt")):
    m = re.match(r"(\w+) rc=(\d+) wall_s=([\d.]+)", l)
    if m:
        times[m.group(1)] = (int(m.group(2)), float(m.group(3)))
for case, (rc, wall) in times.items():
    if case == "n1":
        continue
    lines = [l for l in open(os.path.join(O, case + ".json")) if l.startswith("{")]
    line = json.loads(lines[-1]) if lines else None
    if line:
        for k in ("config5_nominal_share_record", "config4_nominal_record"):
            line.get("extra", {}).pop(k, None)
        line.pop("valu_issue", None)
    rec = {"round": R, "case": case, "what": what.get(case, case), "bench_py_commit": commit,
           "setup": "ONE MI355X: BENCH_SHARE_GPU=1 BENCH_BACKEND=gloo (2 ranks on GPU 0, gloo carries the collectives: RCCL refuses two ranks per device); --steps 5 --warmup 2; tools/rehearse_bench.sh",
           "exit_code": rc, "wall_s": wall, "json_lines_on_stdout": len(lines), "line": line}
    four = case.endswith("clean4")
    if four:
        rec["setup"] = rec["setup"].replace("2 ranks on GPU 0", "4 ranks on GPU 0")
    name = "%s_bench_%s_%s.json" % (R, "4rank" if four else "2rank", case.replace("clean4", "clean") if four else ("clean" if case == "clean" else "inject_" + case))
    json.dump(rec, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
    e = (line or {}).get("extra", {})
    print(name, "rc", rc, "wall", wall, "| failures:", e.get("rank_failures"), "| aborted:", (e.get("bench_aborted") or {}).get("reason"))
