"""Starts the rank processes of the sharded product path (tests/shard_rank_child.py) and collects what they report."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(world, cfg, entropy, tmp_path, roots_path=None, timeout=900, gather="dev"):
    """`world` fresh processes on GPU 0 (gloo collectives: RCCL refuses two ranks per device); returns their result dicts.
    gather: "dev" = the roots stay on the device on both sides of the exchange, "host" = host arrays (round 3's path)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(world), CP2_TEST_GATHER=gather)
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / ("rank%d_of%d.json" % (r, world)))
        outs.append(out)
        argv = [sys.executable, os.path.join(ROOT, "tests", "shard_rank_child.py"), out, "0", json.dumps(cfg), str(entropy)]
        if roots_path:
            argv.append(str(roots_path))
        procs.append(subprocess.Popen(argv, env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    for p, out in zip(procs, outs):
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, se[-3000:]
        res.append(json.load(open(out)))
    return res
