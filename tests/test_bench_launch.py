"""bench.py's launcher contract: `python bench.py --gpus N` must really run N ranks.

CPU part: with PFMI_BENCH_LAUNCH_ONLY=1 the ranks only join a `gloo` world and count themselves, which exercises the
self-launch path (no WORLD_SIZE in the environment -> bench.py re-executes itself under torch.distributed.run on
127.0.0.1) and the WORLD_SIZE / --gpus consistency check.  The GPU part (tests/test_gpu_*.py::test_bench_*) runs the
real step through RCCL.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def _json_line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert lines, out
    return json.loads(lines[-1])


def test_gpus_2_self_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"PFMI_BENCH_LAUNCH_ONLY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["ranks_in_collective"] == 2


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2"], {"PFMI_BENCH_LAUNCH_ONLY": "1", "WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    r = _run(["--gpus", "1"], {"PFMI_BENCH_LAUNCH_ONLY": "1", "WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "--gpus 1" in r.stderr
