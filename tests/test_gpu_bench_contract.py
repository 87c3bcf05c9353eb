"""bench.py's contract with the driver, on a small corpus: exactly ONE JSON line on stdout, the keys the driver and the
judge read, a valid planted-neighbour check, the roofline and cpu_baseline objects."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "300000", "--steps", "3", "--warmup", "1", "--latency-iters", "3",
                        "--cpu-seconds", "0.5", "--cpu-sample-rows", "50000", *extra], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}"
    return json.loads(lines[0])


def test_single_gpu_line():
    line = run_bench()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["results_valid"] is True and line["value"] > 0 and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["launches_timed"] == 3
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_sharded_code_path_line():
    line = run_bench("--force-dist", "--no-cpu-baseline", "--rows", "400000")
    assert line["results_valid"] is True and line["config"]["batches_in_flight"] == 2
    assert line["roofline"]["events_from"].startswith("the same K steps")


def test_driver_command_shape_under_torch_distributed_run():
    """the exact command shape the driver uses for N > 1 — `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` — exercised end to end with N = 1 on the
    one-GPU box: the launcher starts as a fresh child process (before anything in THIS process has touched the GPU for it),
    rank 0 initialises RCCL from the RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* it is handed, takes the sharded code path
    (--force-dist: one all_gather of B*k keys per step, two batches in flight) and prints exactly one line."""
    import socket

    with socket.socket() as s:   # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--rows", "400000", "--steps", "4", "--warmup", "1", "--latency-iters", "3",
           "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["results_valid"] is True and line["value"] > 0
    assert line["config"]["batches_in_flight"] == 2 and line["config"]["rows_per_gpu"] == 400000
    assert line["roofline"]["achieved"] > 0 and "rehearsal" not in line
