"""The exact launch line of the multi-GPU bench, exercised end to end on the one GPU a test
box has: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr
127.0.0.1 --master-port P bench.py --gpus 1 ...` as a FRESH child process, `nccl` (= RCCL)
process group, the library's own RCCL communicator moving the packed accumulators
(SPMF_BENCH_COMM=lib).  No N > 1 number exists anywhere in this repo: this covers the
rendezvous, the env parsing, ShardReducer + LibraryComm + spmf_allreduce, the barrier / max
timing and the one-line JSON contract -- not scaling."""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text[-2000:])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("transport", ["lib", "torch"])
def test_torchrun_one_rank_rccl_matches_the_plain_run(transport):
    """transport "lib": the driver's own launch (no SPMF_BENCH_COMM: since round 4 the library's
    communicator, spmf_allreduce, moves the accumulators); "torch": SPMF_BENCH_COMM=torch,
    torch.distributed's RCCL process group."""
    args = ["bench.py", "--gpus", "1", "--workload", "small", "--steps", "3", "--warmup", "1",
            "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ)
    env.pop("SPMF_BENCH_BACKEND", None)
    env.pop("SPMF_BENCH_ONE_GPU", None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    plain = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True,
                           text=True, timeout=280)
    assert plain.returncode == 0, plain.stderr[-2000:]
    ref = _last_json(plain.stdout)
    if transport == "torch":
        env["SPMF_BENCH_COMM"] = "torch"
    else:
        env.pop("SPMF_BENCH_COMM", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _last_json(r.stdout)
    assert out["metric"] == ref["metric"] == "elbo_steps_per_sec"
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["parallelism"] == "row-shard dp1" and out["scaling"] == "strong"
    assert ("library RCCL" if transport == "lib" else "torch.distributed") in out["config"]["allreduce_transport"]
    assert ref["config"]["allreduce_transport"] is None
    assert out["value"] > 0 and math.isfinite(out["ms_per_step"])
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1
    # the same matrix, the same seeded draw: the all-reduced step gives the same energy
    assert out["config"]["nnz"] == ref["config"]["nnz"]
    assert abs(out["elbo_x"] - ref["elbo_x"]) <= 1e-6 * abs(ref["elbo_x"])
    assert out["n_nonfinite"] == 0
