"""One PROCESS per rank (the production shape), ranks talking through a real out-of-process transport:
torch.distributed.run starts the ranks, gloo carries the library's collectives as host buffers
(alfd_comm_init_host), every rank computes on GPU 0 of the one-GPU box.  Counts and history must equal the
oracle's emulation of the same partition.  With >= 2 GPUs visible the same solve also runs over RCCL
(bench.py --gpus 2), otherwise that test is skipped: RCCL refuses two ranks on one device."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(nproc, script_args, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


@pytest.mark.parametrize("world,mode", [(2, "chebyshev"), (3, "multilevel"), (2, "geometric")])
def test_partitioned_solve_in_separate_processes_over_gloo(built, world, mode):
    p = _launch(world, [os.path.join(ROOT, "tests", "mp_worker.py"), mode])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("MP_RESULT ")][-1]
    out = json.loads(line[len("MP_RESULT "):])
    assert out["ok"] and out["world"] == world and out["outer"] > 0


def test_bench_starts_its_own_ranks(built):
    """`python bench.py --gpus 2` WITHOUT a launcher (how a driver may call it): bench.py starts the two ranks as child
    processes before touching the GPU and relays rank 0's line.  One-GPU box: both ranks on device 0, host transport."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", ALFD_BENCH_SINGLE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "host", "--n-cells", "12",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--general-steps", "0"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["outer_iterations_per_solve"] > 0
    assert out["config"]["transport"] == "host buffers over gloo"


def test_two_gpu_rccl_bench_matches_single_gpu_counts(built):
    """The literal RCCL path (ncclAllGather / grouped ncclSend+ncclRecv) needs two devices."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("fewer than 2 GPUs visible: RCCL cannot place two ranks on one device")
    args = ["--n-cells", "16", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--hierarchy", "aggregation",
            "--general-steps", "0"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, capture_output=True,
                         text=True, timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    two = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args)
    assert two.returncode == 0, two.stderr[-3000:]
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    j2 = json.loads(two.stdout.strip().splitlines()[-1])
    assert j2["n_gpus"] == 2
    assert j2["config"]["outer_iterations_per_solve"] == j1["config"]["outer_iterations_per_solve"]
    assert abs(j2["config"]["final_residual"] - j1["config"]["final_residual"]) <= 1e-6 * j1["config"]["initial_residual"]
