"""CPU: the multi-GPU path (one process per device, flat gradient buffer + bucketed all-reduce, rank-sharded sampler,
epoch metrics over the job) rehearsed with gloo, world_size 2."""
import os
import subprocess
import sys

import socket

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


import pytest


@pytest.mark.parametrize("comm", ["fp32", "bf16"])
def test_ddp_world_size_2_gloo(comm):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", MDX_TEST_GRAD_COMM=comm)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DDP_OK" in out.stdout
