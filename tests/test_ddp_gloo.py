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


@pytest.mark.parametrize("comm,channels_last", [("fp32", "0"), ("bf16", "0"), ("fp32", "1")])
def test_ddp_world_size_2_gloo(comm, channels_last):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", MDX_TEST_GRAD_COMM=comm,
               MDX_TEST_CHANNELS_LAST=channels_last)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DDP_OK" in out.stdout


def test_synchronous_collectives_refuse_stream_capture(monkeypatch):
    """The round-3 core dump: a blocking collective inside torch.cuda.graph hands ProcessGroupNCCL's
    watchdog an event recorded in the capture -> hipErrorCapturedEvent -> std::terminate.  parallel.py's synchronous
    collectives raise in the caller instead (host logic here; on the GPU with a real capture: tests/test_gpu_driver.py)."""
    import importlib
    import sys
    import torch
    sys.path.insert(0, ROOT)
    importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
    from model_tool import parallel
    assert parallel.capturing() is False          # no GPU here / no capture
    monkeypatch.setattr(parallel, "capturing", lambda: True)
    with pytest.raises(RuntimeError, match="synchronous collective"):
        parallel.broadcast_state([torch.nn.Linear(2, 2)])
    monkeypatch.setattr(torch.distributed, "is_initialized", lambda: True)
    with pytest.raises(RuntimeError, match="synchronous collective"):
        parallel.mean_over_ranks([1.0], "cpu")


def test_captured_data_parallel_step_is_opt_in(monkeypatch):
    """With several ranks the exchange goes INSIDE the captured hipGraph only on request (MDX_DP_GRAPH=1); otherwise a captured
    step takes the split form (model_train.graphed_step: the all-reduce issued eagerly between two graphs)."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
    from model_tool import parallel
    monkeypatch.delenv("MDX_DP_GRAPH", raising=False)
    assert parallel.dp_graph_allowed(1) and not parallel.dp_graph_allowed(2) and not parallel.dp_graph_allowed(8)
    monkeypatch.setenv("MDX_DP_GRAPH", "1")
    assert parallel.dp_graph_allowed(8)
