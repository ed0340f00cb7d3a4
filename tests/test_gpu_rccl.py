"""The one collective of the multi-GPU path -- an all_gather of per-rank metrics over RCCL (backend "nccl"), SURVEY section 8e --
on the hardware a one-GPU box has: a ONE-rank RCCL communicator.  (i) in this process: init_process_group("nccl", device_id=...)
-> cdfo_amd.dist.gather_metrics(values, device) -> barrier -> destroy; (ii) `torchrun --nproc-per-node 1 bench.py --gpus 1`, the
driver's launch line at N = 1: bench.py then keeps the process group alive, takes the N > 1 path's barriers and all_gather
over RCCL and prints "backend": "nccl".  (ii) starts through the GPU-free fork server (conftest.clean_process_run)."""
import json
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_metric_all_gather_over_rccl_world_1():
    import torch.distributed as dist
    from cdfo_amd.dist import gather_metrics
    assert not dist.is_initialized()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        vals = [0.125, 3.0, -7.5, 1e-3]
        table = gather_metrics(vals, dev)                   # all_gather on the device through RCCL
        dist.barrier()
        torch.cuda.synchronize()
        assert table.shape == (1, 4) and table.dtype == torch.float64 and table[0].tolist() == vals
        x = torch.arange(8, dtype=torch.float32, device=dev)
        dist.all_reduce(x)                                   # a second collective on the same communicator
        assert x.cpu().tolist() == list(range(8))
    finally:
        dist.destroy_process_group()
    assert any("librccl" in ln for ln in open("/proc/self/maps")), "RCCL is not mapped into this process"


def test_bench_under_torchrun_with_one_rank_runs_over_rccl():
    from conftest import clean_process_run
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CDFO_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "2", "--steps", "2",
           "--warmup", "1", "--height", "64", "--width", "96", "--no-extra-modes", "--no-cpu-baseline"]
    rc, out, err = clean_process_run(cmd, env=env, cwd=ROOT, timeout=600)
    assert rc == 0, (rc, out[-2000:], err[-4000:])
    res = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["world"] == 1 and res["backend"] == "nccl"
    assert res["distinct_devices"] == 1 and len(res["ranks"]) == 1 and res["ranks"][0]["device"]["compute_units"] > 0
    assert res["parity"]["verified"] and res["parity"]["max_abs"] <= 1e-3
