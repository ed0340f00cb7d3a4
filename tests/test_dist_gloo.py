"""The N>1 path (batch sharding + the single metric all_gather) on CPU with the gloo backend, world_size 2."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cdfo_amd.dist import gather_metrics, shard_range, whole_job_rate


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(13, rank, world)
    clips = torch.arange(13)[lo:hi]                       # this rank's shard of the clip batch
    elapsed = 0.5 + 0.25 * rank                          # pretend forward time
    allm = gather_metrics([elapsed, float(len(clips)), float(clips.sum())])
    dist.barrier()
    q.put((rank, allm.tolist()))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1]                               # every rank sees the same gathered table
    table = torch.tensor(res[0])
    assert table[:, 1].sum().item() == 13                 # all clips covered exactly once
    assert table[:, 2].sum().item() == sum(range(13))
    assert abs(whole_job_rate(table[:, 1].tolist(), table[:, 0].tolist()) - 13 / 0.75) < 1e-12


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_gather_without_process_group_is_identity():
    t = gather_metrics([1.0, 2.0])
    assert t.shape == (1, 2) and t[0, 1].item() == 2.0
