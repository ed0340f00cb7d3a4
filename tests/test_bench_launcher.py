"""bench.py's own N > 1 launch path: `python bench.py --gpus N` outside torchrun must start N ranks --
never silently measure one GPU -- and a WORLD_SIZE that disagrees with --gpus must fail.  CPU only: nothing here
touches a GPU (the launcher itself runs before any GPU call)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_command_starts_one_rank_per_gpu_on_localhost():
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "7", "--warmup", "2"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]      # the ranks see the caller's flags unchanged


def test_args_plumbing():
    a = bench.parse_args(["--gpus", "2", "--steps", "5", "--warmup", "1", "--batch", "4"])
    assert (a.gpus, a.steps, a.warmup, a.batch) == (2, 5, 1, 4)
    d = bench.parse_args([])
    assert d.gpus == 1 and d.precision == "fp16x2" and not d.injected_noise


def test_bare_multi_gpu_invocation_becomes_the_launcher(monkeypatch):
    seen = {}

    def fake_launch(n, argv):
        seen["n"], seen["argv"] = n, list(argv)
        return 0

    monkeypatch.setattr(bench, "launch_ranks", fake_launch)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--batch", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and seen == {"n": 2, "argv": ["--gpus", "2", "--batch", "4"]}


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
