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
    assert d.gpus == 1 and d.precision == "fp16x2" and not d.injected_noise and not d.process_group
    assert bench.parse_args(["--process-group"]).process_group
    # the driver reads these keys from the default line: they must be wired into main()
    src = open(bench.__file__).read()
    assert 'res["streaming_b1"] = streaming_b1_line(' in src and 'res["train_step"] = train_step_line(' in src


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


@pytest.mark.gpu
def test_four_rank_rehearsal_over_gloo():
    """The N = 4 control flow on this one-GPU box: four ranks share the GPU (within the box's process limit), gloo
    carries the barriers and the single all_gather; every rank checks c1, nobody runs the full-size oracle clip."""
    import json
    from conftest import clean_process_run
    env = dict(os.environ, CDFO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--batch", "1", "--steps", "1", "--warmup", "1",
           "--height", "64", "--width", "64", "--no-extra-modes", "--no-cpu-baseline", "--no-full-size-parity"]
    rc, out, err = clean_process_run(cmd, env=env, cwd=ROOT, timeout=900)
    assert rc == 0, (rc, out[-2000:], err[-4000:])
    res = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 4 and res["world"] == 4 and res["backend"] == "gloo" and [r["rank"] for r in res["ranks"]] == [0, 1, 2, 3]
    assert res["parity"]["verified"] and len(res["parity"]["per_rank_max_abs"]) == 4


@pytest.mark.gpu
def test_two_rank_bench_line_reports_every_rank():
    """`python bench.py --gpus 2` as a fresh program (its launcher path: the parent starts the ranks before any GPU call), ranks
    sharing this box's GPU with the all_gather over gloo (CDFO_BENCH_BACKEND=gloo; the measured configuration is RCCL, one GPU
    per rank): the JSON line must say n_gpus = 2 and carry one entry per rank (step time, device identity), the backend, and --
    on rank 0 -- the full-size clip-0 parity next to every rank's c1 check.  Started through the GPU-free fork server
    (conftest.clean_process_run): this pytest process has initialised the GPU and must not fork / exec GPU programs."""
    import json
    from conftest import clean_process_run
    env = dict(os.environ, CDFO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--steps", "2", "--warmup", "1",
           "--height", "64", "--width", "96", "--no-extra-modes", "--no-cpu-baseline"]
    rc, out, err = clean_process_run(cmd, env=env, cwd=ROOT, timeout=900)
    assert rc == 0, (rc, out[-2000:], err[-4000:])
    line = [ln for ln in out.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["world"] == 2 and res["backend"] == "gloo"
    assert [r["rank"] for r in res["ranks"]] == [0, 1]
    assert all(r["ms_per_step"] > 0 and r["device"]["compute_units"] > 0 for r in res["ranks"])
    assert res["config"]["clips_per_gpu"] == 2 and res["scaling"] == "weak"
    assert abs(res["value"] - 2 * 2 * 1e3 / res["ms_per_step"]) < 1e-2 * res["value"]          # whole-job rate over the slowest rank
    assert res["ms_per_step"] >= max(r["ms_per_step"] for r in res["ranks"]) - 1e-3
    par = res["parity"]
    assert par["verified"] and len(par["per_rank_max_abs"]) == 2 and all(v <= 1e-3 for v in par["per_rank_max_abs"])
    assert par["full_size_max_abs"] is not None and par["full_size_max_abs"] <= 1e-3
