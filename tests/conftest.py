import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # On a GPU box: start multiprocessing's fork server NOW, while this process has not touched the GPU.  Tests that must start
    # another GPU program (bench.py's multi-rank launcher) ask the server for a child: that child is forked from a GPU-free
    # process, so no process that has initialised the GPU ever forks or execs (see clean_process_run below).
    # The test is "is there a GPU device node", not a torch.cuda call: on ROCm without amdsmi torch.cuda.device_count() falls
    # through to hipGetDeviceCount, which initialises the runtime -- the fork server would then be started by a process that
    # already holds the GPU.
    try:
        if os.path.exists("/dev/kfd") or os.environ.get("CDFO_FORKSERVER", "0") not in ("", "0"):
            import multiprocessing.forkserver as fs
            fs.ensure_running()
    except Exception:
        pass


_PREFETCH = {"wanted": False, "started": False}


def pytest_collection_finish(session):
    """GPU session that includes the full-size parity tests: their CPU-oracle forwards (minutes of host work) run in background
    threads while the GPU tests of the files before them proceed (tests/test_gpu_full_size.py::prefetch).  They are started by
    pytest_runtest_setup below, not here: tests/test_bench_launcher.py runs first and starts bench.py ranks whose own oracle checks
    need the host's cores (with the prefetch beside it that one test took 412 s instead of ~70)."""
    try:
        if os.path.exists("/dev/kfd"):
            wanted = [it for it in session.items if "test_gpu_full_size.py" in it.nodeid and "oracle" in it.nodeid]
            _PREFETCH["wanted"] = len(wanted) >= 3               # a run of the whole file, not one selected case
    except Exception:
        pass


def pytest_runtest_setup(item):
    if _PREFETCH["wanted"] and not _PREFETCH["started"] and "test_bench_launcher.py" not in item.nodeid:
        _PREFETCH["started"] = True
        try:
            import test_gpu_full_size as fs
            fs.prefetch()
        except Exception:
            pass


def _run_and_report(cmd, env, cwd, timeout, q):
    import subprocess
    try:
        r = subprocess.run(cmd, env=env, cwd=cwd, capture_output=True, text=True, timeout=timeout)
        q.put((r.returncode, r.stdout[-40000:], r.stderr[-8000:]))
    except Exception as e:      # noqa: BLE001
        q.put((-999, "", repr(e)))


def clean_process_run(cmd, env=None, cwd=None, timeout=900):
    """subprocess.run(cmd) executed by a child of the fork server (a process that never initialised the GPU):
    returns (returncode, stdout tail, stderr tail)."""
    import multiprocessing as mp
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    p = ctx.Process(target=_run_and_report, args=(cmd, env, cwd, timeout, q))
    p.start()
    try:
        res = q.get(timeout=timeout + 60)
    finally:
        p.join(30)
    return res


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
