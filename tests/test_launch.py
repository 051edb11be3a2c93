"""The one-node launcher behind `bench.py --gpus N` (shogidrl_amd/launch.py), exercised on CPU with gloo ranks:
N asked for = N ranks started; rank 0 owns stdout; a failing rank fails the job; fewer GPUs than ranks is an error."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "helpers", "launch_probe.py")


def _launch(mode, n=2, require_gpus=False, env=None, timeout=180):
    code = ("import sys; sys.path.insert(0, %r); from shogidrl_amd import launch; "
            "sys.exit(launch.spawn_ranks([sys.executable, %r, %r], %d, require_gpus=%r))" % (ROOT, PROBE, mode, n, require_gpus))
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=e)


def test_launcher_starts_the_ranks_it_was_asked_for_and_rank0_owns_stdout():
    r = _launch("ok", n=2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip() and not l.startswith("[Gloo]")]   # (gloo's own C++ banner)
    assert len(lines) == 1, lines                              # exactly ONE line on stdout: rank 0's JSON
    out = json.loads(lines[0])
    assert out == {"world_size": 2, "backend": "gloo", "sum_of_ranks": 1.0, "launched": "1"}
    assert "rank 1 chatter" in r.stderr                        # the other ranks' stdout is diverted to stderr


def test_launcher_fails_the_job_when_one_rank_fails():
    t0 = time.time()
    r = _launch("fail", n=2)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 120                              # rank 0 (sleeping) was stopped, not waited for
    assert "rank 1 exited with code 3" in r.stderr


def test_launcher_refuses_more_ranks_than_gpus():
    """On a box with fewer GPUs than requested ranks the launcher must fail loudly (`--gpus 8` silently ran
    one rank).  This container has no GPU at all, so 2 ranks with require_gpus=True is such a case."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = _launch("ok", n=2, require_gpus=True)
    assert r.returncode == 2 and "refusing to run fewer ranks" in r.stderr
    assert r.stdout.strip() == ""


def test_bench_routes_multi_gpu_requests_through_the_launcher():
    """`python bench.py --gpus 2` without WORLD_SIZE must not fall through to a single-rank run."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=180, env=e)
    assert r.returncode == 2 and "refusing to run fewer ranks" in r.stderr and r.stdout.strip() == ""
