"""bench.py's process structure and the multi-GPU dress rehearsal, on CPU (tests/helpers/bench_rehearsal.py = bench.py with
the rank's GPU job replaced by a stand-in; everything around it -- launcher, supervisor / worker / stage file, process
group, collectives, timing, teardown -- is bench.py's own code):

  * the ONE JSON line survives a worker that dies after the timed region (the way a GPU fault kills a process: abort, no
    exception), and a process-group teardown that never returns;
  * `--gpus 8` through the package's launcher and through torch.distributed.run (the driver's way): 8 fresh ranks over gloo,
    rank 0 prints, every rank has its own MIOpen user-db directory.
The reference has no counterpart (single process, `ddp` flag dead: keisei/config_schema.py:81)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "helpers", "bench_rehearsal.py")
TOY = ["--steps", "2", "--warmup", "1", "--games", "8", "--rollout", "4", "--epochs", "2", "--minibatch", "16"]


def _env(**extra):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KSH_DIST_REHEARSAL", "KSH_BENCH_STAGES",
              "KSH_BENCH_TEST_ABORT_AT", "MIOPEN_USER_DB_PATH", "MIOPEN_CUSTOM_CACHE_DIR"):
        e.pop(k, None)
    e.update(KSH_DIST_BACKEND="gloo")
    e.update(extra)
    return e


def _run(args, timeout=300, **extra):
    return subprocess.run([sys.executable, HELPER] + args, capture_output=True, text=True, timeout=timeout, env=_env(**extra))


def _one_line(r):
    lines = [l for l in r.stdout.splitlines() if l.strip() and not l.startswith("[Gloo]")]
    assert len(lines) == 1, (lines, r.stderr[-2000:])
    return json.loads(lines[0])


def test_single_rank_line_comes_from_the_supervisor_and_has_every_part():
    r = _run(TOY)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _one_line(r)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "world_size", "backend", "warmup_s", "first_warmup_step_s",
              "roofline_8192", "fused_policy_head"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and "worker_exit" not in out
    assert out["config"]["old_log_probs"].startswith("as stored")      # the reference's semantics are the default
    assert out["rehearsal_stand_in"] is True                          # a line from the stand-in can never pass for a measurement


def test_a_worker_that_dies_in_a_side_figure_cannot_remove_the_headline():
    """abort (no exception, no cleanup: what a GPU memory fault does to a process) just before the fused-head figure: value,
    roofline and cpu_baseline were already in the stage file, so the line is complete and the run counts"""
    r = _run(TOY, KSH_BENCH_TEST_ABORT_AT="fused_policy_head")
    out = _one_line(r)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["value"] > 0 and out["roofline"]["bound"] == "hbm" and out["cpu_baseline"]["kind"] == "port"
    assert "fused_policy_head" not in out
    assert out["worker_exit"]["last_stage"] == "roofline_8192" and out["worker_exit"]["code"] != 0


def test_a_worker_that_dies_before_the_roofline_still_reports_value_but_fails_the_run():
    r = _run(TOY, KSH_BENCH_TEST_ABORT_AT="roofline")
    out = _one_line(r)
    assert r.returncode != 0
    assert out["value"] > 0 and "roofline" not in out and out["worker_exit"]["last_stage"] == "headline"


def test_a_worker_that_dies_inside_the_timed_region_prints_nothing():
    r = _run(["--gpus", "2"] + TOY, KSH_ALLOW_GPU_SHARING="1", KSH_REHEARSAL_FAIL_RANK="1")
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_a_hung_teardown_cannot_eat_the_line():
    """every figure reported, then the worker never comes back from its final barrier (a late RCCL teardown hang): the
    supervisor prints after its grace period and stops the worker"""
    r = _run(TOY, KSH_DIST_REHEARSAL="1", MASTER_PORT=str(_free_port()), KSH_REHEARSAL_HANG_TEARDOWN="1",
             KSH_BENCH_TEARDOWN_GRACE_S="3")
    out = _one_line(r)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out["value"] > 0 and out["world_size"] == 1 and out["backend"] == "gloo"
    assert out["worker_exit"]["last_stage"] == "complete" and "teardown" in out["worker_exit"]["note"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _check_eight(out, tmp):
    assert out["n_gpus"] == 8 and out["world_size"] == 8 and out["backend"] == "gloo" and out["scaling"] == "weak"
    assert out["value"] > 0 and out["roofline"]["bound"] == "hbm" and "cpu_baseline" not in out      # (rank 0 at N = 1 only)
    assert all(v == v for v in out["ppo_metrics"].values())
    dirs = out["miopen_user_db_paths"]
    assert len(set(dirs)) == 8 and all(d and d.startswith(str(tmp)) and os.path.isdir(d) for d in dirs)
    # whole-job value = units of ALL ranks / max-over-ranks time
    assert abs(out["value"] - 8 * 2 * 4 * 8 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]


def test_eight_rank_rehearsal_through_the_launcher(tmp_path):
    """`bench.py --gpus 8` with no WORLD_SIZE in the environment: the package's launcher starts 8 fresh ranks"""
    r = _run(["--gpus", "8"] + TOY, timeout=600, KSH_ALLOW_GPU_SHARING="1", KSH_MIOPEN_DB_BASE=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    _check_eight(_one_line(r), tmp_path)


def test_eight_rank_rehearsal_through_torch_distributed_run(tmp_path):
    """the driver's command line: python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 ..."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), HELPER, "--gpus", "8"] + TOY
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                       env=_env(KSH_ALLOW_GPU_SHARING="1", KSH_MIOPEN_DB_BASE=str(tmp_path)))
    assert r.returncode == 0, r.stderr[-3000:]
    _check_eight(_one_line(r), tmp_path)


def test_launcher_counts_gpus_in_a_child_process():
    """launch.visible_gpus() must not be able to initialise HIP in the parent: it asks a short-lived child"""
    import inspect
    from shogidrl_amd import launch
    src = inspect.getsource(launch.visible_gpus)
    assert "subprocess.run" in src and "import torch" in src and "torch.cuda" not in src.replace("torch.cuda.device_count()", "")
    import torch
    n = launch.visible_gpus()
    assert n == torch.cuda.device_count()                  # (None would mean "could not count": not a refusal, see spawn_ranks)
    assert not torch.cuda.is_initialized()


def test_every_graph_replay_site_requires_the_runtime_setting():
    """no hipGraph replay anywhere in the package without graphs.require_safe_replay"""
    pkg = os.path.join(ROOT, "shogidrl_amd")
    sites = []
    for name in sorted(os.listdir(pkg)):
        if name.endswith(".py") and name != "graphs.py":
            src = open(os.path.join(pkg, name)).read()
            if "CUDAGraph(" in src or ".replay()" in src:
                sites.append(name)
                assert "require_safe_replay(" in src, name
    assert set(sites) >= {"agent.py", "rollout.py"}
    assert "require_safe_replay(" in open(os.path.join(pkg, "async_learner.py")).read()


def test_graph_guard_refuses_and_enables(monkeypatch):
    from shogidrl_amd import graphs
    monkeypatch.setattr(graphs, "_set_in_time", False)
    monkeypatch.delenv(graphs.PACKET_CAPTURE_ENV, raising=False)
    with pytest.raises(graphs.UnsafeGraphReplay, match=graphs.PACKET_CAPTURE_ENV):
        graphs.require_safe_replay("test")
    monkeypatch.setenv(graphs.PACKET_CAPTURE_ENV, "1")
    with pytest.raises(graphs.UnsafeGraphReplay):
        graphs.enable_safe_replay()                     # an explicit other value is not overridden
    monkeypatch.delenv(graphs.PACKET_CAPTURE_ENV, raising=False)
    graphs.enable_safe_replay()                          # (no HIP call has been made in this CPU process)
    assert os.environ[graphs.PACKET_CAPTURE_ENV] == "0" and graphs.packet_capture_off()
    graphs.require_safe_replay("test")
    # setting the variable by hand AFTER the runtime may have read its flags does not count
    monkeypatch.setattr(graphs, "_set_in_time", False)
    assert not graphs.packet_capture_off()


def test_cpu_baseline_uses_every_core_of_the_affinity_mask_and_is_bounded_by_wall_time():
    """`cpu_baseline.all_cores.cores` = len(os.sched_getaffinity(0)), not a cap; the sample is bounded by
    wall time whatever the cgroup grants (a fixed step count took 190 s on a 256-thread / 16-CPU box)."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    t0 = time.time()
    cb = bench.cpu_baseline(0.5)
    assert time.time() - t0 < 20
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "NOT with `value`" in cb["sample"]
    ac = cb["all_cores"]
    assert ac["cores"] == len(os.sched_getaffinity(0)) == ac["usable_cores"] and ac["host_cpu_count"] == os.cpu_count()
    assert ac["value"] > 0 and "cgroup_cpu_limit" in ac
