"""SURVEY 8(e) parity statement on real kernels: a 2-rank data-parallel PPO update (games sharded, flat-gradient
all-reduce, global advantage moments) equals the 1-rank update on the concatenated batch.  On a one-GPU box the two
ranks share the GPU and the collectives run over gloo (KSH_ALLOW_GPU_SHARING=1); when the box has two GPUs the same
checks run once more over RCCL (backend "nccl"), one rank per GPU (skipped otherwise)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
A = 13527


def _batch(seed, n):
    g = torch.Generator().manual_seed(seed)
    obs = (torch.rand(n, 46, 9, 9, generator=g) < 0.05).float()
    mask = torch.zeros(n, A, dtype=torch.bool)
    actions = torch.zeros(n, dtype=torch.int64)
    for b in range(n):
        ids = torch.randperm(A, generator=g)[:40].sort().values
        mask[b, ids] = True
        actions[b] = ids[int(torch.randint(0, 40, (1,), generator=g))]
    return dict(obs=obs, mask=mask, actions=actions, logp=torch.randn(n, generator=g) * 0.3 - 3.5,
                values=torch.randn(n, generator=g) * 0.5, adv=torch.randn(n, generator=g), ret=torch.randn(n, generator=g))


def _learn(batch, lo, hi, minibatch, group=None, dev="cuda:0", bn=False, epochs=1):
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import ExperienceBuffer
    from shogidrl_amd.models import ResTowerActorCritic, TinyActorCritic
    torch.manual_seed(5)
    model = ResTowerActorCritic(46, A, 1, 16, 0.25) if bn else TinyActorCritic()
    agent = VecPPOAgent(model, PPOConfig(ppo_epochs=epochs, minibatch_size=minibatch, normalize_advantages=True, seed=1),
                        dev, group=group)
    n = hi - lo
    buf = ExperienceBuffer(n, 0.99, 0.95, dev)
    for b in range(lo, hi):
        buf.add(batch["obs"][b], int(batch["actions"][b]), 0.0, float(batch["logp"][b]), float(batch["values"][b]), False,
                batch["mask"][b])
    buf.compute_advantages_and_returns(0.0)
    buf.advantages[:] = batch["adv"][lo:hi].to(dev)
    buf.returns[:] = batch["ret"][lo:hi].to(dev)
    metrics = agent.learn(buf)
    flat = torch.cat([p.detach().reshape(-1) for p in agent.model.parameters()]).cpu()
    if bn:      # the whole state dict: parameters AND BatchNorm buffers
        flat = torch.cat([flat] + [b.detach().reshape(-1).double().cpu().float() for b in agent.model.buffers()])
    return metrics, flat


def _worker(rank, world, port, n, outdir, backend="gloo", bn=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), KSH_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "gloo":
        os.environ["KSH_ALLOW_GPU_SHARING"] = "1"          # both ranks on the one GPU of the test box
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from shogidrl_amd import parallel
    _, local_rank, _ = parallel.init_from_env()
    assert dist.get_backend() == backend and dist.get_world_size() == world
    dev = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)
    batch = _batch(77, n)
    lo, hi = parallel.shard_range(n, rank, world)
    if bn:   # BatchNorm model, several minibatches per rank: replicas must end with identical state dicts
        metrics, flat = _learn(batch, lo, hi, minibatch=(hi - lo) // 3, dev=dev, bn=True, epochs=2)
    else:
        metrics, flat = _learn(batch, lo, hi, minibatch=hi - lo, dev=dev)
    torch.save({"metrics": metrics, "flat": flat, "backend": dist.get_backend(), "device": dev},
               os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(tmp_path, n, backend, bn=False):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, str(tmp_path), backend, bn)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=True) for r in range(2)]


def _check_against_single_rank(r0, r1, n):
    single_metrics, single_flat = _learn(_batch(77, n), 0, n, minibatch=n)
    assert torch.equal(r0["flat"], r1["flat"])                                   # replicas stay in lock-step
    # (Adam divides by sqrt(v): an element whose gradient is pure float32 summation noise moves by up to lr in either run)
    np.testing.assert_allclose(r0["flat"].numpy(), single_flat.numpy(), rtol=0, atol=1e-5)
    for k in ("ppo/policy_loss", "ppo/value_loss", "ppo/entropy", "ppo/kl_divergence_approx", "ppo/clip_fraction"):
        assert abs(r0["metrics"][k] - single_metrics[k]) < 1e-5, k                # metrics are rank-averaged


def test_two_rank_update_equals_single_rank_on_concatenated_batch(tmp_path):
    n = 24
    r0, r1 = _run_two_ranks(tmp_path, n, "gloo")
    _check_against_single_rank(r0, r1, n)


def test_two_rank_batchnorm_model_ends_with_identical_state_dicts(tmp_path):
    """with a BatchNorm tower every rank advances the running statistics with its own games; learn() must
    leave parameters AND buffers identical on all ranks (sync_buffers), also with several minibatches per rank."""
    r0, r1 = _run_two_ranks(tmp_path, 24, "gloo", bn=True)
    assert torch.equal(r0["flat"], r1["flat"])
    assert all(np.isfinite(v) for v in r0["metrics"].values())


def test_two_gpu_rccl_update_equals_single_rank(tmp_path):
    """The same parity check over RCCL ("nccl" backend), one rank per GPU: only runs on a box with >= 2 GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the RCCL path); the one-GPU box covers the same code over gloo")
    n = 24
    r0, r1 = _run_two_ranks(tmp_path, n, "nccl")
    assert r0["backend"] == "nccl" and {r0["device"], r1["device"]} == {"cuda:0", "cuda:1"}
    _check_against_single_rank(r0, r1, n)
    b0, b1 = _run_two_ranks(tmp_path, n, "nccl", bn=True)
    assert torch.equal(b0["flat"], b1["flat"])


def test_bench_launcher_starts_two_ranks_and_reports_them(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment goes through shogidrl_amd/launch.py: two fresh
    ranks, rank 0 prints the JSON line with world_size 2 and the backend.  On a one-GPU box this is a rehearsal (both
    ranks share the GPU, gloo collectives); without the explicit sharing switch the same command must fail loudly."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    small = ["--gpus", "2", "--steps", "1", "--warmup", "0", "--games", "64", "--rollout", "4", "--epochs", "1",
             "--minibatch", "128", "--depth", "1", "--width", "32", "--replay", "4", "--no-cpu-baseline"]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KSH_ALLOW_GPU_SHARING", "KSH_DIST_BACKEND"):
        env.pop(k, None)
    two = torch.cuda.device_count() >= 2
    if not two:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + small, capture_output=True, text=True,
                           timeout=300, env=env)
        assert r.returncode == 2 and "refusing to run fewer ranks" in r.stderr and not r.stdout.strip()
        env.update(KSH_ALLOW_GPU_SHARING="1", KSH_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + small, capture_output=True, text=True, timeout=900,
                       env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["scaling"] == "weak"
    assert out["backend"].startswith("nccl" if two else "gloo")
    assert out["value"] > 0 and out["roofline"]["frac"] > 0


def test_bench_single_rank_line_has_the_contract_keys():
    """`python bench.py` (one rank, no launcher) at toy sizes: ONE JSON line with every key of the driver's contract,
    the roofline object of the env kernel and the CPU baseline object; side figures that do not fit a toy buffer report
    an error string instead of taking the line down."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KSH_ALLOW_GPU_SHARING", "KSH_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--games", "64", "--rollout", "4",
                        "--epochs", "1", "--minibatch", "128", "--depth", "1", "--width", "32", "--replay", "4", "--cpu-seconds", "0.5"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip() == lines[0]
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "world_size", "backend"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 1 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["unit"] == "env-steps/s" and out["value"] > 0 and out["vs_baseline"] is None and out["data"] == "synthetic"
    assert "workload" in out["config"] and "model" not in out["config"]
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and "traffic" in roof
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "env-steps/s" and cb["sample"]
    assert "error" in out["ppo_reference_minibatch64"] or out["ppo_reference_minibatch64"]["value"] > 0


def test_rccl_one_rank_group_accepts_every_collective_the_path_uses():
    """RCCL needs one GPU per rank, so a one-GPU box cannot run the 2-rank test above over "nccl"; what it CAN run is a
    one-rank RCCL communicator: library initialisation on this box plus every collective / dtype / reduce-op combination
    of shogidrl_amd.parallel and bench.py at the bench's sizes (tests/helpers/rccl_single_rank.py)."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(here, "helpers", "rccl_single_rank.py")], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["world_size"] == 1
    for k in ("f32_sum_ok", "f64_sum_ok", "i64_max_ok", "f64_max_ok", "broadcast_ok"):
        assert out[k], (k, out)


def test_bench_one_rank_rccl_rehearsal():
    """KSH_DIST_REHEARSAL=1 python bench.py at toy sizes: a ONE-rank "nccl" (RCCL) process group, and with it the whole
    multi-GPU code path -- parameter broadcast, the flat gradient all-reduce of every minibatch, the advantage moments,
    the equal-batches check, the BatchNorm buffer sync, barriers -- runs over RCCL on the one GPU of this box."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KSH_ALLOW_GPU_SHARING", "KSH_DIST_BACKEND", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(KSH_DIST_REHEARSAL="1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--games", "64", "--rollout", "4",
                        "--epochs", "2", "--minibatch", "128", "--depth", "1", "--width", "32", "--replay", "4", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[:500]      # RCCL's version banner must not reach stdout
    assert "RCCL version" in r.stderr
    out = json.loads(r.stdout)
    assert out["backend"] == "nccl (RCCL)" and out["world_size"] == 1 and out["n_gpus"] == 1 and out["value"] > 0
    assert all(v == v for v in out["ppo_metrics"].values())      # finite metrics after all-reduced updates


def test_bench_under_torch_distributed_run_with_four_ranks_sharing_the_gpu(tmp_path):
    """The driver's command line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- with the REAL job (env kernel, tower, PPO update) at toy size: four
    ranks, rank 0 as supervisor + worker, gloo collectives, every rank with its own MIOpen user-db / kernel-cache directory
    (MIOpen really creates its databases there).  On a box with fewer than four GPUs the ranks share one (explicit switch);
    with four or more they take one each and the collectives run over RCCL."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KSH_ALLOW_GPU_SHARING", "KSH_DIST_BACKEND", "MIOPEN_USER_DB_PATH",
              "MIOPEN_CUSTOM_CACHE_DIR"):
        env.pop(k, None)
    four = torch.cuda.device_count() >= 4
    if not four:
        env.update(KSH_ALLOW_GPU_SHARING="1", KSH_DIST_BACKEND="gloo")
    env["KSH_MIOPEN_DB_BASE"] = str(tmp_path)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1", "--games", "64",
           "--rollout", "4", "--epochs", "1", "--minibatch", "128", "--depth", "1", "--width", "32", "--replay", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["world_size"] == 4 and out["backend"].startswith("nccl" if four else "gloo")
    assert out["value"] > 0 and out["roofline"]["frac"] > 0 and "cpu_baseline" not in out and "worker_exit" not in out
    assert all(v == v for v in out["ppo_metrics"].values())
    made = sorted(d for d in os.listdir(tmp_path) if d.startswith("rank"))
    assert made == ["rank0", "rank1", "rank2", "rank3"]
    assert all(os.listdir(os.path.join(tmp_path, d, "db")) or os.listdir(os.path.join(tmp_path, d, "cache")) for d in made), \
        "MIOpen did not use the per-rank directories"
