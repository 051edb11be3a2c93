"""N > 1 host logic of the data-parallel path on CPU: world_size-2 gloo processes exercise game sharding,
the flat-gradient all-reduce and the 3-moment advantage all-reduce of shogidrl_amd.parallel and check that
the 2-rank result equals the 1-rank result on the concatenated batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from shogidrl_amd import parallel


def test_shard_range_partitions_exactly():
    import pytest
    with pytest.raises(ValueError):
        parallel.shard_range(10, 0, 3)                      # the training path insists on equal shards
    assert [parallel.shard_range(4096, r, 8) for r in range(8)] == [(512 * r, 512 * r + 512) for r in range(8)]
    for n, w in ((4096, 8), (10, 3), (7, 8), (32768, 8)):
        spans = [parallel.shard_range(n, r, w, equal=False) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, lr, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.Flatten(), torch.nn.Linear(4 * 9, 2))
    model = model.to(memory_format=torch.channels_last)
    bucket = parallel.FlatGradBucket(model.parameters())
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 3, 3, 3, generator=g)
    y = torch.randn(8, 2, generator=g)
    lo, hi = parallel.shard_range(8, rank, world)
    bucket.zero()
    loss = ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean()
    loss.backward()
    bucket.all_reduce_mean()
    adv = torch.arange(8, dtype=torch.float32) ** 1.5
    n, mean, std = parallel.global_advantage_moments(adv[lo:hi])
    m = parallel.all_reduce_mean_(torch.tensor([float(rank)]))
    parallel.broadcast_parameters(model, src=0)
    q.put((rank, bucket.flat.clone(), float(n), float(mean), float(std), float(m)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_equals_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the concatenated batch
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.Flatten(), torch.nn.Linear(4 * 9, 2))
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 3, 3, 3, generator=g)
    y = torch.randn(8, 2, generator=g)
    loss = ((model(x) - y) ** 2).mean()
    loss.backward()
    want = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    for rank, flat, n, mean, std, m in res:
        # the bucket keeps each gradient in the parameter's own (channels_last) layout: compare as sets of values per tensor
        off = 0
        for p in model.parameters():
            got = flat[off:off + p.numel()].as_strided(p.size(), p.contiguous(memory_format=torch.channels_last).stride()
                                                         if p.dim() == 4 else p.stride())
            assert torch.allclose(got, p.grad, atol=1e-6), rank
            off += p.numel()
        adv = torch.arange(8, dtype=torch.float32) ** 1.5
        assert n == 8 and abs(mean - float(adv.mean())) < 1e-6 and abs(std - float(adv.std())) < 1e-5
        assert m == 0.5
    assert torch.equal(res[0][1], res[1][1])
    assert want.numel() == res[0][1].numel()


def _bn_worker(rank, world, port, q):
    """(1) BatchNorm running statistics advance with each rank's own games and must be re-synchronised;
    (2) ranks holding different batch sizes must fail loudly instead of hanging in the gradient all-reduce."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                                torch.nn.Flatten(), torch.nn.Linear(4 * 9, 2))
    bucket = parallel.FlatGradBucket(model.parameters())
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(100 + rank)           # every rank sees its OWN data
    model.train()
    for _ in range(3):                                      # more than one minibatch per rank
        x = torch.randn(6, 3, 3, 3, generator=g) * (1.0 + rank) + rank
        bucket.begin()
        model(x).square().mean().backward()
        bucket.gather()
        bucket.all_reduce_mean()
        opt.step()
    bn = model[1]
    before = (bn.running_mean.clone(), bn.running_var.clone())
    parallel.sync_buffers(model)
    state = torch.cat([t.detach().reshape(-1).double() for t in model.state_dict().values()])
    raised = False
    try:
        parallel.check_equal_batches(64 + rank, "cpu")      # unequal on purpose
    except ValueError:
        raised = True
    parallel.check_equal_batches(64, "cpu")                 # equal: passes on every rank
    n = lambda t: t.detach().numpy().copy()                 # by value (tensors travel as fds that die with the sender)
    q.put((rank, n(before[0]), n(before[1]), n(bn.running_mean), n(bn.running_var), int(bn.num_batches_tracked),
           n(state), raised))
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_keep_identical_state_dicts():
    """the same worker at world_size 4 (the N = 4 point of the scaling run): every rank ends with the same state dict,
    running statistics = the mean over the four ranks, and the batch-size check raises everywhere"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = [tuple(torch.from_numpy(x) if hasattr(x, "dtype") else x for x in r) for r in res]
    states = [r[6] for r in res]
    assert all(torch.equal(states[0], s) for s in states[1:])
    mean_before = sum(r[1] for r in res) / 4
    assert torch.allclose(res[0][3], mean_before, atol=1e-6)
    assert all(r[7] for r in res) and all(r[5] == 3 for r in res)


def test_two_rank_batchnorm_buffers_are_synchronised_and_unequal_batches_fail_loudly():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = [tuple(torch.from_numpy(x) if hasattr(x, "dtype") else x for x in r) for r in res]
    (_, m0, v0, sm0, sv0, nb0, st0, r0), (_, m1, v1, sm1, sv1, nb1, st1, r1) = res
    assert not torch.allclose(m0, m1)                       # the ranks really had diverged ...
    assert torch.equal(sm0, sm1) and torch.equal(sv0, sv1)  # ... and agree bit for bit afterwards
    assert torch.allclose(sm0, (m0 + m1) / 2) and torch.allclose(sv0, (v0 + v1) / 2)
    assert nb0 == nb1 == 3
    assert torch.equal(st0, st1)                            # whole state dict (parameters + buffers) identical
    assert r0 and r1                                        # the size check raised on BOTH ranks (nobody is left waiting)


def test_one_rank_rehearsal_group_runs_the_collectives():
    """KSH_DIST_REHEARSAL=1: a ONE-rank process group counts as distributed, so a single-GPU job goes through the same
    collectives as a multi-GPU one (how the "nccl" path is exercised on a one-GPU box; here over gloo on the CPU)."""
    import subprocess
    import sys
    code = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from shogidrl_amd import parallel
assert not parallel.collectives_active()
os.environ["KSH_DIST_REHEARSAL"] = "1"
rank, local_rank, world = parallel.init_from_env(backend="gloo")
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1 and parallel.collectives_active()
lin = torch.nn.Linear(5, 3)
b = parallel.FlatGradBucket(list(lin.parameters()))
b.flat.copy_(torch.arange(18, dtype=torch.float32))
b.all_reduce_mean()
assert torch.equal(b.flat, torch.arange(18, dtype=torch.float32))
n, mean, std = parallel.global_advantage_moments(torch.tensor([1.0, 2.0, 3.0, 6.0]))
assert float(n) == 4 and abs(float(mean) - 3.0) < 1e-12 and abs(float(std) - torch.tensor([1.0, 2.0, 3.0, 6.0]).double().std().item()) < 1e-12
parallel.check_equal_batches(7, "cpu")
bn = torch.nn.BatchNorm1d(4); bn.running_mean.fill_(2.0)
parallel.sync_buffers(bn)
assert torch.equal(bn.running_mean, torch.full((4,), 2.0))
parallel.broadcast_parameters(lin)
dist.destroy_process_group()
os.environ["KSH_DIST_REHEARSAL"] = "0"
assert not parallel.collectives_active()
print("REHEARSAL_OK")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KSH_DIST_REHEARSAL"):
        env.pop(k, None)
    env["MASTER_PORT"] = "29561"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "REHEARSAL_OK" in r.stdout, r.stderr[-2000:]
