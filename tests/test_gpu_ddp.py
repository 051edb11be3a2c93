"""SURVEY 8(e) parity statement on real kernels: a 2-rank data-parallel PPO update (games sharded, flat-gradient
all-reduce, global advantage moments) equals the 1-rank update on the concatenated batch.  The two ranks share the
single GPU of the test box, so the collective runs over gloo here; on a multi-GPU node the same code path uses
RCCL (backend "nccl")."""
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


def _learn(batch, lo, hi, minibatch, group=None):
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    from shogidrl_amd.buffer import ExperienceBuffer
    from shogidrl_amd.models import TinyActorCritic
    torch.manual_seed(5)
    agent = VecPPOAgent(TinyActorCritic(), PPOConfig(ppo_epochs=1, minibatch_size=minibatch, normalize_advantages=True, seed=1),
                        "cuda:0", group=group)
    n = hi - lo
    buf = ExperienceBuffer(n, 0.99, 0.95, "cuda:0")
    for b in range(lo, hi):
        buf.add(batch["obs"][b], int(batch["actions"][b]), 0.0, float(batch["logp"][b]), float(batch["values"][b]), False,
                batch["mask"][b])
    buf.compute_advantages_and_returns(0.0)
    buf.advantages[:] = batch["adv"][lo:hi].cuda()
    buf.returns[:] = batch["ret"][lo:hi].cuda()
    metrics = agent.learn(buf)
    flat = torch.cat([p.detach().reshape(-1) for p in agent.model.parameters()]).cpu()
    return metrics, flat


def _worker(rank, world, port, n, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), KSH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from shogidrl_amd import parallel
    parallel.init_from_env()
    batch = _batch(77, n)
    lo, hi = parallel.shard_range(n, rank, world)
    metrics, flat = _learn(batch, lo, hi, minibatch=hi - lo)
    torch.save({"metrics": metrics, "flat": flat}, os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_update_equals_single_rank_on_concatenated_batch(tmp_path):
    n = 24
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    single_metrics, single_flat = _learn(_batch(77, n), 0, n, minibatch=n)
    r0 = torch.load(os.path.join(str(tmp_path), "rank0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(str(tmp_path), "rank1.pt"), weights_only=True)
    assert torch.equal(r0["flat"], r1["flat"])                                   # replicas stay in lock-step
    np.testing.assert_allclose(r0["flat"].numpy(), single_flat.numpy(), rtol=0, atol=2e-6)
    for k in ("ppo/policy_loss", "ppo/value_loss", "ppo/entropy", "ppo/kl_divergence_approx", "ppo/clip_fraction"):
        assert abs(r0["metrics"][k] - single_metrics[k]) < 1e-5, k                # metrics are rank-averaged
