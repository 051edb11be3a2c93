"""Host logic of the asynchronous learner on CPU (shogidrl_amd/async_learner.py): the reference's sync rule
(keisei/training/parallel/model_sync.py:50) and the actor / learner role split over a gloo process group --
experience actor -> learner by point-to-point sends, weights learner -> actors by broadcast every `sync_interval`
rounds, actors running ahead with stale weights in between."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_policy_sync_rule_matches_the_reference():
    from shogidrl_amd.async_learner import PolicySync
    s = PolicySync(100)                                   # default_config.yaml:458
    assert not s.should_sync(0) and not s.should_sync(99) and s.should_sync(100) and s.should_sync(250)
    s.mark(100)
    assert not s.should_sync(199) and s.should_sync(200) and s.sync_count == 1
    import pytest
    with pytest.raises(ValueError):
        PolicySync(0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_learners, interval, rounds, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from shogidrl_amd import parallel
    from shogidrl_amd.async_learner import ActorLearnerLoop
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(100 + rank)                          # ranks start DIFFERENT: the constructor's broadcast aligns them
    model = torch.nn.Linear(3, 2)
    template = {"weights_seen": ((1,), torch.float64), "round": ((1,), torch.int64), "payload": ((4, 3), torch.float32)}
    loop = ActorLearnerLoop(model, n_learners, sync_interval=interval, template=template)
    w0 = float(model.weight.detach().double().sum())

    def collect(m, r):                                     # the "rollout": stamps the data with the acting weights
        return {"weights_seen": torch.tensor([float(m.weight.detach().double().sum())], dtype=torch.float64),
                "round": torch.tensor([r]), "payload": torch.full((4, 3), float(rank))}

    def learn(batches, r):                                 # the "update": a deterministic change of the weights
        with torch.no_grad():
            model.weight.add_(1.0)
        return [(float(b["weights_seen"]), int(b["round"]), float(b["payload"].mean())) for b in batches]

    log = []
    for _ in range(rounds):
        out = loop.run_round(collect, learn)
        log.append(out if loop.is_learner else float(out["weights_seen"]))
    q.put((rank, loop.is_learner, w0, log, float(model.weight.detach().double().sum()), loop.sync.sync_count))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n_learners, interval, rounds):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_learners, interval, rounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_actor_and_learner_ranks_exchange_experience_and_stale_weights():
    rounds, interval = 7, 3
    res = _run(2, 1, interval, rounds)
    (_, is_l0, w0, llog, lw, lsync), (_, is_l1, a0, alog, aw, async_count) = res
    assert is_l0 and not is_l1
    assert a0 == w0                                        # the constructor broadcast the learner's initial weights
    step = 6.0                                             # weight.add_(1) on a [2,3] matrix moves the sum by 6
    for r in range(rounds):
        synced_after = (r // interval) * interval          # rounds completed at the last weight push
        want = w0 + step * synced_after                    # the actor acted with weights that old
        assert abs(alog[r] - want) < 1e-4, (r, alog[r], want)
        (seen, rr, payload), = llog[r]                     # and the learner received exactly that batch, in order
        assert rr == r and abs(seen - want) < 1e-4 and payload == 1.0
    assert abs(lw - (w0 + step * rounds)) < 1e-4           # the learner itself never waited: one update per round
    assert lsync == async_count == rounds // interval
    assert abs(aw - (w0 + step * (rounds // interval) * interval)) < 1e-4


def test_two_actors_feed_one_learner():
    res = _run(3, 1, 1, 3)
    learner = res[0]
    for r, got in enumerate(learner[3]):
        assert sorted(p for _, _, p in got) == [1.0, 2.0] and all(rr == r for _, rr, _ in got)
        assert all(abs(seen - (learner[2] + 6.0 * r)) < 1e-4 for seen, _, _ in got)     # interval 1: fresh weights every round
