"""One-off long differential run: the HIP env vs the oracle over hundreds of thousands of env steps (uniform and
drop-heavy random self-play, several max_moves). Not part of the test suite (takes minutes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np
import torch
from oracle import ko
from shogidrl_amd import ops
from shogidrl_amd.env import VecShogiEnv

COMPACT = len(sys.argv) > 1 and sys.argv[1] == "compact"     # packed masks + bf16 channels-last observations


def run(n, steps, max_moves, drop_bias, seed):
    env = VecShogiEnv(n, max_moves, packed_masks=True, obs_format="bf16_nhwc") if COMPACT else VecShogiEnv(n, max_moves)
    ora = ko.VecGames(n, max_moves)
    env.reset()
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    stats = np.zeros(6, np.int64)
    t0 = time.time()
    for k in range(steps):
        m = ops.unpack_mask(env.legal_bits) if COMPACT else env.legal_mask
        w = torch.rand(m.shape, device="cuda:0", generator=g) * m
        if drop_bias > 0:
            boost = (torch.rand(n, 1, device="cuda:0", generator=g) < drop_bias).float()
            w[:, 12960:] += boost * m[:, 12960:]
        a = w.argmax(dim=1)
        env.step(a)
        o = ora.step(a.cpu().numpy())
        if COMPACT:
            assert env.obs.is_contiguous(memory_format=torch.channels_last)
            assert torch.equal(env.obs.cpu(), torch.from_numpy(o[0]).to(torch.bfloat16)), ("obs", k)
            assert np.array_equal(ops.unpack_mask(env.legal_bits).cpu().numpy().astype(np.uint8), o[1]), ("mask", k)
        else:
            assert np.array_equal(env.obs.cpu().numpy(), o[0]), ("obs", k)
            assert np.array_equal(env.legal_mask.cpu().numpy().astype(np.uint8), o[1]), ("mask", k)
            assert torch.equal(ops.unpack_mask(env.legal_bits), env.legal_mask), ("packed twin of the byte mask", k)
        assert np.array_equal(env.reward.cpu().numpy(), o[3]) and np.array_equal(env.done.cpu().numpy().astype(np.uint8), o[4]), ("done", k)
        assert np.array_equal(env.reason.cpu().numpy(), o[5]) and np.array_equal(env.winner.cpu().numpy(), o[6]), ("reason", k)
        for r in range(1, 6):
            stats[r] += int((o[5][o[4] != 0] == r).sum())
        if k % 100 == 99:
            print("  %d x %d steps ok (%.0f s) ends: tsumi %d stalemate %d maxmoves %d sennichite %d" %
                  (n, k + 1, time.time() - t0, stats[1], stats[2], stats[3], stats[4]), flush=True)
    return stats

print("attack tables: %s" % os.environ.get("KSH_ENV_TABLES", "library default (LDS copy below 65 536 games)"), flush=True)
tot = np.zeros(6, np.int64)
for (n, steps, mm, db, seed) in [(768, 620, 500, 0.0, 1), (512, 400, 500, 0.6, 2), (512, 300, 90, 0.3, 3)]:
    print("run n=%d steps=%d max_moves=%d drop_bias=%.1f" % (n, steps, mm, db), flush=True)
    tot += run(n, steps, mm, db, seed)
print("ALL OK; game ends by reason:", tot.tolist())
