"""One-off long differential run of the fused policy head (sampler's log-prob, evaluate, loss forward + backward) against
the float64 oracle over a few hundred random cases: batch sizes around every tile / word / segment boundary, K = 162 and
others, dense and empty rows, value clipping, no bias, peaked soft-max rows.  Not part of the test suite (minutes)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests import headcheck
from tests.gpu_backends import GpuOps

ops = GpuOps()
rng = np.random.default_rng(2026)
t0 = time.time()
n = 0
Bs = [1, 2, 31, 32, 33, 63, 64, 65, 70, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 600, 1000]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for B in Bs:
        K = int(rng.choice([162, 162, 162, 10, 64, 190, 192, 194, 320]))
        dense_rows = int(rng.integers(0, 3)) if B >= 3 else 0
        empty_rows = int(rng.integers(0, 2)) if B >= 4 else 0
        dense = bool(dense_rows) or bool(rng.integers(0, 2))
        vc = int(rng.integers(0, 2))
        scale = float(rng.choice([1.0, 1.0, 6.0, 30.0]))
        headcheck.check_head_loss_vs_oracle(ops, B, K, seed=int(rng.integers(1 << 30)), value_clip=vc, dense_rows=dense_rows,
                                            empty_rows=empty_rows, dense=dense, with_bias=bool(rng.integers(0, 2)),
                                            w_scale=scale, greedy_actions=bool(rng.integers(0, 2)))
        n += 1
    headcheck.check_head_evaluate(ops, B=int(rng.choice(Bs[4:])), K=162, seed=int(rng.integers(1 << 30)))
    print("  %d loss cases + %d evaluate cases ok (%.0f s)" % (n, rep + 1, time.time() - t0), flush=True)
print("ALL OK: %d cases" % n)
