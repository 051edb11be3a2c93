"""End-to-end learning sanity check: the assembled path -- env kernel, fused sampler, rollout buffer, GAE
kernel, fused PPO loss, Adam, driven by SelfPlayRunner -- must change a policy in a way the evaluation loop of SURVEY 8 row f2
can see.  tools/learn_sanity.py trains 60 self-play iterations (1024 games x 64 plies each, ResNet 3x64, the REFERENCE's
hyper-parameters: default_config.yaml:79-116 lr 3e-4, clip 0.2, entropy 0.01, stored old log-probs) from random init and
plays `evaluate_match(policy, UniformRandomPolicy)` (single_opponent.py:170-220 batched, colours balanced, 512 games) before
and after.  Measured on the MI355X (profiles/r03_learn_sanity_reference_hyperparameters_*.log): arg-max win rate 0.14 -> 0.39, max-move draws
224 -> 118 of 512, finished self-play games per iteration 136 -> 235; 46 s.  Runs differ (MIOpen's atomics-based weight
gradients, sampling): the weakest of four runs had arg-max 0.15 -> 0.30, sampled 0.28 -> 0.39, draws 194 -> 157 / 234 -> 131.
The margins below are about half of THAT run's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_play_training_beats_its_untrained_self_against_a_random_opponent():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "learn_sanity.py"), "--iters", "60", "--games", "1024"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    u, t = d["untrained"], d["trained"]
    print("untrained %s\ntrained   %s" % (json.dumps(u), json.dumps(t)))
    assert t["argmax"]["win_rate"] >= u["argmax"]["win_rate"] + 0.08, (u["argmax"], t["argmax"])
    assert t["sampled"]["win_rate"] >= u["sampled"]["win_rate"] + 0.05, (u["sampled"], t["sampled"])
    # games get DECIDED: that is what +1-for-the-mate teaches (both evaluation modes together, 1024 games)
    assert t["argmax"]["draws"] + t["sampled"]["draws"] <= u["argmax"]["draws"] + u["sampled"]["draws"] - 60
    assert d["episodes_per_iter_last_fifth"] >= 1.2 * d["episodes_per_iter_first_fifth"]
    for side in (u, t):
        for mode in ("sampled", "argmax"):
            m = side[mode]
            assert m["wins"] + m["losses"] + m["draws"] == 512
