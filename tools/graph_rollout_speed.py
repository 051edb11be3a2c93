"""Diagnostic: time of one T-step self-play rollout, eager launches vs replayed from its captured hipGraph, at a launch-bound
size -- to see what DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (which the package requires for every replay, shogidrl_amd/graphs.py)
costs.  Run it twice: as is (the setting in place), and with KSH_PROBE_DEFAULT_PACKET_CAPTURE=1 (this TOOL patches the guard and
leaves the runtime's default packet capture on).
usage: graph_rollout_speed.py [depth width games T]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shogidrl_amd import graphs

if os.environ.get("KSH_PROBE_DEFAULT_PACKET_CAPTURE") == "1":
    graphs.packet_capture_off = lambda: True
else:
    graphs.enable_safe_replay()
import torch

from shogidrl_amd.agent import PPOConfig, VecPPOAgent
from shogidrl_amd.buffer import VecExperienceBuffer
from shogidrl_amd.env import VecShogiEnv
from shogidrl_amd.models import ResTowerActorCritic
from shogidrl_amd.rollout import SelfPlayRunner

depth, width, N, T = (int(x) for x in (sys.argv[1:5] + ["2", "64", "256", "32"][len(sys.argv) - 1:]))
DEV = "cuda:0"
torch.backends.cudnn.benchmark = True


def make():
    torch.manual_seed(0)
    cfg = PPOConfig(seed=11, mixed_precision=True, channels_last=True)
    agent = VecPPOAgent(ResTowerActorCritic(46, 13527, depth, width, 0.25), cfg, DEV)
    return SelfPlayRunner(VecShogiEnv(N, 500, device=DEV), agent, VecExperienceBuffer(T, N, cfg.gamma, cfg.lambda_gae, DEV))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


e, g = make(), make()


def eager():
    e.collect()
    e.buffer.clear()


def graphed():
    g.collect_graphed()
    g.buffer.clear()


ms_e, ms_g = timed(eager), timed(graphed)
print("%s=%s  tower %dx%d, %d games, T=%d: eager %.2f ms per rollout (%.0f k env-steps/s), graph replay %.2f ms (%.0f k env-steps/s)"
      % (graphs.PACKET_CAPTURE_ENV, os.environ.get(graphs.PACKET_CAPTURE_ENV, "<runtime default>"), depth, width, N, T, ms_e,
         N * T / ms_e, ms_g, N * T / ms_g))
