"""shogidrl_amd -- MI355X-native (gfx950) hot path of the Keisei Shogi self-play + PPO system.

Only the path BASELINE.json's north_star names lives here: the batched Shogi env kernel (movegen,
legality, termination, 46-plane observation, 13 527-wide legal mask), masked categorical sampling, the
GAE scan and the fused PPO loss, all hand-written HIP behind the C-ABI of include/keisei_hip.h, plus
the thin host-side mirror of the reference's Python interface for that path.
"""
from ._lib import NUM_ACTIONS, OBS_SHAPE, KeiseiHipError  # noqa: F401

from . import ops  # noqa: E402,F401
from .env import DeviceTimer, VecShogiEnv  # noqa: E402,F401

__all__ = ["NUM_ACTIONS", "OBS_SHAPE", "KeiseiHipError", "VecShogiEnv", "DeviceTimer", "ops"]
