"""The env kernel source (shogidrl_amd/csrc/ksh_env_kernel.h) executed on the HOST through the fiber
emulator in tests/emu, checked against the golden vectors and the oracle.  This exercises the kernel's
logic (bitboards, pins, drops, uchi-fu-zume, termination, encoders) without a GPU; the same checks
run against the real HIP library in tests/test_gpu_env.py."""
import numpy as np

from tests import envcheck
from tests.emu.emu import EmuEnv


def make_env(n, max_moves):
    return EmuEnv(n, max_moves)


def test_emu_positions_fixture():
    assert envcheck.check_positions(make_env) > 500


def test_emu_sequences_fixture():
    assert envcheck.check_sequences(make_env) > 80


def test_emu_deep_sequences_sennichite_far_back_and_inert_games():
    """repetitions found 70+ / 520+ plies back in the history (hv[1..7], the `base = 512` loop at max_moves 700) and
    steps on finished games -- every output as the reference produced it (sequences_deep.json)"""
    assert envcheck.check_sequences(make_env, deep=True) > 600


def test_emu_traces_fixture():
    # 16 of the 48 golden games here (the GPU suite replays all of them through the C-ABI)
    assert envcheck.check_traces(make_env, max_games=16) > 4500


def test_emu_traces_no_autoreset_terminal_obs():
    assert envcheck.check_traces(make_env, max_games=6, auto_reset=False) > 1000


def test_emu_random_vs_oracle_drop_heavy():
    envcheck.check_random_vs_oracle(make_env, n_games=24, n_steps=220, seed=5, drop_bias=0.5)


def test_emu_random_short_games_many_resets():
    ndone = envcheck.check_random_vs_oracle(make_env, n_games=16, n_steps=150, seed=9, max_moves=40)
    assert ndone >= 16 * 3


def test_emu_obs_rows_at_every_alignment_phase():
    import pytest
    from tests.emu.emu import EmuEnv
    for off in (1, 2, 3):
        envcheck.check_random_vs_oracle(lambda n, mm, off=off: EmuEnv(n, mm, obs_offset=off), n_games=5, n_steps=30, seed=off)


def test_emu_illegal_actions_rejected():
    envcheck.check_illegal_actions(make_env)


def test_emu_packed_only_masks_random_vs_oracle():
    """byte-mask pointer NULL: the kernel writes only the 1 696-byte packed mask and validates actions against packed bits"""
    def mk(n, mm):
        e = EmuEnv(n, mm)
        e.packed_only = True
        return e
    envcheck.check_random_vs_oracle(mk, n_games=9, n_steps=60, seed=4, drop_bias=0.4)
    envcheck.check_illegal_actions(mk)


def test_emu_compact_bf16_channels_last_obs_vs_oracle():
    """observation planes written as bfloat16, channels last -- bit-exact round-to-nearest-even of the
    oracle's float32 planes, at every 16-byte alignment phase of the 7 452-byte rows, nothing stored outside them."""
    from tests.emu.emu import from_bf16_bits, to_bf16_bits

    def mk(off):
        def f(n, mm):
            e = EmuEnv(n, mm, obs_offset=off)
            e.obs16 = True
            e.obs_round = lambda x: from_bf16_bits(to_bf16_bits(x))
            return e
        return f
    ndone = envcheck.check_random_vs_oracle(mk(0), n_games=12, n_steps=120, seed=21, max_moves=60, drop_bias=0.4)
    assert ndone >= 12
    for off in (1, 2, 3):
        envcheck.check_random_vs_oracle(mk(off), n_games=5, n_steps=24, seed=30 + off)
