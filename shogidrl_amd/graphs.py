"""One rule for every hipGraph this package replays: the HIP runtime's AQL-packet capture of graph kernel nodes is OFF.

Why.  Replays of the captured PPO-update graph came back corrupted (garbage gradients, twice a GPU memory fault) once some
10^4 eager launches had run on the replay stream between two replays; with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 they stay
correct at the same replay speed (profiles/r02_update_graph_corruption.txt).  The mechanism that fits the evidence --
kernel-argument storage of captured packets recycled by later eager launches -- is not specific to that graph, so the cure
is applied to EVERY replay path, not only the one that was seen to fail: SelfPlayRunner.collect_graphed (rollout graph),
OverlappedSelfPlayLearner(graphed=True), PPOConfig.graph_minibatch_step (update graph).  Each of them calls
`require_safe_replay()` before it captures and refuses to run otherwise.

The runtime reads the variable when it initialises (the first HIP call of the process), so it has to be in the environment
BEFORE that: export it in the shell, or call `enable_safe_replay()` at the top of the program (bench.py does for --graph /
--async-learner; tests/conftest.py does for the GPU test process).  There is no override switch in the product; the
diagnostic probes under tools/ patch `packet_capture_off` themselves when they want to reproduce the corruption.

The reference has no counterpart (its only "validate, then fall back" logic is keisei/utils/compilation_validator.py:103-135
around torch.compile)."""
import os

PACKET_CAPTURE_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"

# what the variable was when this module was first imported (normally with the package, before any HIP call)
_AT_IMPORT = os.environ.get(PACKET_CAPTURE_ENV)
_set_in_time = _AT_IMPORT == "0"


class UnsafeGraphReplay(RuntimeError):
    pass


def _hip_initialised():
    try:
        import torch
        return bool(torch.cuda.is_initialized())
    except Exception:
        return False


def enable_safe_replay():
    """Put DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 into the environment.  Must run before the process makes its first HIP call;
    raises when that is no longer true or when the variable is already set to something else."""
    global _set_in_time
    cur = os.environ.get(PACKET_CAPTURE_ENV)
    if cur == "0" and _set_in_time:
        return
    if cur not in (None, "0"):
        raise UnsafeGraphReplay("%s=%s is set explicitly; hipGraph replay paths of shogidrl_amd need it to be 0"
                                % (PACKET_CAPTURE_ENV, cur))
    if _hip_initialised():
        raise UnsafeGraphReplay("enable_safe_replay() has to run before the first HIP call of the process (the runtime reads "
                                "%s when it initialises)" % PACKET_CAPTURE_ENV)
    os.environ[PACKET_CAPTURE_ENV] = "0"
    _set_in_time = True


def packet_capture_off():
    """True when the setting is in place AND was in place before HIP could have read its flags."""
    return _set_in_time and os.environ.get(PACKET_CAPTURE_ENV) == "0"


def require_safe_replay(what):
    if not packet_capture_off():
        raise UnsafeGraphReplay(
            "%s replays a hipGraph and needs %s=0 in the environment before the process makes its first HIP call "
            "(shogidrl_amd.graphs.enable_safe_replay() at the top of the program, or export it): with the runtime's default "
            "packet capture, replays of a captured graph were corrupted once ~10^4 eager launches had run on the same stream "
            "(profiles/r02_update_graph_corruption.txt)" % (what, PACKET_CAPTURE_ENV))
