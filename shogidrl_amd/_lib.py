"""ctypes binding of libkeisei_hip.so (include/keisei_hip.h).  There is NO fallback: if the HIP
library is missing or fails to load, importing anything that computes raises."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# SHOGIDRL_AMD_LIB lets a developer point at an alternative build of the SAME library (diagnostic / A-B builds)
LIB_PATH = os.environ.get("SHOGIDRL_AMD_LIB") or os.path.join(PKG, "libkeisei_hip.so")

NUM_ACTIONS = 13527
MASK_WORDS = 424     # int32 words per row of a bit-packed legal mask (include/keisei_hip.h KSH_MASK_WORDS)
OBS_SHAPE = (46, 9, 9)

_lib = None

# every symbol include/keisei_hip.h declares (tests check the .so exports each of them)
SYMBOLS = [
    "ksh_last_error", "ksh_version", "ksh_env_create", "ksh_env_destroy", "ksh_env_num_games", "ksh_env_max_moves",
    "ksh_env_set_obs_format", "ksh_env_obs_format", "ksh_env_set_table_mode", "ksh_env_table_mode",
    "ksh_env_reset", "ksh_env_observe", "ksh_env_evaluate_termination", "ksh_env_step", "ksh_env_reset_packed", "ksh_env_step_packed", "ksh_env_set_state", "ksh_env_get_state", "ksh_masked_sample", "ksh_masked_sample_packed", "ksh_masked_sample_graph", "ksh_ppo_loss_fwd_bwd_packed",
    "ksh_evaluate_actions", "ksh_evaluate_actions_workspace_bytes",
    "ksh_pack_mask", "ksh_policy_head_sample", "ksh_policy_head_evaluate", "ksh_policy_head_evaluate_workspace_bytes",
    "ksh_policy_head_workspace_bytes", "ksh_policy_head_loss_fwd_bwd",
    "ksh_gae_scan", "ksh_ppo_loss_workspace_bytes", "ksh_ppo_loss_fwd_bwd", "ksh_event_create", "ksh_event_record",
    "ksh_event_elapsed_ms", "ksh_event_destroy",
    "ksh_tower_workspace_bytes", "ksh_bn_fwd_train", "ksh_bn_fwd_eval", "ksh_bn_bwd", "ksh_se_pool", "ksh_se_apply",
    "ksh_se_bwd_reduce", "ksh_se_bwd_apply", "ksh_bn_pool_fwd_train", "ksh_se_bn_bwd",
]


class KeiseiHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # not a fallback: build the HIP library itself (hipcc --offload-arch=gfx950) when a checkout has no binary yet
        try:
            from . import build as _build
            if os.environ.get("SHOGIDRL_AMD_LIB"):
                raise RuntimeError("SHOGIDRL_AMD_LIB points at a missing file")
            _build.build(verbose=False)
        except Exception as e:
            raise KeiseiHipError(
                "shogidrl_amd: %s is missing and could not be built (%s). Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback." % (LIB_PATH, e))
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double
    L.ksh_last_error.restype = C.c_char_p
    L.ksh_env_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
    L.ksh_env_destroy.argtypes = [vp]
    L.ksh_env_num_games.argtypes = [vp]
    L.ksh_env_max_moves.argtypes = [vp]
    L.ksh_env_set_obs_format.argtypes = [vp, C.c_int]
    L.ksh_env_obs_format.argtypes = [vp]
    L.ksh_env_set_table_mode.argtypes = [vp, C.c_int]
    L.ksh_env_table_mode.argtypes = [vp]
    L.ksh_env_reset.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ksh_env_observe.argtypes = [vp, vp, vp, vp, vp]
    L.ksh_env_evaluate_termination.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.ksh_env_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    L.ksh_env_reset_packed.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.ksh_env_step_packed.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    L.ksh_env_set_state.argtypes = [vp, i32, vp, vp]
    L.ksh_env_get_state.argtypes = [vp, i32, vp, vp]
    L.ksh_masked_sample.argtypes = [vp, i32, vp, i32, u64, u64, i32, vp, vp, vp]
    L.ksh_masked_sample_packed.argtypes = [vp, i32, vp, i32, u64, u64, i32, vp, vp, vp]
    L.ksh_masked_sample_graph.argtypes = [vp, i32, vp, vp, i32, u64, vp, i32, vp, vp, vp]
    L.ksh_ppo_loss_fwd_bwd_packed.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, i32, vp, vp, vp, vp, vp]
    L.ksh_gae_scan.argtypes = [vp, vp, vp, vp, f64, f64, i32, i32, vp, vp, vp]
    L.ksh_ppo_loss_workspace_bytes.restype = u64
    L.ksh_ppo_loss_workspace_bytes.argtypes = [i32]
    L.ksh_evaluate_actions_workspace_bytes.restype = u64
    L.ksh_evaluate_actions_workspace_bytes.argtypes = [i32]
    L.ksh_evaluate_actions.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]
    L.ksh_pack_mask.argtypes = [vp, i32, vp, vp]
    L.ksh_policy_head_sample.argtypes = [vp, vp, vp, i32, vp, i32, u64, u64, vp, i32, vp, vp, vp]
    L.ksh_policy_head_evaluate_workspace_bytes.restype = u64
    L.ksh_policy_head_evaluate_workspace_bytes.argtypes = [i32, i32, i32]
    L.ksh_policy_head_evaluate.argtypes = [vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, vp, vp]
    L.ksh_policy_head_workspace_bytes.restype = u64
    L.ksh_policy_head_workspace_bytes.argtypes = [i32, i32, i32]
    L.ksh_policy_head_loss_fwd_bwd.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, i32, i32,
                                               vp, vp, vp, vp, vp, vp, vp]
    L.ksh_ppo_loss_fwd_bwd.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, i32, vp, vp, vp, vp, vp]
    i64 = C.c_int64
    L.ksh_tower_workspace_bytes.restype = u64
    L.ksh_tower_workspace_bytes.argtypes = [i32]
    L.ksh_bn_fwd_train.argtypes = [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp]
    L.ksh_bn_fwd_eval.argtypes = [vp, vp, vp, i32, i64, i32, vp, i32, vp]
    L.ksh_bn_bwd.argtypes = [vp, vp, vp, vp, vp, i32, i64, i32, vp, vp, vp, i32, vp, vp, vp, vp]
    L.ksh_se_pool.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    L.ksh_se_apply.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.ksh_se_bwd_reduce.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp]
    L.ksh_se_bwd_apply.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.ksh_bn_pool_fwd_train.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp]
    L.ksh_se_bn_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.ksh_event_create.argtypes = [C.POINTER(vp)]
    L.ksh_event_record.argtypes = [vp, vp]
    L.ksh_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(f32)]
    L.ksh_event_destroy.argtypes = [vp]
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise KeiseiHipError("%s failed (%d): %s" % (what, rc, lib().ksh_last_error().decode()))
