"""Builds libkeisei_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No JIT cache, no pip."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG, "csrc", "ksh_lib.hip")
OUT = os.path.join(PKG, "libkeisei_hip.so")
DEPS = [SRC] + [os.path.join(PKG, "csrc", f) for f in ("ksh_env_kernel.h", "ksh_ppo_kernels.h", "ksh_head_kernels.h", "ksh_tables.h", "ksh_tower_kernels.h",
                                                            "ksh_device_util.h")] + \
       [os.path.join(os.path.dirname(PKG), "include", "keisei_hip.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> shogidrl_amd/libkeisei_hip.so"""
    if not force and not stale():
        return OUT
    cmd = [hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT
