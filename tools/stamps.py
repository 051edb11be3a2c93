"""Scratch diagnostic: builds a -DKSH_STAMPS copy of the library and prints the per-phase shader-clock
breakdown of the env kernel (guide section 7, 'In-kernel stamps'). Never used by the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "gpurun_out", "libkeisei_hip_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-DKSH_STAMPS",
                       "-o", so, os.path.join(ROOT, "shogidrl_amd/csrc/ksh_lib.hip")])
import shogidrl_amd._lib as _lib
_lib.LIB_PATH = so
from shogidrl_amd import ops
from shogidrl_amd.env import VecShogiEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = _lib.lib()
L.ksh_debug_set_stamps.argtypes = [C.c_void_p]
env = VecShogiEnv(N, 500)
env.reset()
logits = torch.zeros(N, 13527, device="cuda:0")
for t in range(80):
    a, _ = ops.masked_sample(logits, env.legal_mask, 42, t)
    env.step(a, validate=False)
st = torch.zeros(N, 32, dtype=torch.int64, device="cuda:0")
L.ksh_debug_set_stamps(st.data_ptr())
order = [0, 1, 3, 2, 5, 6, 4, 7, 9, 10]      # stamp ids in the order the kernel passes them (round 3 phase order)
names = ["entry", "state in LDS", "applied+key", "tables+barrier", "obs stores", "movegen", "history", "term/regen", "late obs", "mask stores"]
acc = np.zeros((9,), np.float64)
gacc = np.zeros((7,), np.float64)
nested = 0.0
span = 0.0
reps = 20
for t in range(reps):
    a, _ = ops.masked_sample(logits, env.legal_mask, 42, 100 + t)
    st.zero_()
    env.step(a, validate=False)
    torch.cuda.synchronize()
    s = st.cpu().numpy().astype(np.float64)
    ok = (s[:, 3] > 0)
    d = np.diff(s[ok][:, order], axis=1)
    acc += np.median(d, axis=0)
    gs = s[ok]
    # inside the first move generation: stamps 16+11..16+15, end of main gen at 16+10, flag nested at 16+9
    seq = np.stack([gs[:, 5], gs[:, 27], gs[:, 28], gs[:, 29], gs[:, 30], gs[:, 31], gs[:, 26], gs[:, 6]], axis=1)
    gacc += np.median(np.diff(seq, axis=1), axis=0)
    nested += gs[:, 25].mean()
    span += (s[:, 10].max() - s[:, 0].min())
L.ksh_debug_set_stamps(None)
print("N=%d median per-wave phase durations (s_memtime ticks = 100 MHz? or shader clk), avg over %d steps" % (N, reps))
for i in range(9):
    print("  %-14s -> %-14s %9.1f" % (names[i], names[i + 1], acc[i] / reps))
print("  sum %.1f ; kernel span (first entry -> last end) %.1f" % (acc.sum() / reps, span / reps))
gn = ["obs issued->ballots+compact", "attack sets", "reductions+checkers", "pins+block", "board emit", "drops+count (main gen end)", "ufz nested + rest"]
for i in range(7):
    print("    gen: %-30s %9.1f" % (gn[i], gacc[i] / reps))
print("    fraction of positions needing the nested uchi-fu-zume search: %.3f" % (nested / reps))
