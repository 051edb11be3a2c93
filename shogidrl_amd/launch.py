"""One-node launcher: start one fresh process per GPU and wire them up for torch.distributed (RCCL over xGMI).

`python bench.py --gpus N` (and tools built the same way) call `spawn_ranks` when they were NOT started by
torch.distributed.run, so that asking for N GPUs always means N ranks.  The reference has no launcher to mirror
(its `ddp` flag is dead, keisei/config_schema.py:81).

Rules this follows (they matter on the GPU pool):
  * the parent makes NO GPU call -- it only counts devices (`torch.cuda.device_count()` does not create a context) and
    starts children with `subprocess`; a process that has touched the GPU is never re-exec'ed;
  * every child is a fresh interpreter with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set;
  * rank 0's stdout is the job's stdout (the one JSON line); the other ranks' stdout goes to stderr;
  * if any child exits non-zero (or the time limit passes) the others are terminated and the parent exits non-zero;
  * more ranks than GPUs is an error unless KSH_ALLOW_GPU_SHARING=1 (rehearsals of the N-rank path over gloo on a
    one-GPU box), so `--gpus 2` on a one-GPU machine fails loudly instead of quietly running one rank.
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    """Number of GPUs without creating a HIP context in this process."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def spawn_ranks(argv, n_ranks, timeout=None, require_gpus=True, extra_env=None):
    """Run `argv` (a full command line, e.g. [sys.executable, "bench.py", ...]) as `n_ranks` processes.
    Returns the exit code for the parent: 0 only when every rank exited 0."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    sharing = os.environ.get("KSH_ALLOW_GPU_SHARING") == "1"
    if require_gpus and not sharing:
        have = visible_gpus()
        if have < n_ranks:
            print("launch: %d ranks requested but only %d GPU(s) visible; refusing to run fewer ranks than asked "
                  "(set KSH_ALLOW_GPU_SHARING=1 with KSH_DIST_BACKEND=gloo to rehearse on a shared GPU)" % (n_ranks, have),
                  file=sys.stderr, flush=True)
            return 2
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", KSH_LAUNCHED="1")
        if extra_env:
            env.update(extra_env)
        out = None if r == 0 else sys.stderr          # rank 0 owns stdout (the JSON line)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=out, stderr=None))
    t0 = time.time()
    rc = 0
    alive = set(range(n_ranks))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print("launch: rank %d exited with code %d; stopping the other ranks" % (r, code), file=sys.stderr, flush=True)
        if rc != 0 or (timeout is not None and time.time() - t0 > timeout):
            if rc == 0:
                rc = 124
                print("launch: time limit of %.0f s passed; stopping all ranks" % timeout, file=sys.stderr, flush=True)
            for r in sorted(alive):
                procs[r].terminate()                  # exact PIDs we started, never a pattern
            deadline = time.time() + 10
            for r in sorted(alive):
                try:
                    procs[r].wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            break
        time.sleep(0.05)
    return rc
