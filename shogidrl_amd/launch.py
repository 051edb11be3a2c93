"""One-node launcher: start one fresh process per GPU and wire them up for torch.distributed (RCCL over xGMI).

`python bench.py --gpus N` (and tools built the same way) call `spawn_ranks` when they were NOT started by
torch.distributed.run, so that asking for N GPUs always means N ranks.  The reference has no launcher to mirror
(its `ddp` flag is dead, keisei/config_schema.py:81).

Rules this follows (they matter on the GPU pool):
  * the parent makes NO GPU call, and cannot make one by accident: it does not even count the devices itself
    (`torch.cuda.device_count()` stays HIP-free only while the optional amdsmi module imports; its fallback is
    hipGetDeviceCount) -- a short-lived child process counts them and prints the number; a process that has touched the
    GPU is never re-exec'ed;
  * every rank gets its own MIOpen user find-db / kernel-cache directory (`per_rank_miopen_env`): N ranks that all run
    MIOpen's exhaustive search at start-up would otherwise queue on the file locks of one sqlite file;
  * every child is a fresh interpreter with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set;
  * rank 0's stdout is the job's stdout (the one JSON line); the other ranks' stdout goes to stderr;
  * if any child exits non-zero (or the time limit passes) the others are terminated and the parent exits non-zero;
  * more ranks than GPUs is an error unless KSH_ALLOW_GPU_SHARING=1 (rehearsals of the N-rank path over gloo on a
    one-GPU box), so `--gpus 2` on a one-GPU machine fails loudly instead of quietly running one rank.
"""
import ctypes
import os
import signal
import socket
import subprocess
import sys
import time


def die_with_parent():
    """preexec_fn for the ranks: PR_SET_PDEATHSIG = SIGTERM, so that a launcher that is killed outright (SIGKILL from a
    driver's time limit cannot be caught) does not leave ranks behind holding GPUs and the rendezvous port"""
    try:
        ctypes.CDLL(None, use_errno=True).prctl(1, signal.SIGTERM)
    except (OSError, AttributeError):
        pass


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus(timeout=600):
    """Number of GPUs a fresh process of this interpreter sees, or None when it could not be counted (the child failed or
    took longer than `timeout`: the first `import torch` on a cold box can take minutes).  Counted in a CHILD process: whatever
    torch.cuda.device_count() has to initialise to answer (amdsmi, or HIP itself when amdsmi is missing) is gone when the
    child exits; this process stays free of any GPU runtime state.  The visibility masks the child inherits
    (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES) are honoured by construction."""
    code = "import torch, sys; sys.stdout.write('KSH_GPUS=%d\\n' % torch.cuda.device_count())"
    try:
        r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout)
    except (OSError, subprocess.TimeoutExpired):
        return None
    for line in r.stdout.decode(errors="replace").splitlines():
        if line.startswith("KSH_GPUS="):
            try:
                return int(line[len("KSH_GPUS="):])
            except ValueError:
                return None
    return None


def per_rank_miopen_env(local_rank, base=None):
    """Environment entries that give rank `local_rank` a private MIOpen user-db and kernel-cache directory (created
    here).  KSH_MIOPEN_DB_BASE overrides the base directory; variables the user has already set are left alone by the
    callers (they use setdefault)."""
    base = base or os.environ.get("KSH_MIOPEN_DB_BASE") or os.path.join(os.path.expanduser("~"), ".config", "miopen_ksh_ranks")
    d = os.path.join(base, "rank%d" % int(local_rank))
    out = {"MIOPEN_USER_DB_PATH": os.path.join(d, "db"), "MIOPEN_CUSTOM_CACHE_DIR": os.path.join(d, "cache")}
    for v in out.values():
        try:
            os.makedirs(v, exist_ok=True)
        except OSError:
            return {}                                  # no writable place: keep MIOpen's defaults rather than fail the launch
    return out


def spawn_ranks(argv, n_ranks, timeout=None, require_gpus=True, extra_env=None):
    """Run `argv` (a full command line, e.g. [sys.executable, "bench.py", ...]) as `n_ranks` processes.
    Returns the exit code for the parent: 0 only when every rank exited 0."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    sharing = os.environ.get("KSH_ALLOW_GPU_SHARING") == "1"
    if require_gpus and not sharing:
        have = visible_gpus()
        if have is None:
            # not a refusal: every rank checks LOCAL_RANK against the devices it sees (parallel.init_from_env) and fails loudly
            print("launch: could not count the GPUs beforehand; starting %d ranks, each of which checks its own device" % n_ranks,
                  file=sys.stderr, flush=True)
        elif have < n_ranks:
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
        if n_ranks > 1:
            for k, v in per_rank_miopen_env(r).items():
                env.setdefault(k, v)
        if extra_env:
            env.update(extra_env)
        out = None if r == 0 else sys.stderr          # rank 0 owns stdout (the JSON line)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=out, stderr=None, preexec_fn=die_with_parent))
    t0 = time.time()
    rc = 0
    stop = []

    def on_signal(signum, frame):                     # the launcher is told to stop (e.g. by a driver): so are its ranks
        stop.append(signum)
    old = {s_: signal.signal(s_, on_signal) for s_ in (signal.SIGTERM, signal.SIGINT)}
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
        if rc != 0 or stop or (timeout is not None and time.time() - t0 > timeout):
            if rc == 0 and stop:
                rc = 128 + stop[0]
                print("launch: signal %d received; stopping all ranks" % stop[0], file=sys.stderr, flush=True)
            elif rc == 0:
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
    for s_, h in old.items():
        signal.signal(s_, h)
    return rc
