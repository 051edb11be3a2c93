"""Data-parallel self-play PPO over the GPUs of one node: one process per GPU, games sharded across ranks
(no env-side communication), RCCL all-reduce of a flat gradient buffer per optimizer update and of the
three advantage moments so that whole-batch normalisation (keisei/core/ppo_agent.py:276-282) is preserved.
The reference has no distributed path (its `ddp` flag is dead, keisei/config_schema.py:81).

Parity statement, exactly as wide as the tests that back it: for a model WITHOUT batch statistics a G-rank update
equals the 1-rank update on the concatenated batch within fp tolerance (tests/test_gpu_ddp.py,
tests/test_parallel_gloo.py).  With BatchNorm the batch statistics are per rank (no SyncBN), so
the G-rank run is a different -- equally valid -- estimator; what IS guaranteed is that parameters stay bit-identical
across ranks (one summed gradient) and that the running statistics are re-synchronised (`sync_buffers`) at the end of
every learn(), so checkpoints and eval-mode values agree on all ranks.  Every rank must hold the same number of
transitions (`check_equal_batches`): a different minibatch count per rank would issue a different number of
collectives and hang."""
import os

import torch
import torch.distributed as dist


def collectives_active(group=None):
    """True when the collectives of this module have to run: an initialised process group with more than one rank -- or
    with ONE rank under KSH_DIST_REHEARSAL=1, which sends a single-GPU job through exactly the code and the RCCL calls
    of a multi-GPU one (the only way to run the "nccl" path on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("KSH_DIST_REHEARSAL") == "1"


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run). Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        ngpu = torch.cuda.device_count()
        if local_rank >= ngpu:
            # one process per GPU.  Ranks may share a GPU only in an explicit rehearsal (gloo collectives; RCCL refuses
            # two ranks on one device): never silently fold N ranks onto fewer GPUs.
            if os.environ.get("KSH_ALLOW_GPU_SHARING") != "1":
                raise RuntimeError("LOCAL_RANK %d but only %d GPU(s) visible: one process per GPU "
                                   "(KSH_ALLOW_GPU_SHARING=1 + KSH_DIST_BACKEND=gloo rehearses on a shared GPU)"
                                   % (local_rank, ngpu))
            local_rank = local_rank % ngpu
    rehearsal = os.environ.get("KSH_DIST_REHEARSAL") == "1"
    if rehearsal and world == 1:      # a plain `python script.py` may rehearse too: the rendezvous of one rank with itself
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29544"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    if (world > 1 or rehearsal) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("KSH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend)
    return rank, local_rank, world


def shard_range(n_total, rank, world, equal=True):
    """Contiguous [lo, hi) slice of the games owned by `rank`.  equal=True (what the training path needs: the same
    number of minibatches, hence collectives, on every rank) raises unless n_total divides evenly; equal=False gives
    remainders to the low ranks (evaluation-style sharding with no collective inside the loop)."""
    base, rem = divmod(n_total, world)
    if equal and rem:
        raise ValueError("%d games do not shard evenly over %d ranks; the training path needs equal shards "
                         "(same number of gradient all-reduces per rank)" % (n_total, world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def check_equal_batches(num, device, group=None):
    """Raise on EVERY rank when the ranks do not hold the same number of transitions (one tiny all-reduce + host read
    per learn() call).  Without it a short rank would leave the minibatch loop early and the others would wait for
    its gradient all-reduce forever."""
    if not collectives_active(group):
        return
    t = torch.tensor([num, -num], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, lo = int(t[0]), -int(t[1])
    if hi != lo:
        raise ValueError("ranks hold different batch sizes (%d..%d transitions); shard the games evenly" % (lo, hi))


class FlatGradBucket:
    """All parameters' .grad are views into ONE contiguous buffer, so the gradient exchange is a single
    in-place all-reduce (53 MB fp32 for the 13.24 M-parameter tower) with no flatten / unflatten copies."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views = []
        off = 0
        for p in self.params:
            # same sizes AND strides as the parameter (channels_last weights keep their layout)
            v = self.flat[off:off + p.numel()].as_strided(p.size(), p.stride())
            self.views.append(v)
            p.grad = v
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def begin(self):
        """Before backward(): drop the .grad views, so that autograd hands each parameter its gradient tensor instead
        of launching one small `view += grad` kernel per parameter (187 of them for the tower)."""
        for p in self.params:
            p.grad = None

    def view_of(self, param):
        """the slice of the flat buffer that holds `param`'s gradient (same sizes and strides as the parameter): a kernel
        may write a gradient straight into it; assign it to param.grad afterwards and gather() leaves it alone"""
        for p, v in zip(self.params, self.views):
            if p is param:
                return v
        raise KeyError("parameter is not in this bucket")

    def gather(self):
        """After backward(): move every gradient into the flat buffer with one multi-tensor copy and point .grad at the
        views again (the optimizer and the all-reduce work on the flat buffer)."""
        dst, src = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()                      # parameter not reached by this backward
            elif p.grad is v:
                pass                           # written in place (view_of): nothing to move
            elif p.grad.dtype == v.dtype:
                dst.append(v)
                src.append(p.grad)
            else:
                v.copy_(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce_mean(self):
        if collectives_active(self.group):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(dist.get_world_size(self.group))


def global_advantage_moments(adv, group=None):
    """(count, mean, unbiased std) of `adv` over ALL ranks from one 3-float all-reduce."""
    a = adv.reshape(-1).double()
    m = torch.stack([torch.tensor(float(a.numel()), dtype=torch.float64, device=a.device), a.sum(), (a * a).sum()])
    if collectives_active(group):
        dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)
    n, s, ss = m[0], m[1], m[2]
    mean = s / n
    var = (ss - n * mean * mean) / torch.clamp(n - 1, min=1.0)
    return n, mean, torch.sqrt(torch.clamp(var, min=0.0))


def all_reduce_mean_(t, group=None):
    if collectives_active(group):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.div_(dist.get_world_size(group))
    return t


def broadcast_parameters(module, src=0, group=None):
    """Learner -> actors weight sync (the reference ships gzip'd state dicts through mp.Queue,
    keisei/training/parallel/communication.py:63-89); here: one RCCL broadcast per tensor."""
    if collectives_active(group):
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


def sync_buffers(module, group=None):
    """Make the module's buffers identical on all ranks: floating buffers (BatchNorm running_mean / running_var, which
    every rank advances with its OWN games' batch statistics) become the mean over ranks through ONE flat all-reduce;
    integer buffers (num_batches_tracked) take the maximum.  Called at the end of every learn(), so parameters AND
    buffers -- i.e. the whole state dict -- agree across ranks whenever a checkpoint can be taken."""
    if not collectives_active(group):
        return
    fb = [b for b in module.buffers() if b.is_floating_point()]
    ib = [b for b in module.buffers() if not b.is_floating_point()]
    if fb:
        flat = torch.cat([b.detach().reshape(-1).float() for b in fb])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(dist.get_world_size(group))
        off = 0
        for b in fb:
            b.data.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
    if ib:
        flat = torch.stack([b.detach().reshape(-1)[0].to(torch.int64) for b in ib])
        dist.all_reduce(flat, op=dist.ReduceOp.MAX, group=group)
        for k, b in enumerate(ib):
            b.data.fill_(flat[k])
