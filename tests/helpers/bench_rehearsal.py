"""TEST INFRASTRUCTURE: bench.py with the GPU job of one rank (bench.SelfPlayJob: env kernel, network, PPO update) replaced
by a CPU stand-in, so that everything AROUND it runs for real on a box without GPUs and with as many ranks as the day an
8-GPU node appears: the launcher / torch.distributed.run, the supervisor + worker + stage-file structure of rank 0, the
process group (gloo), every collective of shogidrl_amd.parallel that VecPPOAgent.learn() issues (equal-batches check,
advantage moments, the flat-gradient all-reduce per update, the buffer sync, the metrics all-reduce), the max-over-ranks
timing, the per-rank MIOpen directories, and the teardown.  Nothing here is a product path: the stand-in computes nothing
a user could want (a linear model on random tensors) and says so in every object it reports.

usage: exactly bench.py's command line (`python tests/helpers/bench_rehearsal.py --gpus 8 --steps 2 ...`)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


class _WallTimer:
    def start(self):
        self.t0 = time.perf_counter()

    def stop(self):
        self.t1 = time.perf_counter()

    def elapsed_ms(self):
        return (self.t1 - self.t0) * 1e3


class CpuStandInJob:
    def __init__(self, args, rank, local_rank, world):
        import torch
        import torch.distributed as dist
        from shogidrl_amd import parallel
        self.args, self.rank, self.world, self.device = args, rank, world, "cpu"
        torch.set_num_threads(1)                   # (8 ranks on a handful of cores)
        torch.manual_seed(42)
        self.model = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(46 * 81, 32), torch.nn.BatchNorm1d(32),
                                         torch.nn.ReLU(), torch.nn.Linear(32, 8))
        if dist.is_initialized():
            parallel.broadcast_parameters(self.model, src=0)
        self.bucket = parallel.FlatGradBucket(self.model.parameters())
        self.opt = torch.optim.Adam(self.model.parameters(), lr=1e-3)
        self.gen = torch.Generator().manual_seed(1000 + rank)
        self.batch = None
        mine = os.environ.get("MIOPEN_USER_DB_PATH")
        self.miopen_dirs = [mine]
        if dist.is_initialized():                  # (a collective: every rank, here -- not in rank 0's reporting code)
            self.miopen_dirs = [None] * dist.get_world_size()
            dist.all_gather_object(self.miopen_dirs, mine)

    def synchronize(self):
        # KSH_REHEARSAL_HANG_TEARDOWN=1: a teardown that never comes back (armed by the last figure; the synchronize() after
        # that figure passes, the one inside the teardown barrier hangs)
        if getattr(self, "_hang", None) == "armed":
            self._hang = "next"
        elif getattr(self, "_hang", None) == "next":
            time.sleep(3600)

    def timer(self):
        return _WallTimer()

    def prepare(self, note):
        pass

    def collect(self):
        import torch
        n = self.args.games * self.args.rollout
        self.batch = (torch.randn(n, 46, 9, 9, generator=self.gen), torch.randn(n, 8, generator=self.gen),
                      torch.randn(n, generator=self.gen))

    def update(self):
        import torch
        from shogidrl_amd import parallel
        x, y, adv = self.batch
        if os.environ.get("KSH_REHEARSAL_FAIL_RANK") == str(self.rank):
            sys.exit(3)                            # a rank that dies inside the job: the launcher has to stop the others
        parallel.check_equal_batches(x.shape[0], "cpu")
        n, mean, std = parallel.global_advantage_moments(adv)
        acc = torch.zeros(2)
        for _ in range(self.args.epochs):
            for lo in range(0, x.shape[0], self.args.minibatch):
                self.bucket.begin()
                loss = ((self.model(x[lo:lo + self.args.minibatch]) - y[lo:lo + self.args.minibatch]) ** 2).mean()
                loss.backward()
                self.bucket.gather()
                self.bucket.all_reduce_mean()
                self.opt.step()
                acc += torch.stack([loss.detach(), torch.ones(())])
        parallel.sync_buffers(self.model)
        m = parallel.all_reduce_mean_(acc[:1] / acc[1])
        return {"stand_in/loss": float(m), "stand_in/adv_std": float(std)}

    def finish(self):
        pass

    def episode_stats(self):
        return {"total_episodes_completed": 0, "black_wins": 0, "white_wins": 0, "draws": 0}

    def workload(self):
        return "REHEARSAL: CPU stand-in for the rank's job (tests/helpers/bench_rehearsal.py); no figure here is a measurement"

    def extra_headline(self):
        return {"rehearsal": True, "miopen_user_db_paths": self.miopen_dirs}

    def side_figures(self):
        fake = {"bound": "hbm", "achieved": 0.0, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None,
                "avg_launch_us": 1.0, "note": "rehearsal stand-in"}
        figs = [("roofline", lambda: fake)]
        if self.world == 1 and not self.args.no_cpu_baseline:
            figs.append(("cpu_baseline", lambda: {"value": 0.0, "unit": "env-steps/s", "cores": 1, "kind": "port",
                                                  "sample": "rehearsal stand-in"}))
        figs.append(("roofline_8192", lambda: fake))
        figs.append(("fused_policy_head", lambda: {"note": "rehearsal stand-in"}))
        if os.environ.get("KSH_REHEARSAL_HANG_TEARDOWN") == "1":
            def arm():
                self._hang = "armed"
                return {"note": "the worker will hang in its teardown barrier"}
            figs.append(("last_figure_then_hang", arm))
        return figs


bench.JOB_FACTORY = CpuStandInJob
os.environ["KSH_BENCH_REHEARSAL_STAND_IN"] = "1"      # bench.py refuses a replaced job without this (and marks its line)

if __name__ == "__main__":
    sys.exit(bench.main())
