"""What write-only / read-only / copy bandwidth does this MI355X sustain?  (context for the env kernel's roofline:
its traffic is 96 % stores.)  Times torch's vectorised fill / sum / copy kernels with hipEvents on buffers of the env
kernel's output size (117.6 MB = 4096 games) and larger."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from shogidrl_amd.env import DeviceTimer


def timed(fn, reps):
    tm = DeviceTimer()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    return tm.elapsed_ms() / reps * 1e3      # us per call (back-to-back launches)


for mb in (117.6, 470.4, 1881.6):
    n = int(mb * 1e6) // 16 * 4                      # float32 elements
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    y = torch.empty(n, dtype=torch.float32, device="cuda")
    reps = 200 if mb < 200 else 50
    t_fill = timed(lambda: x.zero_(), reps)
    t_fill1 = timed(lambda: x.fill_(1.0), reps)
    t_sum = timed(lambda: x.sum(), reps)
    t_copy = timed(lambda: y.copy_(x), reps)
    b = n * 4
    print("%7.1f MB: zero_ %.1f us = %.2f TB/s | fill_ %.1f us = %.2f TB/s | sum %.1f us = %.2f TB/s | copy %.1f us = %.2f TB/s (r+w)"
          % (mb, t_fill, b / t_fill / 1e6, t_fill1, b / t_fill1 / 1e6, t_sum, b / t_sum / 1e6, t_copy, 2 * b / t_copy / 1e6), flush=True)
