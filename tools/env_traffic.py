#!/usr/bin/env python3
"""Regenerate profiles/env_kernel_hbm_traffic.json (bench.py's `roofline.traffic`) from two rocprofv3 --pmc passes.

On the GPU box (counters need their own passes: FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2):

    R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --roofline-only
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --roofline-only
    python3 $R/tools/env_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write 4096 > $R/gpurun_out/env_kernel_hbm_traffic.json

then copy gpurun_out/env_kernel_hbm_traffic.json to profiles/.  The file is stamped with a hash of the env kernel's
sources (env_source_hash below); bench.py reports `traffic: null` when the stamp does not match the sources it runs, so a
number measured on an older kernel is never attached to a newer one.

Corrections exactly as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE is reported in KB and
counts 128-byte requests as 64 B on gfx950 -> doubled; WRITE_SIZE (KB) is exact for 16-byte-per-lane streaming stores."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV_SOURCES = ("shogidrl_amd/csrc/ksh_env_kernel.h", "shogidrl_amd/csrc/ksh_tables.h")
KERNEL = "ksh_env_step_kernel"


def env_source_hash():
    h = hashlib.sha256()
    for rel in ENV_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def mean_counter(outdir, counter, last_n):
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no *counter_collection.csv under %s" % outdir)
    per = {}
    order = []
    for fn in files:
        with open(fn, newline="") as f:
            for r in csv.DictReader(f):
                if KERNEL not in r["Kernel_Name"] or r["Counter_Name"] != counter:
                    continue
                k = int(r["Dispatch_Id"])
                if k not in per:
                    per[k] = 0.0
                    order.append(k)
                per[k] += float(r["Counter_Value"])       # one row per XCD / instance
    vals = [per[k] for k in sorted(order)][-last_n:]
    if not vals:
        raise SystemExit("no %s rows for %s in %s" % (counter, KERNEL, outdir))
    return sum(vals) / len(vals), len(vals)


if __name__ == "__main__":
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    n_games = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    last_n = int(sys.argv[4]) if len(sys.argv) > 4 else 96
    fetch_kb, nf = mean_counter(fetch_dir, "FETCH_SIZE", last_n)
    write_kb, nw = mean_counter(write_dir, "WRITE_SIZE", last_n)
    fetch_b, write_b = int(round(fetch_kb * 1024 * 2)), int(round(write_kb * 1024))
    print(json.dumps({
        "kernel": KERNEL, "n_games": n_games, "bytes_per_launch": fetch_b + write_b, "fetch_bytes": fetch_b,
        "write_bytes": write_b, "dispatches_averaged": [nf, nw], "env_source_hash": env_source_hash(),
        "algorithmic_bytes_per_launch": n_games * 28710,
        "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE on `python3 bench.py "
                  "--roofline-only`; mean over the last %d replay launches; counters are in KB; FETCH_SIZE doubled (gfx950 "
                  "tallies 128-B requests at 64 B, MI355X_MICROARCH.md); WRITE_SIZE as is (exact for 16-B/lane streaming "
                  "stores).  The kernel also writes the 1 696-byte packed legal bit-set per game (+6 %%), which the "
                  "algorithmic figure of SURVEY 8(d) does not count." % last_n}, indent=1))
