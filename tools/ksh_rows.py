"""Aggregate the hand-written kernels' rows out of a rocprofv3 kernel trace (csv) -- the trace of a full bench run is
hundreds of MB, so only this summary is kept under profiles/.

usage: python tools/ksh_rows.py <..._kernel_trace.csv> <out.csv> [n_replay]
The last `n_replay` (default 96) launches of ksh_env_step_kernel are bench.py's back-to-back roofline replay and are
also reported on their own."""
import csv
import sys
from collections import OrderedDict


def main():
    src, dst = sys.argv[1], sys.argv[2]
    n_replay = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    rows = OrderedDict()
    with open(src, newline="") as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "ksh_" not in name:
                continue
            rows.setdefault(name, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us", "note"])
        for name, spans in rows.items():
            d = [(e - s) / 1e3 for s, e in spans]
            w.writerow([name, len(d), round(sum(d) / len(d), 2), round(min(d), 2), round(max(d), 2), ""])
        env = [k for k in rows if "ksh_env_step_kernel" in k]
        if env and len(rows[env[0]]) > n_replay:
            spans = sorted(rows[env[0]])[-n_replay:]
            d = [(e - s) / 1e3 for s, e in spans]
            w.writerow(["ksh_env_step_kernel (roofline replay: last %d launches, back-to-back)" % n_replay, len(d),
                        round(sum(d) / len(d), 2), round(min(d), 2), round(max(d), 2),
                        "bench.py roofline.avg_launch_us is measured on exactly these launches"])


if __name__ == "__main__":
    main()
