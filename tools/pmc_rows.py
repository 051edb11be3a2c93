"""Average rocprofv3 --pmc counter values per dispatch of one kernel.
usage: python tools/pmc_rows.py <..._counter_collection.csv> <kernel-substring> [last_n]"""
import csv
import sys
from collections import OrderedDict

src, needle = sys.argv[1], sys.argv[2]
last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 96
per = OrderedDict()
with open(src, newline="") as f:
    for r in csv.DictReader(f):
        if needle not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Counter_Name"], OrderedDict()).setdefault(r["Dispatch_Id"], 0.0)
        per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for name, d in per.items():
    vals = list(d.values())[-last_n:]
    print("%s,%d,%.1f" % (name, len(vals), sum(vals) / len(vals)))
