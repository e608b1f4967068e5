"""Summarises a rocprofv3 --pmc counter_collection.csv: per kernel, the mean of every
counter over its dispatches.  Usage: python scripts/pmc_summary.py counter_collection.csv"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
agg = defaultdict(lambda: defaultdict(list))
for r in rows:
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg for c in agg[k]})
print("kernel".ljust(40), " ".join(n[-18:].rjust(18) for n in names))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", kv[1].get(names[0], [0])))):
    print(k[:40].ljust(40), " ".join(("%.4g" % (sum(d[n]) / len(d[n])) if n in d else "-").rjust(18) for n in names))
