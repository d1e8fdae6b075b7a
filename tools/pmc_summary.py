"""Summarise rocprofv3 --pmc output (counter_collection.csv files under a directory):
per kernel name and counter, the mean value per launch.  usage: python tools/pmc_summary.py DIR [name-filter]"""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if filt and filt not in k:
                continue
            a = acc[k.split("(")[0]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
out = {k: {c: {"mean_per_launch": v[0] / v[1], "launches": v[1]} for c, v in cs.items()} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
