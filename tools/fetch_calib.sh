#!/bin/bash
# FETCH_SIZE on known byte counts in this library's access shapes (tools/fetch_calib.hip) -> gpurun_out/fetch_calib.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib || exit 1
O=gpurun_out/fetch_calib; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -- /tmp/fetch_calib > $O/log 2>&1
python3 - "$O" <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
acc = {}
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != "FETCH_SIZE" or "k_read" not in row["Kernel_Name"]:
            continue
        a = acc.setdefault(row["Kernel_Name"], {})
        a[row["Dispatch_Id"]] = a.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
exp = 2 * 2 ** 30
out = {}
for k, per in sorted(acc.items()):
    vals = [v * 1024.0 for v in per.values()]                      # KiB
    out[k] = {"fetch_size_bytes_per_launch": vals, "ratio_to_bytes_read": [v / exp for v in vals]}
json.dump({"expected_bytes": exp, "kernels": out}, open("gpurun_out/fetch_calib.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
