#!/bin/bash
# GPU clock under the stage kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration, both from ONE rocprofv3 pass
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/clock; rm -rf $O; mkdir -p $O
export ORDER=0,2,3,1
# an alternate build is handed to the loader (hjbdp/core.py reads HJBDP_LIB); the in-tree library is never overwritten
if [ -n "$LIB" ]; then export HJBDP_LIB="$(realpath "$LIB")"; fi
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p -- python3 bench.py --pmc-child --workload c4 --steps 30 --warmup 5 > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
ct = glob.glob("$O/p/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("$O/p/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    if "colsweep" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
cyc = {}
for r in csv.DictReader(open(ct)):
    if "colsweep" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cyc[r["Dispatch_Id"]] = cyc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
ids = [i for i in dur if i in cyc][5:]
d = sum(dur[i] for i in ids) / len(ids); c = sum(cyc[i] for i in ids) / len(ids) / 8
print("launches %d: duration %.3f ms, %.3e cycles -> %.3f GHz" % (len(ids), d * 1e3, c, c / d * 1e-9))
PY
