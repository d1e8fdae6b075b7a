"""Build profiles/pmc_traffic.json from rocprofv3 --pmc passes of bench.py (one directory per pass).
usage: python tools/make_pmc_traffic.py FETCH_DIR WRITE_DIR SQ_DIR OUT.json
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3 on this image; dword-per-lane reads are counted 1x
(calibrated on torch's reduce kernel, which reads J exactly once - see the 'calibration' field)."""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(root):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                a = acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return {k: {c: (v[0] / v[1], v[1]) for c, v in cs.items()} for k, cs in acc.items()}


def pick(d, needle):
    ks = [k for k in d if needle in k]
    if not ks:
        raise SystemExit("no kernel matching %r in %s" % (needle, list(d)))
    return ks[0], d[ks[0]]


fetch, write, sq = (collect(p) for p in sys.argv[1:4])
kname, kf = pick(fetch, "k_backup_packed2")
_, kw = pick(write, "k_backup_packed2")
_, ks = pick(sq, "k_backup_packed2")
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* GRBM_GUI_ACTIVE (separate passes, --kernel-trace), "
              "python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline",
    "kernel": kname,
    "unit_note": "rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KiB",
    "FETCH_SIZE_KiB_per_launch": kf["FETCH_SIZE"][0], "FETCH_SIZE_launches": kf["FETCH_SIZE"][1],
    "WRITE_SIZE_KiB_per_launch": kw["WRITE_SIZE"][0], "WRITE_SIZE_launches": kw["WRITE_SIZE"][1],
    "sq_per_launch": {c: v[0] for c, v in sorted(ks.items())},
}
red = [k for k in fetch if "reduce_kernel" in k]
if red:
    out["FETCH_SIZE_KiB_torch_sum_kernel_reading_J_once"] = fetch[red[0]]["FETCH_SIZE"][0]
    out["calibration"] = ("torch's reduce kernel reads the 4,121,204-byte J exactly once; its FETCH_SIZE shows how "
                          "dword-per-lane reads are counted (1x here; the guide's 1/2 factor applies to 16-B/lane streams)")
cyc = ks.get("GRBM_GUI_ACTIVE", (0, 0))[0] / 8.0          # the counter is summed over the 8 XCDs
if cyc:
    out["gpu_cycles_per_launch"] = cyc
    out["valu_busy_frac"] = ks["SQ_INSTS_VALU"][0] * 4.0 / (1024.0 * cyc)
    out["valu_busy_note"] = ("SQ_INSTS_VALU x 4 cycles (wave64 on a 16-lane SIMD) / (1024 SIMDs x cycles), "
                             "cycles = GRBM_GUI_ACTIVE / 8 XCDs")
out["hbm_bytes_per_launch"] = (out["FETCH_SIZE_KiB_per_launch"] + out["WRITE_SIZE_KiB_per_launch"]) * 1024.0
out["algorithmic_bytes_per_launch"] = 12 * 101 ** 3
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
