"""Table-build time: vector term-sum kernels vs the MFMA outer-sum form (csrc/kernels_prep_mfma.h), same tables.
usage: python tools/time_prep.py   -> one JSON line per workload (C2, C4, C5, a 2-D position channel)"""
import json, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "optimal-control-dynamic-programming_amd")); sys.path.insert(0, ROOT)
import hjbdp
import bench

for wl in ("c2", "c4", "c5"):
    spec, name = bench.build_spec(wl)
    with hjbdp.Backup(spec) as bk:
        res = {"workload": name, "kernel_variant": bk.info()["kernel_variant"]}
        for mode in (0, 1, 0, 1):
            bk.set_option("prep_mfma", mode)
            key = "mfma" if mode else "vector"
            res.setdefault(key + "_us", []).append(bk.get_option("prep_ns") * 1e-3)
            res[key + "_hash"] = bk.get_option("table_hash")
            res["tables"] = bk.get_option("prep_tables")
            if mode:
                res["tables_on_mfma"] = bk.get_option("prep_mfma_tables")
        res["identical"] = res["mfma_hash"] == res["vector_hash"]
        print(json.dumps(res), flush=True)
