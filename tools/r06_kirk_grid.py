"""Kirk's default problem (100 x 100 states x 1000 controls, 199 stages: the control-split kernel, one wave per state) over launch sizes.
usage: python tools/r06_kirk_grid.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimal-control-dynamic-programming_amd"))
import hjbdp
ds = hjbdp.Dynamic_Solver()
spec = ds.build_spec()
with hjbdp.Backup(spec) as bk:
    print("variant", bk.info()["kernel_variant"], "default grid", bk.get_option("grid"))
    for g in (0, 256, 512, 640, 840, 1024, 1250, 1280, 1536, 2048, 2500):
        if g:
            bk.set_option("grid", g)
        best = None
        for rep in range(3):
            out = bk.solve(ds.N - 1)
            best = out["sweep_ms"] if best is None else min(best, out["sweep_ms"])
        print("grid %5s: %.2f ms per %d stages (sum J %.6e)" % (g or "auto", best, ds.N - 1, float(out["J"].sum())), flush=True)
