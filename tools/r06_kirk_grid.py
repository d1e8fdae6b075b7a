"""Kirk's default problem (100 x 100 states x 1000 controls, 199 stages: the control-split kernel, one wave per state) over workgroup
sizes and launch sizes.  usage: python tools/r06_kirk_grid.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimal-control-dynamic-programming_amd"))
import hjbdp
ds = hjbdp.Dynamic_Solver()
spec = ds.build_spec()
with hjbdp.Backup(spec) as bk:
    print("variant", bk.info()["kernel_variant"], "default grid", bk.get_option("grid"))
    for block, grids in ((256, (0, 1024, 2500)), (512, (256, 512, 640, 1024, 1250)), (1024, (128, 256, 320, 512, 625))):
        bk.set_option("block", block)
        for g in grids:
            if g:
                bk.set_option("grid", g)
            best = min(bk.solve(ds.N - 1)["sweep_ms"] for _ in range(3))
            out = bk.solve(ds.N - 1)
            print("block %4d grid %5s: %.2f ms per %d stages (sum J %.6e)" % (block, g or "auto", best, ds.N - 1, float(out["J"].sum())), flush=True)
