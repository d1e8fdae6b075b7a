"""How deep does C5 (C4 with float16 cost-to-go storage) stay finite?  Prints max |J|, min J and finiteness after n stages.
usage: python tools/c5_horizon.py [workload=c5] [n ...]"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "optimal-control-dynamic-programming_amd")); sys.path.insert(0, ROOT)
import hjbdp, bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c5"
ns = [int(a) for a in sys.argv[2:]] or [5, 10, 20, 30, 40, 60, 80, 100, 120, 160, 200]
spec, _ = bench.build_spec(wl)
with hjbdp.Backup(spec) as bk:
    for n in ns:
        J = bk.solve(n)["J"].astype(np.float32)
        fin = np.isfinite(J)
        G = J.reshape(spec.n, order="F")
        bad = np.argwhere(~np.isfinite(G))
        print("%s n=%3d finite=%s nonfinite=%d max=%.4g min=%.4g first_bad=%s" % (wl, n, bool(fin.all()), int((~fin).sum()),
              float(np.nanmax(np.where(fin, J, 0))), float(np.nanmin(np.where(fin, J, 0))), bad[0].tolist() if len(bad) else None), flush=True)
