"""Quick timing of the C2 stage kernel: python tools/time_c2.py [n] [mu] [stages] [variant] [lds_pad]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / 'optimal-control-dynamic-programming_amd'))
import numpy as np, hjbdp
from hjbdp.synthetic import position3d_spec
n = int(sys.argv[1]) if len(sys.argv)>1 else 101
mu = int(sys.argv[2]) if len(sys.argv)>2 else 21
st = int(sys.argv[3]) if len(sys.argv)>3 else 5
var = int(sys.argv[4]) if len(sys.argv)>4 else None
spec = position3d_spec(n, mu)
pad = int(sys.argv[5]) if len(sys.argv)>5 else 0
with hjbdp.Backup(spec, variant=var) as bk:
    if pad: bk.set_option('lds_pad', pad)
    import os
    if os.environ.get("AXIS0_TABLE"): bk.set_option('axis0_table', int(os.environ["AXIS0_TABLE"]))
    print(bk.info(), 'axis0_table', bk.get_option('axis0_table'))
    out = bk.solve(2)
    out = bk.solve(st)
    ms = out['sweep_ms']/st
    print('n=%d mu=%d: %.3f ms/stage, %.3e backups/s' % (n, mu, ms, spec.nS*spec.nU/ (ms*1e-3)))
    print('sum J %.12e' % float(out['J'].astype(np.float64).sum()))
    print('J range', out['J'].min(), out['J'].max(), 'idx range', out['idx'].min(), out['idx'].max())
