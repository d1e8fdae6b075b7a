import sys, time
sys.path.insert(0,'optimal-control-dynamic-programming_amd'); sys.path.insert(0,'.')
import numpy as np, hjbdp
from hjbdp.synthetic import position3d_spec
n = int(sys.argv[1]) if len(sys.argv)>1 else 101
mu = int(sys.argv[2]) if len(sys.argv)>2 else 21
st = int(sys.argv[3]) if len(sys.argv)>3 else 5
spec = position3d_spec(n, mu)
with hjbdp.Backup(spec) as bk:
    print(bk.info())
    out = bk.solve(2)
    out = bk.solve(st)
    ms = out['sweep_ms']/st
    print('n=%d mu=%d: %.3f ms/stage, %.3e backups/s' % (n, mu, ms, spec.nS*spec.nU/ (ms*1e-3)))
    print('J range', out['J'].min(), out['J'].max(), 'idx range', out['idx'].min(), out['idx'].max())
import torch
print('torch sees', torch.cuda.is_available(), torch.cuda.device_count())
