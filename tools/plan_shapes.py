"""Which plan signatures (groups x member slots per window pair) the columns of a pos-att channel have, per grid size and
channel: the survey behind the shape-specialisation experiment of round 3 (profiles/r03_c4_experiments.log).  CPU only."""
import sys, numpy as np, collections
sys.path.insert(0,'/root/repo/optimal-control-dynamic-programming_amd'); sys.path.insert(0,'/root/repo')
import hjbdp
def shapes_for(n, channel=0, fail=False):
    pa = hjbdp.Solver_pos_att(); pa.cost_mode="terms"
    pa.n_mesh_x = pa.n_mesh_v = pa.n_mesh_t = pa.n_mesh_w = n
    sx, sv, st, sw = pa.grids()
    thr = [(pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.J2), (pa.F_Thr2, pa.F_Thr3, pa.F_Thr8, pa.F_Thr9, pa.J3), (pa.F_Thr4, pa.F_Thr5, pa.F_Thr10, pa.F_Thr11, pa.J1)][channel]
    f0 = [0.0] if fail else thr[0]
    spec,_ = pa.build_channel_spec(sx, sv, st[channel], sw, f0, thr[1], thr[2], thr[3], pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, thr[4])
    spec,_ = hjbdp.permute_state_axes(spec,(0,2,3,1))
    nU = spec.nU
    def cells(a):
        k = spec.knots[a]; q=None
        for t in spec.next_terms[a]:
            shape=[1]*5
            for ax,d in enumerate(t.dims): shape[d]=t.data.shape[ax]
            e = np.asarray(t.data,dtype=np.float64).reshape(shape)
            q = e if q is None else q+e
        c = np.clip(np.searchsorted(k,q,side='right')-1,0,len(k)-2)
        return np.squeeze(c)
    c2 = cells(2); c3 = cells(3)
    shapes = collections.Counter()
    for i3 in range(n):
      for i2 in range(n):
        cg = c2[i2]; cw = c3[i3]
        groups = {}
        for u in range(nU): groups.setdefault(cg[u], []).append(u)
        sig=[]
        for g,us in groups.items():
            wmin = min(cw[u] for u in us)
            if wmin+2>n-1: wmin=n-3
            p0 = sum(1 for u in us if cw[u]-wmin==0); p1 = sum(1 for u in us if cw[u]-wmin==1); oth=sum(1 for u in us if cw[u]-wmin>1)
            sig.append((p0,p1) if not oth else (p0,p1,oth))
        shapes[tuple(sorted(sig))]+=1
    tot = n*n
    return [(round(c/tot,3), s) for s,c in shapes.most_common(4)], len(shapes)
for n in (30, 60, 90, 120, 160, 200):
    for ch in (0,1,2):
        print(n, ch, shapes_for(n, ch))
    print(n, 'fail', shapes_for(n, 0, True))
