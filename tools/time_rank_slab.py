"""What one rank of an N-GPU run does per stage on C4 (pos-att 120^4, slabs of the last axis), timed on ONE GPU:
the fused launch (halo exchange first, then one kernel) against interior + two boundary strips in line on one stream
and against the strips on streams of their own (what ShardedSweep / hjb_solve_multi do).  usage: time_rank_slab.py [N=8]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimal-control-dynamic-programming_amd"))
import torch, hjbdp
from hjbdp.sharded import required_halo
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
assert N >= 3, "a middle rank (halos on both sides) needs N >= 3"
pa = hjbdp.Solver_pos_att(); pa.cost_mode = "terms"
pa.n_mesh_x = pa.n_mesh_v = pa.n_mesh_t = pa.n_mesh_w = 120
sx, sv, st, sw = pa.grids()
spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
spec, _ = hjbdp.permute_state_axes(spec, (0, 2, 3, 1))
nl = spec.n[-1]; own = nl // N
b = (N // 2) * own; e = b + own                      # a middle rank
hl, hh = required_halo(spec)
inner = spec.nS // nl
dev = torch.device("cuda:0")
J_in = torch.rand((own + hl + hh, inner), device=dev, dtype=torch.float32)
J_out = torch.empty_like(J_in)
idx = torch.empty((own, inner), device=dev, dtype=torch.int32)
whole = hjbdp.Backup(spec, slab=(b, e, hl, hh))
parts = [(hjbdp.Backup(spec, slab=(b + hl, e - hh, hl, hh)), hl, hl, own - hl - hh, hl, hh),     # interior: rows from hl (its own halo = owned planes)
         (hjbdp.Backup(spec, slab=(b, b + hl, hl, min(hh, own - hl))), 0, 0, hl, hl, min(hh, own - hl)),
         (hjbdp.Backup(spec, slab=(e - hh, e, min(hl, own - hh), hh)), own + hl - min(hl, own - hh) - hh, own - hh, hh, min(hl, own - hh), hh)]
if os.environ.get("CS_SPLIT"):                        # CS_SPLIT=s forces the parts per column of every handle
    for h in [whole] + [p[0] for p in parts]:
        h.set_option("cs_split", int(os.environ["CS_SPLIT"]))
print("parts per column: whole %d, interior %d, strips %d / %d" % tuple(h.get_option("cs_split") for h in [whole] + [p[0] for p in parts]))
main = torch.cuda.current_stream(dev)
side = [torch.cuda.Stream(device=dev) for _ in range(2)]

def launch(p, stream):
    h, row0, own0, planes, a, c = p
    n = planes + a + c
    h.backup_stage_device(J_in[row0:row0 + n], J_out[row0:row0 + n], idx[own0:own0 + planes], stream=stream.cuda_stream)

def fused():
    whole.backup_stage_device(J_in, J_out, idx, stream=main.cuda_stream)
def inline():
    for p in parts: launch(p, main)
def beside():
    for s in side: s.wait_stream(main)
    launch(parts[0], main)
    for p, s in zip(parts[1:], side):
        launch(p, s)
    for s in side: main.wait_stream(s)
rk = hjbdp.RankSlab(spec, 0, N // 2, N, overlap=True)      # the same rank through hjb_rank_create: one call per stage
assert (rk.begin, rk.end, rk.halo_lo, rk.halo_hi) == (b, e, hl, hh) and rk.split
def one_call():
    rk.stage(J_in, J_out, idx, compute_stream=main.cuda_stream)
for name, fn in (("fused", fused), ("interior + strips in line", inline), ("strips beside the interior", beside),
                 ("strips beside, hjb_rank_stage", one_call)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): fn()
    t_cpu = (time.perf_counter() - t0) / 300 * 1e3           # host time to enqueue one stage (the GPU may lag behind)
    torch.cuda.synchronize()
    print("N=%d (%d owned planes, halo %d/%d): %-28s %.3f ms per stage (host enqueue %.3f)" % (N, own, hl, hh, name, (time.perf_counter() - t0) / 300 * 1e3, t_cpu))
