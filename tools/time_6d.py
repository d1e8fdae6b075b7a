"""Time the 6-D attitude model (Solver_attitude.run semantics) on an n^6 grid x nu^3 torques.
usage: python tools/time_6d.py [n=24] [nu=11] [stages=2] [variant=-1]     env MODEL=1: next angles computed in the kernel
(HJB_MODEL_QUAT_EULER321, K3 mode 3 - what C3 runs) instead of tabulated as the reference does (mode 2); env WINDOW=3|4: planes of
the per-state window (option window_planes); env UNIWIN=0|1: K3's window modes / K15 (kernels_uniwin.h); UW_TILE: its tile extents"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimal-control-dynamic-programming_amd"))
import hjbdp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
nu = int(sys.argv[2]) if len(sys.argv) > 2 else 11
stages = int(sys.argv[3]) if len(sys.argv) > 3 else 2
variant = int(sys.argv[4]) if len(sys.argv) > 4 else -1
sa = hjbdp.Solver_attitude(n_mesh_w=n, n_mesh_q=n)
sa.U_vector = np.linspace(-0.11, 0.11, nu)
t0 = time.time()
if os.environ.get("MODEL") == "1":
    pspec = sa.build_spec_model()
else:
    spec = sa.build_spec_full()
    pspec, _ = hjbdp.permute_state_axes(spec, sa.AXIS_ORDER)
print("host table build %.1f s" % (time.time() - t0), flush=True)
with hjbdp.Backup(pspec) as bk:
    if variant >= 0:
        bk.set_option("variant", variant)
    if os.environ.get("WINDOW"):
        bk.set_option("window_planes", int(os.environ["WINDOW"]))
    if os.environ.get("UNIWIN"):                 # K15 (kernels_uniwin.h): 0 = K3's window modes 5 / 6, 1 = K15 whenever its structure holds
        bk.set_option("uniwin", int(os.environ["UNIWIN"]))
    if os.environ.get("UW_TILE"):
        bk.set_option("uw_tile", int(os.environ["UW_TILE"]))
    if os.environ.get("UW_CLAIM"):               # K15: 0 = fixed-stride chunk walk, 1 = positions claimed from per-XCD counters (default)
        bk.set_option("uw_claim", int(os.environ["UW_CLAIM"]))
    if os.environ.get("UW_BLOCK"):               # K15: states per chunk = threads per workgroup (256 | 64)
        bk.set_option("uw_block", int(os.environ["UW_BLOCK"]))
    if os.environ.get("LDS_PAD"):                # extra LDS per workgroup: 16384 leaves three workgroups per CU instead of four
        bk.set_option("lds_pad", int(os.environ["LDS_PAD"]))
    print(bk.info(), "packed2_mode", bk.get_option("packed2_mode"), "grid", bk.get_option("grid"), "slow points", bk.get_option("uniwin_slow_points"), flush=True)
    bk.solve(1)
    out = bk.solve(stages)
b = pspec.nS * pspec.nU * stages
print("n=%d nu=%d: %.2f ms/stage, %.3e backups/s sum J %.9e" % (n, nu, out["sweep_ms"] / stages, b / (out["sweep_ms"] * 1e-3), float(out["J"].astype(np.float64).sum())))
