"""Time one pos-att channel (Solver_pos_att) per stage kernel variant.
usage: python tools/time_posatt.py [n=0 (reference grid 30,30,20,15) | n (n^4 grid)] [stages] [variants...]
env: ORDER=0,2,1,3 relabels the state axes (new axis i = old axis ORDER[i]; old = x,v,theta,w), F16=1 stores J as
float16, CS_XCD_MOD=m sets the residue modulus of variant 7's column -> XCD assignment, CS_DPP=0 its two-loads form, CS_COOP=0 its one-wave-per-column form."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "optimal-control-dynamic-programming_amd"))
import hjbdp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
stages = int(sys.argv[2]) if len(sys.argv) > 2 else 200
variants = [int(v) for v in sys.argv[3:]] or [5]
order = os.environ.get("ORDER")
pa = hjbdp.Solver_pos_att()
pa.cost_mode = os.environ.get("COST", "terms")          # COST=exact | f64 | terms
if n:
    pa.n_mesh_x = pa.n_mesh_v = pa.n_mesh_t = pa.n_mesh_w = n
for k in ("x", "v", "t", "w"):          # N_X / N_V / N_T / N_W override one axis
    if os.environ.get("N_" + k.upper()):
        setattr(pa, "n_mesh_" + k, int(os.environ["N_" + k.upper()]))
sx, sv, st, sw = pa.grids()
spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7, pa.Qx1, pa.Qv1, pa.Qt1,
                                pa.Qw1, pa.R1, pa.J2)
label = "(x,v,theta,w)"
if order:
    perm = tuple(int(c) for c in order.split(","))
    spec, _ = hjbdp.permute_state_axes(spec, perm)
    label = "(" + ",".join("x v theta w".split()[p] for p in perm) + ")"
if os.environ.get("F16") == "1":
    spec = hjbdp.ProblemSpec(spec.knots, spec.m, spec.next_terms, spec.cost_terms, dtype=np.float32, index_base=1,
                             j_storage=np.float16)
    label += " f16"
for v in variants:
    try:
        with hjbdp.Backup(spec, variant=v % 10 if v >= 0 else None) as bk:
            if v >= 10:
                bk.set_option("row_lean", 0)    # 16 = variant 6 without the lean form
            if os.environ.get("CS_XCD_MOD"):
                bk.set_option("cs_xcd_mod", int(os.environ["CS_XCD_MOD"]))
            if os.environ.get("CS_DPP"):
                bk.set_option("cs_dpp", int(os.environ["CS_DPP"]))
            if os.environ.get("CS_XCD_AXIS"):
                bk.set_option("cs_xcd_axis", int(os.environ["CS_XCD_AXIS"]))
            if os.environ.get("GRID"):
                bk.set_option("grid", int(os.environ["GRID"]))
            if os.environ.get("CS_SPLIT"):
                bk.set_option("cs_split", int(os.environ["CS_SPLIT"]))
            if os.environ.get("TABLED_I32"):
                bk.set_option("tabled_i32", int(os.environ["TABLED_I32"]))
            if os.environ.get("CS_COOP"):
                bk.set_option("cs_coop", int(os.environ["CS_COOP"]))
            info = bk.info()
            if info["kernel_variant"] == 7:
                print("variant 7: group axis %d, %d groups, dpp %d, coop %d, rows %d, split %d" % (bk.get_option("cs_group_axis"), bk.get_option("cs_groups"), bk.get_option("cs_dpp"), bk.get_option("cs_coop"), bk.get_option("cs_rows"), bk.get_option("cs_split")))
            bk.solve(2)
            out = bk.solve(stages)
    except hjbdp.HjbError as e:
        print(label, "variant", v, "refused:", e)
        continue
    b = spec.nS * spec.nU * out["stages_done"]
    print("%s grid %s: %d workgroups, variant %d (ran %d): %.4f ms/stage, %.3e backups/s (halo %d/%d) sumJ %.9e" % (
        label, "x".join(str(len(k)) for k in spec.knots), info["grid"], v, info["kernel_variant"], out["sweep_ms"] / out["stages_done"], b / (out["sweep_ms"] * 1e-3),
        info["halo_needed_lo"], info["halo_needed_hi"], float(out["J"].astype(np.float64).sum())), flush=True)
