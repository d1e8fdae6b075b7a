#!/usr/bin/env python3
"""bench.py - the reference's headline metric on MI355X.

Metric  : Bellman backups/s (state x control x stage)                     [BASELINE.json `metric`]
Workload: C4, the configuration the north-star target is quoted on ("the 6-D pos-att grid", BASELINE.json
          configs[3]): ONE channel of Solver_pos_att (pos-att/Solver_pos_att.m:244-297) on a 120^4 = 2.07e8-cell
          sym_linspace grid x the 9 thruster combinations, the reference's typing (single J and
          cost, DOUBLE query tables :299-327 = table_dtype float64; U_Optimal_id as one byte per state), zero terminal
          cost (Solver_pos_att.m:264-265), state axes relabelled (x, theta, w, v) = Solver_pos_att.FAST_AXIS_ORDER.
          The other BASELINE configs are measured in the same run as extra keys of the line (`other_workloads`:
          C5 = C4 with float16 cost-to-go storage, C2 = Solver_position 101^3 x 21^3, 6D = the attitude model of
          Solver_attitude.run on 24^6 states x 11^3 torques - SURVEY 8(d)'s "6-D" figure - and C3 = that model on
          51^6 states with the next angles computed in the kernel: BASELINE configs[2], 176 GB resident, 1 warm-up + 2
          timed stages, skipped with the reason stated when less than 190 GB of HBM are free), each with its own
          roofline object from the same live counter passes; they are also parity tests.
Timing  : the timed region (K steps between barriers) is repeated 3 times; `ms_per_step` / `value` are the MEDIAN
          repetition, `ms_per_step_min` and `ms_per_step_reps` say what the others were.
Step    : ONE stage of the backward sweep = one fused backup kernel over the whole grid (1.866e9 backups).
N GPUs  : one process per GPU (torchrun), the FIXED grid sharded along its last state axis (v: next states move
          < 1 plane, so the halo is one plane each side) = STRONG scaling; neighbour halo exchange per stage over RCCL
          (torch.distributed P2P) overlapped with the interior planes' kernel.  A weak-scaling figure (120 planes of
          the last axis per GPU) is an extra key.

Prints ONE JSON line on rank 0.  `roofline`: the path is VALU-bound at every BASELINE config, so
`achieved` = algorithmic flops F_alg(4) = 71 per backup / launch time against the 157.3 TFLOP/s fp32 vector peak; the
executed-instruction view (`valu_issue_util`) and `traffic` come from rocprofv3 --pmc passes that THIS run makes on
child processes of the same command (rank 0, N = 1; null when rocprofv3 is unavailable).  `cpu_baseline` times the
oracle's C twin (oracle/hjb_oracle.c, OpenMP) on a bounded slab sample of the same workload IN THE SAME TYPING
(float64-built queries, float32 blend: `cpu_baseline.typing`).
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "optimal-control-dynamic-programming_amd"))
sys.path.insert(0, str(ROOT))

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector (= f32-input MFMA rate)
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_CYCLES_PER_WAVE_INSTR = 2.0   # wave64 on a SIMD-32 (MI355X_MICROARCH.md; profiles/r02_valu_rate.json)
MEASURED_PK_FMA_TFLOPS = 118.1     # what v_pk_fma_f32 delivers on this part at four waves per SIMD (tools/valu_rate.hip, profiles/r05_valu_rate.json)
KERNEL_OF_VARIANT = {7: "k_backup_colsweep", 6: "k_backup_row", 5: "k_backup_tabled", 4: "k_backup_packed2",
                     2: "k_backup_packed", 1: "k_backup_nested", 3: "k_backup_ctrlsplit", 0: "k_backup_generic"}
# how the stage kernel of each workload is told apart in one rocprofv3 pass over all of them (demangled names)
# (substring the name must hold, substring it must not hold): the binary16 type is spelt differently by demanglers
# the first entry may be a tuple of alternatives (the window modes of K3: three- or four-plane window)
KERNEL_FILTER = {"c4": ("k_backup_colsweep<float, float", None), "c5": ("k_backup_colsweep", "k_backup_colsweep<float, float"),   # rocprofv3 leaves the binary16 name mangled
                 "c2": ("k_backup_packed2<float, 3", None),
                 # the 6-D grids: K15 (csrc/kernels_uniwin.h) where its structure holds, else K3's window modes
                 "6d": (("k_backup_uniwin<float, 6, false", "k_backup_packed2<float, 6, 5>", "k_backup_packed2<float, 6, 2>"), None),      # tabulated next angles
                 "c3": (("k_backup_uniwin<float, 6, true", "k_backup_packed2<float, 6, 6>", "k_backup_packed2<float, 6, 3>"), None)}      # on-the-fly model
EXTRA_STEPS = {"c5": 20, "c2": 20, "6d": 4, "c3": 2}
C3_NEEDS_GIB = 190          # J_k+1 + J_k (70.4 GB each) + uint16 labels (35.2 GB) = 176 GB resident


def f_alg(D):
    return 3 * (2 ** D - 1) + 6 * D + 2   # SURVEY.md 8(d) "Algorithmic flops per backup"


def build_spec(workload, n_last=None, n=120):
    """-> (ProblemSpec, description).  n_last: planes of the last axis (weak scaling grows it); n: points per axis
    of the pos-att grid (the configs are n = 120; smaller grids are for tests)."""
    import numpy as np
    import hjbdp
    if workload in ("c4", "c5"):
        pa = hjbdp.Solver_pos_att()
        pa.cost_mode = "terms"
        pa.n_mesh_x = pa.n_mesh_v = pa.n_mesh_t = pa.n_mesh_w = n
        sx, sv, st, sw = pa.grids()
        if n_last and n_last != n:       # weak scaling: more v planes at the same spacing
            from hjbdp.matlab_compat import sym_linspace_pos_att
            sv = sym_linspace_pos_att(pa.v_min * n_last / float(n), pa.v_max * n_last / float(n), n_last)
        spec, _ = pa.build_channel_spec(sx, sv, st[0], sw, pa.F_Thr0, pa.F_Thr1, pa.F_Thr6, pa.F_Thr7,
                                        pa.Qx1, pa.Qv1, pa.Qt1, pa.Qw1, pa.R1, pa.J2)
        order = hjbdp.suggest_axis_order(spec)                            # what the library proposes for the reference's (x, v, theta, w):
        assert order == (0, 2, 3, 1), order                              # (x, theta, w, v) - v last = the sharded axis
        assert order == hjbdp.Solver_pos_att.FAST_AXIS_ORDER
        spec, _ = hjbdp.permute_state_axes(spec, order)
        if workload == "c5":
            spec = hjbdp.ProblemSpec(spec.knots, spec.m, spec.next_terms, spec.cost_terms, dtype=np.float32,
                                     index_base=1, j_storage=np.float16, idx_dtype=spec.idx_dtype, table_dtype=spec.table_dtype)
        assert spec.table_dtype == np.float64 and spec.idx_np_dtype == np.uint8
        name = "%s Solver_pos_att channel x: %s states (x,theta,w,v) x %d thruster combinations, %s, float64-built query tables, uint8 argmin, 1 stage per step" % (
            workload.upper(), "x".join(str(k) for k in spec.n), spec.nU,
            "float32" if workload == "c4" else "float32 arithmetic, float16 cost-to-go storage")
        return spec, name
    if workload == "6d":
        # "also a 6-D 24^6 variant of C3's model for the north-star '6-D' figure": Solver_attitude.run
        # (attitude-control/Solver_attitude.m:261-300) with next angles tabulated as the reference does (:449-504)
        sa = hjbdp.Solver_attitude(n_mesh_w=24, n_mesh_q=24)
        sa.U_vector = np.linspace(-0.11, 0.11, 11)
        spec0 = sa.build_spec_full()
        spec, _ = hjbdp.permute_state_axes(spec0, sa.AXIS_ORDER)
        spec = hjbdp.ProblemSpec(spec.knots, spec.m, spec.next_terms, spec.cost_terms, dtype=np.float32, index_base=spec.index_base,
                                 idx_dtype="auto")
        return spec, "6D Solver_attitude.run model: %s states (yaw,pitch,roll,w1,w2,w3) x 11^3 torques, float32, uint16 argmin, 1 stage per step" % "x".join(str(k) for k in spec.n)
    if workload == "c3":
        # BASELINE configs[2]: Solver_attitude.run's model on 51^6 states x 11^3 torques, next angles computed in the stage
        # kernel from four 51^3 quaternion tables (hjbdp.h HJB_MODEL_QUAT_EULER321): nothing nS-sized but J and the labels
        sa = hjbdp.Solver_attitude(n_mesh_w=n, n_mesh_q=n)
        sa.U_vector = np.linspace(-0.11, 0.11, 11)
        s0 = sa.build_spec_model()
        spec = hjbdp.ProblemSpec(s0.knots, s0.m, s0.next_terms, s0.cost_terms, dtype=np.float32, index_base=s0.index_base,
                                 model=s0.model, idx_dtype="auto")
        return spec, "C3 Solver_attitude.run model: %s states (yaw,pitch,roll,w1,w2,w3) x 11^3 torques, float32, on-the-fly quaternion model, uint16 argmin, 1 stage per step" % "x".join(str(k) for k in spec.n)
    if workload == "c2":
        from hjbdp.synthetic import position3d_spec
        spec = position3d_spec(n=101, mu=21, n_last=n_last)
        spec = hjbdp.ProblemSpec(spec.knots, spec.m, spec.next_terms, spec.cost_terms, dtype=np.float32, index_base=spec.index_base,
                                 idx_dtype="auto")
        return spec, "C2 Solver_position 3-DOF: %s states x 21^3 controls, float32, uint16 argmin, 1 stage per step" % "x".join(str(k) for k in spec.n)
    raise ValueError(workload)


def cpu_baseline(spec, budget_s=15.0, dataflow_spec=None):
    """Oracle C twin on this host's cores, on a slab sample (whole planes of the last axis, mid-grid, with
    halos) of the SAME workload."""
    import numpy as np
    from hjbdp import _abi
    from hjbdp.sharded import required_halo
    from oracle import c_oracle
    c_oracle.build()
    lib = c_oracle.lib(_abi)
    cores = int(lib.orc_max_threads())
    inner = spec.nS // spec.n[-1]
    hl, hh = required_halo(spec)
    mid = spec.n[-1] // 2
    rng = np.random.default_rng(0)
    typing = ("float64-built queries (double next-state operands, located and weighted in double, weight rounded once), "
              "float32 blend / cost / argmin: the GPU line's typing" if spec.table_dtype is not None
              else "%s throughout" % np.dtype(spec.dtype).name)

    def run(planes, impl):
        b, e = mid, mid + planes
        J = rng.random(inner * (planes + hl + hh)).astype(spec.j_dtype)
        t0 = time.perf_counter()
        c_oracle.backup_stage(_abi, spec, J, slab=(b, e, hl, hh), nthreads=cores, impl=impl)
        return time.perf_counter() - t0

    def timed(impl, budget):
        t1 = run(1, impl)
        planes = int(max(1, min(spec.n[-1] // 2 - hh - 1, budget / max(t1, 1e-3))))
        t = run(planes, impl) if planes > 1 else t1
        return {"value": inner * planes * spec.nU / t, "unit": "backups/s", "cores": cores, "kind": "port", "typing": typing,
                "sample": "%d of %d planes of the last state axis (%d states x %d controls, 1 stage) in %.1f s" % (
                    planes, spec.n[-1], inner * planes, spec.nU, t)}
    # two forms of the C twin, same results bit for bit (tests/test_oracle_golden.py): the scalar one every parity
    # test compares against, and its AVX2 + FMA row-vectorised form - the faster CPU baseline
    scalar = timed("scalar", budget_s * 0.4)
    scalar["sample"] += "; oracle/hjb_oracle.c backup_f32, scalar, OpenMP"
    try:
        out = timed("avx2", budget_s * 0.6)
        out["sample"] += "; oracle/hjb_oracle.c backup_f32_avx2 (8 states per vector), OpenMP"
        out["scalar_c_openmp"] = scalar
    except RuntimeError:                  # float64 problems: the vector form is float32 only
        out = scalar
    if dataflow_spec is not None:
        # the reference's own dataflow restated in numpy (SURVEY 8d "CPU reference timing"): materialised
        # next-state and cost tables over states x controls, vectorised N-linear interpolation, min over the control axis
        from oracle import hjb_oracle
        ds = dataflow_spec
        prob = hjb_oracle.Problem(ds.knots, ds.m, [[hjb_oracle.Term(tm.dims, tm.data) for tm in ts] for ts in ds.next_terms],
                                  [hjb_oracle.Term(tm.dims, tm.data) for tm in ds.cost_terms], dtype=ds.dtype)
        Jn = rng.random(ds.nS).astype(ds.dtype)
        t0 = time.perf_counter()
        hjb_oracle.backup_stage(prob, Jn)
        td = time.perf_counter() - t0
        out["matlab_dataflow_numpy"] = {"value": ds.nS * ds.nU / td, "unit": "backups/s", "kind": "port",
                                        "sample": "the same problem on a %s grid (%d states x %d controls, 1 stage) in %.1f s; "
                                                  "oracle/hjb_oracle.py: materialised tables + vectorised interpolation + min, "
                                                  "numpy (single process)" % ("x".join(str(k) for k in ds.n), ds.nS, ds.nU, td)}
    return out


def collect_pmc(argv_child, kernel_filters, timeout_s=420):
    """rocprofv3 --pmc passes (one counter set per pass, --kernel-trace only) on child processes running
    `bench.py --pmc-child ...` over every workload of this run; mean per launch of each workload's stage kernel
    (kernel_filters: workload -> substring of the demangled kernel name).  Runs BEFORE this process touches the GPU.
    -> ({workload: {counter: mean per launch}}, note)"""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(rocprof).exists():
        return None, "rocprofv3 not found"
    sets = [["FETCH_SIZE"], ["WRITE_SIZE"],
            ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
            ["SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"]]
    out, notes = {w: {} for w in kernel_filters}, []
    tmp = tempfile.mkdtemp(prefix="hjb_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    try:
        for i, cs in enumerate(sets):
            d = os.path.join(tmp, "p%d" % i)
            cmd = [rocprof, "--pmc", *cs, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, str(ROOT / "bench.py"), "--pmc-child", *argv_child]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                notes.append("pass %d timed out" % i)
                continue
            if r.returncode != 0:
                notes.append("pass %d rc=%d" % (i, r.returncode))
                continue
            acc = {w: {} for w in kernel_filters}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        for w, (kf, knot) in kernel_filters.items():
                            kfs = kf if isinstance(kf, tuple) else (kf,)
                            if any(k in row["Kernel_Name"] for k in kfs) and not (knot and knot in row["Kernel_Name"]):
                                # one row per (dispatch, counter[, dimension instance]): sum the instances of a dispatch, then
                                # average over the dispatches
                                a = acc[w].setdefault(row["Counter_Name"], {})
                                a[row["Dispatch_Id"]] = a.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            for w in kernel_filters:
                for k, per in acc[w].items():
                    out[w][k] = sum(per.values()) / len(per)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {w: v for w, v in out.items() if v}
    return (out or None), "; ".join(notes)


def run_c3(args, steps, warmup, dev, n=51):
    """C3 (BASELINE configs[2]) on ONE GPU, library stage calls on torch-owned device buffers (two J buffers + uint16
    labels = 176 GB at n = 51): `warmup` + `steps` stages from a zero terminal cost, HIP events on the launch stream.
    -> dict like run_workload's, or {"skipped": reason}."""
    import torch
    import hjbdp
    spec, name = build_spec("c3", n=n)
    need = (2 * spec.nS * 4 + spec.nS * spec.idx_np_dtype.itemsize) / 2 ** 30
    free, total = torch.cuda.mem_get_info(dev)
    if n == 51 and free < C3_NEEDS_GIB * 2 ** 30:
        return {"skipped": "C3 needs %d GiB of free HBM (%.0f GiB resident), this device has %.0f of %.0f GiB free"
                           % (C3_NEEDS_GIB, need, free / 2 ** 30, total / 2 ** 30), "workload": name}
    J = [torch.zeros(spec.nS, dtype=torch.float32, device=dev) for _ in range(2)]
    idx = torch.empty(spec.nS, dtype=torch.int16, device=dev)            # uint16 labels: 1331 torque triples
    stream = torch.cuda.current_stream(dev).cuda_stream
    with hjbdp.Backup(spec, device=dev.index or 0) as bk:
        info = bk.info()
        info["packed2_mode"] = bk.get_option("packed2_mode")              # 7 / 8: the rate-shared window kernel (K15) runs
        k = 0
        for _ in range(warmup):
            bk.backup_stage_device(J[k & 1], J[1 - (k & 1)], idx, stream=stream)
            k += 1
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            bk.backup_stage_device(J[k & 1], J[1 - (k & 1)], idx, stream=stream)
            k += 1
        ev1.record()
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        bk.check_device_status(stream)
    dev_ms = ev0.elapsed_time(ev1)
    cs = sum(float(c.double().sum()) for c in J[k & 1].split(1 << 28))      # float64 sums of 1 GiB pieces: no 141 GB temporary
    del J, idx
    torch.cuda.empty_cache()
    return {"spec": spec, "name": name, "info": info, "wall": wall, "dev_ms": dev_ms, "steps": steps, "states_rank": spec.nS,
            "halo": (0, 0), "checksum": cs, "total_backups": spec.nS * spec.nU * steps, "walls": [wall]}


def run_workload(args, workload, steps, warmup, world, rank, dev, dist, weak=False, reps=1, transport=None):
    """Times `steps` stages of `workload` on this rank's slab, `reps` times over (each repetition bracketed by barriers;
    the MEDIAN repetition is reported, all of them are returned).  -> dict of measurements."""
    import torch
    from hjbdp.sharded import ShardedSweep
    n = {"c2": 101, "6d": 24, "c3": args.c3_n}.get(workload, args.grid_n)
    spec, name = build_spec(workload, n_last=n * world if weak else None, n=args.c3_n if workload == "c3" else args.grid_n)
    sw = ShardedSweep(spec, rank, world, dev, overlap=not args.no_overlap, transport=transport or args.transport)
    if args.variant is not None:
        sw.set_option("variant", args.variant)
    info = sw.info()
    info["comm_ranks"] = sw.comm_ranks()
    if info["kernel_variant"] == 4:
        info["packed2_mode"] = sw.get_option("packed2_mode")
    sw.set_terminal(None)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        sw.step()
    walls, devs = [], []
    for _ in range(max(1, reps)):
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()                      # the stream the stage kernels are launched on
        for _ in range(steps):
            sw.step()
        ev1.record()
        barrier()
        w = time.perf_counter() - t0
        devs.append(ev0.elapsed_time(ev1))
        if world > 1:
            tt = torch.tensor([w], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            w = float(tt[0])
        walls.append(w)
    sw.check_device_status()
    med = sorted(range(len(walls)), key=lambda i: walls[i])[len(walls) // 2]      # the median repetition (by the max-over-ranks wall)
    wall, dev_ms = walls[med], devs[med]
    cs = sw.owned_J().double().sum().reshape(1)
    if world > 1:
        cs = cs.cpu() if args.backend != "nccl" else cs
        dist.all_reduce(cs)
    res = {"spec": spec, "name": name, "info": info, "wall": wall, "dev_ms": dev_ms, "steps": steps,
           "states_rank": sw.owned * sw.inner, "halo": (sw.halo_lo, sw.halo_hi), "checksum": float(cs[0]),
           "total_backups": spec.nS * spec.nU * steps, "walls": walls}
    sw.close()
    return res


class OptionalLegsWatchdog:
    """N > 1: the headline leg is measured FIRST; what follows it - the other halo transport's leg, the weak-scaling extra - is
    optional and has never met real multi-GPU hardware.  If those parts hang (a collective whose peers never arrive), every rank
    leaves on its own timer and rank 0 prints the line the headline leg alone supports, instead of the run dying in the driver's
    time limit without a line.  Disarmed before the normal print."""

    def __init__(self, seconds, rank, line_fn):
        import threading
        self._lock = threading.Lock()
        self._done = False
        self._rank, self._line_fn = rank, line_fn
        self._timer = threading.Timer(seconds, self._fire)
        self._timer.daemon = True
        self._timer.start()

    def _fire(self):
        with self._lock:
            if self._done:
                return
            self._done = True
            if self._rank == 0:
                try:
                    print(json.dumps(self._line_fn()), flush=True)
                except Exception as e:       # noqa: BLE001
                    print("bench.py watchdog: could not assemble the line: %s" % e, file=sys.stderr, flush=True)
            sys.stdout.flush()
            os._exit(0)

    def swap(self, line_fn):
        """The optional legs are through: what the timer would print from now on is the complete line (it stays armed across the
        final barrier - a peer that left on its own timer a moment earlier never arrives there).  -> False if it has fired."""
        with self._lock:
            if self._done:
                return False
            self._line_fn = line_fn
        return True

    def disarm(self):
        """-> True if the normal path may print (the timer has not fired)."""
        with self._lock:
            if self._done:
                return False
            self._done = True
        self._timer.cancel()
        return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c4", choices=["c4", "c5", "c2", "6d", "c3"],
                    help="headline workload (default: C4).  c3 = BASELINE configs[2] as the headline, meant for --gpus N: 51^6 states sharded "
                         "along w3 (22 GB of J per rank at N = 8, a halo of one 1.38 GB plane per neighbour and stage); at N = 1 it needs "
                         "190 GB of free HBM; no CPU leg and no extra workloads ride on it")
    ap.add_argument("--c3-n", type=int, default=51, help="points per axis of the c3 workload (config: 51; smaller = testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE configs / the weak-scaling figure")
    ap.add_argument("--no-c3", action="store_true", help="skip the C3 leg (51^6 states, 176 GB, ~20 s + ~50 s of counter passes)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: exchange halos, then compute (no overlap)")
    ap.add_argument("--variant", type=int, default=None, help="force a stage-kernel variant (testing)")
    ap.add_argument("--grid-n", type=int, default=120, help="points per axis of the pos-att grid (config: 120; smaller = testing)")
    ap.add_argument("--backend", default="nccl", help="process-group backend; 'gloo' + --share-gpu is a 1-GPU test mode")
    ap.add_argument("--test-hang-optional", type=int, default=-1,
                    help="test hook (tests/test_gpu_parity.py): rank R never leaves the optional part after the headline leg (-2: every rank)")
    ap.add_argument("--optional-timeout", type=float, default=180.0,
                    help="N > 1: seconds the legs AFTER the headline (other transport, weak scaling) may take before every rank leaves "
                         "and rank 0 prints the headline-only line (at least 120 x the headline leg's own duration)")
    ap.add_argument("--transport", default="torch", choices=["torch", "lib"],
                    help="N > 1: who moves the halo planes - torch.distributed P2P from Python (default), or the RCCL transport inside "
                         "libhjbdp (hjb_rank_step: one library call per stage; needs one GPU per rank)")
    ap.add_argument("--share-gpu", action="store_true", help="testing: every rank uses cuda:0")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.workload == "c3":                # the 176 GB configuration as the headline: nothing else rides on the line
        args.no_extras = args.no_cpu_baseline = True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    # ---- PMC passes on child processes, before this process initialises the GPU ------------------------------
    extras = [w for w in ("c5", "c2", "6d", "c3") if w != args.workload] if (world == 1 and not args.no_extras and args.grid_n == 120) else []
    if args.no_c3 and "c3" in extras:
        extras.remove("c3")
    pmc_all, pmc_note = None, "not collected"
    if world == 1 and not args.pmc_child and not args.no_pmc:
        child = ["--workload", args.workload, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pmc",
                 "--grid-n", str(args.grid_n)]
        if not extras:
            child.append("--no-extras")
        if args.variant is not None:
            child += ["--variant", str(args.variant)]
        filt = {w: KERNEL_FILTER[w] for w in [args.workload] + extras}
        if args.variant is not None:
            filt[args.workload] = (KERNEL_OF_VARIANT.get(args.variant, "k_backup_"), None)
        pmc_all, pmc_note = collect_pmc(child, filt)

    import numpy as np  # noqa: F401
    import torch
    import hjbdp  # noqa: F401

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    head = run_workload(args, args.workload, args.steps, args.warmup, world, rank, dev, dist, reps=1 if args.pmc_child else 3)
    # N > 1 over RCCL: BOTH halo transports sweep the same grid under the same timing contract - `--transport`
    # first, then the other one (the RCCL calls inside libhjbdp, or torch.distributed's P2P from Python).  The line's HEADLINE is the
    # `--transport` leg, whatever the other one measures (the maximum of two noisy legs is biased upward and would not compare with
    # single-leg lines: ADVICE r05); the other leg is reported beside it under `transports`.  A second leg that fails on ANY rank is
    # dropped on EVERY rank: the ranks all-reduce a success flag behind it (ShardedSweep has already agreed across the ranks on
    # whether the library's RCCL transport is reachable BEFORE its collective set-up).
    legs, leg_error, chosen = {args.transport: head}, None, args.transport
    other = "lib" if args.transport == "torch" else "torch"
    watchdog = None
    if world > 1 and not args.pmc_child:
        def headline_only():
            sp = head["spec"]
            launch_ms = head["dev_ms"] / head["steps"]
            tfl = f_alg(sp.D) * head["states_rank"] * sp.nU / (launch_ms * 1e-3) / 1e12
            return {"metric": "bellman_backups_per_s", "value": head["total_backups"] / head["wall"], "unit": "backups/s", "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["wall"] * 1e3 / args.steps,
                    "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                    "dtype": "f32" if sp.j_dtype.itemsize == 4 else "f32 (J stored as f16)", "data": "synthetic",
                    "config": {"workload": head["name"], "states": sp.nS, "states_per_gpu": head["states_rank"], "controls": sp.nU,
                               "stages": args.steps, "kernel_variant": head["info"]["kernel_variant"]},
                    "roofline": {"bound": "valu", "achieved": tfl, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tfl / PEAK_FP32_TFLOPS,
                                 "traffic": None, "avg_launch_ms": launch_ms, "note": "rank 0's stage kernels; per-rank algorithmic flops"},
                    "checksum_sum_J": head["checksum"],
                    "transports": {"headline": chosen, args.transport: {"ms_per_step": head["wall"] * 1e3 / head["steps"],
                                                                         "comm_ranks": head["info"].get("comm_ranks")}},
                    "incomplete": "the optional legs after the headline (the other halo transport, weak scaling) did not finish within "
                                  "%d s and were abandoned; the headline leg above was measured in full" % int(limit)}
        limit = max(args.optional_timeout, 40.0 * head["wall"] * 3)
        watchdog = OptionalLegsWatchdog(limit, rank, headline_only)
        if args.test_hang_optional == rank or args.test_hang_optional == -2:
            time.sleep(36000)
    if world > 1 and args.backend == "nccl" and not args.pmc_child:
        ok = 1
        try:
            legs[other] = run_workload(args, args.workload, args.steps, args.warmup, world, rank, dev, dist, reps=3, transport=other)
        except Exception as e:               # noqa: BLE001 - recorded in the line, the measured leg stands
            leg_error = "%s: %s" % (type(e).__name__, e)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            legs.pop(other, None)
            leg_error = leg_error or "the %s leg failed on another rank" % other
    if args.pmc_child:                      # the counter passes: a few launches of every workload's stage kernel, nothing else
        for w in extras:
            if w == "c3":
                run_c3(args, 1, 0, dev)
            else:
                run_workload(args, w, 2, 1, world, rank, dev, dist)
        return
    spec, info = head["spec"], head["info"]
    value = head["total_backups"] / head["wall"]

    def roofline_of(res, workload):
        """The roofline object of one workload: algorithmic flops and bytes per launch over the HIP-event launch time,
        the executed-instruction view and the HBM traffic from this run's own counter passes."""
        sp, inf = res["spec"], res["info"]
        launch_ms = res["dev_ms"] / res["steps"]                   # rank 0's stage time (one fused kernel; N > 1: + boundary launches)
        backups = res["states_rank"] * sp.nU
        bytes_state = 2 * sp.j_dtype.itemsize + sp.idx_np_dtype.itemsize      # read J_{k+1}, write J_k + the argmin label
        tflops = f_alg(sp.D) * backups / (launch_ms * 1e-3) / 1e12
        gbs = bytes_state * res["states_rank"] / (launch_ms * 1e-3) / 1e9
        kname = KERNEL_OF_VARIANT.get(inf["kernel_variant"], "k_backup")
        if inf["kernel_variant"] == 4 and inf.get("packed2_mode", 0) >= 7:
            kname = "k_backup_uniwin"                              # variant 4, modes 7 / 8 (kernels_uniwin.h)
        pmc = (pmc_all or {}).get(workload)
        traffic = traffic_raw = valu_util = None
        if pmc:
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                # rocprofv3 reports KiB.  On gfx950 FETCH_SIZE tallies the L2's 128-byte fabric requests at 64 bytes
                # (MI355X_MICROARCH.md, HBM: "double it before comparing with a byte count"); calibrated on THIS library's
                # access shapes - 4 / 8 / 16 bytes per lane, streamed and at scattered rows, each byte of 2 GiB read once:
                # FETCH_SIZE = 0.5000 x bytes in all six (tools/fetch_calib.hip, profiles/r04_fetch_calib.json).  64-byte
                # requests (re-reads of small L2-evicted pieces) would be counted in full, so the corrected figure is an
                # upper bound and the uncorrected one a lower bound of the bytes fetched.
                traffic_raw = (pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
                traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
            if "SQ_INSTS_VALU" in pmc and pmc.get("GRBM_GUI_ACTIVE"):
                cyc = pmc["GRBM_GUI_ACTIVE"] / 8.0                                   # summed over the 8 XCDs
                valu_util = pmc["SQ_INSTS_VALU"] * VALU_CYCLES_PER_WAVE_INSTR / (1024.0 * cyc)
        valu_busy = wait_frac = None
        if pmc and pmc.get("GRBM_GUI_ACTIVE"):
            cyc = pmc["GRBM_GUI_ACTIVE"] / 8.0
            if "SQ_ACTIVE_INST_VALU" in pmc:          # rocprof's derived VALUBusy: 4 cycles per counted unit, over SIMDs x cycles
                valu_busy = pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)
            if "SQ_WAIT_ANY" in pmc and pmc.get("SQ_WAVE_CYCLES"):
                wait_frac = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]
        rf = {"bound": "valu", "achieved": tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_FP32_TFLOPS,
              # the same achieved rate against what the packed-fma pipe measurably delivers (the spec peak is not reachable by any
              # instruction mix on this part: profiles/r05_valu_rate.json) - beside `frac`, not instead of it
              "frac_of_measured_pk_fma_peak": tflops / MEASURED_PK_FMA_TFLOPS, "measured_pk_fma_peak": MEASURED_PK_FMA_TFLOPS,
              "traffic": traffic, "traffic_uncorrected": traffic_raw, "kernel": kname, "avg_launch_ms": launch_ms,
              "alg_flop_per_backup": f_alg(sp.D),
              "alg_bytes_per_launch": bytes_state * res["states_rank"], "valu_issue_util": valu_util,
              "valu_busy": valu_busy, "waves_waiting_frac": wait_frac, "pmc": pmc,
              "hbm": {"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "alg_bytes_per_state": bytes_state}}
        if workload == "6d":
            # this workload's next angles are TABULATED: one 8-byte (cell, weight) entry per state and angle axis, read once per
            # stage beside J (the on-the-fly quaternion model of C3's kernel mode computes them instead) - part of `traffic`,
            # not of the algorithmic bytes
            rf["model_table_bytes_per_launch"] = 3 * 8 * res["states_rank"]
        if rf["frac"] > 1.0:
            # F_alg prices every backup at a full N-linear interpolation; this kernel contracts the axes the innermost
            # controls do not move once per outer control step, so it executes a fraction of those flops: the credit is
            # not a utilisation.  The executed-instruction view (valu_issue_util) is the fraction to read.
            rf["alg_flops_credit_TFLOPs"] = tflops
            rf["achieved"] = None if valu_util is None else valu_util * PEAK_FP32_TFLOPS
            rf["frac"] = valu_util
            rf["frac_of_measured_pk_fma_peak"] = None
            rf["frac_is"] = "valu_issue_util (F_alg credit %.1f TFLOP/s exceeds the vector peak: shared interpolation work)" % tflops
        return rf

    rf = roofline_of(head, args.workload)
    kname = rf["kernel"]
    rf["pmc_source"] = ("rocprofv3 --pmc passes made by this run on `bench.py --pmc-child --workload %s --steps 3` (one child per "
                        "counter set, every workload of this line in it; mean per launch of each stage kernel; FETCH_SIZE / "
                        "WRITE_SIZE in KiB; traffic = 2 x FETCH_SIZE + WRITE_SIZE: gfx950 tallies 128-byte fabric reads at 64 bytes - "
                        "0.5000 x bytes on every access shape of this library, profiles/r04_fetch_calib.json; traffic_uncorrected = "
                        "FETCH_SIZE + WRITE_SIZE is the lower bound; FETCH_SIZE counts the L2's requests to the fabric, Infinity-Cache "
                        "hits included: fabric traffic, an upper bound of the HBM bytes)" % args.workload) if pmc_all else pmc_note
    rf["note"] = ("fp32 VALU binds (SURVEY 8d), not HBM and not MFMA (interpolation is a gather; K = D <= 6); peak = fp32 vector "
                  "peak = f32-input MFMA peak.  achieved = ALGORITHMIC flops (F_alg(D) per backup) / launch time; the "
                  "kernel shares the control-independent lerps between the controls, so it executes fewer.  "
                  "valu_issue_util = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE/8): the executed-instruction view at "
                  "the cheapest instruction class; valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 x cycles) = rocprof's VALUBusy (4 "
                  "cycles per instruction: packed fp32, v_min3, v_cndmask, DPP measure 4.1 - 4.3 on this part, plain add / mul "
                  "2.1 - 2.8, profiles/r03_valu_rate.json - the pipe's real occupancy lies between the two); "
                  "waves_waiting_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES")
    out = {
        "metric": "bellman_backups_per_s", "value": value, "unit": "backups/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["wall"] * 1e3 / args.steps,
        "ms_per_step_min": min(head["walls"]) * 1e3 / args.steps,
        "ms_per_step_reps": [w * 1e3 / args.steps for w in head["walls"]],
        "timing": "3 repetitions of the K-step timed region, each between barriers; value / ms_per_step = the median repetition",
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if spec.j_dtype.itemsize == 4 else "f32 (J stored as f16)", "data": "synthetic",
        "config": {"workload": head["name"], "states": spec.nS, "states_per_gpu": head["states_rank"], "controls": spec.nU,
                   "stages": args.steps,
                   "sharding": ("last state axis: %d of %d planes per GPU, halo %d/%d planes (rank 0) exchanged per stage over %s%s"
                                % (head["states_rank"] // (spec.nS // spec.n[-1]), spec.n[-1], head["halo"][0], head["halo"][1],
                                   ("RCCL inside libhjbdp" if chosen == "lib" else "RCCL (torch.distributed P2P)") if args.backend == "nccl" else args.backend + " (test transport)",
                                   "" if args.no_overlap else ", overlapped with the interior planes")) if world > 1 else "none",
                   "kernel_variant": info["kernel_variant"]},
        "roofline": rf,
        "checksum_sum_J": head["checksum"],
    }
    if world > 1:
        # Both transports' legs (measured above): each carries the rank count its communicator ITSELF reports (ncclCommCount
        # through the library / the process group's size), so that a first multi-GPU run verifies what it measured.  Equal
        # checksums = the two transports delivered the same halo planes.
        out["transports"] = {k: {"ms_per_step": r["wall"] * 1e3 / r["steps"], "value": r["total_backups"] / r["wall"],
                                 "checksum_sum_J": r["checksum"], "comm_ranks": r["info"]["comm_ranks"],
                                 "what": "RCCL inside libhjbdp (hjb_rank_step: ncclSend / ncclRecv on the library's transfer stream)"
                                         if k == "lib" else "torch.distributed batch_isend_irecv (" + args.backend + ")"}
                             for k, r in legs.items()}
        if leg_error is not None:
            out["transports"][other] = {"error": leg_error}
        out["transports"]["headline"] = chosen
        out["transports"]["headline_rule"] = "the --transport leg (%s); the other leg is reported beside it, never chosen" % args.transport
        out["transports"]["checksums_equal"] = len({r["checksum"] for r in legs.values()}) == 1
    if not args.no_extras:
        if world == 1:
            others = {}
            for w in extras:
                r = run_c3(args, EXTRA_STEPS[w], 1, dev) if w == "c3" else run_workload(args, w, EXTRA_STEPS[w], 2, world, rank, dev, dist)
                if "skipped" in r:
                    others[w] = r
                    continue
                others[w] = {"workload": r["name"], "value": r["total_backups"] / r["wall"], "unit": "backups/s",
                             "ms_per_step": r["wall"] * 1e3 / r["steps"], "kernel_variant": r["info"]["kernel_variant"],
                             "roofline": roofline_of(r, w), "checksum_sum_J": r["checksum"]}
            out["other_workloads"] = others
        else:
            try:                                 # (an extra: never at the price of the measured headline)
                r = run_workload(args, args.workload, max(10, args.steps // 2), 3, world, rank, dev, dist, weak=True)
                out["weak_scaling"] = {"workload": r["name"], "value": r["total_backups"] / r["wall"], "unit": "backups/s",
                                       "ms_per_step": r["wall"] * 1e3 / r["steps"], "states_per_gpu": r["states_rank"]}
            except Exception as e:               # noqa: BLE001
                out["weak_scaling"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        small = build_spec(args.workload, n=32)[0] if args.workload in ("c4", "c5") else None
        out["cpu_baseline"] = cpu_baseline(spec, dataflow_spec=small)      # the GPU line's own problem, typing included
    if watchdog is not None and not watchdog.swap(lambda: out):
        return                               # (the timer fired and is printing / leaving)
    if world > 1:
        dist.barrier()
    if watchdog is not None and not watchdog.disarm():
        return
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        OptionalLegsWatchdog(20.0, 1, lambda: None)      # (tearing the group down must not outlive the line by more than this)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
