"""Randomised stress of hjb_solve_batch's host loop (csrc/hjbdp_batch.hip): 2 .. 5 random problems of one column-sweep shape (or small
problems on the table kernel) swept as ONE launch chain - random stage counts around the 32-stage graph, monitor periods and tolerances
that stop some problems early at different stages - against the same problems swept one by one with hjb_solve: values, labels, stages
done and stop flags must be equal.  usage: python tools/stress_batch.py [seconds=120] [seed=0]"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "optimal-control-dynamic-programming_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hjbdp
from hjbdp import core
from problems import colsweep_problem, random_problem

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_calls = n_batched = n_stopped = n_nonfinite = 0
while time.time() < t_end:
    k = int(rng.integers(2, 6))
    if rng.random() < 0.7:            # column-sweep problems of one group axis (the grids may differ)
        gax = int(rng.choice([2, 3]))
        specs = []
        for _ in range(k):
            n = (int(rng.choice([60, 61, 120, 30, 45, 90])), int(rng.integers(4, 24)), int(rng.integers(4, 10)), int(rng.integers(4, 10)))
            specs.append(colsweep_problem(int(rng.integers(1 << 30)), n, nU=int(rng.integers(2, 12)), gax=gax, big=float(rng.choice([1.3, 2.7, 3.8])),
                                          small=float(rng.choice([0.2, 0.6])), cost="fast", levels=int(rng.integers(2, 6))))
    else:                             # small problems on the table kernel, one (dtype, D)
        D = int(rng.integers(2, 5))
        dtype = np.float32 if rng.random() < 0.5 else np.float64
        specs = []
        for _ in range(k):
            n = tuple(int(rng.integers(3, {2: 60, 3: 16, 4: 8}[D])) for _ in range(D))
            specs.append(random_problem(int(rng.integers(1 << 30)), n, (int(rng.integers(2, 7)),), dtype=dtype, spread=float(rng.choice([0.3, 0.8, 2.5]))))
    n_st = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 97, 130, int(rng.integers(1, 200))]))
    period = int(rng.choice([0, 0, 1, 3, 7, 10, 32, 50]))
    tol = 0.0
    single = bool(rng.random() < 0.5)
    ones = []
    for s in specs:
        with hjbdp.Backup(s) as bk:
            ones.append((bk.solve(n_st, monitor_period=period, monitor_tol=0.0, monitor_single=single), bk.info()["kernel_variant"]))
    if period > 0 and rng.random() < 0.8:     # a tolerance that stops some of them: between the monitor differences the problems end on
        es = sorted(abs(o["last_e"]) for o, _ in ones)
        tol = float(es[len(es) // 2]) * float(rng.choice([1.5, 30.0, 1e3]))
        ones = []
        for s in specs:
            with hjbdp.Backup(s) as bk:
                ones.append((bk.solve(n_st, monitor_period=period, monitor_tol=tol, monitor_single=single), bk.info()["kernel_variant"]))
    outs, wall, variants, sizes = core.solve_batch(specs, n_st, monitor_period=period, monitor_tol=tol, monitor_single=single)
    n_calls += 1
    n_batched += max(sizes) > 1
    for i, (o, (r, v)) in enumerate(zip(outs, ones)):
        n_stopped += bool(r["stopped_early"])
        if not np.all(np.isfinite(r["J"])):      # (a random problem whose values left the float range: labels of NaN totals are not defined)
            n_nonfinite += 1
            continue
        same = o["stages_done"] == r["stages_done"] and o["stopped_early"] == r["stopped_early"] and \
            np.array_equal(o["J"], r["J"]) and np.array_equal(o["idx"], r["idx"])
        if not same:
            print("   J differs at %d entries, labels at %d" % (np.count_nonzero(o["J"] != r["J"]), np.count_nonzero(o["idx"] != r["idx"])))
            print("MISMATCH problem %d of %d (variant %d, groups %s): n_stages %d period %d tol %g single %s: stages %d vs %d, early %s vs %s"
                  % (i, k, v, sizes, n_st, period, tol, single, o["stages_done"], r["stages_done"], o["stopped_early"], r["stopped_early"]), flush=True)
            sys.exit(1)
print("stress ok: %d calls (%d with a batch of two or more), %d early stops, %d problems left the float range (skipped)" % (n_calls, n_batched, n_stopped, n_nonfinite))
