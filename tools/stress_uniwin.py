"""Randomised GPU-vs-oracle parity stress of K15 (k_backup_uniwin, kernels_uniwin.h): random rate-shared shapes (D = 4 .. 6, 11 or 12
inner controls, outer control counts, gains from sub-cell to several cells - points that leave the two-cell windows take the kernel's
plain path -, uniform and non-uniform knots, float32 / float16 cost-to-go storage), whole grids and slabs, the walk options (claimed /
static, tile shapes, 64- / 256-state chunks).  Two stages from a random terminal cost; every value and label must equal the C oracle's.
usage: python tools/stress_uniwin.py [seconds=120] [seed=0]"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "optimal-control-dynamic-programming_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hjbdp
from hjbdp import _abi
from oracle import c_oracle
from problems import rate_shared_problem, random_terminal

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_prob = n_runs = n_k15 = n_slowpts = 0
while time.time() < t_end:
    NP = int(rng.integers(1, 4))                                   # state-only axes: D = 4 .. 6
    while True:
        n_so = tuple(int(rng.integers(2, {1: 400, 2: 24, 3: 9}[NP] + 1)) for _ in range(NP))
        if 128 <= int(np.prod(n_so)) <= 700:
            break
    n_rates = tuple(int(rng.integers(3, 7)) for _ in range(3))
    m = (int(rng.integers(1, 13)), int(rng.integers(1, 13)), int(rng.choice([11, 12])))
    gain = tuple(float(rng.choice([0.02, 0.2, 0.5, 0.6, 0.9, 1.4])) for _ in range(3))
    nonuniform = bool(rng.random() < 0.4)
    j_storage = np.float16 if rng.random() < 0.2 else None
    seed = int(rng.integers(1 << 30))
    spec = rate_shared_problem(seed, n_so, n_rates, m=m, gain=gain, nonuniform=nonuniform, so_move=float(rng.choice([0.3, 0.7, 0.95])),
                               j_storage=j_storage)
    if spec.nS * spec.nU > 6e8:
        continue
    term = random_terminal(spec, seed % 1000)
    ref = c_oracle.sweep(_abi, spec, 2, terminal=term, keep_J=True, keep_idx=True)
    n_prob += 1
    desc = "n_so=%s n_rates=%s m=%s gain=%s nonuniform=%s j_storage=%s seed=%d" % (n_so, n_rates, m, gain, nonuniform, j_storage, seed)
    with hjbdp.Backup(spec) as bk:
        if bk.info()["kernel_variant"] != 4 or bk.get_option("uniwin_ok") != 1:
            continue
        bk.set_option("uniwin", 1)                                 # also where the host's 2 % rule would not pick it: the plain path at work
        if bk.get_option("packed2_mode") not in (7, 8):
            continue
        n_k15 += 1
        n_slowpts += bk.get_option("uniwin_slow_points")
        forms = [("default", {})]
        forms.append(("static walk", {"uw_claim": 0}) if rng.random() < 0.5 else ("64-state chunks", {"uw_block": 64}))
        forms.append(("tile", {"uw_tile": int(rng.integers(0, 4)) + 8 * int(rng.integers(0, 3)) + 64 * int(rng.integers(0, 3))}))
        for name, opts in forms:
            for k, v in opts.items():
                bk.set_option(k, v)
            out = bk.solve(2, terminal=term, keep_J=True, keep_idx=True)
            n_runs += 1
            for key in ("J_stages", "idx_stages"):
                bad = np.flatnonzero(out[key] != ref[key])
                if bad.size:
                    print("MISMATCH (%s, %s) %s: %d entries, first %s" % (name, key, desc, bad.size, bad[:6]), flush=True)
                    sys.exit(1)
    if rng.random() < 0.35 and n_rates[2] >= 4:                    # a slab of the last axis with one halo plane each way where there is one
        nl = n_rates[2]
        b = int(rng.integers(0, nl - 1))
        e = int(rng.integers(b + 1, nl + 1))
        lo, hi = min(1, b), min(1, nl - e)
        inner = spec.nS // nl
        J2 = ref["J_stages"].reshape(spec.nS, 2, order="F")
        I2 = ref["idx_stages"].reshape(spec.nS, 2, order="F")
        Jn = J2[:, 1].reshape(inner, nl, order="F")               # the second-to-last stage's values are the last stage's input
        try:
            with hjbdp.Backup(spec, slab=(b, e, lo, hi)) as bk:
                if bk.info()["kernel_variant"] == 4 and bk.get_option("uniwin_ok") == 1:
                    bk.set_option("uniwin", 1)
                    Jo, io = bk.backup_stage(np.asfortranarray(Jn[:, b - lo:e + hi]).reshape(-1, order="F"))
                    n_runs += 1
                    ok = np.array_equal(Jo.reshape(inner, -1, order="F")[:, lo:lo + e - b], J2[:, 0].reshape(inner, nl, order="F")[:, b:e]) and \
                        np.array_equal(io, I2[:, 0].reshape(inner, nl, order="F")[:, b:e].reshape(-1, order="F"))
                    if not ok:
                        print("MISMATCH (slab %d:%d halo %d/%d) %s" % (b, e, lo, hi, desc), flush=True)
                        sys.exit(1)
        except hjbdp.HjbError as ex:
            if ex.status != _abi.HJB_E_HALO:                       # (a query that leaves a one-plane halo is reported, never silent)
                raise
print("stress ok: %d problems, %d on K15 (%d slow points in all), %d GPU runs" % (n_prob, n_k15, n_slowpts, n_runs))
