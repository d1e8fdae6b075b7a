"""Replay the case tools/stress_parity.py dumped at a mismatch (default: the seed-11 case kept as tests/golden/runaway_60x9x9x8.pkl; or the path given): every
applicable variant against the oracle, with the count of differing states.  usage: python tools/replay_stress_fail.py [pkl]"""
import os, pickle, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "optimal-control-dynamic-programming_amd"))
sys.path.insert(0, ROOT)
import hjbdp
from hjbdp import Term, _abi
from oracle import c_oracle

d = pickle.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "runaway_60x9x9x8.pkl"), "rb"))
idx_dtype = {"<class 'numpy.uint16'>": np.uint16, "<class 'numpy.uint8'>": np.uint8, "auto": "auto"}.get(d["idx_dtype"])
spec = hjbdp.ProblemSpec(d["knots"], d["m"], [[Term(dims, data) for dims, data in ts] for ts in d["next_terms"]],
                         [Term(dims, data) for dims, data in d["cost_terms"]], dtype=np.dtype(d["dtype"]), index_base=1,
                         j_storage=None if d["j_dtype"] == d["dtype"] else np.dtype(d["j_dtype"]), idx_dtype=idx_dtype,
                         table_dtype=np.float64 if d["tab64"] else None, cost_dtype=np.float64 if d.get("cost64") else None)
ref = c_oracle.sweep(_abi, spec, d["stages"], terminal=d["term"], nthreads=16, **d["mon"])
for v in (None, 0, 1, 2, 3, 4, 5, 6, 7):
    try:
        bk = hjbdp.Backup(spec, variant=v)
    except hjbdp.HjbError:
        continue
    with bk:
        out = bk.solve(d["stages"], terminal=d["term"], **d["mon"])
    same = (out["J"] == ref["J"]) | (np.isnan(out["J"].astype(np.float64)) & np.isnan(ref["J"].astype(np.float64)))
    print("variant", v, "J differs at", int((~same).sum()), "idx at", int((out["idx"] != ref["idx"]).sum()), "of", spec.nS)
