"""What storage precision does the pos-att sweep (C4's problem, 120^4 x 9) need to run its horizon?
The stage kernel runs in float32 on a float32 J; after every stage J is ROUNDED on the device to the storage format under
test - which is exactly what a narrower storage type does (HJB_F16S = float32 arithmetic on widened values, one rounding
on store) - and the sweep is compared with the float32 one.  Formats: 'm<p>' = p explicit mantissa bits, float32's exponent
(m10 ~ binary16 without its range limits, m7 ~ bfloat16), 'f16' = IEEE binary16, 'q16' = 16-bit fixed point of step
96 / 65536, 'f16+w<k>' / 'f16+all<k>' = binary16 except the k outermost knots at both ends of the w axis / of every axis,
which stay float32 (a mixed layout's numerics).
usage: python tools/c5_precision.py [stages=200] [formats ...]"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "optimal-control-dynamic-programming_amd")); sys.path.insert(0, ROOT)
import torch
import hjbdp, bench

stages = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fmts = sys.argv[2:] or ["f16", "m10", "m12", "m14", "m16", "m18", "q16", "f16+w2", "f16+all2"]
n = int(os.environ.get("GRID_N", "120"))
spec, _ = bench.build_spec("c4", n=n)
dev = torch.device("cuda:0")
shape = tuple(reversed(spec.n))            # torch row-major view of the column-major grid: (v, w, theta, x)
checks = [s for s in (10, 25, 50, 100, 150, 200, 400) if s <= stages]


def rounder(fmt):
    if fmt == "f32":
        return lambda J: J
    if fmt == "f16":
        return lambda J: J.half().float()
    if fmt == "q16":
        return lambda J: torch.clamp(torch.round(J * (65536.0 / 96.0)), -32768 * 2.0, 65535.0) * (96.0 / 65536.0)
    if fmt[0] == "m":
        p = int(fmt[1:]); sh = 23 - p
        def f(J):
            i = J.view(torch.int32)
            i = (i + ((1 << (sh - 1)) - 1) + ((i >> sh) & 1)) & ~((1 << sh) - 1)      # round to nearest even on the kept bits
            return i.view(torch.float32)
        return f
    if fmt.startswith("f16+"):
        which, k = (fmt[4:5], int(fmt[5:])) if fmt[4] == "w" else ("all", int(fmt[7:]))
        def f(J):
            G = J.view(shape)
            H = G.half().float()
            axes = [1] if which == "w" else [0, 1, 2, 3]          # torch dim 1 = the w axis (grid axis 2)
            for a in axes:
                for sl in (slice(0, k), slice(G.shape[a] - k, G.shape[a])):
                    ix = [slice(None)] * 4; ix[a] = sl
                    H[tuple(ix)] = G[tuple(ix)]
            return H.view(-1)
        return f
    raise ValueError(fmt)


ref = {}
with hjbdp.Backup(spec) as bk:
    assert bk.info()["kernel_variant"] == 7
    for fmt in ["f32"] + fmts:
        rd = rounder(fmt)
        J = [torch.zeros(spec.nS, dtype=torch.float32, device=dev) for _ in range(2)]
        idx = torch.zeros(spec.nS, dtype=torch.uint8, device=dev)
        cur, line = 0, []
        for s in range(1, stages + 1):
            bk.backup_stage_device(J[cur], J[1 - cur], idx)
            cur = 1 - cur
            J[cur].copy_(rd(J[cur]))
            if s in checks:
                x = J[cur]
                if fmt == "f32":
                    ref[s] = x.clone()
                    line.append("%d: [%.4g, %.4g]" % (s, float(x.min()), float(x.max())))
                else:
                    fin = bool(torch.isfinite(x).all())
                    err = float((x - ref[s]).abs().max() / ref[s].max()) if fin else float("inf")
                    line.append("%d: min %.4g max %.4g relerr %.2e" % (s, float(x.min()), float(x.max()), err))
        bk.check_device_status()
        print("%-9s %s" % (fmt, " | ".join(line)), flush=True)
