"""What ONE rank of an N-GPU strong-scaling run of C4 does per stage, emulated on ONE GPU with the library's own RCCL path:
a middle rank's slab (120 / N planes of v), hjb_rank_sweep with the RCCL loopback communicator (the rank is both of its
neighbours: real ncclSend / ncclRecv of its two halo planes, 6.9 MB each, on the transfer stream beside the interior planes'
kernel) plus an INJECTED delay of 0 / 20 / 40 / 80 us per exchange standing in for xGMI latency and the neighbour's skew.
Prints ms per stage and the strong-scaling efficiency it would imply, T1 / (N * T_rank) - EMULATED, ONE GPU: no multi-GPU
hardware was available; the driver's SCALE run is the measurement.
usage: python tools/emulate_ranks.py [steps=200] [workload=c4|c3]      (c3: BASELINE configs[2], 51^6 sharded along w3 - a middle rank of
eight holds 6 planes + 2 halo planes of 1.38 GB; steps ~4; the one-rank reference run needs 176 GB of free HBM)"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "optimal-control-dynamic-programming_amd")); sys.path.insert(0, ROOT)
import hjbdp, bench
from hjbdp import _abi

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
workload = sys.argv[2] if len(sys.argv) > 2 else "c4"
spec, name = bench.build_spec(workload, n=51) if workload == "c3" else bench.build_spec("c4")
warm = 1 if workload == "c3" else 20
lib = hjbdp.load_library()
inner = spec.nS // spec.n[-1]


def sweep_ms(rank, world, overlap, delay_us, loopback):
    rk = hjbdp.core.RankSlab(spec, 0, rank, world, overlap=overlap)
    if loopback:
        rk.set_option("comm_loopback", 1)
        rk.set_option("xfer_delay_us", delay_us)
        uid = (C.c_char * 128)()
        assert lib.hjb_rank_comm_unique_id(uid) == 0, lib.hjb_rank_last_error(None)
        assert lib.hjb_rank_comm_init(rk._r, uid) == 0, lib.hjb_rank_last_error(rk._r)
    planes = rk.end - rk.begin + rk.halo_lo + rk.halo_hi
    nb = inner * planes * 4
    with hjbdp.DeviceBuffer(nb) as d0, hjbdp.DeviceBuffer(nb) as d1, hjbdp.DeviceBuffer(inner * (rk.end - rk.begin) * rk.idx_bytes) as dI:
        if workload == "c3":                         # (a 70 GB host array of zeros is not an option: separable zeros built on the device)
            rk.fill_separable([np.zeros(n, dtype=np.float32) for n in spec.n], d0)
            rk.fill_separable([np.zeros(n, dtype=np.float32) for n in spec.n], d1)
        else:
            z = np.zeros(inner * planes, dtype=np.float32)
            d0.upload(z); d1.upload(z)
        done, early, in0, ms = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double()
        out = []
        for n_st in (warm, steps):                    # warm-up, then the timed sweep
            st = lib.hjb_rank_sweep(rk._r, n_st, 0, 0.0, int(d0), int(d1), int(dI), None, C.byref(done), C.byref(early), C.byref(in0), C.byref(ms))
            assert st == 0, lib.hjb_rank_last_error(rk._r)
            out.append(ms.value / n_st)
    info = (rk.split, rk.halo_lo, rk.halo_hi, rk.end - rk.begin)
    rk.close()
    return out[1], info


t1, _ = sweep_ms(0, 1, True, 0, False)
print("%s" % name)
print("EMULATED ON ONE GPU (RCCL loopback of a middle rank + injected delay per exchange); not a multi-GPU measurement")
print("whole grid, one rank: %.4f ms per stage" % t1)
print("%-4s %-9s %-10s %-28s %s" % ("N", "planes", "delay us", "ms per stage (rank)", "implied efficiency T1 / (N T_rank)"))
for N in ((8,) if workload == "c3" else (2, 4, 8)):
    for overlap in (True, False):
        for delay in ((0, 1000, 10000) if workload == "c3" else (0, 20, 40, 80)):
            t, (split, hlo, hhi, own) = sweep_ms(N // 2, N, overlap, delay, True)
            print("%-4d %-9d %-10d %-28s %.3f" % (N, own, delay, "%.4f (%s, halo %d/%d)" % (t, "interior + strips beside the transfer" if split else "exchange, then one kernel", hlo, hhi),
                                                 t1 / (N * t)), flush=True)
