"""Derived quantities from a tools/pmc_summary.py JSON (one kernel): per-launch means of each counter plus
cycles, VALU issue utilisation (2 SIMD cycles per wave64 instruction, profiles/r02_valu_rate.json), wait fractions,
cache hit rates and HBM traffic.  usage: python tools/pmc_derive.py raw.json"""
import json, sys

raw = json.load(open(sys.argv[1]))
out = {}
for k, cs in raw.items():
    c = {n: v["mean_per_launch"] for n, v in cs.items()}
    d = {}
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0            # summed over the 8 XCDs
    if cyc:
        d["gpu_cycles_per_launch"] = cyc
        if "SQ_INSTS_VALU" in c:
            # 1024 SIMDs; a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (4 for packed f32: valu_rate)
            d["valu_issue_util_at_2cyc"] = c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cyc)
        if "SQ_INSTS_SALU" in c:
            d["salu_per_cu_cycle"] = c["SQ_INSTS_SALU"] / (256.0 * cyc)
        if "SQ_INSTS_VMEM_RD" in c:
            d["vmem_rd_instr_per_cu_cycle"] = c["SQ_INSTS_VMEM_RD"] / (256.0 * cyc)
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"):
            if n in c:
                d[n.lower() + "_frac_of_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]):
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c and c["TCP_TOTAL_CACHE_ACCESSES_sum"]:
        d["l1_hit_rate"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if "SQ_INSTS_VMEM_RD" in c and "TCP_TOTAL_CACHE_ACCESSES_sum" in c and c["SQ_INSTS_VMEM_RD"]:
        d["tcp_line_accesses_per_vmem_instr"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["SQ_INSTS_VMEM_RD"]
    if cyc and "TA_TA_BUSY_sum" in c:
        d["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / (256.0 * cyc)
    if "FETCH_SIZE" in c:
        d["hbm_fetch_bytes_per_launch"] = c["FETCH_SIZE"] * 1024.0      # rocprofv3 reports KiB; dword-per-lane reads count 1x
    if "WRITE_SIZE" in c:
        d["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0
    out[k] = {"counters_mean_per_launch": c, "derived": d}
print(json.dumps(out, indent=1))
