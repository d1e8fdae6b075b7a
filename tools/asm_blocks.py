#!/usr/bin/env python3
"""Basic blocks of one kernel's assembly listing with their vector / scalar / LDS / VMEM instruction counts and loop annotations.
usage: python tools/asm_blocks.py /tmp/k3_65.s [loop-header-label]   (only blocks inside that loop when given)"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else None
blocks, cur = [], None
for ln in lines:
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln)
    if m:
        cur = {"label": m.group(1), "note": (m.group(2) or ""), "v": 0, "s": 0, "lds": 0, "vmem": 0, "pk": 0, "br": []}
        blocks.append(cur)
        continue
    if cur is None:
        cur = {"label": "entry", "note": "", "v": 0, "s": 0, "lds": 0, "vmem": 0, "pk": 0, "br": []}
        blocks.append(cur)
    t = ln.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if t.startswith(";") and ("Loop" in t) and cur["v"] + cur["s"] == 0:
            cur["note"] += " " + t
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cur["v"] += 1
        if op.startswith("v_pk_"): cur["pk"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_"): cur["vmem"] += 1
    elif op.startswith("s_"):
        cur["s"] += 1
        if op.startswith("s_cbranch") or op == "s_branch": cur["br"].append(t.split()[-1])
for b in blocks:
    if want and want not in b["note"] and b["label"] != "." + want.lstrip("."):
        continue
    print(f'{b["label"]:12s} v={b["v"]:3d} (pk {b["pk"]:2d}) s={b["s"]:3d} lds={b["lds"]:2d} vmem={b["vmem"]:2d} -> {",".join(x.replace(".LBB","") for x in b["br"]):30s} {b["note"][:70]}')
