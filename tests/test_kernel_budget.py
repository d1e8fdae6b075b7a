"""CPU test: the pos-att stage kernel's register budget, read from the code object hipcc produces here (no GPU).

K10 (kernels_colsweep.h) issues its corner-row gathers from inline asm and waits for them by count; between a gather and
its wait the compiler believes the destination register already holds its value, so it must never have a reason to
move or spill one - and the kernel's speed rests on six waves per SIMD (round 4; five before).  Both come down to: the headline
instantiations fit 80 VGPRs with at most one loop-invariant in scratch, touched outside the step body only."""
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
SRC = """
#include <hip/hip_runtime.h>
#include "%s/optimal-control-dynamic-programming_amd/csrc/kernels_colsweep.h"
using namespace hjb;
#define INST(TJ, GAX, NG) template __global__ void hjb::k_backup_colsweep<float, TJ, GAX, NG, true, true>( \\
    const DParams *, const DTabled *, const DColSweep *, const TJ *, TJ *, void *);
INST(float, 2, 5) INST(float, 3, 5) INST(_Float16, 2, 5) INST(float, 3, 4)
"""


@pytest.fixture(scope="module")
def cs_asm(tmp_path_factory):
    d = tmp_path_factory.mktemp("cs_budget")
    src = d / "cs_budget.hip"
    src.write_text(SRC % ROOT)
    asm = d / "cs_budget.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                        "-o", str(asm), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # ... and ASSEMBLED: -S prints inline asm operands without checking them (a VGPR handed to an "s" operand of the
    # counted-wait jump passes -S and fails here)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", "--cuda-device-only",
                        "-o", str(d / "cs_budget.o"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return asm.read_text()


def test_column_sweep_kernels_fit_six_waves(cs_asm):
    text = cs_asm
    kernels = re.findall(r"\.name:\s+(\S*k_backup_colsweep\S*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(kernels) == 4, [k[0] for k in kernels]
    # round 4: the usual cost shape runs at SIX waves per SIMD (80 registers, 25 KB of LDS per workgroup); at that bound the compiler
    # keeps ONE loop-invariant in scratch, stored before the step loop and reloaded where the parked results are written out (every
    # kCsFlush steps) - never between a gather and its wait (test_gathers_in_flight_are_never_touched below)
    for name, vgprs, spills in kernels:
        six = "IffLi2ELi5E" in name or "IffLi3ELi4E" in name       # float32 J, (group axis 2, five groups) / (axis 3, four groups): six waves
        assert int(vgprs) <= (80 if six else 96) and int(spills) <= (1 if six else 0), (name, vgprs, spills, kernels)
    assert all(int(x) <= 8 for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text))
    # scalar registers may overflow into lanes of a vector register (v_writelane / v_readlane: no memory involved) - a few
    assert all(int(x) <= 8 for x in re.findall(r"\.sgpr_spill_count:\s+(\d+)", text))
    assert text.count("scratch_") <= 3 * len(kernels), text.count("scratch_")


def _kernel_bodies(text):
    """{mangled name: [lines]} of every column-sweep kernel in the assembly."""
    out, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^(_ZN3hjb17k_backup_colsweep\S*):", ln)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is not None:
            if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
                cur = None
            else:
                cur.append(ln)
    return out


def _mentions(line, reg):
    """Does the instruction on `line` name VGPR number `reg` (alone or inside a v[a:b] range)?"""
    code = line.split(";")[0]
    if re.search(r"\bv%d\b" % reg, code):
        return True
    return any(int(a) <= reg <= int(b) for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code))


def test_gathers_in_flight_are_never_touched(cs_asm):
    """The structure K10's hand-issued gathers rely on, read off the code object (kernels_colsweep.h: gather_async /
    wait_gathers_n).  The corner-row gathers of the column loop are inline asm the compiler does not track: between a
    gather and the counted wait that retires it the destination register holds nothing yet, and the compiler - which
    believes it is already written - must not read, copy or reuse it.  The schedule: gathers are issued in batches
    separated by the hand-written waits (H1 of this step | wait | H0 of the next step | wait | ...), and a wait retires
    the batch issued BEFORE the previous wait; the full drain behind the stores of a write-out retires everything.  So:
    walking the loop body from a gather (cyclically - a batch crosses the back-edge), no instruction may name its
    destination before the second hand-written wait (or a full drain).  Also pinned: gathers per step of the headline
    shape (5 groups x 3 window knots x 2 rows, one load each)."""
    bodies = _kernel_bodies(cs_asm)
    assert len(bodies) == 4, list(bodies)
    for name, lines in bodies.items():
        _check_gathers(name, lines)
    # the checker itself: a copy of a gather's destination planted right behind a step gather must be caught
    name, lines = next(iter(bodies.items()))
    bar = next(j for j, l in enumerate(lines) if "s_barrier" in l)
    at = next(i for i, ln in enumerate(lines) if i > bar and re.search(r"global_load_dword\s+v\d+,", ln) and ";;#ASMSTART" in lines[i - 1])
    reg = re.search(r"global_load_dword\s+v(\d+),", lines[at]).group(1)
    end = next(i for i in range(at, len(lines)) if ";;#ASMEND" in lines[i])
    planted = lines[:end + 1] + ["\tv_mov_b32_e32 v200, v%s" % reg] + lines[end + 1:]
    with pytest.raises(AssertionError, match="in flight"):
        _check_gathers(name, planted)


def _check_gathers(name, lines):
    """Control-flow walk (not file order: hipcc lays blocks of the loop out of line).  From every hand-issued gather follow
    every path - fall-through, s_branch, both sides of an s_cbranch - until the gather is retired: the SECOND counted wait
    (a computed jump into a table of s_waitcnt: the asm block with s_setpc_b64) or a full drain (`s_waitcnt vmcnt(0)` the
    compiler wrote, or the kernel's own one-instruction asm drain behind the (re-)prime's gathers).  No instruction on the way
    may name the gather's destination register."""
    ins = []                       # (text, in_asm, asm_block_id)
    label_at = {}
    in_asm, blk = False, -1
    loop_headers = []              # instruction index behind every depth-1 loop header label
    for ln in lines:
        if re.match(r"^\.LBB\d+_\d+:.*Loop Header: Depth=1", ln):
            loop_headers.append(len(ins))
        if ";;#ASMSTART" in ln:
            in_asm, blk = True, blk + 1
            continue
        if ";;#ASMEND" in ln:
            in_asm = False
            continue
        code = ln.split(";")[0].rstrip()
        m = re.match(r"^(\.L[A-Za-z0-9_]+):", code)
        if m:
            label_at[m.group(1)] = len(ins)
            continue
        if not code.strip() or code.strip().startswith("."):
            continue
        ins.append((code.strip(), in_asm, blk if in_asm else -1))
    n = len(ins)
    blocks = {}
    for i, (c, a, b) in enumerate(ins):
        if a:
            blocks.setdefault(b, []).append(i)
    asm_drain = {idx[0] for idx in blocks.values() if len(idx) == 1 and re.fullmatch(r"s_waitcnt\s+vmcnt\(0\)", ins[idx[0]][0])}
    wait_jump_end = {}             # index of the s_setpc_b64 of a counted wait -> first instruction behind its asm block
    for idx in blocks.values():
        for i in idx:
            if ins[i][0].startswith("s_setpc_b64"):
                wait_jump_end[i] = idx[-1] + 1
    bar = next(i for i, (c, a, b) in enumerate(ins) if c.startswith("s_barrier"))
    hdr = max(h for h in loop_headers if h <= bar)            # the column loop = the depth-1 loop that holds the s_barrier
    # (the prologue's gathers, before the loop, are retired by the FIRST wait and are not walked here)
    gathers = [(i, int(re.search(r"global_load_(?:dword|ushort)\s+v(\d+),", c).group(1))) for i, (c, a, b) in enumerate(ins)
               if a and i >= hdr and re.search(r"global_load_(?:dword|ushort)\s+v\d+,", c)]
    n_groups = 5 if "Li5E" in name else 4
    # per step each row is gathered by ONE load; as many again in the (re-)prime block (round 5: all requested before one drain)
    assert len(gathers) == 2 * n_groups * 3 * 2, (name, len(gathers))
    for pos, reg in gathers:
        seen, todo = set(), [(pos + 1, 0)]
        while todo:
            i, waits = todo.pop()
            while True:
                if i >= n or (i, waits) in seen:
                    break
                seen.add((i, waits))
                c, a, b = ins[i]
                if c.startswith("s_endpgm"):
                    break
                if i in asm_drain or (not a and re.fullmatch(r"s_waitcnt\s+vmcnt\(0\)(\s+lgkmcnt\(\d+\))?", c)):
                    break                                   # a full drain: everything has landed
                if i in wait_jump_end:
                    waits += 1
                    if waits == 2:
                        break
                    i = wait_jump_end[i]
                    continue
                if not (a and re.match(r"s_(waitcnt|branch|getpc|add_u32|addc_u32)", c)):
                    assert not _mentions(c, reg), "%s: v%d is named while its gather is in flight: %s" % (name, reg, c)
                m = re.match(r"s_branch\s+(\S+)", c)
                if m and not a:
                    i = label_at[m.group(1)]
                    continue
                m = re.match(r"s_cbranch_\w+\s+(\S+)", c)
                if m and not a:
                    todo.append((label_at[m.group(1)], waits))
                i += 1
        assert seen, name


def test_packed2_occupancy_budgets():
    """K3's occupancy rests on register counts the compiler could quietly exceed: the C2 modes are held to 96 VGPRs (five
    waves per SIMD, `amdgpu_waves_per_eu`), the three-plane window modes of the 6-D grids must stay within 128 (the LDS they
    save buys a fourth workgroup per CU only then)."""
    import tempfile
    import __graft_entry__ as g
    got = {}
    with tempfile.TemporaryDirectory() as d:
        for unit in ("stage_packed2_f32.hip", "stage_packed2w_f32.hip"):      # plain / C2 modes; window modes (own flags)
            asm = "%s/p2.s" % d
            r = subprocess.run([HIPCC, *g.HIPCC_FLAGS, *g.UNIT_FLAGS.get(unit, []), "-S", "--cuda-device-only",
                                "-I%s/include" % ROOT, "-o", asm, "%s/optimal-control-dynamic-programming_amd/csrc/%s" % (ROOT, unit)],
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            text = open(asm).read()
            if unit == "stage_packed2_f32.hip":
                text_c2 = text
            got.update((m[0], (int(m[1]), int(m[2]))) for m in re.findall(
                r"\.name:\s+_ZN3hjb16k_backup_packed2IfLi(\d+ELi\d+)E\S*\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text))
    assert got["3ELi4"][0] <= 96 and got["3ELi1"][0] <= 96, got          # C2 modes: five waves per SIMD
    # ... with little in scratch (round 4: the 22 straight-line sweeps of the 21-control trip cost 11 spilled registers) and NOTHING of it
    # inside the two-step trip loop (loop depth 3: state chunk > o0 step > trip), where a scratch access per trip would cost what C3's
    # pass-loop experiment showed (profiles/r04_k3_experiments.log)
    # round 5: the per-state values the compiler kept in scratch are gone (grid-size divisions with the host's multipliers instead
    # of per-kernel reciprocals held in vector registers, 32-bit table offsets instead of per-lane 64-bit pointers, two per-state
    # offsets parked in LDS by hand): what is left is ONE uniform pointer of the LDS staging loops, touched before the state loop only
    assert got["3ELi4"][1] <= 4 and got["3ELi1"][1] <= 4, got
    body = text_c2[text_c2.index("_ZN3hjb16k_backup_packed2IfLi3ELi4"):]
    body = body[body.index("\n_ZN3hjb16k_backup_packed2IfLi3ELi4") + 1:] if "\n_ZN3hjb16k_backup_packed2IfLi3ELi4" in body else body
    body = body[:body.index("s_endpgm")]
    depth, deep = 0, []
    for ln in body.splitlines():
        m = re.match(r"^\.LBB\d+_\d+:\s*(;.*)?$", ln)
        if m:
            d = re.search(r"Depth=(\d+)", m.group(1) or "")
            depth = int(d.group(1)) if d else 0
        elif "scratch_" in ln and depth >= 2:
            deep.append(ln.strip())
    assert not deep, deep[:5]
    # ... and nothing of it inside the loop over the states (the last depth-1 loop of the kernel; the ones before it stage tables in LDS)
    state_loop = [m.start() for m in re.finditer(r"^\.LBB\d+_\d+:.*Loop Header: Depth=1", body, re.M)][-1]
    after = body[state_loop:]
    assert "Depth=2" in after and "scratch_" not in after, [ln for ln in after.splitlines() if "scratch_" in ln][:5]
    for key in ("6ELi5", "6ELi6"):                                          # three-plane window: four waves per SIMD, no spill
        assert got[key][0] <= 128 and got[key][1] == 0, got


def test_table_kernel_32bit_form_budget():
    """K7's 32-bit form (kernels_tabled.h::k_backup_tabled32) is worth its second set of instantiations only while it is the leaner
    kernel: nothing spilled in any instantiation, the 4-D float32 one (the reference's pos-att grid) within 80 VGPRs (six waves per
    SIMD), its corners loaded as axis-0 PAIRS and its table entries through global loads (eight-byte global loads: 8 + one per axis)."""
    import tempfile
    import __graft_entry__ as g
    with tempfile.TemporaryDirectory() as d:
        asm = "%s/tabled.s" % d
        r = subprocess.run([HIPCC, *g.HIPCC_FLAGS, *g.UNIT_FLAGS.get("stage_tabled.hip", []), "-S", "--cuda-device-only",
                            "-I%s/include" % ROOT, "-o", asm, "%s/optimal-control-dynamic-programming_amd/csrc/stage_tabled.hip" % ROOT],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        text = open(asm).read()
    got = dict((m[0], (int(m[1]), int(m[2]))) for m in re.findall(
        r"\.name:\s+_ZN3hjb17k_backup_tabled32I(\w+?Li\d)E\S*\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text))
    assert len(got) == 18, sorted(got)                          # float32 / binary16-stored / float64 x D = 1 .. 6
    assert all(sp == 0 for _, sp in got.values()), got
    assert got["ffLi4"][0] <= 80, got
    body = text[text.index("\n_ZN3hjb17k_backup_tabled32IffLi4E") + 1:]
    body = body[:body.index("s_endpgm")]
    assert len(re.findall(r"global_load_dwordx2", body)) >= 8 + 4, "corner pairs and table entries through global loads"


def test_uniwin_occupancy_budget():
    """K15 (kernels_uniwin.h, variant 4 modes 7 / 8) exists for its fifth wave per SIMD: every instantiation of the 256-state form
    within 96 VGPRs with NOTHING in scratch, the sweep's weights and control costs as SCALAR operands of the packed instructions
    (the register budget rests on it), and no scratch or LDS traffic inside the two-step trip (loop depth >= 3: chunk > o0 > trip)."""
    import tempfile
    import __graft_entry__ as g
    with tempfile.TemporaryDirectory() as d:
        asm = "%s/uw.s" % d
        r = subprocess.run([HIPCC, *g.HIPCC_FLAGS, "-S", "--cuda-device-only", "-I%s/include" % ROOT, "-o", asm,
                            "%s/optimal-control-dynamic-programming_amd/csrc/stage_uniwin_f32.hip" % ROOT], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        text = open(asm).read()
    got = dict((m[0], (int(m[1]), int(m[2]), int(m[3]))) for m in re.findall(
        r"\.name:\s+_ZN3hjb15k_backup_uniwinIf(Li\dELb[01]ELi\d+)E\S*\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?"
        r"\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text))
    assert sorted(got) == sorted("Li%dELb%dELi%d" % (dd, q, b) for (dd, q) in ((4, 0), (5, 0), (6, 0), (6, 1)) for b in (64, 256)), sorted(got)
    for key, (scratch, vgprs, spills) in got.items():
        assert scratch == 0 and spills == 0 and vgprs <= 96, (key, got[key])          # five waves per SIMD, nothing spilled
    body = text[text.index("\n_ZN3hjb15k_backup_uniwinIfLi6ELb1ELi256E") + 1:]
    body = body[:body.index("s_endpgm")]
    # the sweep: packed fma / add with a scalar register pair as an operand, operand selects written out
    assert len(re.findall(r"v_pk_fma_f32 v\[\d+:\d+\], s\[\d+:\d+\], v\[\d+:\d+\], v\[\d+:\d+\] op_sel", body)) >= 200
    assert len(re.findall(r"v_pk_add_f32 v\[\d+:\d+\], v\[\d+:\d+\], s\[\d+:\d+\] op_sel", body)) >= 200
    # the two-step trips are the INNERMOST depth-3 loops (chunk > o0 > trip; the slow per-backup path nests deeper): nothing but
    # vector / scalar arithmetic, lane reads and branches in them
    active, header, deep, n_trip_loops = False, None, [], 0
    lines = body.splitlines()
    i = 0
    while i < len(lines):
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", lines[i])
        if m:
            c = m.group(2) or ""
            while i + 1 < len(lines) and re.match(r"^\s+;", lines[i + 1]):        # the label's comment continues on the next lines
                i += 1
                c += lines[i]
            if "Inner Loop Header: Depth=3" in c:
                active, header = True, m.group(1).lstrip(".L")
                n_trip_loops += 1
            elif header and ("Header=%s Depth=3" % header) in c:
                active = True
            else:
                active = False
        elif active and re.search(r"\b(scratch_|ds_read|ds_write|global_load|global_store|s_load|buffer_)", lines[i]):
            deep.append(lines[i].strip())
        i += 1
    assert n_trip_loops >= 20, n_trip_loops                 # one per sweep shape (the loop nest is instantiated per shape)
    assert not deep, deep[:5]
