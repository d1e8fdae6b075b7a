"""CPU test: the pos-att stage kernel's register budget, read from the code object hipcc produces here (no GPU).

K10 (kernels_colsweep.h) issues its corner-row gathers from inline asm and waits for them by count; between a gather and
its wait the compiler believes the destination register already holds its value, so it must never have a reason to
move or spill one - and the kernel's speed rests on five waves per SIMD.  Both come down to: the headline
instantiations fit 96 VGPRs without a spill and without scratch memory."""
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


def test_column_sweep_kernels_fit_five_waves_without_spills(tmp_path):
    src = tmp_path / "cs_budget.hip"
    src.write_text(SRC % ROOT)
    asm = tmp_path / "cs_budget.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                        "-o", str(asm), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    text = asm.read_text()
    kernels = re.findall(r"\.name:\s+(\S*k_backup_colsweep\S*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)
    assert len(kernels) == 4, [k[0] for k in kernels]
    for name, vgprs, spills in kernels:
        assert int(vgprs) <= 96 and int(spills) == 0, (name, vgprs, spills)
    assert all(int(x) == 0 for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text))
    # scalar registers may overflow into lanes of a vector register (v_writelane / v_readlane: no memory involved) - a few
    assert all(int(x) <= 8 for x in re.findall(r"\.sgpr_spill_count:\s+(\d+)", text))
    assert "scratch_" not in text
