#!/bin/bash
# Compile the packed2 float32 unit to assembly and print register use of the window / C2 modes (+ the min instruction mix of mode <6,5>).
# usage: bash tools/k3_asm.sh [extra hipcc flags]   -> /tmp/k3.s, /tmp/k3_65.s
cd /root/repo/optimal-control-dynamic-programming_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize "$@" -S --cuda-device-only stage_packed2w_f32.hip -o /tmp/k3.s 2>&1 | grep -v "hip-link" | head -5
grep -E "^\s+\.(vgpr_count|sgpr_spill_count|private_segment_fixed_size|vgpr_spill_count):|^\s+\.name:" /tmp/k3.s | paste - - - - - | sed 's/  */ /g' | grep -E "Li6ELi[2356]|Li4ELi5|Li3ELi[14]" | sed 's/_ZN3hjb16k_backup_packed2IfLi/D=/; s/EEEvPK.*Pv//'
L=$(grep -n "^_ZN3hjb16k_backup_packed2IfLi6ELi5" /tmp/k3.s | head -1 | cut -d: -f1)
sed -n "${L},\$p" /tmp/k3.s | awk '{print} /s_endpgm/{exit}' > /tmp/k3_65.s
echo "mode <6,5>: lines $(wc -l < /tmp/k3_65.s), v_min3 $(grep -c v_min3_f32 /tmp/k3_65.s), v_min $(grep -c 'v_min_f32' /tmp/k3_65.s), v_pk $(grep -c 'v_pk_' /tmp/k3_65.s), readlane $(grep -c v_readlane /tmp/k3_65.s)"
