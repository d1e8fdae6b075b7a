#!/bin/bash
# CPU-side sanitizer runs (SURVEY section 5 "ASan on CPU oracle").  HOST CODE ONLY - never on the GPU:
#   1. oracle/hjb_oracle.c (the checker) built with -fsanitize=address,undefined, under the oracle / host-solver CPU tests;
#   2. the host translation units of libhjbdp built by hipcc with -fsanitize=address,undefined -fno-gpu-sanitize (device code is
#      compiled as always), under tests/test_abi.py (argument validation, the flat builder, loader failures - no device needed).
# usage: bash tools/sanitize_cpu.sh [oracle|host|all]      (from the repository root; ~2 + ~4 minutes on 8 cores)
set -u
cd "$(dirname "$0")/.." || exit 1
what="${1:-all}"
out=build/san
mkdir -p "$out"
rc=0
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=87"     # (CPython itself leaks by design)
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=88"

if [ "$what" = oracle ] || [ "$what" = all ]; then
  echo "== oracle/hjb_oracle.c with -fsanitize=address,undefined"
  gcc -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -fopenmp -mfma -mavx2 -mf16c \
      -ffp-contract=off -Wall -Wno-unknown-pragmas -shared oracle/hjb_oracle.c -o "$out/libhjb_oracle_san.so" -lm || exit 1
  LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" HJB_ORACLE_LIB="$PWD/$out/libhjb_oracle_san.so" \
      python3 -m pytest tests/test_oracle_golden.py tests/test_host_solvers.py -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
  r=${PIPESTATUS[0]}; [ "$r" -ne 0 ] && rc=$r
fi

if [ "$what" = host ] || [ "$what" = all ]; then
  echo "== libhjbdp host units with -fsanitize=address,undefined -fno-gpu-sanitize"
  HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
  rt="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
  mkdir -p "$out/obj"
  ls optimal-control-dynamic-programming_amd/csrc/*.hip | xargs -P "$(nproc)" -I{} sh -c \
      'o='"$out"'/obj/$(basename {} .hip).o; [ "$o" -nt {} ] && [ -z "$(find optimal-control-dynamic-programming_amd/csrc include -newer "$o" -name "*.h*" | head -1)" ] || \
       '"$HIPCC"' --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-omit-frame-pointer -fsanitize=address,undefined -fno-gpu-sanitize -c {} -o "$o"' || exit 1
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan "$out"/obj/*.o -o "$out/libhjbdp_san.so" || exit 1
  LD_PRELOAD="$rt" HJBDP_LIB="$PWD/$out/libhjbdp_san.so" python3 -m pytest tests/test_abi.py -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
  r=${PIPESTATUS[0]}; [ "$r" -ne 0 ] && rc=$r
fi
echo "sanitize_cpu: exit $rc"
exit $rc
