#!/bin/bash
# Builds tools/bench_ranks (the C++ one-process-per-GPU driver, SURVEY 8b iii) against the in-tree libhjbdp.so.
cd "$(dirname "$0")/.." || exit 1
L=optimal-control-dynamic-programming_amd/hjbdp
g++ -O2 -std=c++17 -Wall -Wextra tools/bench_ranks.cpp -Iinclude -L$L -lhjbdp -Wl,-rpath,'$ORIGIN/../'$L -Wl,-rpath-link,/opt/rocm/lib \
    -Wl,--allow-shlib-undefined -o tools/bench_ranks && ls -la tools/bench_ranks
