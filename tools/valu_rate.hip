// valu_rate.hip - microbenchmark: what does a wave64 vector instruction cost on a gfx950 SIMD?
//
// VERDICT r1 asked for this calibration before any "VALU issue utilisation" is quoted: is a wave64 `v_fma_f32` 2 or 4
// cycles of a SIMD, does `v_pk_fma_f32` cost the same slot, is `v_min3_f32` full rate, and do scalar instructions
// share the issue.  Every CU gets W workgroups of 256 threads (= W waves on each of its 4 SIMDs); each wave runs
// ITERS trips over a block of 16 independent instructions of one kind (inline asm on 16 separate accumulators, so
// neither the compiler nor dependences pace the stream).  Reported per op and W: cycles per wave-instruction per SIMD
// from s_memtime (shader clock, median over waves) and from the kernel's wall time at the clock the run sustained.
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/valu_rate.hip ; run: ./valu_rate  (prints a JSON document)
// Also checks which lane a DPP wave shift reads (used by the column-sweep stage kernel).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

enum Op { FMA = 0, PKFMA, MIN3, ADD, CNDMASK, DPPMOV, FMA_SALU, FMA_SGPR, PKADD, MUL, DPPMOV_ROW, SUBDPP_WAVE, SUBDPP_ROW, SUB, NOPS };
static const char *kOpName[] = {"v_fma_f32", "v_pk_fma_f32", "v_min3_f32", "v_add_f32", "v_cndmask_b32 (SGPR-pair mask)",
                                "v_mov_b32_dpp(wave_shl:1)", "v_fma_f32 + s_add_u32 (1:1)", "v_fma_f32 (SGPR operand)",
                                "v_pk_add_f32", "v_mul_f32", "v_mov_b32_dpp(row_shl:1)", "v_sub_f32_dpp(wave_shl:1)",
                                "v_sub_f32_dpp(row_shl:1)", "v_sub_f32"};

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void __launch_bounds__(256) k_rate(float *out, int iters, unsigned long long *cyc) {
    extern __shared__ float pad_lds[];            // sized by the host so that exactly W workgroups fit on a CU
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a[16];
    v2f p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = 1.0f + 1e-3f * (float)(tid + i);
        p[i] = v2f{a[i], a[i] + 0.5f};
    }
    float b = 0.9999f + 1e-9f * (float)tid, c = 1e-7f * (float)(tid & 7);
    v2f pb = v2f{b, b}, pc = v2f{c, c};
    float sb = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(0.99991f + 1e-9f * (float)blockIdx.x)));
    unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane(iters), s1 = 3u;
    const unsigned long long lanemask = __builtin_amdgcn_ballot_w64((threadIdx.x & 3) != 0);   // a lane mask in an SGPR pair
    if (iters < 0) pad_lds[threadIdx.x] = b;      // keeps the allocation
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (OP == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X)
#undef X
        } else if (OP == MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(X)
#undef X
        } else if (OP == PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
            REP16(X)
#undef X
        } else if (OP == PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            REP16(X)
#undef X
        } else if (OP == MIN3) {
#define X(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X)
#undef X
        } else if (OP == ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(lanemask));
            REP16(X)
#undef X
        } else if (OP == DPPMOV) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == DPPMOV_ROW) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (OP == SUBDPP_WAVE) {
#define X(i) asm volatile("v_sub_f32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == SUBDPP_ROW) {
#define X(i) asm volatile("v_sub_f32_dpp %0, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == SUB) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X)
#undef X
        } else if (OP == FMA_SALU) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_add_u32 %1, %1, %4" : "+v"(a[i]), "+s"(s0) : "v"(b), "v"(c), "s"(s1) : "scc");
            REP16(X)
#undef X
        } else if (OP == FMA_SGPR) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c));
            REP16(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)s0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    out[tid] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[2 * (tid >> 6)] = t1 - t0;            // shader clocks
        cyc[2 * (tid >> 6) + 1] = r1 - r0;        // 100 MHz reference clocks
    }
}

__global__ void k_dpp_probe(int *out) {
    const int lane = threadIdx.x;
    int shl = -1, shr = -1;
    asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(shl) : "v"(lane));
    asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(shr) : "v"(lane));
    out[lane] = shl;
    out[64 + lane] = shr;
}

// W workgroups resident per CU (LDS-limited), kRounds times as many dispatched: every SIMD runs W waves for the whole
// kernel, and dispatch imbalance averages out.  SIMD cycles per wave-instruction =
//   kernel time x sustained shader clock / (instructions per SIMD), with the clock from s_memtime / s_memrealtime.
constexpr int kRounds = 6;

template <int OP>
void run(int W, int iters, float *dout, unsigned long long *dcyc, int ncu, std::string &json) {
    const int blocks = ncu * W * kRounds;
    const size_t lds = (size_t)(160 * 1024) / W - 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rate<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(256), lds, 0, dout, iters / 8, dcyc);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(256), lds, 0, dout, iters, dcyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc((size_t)blocks * 4 * 2);
    CHECK(hipMemcpy(cyc.data(), dcyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk, wc;
    for (size_t i = 0; i < cyc.size(); i += 2) {
        clk.push_back((double)cyc[i] / (double)cyc[i + 1] * 100e6);
        wc.push_back((double)cyc[i]);
    }
    std::sort(clk.begin(), clk.end());
    std::sort(wc.begin(), wc.end());
    const double clock = clk[clk.size() / 2], wave_cycles = wc[wc.size() / 2];
    const double n_inst = 16.0 * iters;                                   // vector instructions per wave
    const double per_simd = n_inst * (double)blocks * 4.0 / (ncu * 4.0);  // vector instructions per SIMD
    const double cyc_per_inst = ms * 1e-3 * clock / per_simd;
    char buf[640];
    snprintf(buf, sizeof buf,
             "  {\"op\": \"%s\", \"waves_per_simd\": %d, \"iters\": %d, \"kernel_ms\": %.4f, \"shader_clock_GHz\": %.3f, "
             "\"simd_cycles_per_wave_instr\": %.3f, \"wave_cycles_per_own_instr\": %.3f, \"fp32_TFLOPs_if_fma\": %.1f},\n",
             kOpName[OP], W, iters, ms, clock * 1e-9, cyc_per_inst, wave_cycles / n_inst,
             per_simd * ncu * 4.0 * 64.0 * 2.0 * (OP == PKFMA ? 2.0 : 1.0) / (ms * 1e-3) * 1e-12);
    json += buf;
    fputs(buf, stderr);
    fflush(stderr);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

template <int OP>
void sweep(int iters, float *dout, unsigned long long *dcyc, int ncu, std::string &json) {
    for (int W : {1, 2, 4, 8}) run<OP>(W, iters, dout, dcyc, ncu, json);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    float *dout;
    unsigned long long *dcyc;
    CHECK(hipMalloc(&dout, (size_t)ncu * 8 * kRounds * 256 * 4));
    CHECK(hipMalloc(&dcyc, (size_t)ncu * 8 * kRounds * 4 * 2 * 8));
    std::string json = "{\"device\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(ncu) +
                       ", \"clock_khz\": " + std::to_string(prop.clockRate) + ",\n \"rates\": [\n";
    const int iters = 6000;
    sweep<FMA>(iters, dout, dcyc, ncu, json);
    sweep<MUL>(iters, dout, dcyc, ncu, json);
    sweep<ADD>(iters, dout, dcyc, ncu, json);
    sweep<PKFMA>(iters, dout, dcyc, ncu, json);
    sweep<PKADD>(iters, dout, dcyc, ncu, json);
    sweep<MIN3>(iters, dout, dcyc, ncu, json);
    sweep<CNDMASK>(iters, dout, dcyc, ncu, json);
    sweep<DPPMOV>(iters, dout, dcyc, ncu, json);
    sweep<DPPMOV_ROW>(iters, dout, dcyc, ncu, json);
    sweep<SUBDPP_WAVE>(iters, dout, dcyc, ncu, json);
    sweep<SUBDPP_ROW>(iters, dout, dcyc, ncu, json);
    sweep<SUB>(iters, dout, dcyc, ncu, json);
    sweep<FMA_SGPR>(iters, dout, dcyc, ncu, json);
    sweep<FMA_SALU>(iters, dout, dcyc, ncu, json);
    json.erase(json.size() - 2, 1);   // last comma
    json += " ],\n";
    int *dp;
    CHECK(hipMalloc(&dp, 128 * 4));
    hipLaunchKernelGGL(k_dpp_probe, dim3(1), dim3(64), 0, 0, dp);
    int hp[128];
    CHECK(hipMemcpy(hp, dp, sizeof hp, hipMemcpyDeviceToHost));
    char buf[256];
    snprintf(buf, sizeof buf,
             " \"dpp\": {\"wave_shl1_lane0_reads\": %d, \"wave_shl1_lane5_reads\": %d, \"wave_shl1_lane15_reads\": %d, "
             "\"wave_shl1_lane31_reads\": %d, \"wave_shl1_lane63_reads\": %d, \"wave_shr1_lane0_reads\": %d, "
             "\"wave_shr1_lane16_reads\": %d, \"wave_shr1_lane32_reads\": %d}\n}\n",
             hp[0], hp[5], hp[15], hp[31], hp[63], hp[64], hp[64 + 16], hp[64 + 32]);
    json += buf;
    fputs(json.c_str(), stdout);
    return 0;
}
