// fetch_calib.hip - what does rocprofv3's FETCH_SIZE count on gfx950 for the access shapes of THIS library's kernels?
// MI355X_MICROARCH.md calibrates one shape (16 B per lane, streaming: FETCH_SIZE = 1/2 of the bytes) and says "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below reads each byte
// of a 2 GiB buffer exactly once: expected = 2 GiB.  Shapes: W = 4 / 8 / 16 bytes per lane, a wave's 64 lanes contiguous
// (64 W bytes); the wave's chunks in linear order ("stream") or in a scattered order ("rows": K10's row gathers are 4 B per
// lane at an arbitrary row, K3's window fill 8 B per lane).  Run under  rocprofv3 --pmc FETCH_SIZE --kernel-trace
// (tools/fetch_calib.sh);  build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename V, bool SCATTER>
__global__ void __launch_bounds__(256) k_read(const V *__restrict__ buf, uint64_t n_chunks, float *__restrict__ sink) {
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const unsigned lane = threadIdx.x & 63u;
    float acc = 0.f;
    for (uint64_t c = wave; c < n_chunks; c += n_waves) {
        // n_chunks is a power of two: an odd multiplier is a bijection of the chunk numbers
        const uint64_t chunk = SCATTER ? (c * 0x9E3779B97F4A7C15ull) & (n_chunks - 1) : c;
        const V v = buf[chunk * 64 + lane];
        if constexpr (sizeof(V) == 4) acc += __builtin_bit_cast(float, v);
        else if constexpr (sizeof(V) == 8) acc += v.x + v.y;
        else acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;       // never true: keeps the loads
}

template <typename V, bool SCATTER>
static void run(const void *buf, size_t bytes, float *sink) {
    const uint64_t n_chunks = bytes / (64 * sizeof(V));
    hipLaunchKernelGGL((k_read<V, SCATTER>), dim3(256 * 8), dim3(256), 0, nullptr, (const V *)buf, n_chunks, sink);
    hipDeviceSynchronize();
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    void *buf = nullptr;
    float *sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        run<uint32_t, false>(buf, bytes, sink);
        run<float2, false>(buf, bytes, sink);
        run<float4, false>(buf, bytes, sink);
        run<uint32_t, true>(buf, bytes, sink);
        run<float2, true>(buf, bytes, sink);
        run<float4, true>(buf, bytes, sink);
    }
    printf("expected bytes per kernel: %zu\n", bytes);
    return 0;
}
