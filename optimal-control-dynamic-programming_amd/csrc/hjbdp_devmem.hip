// hjbdp_devmem.hip - device-buffer helpers of libhjbdp for hosts without a HIP binding of their own.
// gfx950 (MI355X) only; no CPU fallback - without a HIP device every compute entry point returns HJB_E_DEVICE.
#include "hjbdp_host.h"
#include "kernels_devmem.h"

using namespace hjbhost;

extern "C" {

// ---- device-buffer helpers -------------------------------------------------------------------------------------------
// hjb_backup_stage_device runs on buffers the caller owns.  A host without a HIP binding of its own (MATLAB, plain C)
// gets them here: allocation, copies, free memory, a separable fill and a gather - enough to drive grids that never
// exist on the host (C3: 51^6 states, 70 GB per buffer).
int32_t hjb_device_malloc(int32_t device, int64_t bytes, void **out) {
    if (!out || bytes < 0) return fail(nullptr, HJB_E_INVALID, "hjb_device_malloc: bad argument");
    *out = nullptr;
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipSetDevice(%d) failed", device);
    void *d = nullptr;
    const hipError_t e = hipMalloc(&d, (size_t)std::max<int64_t>(bytes, 16));
    if (e != hipSuccess) return fail(nullptr, HJB_E_NOMEM, "hipMalloc of %lld bytes: %s", (long long)bytes, hipGetErrorString(e));
    *out = d;
    return HJB_OK;
}

int32_t hjb_device_free(int32_t device, void *p) {
    if (!p) return HJB_OK;
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipSetDevice(%d) failed", device);
    (void)hipDeviceSynchronize();
    return hipFree(p) == hipSuccess ? HJB_OK : fail(nullptr, HJB_E_DEVICE, "hipFree failed");
}

int32_t hjb_device_mem_info(int32_t device, int64_t *free_bytes, int64_t *total_bytes) {
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipSetDevice(%d) failed", device);
    size_t f = 0, t = 0;
    if (hipMemGetInfo(&f, &t) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipMemGetInfo failed");
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return HJB_OK;
}

int32_t hjb_device_copy(int32_t device, void *dst, const void *src, int64_t bytes, int32_t kind) {
    if (!dst || !src || bytes < 0) return fail(nullptr, HJB_E_INVALID, "hjb_device_copy: bad argument");
    const hipMemcpyKind k = kind == HJB_COPY_H2D ? hipMemcpyHostToDevice : kind == HJB_COPY_D2H ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (kind < HJB_COPY_H2D || kind > HJB_COPY_D2D) return fail(nullptr, HJB_E_INVALID, "hjb_device_copy: kind %d", kind);
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipSetDevice(%d) failed", device);
    const hipError_t e = hipMemcpy(dst, src, (size_t)bytes, k);
    return e == hipSuccess ? HJB_OK : fail(nullptr, HJB_E_DEVICE, "hipMemcpy: %s", hipGetErrorString(e));
}

int32_t hjb_device_fill_separable(hjb_handle hh, const void *const *vecs, void *dJ, void *stream) {
    Handle *h = (Handle *)hh;
    if (!h || !vecs || !dJ) return fail(h, HJB_E_INVALID, "null argument");
    // a slab handle: its haloed buffer = planes [slab_begin - halo_lo, slab_end + halo_hi) of the GLOBAL separable function
    const bool slab = h->j_elems != h->n_owned || h->plane0 != 0 || h->nplanes != h->prob.n[h->hp.D - 1];
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    HIP_TRY(h, hipSetDevice(h->device));
    const int D = h->hp.D;
    const size_t tsz = h->dtype == HJB_F64 ? 8 : 4;
    DSeparable S{};
    std::vector<void *> tmp;
    for (int a = 0; a < D; ++a) {
        if (!vecs[a]) return fail(h, HJB_E_INVALID, "vecs[%d] is null", a);
        void *d = nullptr;
        if (hipMalloc(&d, (size_t)h->prob.n[a] * tsz) != hipSuccess) { for (void *t : tmp) (void)hipFree(t); return fail(h, HJB_E_NOMEM, "fill vectors"); }
        tmp.push_back(d);
        if (hipMemcpy(d, vecs[a], (size_t)h->prob.n[a] * tsz, hipMemcpyHostToDevice) != hipSuccess) { for (void *t : tmp) (void)hipFree(t); return fail(h, HJB_E_DEVICE, "fill vectors"); }
        S.v[a] = d;
        S.n[a] = h->prob.n[a];
    }
    S.D = D;
    S.total = h->n_owned;
    if (slab) {
        S.v[D - 1] = (const char *)S.v[D - 1] + (size_t)h->plane0 * tsz;
        S.n[D - 1] = h->nplanes;
        S.total = h->j_elems;
    }
    const unsigned grid = (unsigned)std::min<int64_t>((S.total + 255) / 256, 256 * 64);
    if (h->dtype == HJB_F16S) hipLaunchKernelGGL((k_fill_separable<float, _Float16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, S, (_Float16 *)dJ);
    else if (h->dtype == HJB_F32) hipLaunchKernelGGL((k_fill_separable<float, float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, S, (float *)dJ);
    else hipLaunchKernelGGL((k_fill_separable<double, double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, S, (double *)dJ);
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);      // the vectors are freed below
    for (void *t : tmp) (void)hipFree(t);
    if (e != hipSuccess) return fail(h, HJB_E_DEVICE, "hjb_device_fill_separable: %s", hipGetErrorString(e));
    return HJB_OK;
}

int32_t hjb_device_gather(int32_t device, const void *d_src, int32_t elem_bytes, const int64_t *sel, int64_t n_sel, void *out) {
    if (!d_src || !sel || !out || n_sel < 0) return fail(nullptr, HJB_E_INVALID, "hjb_device_gather: bad argument");
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) return fail(nullptr, HJB_E_INVALID, "hjb_device_gather: elem_bytes %d", elem_bytes);
    if (n_sel == 0) return HJB_OK;
    std::shared_lock<std::shared_mutex> lk(g_capture_mu);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, HJB_E_DEVICE, "hipSetDevice(%d) failed", device);
    void *dsel = nullptr, *dout = nullptr;
    hipError_t e = hipMalloc(&dsel, (size_t)n_sel * 8);
    if (e == hipSuccess) e = hipMalloc(&dout, (size_t)n_sel * elem_bytes);
    if (e == hipSuccess) e = hipMemcpy(dsel, sel, (size_t)n_sel * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gather_bytes, dim3((unsigned)std::min<int64_t>((n_sel + 255) / 256, 65536)), dim3(256), 0, nullptr,
                           (const unsigned char *)d_src, elem_bytes, (const int64_t *)dsel, n_sel, (unsigned char *)dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n_sel * elem_bytes, hipMemcpyDeviceToHost);
    if (dsel) (void)hipFree(dsel);
    if (dout) (void)hipFree(dout);
    return e == hipSuccess ? HJB_OK : fail(nullptr, HJB_E_DEVICE, "hjb_device_gather: %s", hipGetErrorString(e));
}

}  // extern "C"
