// Library-level entry points: version, error strings, device probe.
#include "vt_common.h"
#include <atomic>

extern "C" int vt_version(void) { return 102; }   // 101: global_terms argument of vt_vi_*; 102: the *_f32 entry points

extern "C" const char* vt_error_string(int code) {
    switch (code) {
        case VT_OK: return "ok";
        case VT_ERR_BAD_ARG: return "bad argument (null pointer, negative size or leading dimension too small)";
        case VT_ERR_UNSUPPORTED: return "shape not supported by the HIP kernels";
        case VT_ERR_WORKSPACE: return "workspace missing or too small (see *_workspace_bytes)";
        case VT_ERR_HIP: return "a HIP runtime call or kernel launch failed";
        case VT_ERR_NO_DEVICE: return "no HIP device";
        case VT_ERR_SCRATCH_LIMIT:
            return "input exceeds the static scratch of the no-alloc entry point (D <= 20448 on a GPU of <= 256 CUs)";
        default: return code > 0 ? "LAPACK-style info > 0" : "unknown error";
    }
}

extern "C" int vt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return VT_ERR_NO_DEVICE;
    return n;
}

int vt_num_cu() {
    static std::atomic<int> cache[VT_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VT_MAX_DEVICES) return 256;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    v = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    cache[dev].store(v, std::memory_order_relaxed);
    return v;
}

int vt_allow_big_lds(const void* func, VtLdsAttr& flag, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VT_MAX_DEVICES) return VT_ERR_HIP;
    if (flag.done[dev]) return VT_OK;
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return VT_ERR_HIP;
    flag.done[dev] = true;
    return VT_OK;
}
