// Dev check: semantics of buffer_load_dwordx4 ... lds on gfx950 (lane placement, OOB zero fill, exec mask).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* X, double* out, int n) {
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) lds[i] = -7.0;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, n * 8, 0x00020000);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < 56)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 112 * w), 16, lane * 16 + w * 800, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = lds[i];
}
int main() {
    int n = 150;  // row 0: elements 0..99 (+ spill into 100..111 = next row head); row 1 (w=1): 100..149 valid, then OOB
    std::vector<double> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = i + 1;
    double *X, *out;
    hipMalloc(&X, 8192); hipMalloc(&out, 4096);
    hipMemcpy(X, h.data(), 8192, hipMemcpyHostToDevice);
    k<<<1, 128, 4096>>>(X, out, n);
    std::vector<double> o(512);
    hipMemcpy(o.data(), out, 4096, hipMemcpyDeviceToHost);
    for (int i = 0; i < 240; ++i) printf("%g%c", o[i], (i % 16 == 15) ? '\n' : ' ');
    printf("err=%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
