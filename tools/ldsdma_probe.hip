#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// probe: LDS-DMA buffer loads -- out-of-range lanes, 4- and 16-byte forms
template <int SZ>
__global__ void k(const double *src, int nbytes, int shift_bytes, double *out)
{
    __shared__ __attribute__((aligned(16))) double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64)
        lds[i] = -7.0;                       // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nbytes, 0x00020000);
    // lane l reads SZ bytes at (l*SZ - shift); LDS dest = lds base + lane*SZ
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (SZ == 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds, 4,
                                                 (int)threadIdx.x * 4 - shift_bytes, 0, 0, 0);
    else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds, 16,
                                                 (int)threadIdx.x * 16 - shift_bytes, 0, 0, 0);
#endif
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
    __syncthreads();
    for (int i = threadIdx.x; i < 160; i += 64)
        out[i] = lds[i];
}
int main()
{
    std::vector<double> h(256);
    for (int i = 0; i < 256; i++) h[i] = 100 + i;
    double *d, *o; hipMalloc(&d, 2048); hipMalloc(&o, 160 * 8);
    hipMemcpy(d, h.data(), 2048, hipMemcpyHostToDevice);
    std::vector<double> r(160);
    auto show = [&](const char *t) {
        hipMemcpy(r.data(), o, 160 * 8, hipMemcpyDeviceToHost);
        printf("%s\n", t);
        for (int i = 0; i < 140; i++) printf("%g%c", r[i], (i % 20 == 19) ? '\n' : ' ');
        printf("\n");
    };
    // 4-byte form: 64 lanes x 4 B = 32 doubles; window of 10 doubles (80 B), shifted by 3 doubles (24 B)
    k<4><<<1, 64>>>(d, 80, 24, o); hipDeviceSynchronize(); show("SZ=4 nbytes=80 shift=24: expect lds[0..2]=0?, lds[3..12]=100..109, then 0 or sentinel up to 31");
    // 16-byte form: 64 lanes x 16 B = 128 doubles; window 11 doubles (88 B, odd), shift 3 doubles (24 B: lane pairs straddle both edges)
    k<16><<<1, 64>>>(d, 88, 24, o); hipDeviceSynchronize(); show("SZ=16 nbytes=88 shift=24: lane0 covers doubles -3,-2; lane1 -1,0 (straddles); ...");
    k<16><<<1, 64>>>(d, 88, 0, o); hipDeviceSynchronize(); show("SZ=16 nbytes=88 shift=0: last lane5 covers doubles 10,11 (11 is out)");
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
