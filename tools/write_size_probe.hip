// What does the WRITE_SIZE counter tally for 8-byte-per-lane stores?  (k_ext_staged
// shows WRITE_SIZE = 2.02 x the 64 MB it writes; k_records, which stores 16 bytes per lane, shows 1.0 x.)
// Two kernels write the SAME 64 MiB once, with 8-byte and with 16-byte stores per lane, to rows of
// 100 001 doubles (the pitch of C2's `ec`: rows are 8-byte aligned, not 128) and to 128-byte-aligned
// rows.  Run under `rocprofv3 --pmc WRITE_SIZE` (tools/profile_round.sh) and compare per-dispatch values.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_store8(double *out, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = 1.0;
}
__global__ void k_store16(double2 *out, long n2)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n2)
        out[i] = make_double2(1.0, 2.0);
}
// the gather's pattern: a wavefront writes 64 consecutive doubles of a row whose pitch is odd
__global__ void k_store8_rows(double *out, int nrows, int pitch)
{
    const int row = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < nrows && j < pitch)
        out[(long)row * pitch + j] = 1.0;
}
int main()
{
    const long n = 8L << 20;                      // 8 Mi doubles = 64 MiB
    double *d;
    hipMalloc(&d, (n + 1024) * 8);
    for (int rep = 0; rep < 3; rep++) {
        k_store8<<<(unsigned)(n / 256), 256>>>(d, n);
        k_store16<<<(unsigned)(n / 512), 256>>>((double2 *)d, n / 2);
        k_store8_rows<<<dim3((100001 + 255) / 256, 80), 256>>>(d, 80, 100001);     // 64.0 MB
        k_store8_rows<<<dim3(100000 / 256 + 1, 80), 256>>>(d, 80, 100000 - 100000 % 16);
    }
    hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
