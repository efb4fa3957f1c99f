// probe: pb::quot(x, d, RN(1/d)) against the true division x / d, bit for bit, on the device
// (ADVICE round 4).  Build and run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Ipyratbay_amd/csrc -Iinclude \
//         tools/quot_probe.hip -o /tmp/quot_probe && /tmp/quot_probe [rounds]
// Draws (x, d) with a counter-based generator: d over the callers' ranges (temperatures 1..1e5,
// mu 0.01..1, kT 1e-16..1e-11) and over 600 binades; x over 600 binades, both signs; plus the
// special values +-0, +-inf, NaN, subnormals for both operands.  Prints the number of pairs whose
// bits differ (specials are held to IEEE class + sign, subnormal results to 1 ulp).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "pb_common.h"

__device__ inline uint64_t mix(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__device__ inline double unit(uint64_t r)            // [1, 2)
{
    return __longlong_as_double((int64_t)((r >> 12) | 0x3ff0000000000000ull));
}

__global__ void k_quot(uint64_t seed, uint64_t *counts)
{
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t diff = 0, diff_ulp = 0, n = 0;
    for (int it = 0; it < 4096; it++) {
        const uint64_t r0 = mix(seed + id * 4096 + it), r1 = mix(r0), r2 = mix(r1);
        double d;
        switch (r2 & 3) {
        case 0: d = unit(r0) * exp2((double)((r2 >> 8) % 17));            break;   // temperatures
        case 1: d = unit(r0) * exp2(-(double)((r2 >> 8) % 7));            break;   // mu
        case 2: d = unit(r0) * exp2(-53.0 + (double)((r2 >> 8) % 17));    break;   // kT
        default: d = unit(r0) * exp2((double)((int)((r2 >> 8) % 600) - 300));
        }
        double x = unit(r1) * exp2((double)((int)((r2 >> 24) % 600) - 300));
        if (r2 & (1ull << 40))
            x = -x;
        const double inv = 1.0 / d;
        const double a = pb::quot(x, d, inv), b = x / d;
        n++;
        if (__double_as_longlong(a) != __double_as_longlong(b)) {
            diff++;
            const int64_t u = __double_as_longlong(a) - __double_as_longlong(b);
            if (u > 1 || u < -1)
                diff_ulp++;
        }
    }
    atomicAdd((unsigned long long *)&counts[0], (unsigned long long)n);
    atomicAdd((unsigned long long *)&counts[1], (unsigned long long)diff);
    atomicAdd((unsigned long long *)&counts[2], (unsigned long long)diff_ulp);
}

__global__ void k_special(const double *vals, int nv, uint64_t *counts)
{
    const int i = threadIdx.x / nv, j = threadIdx.x % nv;
    if (i >= nv)
        return;
    const double x = vals[i], d = vals[j];
    const double a = pb::quot(x, d, 1.0 / d), b = x / d;
    bool same = __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b);
    if (!same && fabs(b) < 4.5e-308) {             // subnormal quotient: 1 ulp allowed
        const int64_t u = __double_as_longlong(a) - __double_as_longlong(b);
        same = u >= -1 && u <= 1;
    }
    if (!same) {
        atomicAdd((unsigned long long *)&counts[3], 1ull);
        printf("special: %a / %a = %a, quot %a\n", x, d, b, a);
    }
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 4;
    uint64_t *c;
    hipMalloc(&c, 32);
    hipMemset(c, 0, 32);
    for (int r = 0; r < rounds; r++)
        k_quot<<<1024, 256>>>(0x1234567ull + (uint64_t)r * 0x100000000ull, c);
    const double vals[] = {0.0, -0.0, INFINITY, -INFINITY, NAN, 4.9e-324, -4.9e-324, 2.2e-308,
                           1e-300, 1.0, -3.0, 7.5e2, 1e300, 1.7e308, -1.7e308};
    const int nv = sizeof(vals) / sizeof(vals[0]);
    double *v;
    hipMalloc(&v, sizeof(vals));
    hipMemcpy(v, vals, sizeof(vals), hipMemcpyHostToDevice);
    k_special<<<1, nv * nv>>>(v, nv, c);
    hipDeviceSynchronize();
    uint64_t h[4];
    hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
    printf("quot_probe: %llu pairs, %llu differ from x / d (%llu by more than 1 ulp); "
           "%llu of %d special pairs differ\n", (unsigned long long)h[0], (unsigned long long)h[1],
           (unsigned long long)h[2], (unsigned long long)h[3], nv * nv);
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return (h[2] || h[3]) ? 1 : 0;
}
