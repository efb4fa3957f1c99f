// Co-add grouping of a line list on the device (the static pre-pass of pb_lines_create).
//
// Reference: the sequential loop of _extcoeff.extinction, src_c/_extcoeff.c:230-262 -- for every
// in-range line in file order: nearest fine-grid index iown (:243-245), then the FOLLOWING lines
// of the same isotope with lwn <= own[last] and |lwn - own[iown]| < ownstep are co-added to it
// (:248-262) and skipped.  Which lines lead a group depends only on (lwn, lID, own): it is done
// once per line list.
//
// The greedy scan is a chain (the line after a group's last member leads the next group), made
// parallel with forward pointers:
//   k_classify   per line: in range?, iown, the fine-grid wavenumber it snaps to
//   k_next       per line: the end of the group it WOULD lead (bisection on the sorted wavenumbers
//                of its isotope) and nxt = the next in-range line at or after that end
//   k_block_exit per block of 1024 lines: pointer doubling inside the block -> for every line the
//                first line of its chain that lies beyond the block
//   k_chain      one thread: from the first in-range line, block to block through the exits
//                (n/1024 dependent steps), noting where the chain enters every block
//   k_mark       one thread per block: follows nxt from the block's entry and marks the leaders
//   scan + k_emit: leaders compacted in file order = (isotope, position) order.
// Requires the TLI invariant: lines sorted by isotope, then wavenumber (checked on the device;
// otherwise the host loop of pb_lines_create does the work).
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <vector>

#include "pb_common.h"
#include "pb_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kChainBlock = 1024;

struct GridDesc {
    const double *own;       // fine grid on the device, or null: own0 + i*step
    double own0, lo, hi, step;
    int64_t onwn;
};

__device__ inline double own_at(const GridDesc &g, int64_t i)
{
    if (g.own)
        return g.own[i];
    const double prod = __dmul_rn((double)i, g.step);       // multiply, then add, like NumPy
    return __dadd_rn(g.own0, prod);
}

__global__ __launch_bounds__(kBlock) void k_check_sorted(int *bad, const double *lwn,
                                                         const int32_t *lid, int64_t n)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j == 0 || j >= n)
        return;
    if (lid[j] < lid[j - 1] || (lid[j] == lid[j - 1] && lwn[j] < lwn[j - 1]))
        *bad = 1;
}

__global__ __launch_bounds__(kBlock) void k_classify(int32_t *iown_out, double *centre_out,
                                                     const double *lwn, GridDesc g, int64_t n)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n)
        return;
    const double v = lwn[j];
    if (v < g.lo || v > g.hi) {
        iown_out[j] = -1;
        centre_out[j] = 0.0;
        return;
    }
    int64_t iown = (int64_t)((v - g.lo) / g.step);
    if (iown + 1 < g.onwn && fabs(v - own_at(g, iown + 1)) < fabs(v - own_at(g, iown)))
        iown++;
    iown_out[j] = (int32_t)iown;
    centre_out[j] = own_at(g, iown);
}

// end[j]: first line after j that is NOT co-added to a group led by j; nxt[j]: the next group
// leader if j leads one (first in-range line at or after end[j]).  iso_lend[s] = one past the
// last line of isotope s, iso_next_first[s] = first in-range line of the isotopes after s.
__global__ __launch_bounds__(kBlock) void k_next(int32_t *end_out, int32_t *nxt_out,
                                                 const double *lwn, const int32_t *lid,
                                                 const int32_t *iown, const double *centre,
                                                 const int64_t *iso_lend,
                                                 const int64_t *iso_next_first, GridDesc g,
                                                 int64_t n)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n)
        return;
    const int s = lid[j];
    const int64_t send = iso_lend[s];
    if (iown[j] < 0) {
        // not a leader candidate: only a stepping stone for the pointer doubling
        end_out[j] = (int32_t)(j + 1);
        int64_t k = j + 1;
        if (k >= send || lwn[j] > g.hi)
            k = iso_next_first[s];
        // lines below the range precede the in-range ones of their isotope
        nxt_out[j] = (int32_t)min(k, n);
        return;
    }
    const double c = centre[j];
    int64_t lo = j + 1, hi = send;                // first k in (j, send) failing the predicate
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const double v = lwn[mid];
        if (v <= g.hi && fabs(v - c) < g.step)
            lo = mid + 1;
        else
            hi = mid;
    }
    end_out[j] = (int32_t)lo;
    int64_t k = lo;
    if (k >= send || lwn[k] > g.hi)
        k = iso_next_first[s];
    nxt_out[j] = (int32_t)min(k, n);
}

// lines of an isotope that lie below the range point to ... the next line; fix them up so that a
// chain entering there reaches the first in-range line (only the pointer doubling walks them)
__global__ __launch_bounds__(kChainBlock) void k_block_exit(int32_t *exit_out, const int32_t *nxt,
                                                            int64_t n)
{
    __shared__ int32_t s_e[2][kChainBlock];
    const int64_t base = (int64_t)blockIdx.x * kChainBlock;
    const int t = threadIdx.x;
    const int64_t j = base + t;
    const int64_t bend = base + kChainBlock;
    int32_t e = j < n ? nxt[j] : (int32_t)n;
    int cur = 0;
    s_e[0][t] = e;
    __syncthreads();
    for (int round = 0; round < 10; round++) {
        if (e < bend && e < n)
            e = s_e[cur][e - base];
        s_e[cur ^ 1][t] = e;
        cur ^= 1;
        __syncthreads();
    }
    if (j < n)
        exit_out[j] = e;
}

__global__ void k_chain(int32_t *entry, const int32_t *exit_ptr, int64_t first, int64_t n)
{
    int64_t e = first;
    while (e < n) {
        entry[e / kChainBlock] = (int32_t)e;
        e = exit_ptr[e];
    }
}

__global__ __launch_bounds__(kBlock) void k_mark(int32_t *leader, const int32_t *entry,
                                                 const int32_t *nxt, int64_t n, int nblocks)
{
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= nblocks)
        return;
    const int64_t bend = min((int64_t)(b + 1) * kChainBlock, n);
    int64_t j = entry[b];
    while (j >= 0 && j < bend) {
        leader[j] = 1;
        j = nxt[j];
    }
}

__global__ __launch_bounds__(kBlock) void k_emit(int32_t *gfirst, int32_t *gcount, int32_t *giown,
                                                 int32_t *giso, const int32_t *leader,
                                                 const int32_t *gpos, const int32_t *end_ptr,
                                                 const int32_t *iown, const int32_t *lid,
                                                 int64_t n)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n || !leader[j])
        return;
    const int g = gpos[j];
    gfirst[g] = (int32_t)j;
    gcount[g] = end_ptr[j] - (int32_t)j;
    giown[g] = iown[j];
    giso[g] = lid[j];
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)); }
};

}  // namespace

// Groups of the line list already on the device (l->d_lwn, l->d_lid); fills l->d_gfirst,
// d_gcount, d_giown, d_giso, the host mirrors, iso_gstart, ngroups, nadd, ninrange.
// Returns PB_OK, an error, or 1 when the list is not sorted by (isotope, wavenumber): the caller
// then runs the host loop.
int pb_lines_group_device(pb_lines *l, const double *lwn_h, const int32_t *lid_h,
                          const double *own_h)
{
    const int64_t n = l->nlines;
    const int niso = l->niso;
    if (n == 0)
        return 1;
    hipStream_t s = nullptr;
    DevBuf<int> bad;
    PB_HIP(bad.alloc(1));
    PB_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), s));
    const unsigned nb = (unsigned)pb::div_up(n, kBlock);
    k_check_sorted<<<nb, kBlock, 0, s>>>(bad.p, l->d_lwn, l->d_lid, n);
    PB_LAUNCH_CHECK();
    int unsorted = 0;
    PB_HIP(hipMemcpy(&unsorted, bad.p, sizeof(int), hipMemcpyDeviceToHost));
    if (unsorted)
        return 1;

    // per isotope: its line segment and the first in-range line of the isotopes after it
    // (bisections over the sorted host arrays)
    std::vector<int64_t> lend((size_t)niso, 0), next_first((size_t)niso, n);
    std::vector<int64_t> first_in((size_t)niso, -1);
    {
        int64_t at = 0;
        for (int i = 0; i < niso; i++) {
            const int32_t *e = std::upper_bound(lid_h + at, lid_h + n, (int32_t)i);
            const int64_t b0 = at, b1 = e - lid_h;
            lend[(size_t)i] = b1;
            const double *f = std::lower_bound(lwn_h + b0, lwn_h + b1, l->own0);
            if (f != lwn_h + b1 && *f <= l->own_last)
                first_in[(size_t)i] = f - lwn_h;
            at = b1;
        }
        int64_t nf = n;
        for (int i = niso - 1; i >= 0; i--) {
            next_first[(size_t)i] = nf;
            if (first_in[(size_t)i] >= 0)
                nf = first_in[(size_t)i];
        }
        // nf = first in-range line of the whole list
        l->iso_gstart.assign((size_t)niso + 1, 0);
        if (nf >= n) {                                   // nothing in range
            l->ngroups = 0;
            l->h_gfirst.clear();
            l->h_gcount.clear();
            l->h_giown.clear();
            PB_HIP(hipMalloc(&l->d_gfirst, 4));
            PB_HIP(hipMalloc(&l->d_gcount, 4));
            PB_HIP(hipMalloc(&l->d_giown, 4));
            PB_HIP(hipMalloc(&l->d_giso, 4));
            return PB_OK;
        }
        first_in.push_back(nf);
    }
    const int64_t first = first_in.back();

    GridDesc g;
    DevBuf<double> own_d;
    g.own = nullptr;
    if (own_h) {
        PB_HIP(own_d.alloc((size_t)l->onwn));
        PB_HIP(hipMemcpy(own_d.p, own_h, (size_t)l->onwn * 8, hipMemcpyHostToDevice));
        g.own = own_d.p;
    }
    g.own0 = l->own0;
    g.lo = l->own0;
    g.hi = l->own_last;
    g.step = l->ownstep;
    g.onwn = l->onwn;

    DevBuf<int32_t> iown, endp, nxt, exitp, entry, leader, gpos;
    DevBuf<double> centre;
    DevBuf<int64_t> lend_d, nf_d;
    const int nblocks = pb::div_up(n, kChainBlock);
    PB_HIP(iown.alloc((size_t)n));
    PB_HIP(centre.alloc((size_t)n));
    PB_HIP(endp.alloc((size_t)n));
    PB_HIP(nxt.alloc((size_t)n));
    PB_HIP(exitp.alloc((size_t)n));
    PB_HIP(entry.alloc((size_t)nblocks));
    PB_HIP(leader.alloc((size_t)n));
    PB_HIP(gpos.alloc((size_t)n + 1));
    PB_HIP(lend_d.alloc((size_t)niso));
    PB_HIP(nf_d.alloc((size_t)niso));
    PB_HIP(hipMemcpy(lend_d.p, lend.data(), (size_t)niso * 8, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(nf_d.p, next_first.data(), (size_t)niso * 8, hipMemcpyHostToDevice));
    PB_HIP(hipMemsetAsync(entry.p, 0xff, (size_t)nblocks * 4, s));
    PB_HIP(hipMemsetAsync(leader.p, 0, (size_t)n * 4, s));

    k_classify<<<nb, kBlock, 0, s>>>(iown.p, centre.p, l->d_lwn, g, n);
    PB_LAUNCH_CHECK();
    k_next<<<nb, kBlock, 0, s>>>(endp.p, nxt.p, l->d_lwn, l->d_lid, iown.p, centre.p, lend_d.p,
                                 nf_d.p, g, n);
    PB_LAUNCH_CHECK();
    k_block_exit<<<nblocks, kChainBlock, 0, s>>>(exitp.p, nxt.p, n);
    PB_LAUNCH_CHECK();
    k_chain<<<1, 1, 0, s>>>(entry.p, exitp.p, first, n);
    PB_LAUNCH_CHECK();
    k_mark<<<pb::div_up(nblocks, kBlock), kBlock, 0, s>>>(leader.p, entry.p, nxt.p, n, nblocks);
    PB_LAUNCH_CHECK();
    // exclusive scan of the leader flags -> position of every group
    {
        size_t tb = 0;
        PB_HIP(rocprim::exclusive_scan(nullptr, tb, leader.p, gpos.p, 0, (size_t)n,
                                       rocprim::plus<int32_t>(), s));
        DevBuf<unsigned char> tmp;
        PB_HIP(tmp.alloc(tb));
        PB_HIP(rocprim::exclusive_scan(tmp.p, tb, leader.p, gpos.p, 0, (size_t)n,
                                       rocprim::plus<int32_t>(), s));
        PB_HIP(hipStreamSynchronize(s));
    }
    int32_t last_pos = 0, last_flag = 0;
    PB_HIP(hipMemcpy(&last_pos, gpos.p + (n - 1), 4, hipMemcpyDeviceToHost));
    PB_HIP(hipMemcpy(&last_flag, leader.p + (n - 1), 4, hipMemcpyDeviceToHost));
    const int64_t ng = (int64_t)last_pos + last_flag;
    l->ngroups = ng;
    PB_HIP(hipMalloc(&l->d_gfirst, std::max<int64_t>(ng, 1) * 4));
    PB_HIP(hipMalloc(&l->d_gcount, std::max<int64_t>(ng, 1) * 4));
    PB_HIP(hipMalloc(&l->d_giown, std::max<int64_t>(ng, 1) * 4));
    PB_HIP(hipMalloc(&l->d_giso, std::max<int64_t>(ng, 1) * 4));
    k_emit<<<nb, kBlock, 0, s>>>(l->d_gfirst, l->d_gcount, l->d_giown, l->d_giso, leader.p,
                                 gpos.p, endp.p, iown.p, l->d_lid, n);
    PB_LAUNCH_CHECK();
    l->h_gfirst.resize((size_t)ng);
    l->h_gcount.resize((size_t)ng);
    l->h_giown.resize((size_t)ng);
    PB_HIP(hipMemcpy(l->h_gfirst.data(), l->d_gfirst, (size_t)ng * 4, hipMemcpyDeviceToHost));
    PB_HIP(hipMemcpy(l->h_gcount.data(), l->d_gcount, (size_t)ng * 4, hipMemcpyDeviceToHost));
    PB_HIP(hipMemcpy(l->h_giown.data(), l->d_giown, (size_t)ng * 4, hipMemcpyDeviceToHost));
    // per-isotope group segments: groups before the first line of each isotope
    for (int i = 0; i <= niso; i++) {
        const int64_t line = i == 0 ? 0 : lend[(size_t)i - 1];
        int32_t v = (int32_t)ng;
        if (line < n)
            PB_HIP(hipMemcpy(&v, gpos.p + line, 4, hipMemcpyDeviceToHost));
        l->iso_gstart[(size_t)i] = v;
    }
    int64_t members = 0;
    for (int32_t c : l->h_gcount)
        members += c;
    l->ninrange = members;
    l->nadd = members - ng;
    return PB_OK;
}
