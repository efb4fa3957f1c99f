// Walker-batched retrieval path (BASELINE config 5) and the fused transit column kernel.
//
// The reference evaluates one model per Pyrat.eval() call (pyratbay/pyrat/pyrat_obj.py:225-385):
// interp_ec over the cross-section table (opacity/line_sampling.py:394-463 ->
// src_c/_extcoeff.c:367-418), transit_path (atmosphere/atmosphere.py:782-802), the optical-depth
// loop (opacity/optic_depth.py:103-112 -> src_c/_trapezoid.c:238-276), transmission
// (spectrum/radiative_transfer.py:57-71) and band integration (spectrum/spec_tools.py:193-233).
// Here a batch of nw walkers goes through every stage in ONE launch each, the walker index being
// a grid dimension: no per-walker Python, no host synchronisation, and the cross-section table is
// read once per chunk of walkers instead of once per walker.
//
//   k_transit_path        raypath[w][r(r-1)/2 + i] from radius[w][L]
//   k_interp_ec_batch     ec[w][L][W] = sum_s dens[w][L][s] * lerp_T(etable[s][.][L][W])
//   k_transit_fused       ec -> (depth, ideep, spectrum): tau for all impact parameters, the
//                         reference's early exit and the transmission integral in one pass
//   k_band_integrate_batch bandflux[w][nbands]
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <type_traits>

#include "pb_common.h"

namespace {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------
// atmosphere.transit_path: path_r[i] = sqrt(rad_i^2 - rad_r^2) - sqrt(rad_{i+1}^2 - rad_r^2),
// rad = radius[itop:], packed lower triangle (row r has r entries from r(r-1)/2).  One multiply per
// square; the reference's pow(x, 2) is 1 ulp off x*x for 0.09 % of values: those rows agree to 1e-12.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_transit_path(double *raypath, const double *radius,
                                                         int itop, int nlayers, int64_t npath)
{
    const int w = blockIdx.y;
    const double *rad = radius + (int64_t)w * nlayers + itop;
    double *out = raypath + (int64_t)w * npath;
    const int nrow = nlayers - itop;
    for (int r = blockIdx.x; r < nrow; r += gridDim.x) {
        const double rr = rad[r] * rad[r];
        for (int i = threadIdx.x; i < r; i += kBlock) {
            const double a = rad[i] * rad[i] - rr;
            const double b = rad[i + 1] * rad[i + 1] - rr;
            out[((int64_t)r * (r - 1)) / 2 + i] = sqrt(a) - sqrt(b);
        }
    }
}

// ---------------------------------------------------------------------------
// Partition functions Z_i(T) of the isotopes of one TLI database at the layer temperatures of a
// batch of atmospheres (line_by_line.py:156-158: interp1d(db.temp, db.iso_pf[j], kind='slinear');
// :219-222: evaluated at the temperature profile on every extinction call).  SciPy's first-order
// spline is evaluated as its de Boor recurrence does (w = 1/(t_hi - t_lo); Z = z_lo (w (t_hi - T))
// + z_hi (w (T - t_lo)), interval t_lo <= T < t_hi, the last one closed): bit-equal to it.  A
// temperature outside the table is an error in the reference (interp1d raises): NaN is written
// and counted in *nbad.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_iso_partition(
    double *z, int64_t z_iso_stride, int64_t z_t_stride, const double *temp, int64_t ntemp,
    const double *ttab, int ntab, const double *pf, int niso, int32_t *nbad)
{
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= ntemp)
        return;
    const double x = temp[t];
    if (!(x >= ttab[0] && x <= ttab[ntab - 1])) {
        for (int i = 0; i < niso; i++)
            z[i * z_iso_stride + t * z_t_stride] = __longlong_as_double(0x7ff8000000000000ll);
        if (nbad)
            atomicAdd(nbad, 1);
        return;
    }
    int lo = 0, hi = ntab - 1;                 // largest lo <= ntab - 2 with ttab[lo] <= x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ttab[mid] <= x)
            lo = mid;
        else
            hi = mid;
    }
    const double xa = ttab[lo], xb = ttab[lo + 1];
    const double w = 1.0 / (xb - xa);
    const double h0 = w * (xb - x), h1 = w * (x - xa);
    for (int i = 0; i < niso; i++) {
        const double *row = pf + (int64_t)i * ntab;
        z[i * z_iso_stride + t * z_t_stride] = row[lo] * h0 + row[lo + 1] * h1;
    }
}

// ---------------------------------------------------------------------------
// Layers nobody reads (round 5).  With the columns of a retrieval batch in the depth order of a base
// model (TableSpectrum.order_columns) the row tile at which the transit kernel leaves a column is
// known in advance to within a layer or two: tile[b] = the last ROW TILE (16 impact parameters)
// the columns 256 b ... 256 b + 255 can need (base model's deepest crossing in the block + a
// margin).  The interpolation then writes the layers row0 ... row0 + 16 (tile[b] + 1) - 1 only
// (at C5's shape 80 % of ec: 0.8 GB of 4.1 GB per 64 walkers less), and the transit kernel, should
// a walker's column still be open beyond that tile, raises flags[walker] and flags[nwalkers]
// instead of reading what was never written.  `gate` (repair pass): a launch whose workgroups
// return at once unless the flag it points to is set -- the full interpolation gated on
// flags[nwalkers], the transit of walker w gated on flags[w] -- so the repair costs two nearly
// empty launches when nothing was flagged, and no host synchronisation ever.
// ---------------------------------------------------------------------------
using pb::TileLimit;
using pb::uniform_i32;
using pb::layer_wanted;

// ---------------------------------------------------------------------------
// interp_ec for a batch of walkers, assigning form.  Workgroup = (256 wavenumbers, layer,
// chunk of walkers).  Walkers of a chunk that share a temperature bracket share its two table
// slices: the brackets the chunk uses at this layer are walked in ascending order, the upper
// node of one bracket staying in registers as the lower node of the next; a walker is computed
// in the pass of its own bracket.  Per-(walker, layer) brackets and weights come from
// k_interp_weights (wave-uniform loads).  kS = species held in registers (<= 8).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_interp_weights(
    int32_t *tlo_out, double *coef_out, const double *ttable, const double *temps,
    const double *density, int nmol, int ncoef, int ntemp, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    const double t = temps[i];
    // same bracket rule as k_interp_ec (src_c/_extcoeff.c:394-398), clamped at the table's ends
    int tlo = pb::nearest_index(ttable, t, 0, ntemp - 1);
    if (t < ttable[tlo] || tlo == ntemp - 1)
        tlo--;
    tlo = max(tlo, 0);
    const double span = ttable[tlo + 1] - ttable[tlo];
    const double a = (ttable[tlo + 1] - t) / span, c = (t - ttable[tlo]) / span;
    tlo_out[i] = tlo;
    // the products interp_ec forms per sample, once per (walker, layer): w_lo*d_j, w_hi*d_j
    double *co = coef_out + i * 2 * ncoef;
    for (int j = 0; j < ncoef; j++) {
        const double d = j < nmol ? density[i * nmol + j] : 0.0;
        co[j] = a * d;
        co[ncoef + j] = c * d;
    }
}

// kFull: nmol == kS, no per-species predicate (the coefficient loads of a walker then merge into
// one scalar load and one wait)
template <int kS, bool kFull>
__global__ __launch_bounds__(kBlock) void k_interp_ec_batch(
    double *ec, const double *etable, const int32_t *tlo, const double *coef, int nmol,
    int ntemp, int nlayers, int nwave, int nwalkers, int chunk, TileLimit lim)
{
    if (!layer_wanted(lim, blockIdx.y, blockIdx.x * kBlock, blockIdx.x * kBlock + kBlock, nwave))
        return;
    // per-(walker, layer) brackets and coefficients are wave-uniform: through the constant
    // address space they are SCALAR loads (one s_load_dwordx16 per walker at four species)
    typedef const double __attribute__((address_space(4))) *ccoef_t;
    typedef const int32_t __attribute__((address_space(4))) *ctlo_t;
    const ctlo_t ctlo = (ctlo_t)(unsigned long long)tlo;
    const int k = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int w0 = blockIdx.z * chunk, w1 = min(w0 + chunk, nwalkers);
    // brackets used by the chunk at this layer
    int bmin = ntemp, bmax = -1;
    for (int w = w0; w < w1; w++) {
        const int b = ctlo[(int64_t)w * nlayers + k];
        bmin = min(bmin, b);
        bmax = max(bmax, b);
    }
    if (i >= nwave)
        return;
    const int64_t slice = (int64_t)nlayers * nwave;
    const double *tab = etable + (int64_t)k * nwave + i;      // + (j*ntemp + t)*slice
    double lo[kS], hi[kS];
#pragma unroll
    for (int j = 0; j < kS; j++)
        hi[j] = kFull || j < nmol ? tab[((int64_t)j * ntemp + bmin) * slice] : 0.0;
    for (int b = bmin; b <= bmax; b++) {
#pragma unroll
        for (int j = 0; j < kS; j++) {
            lo[j] = hi[j];
            hi[j] = kFull || j < nmol ? tab[((int64_t)j * ntemp + b + 1) * slice] : 0.0;
        }
        // (the walkers of bracket b as the set bits of a ballot over per-lane brackets -- no
        // scalar load and wait per walker and bracket -- measured slower: 1.21 against 1.11 ms)
        for (int w = w0; w < w1; w++) {
            const int64_t wk = (int64_t)w * nlayers + k;
            if (ctlo[wk] != b)
                continue;                                   // wave-uniform
            const ccoef_t co = (ccoef_t)(unsigned long long)(coef + wk * 2 * kS);
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < kS; j++)
                if (kFull || j < nmol)
                    acc += lo[j] * co[j] + hi[j] * co[kS + j];
            ec[wk * nwave + i] = acc;
        }
    }
}

// Two adjacent samples per thread (16 bytes per lane and access: 1 KiB per wavefront store instead of
// 512 B).  The rows of ec and of the table start at layer * nwave samples, 8-byte aligned only when
// nwave is odd, so the pairs start at the first EVEN absolute element of the row: the accesses are
// then 16-byte aligned; the odd sample in front of / behind the pairs is done by one lane on its own.
// NP = pair slots per thread, kBlock slots apart (every store instruction of a workgroup still
// covers 4 KiB of consecutive samples): NP = 2 halves the per-walker scalar loads, waits and
// branches per byte written (0.902 against 0.914 ms per 64 walkers at C5's shape, same bits).
template <int kS, bool kFull, int NP>
__global__ __launch_bounds__(kBlock) void k_interp_ec_batch2(
    double *ec, const double *etable, const int32_t *tlo, const double *coef, int nmol,
    int ntemp, int nlayers, int nwave, int nwalkers, int chunk, TileLimit lim)
{
    // (slots [x NP kBlock, (x + 1) NP kBlock) hold the samples 2 q - 1 ... 2 q + 1)
    if (!layer_wanted(lim, blockIdx.y, 2 * (int)blockIdx.x * NP * kBlock - 1,
                      2 * ((int)blockIdx.x + 1) * NP * kBlock + 1, nwave))
        return;
    typedef const double __attribute__((address_space(4))) *ccoef_t;
    typedef const int32_t __attribute__((address_space(4))) *ctlo_t;
    typedef double d2 __attribute__((ext_vector_type(2)));
    const ctlo_t ctlo = (ctlo_t)(unsigned long long)tlo;
    const int k = blockIdx.y;
    const int w0 = blockIdx.z * chunk, w1 = min(w0 + chunk, nwalkers);
    int bmin = ntemp, bmax = -1;
    for (int w = w0; w < w1; w++) {
        const int b = ctlo[(int64_t)w * nlayers + k];
        bmin = min(bmin, b);
        bmax = max(bmax, b);
    }
    const int64_t slice = (int64_t)nlayers * nwave;
    // first sample of the row whose absolute element index is even (all slices / walkers share
    // the parity when slice and nlayers * nwave are of one parity -- checked by the launcher)
    const int head = (int)(((int64_t)k * nwave) & 1);
    // slot q of a row: its first sample alone when the row starts at an odd element, then pairs
    int ii[NP];
    bool live[NP], pair[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int q = (blockIdx.x * NP + p) * kBlock + threadIdx.x;
        ii[p] = head ? (q == 0 ? 0 : 1 + 2 * (q - 1)) : 2 * q;
        live[p] = ii[p] < nwave;
        pair[p] = live[p] && !(head && q == 0) && ii[p] + 1 < nwave;
        if (!live[p])
            ii[p] = 0;                                      // (a valid address; never stored)
    }
    if (!live[0])
        return;                                             // (slots ascend with p)
    const double *tab = etable + (int64_t)k * nwave;
    d2 lo[NP][kS], hi[NP][kS];
    auto load = [&](int p, int j, int b) -> d2 {
        const double *ptr = tab + ii[p] + ((int64_t)j * ntemp + b) * slice;
        if (pair[p])
            return *reinterpret_cast<const d2 *>(ptr);
        d2 v;
        v.x = ptr[0];
        v.y = 0.0;
        return v;
    };
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
        for (int j = 0; j < kS; j++)
            hi[p][j] = kFull || j < nmol ? load(p, j, bmin) : d2{0.0, 0.0};
    for (int b = bmin; b <= bmax; b++) {
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int j = 0; j < kS; j++) {
                lo[p][j] = hi[p][j];
                hi[p][j] = kFull || j < nmol ? load(p, j, b + 1) : d2{0.0, 0.0};
            }
        for (int w = w0; w < w1; w++) {
            const int64_t wk = (int64_t)w * nlayers + k;
            if (ctlo[wk] != b)
                continue;                                   // wave-uniform
            const ccoef_t co = (ccoef_t)(unsigned long long)(coef + wk * 2 * kS);
#pragma unroll
            for (int p = 0; p < NP; p++) {
                d2 acc = {0.0, 0.0};
#pragma unroll
                for (int j = 0; j < kS; j++)
                    if (kFull || j < nmol) {
                        // same products and sums per sample as the one-sample kernel
                        acc.x += lo[p][j].x * co[j] + hi[p][j].x * co[kS + j];
                        acc.y += lo[p][j].y * co[j] + hi[p][j].y * co[kS + j];
                    }
                double *dst = ec + wk * nwave + ii[p];
                if (pair[p])
                    *reinterpret_cast<d2 *>(dst) = acc;
                else if (live[p])
                    dst[0] = acc.x;
            }
        }
    }
}

// Ray paths re-laid for the fused kernel: for every block of kRows impact parameters the segments
// [i][row], zero where segment >= row -- contiguous per (block, segment), so that the kernel can
// take them with wide SCALAR loads (they are wave-uniform) and feed v_fma_f64 from SGPRs.
__global__ __launch_bounds__(kBlock) void k_path_blocks(double *blocked, const double *raypath,
                                                        int64_t npath, int64_t nblocked, int rows,
                                                        int nimpact)
{
    const int w = blockIdx.y;
    const double *path = raypath + (int64_t)w * npath;
    double *out = blocked + (int64_t)w * nblocked;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < nblocked;
         e += (int64_t)gridDim.x * kBlock) {
        // block b starts at rows * sum_{b'<b} nseg_b', nseg_b = min(rows*b + rows, nimpact) - 1
        int b = 0;
        int64_t off = 0;
        for (;;) {
            const int64_t n = (int64_t)max(min(rows * b + rows, nimpact) - 1, 0) * rows;
            if (e < off + n)
                break;
            off += n;
            b++;
        }
        const int i = (int)((e - off) / rows), k = (int)((e - off) % rows);
        const int r = rows * b + k;
        out[e] = (r < nimpact && i < r) ? path[((int64_t)r * (r - 1)) / 2 + i] : 0.0;
    }
}

// ---------------------------------------------------------------------------
// Transit optical depth + transmission, one pass per column (optic_depth.py:103-112 with the
// early exit of _trapezoid.c:259-273, radiative_transfer.py:57-71 incl. the cloud deck).
// thread = column (x walker); the impact parameters are taken kRows at a time: for one block of
// rows the column of ec is streamed from its top (coalesced over columns; the re-reads of
// later blocks come from L2 / the Infinity Cache), the kRows running sums stay in registers and
// the ray-path segments of the block sit in LDS ([segment][row], zero where segment >= row, so one
// predicate-free loop serves all rows with the reference's products and additions).  After each
// block its rows are examined in order: pb::exp_s(-tau)*r joins the trapezoid, the first tau above
// maxdepth ends the column -- later blocks are not computed at all, which is where the time
// of the two-kernel form went (every row of every column, then a second pass to find the exit).
// depth and ideep are optional outputs (a retrieval needs neither).
// ---------------------------------------------------------------------------
__device__ inline double deck_integrand(double f_above, double f_below, double r_above,
                                        double r_below, double rsurf)
{
    const double slope = (f_above - f_below) / (r_above - r_below);
    return slope * (rsurf - r_below) + f_below;
}

// kFma: tau += path*s as ONE fused multiply-add instead of the reference's product-then-sum
// (two roundings): half the FP64 instructions, results equal to ~1e-16 relative.  Used only
// where no optical depth is returned (the retrieval batch: spectrum only).
template <int kRows, bool kScalarPath, bool kFma>
__global__ __launch_bounds__(kBlock) void k_transit_fused(
    double *depth, int32_t *ideep, double *spectrum, const double *ec, const double *raypath,
    const double *radius, int64_t npath, double rstar, int itop, int ibottom, double maxdepth,
    int nlayers, int nwave, int deck_row, double rsurf)
{
    // kScalarPath: raypath is the blocked layout of k_path_blocks (npath = its length per
    // walker), read through the constant address space = scalar loads; else the packed lower
    // triangle, staged per block in LDS
    extern __shared__ __align__(16) double s_path[];      // [segment][kRows]
    const int w = blockIdx.y;
    const int col = blockIdx.x * kBlock + threadIdx.x;
    const bool active = col < nwave;
    const int64_t plane = (int64_t)nlayers * nwave;
    ec += (int64_t)w * plane;
    if (depth)
        depth += (int64_t)w * plane;
    const double *path = raypath ? raypath + (int64_t)w * npath : nullptr;
    // the walker's radii are wave-uniform: scalar loads through the constant address space
    typedef const double __attribute__((address_space(4))) *crad_t;
    const crad_t rad = (crad_t)(unsigned long long)(radius ? radius + (int64_t)w * nlayers : nullptr);
    const int nimpact = min(ibottom, nlayers) - itop;     // rows 0..nimpact-1 are evaluated
    const double *src = ec + (int64_t)itop * nwave + (active ? col : 0);

    int stop = -1;
    double acc = 0.0, fprev = 0.0, rprev = 0.0;
    if (depth && active)
        for (int r = 0; r < itop; r++)
            depth[(int64_t)r * nwave + col] = 0.0;
    int rdone = 0;                       // rows examined so far (uniform)
    int64_t boff = 0;                    // start of the current block in the blocked path layout
    for (int rb = 0; rb < nimpact; rb += kRows) {
        // every column of the workgroup has met its exit: nothing left to compute
        if (__syncthreads_count(active && stop < 0) == 0)
            break;
        const int rlast = min(rb + kRows, nimpact) - 1;
        const int nseg = max(rlast, 0);
        if (!kScalarPath) {
            for (int e = threadIdx.x; e < nseg * kRows; e += kBlock) {
                const int i = e / kRows, k = e % kRows;
                const int r = rb + k;
                s_path[e] = (r <= rlast && i < r) ? path[((int64_t)r * (r - 1)) / 2 + i] : 0.0;
            }
            __syncthreads();
        }
        rdone = rlast + 1;
        if (!active) {
            boff += (int64_t)nseg * kRows;
            continue;
        }
        if (stop >= 0) {
            // below the first crossing the reference leaves zeros
            if (depth)
                for (int r = rb; r <= rlast; r++)
                    depth[(int64_t)(itop + r) * nwave + col] = 0.0;
            boff += (int64_t)nseg * kRows;
            continue;
        }
        double tau[kRows];
#pragma unroll
        for (int k = 0; k < kRows; k++)
            tau[k] = 0.0;
        if (nseg > 0) {
            double prev = src[0];
            if (kScalarPath) {
                typedef const double __attribute__((address_space(4))) *cpath_t;
                const cpath_t pb_ = (cpath_t)(unsigned long long)(path + boff);
#pragma unroll 2
                for (int i = 0; i < nseg; i++) {
                    const double next = src[(int64_t)(i + 1) * nwave];
                    const double s = next + prev;
                    prev = next;
                    double pv[kRows];                           // wave-uniform: scalar loads
#pragma unroll
                    for (int k = 0; k < kRows; k++)
                        pv[k] = pb_[i * kRows + k];
#pragma unroll
                    for (int k = 0; k < kRows; k++)
                        tau[k] = kFma ? fma(pv[k], s, tau[k]) : tau[k] + pv[k] * s;
                }
            } else {
#pragma unroll 4
                for (int i = 0; i < nseg; i++) {
                    const double next = src[(int64_t)(i + 1) * nwave];
                    const double s = next + prev;
                    prev = next;
                    const double *pk = s_path + i * kRows;      // LDS broadcast reads
#pragma unroll
                    for (int k = 0; k < kRows; k++)
                        tau[k] += pk[k] * s;
                }
            }
        }
        boff += (int64_t)nseg * kRows;
#pragma unroll
        for (int k = 0; k < kRows; k++) {
            const int r = rb + k;
            if (r <= rlast) {
                double t = tau[k];
                if (stop < 0) {
                    if (spectrum) {
                        const double rr = rad[itop + r];
                        double f = pb::exp_s(-t) * rr;
                        if (r > 0 && r == deck_row) {
                            f = deck_integrand(fprev, f, rprev, rr, rsurf);
                            acc += (rsurf - rprev) * (fprev + f);
                        } else if (r > 0) {
                            acc += (rr - rprev) * (fprev + f);
                        }
                        fprev = f;
                        rprev = rr;
                    }
                    if (t > maxdepth)
                        stop = r;
                } else {
                    t = 0.0;
                }
                if (depth)
                    depth[(int64_t)(itop + r) * nwave + col] = t;
            }
        }
    }
    if (!active)
        return;
    if (depth)
        for (int r = max(rdone, 0); r < nlayers - itop; r++)
            depth[(int64_t)(itop + r) * nwave + col] = 0.0;   // rows never reached, rows >= ibottom
    // ideep[ideep<0] = r with r the last loop value (itop if the loop is empty)
    const int last = nimpact > 0 ? itop + nimpact - 1 : itop;
    if (ideep)
        ideep[(int64_t)w * nwave + col] = stop >= 0 ? itop + stop : last;
    if (spectrum) {
        const double rtop = rad[itop];
        spectrum[(int64_t)w * nwave + col] = (rtop * rtop + 2 * (acc * 0.5)) / (rstar * rstar);
    }
}

// ---------------------------------------------------------------------------
// The retrieval batch (spectrum only, no cloud deck): the same pass with TWO columns per thread,
// so that every ray-path value fetched by a scalar load feeds two fused multiply-adds -- the
// scalar cache cannot hold the ray paths of the few walkers a CU works on at once (30 KB each),
// and with one column per thread the waits for those loads are half of the kernel's time.
// ---------------------------------------------------------------------------
template <int kRows>
__global__ __launch_bounds__(kBlock) void k_transit_pair(
    double *spectrum, const double *ec, const double *blocked, const double *radius, int64_t plen,
    double rstar, int itop, int ibottom, double maxdepth, int nlayers, int nwave)
{
    const int w = blockIdx.y;
    const int col[2] = {(int)(blockIdx.x * 2 * kBlock + threadIdx.x),
                        (int)(blockIdx.x * 2 * kBlock + kBlock + threadIdx.x)};
    const bool active[2] = {col[0] < nwave, col[1] < nwave};
    const int64_t plane = (int64_t)nlayers * nwave;
    ec += (int64_t)w * plane;
    typedef const double __attribute__((address_space(4))) *cdbl_t;
    const cdbl_t rad = (cdbl_t)(unsigned long long)(radius + (int64_t)w * nlayers);
    const cdbl_t path = (cdbl_t)(unsigned long long)(blocked + (int64_t)w * plen);
    const int nimpact = min(ibottom, nlayers) - itop;
    const double *src[2] = {ec + (int64_t)itop * nwave + (active[0] ? col[0] : 0),
                            ec + (int64_t)itop * nwave + (active[1] ? col[1] : 0)};
    int stop[2] = {-1, -1};
    double acc[2] = {0.0, 0.0}, fprev[2] = {0.0, 0.0};
    int64_t boff = 0;
    for (int rb = 0; rb < nimpact; rb += kRows) {
        if (__syncthreads_count((active[0] && stop[0] < 0) || (active[1] && stop[1] < 0)) == 0)
            break;
        const int rlast = min(rb + kRows, nimpact) - 1;
        const int nseg = max(rlast, 0);
        double tau[2][kRows];
#pragma unroll
        for (int k = 0; k < kRows; k++)
            tau[0][k] = tau[1][k] = 0.0;
        if (nseg > 0) {
            double prev0 = src[0][0], prev1 = src[1][0];
            const cdbl_t pb_ = path + boff;
            // rows of ec are fetched kAhead at a time, one group ahead of the sums that use them
            // (a wavefront then keeps 2 x kAhead row loads in flight: the stream comes from HBM)
            constexpr int kAhead = 4;
            double nx0[kAhead], nx1[kAhead];
#pragma unroll
            for (int j = 0; j < kAhead; j++) {
                const int64_t row = (int64_t)min(j + 1, nseg) * nwave;
                nx0[j] = src[0][row];
                nx1[j] = src[1][row];
            }
            for (int i0 = 0; i0 < nseg; i0 += kAhead) {
                double c0[kAhead], c1[kAhead];
#pragma unroll
                for (int j = 0; j < kAhead; j++) {
                    c0[j] = nx0[j];
                    c1[j] = nx1[j];
                }
#pragma unroll
                for (int j = 0; j < kAhead; j++) {
                    const int64_t row = (int64_t)min(i0 + kAhead + j + 1, nseg) * nwave;
                    nx0[j] = src[0][row];
                    nx1[j] = src[1][row];
                }
#pragma unroll
                for (int j = 0; j < kAhead; j++) {
                    const int i = i0 + j;
                    if (i < nseg) {                             // uniform
                        const double s0 = c0[j] + prev0, s1 = c1[j] + prev1;
                        prev0 = c0[j];
                        prev1 = c1[j];
                        double pv[kRows];                       // wave-uniform: scalar loads
#pragma unroll
                        for (int k = 0; k < kRows; k++)
                            pv[k] = pb_[i * kRows + k];
#pragma unroll
                        for (int k = 0; k < kRows; k++) {
                            tau[0][k] = fma(pv[k], s0, tau[0][k]);
                            tau[1][k] = fma(pv[k], s1, tau[1][k]);
                        }
                    }
                }
            }
        }
        boff += (int64_t)nseg * kRows;
#pragma unroll
        for (int c = 0; c < 2; c++) {
#pragma unroll
            for (int k = 0; k < kRows; k++) {
                const int r = rb + k;
                if (r <= rlast && active[c] && stop[c] < 0) {
                    const double t = tau[c][k];
                    const double rr = rad[itop + r];
                    const double f = pb::exp_s(-t) * rr;
                    if (r > 0)
                        acc[c] += (rr - rad[itop + r - 1]) * (fprev[c] + f);
                    fprev[c] = f;
                    if (t > maxdepth)
                        stop[c] = r;
                }
            }
        }
    }
    const double rtop = rad[itop];
#pragma unroll
    for (int c = 0; c < 2; c++)
        if (active[c])
            spectrum[(int64_t)w * nwave + col[c]] =
                (rtop * rtop + 2 * (acc[c] * 0.5)) / (rstar * rstar);
}

// ---------------------------------------------------------------------------
// The retrieval batch on the matrix cores.  Per walker the optical depths are ONE triangular
// matrix product shared by all of its columns,
//     tau[r][col] = sum_{j<=r} Q[r][j] * ec[itop + j][col],   Q[r][j] = P[r][j] + P[r][j-1]
// (P = the ray paths of optic_depth.py:103-112: sum_i P[r][i] (ec[i+1] + ec[i]) regrouped by
// layer), an 80 x 80 lower-triangular Q against an 80 x 1e5 block of ec at C5's shape.
// v_mfma_f64_16x16x4_f64: A = a 16-row x 4-layer block of Q (one double per lane, from LDS, laid
// out in lane order by k_path_qblocks), B = 4 layers x 16 columns of ec (one double per lane,
// straight from global memory: lane = (layer l>>4, column l&15), 128-byte runs), C = 16 rows x 16
// columns of tau (4 doubles per lane).  A wavefront owns NT column tiles and all MT row tiles:
// NT x MT accumulators stay in registers while ec streams past ONCE (the vector form re-read
// every column once per block of 16 rows: 3x at 80 layers), and of the MT x 4MT blocks of Q only
// the 2MT^2 + 2MT on or below the diagonal are multiplied (60 of 100 at 80 layers).
// The epilogue -- exp(-tau) r, first crossing of maxdepth, trapezoid over the rows
// (radiative_transfer.py:57-71) -- runs on the accumulator layout: lane (q = l>>4, n = l&15) holds
// the rows 16m + 4j + q of column n; the previous row's integrand comes from the lane 16 below
// (one cross-lane move per row), the first crossing and the sums are combined over the four lanes
// of a column.  Sums of a column are added in a different order than the reference's loop:
// spectra agree to ~1e-15 relative with the vector form, not bit for bit.
// ---------------------------------------------------------------------------
__host__ __device__ constexpr int qblocks(int mt) { return 2 * mt * mt + 2 * mt; }

__global__ __launch_bounds__(kBlock) void k_path_qblocks(double *out, const double *raypath,
                                                         int64_t npath, int nblk, int nimpact)
{
    const int w = blockIdx.y;
    const double *path = raypath + (int64_t)w * npath;
    double *o = out + (int64_t)w * nblk * 64;
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < nblk * 64; e += gridDim.x * kBlock) {
        const int blk = e >> 6, l = e & 63;
        int m = 0;
        while (qblocks(m + 1) <= blk)
            m++;
        const int ks = blk - qblocks(m);
        const int r = 16 * m + (l & 15), j = 4 * ks + (l >> 4);
        double v = 0.0;
        if (r >= 1 && r < nimpact && j <= r) {
            const int64_t base = ((int64_t)r * (r - 1)) / 2;
            if (j < r)
                v = path[base + j];
            if (j >= 1)
                v += path[base + j - 1];
        }
        o[e] = v;
    }
}

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

// Epilogue of the matrix-core transit kernels on the accumulator layout: lane (kq = l >> 4,
// n = l & 15) holds the rows 16m + 4j + kq of one column per 16-column tile.  One tile:
template <int MT>
__device__ __forceinline__ void mfma_transit_epilogue_tile(
    const v4d (&C)[MT], const double *s_rad, double *dst, bool ok, int lane, int nimpact,
    double maxdepth, double rstar)
{
    const int kq = lane >> 4;
    const double rtop = s_rad[0];
    const double *srad = s_rad + kq;
    const int src_lane = (lane + 48) & 63;                // the lane one row above (16 below)
    int first = INT_MAX;
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = 16 * m + 4 * j + kq;
            if (r < nimpact && C[m][j] > maxdepth)
                first = min(first, r);
        }
    first = min(first, __shfl_xor(first, 16));
    first = min(first, __shfl_xor(first, 32));
    double acc = 0.0, carry = 0.0;                        // carry: row 16m + 4j - 1 seen from q = 0
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = 16 * m + 4 * j + kq;
            const bool in = r < nimpact && r <= first;
            const double rr = srad[16 * m + 4 * j];
            const double f = in ? pb::exp_s(-C[m][j]) * rr : 0.0;
            const double up = __shfl(f, src_lane);        // q > 0: row r - 1; q = 0: row r + 3
            const double fprev = kq > 0 ? up : carry;
            carry = up;
            if (in && r >= 1)
                acc += (rr - srad[16 * m + 4 * j - 1]) * (fprev + f);
        }
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (kq == 0 && ok)
        *dst = (rtop * rtop + 2 * (acc * 0.5)) / (rstar * rstar);
}

// the two column tiles of a wavefront (columns col0 and col0 + 1 of one walker)
template <int MT>
__device__ __forceinline__ void mfma_transit_epilogue(
    const v4d (&C)[2][MT], const double *s_rad, double *spectrum_w, int col0, const bool (&ok)[2],
    int lane, int nimpact, double maxdepth, double rstar)
{
#pragma unroll
    for (int t = 0; t < 2; t++)
        mfma_transit_epilogue_tile<MT>(C[t], s_rad, spectrum_w + col0 + t, ok[t], lane, nimpact,
                                       maxdepth, rstar);
}

// A walker's Q blocks (qblocks(MT) x 64 doubles: 30 KB at 80 layers) -> LDS.  The count is a
// compile-time constant of the instantiation, so every thread issues ALL of its 16-byte loads
// before the first LDS store: one L2 round trip per workgroup instead of one per 8-byte element of
// a run-time loop (7.5 at 512 threads).
template <int MT, int TB>
__device__ __forceinline__ void stage_qblocks(double *s_q, const double *q, int tid)
{
    constexpr int N2 = qblocks(MT) * 32;                  // 16-byte units
    constexpr int PER = (N2 + TB - 1) / TB;
    const d2u *src = reinterpret_cast<const d2u *>(q);
    d2u tmp[PER];
#pragma unroll
    for (int i = 0; i < PER; i++)
        if (tid + i * TB < N2)
            tmp[i] = src[tid + i * TB];
#pragma unroll
    for (int i = 0; i < PER; i++)
        if (tid + i * TB < N2) {
            s_q[2 * (tid + i * TB)] = tmp[i].x;
            s_q[2 * (tid + i * TB) + 1] = tmp[i].y;
        }
}

#ifdef PB_EXPERIMENTS   // the layers-outer matrix kernel of round 3 (replaced by k_transit_mfma_rows)
// A wavefront owns 32 columns = two 16-column tiles (the even and the odd columns of its range:
// one 16-byte load per lane fetches both) and all MT row tiles: 2 x MT accumulators stay in
// registers while the layers stream past once, four K-steps (16 layers) in flight ahead of the
// four being multiplied.  The loads carry no branch (rows beyond the last layer and columns beyond
// the grid read a clamped address: their Q entries are zero, their results unused): behind a
// divergent `if` the compiler drains every load (s_waitcnt vmcnt(0)) before it issues the next.
template <int MT, int WPS, int TB>
__global__ __launch_bounds__(TB, WPS) void k_transit_mfma(
    double *spectrum, const double *ec, const double *qblk, const double *radius, int nblk,
    double rstar, int itop, int ibottom, double maxdepth, int nlayers, int nwave)
{
    extern __shared__ __align__(16) double s_q[];         // [nblk][64] | rad[16 MT]
    const int w = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nimpact = min(ibottom, nlayers) - itop;
    double *s_rad = s_q + (size_t)nblk * 64;
    {
        stage_qblocks<MT, TB>(s_q, qblk + (int64_t)w * nblk * 64, tid);
        for (int r = tid; r < 16 * MT; r += TB)
            s_rad[r] = r < nimpact ? radius[(int64_t)w * nlayers + itop + r] : 0.0;
    }
    __syncthreads();
    const int c0 = (blockIdx.x * (TB / 64) + wave) * 32;
    if (c0 >= nwave)
        return;                                           // (after the only barrier)
    const int kq = lane >> 4, n = lane & 15;
    const int col0 = c0 + 2 * n;                          // tile 0: even columns, tile 1: odd ones
    const bool ok[2] = {col0 < nwave, col0 + 1 < nwave};
    const int cpair = max(min(col0, nwave - 2), 0);       // first column of the pair I load
    const bool second = col0 != cpair;                    // my column 0 is the pair's second one
    const double *src = ec + ((int64_t)w * nlayers + itop) * nwave + cpair;
    const int KS = (nimpact + 3) / 4;
    v4d C[2][MT];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int m = 0; m < MT; m++)
            C[t][m] = v4d{0.0, 0.0, 0.0, 0.0};
    double bcur[4][2], bnxt[4][2];
    auto loadb = [&](int mb, double (&b)[4][2]) {         // (the launcher guarantees nwave >= 2)
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int j = min(4 * (4 * mb + kk) + kq, nimpact - 1);
            const d2u v = *reinterpret_cast<const d2u *>(src + (int64_t)j * nwave);
            b[kk][0] = second ? v.y : v.x;
            b[kk][1] = v.y;
        }
    };
    loadb(0, bcur);
    const double *sq = s_q + lane;
#pragma unroll
    for (int mb = 0; mb < MT; mb++) {
        if (4 * mb < KS) {                                // uniform
            if (mb + 1 < MT)
                loadb(mb + 1, bnxt);
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int ks = 4 * mb + kk;
#pragma unroll
                for (int m = mb; m < MT; m++) {
                    const double a = sq[(qblocks(m) + ks) * 64];
                    C[0][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bcur[kk][0], C[0][m], 0, 0, 0);
                    C[1][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bcur[kk][1], C[1][m], 0, 0, 0);
                }
            }
            if (mb + 1 < MT) {
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    bcur[kk][0] = bnxt[kk][0];
                    bcur[kk][1] = bnxt[kk][1];
                }
            }
        }
    }
    mfma_transit_epilogue<MT>(C, s_rad, spectrum + (int64_t)w * nwave, col0, ok, lane, nimpact,
                              maxdepth, rstar);
}

#endif  // PB_EXPERIMENTS

// The same products ROW TILE BY ROW TILE, with the reference's early exit at tile granularity
// (_trapezoid.c:259-273: a column is finished at the first row whose optical depth exceeds
// maxdepth).  Row tile m needs the layers 0 .. 16m + 15 only, so the B operands stay in registers
// (2 doubles per K-step and lane: 80 registers at 80 layers) while ONE row tile's accumulators are
// live; its rows go through the epilogue at once (carry, sums and first crossing kept across
// tiles), and when every column of the wavefront has crossed, the remaining tiles -- their products
// AND the loads of their layers, which are issued one tile ahead -- are skipped.  Per column the
// products, their order and the epilogue's arithmetic are those of k_transit_mfma: same bits.
// Columns that cross at similar rows must sit together for the exit to happen: the caller orders
// the columns (TableSpectrum.column_order) and passes `scatter`, the grid index of each column.
template <int B, int E, class F>
__device__ __forceinline__ void static_for_rows(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for_rows<B + 1, E>(f);
    }
}

// pb::exp_s of four values at once, cut into 15 slices of four independent instructions each (the
// same operations in the same order per value: same bits), so that the slices of one column tile's
// epilogue can be issued between the matrix products of the other (k_transit_mfma_rows).
struct Exp4 {
    double x[4], n[4], r[4], p[4];
};
constexpr int kExp4Slices = 15;
template <int S>
__device__ __forceinline__ void exp4_slice(Exp4 &e)
{
    using pb::sgpr_const;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if constexpr (S == 0)
            e.n[j] = rint(e.x[j] * sgpr_const(0x1.71547652b82fep+0));
        else if constexpr (S == 1)
            e.r[j] = fma(e.n[j], sgpr_const(-0x1.62e42fefa39efp-1), e.x[j]);
        else if constexpr (S == 2)
            e.r[j] = fma(sgpr_const(-0x1.abc9e3b39803fp-56), e.n[j], e.r[j]);
        else if constexpr (S == 3)
            e.p[j] = fma(sgpr_const(0x1.ade156a5dcb37p-26), e.r[j], sgpr_const(0x1.28af3fca7ab0cp-22));
        else if constexpr (S == 4)
            e.p[j] = fma(e.r[j], e.p[j], sgpr_const(0x1.71dee623fde64p-19));
        else if constexpr (S == 5)
            e.p[j] = fma(e.r[j], e.p[j], sgpr_const(0x1.a01997c89e6b0p-16));
        else if constexpr (S == 6)
            e.p[j] = fma(e.r[j], e.p[j], sgpr_const(0x1.a01a014761f6ep-13));
        else if constexpr (S == 7)
            e.p[j] = fma(e.r[j], e.p[j], sgpr_const(0x1.6c16c1852b7b0p-10));
        else if constexpr (S == 8)
            e.p[j] = fma(e.r[j], e.p[j], sgpr_const(0x1.1111111122322p-7));
        else if constexpr (S == 9)
            e.p[j] = fma(e.r[j], e.p[j], sgpr_const(0x1.55555555502a1p-5));
        else if constexpr (S == 10)
            e.p[j] = fma(e.r[j], e.p[j], sgpr_const(0x1.5555555555511p-3));
        else if constexpr (S == 11)
            e.p[j] = fma(e.r[j], e.p[j], sgpr_const(0x1.000000000000bp-1));
        else if constexpr (S == 12)
            e.p[j] = fma(e.r[j], e.p[j], 1.0);
        else if constexpr (S == 13)
            e.p[j] = fma(e.r[j], e.p[j], 1.0);
        else if constexpr (S == 14) {
            double v = ldexp(e.p[j], (int)e.n[j]);
            v = e.x[j] > 1024.0 ? __builtin_huge_val() : v;
            e.p[j] = e.x[j] < -1075.0 ? 0.0 : v;            // the result
        }
    }
}
template <int B, int E>
__device__ __forceinline__ void exp4_slices(Exp4 &e)
{
    if constexpr (B < E) {
        exp4_slice<B>(e);
        exp4_slices<B + 1, E>(e);
    }
}

template <int MT, int WPS, int TB>
__global__ __launch_bounds__(TB, WPS) void k_transit_mfma_rows(
    double *spectrum, const double *ec, const double *qblk, const double *radius, int nblk,
    double rstar, int itop, int ibottom, double maxdepth, int nlayers, int nwave,
    const int32_t *scatter, TileLimit lim, int32_t *flags)
{
    extern __shared__ __align__(16) double s_q[];         // [nblk][64] | rad[16 MT]
    const int w = blockIdx.y;
    if (lim.gate && uniform_i32(lim.gate + w) == 0)
        return;                                           // (repair pass: walker w was not flagged)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nimpact = min(ibottom, nlayers) - itop;
    double *s_rad = s_q + (size_t)nblk * 64;
    {
        stage_qblocks<MT, TB>(s_q, qblk + (int64_t)w * nblk * 64, tid);
        for (int r = tid; r < 16 * MT; r += TB)
            s_rad[r] = r < nimpact ? radius[(int64_t)w * nlayers + itop + r] : 0.0;
    }
    __syncthreads();
    const int c0 = (blockIdx.x * (TB / 64) + wave) * 32;
    if (c0 >= nwave)
        return;                                           // (after the only barrier)
    const int kq = lane >> 4, n = lane & 15;
    const int col0 = c0 + 2 * n;
    const bool ok[2] = {col0 < nwave, col0 + 1 < nwave};
    const int cpair = max(min(col0, nwave - 2), 0);
    const bool second = col0 != cpair;
    const double *src = ec + ((int64_t)w * nlayers + itop) * nwave + cpair;
    const int KS = (nimpact + 3) / 4;
    double b[4 * MT][2];
    auto loadb = [&](auto mbc) {
        constexpr int mb = decltype(mbc)::value;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int j = min(4 * (4 * mb + kk) + kq, nimpact - 1);
            const d2u v = *reinterpret_cast<const d2u *>(src + (int64_t)j * nwave);
            b[4 * mb + kk][0] = second ? v.y : v.x;
            b[4 * mb + kk][1] = v.y;
        }
    };
    const double rtop = s_rad[0];
    const double *srad = s_rad + kq;
    const int src_lane = (lane + 48) & 63;                // the lane one row above (16 below)
    int first[2] = {ok[0] ? INT_MAX : -1, ok[1] ? INT_MAX : -1};   // (-1: nothing to wait for)
    double acc[2] = {0.0, 0.0}, carry[2] = {0.0, 0.0};
    const double *sq = s_q + lane;
    bool done = false;
    // the last row tile whose layers were interpolated for these 32 columns (TileLimit)
    const int mlim = lim.tile ? uniform_i32(lim.tile + (c0 >> 8)) : MT;
    bool overrun = false;
    loadb(std::integral_constant<int, 0>{});
    auto tile = [&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if (done || 4 * m >= KS)                          // uniform
            return;
        if (m > mlim) {                                   // uniform: a column is still open beyond
            overrun = true;                               // what was interpolated -> repair pass
            done = true;
            return;
        }
        // the next tile's layers are requested before this tile's products (two tiles ahead:
        // measured slower, 1.27 against 1.16 ms at C5's shape -- the loads an exit wastes)
        // (not beyond the tile limit: those layers were never interpolated and an open column
        // there goes to the repair pass -- a fifth of the kernel's reads at C5's shape)
        if constexpr (m + 1 < MT)                         // (clamped rows: harmless past the end)
            if (m + 1 <= mlim || lim.row0 < 0)            // uniform (row0 < 0: A/B switch)
                loadb(std::integral_constant<int, m + 1>{});
        // Column tile 0's products; then column tile 1's with the exponentials of tile 0's rows
        // between them: a product holds the matrix pipe for 64 cycles, the slices issue meanwhile
        // (pinned by scheduling barriers: left alone, the compiler keeps the products together).
        // Tile 0's exponentials are computed for all four rows and selected afterwards.  Same bits;
        // 1.19 against 1.21 ms per 64 walkers at C5's shape.
        v4d C[2] = {v4d{0.0, 0.0, 0.0, 0.0}, v4d{0.0, 0.0, 0.0, 0.0}};
        Exp4 e0;
        {
            constexpr int K = 4 * m + 4;
#pragma unroll
            for (int ks = 0; ks < K; ks++)
                C[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sq[(qblocks(m) + ks) * 64], b[ks][0], C[0],
                                                            0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; j++)
                e0.x[j] = -C[0][j];
            __builtin_amdgcn_sched_barrier(0);
            static_for_rows<0, K>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
                C[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sq[(qblocks(m) + ks) * 64], b[ks][1], C[1],
                                                            0, 0, 0);
                exp4_slices<(ks * kExp4Slices) / K, ((ks + 1) * kExp4Slices) / K>(e0);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
#pragma unroll
        for (int t = 0; t < 2; t++) {
            int f = INT_MAX;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int r = 16 * m + 4 * j + kq;
                if (r < nimpact && C[t][j] > maxdepth)
                    f = min(f, r);
            }
            f = min(f, __shfl_xor(f, 16));
            f = min(f, __shfl_xor(f, 32));
            // (an earlier tile's crossing is below every row of this one)
            const int fst = first[t] == INT_MAX ? f : first[t];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int r = 16 * m + 4 * j + kq;
                const bool in = r < nimpact && r <= fst;
                const double rr = srad[16 * m + 4 * j];
                double fv;
                if (t == 0)
                    fv = in ? e0.p[j] * rr : 0.0;
                else
                    fv = in ? pb::exp_s(-C[t][j]) * rr : 0.0;
                const double up = __shfl(fv, src_lane);   // q > 0: row r - 1; q = 0: row r + 3
                const double fprev = kq > 0 ? up : carry[t];
                carry[t] = up;
                if (in && r >= 1)
                    acc[t] += (rr - srad[16 * m + 4 * j - 1]) * (fprev + fv);
            }
            first[t] = fst;
        }
        done = __all(first[0] != INT_MAX && first[1] != INT_MAX);
    };
    static_assert(MT <= 8, "row tiles");
    tile(std::integral_constant<int, 0>{});
    if constexpr (MT > 1) tile(std::integral_constant<int, 1>{});
    if constexpr (MT > 2) tile(std::integral_constant<int, 2>{});
    if constexpr (MT > 3) tile(std::integral_constant<int, 3>{});
    if constexpr (MT > 4) tile(std::integral_constant<int, 4>{});
    if constexpr (MT > 5) tile(std::integral_constant<int, 5>{});
    if constexpr (MT > 6) tile(std::integral_constant<int, 6>{});
    if constexpr (MT > 7) tile(std::integral_constant<int, 7>{});
    if (overrun) {
        if (lane == 0 && flags) {
            flags[w] = 1;
            flags[gridDim.y] = 1;                         // flags[nwalkers]: any walker
        }
        return;                                           // (the repair pass writes these columns)
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
        double a = acc[t];
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        if (kq == 0 && ok[t]) {
            const int64_t dst = scatter ? scatter[col0 + t] : col0 + t;
            // (an index outside the grid -- a caller's column_d that is not a permutation -- is
            // dropped, not written out of bounds)
            if (dst >= 0 && dst < nwave)
                spectrum[(int64_t)w * nwave + dst] =
                    (rtop * rtop + 2 * (a * 0.5)) / (rstar * rstar);
        }
    }
}

#ifdef PB_EXPERIMENTS   // one-pass table transit (slower than the two passes at C5; opt-in)
// a wave-uniform flag written by an earlier kernel, by a scalar load
__device__ __forceinline__ int uniform_flag(const int32_t *p)
{
    typedef const int32_t __attribute__((address_space(4))) *cptr;
    return *((cptr)(unsigned long long)p);
}

// ---------------------------------------------------------------------------
// Interpolation + optical depth + transmission of the retrieval batch in ONE pass: the B operand of
// k_transit_mfma -- 4 layers x 16 columns of a walker's ec -- is not loaded but FORMED from the
// cross-section table (the sums of interp_ec, _extcoeff.c:367-418: sum_s d_s (w_lo T[s][tlo] +
// w_hi T[s][tlo+1]), same products and order as k_interp_ec_batch2), so ec[walker][layer][sample]
// -- 4.1 GB per 64 walkers at C5's shape, written by one kernel and read back by the next -- never
// exists.  What makes that affordable is the ORDER of the workgroups: the table slices a walker
// brackets are shared with the other walkers of the batch, so all walkers of one column block run
// at the same time on ONE XCD (workgroup id -> xcd = id & 7; the XCD's slots walk its column
// blocks one after the other, the walkers of a block in consecutive slots): a slice element comes
// from HBM once per batch and from that XCD's L2 for the other walkers.
// LDS: the walker's Q blocks and radii as in k_transit_mfma, its per-layer coefficients
// (k_interp_weights) and table offsets.  PD = K-steps whose table loads are in flight ahead of the
// one being multiplied (2 x kS 16-byte loads per lane and K-step, wave-uniform bases in SGPRs +
// one 32-bit offset per lane: no vector address arithmetic).
// MEASURED (C5's shape, tools/bench_tt.py): 2.86 ms per 64 walkers against 2.67 for the two passes
// (the two-walker form below: 2.54).  HBM traffic is what it should be (PMC: 1.5 GB
// fetched per launch against 4.4 GB for k_transit_mfma alone; L2 hit rate 96 %), but the 8 slice
// reads per walker, layer and sample now come from the XCD's L2 -- 32.8 GB per batch -- and L2 ->
// L1 delivers ~12 TB/s of them: with the table loads taken out the kernel runs 1.4 ms, with the
// matrix products taken out 2.7.  Per-walker Q blocks (30 KB of LDS each) leave no room to share
// the slices of several walkers through LDS.  Kept (opt-in) for what it saves: the 4.1 GB of ec per
// 64 walkers are never allocated.
// ---------------------------------------------------------------------------
template <int MT, int WPS, int TB, int kS, int PD>
__global__ __launch_bounds__(TB, WPS) void k_table_transit_mfma(
    double *spectrum, const double *etable, const int32_t *tlo, const double *coef,
    const double *qblk, const double *radius, int nblk, double rstar, int itop, int ibottom,
    double maxdepth, int nmol, int ntemp, int nlayers, int nwave, int nwalkers, int ncolblk,
    int ginter, const int32_t *skip_if)
{
    extern __shared__ __align__(16) double s_q[];         // [nblk][64] | rad[16 MT] | coef[16 MT][2 kS] | off[16 MT]
    if (skip_if && uniform_flag(skip_if))
        return;                                           // the two-walker kernel runs this batch
    // XCD-aware order: see above
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    // (ginter column blocks of an XCD walked together, their walkers alternating -- PB_TT_GINTER;
    // measured 2.86 / 2.88 / 2.89 / 3.05 / 3.25 ms for 1 / 2 / 3 / 4 / 8: the default is 1)
    const int grp = slot / (ginter * nwalkers), within = slot % (ginter * nwalkers);
    const int cb = (grp * ginter + within % ginter) * 8 + xcd;
    const int w = within / ginter;
    if (cb >= ncolblk)
        return;                                           // (whole workgroup, before the barrier)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nimpact = min(ibottom, nlayers) - itop;
    double *s_rad = s_q + (size_t)nblk * 64;
    double *s_coef = s_rad + 16 * MT;
    uint32_t *s_off = reinterpret_cast<uint32_t *>(s_coef + 16 * MT * 2 * kS);
    {
        stage_qblocks<MT, TB>(s_q, qblk + (int64_t)w * nblk * 64, tid);
        for (int r = tid; r < 16 * MT; r += TB) {
            s_rad[r] = r < nimpact ? radius[(int64_t)w * nlayers + itop + r] : 0.0;
            // byte offset of (bracket's lower slice, layer) inside one species' block of the table
            // (< 4 GiB: checked by the launcher); rows beyond the last layer repeat it (their Q is 0)
            const int rr = min(r, nimpact - 1);
            const int64_t t = tlo[(int64_t)w * nlayers + itop + rr];
            s_off[r] = (uint32_t)(((t * nlayers + itop + rr) * (int64_t)nwave) * 8);
        }
        for (int e = tid; e < 16 * MT * 2 * kS; e += TB) {
            const int r = min(e / (2 * kS), nimpact - 1);
            s_coef[e] = coef[((int64_t)w * nlayers + itop + r) * 2 * kS + e % (2 * kS)];
        }
    }
    __syncthreads();
    const int c0 = (cb * (TB / 64) + wave) * 32;
    if (c0 >= nwave)
        return;                                           // (after the only barrier)
    const int kq = lane >> 4, n = lane & 15;
    const int col0 = c0 + 2 * n;                          // tile 0: even columns, tile 1: odd ones
    const bool ok[2] = {col0 < nwave, col0 + 1 < nwave};
    const int cpair = max(min(col0, nwave - 2), 0);       // first column of the pair I load
    const bool second = col0 != cpair;                    // my column 0 is the pair's second one
    const int64_t slice = (int64_t)nlayers * nwave;
    // wave-uniform bases (SGPR pairs) + one 32-bit byte offset per lane and K-step: the loads need
    // no vector address arithmetic (species beyond nmol: a valid block, their coefficients are 0)
    const char *base_lo[kS], *base_hi[kS];
#pragma unroll
    for (int sp = 0; sp < kS; sp++) {
        base_lo[sp] = reinterpret_cast<const char *>(etable + (int64_t)min(sp, nmol - 1) * ntemp * slice);
        base_hi[sp] = base_lo[sp] + slice * 8;
    }
    const uint32_t lane_off = (uint32_t)cpair * 8u;
    v4d C[2][MT];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int m = 0; m < MT; m++)
            C[t][m] = v4d{0.0, 0.0, 0.0, 0.0};
    d2u raw[PD][2 * kS];
    // K-steps of one group of four (same row tiles) are taken in an order rotated by the walker
    // index: the walkers of a column block run at the same time and would otherwise all ask the
    // XCD's L2 for the same lines at the same moment
    const int rot = w & 3;
    auto kstep = [&](int ks) { return (ks & ~3) | ((ks + rot) & 3); };
    auto issue = [&](int ks0, d2u (&r)[2 * kS]) {
        const int ks = kstep(ks0);
        const uint32_t off = s_off[4 * ks + kq] + lane_off;
#pragma unroll
        for (int sp = 0; sp < kS; sp++) {
            r[sp] = *reinterpret_cast<const d2u *>(base_lo[sp] + off);
            r[kS + sp] = *reinterpret_cast<const d2u *>(base_hi[sp] + off);
        }
    };
    auto combine = [&](int ks0, const d2u (&r)[2 * kS], double (&b)[2]) {
        const int ks = kstep(ks0);
        const double *co = s_coef + (4 * ks + kq) * 2 * kS;
        double ax = 0.0, ay = 0.0;
#pragma unroll
        for (int sp = 0; sp < kS; sp++) {
            // the sums of interp_ec with fused multiply-adds (k_interp_ec_batch2 rounds every
            // product: the two forms agree to ~1e-16 relative)
            ax = fma(r[sp].x, co[sp], ax);
            ay = fma(r[sp].y, co[sp], ay);
            ax = fma(r[kS + sp].x, co[kS + sp], ax);
            ay = fma(r[kS + sp].y, co[kS + sp], ay);
        }
        b[0] = second ? ay : ax;
        b[1] = ay;
    };
    // All 4 MT K-steps run whatever the number of layers (rows beyond the last one: Q = 0, the
    // table address of the last row): no branch inside the loop, ONE basic block, so that the
    // scheduler can place a step's loads and sums between the previous step's matrix products.
#pragma unroll
    for (int d = 0; d < PD; d++)
        if (d < 4 * MT)
            issue(d, raw[d]);
    double bcur[2], bnxt[2] = {0.0, 0.0};
    combine(0, raw[0], bcur);
    if (PD < 4 * MT)
        issue(PD, raw[0]);
    const double *sq = s_q + lane;
#pragma unroll
    for (int ks = 0; ks < 4 * MT; ks++) {
        // the next K-step's operand is formed in the shadow of this step's matrix products
        if (ks + 1 < 4 * MT) {
            combine(ks + 1, raw[(ks + 1) % PD], bnxt);
            if (ks + 1 + PD < 4 * MT)
                issue(ks + 1 + PD, raw[(ks + 1) % PD]);
        }
#pragma unroll
        for (int m = ks / 4; m < MT; m++) {
            const double a = sq[(qblocks(m) + kstep(ks)) * 64];
            C[0][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bcur[0], C[0][m], 0, 0, 0);
            C[1][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bcur[1], C[1][m], 0, 0, 0);
        }
        bcur[0] = bnxt[0];
        bcur[1] = bnxt[1];
    }
    mfma_transit_epilogue<MT>(C, s_rad, spectrum + (int64_t)w * nwave, col0, ok, lane, nimpact,
                              maxdepth, rstar);
}

// ---------------------------------------------------------------------------
// The one-pass kernel with TWO walkers per wavefront.  What binds k_table_transit_mfma is the
// delivery of table slices from L2 (8 loads per walker, layer and sample); two walkers that
// bracket the same table temperatures at a layer need the SAME eight values there -- only their
// coefficients differ.  A wavefront therefore owns 16 columns (one tile) of a PAIR of walkers: one
// set of slice loads per K-step feeds both operands (two sets where the pair's brackets differ at
// one of the step's four layers: a wave-uniform branch), the accumulators are the pair's 2 x MT
// row tiles, the A operands come from the two walkers' Q blocks.  The pairs are neighbours in an
// order of the walkers by their table brackets (k_walker_order), so that pairs share almost all of
// them whenever the walkers of a batch resemble one another (the walkers of a sampler do).
// LDS per workgroup: 2 x (Q blocks + radii + coefficients + offsets) = 73 KB at 80 layers.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_walker_order(int32_t *perm, int32_t *use_pair,
                                                      const int32_t *tlo, const double *temps,
                                                      int nlayers, int nwalkers, int itop,
                                                      int nimpact, int force)
{
    // rank sort of up to 1024 walkers by (sum of their brackets, mid-layer temperature, index),
    // then the share of K-steps (groups of four layers) in which the pairs of neighbours bracket
    // different table temperatures: the two-walker kernel pays for each of those with an exposed
    // load latency, beyond a fifth of the steps the one-walker kernel is the faster one
    __shared__ long long s_key[1024];
    __shared__ double s_t[1024];
    __shared__ int s_perm[1024];
    __shared__ int s_diff;
    const int w = threadIdx.x;
    if (w == 0)
        s_diff = 0;
    // keys: with up to 64 walkers sixteen lanes share a walker's layers (a serial loop over 80
    // layers per thread was most of the kernel's 25 us)
    const int per = nwalkers <= 64 ? 16 : 1;
    {
        const int ww = w / per, part = w % per;
        long long k = 0;
        if (ww < nwalkers)
            for (int l = part; l < nlayers; l += per)
                k += tlo[(int64_t)ww * nlayers + l];
        for (int d = per >> 1; d >= 1; d >>= 1)
            k += __shfl_down(k, d, 16);
        if (ww < nwalkers && part == 0) {
            s_key[ww] = k;
            s_t[ww] = temps[(int64_t)ww * nlayers + nlayers / 2];
        }
    }
    __syncthreads();
    if (w < nwalkers) {
        const long long k = s_key[w];
        const double t = s_t[w];
        int rank = 0;
        for (int v = 0; v < nwalkers; v++) {
            const long long kv = s_key[v];
            const double tv = s_t[v];
            rank += (kv < k) || (kv == k && (tv < t || (tv == t && v < w)));
        }
        perm[rank] = w;
        s_perm[rank] = w;
    }
    __syncthreads();
    const int npair = nwalkers >> 1;
    const int nsteps = (nimpact + 3) / 4;
    // one thread per (pair, K-step) while they fit the workgroup, else per pair
    const bool fine = npair * nsteps <= 1024;
    const int pi = fine ? w / max(nsteps, 1) : w;
    if (pi < npair && (fine ? w < npair * nsteps : true)) {
        const int32_t *ta = tlo + (int64_t)s_perm[2 * pi] * nlayers + itop;
        const int32_t *tb = tlo + (int64_t)s_perm[2 * pi + 1] * nlayers + itop;
        int diff = 0;
        const int k0 = fine ? w % nsteps : 0, k1 = fine ? k0 + 1 : nsteps;
        for (int ks = k0; ks < k1; ks++) {
            bool same = true;
            for (int q = 0; q < 4; q++) {
                const int r = min(4 * ks + q, nimpact - 1);
                same = same && ta[r] == tb[r];
            }
            diff += same ? 0 : 1;
        }
        if (diff)
            atomicAdd(&s_diff, diff);
    }
    __syncthreads();
    if (w == 0)
        *use_pair = force || (npair > 0 && 5 * (int64_t)s_diff <= (int64_t)npair * nsteps);
}

template <int MT, int WPS, int TB, int kS, int PD>
__global__ __launch_bounds__(TB, WPS) void k_table_transit_pair(
    double *spectrum, const double *etable, const int32_t *tlo, const double *coef,
    const double *qblk, const double *radius, const int32_t *perm, const int32_t *use_pair,
    int nblk, double rstar, int itop, int ibottom, double maxdepth, int nmol, int ntemp,
    int nlayers, int nwave, int nwalkers, int ncolblk)
{
    // per walker x: Q[nblk][64] | rad[16 MT] | coef[16 MT][2 kS] ; then off[2][16 MT] | same[4 MT]
    extern __shared__ __align__(16) double s_q[];
    if (!uniform_flag(use_pair))
        return;                                           // the one-walker kernel runs this batch
    constexpr int kPerWalker = qblocks(MT) * 64 + 16 * MT + 16 * MT * 2 * kS;
    const int npair = (nwalkers + 1) >> 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cb = (slot / npair) * 8 + xcd;
    const int pi = slot % npair;
    if (cb >= ncolblk)
        return;                                           // (whole workgroup, before the barrier)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nimpact = min(ibottom, nlayers) - itop;
    const bool haveb = 2 * pi + 1 < nwalkers;             // an odd batch: the last walker twice
    const int wx[2] = {perm[2 * pi], perm[haveb ? 2 * pi + 1 : 2 * pi]};
    double *s_rad[2], *s_coef[2];
    const double *s_qx[2];
    uint32_t *s_off = reinterpret_cast<uint32_t *>(s_q + 2 * kPerWalker);   // [2][16 MT]
    int32_t *s_same = reinterpret_cast<int32_t *>(s_off + 2 * 16 * MT);     // [4 MT]
#pragma unroll
    for (int x = 0; x < 2; x++) {
        double *base = s_q + x * kPerWalker;
        s_qx[x] = base;
        s_rad[x] = base + qblocks(MT) * 64;
        s_coef[x] = s_rad[x] + 16 * MT;
        const int w = wx[x];
        stage_qblocks<MT, TB>(base, qblk + (int64_t)w * nblk * 64, tid);
        for (int r = tid; r < 16 * MT; r += TB) {
            s_rad[x][r] = r < nimpact ? radius[(int64_t)w * nlayers + itop + r] : 0.0;
            const int rr = min(r, nimpact - 1);
            const int64_t t = tlo[(int64_t)w * nlayers + itop + rr];
            s_off[x * 16 * MT + r] = (uint32_t)(((t * nlayers + itop + rr) * (int64_t)nwave) * 8);
        }
        for (int e = tid; e < 16 * MT * 2 * kS; e += TB) {
            const int r = min(e / (2 * kS), nimpact - 1);
            s_coef[x][e] = coef[((int64_t)w * nlayers + itop + r) * 2 * kS + e % (2 * kS)];
        }
    }
    for (int ks = tid; ks < 4 * MT; ks += TB) {
        bool same = true;
        for (int q = 0; q < 4; q++) {
            const int rr = min(4 * ks + q, nimpact - 1);
            same = same && tlo[(int64_t)wx[0] * nlayers + itop + rr] ==
                               tlo[(int64_t)wx[1] * nlayers + itop + rr];
        }
        s_same[ks] = same ? 1 : 0;
    }
    __syncthreads();
    const int c0 = (cb * (TB / 64) + wave) * 16;
    if (c0 >= nwave)
        return;                                           // (after the only barrier)
    const int kq = lane >> 4, n = lane & 15;
    const bool ok = c0 + n < nwave;
    const int col = min(c0 + n, nwave - 1);
    const int64_t slice = (int64_t)nlayers * nwave;
    const char *base_lo[kS], *base_hi[kS];
#pragma unroll
    for (int sp = 0; sp < kS; sp++) {
        base_lo[sp] = reinterpret_cast<const char *>(etable + (int64_t)min(sp, nmol - 1) * ntemp * slice);
        base_hi[sp] = base_lo[sp] + slice * 8;
    }
    const uint32_t lane_off = (uint32_t)col * 8u;
    v4d C[2][MT];
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int m = 0; m < MT; m++)
            C[x][m] = v4d{0.0, 0.0, 0.0, 0.0};
    const int rot = pi & 3;                               // (see k_table_transit_mfma)
    auto kstep = [&](int ks) { return (ks & ~3) | ((ks + rot) & 3); };
    auto issue = [&](int x, int ks, double (&r)[2 * kS]) {
        const uint32_t off = s_off[x * 16 * MT + 4 * ks + kq] + lane_off;
#pragma unroll
        for (int sp = 0; sp < kS; sp++) {
            r[sp] = *reinterpret_cast<const double *>(base_lo[sp] + off);
            r[kS + sp] = *reinterpret_cast<const double *>(base_hi[sp] + off);
        }
    };
    auto combine = [&](int x, int ks, const double (&r)[2 * kS]) {
        const double *co = s_coef[x] + (4 * ks + kq) * 2 * kS;
        double a = 0.0;
#pragma unroll
        for (int sp = 0; sp < kS; sp++) {
            a = fma(r[sp], co[sp], a);
            a = fma(r[kS + sp], co[kS + sp], a);
        }
        return a;
    };
    double rawa[PD][2 * kS];
    double bcur[2], bnxt[2] = {0.0, 0.0};
    static_assert(kS == 4, "the second walker's own loads are written out for four species");
    auto fetch = [&](int ks, double (&ra)[2 * kS]) { issue(0, ks, ra); };
    // Where the pair's brackets differ at one of the K-step's four layers (wave-uniform flag; rare
    // between neighbours in bracket order) the second walker's operand needs slice values of its
    // own.  As a branch in the source that splits the unrolled loop into ~40 basic blocks and the
    // register allocator gives up (218-256 registers, spills); as loads in an asm statement that
    // the compiler cannot see complete, any copy it makes of their destination registers reads
    // stale data.  So the whole exception lives in ONE asm statement: skip if the brackets agree,
    // else eight loads, the wait for them and the eight fused multiply-adds of combine() -- in the
    // same order -- on scratch registers that are dead at its end.  Its latency is exposed; a
    // batch whose pairs disagree often runs the one-walker kernel instead (k_walker_order decides).
    auto form = [&](int ks, const double (&ra)[2 * kS], double (&b)[2]) {
        b[0] = combine(0, ks, ra);
        double bb = combine(1, ks, ra);
        const double *co = s_coef[1] + (4 * ks + kq) * 2 * kS;
        const int same = __builtin_amdgcn_readfirstlane(s_same[ks]);
        const uint32_t offb = s_off[16 * MT + 4 * ks + kq] + lane_off;
        double t0, t1, t2, t3, t4, t5, t6, t7;
        asm volatile("s_cmp_lg_u32 %[same], 0\n\t"
                     "s_cbranch_scc1 .Lpb_tt_skip%=\n\t"
                     "global_load_dwordx2 %[t0], %[off], %[l0]\n\t"
                     "global_load_dwordx2 %[t4], %[off], %[h0]\n\t"
                     "global_load_dwordx2 %[t1], %[off], %[l1]\n\t"
                     "global_load_dwordx2 %[t5], %[off], %[h1]\n\t"
                     "global_load_dwordx2 %[t2], %[off], %[l2]\n\t"
                     "global_load_dwordx2 %[t6], %[off], %[h2]\n\t"
                     "global_load_dwordx2 %[t3], %[off], %[l3]\n\t"
                     "global_load_dwordx2 %[t7], %[off], %[h3]\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "v_fma_f64 %[bb], %[t0], %[c0], 0\n\t"
                     "v_fma_f64 %[bb], %[t4], %[c4], %[bb]\n\t"
                     "v_fma_f64 %[bb], %[t1], %[c1], %[bb]\n\t"
                     "v_fma_f64 %[bb], %[t5], %[c5], %[bb]\n\t"
                     "v_fma_f64 %[bb], %[t2], %[c2], %[bb]\n\t"
                     "v_fma_f64 %[bb], %[t6], %[c6], %[bb]\n\t"
                     "v_fma_f64 %[bb], %[t3], %[c3], %[bb]\n\t"
                     "v_fma_f64 %[bb], %[t7], %[c7], %[bb]\n\t"
                     ".Lpb_tt_skip%=:"
                     : [bb] "+v"(bb), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
                       [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6),
                       [t7] "=&v"(t7)
                     : [same] "s"(same), [off] "v"(offb), [l0] "s"(base_lo[0]),
                       [l1] "s"(base_lo[1]), [l2] "s"(base_lo[2]), [l3] "s"(base_lo[3]),
                       [h0] "s"(base_hi[0]), [h1] "s"(base_hi[1]), [h2] "s"(base_hi[2]),
                       [h3] "s"(base_hi[3]), [c0] "v"(co[0]), [c1] "v"(co[1]), [c2] "v"(co[2]),
                       [c3] "v"(co[3]), [c4] "v"(co[4]), [c5] "v"(co[5]), [c6] "v"(co[6]),
                       [c7] "v"(co[7])
                     : "memory", "scc");
        b[1] = bb;
    };
#pragma unroll
    for (int d = 0; d < PD; d++)
        if (d < 4 * MT)
            fetch(kstep(d), rawa[d]);
    form(kstep(0), rawa[0], bcur);
    if (PD < 4 * MT)
        fetch(kstep(PD), rawa[0]);
    const double *sqa = s_qx[0] + lane, *sqb = s_qx[1] + lane;
#pragma unroll
    for (int ks = 0; ks < 4 * MT; ks++) {
        if (ks + 1 < 4 * MT) {
            // (the scheduling barriers keep a step's sums, loads and products apart: left to
            // itself the scheduler spreads them over the unrolled loop and needs 192-256 registers)
            form(kstep(ks + 1), rawa[(ks + 1) % PD], bnxt);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 + PD < 4 * MT)
                fetch(kstep(ks + 1 + PD), rawa[(ks + 1) % PD]);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int kse = kstep(ks);
#pragma unroll
        for (int m = ks / 4; m < MT; m++) {
            const double aa = sqa[(qblocks(m) + kse) * 64];
            const double ab = sqb[(qblocks(m) + kse) * 64];
            C[0][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa, bcur[0], C[0][m], 0, 0, 0);
            C[1][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(ab, bcur[1], C[1][m], 0, 0, 0);
        }
        bcur[0] = bnxt[0];
        bcur[1] = bnxt[1];
        __builtin_amdgcn_sched_barrier(0);
    }
    mfma_transit_epilogue_tile<MT>(C[0], s_rad[0], spectrum + (int64_t)wx[0] * nwave + c0 + n, ok,
                                   lane, nimpact, maxdepth, rstar);
    mfma_transit_epilogue_tile<MT>(C[1], s_rad[1], spectrum + (int64_t)wx[1] * nwave + c0 + n,
                                   ok && haveb, lane, nimpact, maxdepth, rstar);
}

#endif  // PB_EXPERIMENTS

#ifdef PB_EXPERIMENTS   // LDS column tile of the fused transit kernel (5x slower; A/B only)
// ---------------------------------------------------------------------------
// The same pass with the column tile in LDS: workgroup = 64 columns x NB wavefronts, wavefront b
// owning the impact parameters 16b .. 16b+15.  The tile of s_i = ec[i+1] + ec[i] (64 columns x all
// segments) is read from HBM ONCE, cooperatively and coalesced, and every wavefront then takes
// its operands from LDS (the one-thread-per-column form above re-reads the column once per row
// block: 3x the bytes at 80 layers, all from HBM once the batch outgrows the caches).  The early
// exit and the transmission integral then run down the rows wavefront after wavefront, the
// carried state (first crossing, trapezoid sum, previous integrand) passing through LDS.
// ---------------------------------------------------------------------------
constexpr int kTileRows = 16;

__global__ __launch_bounds__(1024) void k_transit_tile(
    double *depth, int32_t *ideep, double *spectrum, const double *ec, const double *raypath,
    const double *radius, int64_t npath, double rstar, int itop, int ibottom, double maxdepth,
    int nlayers, int nwave, int deck_row, double rsurf)
{
    extern __shared__ __align__(16) double s_mem[];
    const int w = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const bool active = col < nwave;
    const int64_t plane = (int64_t)nlayers * nwave;
    ec += (int64_t)w * plane;
    if (depth)
        depth += (int64_t)w * plane;
    const double *path = raypath ? raypath + (int64_t)w * npath : nullptr;
    const double *rad = radius ? radius + (int64_t)w * nlayers : nullptr;
    const int nimpact = min(ibottom, nlayers) - itop;     // rows 0..nimpact-1 are evaluated
    const int nseg_all = max(nimpact - 1, 0);
    double *s_tile = s_mem;                               // [segment][64]
    double *s_carry = s_tile + (size_t)nseg_all * 64;     // [3][64]: acc, fprev, rprev
    int *s_stop = reinterpret_cast<int *>(s_carry + 3 * 64);   // [64]
    double *s_path = reinterpret_cast<double *>(s_stop + 64);  // per wavefront [segment][16]

    // the tile: row i of s = ec[itop+i+1] + ec[itop+i]; wavefront v loads rows v, v+nwaves, ...
    // (each row 512 contiguous bytes); the two operands of a row are two coalesced loads
    {
        const double *src = ec + (int64_t)itop * nwave + (active ? col : 0);
        for (int i = wave; i < nseg_all; i += nwaves) {
            const double a = src[(int64_t)i * nwave], b = src[(int64_t)(i + 1) * nwave];
            s_tile[i * 64 + lane] = b + a;
        }
    }
    // my block of rows and its ray paths ([segment][row], zero where segment >= row)
    const int rb = wave * kTileRows;
    const int rlast = min(rb + kTileRows, nimpact) - 1;
    const int nseg = rb <= rlast ? max(rlast, 0) : 0;
    int poff = 0;                                         // doubles before my block
    for (int v = 0; v < wave; v++)
        poff += max(min(v * kTileRows + kTileRows, nimpact) - 1, 0) * kTileRows;
    double *mypath = s_path + poff;
    for (int e = lane; e < nseg * kTileRows; e += 64) {
        const int i = e / kTileRows, k = e % kTileRows;
        const int r = rb + k;
        mypath[e] = (r <= rlast && i < r) ? path[((int64_t)r * (r - 1)) / 2 + i] : 0.0;
    }
    if (threadIdx.x < 64) {
        s_stop[lane] = -1;
        s_carry[lane] = 0.0;
        s_carry[64 + lane] = 0.0;
        s_carry[128 + lane] = 0.0;
    }
    __syncthreads();
    double tau[kTileRows];
#pragma unroll
    for (int k = 0; k < kTileRows; k++)
        tau[k] = 0.0;
    for (int i = 0; i < nseg; i++) {
        const double s = s_tile[i * 64 + lane];
        const double *pk = mypath + i * kTileRows;              // LDS broadcast reads
#pragma unroll
        for (int k = 0; k < kTileRows; k++)
            tau[k] += pk[k] * s;
    }
    if (depth && active && wave == 0)
        for (int r = 0; r < itop; r++)
            depth[(int64_t)r * nwave + col] = 0.0;
    // the rows in order, one wavefront after the other
    for (int v = 0; v < nwaves; v++) {
        if (v == wave && rb <= rlast) {
            int stop = s_stop[lane];
            double acc = s_carry[lane], fprev = s_carry[64 + lane], rprev = s_carry[128 + lane];
#pragma unroll
            for (int k = 0; k < kTileRows; k++) {
                const int r = rb + k;
                if (r > rlast)
                    break;
                double t = tau[k];
                if (stop < 0) {
                    if (spectrum) {
                        const double rr = rad[itop + r];
                        double f = pb::exp_s(-t) * rr;
                        if (r > 0 && r == deck_row) {
                            f = deck_integrand(fprev, f, rprev, rr, rsurf);
                            acc += (rsurf - rprev) * (fprev + f);
                        } else if (r > 0) {
                            acc += (rr - rprev) * (fprev + f);
                        }
                        fprev = f;
                        rprev = rr;
                    }
                    if (t > maxdepth)
                        stop = r;
                } else {
                    t = 0.0;
                }
                if (depth && active)
                    depth[(int64_t)(itop + r) * nwave + col] = t;
            }
            s_stop[lane] = stop;
            s_carry[lane] = acc;
            s_carry[64 + lane] = fprev;
            s_carry[128 + lane] = rprev;
        }
        __syncthreads();
    }
    if (!active)
        return;
    if (depth)
        for (int r = max(nimpact, 0) + wave; r < nlayers - itop; r += nwaves)
            depth[(int64_t)(itop + r) * nwave + col] = 0.0;   // rows at and below ibottom
    if (wave == 0) {
        const int stop = s_stop[lane];
        const int last = nimpact > 0 ? itop + nimpact - 1 : itop;
        if (ideep)
            ideep[(int64_t)w * nwave + col] = stop >= 0 ? itop + stop : last;
        if (spectrum) {
            const double rtop = rad[itop];
            spectrum[(int64_t)w * nwave + col] =
                (rtop * rtop + 2 * (s_carry[lane] * 0.5)) / (rstar * rstar);
        }
    }
}

#endif  // PB_EXPERIMENTS

// ---------------------------------------------------------------------------
// PassBand.integrate for a batch of spectra: grid (band, walker); fixed-order tree sum.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_band_integrate_batch(
    double *bandflux, const double *spectrum, const double *wn, const int32_t *band_start,
    const int32_t *band_count, const double *response, const int64_t *response_offset,
    const double *heights, int nbands, int nwave)
{
    __shared__ double s_part[kBlock];
    const int b = blockIdx.x, w = blockIdx.y;
    const int start = band_start[b];
    const int count = band_count[b];
    const double *resp = response + response_offset[b];
    const double *spec = spectrum + (int64_t)w * nwave;
    double acc = 0.0;
    for (int i = threadIdx.x; i + 1 < count; i += kBlock) {
        const int64_t g = (int64_t)start + i;
        const double y0 = spec[g] * resp[i];
        const double y1 = spec[g + 1] * resp[i + 1];
        acc += 0.5 * (wn[g + 1] - wn[g]) * (y0 + y1);
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (int h = kBlock / 2; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h)
            s_part[threadIdx.x] += s_part[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        bandflux[(int64_t)w * nbands + b] = heights ? s_part[0] * heights[b] : s_part[0];
}

// walkers whose temperatures leave the table: every band flux = +inf (eval()'s reject value,
// pyrat_obj.py:302-320, 378-380)
__global__ __launch_bounds__(kBlock) void k_reject_walkers(double *bandflux, const double *temps,
                                                           double tmin, double tmax, int nlayers,
                                                           int nbands)
{
    __shared__ int s_bad;
    const int w = blockIdx.x;
    if (threadIdx.x == 0)
        s_bad = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < nlayers; k += kBlock) {
        const double t = temps[(int64_t)w * nlayers + k];
        if (!(t >= tmin && t <= tmax))
            s_bad = 1;
    }
    __syncthreads();
    if (s_bad)
        for (int b = threadIdx.x; b < nbands; b += kBlock)
            bandflux[(int64_t)w * nbands + b] = INFINITY;
}

// ---------------------------------------------------------------------------
// Loader of sampled cross sections: one species' table of one opacity file brought onto the
// run's (temperature, pressure, wavenumber) grid -- tools.interpolate_opacity
// (pyratbay/tools/tools.py:1026-1107) as called by Line_Sample.__init__
// (opacity/line_sampling.py:245-275): linear in log(cs) over log(p), then over T, constant
// beyond the table; a zero cross section enters as exp(-230).  The brackets and weights of the
// two axes are prepared on the host (a handful of values); `wsel` are the kept wavenumber
// samples (window + thinning); `accumulate` adds to what the table holds (a species spread over
// several files is the sum of its files).  kResample = false: the grids agree with the file's,
// values are copied (or added) untouched, like the reference does.
// ---------------------------------------------------------------------------
template <bool kResample>
__global__ __launch_bounds__(kBlock) void k_resample_cs(
    double *out, const double *in, const int32_t *wsel, const int32_t *tlo, const double *ta,
    const int32_t *plo, const double *pa, int nlay_in, int nwave_in, int ntemp_out, int nlay_out,
    int nwave_out, int accumulate)
{
    const int w = blockIdx.x * kBlock + threadIdx.x;
    const int p2 = blockIdx.y, t2 = blockIdx.z;
    if (w >= nwave_out)
        return;
    const int64_t wi = wsel[w];
    auto at = [&](int t, int p) { return in[((int64_t)t * nlay_in + p) * nwave_in + wi]; };
    double v;
    if (!kResample) {
        v = at(tlo[t2], plo[p2]);
    } else {
        auto lg = [&](int t, int p) {
            const double y = log(at(t, p));
            return isfinite(y) ? y : -230.0;
        };
        auto over_p = [&](int t) {
            const double a = pa[p2];
            const int p = plo[p2];
            if (a == 0.0)
                return lg(t, p);
            const double lo = lg(t, p), hi = lg(t, p + 1);
            return lo + a * (hi - lo);                   // np.interp / slinear: lo + slope*(x - xlo)
        };
        const double b = ta[t2];
        const int t = tlo[t2];
        double y = over_p(t);
        if (b != 0.0) {
            const double hi = over_p(t + 1);
            y = y + b * (hi - y);
        }
        v = exp(y);
    }
    const int64_t o = ((int64_t)t2 * nlay_out + p2) * nwave_out + w;
    out[o] = accumulate ? out[o] + v : v;
}

}  // namespace

// rows per block of the fused kernel for a launch of nwave x nwalkers columns
static int fused_rows(int nwave, int nwalkers, int nrow, bool scalar_path)
{
    // measured at C5's shape (64 walkers x 1e5 columns x 80 layers, ray paths in SGPRs): 8 rows
    // per thread 3.97 ms, 16 rows 3.55 ms, 40 rows 3.70 ms (157 registers); LDS-staged paths 4.68
    int rows = (int64_t)nwave * nwalkers <= 32768 ? 8 : 16;
    (void)scalar_path;
    if (const char *e = getenv("PB_TRANSIT_ROWS"))
        rows = atoi(e) >= 40 ? 40 : atoi(e) >= 16 ? 16 : 8;
    while (rows > 8 && (size_t)std::max(nrow, 1) * rows * 8 > 64 * 1024)
        rows = rows == 40 ? 16 : 8;
    return rows;
}

static int64_t blocked_len(int rows, int nimpact)
{
    int64_t n = 0;
    for (int rb = 0; rb < nimpact; rb += rows)
        n += (int64_t)std::max(std::min(rb + rows, nimpact) - 1, 0) * rows;
    return n;
}

// shared with pb_columns.hip: the single-spectrum entries can use the fused kernel too.
// work_d: nwalkers * blocked_len doubles of scratch for the scalar-path form, or NULL (ray paths
// staged in LDS).
int pb_transit_fused_launch(double *depth_d, int32_t *ideep_d, double *spectrum_d,
                            const double *ec_d, const double *raypath_d, const double *radius_d,
                            int64_t npath, double rstar, int itop, int ibottom, double maxdepth,
                            int nlayers, int nwave, int nwalkers, int deck_row, double rsurf,
                            hipStream_t s, double *work_d, const int32_t *scatter_d,
                            const int32_t *tile_limit_d, int32_t *flags_d, const int32_t *gate_d)
{
    // (PB_C5_PREFETCH_ALL=1: the next tile's layers are requested whatever the limit -- A/B)
    static const bool prefetch_all = getenv("PB_C5_PREFETCH_ALL") && atoi(getenv("PB_C5_PREFETCH_ALL"));
    const TileLimit lim{tile_limit_d, prefetch_all ? -1 : itop, gate_d};
    const int nrow = nlayers - itop;
    const int nimpact = std::min(ibottom, nlayers) - itop;
#ifdef PB_EXPERIMENTS
    // PB_TRANSIT_TILE=1: the LDS-tile form (measured slower: 10.3 ms against 4.5 ms per 64-walker
    // batch at C5's shape -- five wavefronts of uneven length per 70 KB of LDS); kept for A/B
    {
        const int nb = std::max(1, pb::div_up(std::max(nimpact, 1), kTileRows));
        size_t npd = 0;
        for (int v = 0; v < nb; v++)
            npd += (size_t)std::max(std::min(v * kTileRows + kTileRows, nimpact) - 1, 0) * kTileRows;
        const size_t tl = ((size_t)std::max(nimpact - 1, 0) * 64 + 3 * 64 + npd) * 8 + 64 * 4;
        static const bool on = getenv("PB_TRANSIT_TILE") && atoi(getenv("PB_TRANSIT_TILE")) != 0;
        if (on && nb <= 16 && tl <= 150 * 1024) {
            if (tl > 64 * 1024)
                PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_transit_tile),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl));
            dim3 tgrid(pb::div_up(nwave, 64), nwalkers);
            k_transit_tile<<<tgrid, nb * 64, tl, s>>>(depth_d, ideep_d, spectrum_d, ec_d, raypath_d,
                                                     radius_d, npath, rstar, itop, ibottom,
                                                     maxdepth, nlayers, nwave, deck_row, rsurf);
            PB_LAUNCH_CHECK();
            return PB_OK;
        }
    }
#endif  // PB_EXPERIMENTS
    // the retrieval batch (spectrum only, no deck) on the matrix cores: k_transit_mfma_rows
    {
        // (read per call: the tests switch it inside one process)
        const char *mode = getenv("PB_TRANSIT_MFMA");
        const bool no_mfma = mode && atoi(mode) == 0 && !scatter_d;   // (ordered: this form only)
        const int mt = pb::div_up(std::max(nimpact, 1), 16);
        PB_REQUIRE(!scatter_d || (work_d && spectrum_d && !depth_d && !ideep_d && deck_row < 0 &&
                                  nimpact > 1 && mt <= 8 && nwave >= 2),
                   "pb_transit_spectrum_ordered: 2 ... 128 impact parameters and at least 2 "
                   "columns (got %d, %d)", nimpact, nwave);
        if (!no_mfma && work_d && spectrum_d && !depth_d && !ideep_d && deck_row < 0 &&
            nimpact > 1 && mt <= 8 && nwave >= 2 && nwalkers >= 1) {
            const int nblk = qblocks(mt);
            dim3 qgrid((unsigned)std::min(16, pb::div_up((int64_t)nblk * 64, kBlock)), nwalkers);
            if (!gate_d) {      // (a gated repair pass re-uses the Q blocks of its first pass)
                k_path_qblocks<<<qgrid, kBlock, 0, s>>>(work_d, raypath_d, npath, nblk, nimpact);
                PB_LAUNCH_CHECK();
            }
            const size_t lds = ((size_t)nblk * 64 + (size_t)mt * 16) * 8;
#ifdef PB_EXPERIMENTS
#define PB_MFMA(M, W, T)                                                                         \
    do {                                                                                         \
        if (lds > 64 * 1024)                                                                     \
            PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_transit_mfma<M, W, T>), \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   \
        dim3 mgrid(pb::div_up(nwave, (T / 64) * 32), nwalkers);                                  \
        k_transit_mfma<M, W, T><<<mgrid, T, lds, s>>>(spectrum_d, ec_d, work_d, radius_d, nblk,  \
                                                     rstar, itop, ibottom, maxdepth, nlayers,    \
                                                     nwave);                                     \
    } while (0)
#endif  // PB_EXPERIMENTS
            // threads per workgroup: a workgroup stages its walker's Q blocks (30 KB at 80
            // layers) once for TB / 64 x 32 columns
            const int tb = getenv("PB_MFMA_TB") ? atoi(getenv("PB_MFMA_TB")) : 512;
            // row tile by row tile with the early exit (the default: 1.48 against 1.59 ms per 64
            // walkers at C5's shape with the columns in grid order, 1.16 with ordered columns);
            // PB_TRANSIT_MFMA=4 (experiments build): the layers-outer kernel it replaced, for A/B
#ifdef PB_EXPERIMENTS
            if (scatter_d || !(mode && atoi(mode) == 4))
#endif
            {
#define PB_MFMA_ROWS(M, W, T)                                                                    \
    do {                                                                                         \
        if (lds > 64 * 1024)                                                                     \
            PB_HIP(hipFuncSetAttribute(                                                          \
                reinterpret_cast<const void *>(k_transit_mfma_rows<M, W, T>),                    \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                          \
        dim3 mgrid(pb::div_up(nwave, (T / 64) * 32), nwalkers);                                  \
        k_transit_mfma_rows<M, W, T><<<mgrid, T, lds, s>>>(spectrum_d, ec_d, work_d, radius_d,   \
                                                          nblk, rstar, itop, ibottom, maxdepth,  \
                                                          nlayers, nwave, scatter_d, lim,        \
                                                          flags_d);                              \
    } while (0)
                switch (mt) {
                case 1: PB_MFMA_ROWS(1, 4, 256); break;
                case 2: PB_MFMA_ROWS(2, 4, 256); break;
                case 3: PB_MFMA_ROWS(3, 4, 256); break;
                case 4: PB_MFMA_ROWS(4, 4, 256); break;
                case 5:
                    // 80 layers: the B operands alone are 80 registers.  Three wavefronts per SIMD
                    // of up to 168 registers (no spills) beat four of 128 (24 spilled): 1.23
                    // against 1.31 ms per 64 walkers at C5's shape; PB_MFMA_TB=512|1024: the latter
                    if (getenv("PB_MFMA_TB") && tb >= 1024)
                        PB_MFMA_ROWS(5, 4, 1024);
                    else if (getenv("PB_MFMA_TB") && tb >= 512)
                        PB_MFMA_ROWS(5, 4, 512);
                    else
                        PB_MFMA_ROWS(5, 3, 256);
                    break;
                case 6: PB_MFMA_ROWS(6, 2, 256); break;
                case 7: PB_MFMA_ROWS(7, 2, 256); break;
                default: PB_MFMA_ROWS(8, 2, 256); break;
                }
#undef PB_MFMA_ROWS
                PB_LAUNCH_CHECK();
                return PB_OK;
            }
#ifdef PB_EXPERIMENTS
            switch (mt) {
            case 1: PB_MFMA(1, 4, 256); break;
            case 2: PB_MFMA(2, 4, 256); break;
            case 3: PB_MFMA(3, 4, 256); break;
            case 4: PB_MFMA(4, 4, 256); break;
            case 5:
                if (tb >= 1024)
                    PB_MFMA(5, 4, 1024);
                else if (tb >= 512)
                    PB_MFMA(5, 4, 512);
                else
                    PB_MFMA(5, 4, 256);
                break;
            case 6: PB_MFMA(6, 2, 256); break;
            case 7: PB_MFMA(7, 2, 256); break;
            default: PB_MFMA(8, 2, 256); break;
            }
#undef PB_MFMA
            PB_LAUNCH_CHECK();
            return PB_OK;
#endif  // PB_EXPERIMENTS
        }
    }
    static const bool no_scalar = getenv("PB_TRANSIT_SCALAR") && atoi(getenv("PB_TRANSIT_SCALAR")) == 0;
    const bool scalar = work_d != nullptr && nimpact > 1 && !no_scalar;
    const int rows = fused_rows(nwave, nwalkers, nrow, scalar);
    dim3 grid(pb::div_up(nwave, kBlock), nwalkers);
    const double *path_d = raypath_d;
    int64_t plen = npath;
    if (scalar) {
        plen = blocked_len(rows, nimpact);
        dim3 bgrid((unsigned)std::min<int64_t>(64, pb::div_up(plen, kBlock)), nwalkers);
        k_path_blocks<<<bgrid, kBlock, 0, s>>>(work_d, raypath_d, npath, plen, rows, nimpact);
        PB_LAUNCH_CHECK();
        path_d = work_d;
    }
#define PB_FUSED(R, SC)                                                                          \
    k_transit_fused<R, SC, false><<<grid, kBlock, SC ? 0 : (size_t)std::max(nrow, 1) * R * 8, s>>>( \
        depth_d, ideep_d, spectrum_d, ec_d, path_d, radius_d, plen, rstar, itop, ibottom,        \
        maxdepth, nlayers, nwave, deck_row, rsurf)
    static const bool no_fma = getenv("PB_TRANSIT_FMA") && atoi(getenv("PB_TRANSIT_FMA")) == 0;
    static const bool no_pair = getenv("PB_TRANSIT_PAIR") && atoi(getenv("PB_TRANSIT_PAIR")) == 0;
    if (scalar && rows == 16 && !depth_d && !ideep_d && nwalkers > 1 && !no_fma && !no_pair &&
        deck_row < 0 && spectrum_d) {
        // the retrieval batch, two columns per thread
        dim3 pgrid(pb::div_up(nwave, 2 * kBlock), nwalkers);
        k_transit_pair<16><<<pgrid, kBlock, 0, s>>>(spectrum_d, ec_d, path_d, radius_d, plen, rstar,
                                                   itop, ibottom, maxdepth, nlayers, nwave);
    } else if (scalar && rows == 16 && !depth_d && !ideep_d && nwalkers > 1 && !no_fma) {
        // the retrieval batch: spectrum only
        k_transit_fused<16, true, true><<<grid, kBlock, 0, s>>>(
            depth_d, ideep_d, spectrum_d, ec_d, path_d, radius_d, plen, rstar, itop, ibottom,
            maxdepth, nlayers, nwave, deck_row, rsurf);
    } else if (scalar) {
        if (rows == 40)
            PB_FUSED(40, true);
        else if (rows == 16)
            PB_FUSED(16, true);
        else
            PB_FUSED(8, true);
    } else {
        if (rows == 40)
            PB_FUSED(40, false);
        else if (rows == 16)
            PB_FUSED(16, false);
        else
            PB_FUSED(8, false);
    }
#undef PB_FUSED
    PB_LAUNCH_CHECK();
    return PB_OK;
}

// the blocked ray-path layout for one spectrum in the stream's persistent scratch (pb_core.hip);
// PB_ERR_NOMEM when there is none: the caller falls back to the LDS form
int pb_path_blocks_launch(double **blocked_d, int64_t *len, const double *raypath_d, int64_t npath,
                          int rows, int nimpact, hipStream_t s)
{
    const int64_t plen = blocked_len(rows, nimpact);
    *blocked_d = nullptr;
    *len = plen;
    if (plen <= 0)
        return PB_ERR_ARG;
    double *buf = reinterpret_cast<double *>(pb::stream_scratch(s, (size_t)plen * 8));
    if (!buf)
        return PB_ERR_NOMEM;
    dim3 bgrid((unsigned)std::min<int64_t>(64, pb::div_up(plen, kBlock)), 1);
    k_path_blocks<<<bgrid, kBlock, 0, s>>>(buf, raypath_d, npath, plen, rows, nimpact);
    if (hipGetLastError() != hipSuccess)
        return PB_ERR_HIP;
    *blocked_d = buf;
    return PB_OK;
}

extern "C" {

int pb_transit_path(double *raypath_d, const double *radius_d, int itop, int nlayers,
                    int nwalkers, void *stream)
{
    PB_REQUIRE(nlayers >= 1 && itop >= 0 && itop < nlayers && nwalkers >= 0,
               "pb_transit_path: bad shape");
    const int nrow = nlayers - itop;
    if (nwalkers == 0 || nrow < 2)
        return PB_OK;
    PB_REQUIRE(raypath_d && radius_d, "pb_transit_path: null pointer");
    const int64_t npath = ((int64_t)nrow * (nrow - 1)) / 2;
    dim3 grid(std::min(nrow, 64), nwalkers);
    k_transit_path<<<grid, kBlock, 0, pb::as_stream(stream)>>>(raypath_d, radius_d, itop, nlayers,
                                                             npath);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_iso_partition(double *z_d, int64_t z_iso_stride, int64_t z_t_stride,
                     const double *temp_d, int64_t ntemp, const double *ttab_d, int ntab,
                     const double *pf_d, int niso, int32_t *nbad_d, void *stream)
{
    PB_REQUIRE(ntemp >= 0 && niso >= 0 && ntab >= 2, "pb_iso_partition: bad shape (a partition-"
               "function table needs at least two temperatures, got %d)", ntab);
    if (ntemp == 0 || niso == 0)
        return PB_OK;
    PB_REQUIRE(z_d && temp_d && ttab_d && pf_d, "pb_iso_partition: null pointer");
    k_iso_partition<<<(unsigned)pb::div_up(ntemp, kBlock), kBlock, 0, pb::as_stream(stream)>>>(
        z_d, z_iso_stride, z_t_stride, temp_d, ntemp, ttab_d, ntab, pf_d, niso, nbad_d);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

static int interp_ec_batch_launch(double *ec_d, const double *etable_d, const double *ttable_d,
                                  const double *temps_d, const double *density_d, void *work_d,
                                  int nmol, int ntemp, int nlayers, int nwave, int nwalkers,
                                  TileLimit lim, void *stream);

int pb_interp_ec_batch(double *ec_d, const double *etable_d, const double *ttable_d,
                       const double *temps_d, const double *density_d, void *work_d, int nmol,
                       int ntemp, int nlayers, int nwave, int nwalkers, void *stream)
{
    return interp_ec_batch_launch(ec_d, etable_d, ttable_d, temps_d, density_d, work_d, nmol, ntemp,
                                  nlayers, nwave, nwalkers, TileLimit{nullptr, 0, nullptr}, stream);
}

int pb_interp_ec_batch_limited(double *ec_d, const double *etable_d, const double *ttable_d,
                               const double *temps_d, const double *density_d, void *work_d,
                               int nmol, int ntemp, int nlayers, int nwave, int nwalkers,
                               const int32_t *tile_limit_d, int row0, const int32_t *gate_d,
                               void *stream)
{
    PB_REQUIRE(row0 >= 0 && row0 < std::max(nlayers, 1), "pb_interp_ec_batch_limited: row0 out of range");
    return interp_ec_batch_launch(ec_d, etable_d, ttable_d, temps_d, density_d, work_d, nmol, ntemp,
                                  nlayers, nwave, nwalkers, TileLimit{tile_limit_d, row0, gate_d},
                                  stream);
}

static int interp_ec_batch_launch(double *ec_d, const double *etable_d, const double *ttable_d,
                                  const double *temps_d, const double *density_d, void *work_d,
                                  int nmol, int ntemp, int nlayers, int nwave, int nwalkers,
                                  TileLimit lim, void *stream)
{
    PB_REQUIRE(nmol >= 1 && nmol <= 8 && ntemp >= 2 && nlayers >= 1 && nwave >= 0 &&
                   nwalkers >= 0,
               "pb_interp_ec_batch: bad shape (1-8 species, >= 2 table temperatures)");
    if (nwave == 0 || nwalkers == 0)
        return PB_OK;
    PB_REQUIRE(ec_d && etable_d && ttable_d && temps_d && density_d && work_d,
               "pb_interp_ec_batch: null pointer");
    hipStream_t s = pb::as_stream(stream);
    const int64_t n = (int64_t)nwalkers * nlayers;
    const int ncoef = nmol <= 4 ? 4 : 8;
    // workspace: coef[n][2*ncoef] doubles, then tlo[n] ints
    double *coef = reinterpret_cast<double *>(work_d);
    int32_t *tlo = reinterpret_cast<int32_t *>(coef + n * 2 * ncoef);
    // (a gated repair pass runs on the workspace its first pass filled: same walkers, same weights)
    if (!lim.gate) {
        k_interp_weights<<<pb::div_up(n, kBlock), kBlock, 0, s>>>(tlo, coef, ttable_d, temps_d,
                                                                density_d, nmol, ncoef, ntemp, n);
        PB_LAUNCH_CHECK();
    }
    // walkers per chunk: every chunk reads the table slices its walkers bracket again, so as many
    // as the launch can afford while it still fills the chip (C5, 64 walkers: 1.40 ms in chunks
    // of 16, 1.23 in one chunk, 1.11 with the species count a template constant)
    int chunk = 64;
    if (const char *e = getenv("PB_INTERP_CHUNK"))
        chunk = std::max(1, atoi(e));
    while (chunk > 1 && (int64_t)pb::div_up(nwave, kBlock) * nlayers * pb::div_up(nwalkers, chunk) < 2048)
        chunk /= 2;
    dim3 grid(pb::div_up(nwave, kBlock), nlayers, pb::div_up(nwalkers, chunk));
    // two samples per thread when every row of every slice / walker has the parity of its layer
    // index times nwave (slice = nlayers * nwave even, or nwave even), and 16-byte aligned bases
    static const bool pairs_on = !(getenv("PB_INTERP_PAIRS") && atoi(getenv("PB_INTERP_PAIRS")) == 0);
    const bool pairs = pairs_on && nwave >= 4 && (((int64_t)nlayers * nwave) % 2 == 0) &&
                       ((uintptr_t)ec_d % 16 == 0) && ((uintptr_t)etable_d % 16 == 0);
    // two pair slots per thread for launches that still fill the chip with half the workgroups
    // (up to four species: with eight, 146 registers would cost a wavefront per SIMD)
    int np = 1;
    if (pairs && nmol <= 4 && (int64_t)pb::div_up(nwave / 2 + 2, 2 * kBlock) * nlayers *
                         pb::div_up(nwalkers, chunk) >= 4096)
        np = 2;
    if (const char *e = getenv("PB_INTERP_NP"))
        np = atoi(e) == 2 ? 2 : 1;
    if (pairs)
        grid.x = pb::div_up(nwave / 2 + 2, kBlock * np);
#define PB_INTERP(S, FULL)                                                                     \
    do {                                                                                       \
        if (pairs && np == 2)                                                                  \
            k_interp_ec_batch2<S, FULL, 2><<<grid, kBlock, 0, s>>>(ec_d, etable_d, tlo, coef,  \
                                                                   nmol, ntemp, nlayers, nwave, \
                                                                   nwalkers, chunk, lim);      \
        else if (pairs)                                                                        \
            k_interp_ec_batch2<S, FULL, 1><<<grid, kBlock, 0, s>>>(ec_d, etable_d, tlo, coef,  \
                                                                   nmol, ntemp, nlayers, nwave, \
                                                                   nwalkers, chunk, lim);      \
        else                                                                                   \
            k_interp_ec_batch<S, FULL><<<grid, kBlock, 0, s>>>(ec_d, etable_d, tlo, coef, nmol,  \
                                                               ntemp, nlayers, nwave, nwalkers, chunk, lim); \
    } while (0)
    static const bool no_full = getenv("PB_INTERP_FULL") && atoi(getenv("PB_INTERP_FULL")) == 0;
    if (nmol == 4 && !no_full)
        PB_INTERP(4, true);
    else if (nmol <= 4)
        PB_INTERP(4, false);
    else if (nmol == 8)
        PB_INTERP(8, true);
    else
        PB_INTERP(8, false);
#undef PB_INTERP
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_resample_cross_section(double *out_d, const double *in_d, const int32_t *wsel_d,
                              const int32_t *tlo_d, const double *tweight_d,
                              const int32_t *plo_d, const double *pweight_d, int ntemp_in,
                              int nlay_in, int nwave_in, int ntemp_out, int nlay_out,
                              int nwave_out, int resample, int accumulate, void *stream)
{
    PB_REQUIRE(ntemp_in >= 1 && nlay_in >= 1 && nwave_in >= 1 && ntemp_out >= 1 && nlay_out >= 1 &&
                   nwave_out >= 0,
               "pb_resample_cross_section: bad shape");
    if (nwave_out == 0)
        return PB_OK;
    PB_REQUIRE(out_d && in_d && wsel_d && tlo_d && tweight_d && plo_d && pweight_d,
               "pb_resample_cross_section: null pointer");
    PB_REQUIRE(nlay_out <= 65535 && ntemp_out <= 65535, "pb_resample_cross_section: grid too large");
    dim3 grid(pb::div_up(nwave_out, kBlock), nlay_out, ntemp_out);
    if (resample)
        k_resample_cs<true><<<grid, kBlock, 0, pb::as_stream(stream)>>>(
            out_d, in_d, wsel_d, tlo_d, tweight_d, plo_d, pweight_d, nlay_in, nwave_in, ntemp_out,
            nlay_out, nwave_out, accumulate);
    else
        k_resample_cs<false><<<grid, kBlock, 0, pb::as_stream(stream)>>>(
            out_d, in_d, wsel_d, tlo_d, tweight_d, plo_d, pweight_d, nlay_in, nwave_in, ntemp_out,
            nlay_out, nwave_out, accumulate);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int64_t pb_transit_work_doubles(int nlayers, int itop, int ibottom, int nwave, int nwalkers)
{
    if (nlayers < 1 || itop < 0 || itop >= nlayers)
        return 0;
    const int nimpact = std::min(ibottom, nlayers) - itop;
    // the largest layout any row-block choice needs
    int64_t n = 0;
    for (int rows : {8, 16, 40})
        n = std::max(n, blocked_len(rows, nimpact));
    // ... and the 16 x 4 blocks of the matrix-core form (k_path_qblocks)
    n = std::max<int64_t>(n, (int64_t)qblocks(pb::div_up(std::max(nimpact, 1), 16)) * 64);
    (void)nwave;
    return n * std::max(nwalkers, 0) + 8;
}

int pb_transit_spectrum_batch(double *spectrum_d, double *depth_d, int32_t *ideep_d,
                              const double *ec_d, const double *raypath_d,
                              const double *radius_d, double rstar, int itop, int ibottom,
                              double maxdepth, int nlayers, int nwave, int nwalkers,
                              void *work_d, void *stream)
{
    PB_REQUIRE(nlayers > 0 && nwave >= 0 && nwalkers >= 0, "pb_transit_spectrum_batch: bad shape");
    PB_REQUIRE(itop >= 0 && itop < nlayers, "pb_transit_spectrum_batch: itop out of range");
    PB_REQUIRE(ibottom <= nlayers, "pb_transit_spectrum_batch: ibottom > nlayers");
    if (nwave == 0 || nwalkers == 0)
        return PB_OK;
    const int nrow = nlayers - itop;
    PB_REQUIRE(spectrum_d && ec_d && radius_d && (nrow == 1 || raypath_d),
               "pb_transit_spectrum_batch: null pointer");
    const int64_t npath = ((int64_t)nrow * (nrow - 1)) / 2;
    return pb_transit_fused_launch(depth_d, ideep_d, spectrum_d, ec_d, raypath_d, radius_d, npath,
                                   rstar, itop, ibottom, maxdepth, nlayers, nwave, nwalkers, -1,
                                   0.0, pb::as_stream(stream), reinterpret_cast<double *>(work_d),
                                   nullptr, nullptr, nullptr, nullptr);
}

int pb_transit_spectrum_ordered(double *spectrum_d, const double *ec_d, const double *raypath_d,
                                const double *radius_d, const int32_t *column_d, double rstar,
                                int itop, int ibottom, double maxdepth, int nlayers, int nwave,
                                int nwalkers, void *work_d, void *stream)
{
    PB_REQUIRE(nlayers > 0 && nwave >= 0 && nwalkers >= 0, "pb_transit_spectrum_ordered: bad shape");
    PB_REQUIRE(itop >= 0 && itop < nlayers, "pb_transit_spectrum_ordered: itop out of range");
    PB_REQUIRE(ibottom <= nlayers, "pb_transit_spectrum_ordered: ibottom > nlayers");
    if (nwave == 0 || nwalkers == 0)
        return PB_OK;
    const int nrow = nlayers - itop;
    PB_REQUIRE(spectrum_d && ec_d && radius_d && raypath_d && column_d && work_d,
               "pb_transit_spectrum_ordered: null pointer");
    const int64_t npath = ((int64_t)nrow * (nrow - 1)) / 2;
    return pb_transit_fused_launch(nullptr, nullptr, spectrum_d, ec_d, raypath_d, radius_d, npath,
                                   rstar, itop, ibottom, maxdepth, nlayers, nwave, nwalkers, -1,
                                   0.0, pb::as_stream(stream), reinterpret_cast<double *>(work_d),
                                   column_d, nullptr, nullptr, nullptr);
}

int pb_transit_spectrum_limited(double *spectrum_d, const double *ec_d, const double *raypath_d,
                                const double *radius_d, const int32_t *column_d, double rstar,
                                int itop, int ibottom, double maxdepth, int nlayers, int nwave,
                                int nwalkers, void *work_d, const int32_t *tile_limit_d,
                                int32_t *flags_d, const int32_t *gate_d, void *stream)
{
    PB_REQUIRE(nlayers > 0 && nwave >= 0 && nwalkers >= 0, "pb_transit_spectrum_limited: bad shape");
    PB_REQUIRE(itop >= 0 && itop < nlayers, "pb_transit_spectrum_limited: itop out of range");
    PB_REQUIRE(ibottom <= nlayers, "pb_transit_spectrum_limited: ibottom > nlayers");
    if (nwave == 0 || nwalkers == 0)
        return PB_OK;
    const int nrow = nlayers - itop;
    PB_REQUIRE(spectrum_d && ec_d && radius_d && raypath_d && column_d && work_d,
               "pb_transit_spectrum_limited: null pointer");
    PB_REQUIRE(!tile_limit_d || flags_d,
               "pb_transit_spectrum_limited: a tile limit needs flags[nwalkers + 1] to report the "
               "walkers that ran past it");
    const int64_t npath = ((int64_t)nrow * (nrow - 1)) / 2;
    return pb_transit_fused_launch(nullptr, nullptr, spectrum_d, ec_d, raypath_d, radius_d, npath,
                                   rstar, itop, ibottom, maxdepth, nlayers, nwave, nwalkers, -1,
                                   0.0, pb::as_stream(stream), reinterpret_cast<double *>(work_d),
                                   column_d, tile_limit_d, flags_d, gate_d);
}

#ifdef PB_EXPERIMENTS
int pb_table_transit_supported(int nmol, int ntemp, int nlayers, int itop, int ibottom, int nwave)
{
    if (nmol < 1 || nmol > 8 || nlayers < 1 || itop < 0 || itop >= nlayers || ibottom > nlayers ||
        ntemp < 2)
        return 0;
    // one species' block of the table is addressed with 32-bit byte offsets
    if ((int64_t)ntemp * nlayers * nwave * 8 + 16 > 0xffffffffll)
        return 0;
    const int nimpact = std::min(ibottom, nlayers) - itop;
    const int mt = pb::div_up(std::max(nimpact, 1), 16);
    return nimpact > 1 && mt <= 8 && nwave >= 2;
}

int64_t pb_table_transit_work_doubles(int nmol, int nlayers, int itop, int ibottom, int nwalkers)
{
    if (!pb_table_transit_supported(nmol, 2, nlayers, itop, ibottom, 2))
        return 0;
    const int nimpact = std::min(ibottom, nlayers) - itop;
    const int ncoef = nmol <= 4 ? 4 : 8;
    const int64_t n = (int64_t)std::max(nwalkers, 0) * nlayers;
    // Q blocks | coef[n][2 ncoef] | tlo[n] (ints, rounded up to doubles) | walker order (ints) |
    // the flag that selects the two-walker kernel
    return (int64_t)qblocks(pb::div_up(nimpact, 16)) * 64 * std::max(nwalkers, 0) + n * 2 * ncoef +
           (n + 1) / 2 + (std::max(nwalkers, 0) + 1) / 2 + 2 + 8;
}

int pb_table_transit_batch(double *spectrum_d, const double *etable_d, const double *ttable_d,
                           const double *temps_d, const double *density_d,
                           const double *raypath_d, const double *radius_d, double rstar,
                           int itop, int ibottom, double maxdepth, int nmol, int ntemp,
                           int nlayers, int nwave, int nwalkers, void *work_d, void *stream)
{
    PB_REQUIRE(nmol >= 1 && nmol <= 8 && ntemp >= 2 && nlayers >= 1 && nwave >= 0 && nwalkers >= 0,
               "pb_table_transit_batch: bad shape (1-8 species, >= 2 table temperatures)");
    PB_REQUIRE(itop >= 0 && itop < nlayers, "pb_table_transit_batch: itop out of range");
    PB_REQUIRE(ibottom <= nlayers, "pb_table_transit_batch: ibottom > nlayers");
    if (nwave == 0 || nwalkers == 0)
        return PB_OK;
    PB_REQUIRE(pb_table_transit_supported(nmol, ntemp, nlayers, itop, ibottom, nwave),
               "pb_table_transit_batch: shape outside the one-pass form (ask "
               "pb_table_transit_supported; use pb_interp_ec_batch + pb_transit_spectrum_batch)");
    PB_REQUIRE(spectrum_d && etable_d && ttable_d && temps_d && density_d && raypath_d &&
                   radius_d && work_d,
               "pb_table_transit_batch: null pointer");
    PB_REQUIRE((int64_t)nwalkers * pb::div_up(nwave, 128) * 8 + 64 * (int64_t)nwalkers < (1ll << 31),
               "pb_table_transit_batch: too many workgroups");
    hipStream_t s = pb::as_stream(stream);
    const int nrow = nlayers - itop;
    const int nimpact = std::min(ibottom, nlayers) - itop;
    const int64_t npath = ((int64_t)nrow * (nrow - 1)) / 2;
    const int mt = pb::div_up(nimpact, 16);
    const int nblk = qblocks(mt);
    const int ncoef = nmol <= 4 ? 4 : 8;
    const int64_t n = (int64_t)nwalkers * nlayers;
    double *qwork = reinterpret_cast<double *>(work_d);
    double *coef = qwork + (int64_t)nblk * 64 * nwalkers;
    int32_t *tlo = reinterpret_cast<int32_t *>(coef + n * 2 * ncoef);
    k_interp_weights<<<pb::div_up(n, kBlock), kBlock, 0, s>>>(tlo, coef, ttable_d, temps_d,
                                                            density_d, nmol, ncoef, ntemp, n);
    PB_LAUNCH_CHECK();
    dim3 qgrid((unsigned)std::min(16, pb::div_up((int64_t)nblk * 64, kBlock)), nwalkers);
    k_path_qblocks<<<qgrid, kBlock, 0, s>>>(qwork, raypath_d, npath, nblk, nimpact);
    PB_LAUNCH_CHECK();
    // two walkers per wavefront (default from 2 walkers on, up to 1024 of them and 96 layers: the
    // pair's Q blocks must fit the LDS twice per CU); PB_TT_PAIR=0: one walker per wavefront, 2: two
    // whatever the pairs' brackets (the default lets k_walker_order decide)
    const char *pe = getenv("PB_TT_PAIR");
    const bool pair = !(pe && atoi(pe) == 0) && nwalkers >= 2 && nwalkers <= 1024 && mt <= 5 &&
                      ncoef == 4;
    const int32_t *skip_if = nullptr;
    if (pair) {
        int32_t *perm = tlo + ((n + 1) / 2) * 2;
        int32_t *use_pair = perm + ((nwalkers + 1) / 2) * 2;
        skip_if = use_pair;
        k_walker_order<<<1, 1024, 0, s>>>(perm, use_pair, tlo, temps_d, nlayers, nwalkers, itop,
                                          nimpact, pe && atoi(pe) == 2);
        PB_LAUNCH_CHECK();
        const size_t plds = 2 * ((size_t)nblk * 64 + (size_t)mt * 16 * (1 + 2 * ncoef)) * 8 +
                            2 * (size_t)mt * 16 * 4 + (size_t)mt * 4 * 4;
        const int npair = (nwalkers + 1) / 2;
#define PB_TP(M, W, T, P)                                                                          \
    do {                                                                                           \
        if (plds > 64 * 1024)                                                                      \
            PB_HIP(hipFuncSetAttribute(                                                            \
                reinterpret_cast<const void *>(k_table_transit_pair<M, W, T, 4, P>),               \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));                           \
        const int ncb = pb::div_up(nwave, (T / 64) * 16);                                          \
        const unsigned grid = (unsigned)(8 * (int64_t)pb::div_up(ncb, 8) * npair);                 \
        k_table_transit_pair<M, W, T, 4, P><<<grid, T, plds, s>>>(                                 \
            spectrum_d, etable_d, tlo, coef, qwork, radius_d, perm, use_pair, nblk, rstar, itop,   \
            ibottom, maxdepth, nmol, ntemp, nlayers, nwave, nwalkers, ncb);                        \
    } while (0)
        // Geometry measured at C5's shape (ms per 64 walkers, tools/bench_tt.py; the two passes over
        // a stored ec 2.65-2.69 on the same box): 256 threads (4 wavefronts x 16 columns x 2 walkers,
        // two workgroups per CU) with two K-steps of loads in flight 2.54-2.55 (168 registers), one
        // K-step 2.57-2.60; 512 threads 2.59 / 2.62; 384 threads 3.20 / 3.27.
        const int tpd = getenv("PB_TP_PD") ? atoi(getenv("PB_TP_PD")) : 2;
        switch (mt) {
        case 1: PB_TP(1, 2, 256, 2); break;
        case 2: PB_TP(2, 2, 256, 2); break;
        case 3: PB_TP(3, 2, 256, 2); break;
        case 4: PB_TP(4, 2, 256, 2); break;
        default:
            if (tpd == 1)
                PB_TP(5, 2, 256, 1);
            else
                PB_TP(5, 2, 256, 2);
            break;
        }
#undef PB_TP
        PB_LAUNCH_CHECK();
        // (the one-walker kernel follows: it returns at once unless k_walker_order found the
        // pairs to disagree too often)
    }
    const size_t lds = ((size_t)nblk * 64 + (size_t)mt * 16 * (1 + 2 * ncoef)) * 8 + (size_t)mt * 16 * 4;
#define PB_TT(M, W, T, S, P)                                                                       \
    do {                                                                                           \
        if (lds > 64 * 1024)                                                                       \
            PB_HIP(hipFuncSetAttribute(                                                            \
                reinterpret_cast<const void *>(k_table_transit_mfma<M, W, T, S, P>),               \
                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
        const int ncb = pb::div_up(nwave, (T / 64) * 32);                                          \
        const unsigned grid = (unsigned)(8 * (int64_t)pb::div_up(ncb, 8 * ginter) * ginter * nwalkers); \
        k_table_transit_mfma<M, W, T, S, P><<<grid, T, lds, s>>>(                                  \
            spectrum_d, etable_d, tlo, coef, qwork, radius_d, nblk, rstar, itop, ibottom, maxdepth, \
            nmol, ntemp, nlayers, nwave, nwalkers, ncb, ginter, skip_if);                          \
    } while (0)
    const int ginter = getenv("PB_TT_GINTER") ? std::max(1, std::min(8, atoi(getenv("PB_TT_GINTER")))) : 1;
    // Geometry measured at C5's shape (80 layers, 4 species; tools/bench_tt.py, ms per 64 walkers,
    // two passes over a stored ec 2.70, before the one-round-trip Q staging): 384 threads, one
    // K-step of loads in flight, 3 wavefronts per SIMD (148 registers, no spill) 3.10; 512 threads,
    // two K-steps in flight, 2 per SIMD (176 registers) 3.19; every 128-register form spills
    // (4.2-4.3).
    if (ncoef == 8) {
        switch (mt) {
        case 1: PB_TT(1, 2, 256, 8, 1); break;
        case 2: PB_TT(2, 2, 256, 8, 1); break;
        case 3: PB_TT(3, 2, 256, 8, 1); break;
        case 4: PB_TT(4, 2, 256, 8, 1); break;
        case 5: PB_TT(5, 2, 256, 8, 1); break;
        case 6: PB_TT(6, 2, 256, 8, 1); break;
        case 7: PB_TT(7, 2, 256, 8, 1); break;
        default: PB_TT(8, 2, 256, 8, 1); break;
        }
    } else {
        switch (mt) {
        case 1: PB_TT(1, 3, 384, 4, 1); break;
        case 2: PB_TT(2, 3, 384, 4, 1); break;
        case 3: PB_TT(3, 3, 384, 4, 1); break;
        case 4: PB_TT(4, 3, 384, 4, 1); break;
        case 5: PB_TT(5, 3, 384, 4, 1); break;
        case 6: PB_TT(6, 2, 256, 4, 1); break;
        case 7: PB_TT(7, 2, 256, 4, 1); break;
        default: PB_TT(8, 2, 256, 4, 1); break;
        }
    }
#undef PB_TT
    PB_LAUNCH_CHECK();
    return PB_OK;
}
#endif  // PB_EXPERIMENTS

int pb_band_integrate_batch(double *bandflux_d, const double *spectrum_d, const double *wn_d,
                            const int32_t *band_start_d, const int32_t *band_count_d,
                            const double *response_d, const int64_t *response_offset_d,
                            const double *heights_d, int nbands, int nwave, int nwalkers,
                            void *stream)
{
    PB_REQUIRE(nbands >= 0 && nwave >= 0 && nwalkers >= 0, "pb_band_integrate_batch: bad sizes");
    if (nbands == 0 || nwalkers == 0)
        return PB_OK;
    PB_REQUIRE(bandflux_d && spectrum_d && wn_d && band_start_d && band_count_d && response_d &&
                   response_offset_d,
               "pb_band_integrate_batch: null pointer");
    dim3 grid(nbands, nwalkers);
    k_band_integrate_batch<<<grid, kBlock, 0, pb::as_stream(stream)>>>(
        bandflux_d, spectrum_d, wn_d, band_start_d, band_count_d, response_d, response_offset_d,
        heights_d, nbands, nwave);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_reject_walkers(double *bandflux_d, const double *temps_d, double tmin, double tmax,
                      int nlayers, int nbands, int nwalkers, void *stream)
{
    PB_REQUIRE(nlayers >= 1 && nbands >= 0 && nwalkers >= 0, "pb_reject_walkers: bad shape");
    if (nwalkers == 0 || nbands == 0)
        return PB_OK;
    PB_REQUIRE(bandflux_d && temps_d, "pb_reject_walkers: null pointer");
    k_reject_walkers<<<nwalkers, kBlock, 0, pb::as_stream(stream)>>>(bandflux_d, temps_d, tmin,
                                                                   tmax, nlayers, nbands);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // extern "C"
