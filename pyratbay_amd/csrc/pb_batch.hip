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

#include "pb_common.h"

namespace {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------
// atmosphere.transit_path: path_r[i] = sqrt(rad_i^2 - rad_r^2) - sqrt(rad_{i+1}^2 - rad_r^2),
// rad = radius[itop:], packed lower triangle (row r has r entries from r(r-1)/2).  Same
// operations as the NumPy expression (square = one multiply, IEEE subtract and sqrt).
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
// interp_ec for a batch of walkers, assigning form.  Workgroup = (256 wavenumbers, layer,
// chunk of walkers).  Walkers of a chunk that share a temperature bracket share its two table
// slices: the brackets the chunk uses at this layer are walked in ascending order, the upper
// node of one bracket staying in registers as the lower node of the next; a walker is computed
// in the pass of its own bracket.  Per-(walker, layer) brackets and weights come from
// k_interp_weights (wave-uniform loads).  kS = species held in registers (<= 8).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_interp_weights(
    int32_t *tlo_out, double *wlo_out, double *whi_out, const double *ttable,
    const double *temps, int ntemp, int64_t n)
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
    tlo_out[i] = tlo;
    wlo_out[i] = (ttable[tlo + 1] - t) / span;
    whi_out[i] = (t - ttable[tlo]) / span;
}

template <int kS>
__global__ __launch_bounds__(kBlock) void k_interp_ec_batch(
    double *ec, const double *etable, const int32_t *tlo, const double *wlo, const double *whi,
    const double *density, int nmol, int ntemp, int nlayers, int nwave, int nwalkers, int chunk)
{
    const int k = blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int w0 = blockIdx.z * chunk, w1 = min(w0 + chunk, nwalkers);
    // brackets used by the chunk at this layer (wave-uniform)
    int bmin = ntemp, bmax = -1;
    for (int w = w0; w < w1; w++) {
        const int b = tlo[(int64_t)w * nlayers + k];
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
        hi[j] = j < nmol ? tab[((int64_t)j * ntemp + bmin) * slice] : 0.0;
    for (int b = bmin; b <= bmax; b++) {
#pragma unroll
        for (int j = 0; j < kS; j++) {
            lo[j] = hi[j];
            hi[j] = j < nmol ? tab[((int64_t)j * ntemp + b + 1) * slice] : 0.0;
        }
        for (int w = w0; w < w1; w++) {
            const int64_t wk = (int64_t)w * nlayers + k;
            if (tlo[wk] != b)
                continue;                                   // wave-uniform
            const double a = wlo[wk], c = whi[wk];
            const double *d = density + wk * nmol;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < kS; j++)
                if (j < nmol)
                    acc += lo[j] * (a * d[j]) + hi[j] * (c * d[j]);
            ec[wk * nwave + i] = acc;
        }
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
// block its rows are examined in order: exp(-tau)*r joins the trapezoid, the first tau above
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

template <int kRows>
__global__ __launch_bounds__(kBlock) void k_transit_fused(
    double *depth, int32_t *ideep, double *spectrum, const double *ec, const double *raypath,
    const double *radius, int64_t npath, double rstar, int itop, int ibottom, double maxdepth,
    int nlayers, int nwave, int deck_row, double rsurf)
{
    extern __shared__ __align__(16) double s_path[];      // [segment][kRows]
    const int w = blockIdx.y;
    const int col = blockIdx.x * kBlock + threadIdx.x;
    const bool active = col < nwave;
    const int64_t plane = (int64_t)nlayers * nwave;
    ec += (int64_t)w * plane;
    if (depth)
        depth += (int64_t)w * plane;
    const double *path = raypath ? raypath + (int64_t)w * npath : nullptr;
    const double *rad = radius ? radius + (int64_t)w * nlayers : nullptr;
    const int nimpact = min(ibottom, nlayers) - itop;     // rows 0..nimpact-1 are evaluated
    const double *src = ec + (int64_t)itop * nwave + (active ? col : 0);

    int stop = -1;
    double acc = 0.0, fprev = 0.0, rprev = 0.0;
    if (depth && active)
        for (int r = 0; r < itop; r++)
            depth[(int64_t)r * nwave + col] = 0.0;
    int rdone = 0;                       // rows examined so far (uniform)
    for (int rb = 0; rb < nimpact; rb += kRows) {
        // every column of the workgroup has met its exit: nothing left to compute
        if (__syncthreads_count(active && stop < 0) == 0)
            break;
        const int rlast = min(rb + kRows, nimpact) - 1;
        const int nseg = max(rlast, 0);
        for (int e = threadIdx.x; e < nseg * kRows; e += kBlock) {
            const int i = e / kRows, k = e % kRows;
            const int r = rb + k;
            s_path[e] = (r <= rlast && i < r) ? path[((int64_t)r * (r - 1)) / 2 + i] : 0.0;
        }
        __syncthreads();
        rdone = rlast + 1;
        if (!active)
            continue;
        if (stop >= 0) {
            // below the first crossing the reference leaves zeros
            if (depth)
                for (int r = rb; r <= rlast; r++)
                    depth[(int64_t)(itop + r) * nwave + col] = 0.0;
            continue;
        }
        double tau[kRows];
#pragma unroll
        for (int k = 0; k < kRows; k++)
            tau[k] = 0.0;
        if (nseg > 0) {
            double prev = src[0];
#pragma unroll 4
            for (int i = 0; i < nseg; i++) {
                const double next = src[(int64_t)(i + 1) * nwave];
                const double s = next + prev;
                prev = next;
                const double *pk = s_path + i * kRows;          // LDS broadcast reads
#pragma unroll
                for (int k = 0; k < kRows; k++)
                    tau[k] += pk[k] * s;
            }
        }
#pragma unroll
        for (int k = 0; k < kRows; k++) {
            const int r = rb + k;
            if (r > rlast)
                break;
            double t = tau[k];
            if (stop < 0) {
                if (spectrum) {
                    const double rr = rad[itop + r];
                    double f = exp(-t) * rr;
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

}  // namespace

// shared with pb_columns.hip: the single-spectrum entries use the fused kernel too
int pb_transit_fused_launch(double *depth_d, int32_t *ideep_d, double *spectrum_d,
                            const double *ec_d, const double *raypath_d, const double *radius_d,
                            int64_t npath, double rstar, int itop, int ibottom, double maxdepth,
                            int nlayers, int nwave, int nwalkers, int deck_row, double rsurf,
                            hipStream_t s)
{
    // rows per block: 16 when the grid fills the chip, 8 for narrow launches (shards)
    const int nrow = nlayers - itop;
    const bool narrow = (int64_t)nwave * nwalkers <= 32768;
    const int rows = narrow ? 8 : 16;
    const size_t lds = (size_t)std::max(nrow, 1) * rows * sizeof(double);
    if (lds > 64 * 1024) {
        pb::set_error("transit: %d layers need %zu B of LDS", nrow, lds);
        return PB_ERR_UNSUPPORTED;
    }
    dim3 grid(pb::div_up(nwave, kBlock), nwalkers);
    if (narrow)
        k_transit_fused<8><<<grid, kBlock, lds, s>>>(depth_d, ideep_d, spectrum_d, ec_d, raypath_d,
                                                   radius_d, npath, rstar, itop, ibottom, maxdepth,
                                                   nlayers, nwave, deck_row, rsurf);
    else
        k_transit_fused<16><<<grid, kBlock, lds, s>>>(depth_d, ideep_d, spectrum_d, ec_d,
                                                    raypath_d, radius_d, npath, rstar, itop,
                                                    ibottom, maxdepth, nlayers, nwave, deck_row,
                                                    rsurf);
    PB_LAUNCH_CHECK();
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

int pb_interp_ec_batch(double *ec_d, const double *etable_d, const double *ttable_d,
                       const double *temps_d, const double *density_d, void *work_d, int nmol,
                       int ntemp, int nlayers, int nwave, int nwalkers, void *stream)
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
    // workspace: wlo[n] | whi[n] doubles, then tlo[n] ints
    double *wlo = reinterpret_cast<double *>(work_d);
    double *whi = wlo + n;
    int32_t *tlo = reinterpret_cast<int32_t *>(whi + n);
    k_interp_weights<<<pb::div_up(n, kBlock), kBlock, 0, s>>>(tlo, wlo, whi, ttable_d, temps_d,
                                                            ntemp, n);
    PB_LAUNCH_CHECK();
    // walkers per chunk: enough to amortise the table reads, few enough to fill the chip
    int chunk = 16;
    while (chunk > 1 && (int64_t)pb::div_up(nwave, kBlock) * nlayers * pb::div_up(nwalkers, chunk) < 2048)
        chunk /= 2;
    dim3 grid(pb::div_up(nwave, kBlock), nlayers, pb::div_up(nwalkers, chunk));
    if (nmol <= 4)
        k_interp_ec_batch<4><<<grid, kBlock, 0, s>>>(ec_d, etable_d, tlo, wlo, whi, density_d, nmol,
                                                   ntemp, nlayers, nwave, nwalkers, chunk);
    else
        k_interp_ec_batch<8><<<grid, kBlock, 0, s>>>(ec_d, etable_d, tlo, wlo, whi, density_d, nmol,
                                                   ntemp, nlayers, nwave, nwalkers, chunk);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_transit_spectrum_batch(double *spectrum_d, double *depth_d, int32_t *ideep_d,
                              const double *ec_d, const double *raypath_d,
                              const double *radius_d, double rstar, int itop, int ibottom,
                              double maxdepth, int nlayers, int nwave, int nwalkers,
                              void *stream)
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
                                   0.0, pb::as_stream(stream));
}

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
