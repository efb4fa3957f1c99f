// Column (per-wavenumber) kernels: optical depth, transmission / emission integrals,
// Planck function, Simpson.  All are HBM-streaming: the [layer][wavenumber] arrays are
// row-major so that the 64 lanes of a wavefront read 512 contiguous bytes of one layer
// and walk down the layers; per-layer scalars (ray paths, radii, temperatures, mu) are
// wave-uniform and come through the scalar cache or LDS.
//
// Reference functions restated (pyratbay v2.0.1): src_c/_trapezoid.c:70-341,
// src_c/_blackbody.c:35-130, src_c/_simpson.c:167-203, src_c/cutils.c:27-42 and the
// Python loops pyratbay/opacity/optic_depth.py:103-112,
// pyratbay/spectrum/radiative_transfer.py:57-71, pyratbay/pyrat/spectrum.py:366-377.
#include <algorithm>
#include <cstdlib>

#include "pb_common.h"

int pb_transit_fused_launch(double *depth_d, int32_t *ideep_d, double *spectrum_d,
                            const double *ec_d, const double *raypath_d, const double *radius_d,
                            int64_t npath, double rstar, int itop, int ibottom, double maxdepth,
                            int nlayers, int nwave, int nwalkers, int deck_row, double rsurf,
                            hipStream_t s, double *work_d, const int32_t *scatter_d = nullptr,
                            const int32_t *tile_limit_d = nullptr, int32_t *flags_d = nullptr,
                            const int32_t *gate_d = nullptr);
int pb_path_blocks_launch(double **blocked_d, int64_t *len, const double *raypath_d, int64_t npath,
                          int rows, int nimpact, hipStream_t s);

namespace {

constexpr int kBlock = 256;
constexpr int kMaxMu = 16;

// ---------------------------------------------------------------------------
// cutils.ediff
// ---------------------------------------------------------------------------
__global__ void k_ediff(double *out, const double *arr, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < n)
        out[i] = arr[i + 1] - arr[i];
}

// ---------------------------------------------------------------------------
// _trapezoid.optdepth: one impact parameter (src_c/_trapezoid.c:238-276)
// ---------------------------------------------------------------------------
__global__ void k_optdepth(double *tau, const double *data, int64_t row_stride,
                           const double *h, int nint, double taumax, int32_t *ideep,
                           int ilay, int nwave)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nwave)
        return;
    double acc = 0.0;
    if (ideep[j] < 0) {
        double prev = nint > 0 ? data[j] : 0.0;
        for (int i = 0; i < nint; i++) {
            double next = data[(int64_t)(i + 1) * row_stride + j];
            acc += h[i] * (next + prev);
            prev = next;
        }
        if (acc > taumax)
            ideep[j] = ilay;
    }
    tau[j] = acc;
}

// ---------------------------------------------------------------------------
// Transit optical depth for all impact parameters (optic_depth.py:103-112), two kernels.
//
// k_transit_tau: thread = (column, block of kRowsPerThread impact parameters).  The
// column of ec is streamed once per row block (coalesced over columns, served by L2 after
// the first block), the kRowsPerThread running sums stay in registers and the ray-path
// segments are wave-uniform scalar loads.  tau_r = sum_{i<r} path_r[i]*(ec[i+1]+ec[i])
// is accumulated in the reference's order (same products, same additions), for EVERY row.
//
// k_transit_finish: thread = column.  Walks the rows once: finds the first crossing of
// maxdepth (the reference's early exit), zeroes the rows below it, writes ideep
// (including the final `ideep[ideep<0] = r`) and, when asked, integrates the
// transmission spectrum exp(-tau)*r over the rows down to ideep
// (radiative_transfer.py:57-71) in the same pass.
// ---------------------------------------------------------------------------
constexpr int kRowsPerThread = 16;     // 32 and 40 measured slower at W = 1e5
// Narrow grids (a wavenumber shard of a multi-GPU run: 12 500 columns) do not fill the chip
// with one thread per (column, 16 rows): they take 4 rows per thread and 64-thread workgroups.
constexpr int kRowsPerThreadNarrow = 4;
constexpr int kNarrowColumns = 32768;

// kScalar: `raypath` is the blocked layout of pb_path_blocks_launch ([block][segment][row], zero
// where segment >= row) and is read through the constant address space = scalar loads, the
// products taking the path from SGPRs; else the packed triangle, staged per block in LDS.
template <int kRows, bool kScalar>
__global__ __launch_bounds__(kBlock) void k_transit_tau(double *depth, const double *ec,
                                                        const double *raypath, int itop,
                                                        int ibottom, int nlayers, int nwave)
{
    // ray-path segments of this block's rows, [segment i][row k], zero where i >= r:
    // adding 0 * s leaves a sum unchanged, so one predicate-free loop serves all rows
    extern __shared__ __align__(16) double s_path[];
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int nrow = nlayers - itop;
    const int nimpact = min(ibottom, nlayers) - itop;     // rows 0..nimpact-1 are evaluated
    const int rb = blockIdx.y * kRows;
    const int rlast = min(rb + kRows, nimpact) - 1;       // last evaluated row here
    const int nseg = rb < nimpact ? max(rlast, 0) : 0;    // segments i < rlast; none below ibottom
    if (!kScalar) {
        for (int e = threadIdx.x; e < nseg * kRows; e += blockDim.x) {
            const int i = e / kRows, k = e % kRows;
            const int r = rb + k;
            s_path[e] = (r <= rlast && i < r) ? raypath[(r * (r - 1)) / 2 + i] : 0.0;
        }
        __syncthreads();
    }
    if (col >= nwave)
        return;
    double tau[kRows];
#pragma unroll
    for (int k = 0; k < kRows; k++)
        tau[k] = 0.0;
    if (nseg > 0) {
        const double *src = ec + (int64_t)itop * nwave + col;
        double prev = src[0];
        if (kScalar) {
            // blocks before this one are full: nseg_b = kRows*b + kRows - 1
            const int64_t b = blockIdx.y;
            const int64_t boff = (int64_t)kRows * (kRows * b * (b - 1) / 2 + b * (kRows - 1));
            typedef const double __attribute__((address_space(4))) *cpath_t;
            const cpath_t pb_ = (cpath_t)(unsigned long long)(raypath + boff);
            // rows of ec are fetched kAhead at a time, one group ahead of the sums that use them
            constexpr int kAhead = 4;
            double nx[kAhead];
#pragma unroll
            for (int j = 0; j < kAhead; j++)
                nx[j] = src[(int64_t)min(j + 1, nseg) * nwave];
            for (int i0 = 0; i0 < nseg; i0 += kAhead) {
                double cur[kAhead];
#pragma unroll
                for (int j = 0; j < kAhead; j++)
                    cur[j] = nx[j];
#pragma unroll
                for (int j = 0; j < kAhead; j++)
                    nx[j] = src[(int64_t)min(i0 + kAhead + j + 1, nseg) * nwave];
#pragma unroll
                for (int j = 0; j < kAhead; j++) {
                    const int i = i0 + j;
                    if (i < nseg) {                             // uniform
                        const double s = cur[j] + prev;
                        prev = cur[j];
                        double pv[kRows];
#pragma unroll
                        for (int k = 0; k < kRows; k++)
                            pv[k] = pb_[i * kRows + k];
#pragma unroll
                        for (int k = 0; k < kRows; k++)
                            tau[k] += pv[k] * s;
                    }
                }
            }
        } else {
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
    }
#pragma unroll
    for (int k = 0; k < kRows; k++) {
        const int r = rb + k;
        if (r < nrow)
            depth[(int64_t)(itop + r) * nwave + col] = tau[k];     // 0 for r >= nimpact
    }
    if (blockIdx.y == 0)
        for (int r = 0; r < itop; r++)
            depth[(int64_t)r * nwave + col] = 0.0;
}

// Opaque cloud deck (radiative_transfer.py:62-66): at row deck_row = deck_itop - itop the
// interval's far end is the cloud top: h = rsurf - radius[deck_itop-1] and the integrand is
// the linear interpolation (scipy interp1d) of exp(-tau)*r between the two layers at rsurf.
__device__ inline double deck_integrand(double f_above, double f_below, double r_above,
                                        double r_below, double rsurf)
{
    const double slope = (f_above - f_below) / (r_above - r_below);
    return slope * (rsurf - r_below) + f_below;
}

__global__ __launch_bounds__(kBlock) void k_transit_finish(
    double *depth, int32_t *ideep, double *spectrum, const double *radius_g, double rstar,
    int itop, int ibottom, double maxdepth, int nlayers, int nwave, int deck_row, double rsurf)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= nwave)
        return;
    // the radii are wave-uniform: scalar loads through the constant address space (as
    // same-address vector loads they sit in the dependent chain of every row)
    typedef const double __attribute__((address_space(4))) *crad_t;
    const crad_t radius = (crad_t)(unsigned long long)radius_g;
    const int nimpact = min(ibottom, nlayers) - itop;
    int stop = -1;
    double acc = 0.0, fprev = 0.0, rprev = 0.0;
    // rows are fetched eight at a time (independent loads in flight), then examined in order
    constexpr int kFetch = 8;
    double nx[kFetch];
#pragma unroll
    for (int k = 0; k < kFetch; k++)
        nx[k] = k < nimpact ? depth[(int64_t)(itop + k) * nwave + col] : 0.0;
    for (int r0 = 0; r0 < nimpact; r0 += kFetch) {
        double t[kFetch];
#pragma unroll
        for (int k = 0; k < kFetch; k++)
            t[k] = nx[k];
        // the next group is requested before this one is examined
#pragma unroll
        for (int k = 0; k < kFetch; k++)
            nx[k] = (r0 + kFetch + k < nimpact)
                        ? depth[(int64_t)(itop + r0 + kFetch + k) * nwave + col]
                        : 0.0;
#pragma unroll
        for (int k = 0; k < kFetch; k++) {
            const int r = r0 + k;
            if (r >= nimpact)
                break;
            if (stop < 0) {
                if (spectrum) {
                    const double rad = radius[itop + r];
                    double f = pb::exp_s(-t[k]) * rad;
                    if (r > 0 && r == deck_row) {
                        f = deck_integrand(fprev, f, rprev, rad, rsurf);
                        acc += (rsurf - rprev) * (fprev + f);
                    } else if (r > 0) {
                        acc += (rad - rprev) * (fprev + f);
                    }
                    fprev = f;
                    rprev = rad;
                }
                if (t[k] > maxdepth)
                    stop = r;
            } else if (t[k] != 0.0) {
                depth[(int64_t)(itop + r) * nwave + col] = 0.0;   // below the first crossing
            }
        }
    }
    // ideep[ideep<0] = r with r the last loop value (itop if the loop is empty)
    const int last = nimpact > 0 ? itop + nimpact - 1 : itop;
    ideep[col] = stop >= 0 ? itop + stop : last;
    if (spectrum) {
        const double rtop = radius[itop];
        spectrum[col] = (rtop * rtop + 2 * (acc * 0.5)) / (rstar * rstar);
    }
}

// ---------------------------------------------------------------------------
// What the reference makes of a plane-parallel flux AFTER the radiative transfer, per sample
// (pyrat/spectrum.py:394-405 and eval()'s unit conversion, pyrat_obj.py:323-329):
//   fplanet  = flux [* f_dilution]
//   emission : spectrum = fplanet
//   eclipse  : spectrum = fplanet * (1/starflux * rprs2),     rprs2 = (rplanet/rstar)^2
//   f_lambda : spectrum = 10.0 * fplanet * (rd * wn * 1e-4)^2, rd = rplanet/distance
// Same products in the same order as the NumPy expressions (bit-equal); `scale` is rprs2 or rd.
// spectrum and fplanet may alias flux (the reference's `spec.fplanet = spec.spectrum`).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_emission_observables(
    double *spectrum, double *fplanet, const double *flux, const double *starflux,
    const double *wn, int64_t n, int mode, int dilute, double f_dilution, double scale)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    double f = flux[i];
    if (dilute)
        f *= f_dilution;
    double s = f;
    if (mode == 1) {
        const double fstar_rprs = 1.0 / starflux[i] * scale;
        s = f * fstar_rprs;
    } else if (mode == 2) {
        const double t = scale * wn[i] * 1.0e-4;
        s = 10.0 * f * (t * t);
    }
    if (fplanet)
        fplanet[i] = f;
    spectrum[i] = s;
}

// band fluxes of a batch times a per-walker factor (f_dilution, pyrat_obj.py:296-297: applied to
// the band integral here, to the spectrum in the reference -- the same up to one rounding) and a
// per-band factor (eclipse: rprs^2 / bandflux_star, pyrat_obj.py:662-665), either may be absent
__global__ __launch_bounds__(kBlock) void k_band_scale(double *bandflux, const double *band_scale,
                                                       const double *walker_scale, int nbands,
                                                       int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    double v = bandflux[i];
    if (walker_scale)
        v *= walker_scale[i / nbands];
    if (band_scale)
        v *= band_scale[i % nbands];
    bandflux[i] = v;
}

// ---------------------------------------------------------------------------
// _trapezoid.plane_parallel_optical_depth (src_c/_trapezoid.c:175-213)
// ---------------------------------------------------------------------------
__global__ void k_plane_depth(double *depth, int32_t *ideep, const double *ec,
                              const double *h, double maxdepth, int itop, int ibottom,
                              int nlayers, int nwave)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwave)
        return;
    double acc = 0.0;
    for (int k = 0; k <= itop && k < nlayers; k++)
        depth[(int64_t)k * nwave + i] = 0.0;
    double prev = itop < nlayers ? ec[(int64_t)itop * nwave + i] : 0.0;
    // The rows are fetched eight at a time and then examined in order: with the stop rule
    // inside a one-row loop every load waited for the previous row's branch (64 us for 128 MB
    // at 1e5 samples: 80 exposed latencies per column).  Rows past the stop are loaded (inside
    // the array) and dropped; the sums and the stop are the reference's.
    constexpr int kFetch = 8;
    int k = min(itop, nlayers - 1), stop = -1;
    for (int k0 = itop + 1; k0 < nlayers && stop < 0; k0 += kFetch) {
        double cur[kFetch];
#pragma unroll
        for (int q = 0; q < kFetch; q++)
            cur[q] = ec[(int64_t)min(k0 + q, nlayers - 1) * nwave + i];
#pragma unroll
        for (int q = 0; q < kFetch; q++) {
            if (stop >= 0 || k0 + q >= nlayers)
                continue;
            k = k0 + q;
            acc += 0.5 * h[k - 1] * (cur[q] + prev);
            prev = cur[q];
            depth[(int64_t)k * nwave + i] = acc;
            if (acc >= maxdepth || k == ibottom || k == nlayers - 1)
                stop = k;
        }
    }
    // (the one-row loop ended with k == nlayers when no row was examined after itop)
    ideep[i] = stop >= 0 ? stop : (itop + 1 >= nlayers ? nlayers : k);
}

// ---------------------------------------------------------------------------
// _trapezoid.trapezoid2D (src_c/_trapezoid.c:70-90)
// ---------------------------------------------------------------------------
__global__ void k_trapezoid2d(double *out, const double *data, const double *h,
                              const int32_t *nint, int nrows, int nwave)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nwave)
        return;
    int n = nint[j];
    if (n > nrows - 1)
        n = nrows - 1;      // the reference would read past the array
    double acc = 0.0;
    if (n > 0) {
        double prev = data[j];
        for (int i = 0; i < n; i++) {
            double next = data[(int64_t)(i + 1) * nwave + j];
            acc += h[i] * (prev + next);
            prev = next;
        }
    }
    out[j] = acc * 0.5;
}

// ---------------------------------------------------------------------------
// Fused transmission spectrum (radiative_transfer.py:57-71, no cloud deck)
// ---------------------------------------------------------------------------
__global__ void k_transmission(double *spectrum, const double *depth,
                               const int32_t *ideep, const double *radius, int itop,
                               double rstar, int nlayers, int nwave, int deck_row, double rsurf)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nwave)
        return;
    int n = ideep[j] - itop;                 // number of intervals
    if (n > nlayers - 1 - itop)
        n = nlayers - 1 - itop;
    double acc = 0.0;
    if (n > 0) {
        double rprev = radius[itop];
        double prev = pb::exp_s(-depth[(int64_t)itop * nwave + j]) * rprev;
        for (int i = 0; i < n; i++) {
            double rnext = radius[itop + i + 1];
            double next = pb::exp_s(-depth[(int64_t)(itop + i + 1) * nwave + j]) * rnext;
            if (i + 1 == deck_row) {
                next = deck_integrand(prev, next, rprev, rnext, rsurf);
                acc += (rsurf - rprev) * (prev + next);
            } else {
                acc += (rnext - rprev) * (prev + next);
            }
            prev = next;
            rprev = rnext;
        }
    }
    acc *= 0.5;
    double rtop = radius[itop];
    spectrum[j] = (rtop * rtop + 2 * acc) / (rstar * rstar);
}

// ---------------------------------------------------------------------------
// Planck function (src_c/_blackbody.c:35-130)
// ---------------------------------------------------------------------------
__device__ inline double planck_factor(double wn)
{
    return 2 * pb::kH * pb::kLS * pb::kLS * pow(wn, 3.0);
}
// kt = kKB * temp and its rounded reciprocal: the same for every sample of a layer, so the
// column kernels prepare them once per (workgroup, layer) in LDS (planck_terms) and the exponent's
// division is pb::quot's three instructions; every kernel forms B this way (same bits)
__device__ inline double planck_q(double factor, double wn, double kt, double inv_kt)
{
    return factor / (pb::exp_s(pb::quot_fast(pb::kH * pb::kLS * wn, kt, inv_kt)) - 1.0);
}
__device__ inline double planck(double factor, double wn, double temp)
{
    double kt, inv_kt;
    pb::sane_divisor(pb::kKB * temp, kt, inv_kt);      // (T = 0 -> B = 0 like the reference)
    return planck_q(factor, wn, kt, inv_kt);
}
// s_kt[0 .. 2 nlayers): kKB * temp[k] and 1 / (kKB * temp[k]) of one temperature profile, then
// 1 / mu[m] for the nmu quadrature angles (the angles themselves stay scalar loads)
__device__ inline void planck_terms(double *s_kt, const double *temp, int nlayers,
                                    const double *mu = nullptr, int nmu = 0)
{
    // (sane_divisor: a layer at T = 0 or a ray at mu = 0 keeps pb::quot_fast finite; the results
    // are those of the true divisions)
    for (int k = threadIdx.x; k < nlayers; k += blockDim.x)
        pb::sane_divisor(pb::kKB * temp[k], s_kt[k], s_kt[nlayers + k]);
    for (int m = threadIdx.x; m < nmu; m += blockDim.x) {
        double ms;
        pb::sane_divisor(mu[m], ms, s_kt[2 * nlayers + m]);
    }
    __syncthreads();
}

__global__ void k_blackbody2d(double *B, const double *wn, int nwave, const double *temp,
                              int nlayers, const int32_t *last)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y;
    if (i >= nwave)
        return;
    int ilast = last ? last[i] : nlayers - 1;
    if (j > ilast)
        return;
    double w = wn[i];
    B[(int64_t)j * nwave + i] = planck(planck_factor(w), w, temp[j]);
}

__global__ void k_blackbody1d(double *B, const double *wn, int nwave, double temp)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwave)
        return;
    double w = wn[i];
    B[i] = planck(planck_factor(w), w, temp);
}

// ---------------------------------------------------------------------------
// _trapezoid.intensity (src_c/_trapezoid.c:304-341; tdiff/itrapezoid utils.h:6-41)
// one thread per (column, mu)
// ---------------------------------------------------------------------------
__global__ void k_intensity(double *out, const double *tau, const int32_t *ideep,
                            const double *bbody, const double *mu, int nmu, int rtop,
                            int nlayers, int nwave)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (j >= nwave)
        return;
    int last = ideep[j];
    if (last > nlayers - 1)
        last = nlayers - 1;
    double m, im;
    pb::sane_divisor(mu[k], m, im);
    double blast = bbody[(int64_t)last * nwave + j];
    double result;
    if (last - rtop == 1) {
        result = blast;
    } else {
        double acc = 0.0;
        if (last > rtop) {
            double eprev = pb::exp_s(pb::quot_fast(-pb::clamp_depth(tau[(int64_t)rtop * nwave + j]), m, im));
            double bprev = bbody[(int64_t)rtop * nwave + j];
            for (int i = rtop; i < last; i++) {
                double enext = pb::exp_s(pb::quot_fast(-pb::clamp_depth(tau[(int64_t)(i + 1) * nwave + j]), m, im));
                double bnext = bbody[(int64_t)(i + 1) * nwave + j];
                acc += (enext - eprev) * (bnext + bprev);
                eprev = enext;
                bprev = bnext;
            }
        }
        result = blast * pb::exp_s(pb::quot_fast(-pb::clamp_depth(tau[(int64_t)last * nwave + j]), m, im)) - 0.5 * acc;
    }
    out[(int64_t)k * nwave + j] = result;
}

// ---------------------------------------------------------------------------
// Fused emission: Planck in registers + intensity for every mu + quadrature sum
// (pyrat/spectrum.py:366-377).  One thread per column, layers outermost so that
// tau is read once; the nmu running sums live in registers.
// ---------------------------------------------------------------------------
template <int MU>
__global__ void k_emission_flux(double *flux, double *intensity, const double *tau,
                                const int32_t *ideep, const double *wn,
                                const double *temp, const double *mu,
                                const double *weights, int nmu, int rtop, int nlayers,
                                int nwave, int ideep_max)
{
    extern __shared__ double s_kt[];        // [2][nlayers]: kKB T and its reciprocal
    planck_terms(s_kt, temp, nlayers, mu, nmu);
    const double *s_imu = s_kt + 2 * nlayers;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nwave)
        return;
    int last = min(ideep[j], ideep_max);    // np.clip(ideep, 0, cloud_itop) with a cloud deck
    if (last > nlayers - 1)
        last = nlayers - 1;
    const double w = wn[j];
    const double factor = planck_factor(w);
    double acc[MU], eprev[MU];
    double t0 = pb::clamp_depth(tau[(int64_t)rtop * nwave + j]);
#pragma unroll
    for (int k = 0; k < MU; k++) {
        acc[k] = 0.0;
        eprev[k] = k < nmu ? pb::exp_s(pb::quot_fast(-t0, mu[k], s_imu[k])) : 0.0;
    }
    double bprev = planck_q(factor, w, s_kt[rtop], s_kt[nlayers + rtop]);
    double tlast = t0;
    for (int i = rtop; i < last; i++) {
        double t = pb::clamp_depth(tau[(int64_t)(i + 1) * nwave + j]);
        double bnext = planck_q(factor, w, s_kt[i + 1], s_kt[nlayers + i + 1]);
        double bsum = bnext + bprev;
#pragma unroll
        for (int k = 0; k < MU; k++) {
            if (k < nmu) {
                double enext = pb::exp_s(pb::quot_fast(-t, mu[k], s_imu[k]));
                acc[k] += (enext - eprev[k]) * bsum;
                eprev[k] = enext;
            }
        }
        bprev = bnext;
        tlast = t;
    }
    double blast = (last > rtop) ? bprev : planck_q(factor, w, s_kt[last], s_kt[nlayers + last]);
    if (last <= rtop)
        tlast = pb::clamp_depth(tau[(int64_t)last * nwave + j]);
    double total = 0.0;
#pragma unroll
    for (int k = 0; k < MU; k++) {
        if (k < nmu) {
            double val;
            if (last - rtop == 1)
                val = blast;
            else
                val = blast * pb::exp_s(pb::quot_fast(-tlast, mu[k], s_imu[k])) - 0.5 * acc[k];
            if (intensity)
                intensity[(int64_t)k * nwave + j] = val;
            total += val * weights[k];
        }
    }
    flux[j] = total;
}

// ---------------------------------------------------------------------------
// Plane-parallel optical depth + emission flux in ONE pass, for a batch of walkers (the
// retrieval inner loop in emission geometry): the running optical depth of k_plane_depth
// (_trapezoid.c:175-213) feeds the intensity sums of k_emission_flux (_trapezoid.c:304-341 +
// pyrat/spectrum.py:366-377) layer by layer, so depth is neither written nor read.  Same
// operations in the same order as the two kernels run one after the other.
// grid (columns, walkers); ec[nw][L][W], intervals[nw][L-1], temp[nw][L] -> flux[nw][W].
// ---------------------------------------------------------------------------
template <int MU>
__global__ void k_emission_fused(double *flux, const double *ec, const double *intervals,
                                 const double *wn, const double *temp, const double *mu,
                                 const double *weights, int nmu, double maxdepth, int itop,
                                 int ibottom, int nlayers, int nwave, const int32_t *scatter,
                                 pb::TileLimit lim, int32_t *flags)
{
    extern __shared__ double s_kt[];        // [2][nlayers]: kKB T of this walker and its reciprocal
    const int wk = blockIdx.y;
    if (lim.gate && pb::uniform_i32(lim.gate + wk) == 0)
        return;                             // (repair pass: walker wk was not flagged)
    planck_terms(s_kt, temp + (int64_t)wk * nlayers, nlayers, mu, nmu);
    const double *s_imu = s_kt + 2 * nlayers;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nwave)
        return;
    ec += (int64_t)wk * nlayers * nwave;
    const double *h = intervals + (int64_t)wk * (nlayers - 1);
    const double w = wn[j];
    const double factor = planck_factor(w);
    const int rtop = itop;
    double acc[MU], eprev[MU];
#pragma unroll
    for (int k = 0; k < MU; k++) {
        acc[k] = 0.0;
        eprev[k] = k < nmu ? pb::exp_s(pb::quot_fast(-0.0, mu[k], s_imu[k])) : 0.0;   // depth[rtop] = 0
    }
    double bprev = planck_q(factor, w, s_kt[rtop], s_kt[nlayers + rtop]);
    double depth = 0.0, tlast = 0.0;
    double prev = ec[(int64_t)itop * nwave + j];
    int last = rtop;                                          // deepest layer seen so far
    // the last layer that was interpolated for this block of 256 (ordered) columns (TileLimit:
    // workgroups are 256 columns wide, so the limit is uniform); a column still open beyond it
    // goes to the repair pass
    const int klim = lim.tile ? lim.row0 + 16 * (pb::uniform_i32(lim.tile + blockIdx.x) + 1) - 1
                              : nlayers;
    bool overrun = false;
    for (int k = itop + 1; k < nlayers; k++) {
        if (k > klim) {
            overrun = true;
            break;
        }
        const double cur = ec[(int64_t)k * nwave + j];
        depth += 0.5 * h[k - 1] * (cur + prev);
        prev = cur;
        // intensity terms of the interval (k-1, k)
        const double bnext = planck_q(factor, w, s_kt[k], s_kt[nlayers + k]);
        const double bsum = bnext + bprev;
        const double dq = pb::clamp_depth(depth);             // (exp(-inf / mu) = 0 without a NaN)
#pragma unroll
        for (int m = 0; m < MU; m++) {
            if (m < nmu) {
                const double enext = pb::exp_s(pb::quot_fast(-dq, mu[m], s_imu[m]));
                acc[m] += (enext - eprev[m]) * bsum;
                eprev[m] = enext;
            }
        }
        bprev = bnext;
        tlast = depth;
        last = k;
        if (depth >= maxdepth || k == ibottom || k == nlayers - 1)
            break;
    }
    if (overrun) {
        if (flags) {
            flags[wk] = 1;
            flags[gridDim.y] = 1;                             // flags[nwalkers]: any walker
        }
        return;                                               // (the repair pass writes this column)
    }
    // (itop == nlayers-1: no interval; the reference's loop then leaves ideep = nlayers, clipped)
    const double blast =
        (last > rtop) ? bprev : planck_q(factor, w, s_kt[last], s_kt[nlayers + last]);
    double total = 0.0;
#pragma unroll
    for (int m = 0; m < MU; m++) {
        if (m < nmu) {
            double val;
            if (last - rtop == 1)
                val = blast;
            else
                val = blast * pb::exp_s(pb::quot_fast(-pb::clamp_depth(tlast), mu[m], s_imu[m])) - 0.5 * acc[m];
            total += val * weights[m];
        }
    }
    // (ordered batches: column j of ec / wn is column scatter[j] of the grid; an index outside
    // the grid is dropped, not written out of bounds)
    const int dst = scatter ? scatter[j] : j;
    if (dst >= 0 && dst < nwave)
        flux[(int64_t)wk * nwave + dst] = total;
}

// ---------------------------------------------------------------------------
// _simpson.simps2D (src_c/_simpson.c:167-203, include/simpson.h:29-47)
// ---------------------------------------------------------------------------
__global__ void k_simps2d(double *out, const double *y, int ny, int nwave,
                          const double *h, const int32_t *nint, const double *hsum,
                          const double *hratio, const double *hfactor)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwave)
        return;
    int n = nint[i];
    if (n > ny)
        n = ny;
    double res;
    if (n < 2) {
        res = 0.0;
    } else if (n == 2) {
        res = h[0] * 0.5 * (y[i] + y[(int64_t)nwave + i]);
    } else {
        double acc = 0.0;
        for (int p = 0; p < (n - 1) / 2; p++) {
            int j = 2 * p;
            acc += (y[(int64_t)j * nwave + i] * (2.0 - 1.0 / hratio[p]) +
                    y[(int64_t)(j + 1) * nwave + i] * hfactor[p] +
                    y[(int64_t)(j + 2) * nwave + i] * (2.0 - hratio[p])) * hsum[p];
        }
        res = acc / 6.0;
        if (n % 2 == 0)
            res += h[n - 2] * 0.5 *
                   (y[(int64_t)(n - 2) * nwave + i] + y[(int64_t)(n - 1) * nwave + i]);
    }
    out[i] = res;
}

// ---------------------------------------------------------------------------
// Band integration (PassBand.integrate, pyratbay/spectrum/spec_tools.py:193-233):
// trapezoid over wavenumber of spectrum[idx]*response for each band's contiguous
// index range.  One workgroup per band; only the pairs (i, i+1) whose left sample lies
// in [wbegin, wbegin+wcount) are summed, so that wavenumber shards add up.  The
// per-thread partial sums are combined in a fixed order (reproducible).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_band_integrate(
    double *bandflux, const double *spectrum, const double *wn, const int32_t *band_start,
    const int32_t *band_count, const double *response, const int64_t *response_offset,
    int64_t wbegin, int64_t wcount)
{
    __shared__ double s_part[kBlock];
    const int b = blockIdx.x;
    const int start = band_start[b];
    const int count = band_count[b];
    const double *resp = response + response_offset[b];
    double acc = 0.0;
    for (int i = threadIdx.x; i + 1 < count; i += kBlock) {
        const int64_t g = (int64_t)start + i;
        if (g < wbegin || g >= wbegin + wcount)
            continue;
        const double y0 = spectrum[g] * resp[i];
        const double y1 = spectrum[g + 1] * resp[i + 1];
        acc += 0.5 * (wn[g + 1] - wn[g]) * (y0 + y1);
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = kBlock / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            s_part[threadIdx.x] += s_part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        bandflux[b] = s_part[0];
}

// ---------------------------------------------------------------------------
// Two-stream fluxes (pyratbay/pyrat/spectrum.py:454-522; Heng et al. 2014 Eqs. B5-B6).
// exp1 = scipy.special.exp1 for real arguments (xsf/expint.h:22-52, the E1XB routine of
// Zhang & Jin 1996): power series for x <= 1, backward continued fraction otherwise.
// ---------------------------------------------------------------------------
__device__ inline double exp1_real(double x)
{
    const double ga = 0.5772156649015328606065120900824024;
    if (x == 0.0)
        return INFINITY;
    if (x <= 1.0) {
        double e1 = 1.0, r = 1.0;
        for (int k = 1; k < 26; k++) {
            const double k1 = k + 1.0;
            r = -r * k * x / (k1 * k1);
            e1 += r;
            if (fabs(r) <= fabs(e1) * 1e-15)
                break;
        }
        return -ga - log(x) + x * e1;
    }
    const int m = 20 + (int)(80.0 / x);
    double t0 = 0.0;
    for (int k = m; k > 0; k--)
        t0 = k / (1.0 + k / (x + t0));
    return pb::exp_s(-x) * (1.0 / (x + t0));
}

// The diffusivity transmission of every (layer interval, sample), trans[i][j] written to
// flux_up[i][j]: the expensive part of the two-stream solver (exp1: a series or a continued
// fraction of 20 + 80/x terms, one division each) has no dependence between layers, so it gets a
// thread per (interval, sample) instead of sitting in the column sweeps (one thread per column =
// 1.5 wavefronts per SIMD at 1e5 samples).  Same expression as the sweep evaluated before.
__global__ __launch_bounds__(kBlock) void k_two_stream_trans(double *flux_up, const double *depth,
                                                            int nlayers, int nwave)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= nwave || i >= nlayers - 1)
        return;
    const double dtau0 = depth[(int64_t)(i + 1) * nwave + j] - depth[(int64_t)i * nwave + j];
    flux_up[(int64_t)i * nwave + j] =
        (1 - dtau0) * pb::exp_s(-dtau0) + dtau0 * dtau0 * exp1_real(dtau0);
}

// One column per thread.  trans[i] comes from k_two_stream_trans (parked in flux_up[i], which
// the upward sweep then overwrites); Planck values are recomputed in registers.
__global__ __launch_bounds__(kBlock) void k_two_stream(
    double *flux_down, double *flux_up, const double *depth, const double *wn,
    const double *temp, const double *f_int, const double *flux_top, int rtop, int nlayers,
    int nwave)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= nwave)
        return;
    const double pi = 3.141592653589793;
    const double w = wn[j];
    const double factor = planck_factor(w);
    // the irradiation is written into row rtop and the sweep overwrites rows 1..L-1
    // (spectrum.py:498-509): it survives only when rtop == 0
    double down = (flux_top && rtop == 0) ? flux_top[j] : 0.0;
    flux_down[j] = down;
    double dprev = depth[j];
    double bprev = planck(factor, w, temp[0]);
    for (int i = 0; i < nlayers - 1; i++) {
        const double dnext = depth[(int64_t)(i + 1) * nwave + j];
        const double bnext = planck(factor, w, temp[i + 1]);
        const double dtau0 = dnext - dprev;
        const double trans = flux_up[(int64_t)i * nwave + j];
        const double bp = (bnext - bprev) / dtau0;
        down = trans * down + pi * bprev * (1 - trans) +
               pi * bp * (-2.0 / 3 * (1 - pb::exp_s(-dtau0)) + dtau0 * (1 - trans / 3));
        flux_down[(int64_t)(i + 1) * nwave + j] = down;
        dprev = dnext;
        bprev = bnext;
    }
    double up = down + (f_int ? f_int[j] : 0.0);
    flux_up[(int64_t)(nlayers - 1) * nwave + j] = up;
    // bprev = B[L-1], dprev = depth[L-1]
    for (int i = nlayers - 2; i >= 0; i--) {
        const double dlo = depth[(int64_t)i * nwave + j];
        const double blo = planck(factor, w, temp[i]);
        const double dtau0 = dprev - dlo;
        const double trans = flux_up[(int64_t)i * nwave + j];
        const double bp = (bprev - blo) / dtau0;
        up = trans * up + pi * bprev * (1 - trans) +
             pi * bp * (2.0 / 3 * (1 - pb::exp_s(-dtau0)) - dtau0 * (1 - trans / 3));
        flux_up[(int64_t)i * nwave + j] = up;
        dprev = dlo;
        bprev = blo;
    }
}

// f_int (spectrum.py:475-478): Planck at tint, scaled so that its trapezoid integral over
// wn is sigma*tint^4.  One workgroup; fixed-order tree sum.
__global__ __launch_bounds__(kBlock) void k_internal_flux(double *f_int, const double *wn,
                                                         double tint, int nwave)
{
    __shared__ double s_part[kBlock];
    const double sigma = 5.6703744191844314e-05;      // constants/astrophysical_constants.py:71
    double acc = 0.0;
    for (int i = threadIdx.x; i < nwave; i += kBlock) {
        const double w = wn[i];
        const double b = planck(planck_factor(w), w, tint);
        f_int[i] = b;
        if (i + 1 < nwave) {
            const double w1 = wn[i + 1];
            acc += (w1 - w) * (planck(planck_factor(w1), w1, tint) + b) / 2.0;
        }
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    for (int k = kBlock / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k)
            s_part[threadIdx.x] += s_part[threadIdx.x + k];
        __syncthreads();
    }
    const double total = s_part[0];
    if (total > 0) {
        const double scale = sigma * pow(tint, 4.0) / total;
        for (int i = threadIdx.x; i < nwave; i += kBlock)
            f_int[i] *= scale;
    }
}

// ---------------------------------------------------------------------------
// Gaussian log-likelihood of band fluxes (pyratbay/tools/retrieval_tools.py:98-104): one
// walker per wavefront, -0.5*sum(((data-model)/uncert)^2) - 0.5*sum(log(2*pi*uncert^2)),
// -inf when not finite.  Sums in lane-strided order, then a fixed butterfly.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_loglike(double *loglike, const double *model,
                                                const double *data, const double *uncert,
                                                int nbands)
{
    const int w = blockIdx.x;
    const double *m = model + (int64_t)w * nbands;
    double chi = 0.0, norm = 0.0;
    for (int b = threadIdx.x; b < nbands; b += 64) {
        const double r = (data[b] - m[b]) / uncert[b];
        chi += r * r;
        norm += log(2.0 * 3.141592653589793 * (uncert[b] * uncert[b]));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        chi += __shfl_xor(chi, d);
        norm += __shfl_xor(norm, d);
    }
    if (threadIdx.x == 0) {
        const double ll = -0.5 * chi - 0.5 * norm;
        loglike[w] = isfinite(ll) ? ll : -1.0e98;     // retrieval_tools.py:101-103
    }
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int pb_ediff(double *out_d, const double *arr_d, int n, void *stream)
{
    PB_REQUIRE(n >= 0, "pb_ediff: n < 0");
    if (n < 2)
        return PB_OK;
    PB_REQUIRE(out_d && arr_d, "pb_ediff: null pointer");
    k_ediff<<<pb::div_up(n - 1, kBlock), kBlock, 0, pb::as_stream(stream)>>>(out_d, arr_d, n);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_optdepth(double *tau_d, const double *data_d, int64_t row_stride,
                const double *intervals_d, int nint, double taumax, int32_t *ideep_d,
                int ilay, int nwave, void *stream)
{
    PB_REQUIRE(nwave >= 0 && nint >= 0, "pb_optdepth: negative size");
    if (nwave == 0)
        return PB_OK;
    PB_REQUIRE(tau_d && data_d && ideep_d && (nint == 0 || intervals_d),
               "pb_optdepth: null pointer");
    PB_REQUIRE(row_stride >= nwave, "pb_optdepth: row_stride < nwave");
    k_optdepth<<<pb::div_up(nwave, kBlock), kBlock, 0, pb::as_stream(stream)>>>(
        tau_d, data_d, row_stride, intervals_d, nint, taumax, ideep_d, ilay, nwave);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

static int transit_launch(double *depth_d, int32_t *ideep_d, double *spectrum_d,
                          const double *ec_d, const double *raypath_d, const double *radius_d,
                          double rstar, int itop, int ibottom, double maxdepth, int nlayers,
                          int nwave, void *stream, const char *who, int deck_itop = -1,
                          double deck_rsurf = 0.0)
{
    PB_REQUIRE(nlayers > 0 && nwave >= 0, "%s: bad shape", who);
    PB_REQUIRE(itop >= 0 && itop < nlayers, "%s: itop out of range", who);
    PB_REQUIRE(ibottom <= nlayers, "%s: ibottom > nlayers", who);
    if (nwave == 0)
        return PB_OK;
    PB_REQUIRE(depth_d && ideep_d && ec_d, "%s: null pointer", who);
    const int nrow = nlayers - itop;
    PB_REQUIRE(nrow == 1 || raypath_d, "%s: null raypath", who);
    PB_REQUIRE(!spectrum_d || radius_d, "%s: null radius", who);
    // radiative_transfer.py:63: the deck matters only when it lies below the top layer
    const int deck_row0 = deck_itop > itop ? deck_itop - itop : -1;
    // One spectrum: two kernels (tau for every row with grid.y over blocks of impact parameters,
    // then the early exit): a single grid of columns has too few wavefronts for the fused
    // one-pass kernel of pb_batch.hip (C2: 155 us fused against 97 us), which is what the
    // walker-batched path uses.  PB_TRANSIT=fused|split forces one form (tests compare them).
    const char *mode = getenv("PB_TRANSIT");
    const bool fused = mode ? !strcmp(mode, "fused") : false;
    if (fused)
        return pb_transit_fused_launch(depth_d, ideep_d, spectrum_d, ec_d, raypath_d, radius_d,
                                       ((int64_t)nrow * (nrow - 1)) / 2, rstar, itop, ibottom,
                                       maxdepth, nlayers, nwave, 1, deck_row0, deck_rsurf,
                                       pb::as_stream(stream), nullptr);
    const bool narrow = nwave <= kNarrowColumns;
    const int rows = narrow ? kRowsPerThreadNarrow : kRowsPerThread;
    const int threads = narrow ? 64 : kBlock;
    dim3 grid(pb::div_up(nwave, threads), pb::div_up(nrow, rows));
    const size_t lds = (size_t)nrow * rows * sizeof(double);
    if (lds > 64 * 1024) {
        pb::set_error("%s: %d layers need %zu B of LDS", who, nrow, lds);
        return PB_ERR_UNSUPPORTED;
    }
    // the ray paths re-laid per block of rows in the stream's scratch, so that the kernel can
    // take them with scalar loads (PB_TRANSIT_SCALAR=0: packed triangle staged in LDS)
    static const bool no_scalar = getenv("PB_TRANSIT_SCALAR") && atoi(getenv("PB_TRANSIT_SCALAR")) == 0;
    const int nimpact = std::min(ibottom, nlayers) - itop;
    double *blocked = nullptr;
    // (not while the stream is being captured: a graph must not depend on scratch that other work
    // on a stream of the same handle may overwrite when the graph is replayed elsewhere)
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(pb::as_stream(stream), &capturing) != hipSuccess)
        (void)hipGetLastError();
    if (!no_scalar && nimpact > 1 && capturing == hipStreamCaptureStatusNone) {
        int64_t plen = 0;
        if (pb_path_blocks_launch(&blocked, &plen, raypath_d, ((int64_t)nrow * (nrow - 1)) / 2,
                                  rows, nimpact, pb::as_stream(stream)) != PB_OK)
            blocked = nullptr;                     // fall back to the LDS form
    }
    if (blocked) {
        if (narrow)
            k_transit_tau<kRowsPerThreadNarrow, true><<<grid, threads, 0, pb::as_stream(stream)>>>(
                depth_d, ec_d, blocked, itop, ibottom, nlayers, nwave);
        else
            k_transit_tau<kRowsPerThread, true><<<grid, threads, 0, pb::as_stream(stream)>>>(
                depth_d, ec_d, blocked, itop, ibottom, nlayers, nwave);
    } else if (narrow)
        k_transit_tau<kRowsPerThreadNarrow, false><<<grid, threads, lds, pb::as_stream(stream)>>>(
            depth_d, ec_d, raypath_d, itop, ibottom, nlayers, nwave);
    else
        k_transit_tau<kRowsPerThread, false><<<grid, threads, lds, pb::as_stream(stream)>>>(
            depth_d, ec_d, raypath_d, itop, ibottom, nlayers, nwave);
    PB_LAUNCH_CHECK();
    // end of the reference's 'odepth' stage when this call also integrates the spectrum (the
    // finish pass resolves the early exit AND integrates: it is counted as 'spectrum')
    if (spectrum_d)
        pb::stage_boundary("odepth", "spectrum", pb::as_stream(stream));
    // radiative_transfer.py:63: the deck matters only when it lies below the top layer
    const int deck_row = deck_itop > itop ? deck_itop - itop : -1;
    k_transit_finish<<<pb::div_up(nwave, threads), threads, 0, pb::as_stream(stream)>>>(
        depth_d, ideep_d, spectrum_d, radius_d, rstar, itop, ibottom, maxdepth, nlayers,
        nwave, deck_row, deck_rsurf);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_optical_depth_transit(double *depth_d, int32_t *ideep_d, const double *ec_d,
                             const double *raypath_d, int itop, int ibottom,
                             double maxdepth, int nlayers, int nwave, void *stream)
{
    return transit_launch(depth_d, ideep_d, nullptr, ec_d, raypath_d, nullptr, 1.0, itop,
                          ibottom, maxdepth, nlayers, nwave, stream,
                          "pb_optical_depth_transit");
}

int pb_transit_spectrum(double *spectrum_d, double *depth_d, int32_t *ideep_d,
                        const double *ec_d, const double *raypath_d, const double *radius_d,
                        double rstar, int itop, int ibottom, double maxdepth, int nlayers,
                        int nwave, void *stream)
{
    PB_REQUIRE(spectrum_d, "pb_transit_spectrum: null spectrum");
    return transit_launch(depth_d, ideep_d, spectrum_d, ec_d, raypath_d, radius_d, rstar, itop,
                          ibottom, maxdepth, nlayers, nwave, stream, "pb_transit_spectrum");
}

int pb_transit_spectrum_deck(double *spectrum_d, double *depth_d, int32_t *ideep_d,
                             const double *ec_d, const double *raypath_d,
                             const double *radius_d, double rstar, int itop, int ibottom,
                             double maxdepth, int deck_itop, double deck_rsurf, int nlayers,
                             int nwave, void *stream)
{
    PB_REQUIRE(spectrum_d, "pb_transit_spectrum_deck: null spectrum");
    PB_REQUIRE(deck_itop < nlayers, "pb_transit_spectrum_deck: deck_itop out of range");
    return transit_launch(depth_d, ideep_d, spectrum_d, ec_d, raypath_d, radius_d, rstar, itop,
                          ibottom, maxdepth, nlayers, nwave, stream,
                          "pb_transit_spectrum_deck", deck_itop, deck_rsurf);
}

int pb_plane_parallel_optical_depth(double *depth_d, int32_t *ideep_d,
                                    const double *ec_d, const double *intervals_d,
                                    double maxdepth, int itop, int ibottom, int nlayers,
                                    int nwave, void *stream)
{
    PB_REQUIRE(nlayers > 0 && nwave >= 0, "pb_plane_parallel_optical_depth: bad shape");
    PB_REQUIRE(itop >= 0, "pb_plane_parallel_optical_depth: itop < 0");
    if (nwave == 0)
        return PB_OK;
    PB_REQUIRE(depth_d && ideep_d && ec_d && (nlayers == 1 || intervals_d),
               "pb_plane_parallel_optical_depth: null pointer");
    k_plane_depth<<<pb::div_up(nwave, kBlock), kBlock, 0, pb::as_stream(stream)>>>(
        depth_d, ideep_d, ec_d, intervals_d, maxdepth, itop, ibottom, nlayers, nwave);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_trapezoid2D(double *out_d, const double *data_d, const double *intervals_d,
                   const int32_t *nint_d, int nrows, int nwave, void *stream)
{
    PB_REQUIRE(nrows >= 1 && nwave >= 0, "pb_trapezoid2D: bad shape");
    if (nwave == 0)
        return PB_OK;
    PB_REQUIRE(out_d && data_d && nint_d && (nrows == 1 || intervals_d),
               "pb_trapezoid2D: null pointer");
    k_trapezoid2d<<<pb::div_up(nwave, kBlock), kBlock, 0, pb::as_stream(stream)>>>(
        out_d, data_d, intervals_d, nint_d, nrows, nwave);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_transmission_deck(double *spectrum_d, const double *depth_d, const int32_t *ideep_d,
                         const double *radius_d, int itop, double rstar, int deck_itop,
                         double deck_rsurf, int nlayers, int nwave, void *stream)
{
    PB_REQUIRE(nlayers > 0 && nwave >= 0, "pb_transmission: bad shape");
    PB_REQUIRE(itop >= 0 && itop < nlayers, "pb_transmission: itop out of range");
    PB_REQUIRE(deck_itop < nlayers, "pb_transmission: deck_itop out of range");
    if (nwave == 0)
        return PB_OK;
    PB_REQUIRE(spectrum_d && depth_d && ideep_d && radius_d, "pb_transmission: null pointer");
    const int deck_row = deck_itop > itop ? deck_itop - itop : -1;
    k_transmission<<<pb::div_up(nwave, kBlock), kBlock, 0, pb::as_stream(stream)>>>(
        spectrum_d, depth_d, ideep_d, radius_d, itop, rstar, nlayers, nwave, deck_row,
        deck_rsurf);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_transmission(double *spectrum_d, const double *depth_d, const int32_t *ideep_d,
                    const double *radius_d, int itop, double rstar, int nlayers,
                    int nwave, void *stream)
{
    return pb_transmission_deck(spectrum_d, depth_d, ideep_d, radius_d, itop, rstar, -1, 0.0,
                                nlayers, nwave, stream);
}

int pb_blackbody_wn_2D(double *B_d, const double *wn_d, int nwave, const double *temp_d,
                       int nlayers, const int32_t *last_d, void *stream)
{
    PB_REQUIRE(nwave >= 0 && nlayers >= 0, "pb_blackbody_wn_2D: bad shape");
    if (nwave == 0 || nlayers == 0)
        return PB_OK;
    PB_REQUIRE(B_d && wn_d && temp_d, "pb_blackbody_wn_2D: null pointer");
    PB_REQUIRE(nlayers <= 65535, "pb_blackbody_wn_2D: too many layers");
    dim3 grid(pb::div_up(nwave, kBlock), nlayers);
    k_blackbody2d<<<grid, kBlock, 0, pb::as_stream(stream)>>>(B_d, wn_d, nwave, temp_d,
                                                            nlayers, last_d);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_blackbody_wn(double *B_d, const double *wn_d, int nwave, double temp, void *stream)
{
    PB_REQUIRE(nwave >= 0, "pb_blackbody_wn: bad shape");
    if (nwave == 0)
        return PB_OK;
    PB_REQUIRE(B_d && wn_d, "pb_blackbody_wn: null pointer");
    k_blackbody1d<<<pb::div_up(nwave, kBlock), kBlock, 0, pb::as_stream(stream)>>>(
        B_d, wn_d, nwave, temp);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_emission_observables(double *spectrum_d, double *fplanet_d, const double *flux_d,
                            const double *starflux_d, const double *wn_d, int64_t nwave, int mode,
                            int dilute, double f_dilution, double scale, void *stream)
{
    PB_REQUIRE(nwave >= 0, "pb_emission_observables: bad shape");
    PB_REQUIRE(mode >= 0 && mode <= 2, "pb_emission_observables: mode %d (0 emission, 1 eclipse, "
                                      "2 f_lambda)", mode);
    if (nwave == 0)
        return PB_OK;
    PB_REQUIRE(spectrum_d && flux_d, "pb_emission_observables: null pointer");
    PB_REQUIRE(mode != 1 || starflux_d, "pb_emission_observables: eclipse needs the stellar flux");
    PB_REQUIRE(mode != 2 || wn_d, "pb_emission_observables: f_lambda needs the wavenumbers");
    k_emission_observables<<<(unsigned)pb::div_up(nwave, (int64_t)kBlock), kBlock, 0,
                             pb::as_stream(stream)>>>(spectrum_d, fplanet_d, flux_d, starflux_d,
                                                      wn_d, nwave, mode, dilute, f_dilution, scale);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_band_scale(double *bandflux_d, const double *band_scale_d, const double *walker_scale_d,
                  int nbands, int nwalkers, void *stream)
{
    PB_REQUIRE(nbands >= 0 && nwalkers >= 0, "pb_band_scale: bad sizes");
    const int64_t n = (int64_t)nbands * nwalkers;
    if (n == 0 || (!band_scale_d && !walker_scale_d))
        return PB_OK;
    PB_REQUIRE(bandflux_d, "pb_band_scale: null pointer");
    k_band_scale<<<(unsigned)pb::div_up(n, (int64_t)kBlock), kBlock, 0, pb::as_stream(stream)>>>(
        bandflux_d, band_scale_d, walker_scale_d, nbands, n);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_intensity(double *out_d, const double *tau_d, const int32_t *ideep_d,
                 const double *bbody_d, const double *mu_d, int nmu, int rtop,
                 int nlayers, int nwave, void *stream)
{
    PB_REQUIRE(nlayers > 0 && nwave >= 0 && nmu >= 0, "pb_intensity: bad shape");
    PB_REQUIRE(rtop >= 0 && rtop < nlayers, "pb_intensity: rtop out of range");
    if (nwave == 0 || nmu == 0)
        return PB_OK;
    PB_REQUIRE(out_d && tau_d && ideep_d && bbody_d && mu_d, "pb_intensity: null pointer");
    dim3 grid(pb::div_up(nwave, kBlock), nmu);
    k_intensity<<<grid, kBlock, 0, pb::as_stream(stream)>>>(out_d, tau_d, ideep_d, bbody_d,
                                                          mu_d, nmu, rtop, nlayers, nwave);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_emission_flux(double *flux_d, double *intensity_d, const double *tau_d,
                     const int32_t *ideep_d, const double *wn_d, const double *temp_d,
                     const double *mu_d, const double *weights_d, int nmu, int rtop,
                     int nlayers, int nwave, void *stream)
{
    return pb_emission_flux_deck(flux_d, intensity_d, tau_d, ideep_d, wn_d, temp_d, mu_d,
                                 weights_d, nmu, rtop, -1, nlayers, nwave, stream);
}

int pb_emission_flux_deck(double *flux_d, double *intensity_d, const double *tau_d,
                          const int32_t *ideep_d, const double *wn_d, const double *temp_d,
                          const double *mu_d, const double *weights_d, int nmu, int rtop,
                          int cloud_itop, int nlayers, int nwave, void *stream)
{
    PB_REQUIRE(nlayers > 0 && nwave >= 0, "pb_emission_flux: bad shape");
    PB_REQUIRE(cloud_itop < nlayers, "pb_emission_flux: cloud_itop out of range");
    PB_REQUIRE(nmu >= 1 && nmu <= kMaxMu, "pb_emission_flux: nmu must be in [1,%d]", kMaxMu);
    PB_REQUIRE(rtop >= 0 && rtop < nlayers, "pb_emission_flux: rtop out of range");
    if (nwave == 0)
        return PB_OK;
    PB_REQUIRE(flux_d && tau_d && ideep_d && wn_d && temp_d && mu_d && weights_d,
               "pb_emission_flux: null pointer");
    if (nmu <= 8)
        k_emission_flux<8><<<pb::div_up(nwave, kBlock), kBlock, ((size_t)2 * nlayers + kMaxMu) * 8, pb::as_stream(stream)>>>(
        flux_d, intensity_d, tau_d, ideep_d, wn_d, temp_d, mu_d, weights_d, nmu, rtop,
        nlayers, nwave, cloud_itop >= 0 ? cloud_itop : nlayers - 1);
    else
        k_emission_flux<kMaxMu><<<pb::div_up(nwave, kBlock), kBlock, ((size_t)2 * nlayers + kMaxMu) * 8, pb::as_stream(stream)>>>(
        flux_d, intensity_d, tau_d, ideep_d, wn_d, temp_d, mu_d, weights_d, nmu, rtop,
        nlayers, nwave, cloud_itop >= 0 ? cloud_itop : nlayers - 1);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_internal_flux(double *f_int_d, const double *wn_d, double tint, int nwave, void *stream)
{
    PB_REQUIRE(nwave >= 0, "pb_internal_flux: bad shape");
    if (nwave == 0)
        return PB_OK;
    PB_REQUIRE(f_int_d && wn_d, "pb_internal_flux: null pointer");
    k_internal_flux<<<1, kBlock, 0, pb::as_stream(stream)>>>(f_int_d, wn_d, tint, nwave);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_two_stream(double *flux_down_d, double *flux_up_d, const double *depth_d,
                  const double *wn_d, const double *temp_d, const double *f_int_d,
                  const double *flux_top_d, int rtop, int nlayers, int nwave, void *stream)
{
    PB_REQUIRE(nlayers >= 1 && nwave >= 0, "pb_two_stream: bad shape");
    PB_REQUIRE(rtop >= 0 && rtop < nlayers, "pb_two_stream: rtop out of range");
    if (nwave == 0)
        return PB_OK;
    PB_REQUIRE(flux_down_d && flux_up_d && depth_d && wn_d && temp_d,
               "pb_two_stream: null pointer");
    if (nlayers > 1) {
        dim3 tgrid((unsigned)pb::div_up(nwave, kBlock), (unsigned)(nlayers - 1));
        k_two_stream_trans<<<tgrid, kBlock, 0, pb::as_stream(stream)>>>(flux_up_d, depth_d,
                                                                         nlayers, nwave);
        PB_LAUNCH_CHECK();
    }
    k_two_stream<<<pb::div_up(nwave, kBlock), kBlock, 0, pb::as_stream(stream)>>>(
        flux_down_d, flux_up_d, depth_d, wn_d, temp_d, f_int_d, flux_top_d, rtop, nlayers,
        nwave);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_loglike(double *loglike_d, const double *bandflux_d, const double *data_d,
               const double *uncert_d, int nwalkers, int nbands, void *stream)
{
    PB_REQUIRE(nwalkers >= 0 && nbands >= 1, "pb_loglike: bad shape");
    if (nwalkers == 0)
        return PB_OK;
    PB_REQUIRE(loglike_d && bandflux_d && data_d && uncert_d, "pb_loglike: null pointer");
    k_loglike<<<nwalkers, 64, 0, pb::as_stream(stream)>>>(loglike_d, bandflux_d, data_d, uncert_d,
                                                         nbands);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

static int emission_batch(double *flux_d, const double *ec_d, const double *intervals_d,
                          const double *wn_d, const double *temp_d, const double *mu_d,
                          const double *weights_d, const int32_t *column_d, int nmu,
                          double maxdepth, int itop, int ibottom, int nlayers, int nwave,
                          int nwalkers, void *stream, const int32_t *tile_limit_d = nullptr,
                          int32_t *flags_d = nullptr, const int32_t *gate_d = nullptr);

int pb_emission_flux_batch(double *flux_d, const double *ec_d, const double *intervals_d,
                           const double *wn_d, const double *temp_d, const double *mu_d,
                           const double *weights_d, int nmu, double maxdepth, int itop,
                           int ibottom, int nlayers, int nwave, int nwalkers, void *stream)
{
    return emission_batch(flux_d, ec_d, intervals_d, wn_d, temp_d, mu_d, weights_d, nullptr, nmu,
                          maxdepth, itop, ibottom, nlayers, nwave, nwalkers, stream);
}

int pb_emission_flux_ordered(double *flux_d, const double *ec_d, const double *intervals_d,
                             const double *wn_d, const double *temp_d, const double *mu_d,
                             const double *weights_d, const int32_t *column_d, int nmu,
                             double maxdepth, int itop, int ibottom, int nlayers, int nwave,
                             int nwalkers, void *stream)
{
    PB_REQUIRE(column_d || nwave == 0, "pb_emission_flux_ordered: null column index");
    return emission_batch(flux_d, ec_d, intervals_d, wn_d, temp_d, mu_d, weights_d, column_d, nmu,
                          maxdepth, itop, ibottom, nlayers, nwave, nwalkers, stream);
}

int pb_emission_flux_limited(double *flux_d, const double *ec_d, const double *intervals_d,
                             const double *wn_d, const double *temp_d, const double *mu_d,
                             const double *weights_d, const int32_t *column_d, int nmu,
                             double maxdepth, int itop, int ibottom, int nlayers, int nwave,
                             int nwalkers, const int32_t *tile_limit_d, int32_t *flags_d,
                             const int32_t *gate_d, void *stream)
{
    PB_REQUIRE(column_d || nwave == 0, "pb_emission_flux_limited: null column index");
    PB_REQUIRE(!tile_limit_d || flags_d,
               "pb_emission_flux_limited: a tile limit needs flags[nwalkers + 1] to report the "
               "walkers that ran past it");
    return emission_batch(flux_d, ec_d, intervals_d, wn_d, temp_d, mu_d, weights_d, column_d, nmu,
                          maxdepth, itop, ibottom, nlayers, nwave, nwalkers, stream, tile_limit_d,
                          flags_d, gate_d);
}

static int emission_batch(double *flux_d, const double *ec_d, const double *intervals_d,
                          const double *wn_d, const double *temp_d, const double *mu_d,
                          const double *weights_d, const int32_t *column_d, int nmu,
                          double maxdepth, int itop, int ibottom, int nlayers, int nwave,
                          int nwalkers, void *stream, const int32_t *tile_limit_d,
                          int32_t *flags_d, const int32_t *gate_d)
{
    const pb::TileLimit lim{tile_limit_d, itop, gate_d};
    PB_REQUIRE(nlayers >= 1 && nwave >= 0 && nwalkers >= 0, "pb_emission_flux_batch: bad shape");
    PB_REQUIRE(nmu >= 1 && nmu <= kMaxMu, "pb_emission_flux_batch: nmu must be 1..%d", kMaxMu);
    PB_REQUIRE(itop >= 0 && itop < nlayers, "pb_emission_flux_batch: itop out of range");
    if (nwave == 0 || nwalkers == 0)
        return PB_OK;
    PB_REQUIRE(flux_d && ec_d && wn_d && temp_d && mu_d && weights_d &&
                   (nlayers == 1 || intervals_d),
               "pb_emission_flux_batch: null pointer");
    dim3 grid(pb::div_up(nwave, kBlock), nwalkers);
    // (quadratures of up to 8 angles -- the default raygrid has 5 -- keep 2 x 8 running values per
    // lane instead of 2 x 16: half the registers, twice the wavefronts per SIMD)
    if (nmu <= 8)
        k_emission_fused<8><<<grid, kBlock, ((size_t)2 * nlayers + kMaxMu) * 8, pb::as_stream(stream)>>>(
        flux_d, ec_d, intervals_d, wn_d, temp_d, mu_d, weights_d, nmu, maxdepth, itop, ibottom,
        nlayers, nwave, column_d, lim, flags_d);
    else
        k_emission_fused<kMaxMu><<<grid, kBlock, ((size_t)2 * nlayers + kMaxMu) * 8, pb::as_stream(stream)>>>(
        flux_d, ec_d, intervals_d, wn_d, temp_d, mu_d, weights_d, nmu, maxdepth, itop, ibottom,
        nlayers, nwave, column_d, lim, flags_d);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_simps2D(double *out_d, const double *y_d, int ny, int nwave, const double *h_d,
               const int32_t *nint_d, const double *hsum_d, const double *hratio_d,
               const double *hfactor_d, void *stream)
{
    PB_REQUIRE(ny >= 0 && nwave >= 0, "pb_simps2D: bad shape");
    if (nwave == 0)
        return PB_OK;
    PB_REQUIRE(out_d && nint_d && (ny == 0 || y_d), "pb_simps2D: null pointer");
    PB_REQUIRE(ny < 3 || (h_d && hsum_d && hratio_d && hfactor_d), "pb_simps2D: null h");
    k_simps2d<<<pb::div_up(nwave, kBlock), kBlock, 0, pb::as_stream(stream)>>>(
        out_d, y_d, ny, nwave, h_d, nint_d, hsum_d, hratio_d, hfactor_d);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_band_integrate(double *bandflux_d, const double *spectrum_d, const double *wn_d,
                      const int32_t *band_start_d, const int32_t *band_count_d,
                      const double *response_d, const int64_t *response_offset_d,
                      int nbands, int64_t wbegin, int64_t wcount, void *stream)
{
    PB_REQUIRE(nbands >= 0 && wbegin >= 0 && wcount >= 0, "pb_band_integrate: bad sizes");
    if (nbands == 0)
        return PB_OK;
    PB_REQUIRE(bandflux_d && spectrum_d && wn_d && band_start_d && band_count_d &&
                   response_d && response_offset_d,
               "pb_band_integrate: null pointer");
    k_band_integrate<<<nbands, kBlock, 0, pb::as_stream(stream)>>>(
        bandflux_d, spectrum_d, wn_d, band_start_d, band_count_d, response_d,
        response_offset_d, wbegin, wcount);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // extern "C"
