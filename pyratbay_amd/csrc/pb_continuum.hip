// Continuum opacity terms added to the extinction coefficient on the device (SURVEY.md 8f
// rank 4): Rayleigh / Lecavelier / gray clouds (rank-1 terms), collision-induced absorption
// (linear in temperature between table nodes), H- bound-free + free-free, and the alkali
// resonance doublets.  All FP64, one pass over ec for the first three families
// (k_continuum), a windowed pass for the alkali lines (k_alkali).
//
// Reference arithmetic:
//   rank-1      pyratbay/opacity/rayleigh/rayleigh.py:85-107, clouds/lecavelier.py:73-100,
//               clouds/gray.py:63-75: cross_section[w] * density[l]
//   CIA         pyratbay/opacity/cia.py:119-215 -> src_c/_spline.c:219-260 (lin_interp_2D)
//   H-          pyratbay/opacity/hydrogen_ion.py:157-276 (John 1988, A&A 193, 189)
//   alkali      src_c/_alkali.c:30-106
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "pb_common.h"
#include "pbhip.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxCia = 4;
constexpr int kMaxAlkaliLines = 8;

// pyratbay/constants/astrophysical_constants.py:67-69 (scipy.constants, CODATA 2018): the
// Python-level models use these, not the legacy values of src_c/include/constants.h
constexpr double kPcH = 6.62607015e-27;
constexpr double kPcK = 1.380649e-16;
constexpr double kPcC = 29979245800.0;
constexpr double kWn0Bf = 6090.5;          // H- photo-detachment threshold (hydrogen_ion.py:30)

struct ContArgs {
    double *ec;
    const double *wn, *temp;
    int nlayers, nwave;
    int nrank1;
    const double *cs, *f;                  // [nrank1][nwave], [nrank1][nlayers]
    int ncia;
    const double *cia_tab[kMaxCia];        // [ntemp][nwave]
    const double *cia_temps[kMaxCia];      // [ntemp]
    int cia_ntemp[kMaxCia], cia_lo[kMaxCia], cia_hi[kMaxCia];
    const double *cia_f;                   // [ncia][nlayers]
    const double *hm_sigma_bf, *hm_ff, *hm_f;   // [nwave], [6][nwave], [nlayers]; null = no H-
};

// One workgroup = kContLayers layers x 256 samples.  Per-layer scalars (CIA brackets, H- powers)
// are prepared by the first lanes in LDS.  A thread keeps the layer-independent operands of its
// sample -- the rank-1 cross sections, the H- bound-free cross section, the six free-free rows and
// the wavenumber -- in registers across the layers (one layer per workgroup re-read those ~1.1 GB
// per C2-shaped call from L2, once per layer: 2.0 TB/s on the 394 MB the pass must move); per
// (layer, sample) the same products are added in the same order.
constexpr int kContLayers = 8;
constexpr int kContRank1 = 8;          // rank-1 terms kept in registers (more are re-read per layer)

__global__ __launch_bounds__(kBlock) void k_continuum(ContArgs a)
{
    __shared__ int s_idx[kContLayers][kMaxCia];
    __shared__ double s_dt[kContLayers][kMaxCia], s_inv[kContLayers][kMaxCia], s_beta[kContLayers][6],
        s_bfpre[kContLayers], s_ffpost[kContLayers];
    const int l0 = blockIdx.y * kContLayers;
    const int nl = min(kContLayers, a.nlayers - l0);
    if ((int)threadIdx.x < a.ncia * kContLayers && (int)threadIdx.x / a.ncia < nl) {
        // _spline.c:235-251: index = nearest node, stepped down unless it is at or below
        // the temperature; a temperature on a node takes that row unchanged
        const int q = threadIdx.x / a.ncia, c = threadIdx.x % a.ncia;
        const double temp = a.temp[l0 + q];
        const double *t = a.cia_temps[c];
        const int n = a.cia_ntemp[c];
        int idx = -1;
        double dt = 0.0, inv = 0.0;
        if (!(temp < t[0] || temp > t[n - 1])) {
            idx = pb::nearest_index(t, temp, 0, n - 1);
            if (idx == n - 1 || temp < t[idx])
                idx--;
            if (t[idx] != temp) {
                dt = temp - t[idx];
                inv = t[idx + 1] - t[idx];
            }
        }
        s_idx[q][c] = idx;
        s_dt[q][c] = dt;
        s_inv[q][c] = inv;
    }
    if (a.hm_sigma_bf && threadIdx.x >= 64 && threadIdx.x < 64 + 6 * kContLayers &&
        (int)(threadIdx.x - 64) / 6 < nl) {
        const int q = (threadIdx.x - 64) / 6, i = (threadIdx.x - 64) % 6;
        const double temp = a.temp[l0 + q];
        const double tc = fmin(fmax(temp, 1000.0), 10080.0);
        s_beta[q][i] = pow(sqrt(5040.0 / tc), (double)(i + 2));
        if (i == 0) {
            const double alpha = kPcH * kPcC / kPcK;
            s_bfpre[q] = 0.75 * pow(temp, -1.5) * kPcK * exp(kWn0Bf * alpha / temp);
            s_ffpost[q] = kPcK * tc;
        }
    }
    __syncthreads();
    const int w = blockIdx.x * kBlock + threadIdx.x;
    if (w >= a.nwave)
        return;
    // layer-independent operands of this sample
    double csv[kContRank1];
#pragma unroll
    for (int m = 0; m < kContRank1; m++)
        csv[m] = m < a.nrank1 ? a.cs[(int64_t)m * a.nwave + w] : 0.0;
    double wn = 0.0, sig = 0.0, ffv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (a.hm_sigma_bf) {
        wn = a.wn[w];
        sig = a.hm_sigma_bf[w];
#pragma unroll
        for (int i = 0; i < 6; i++)
            ffv[i] = a.hm_ff[(int64_t)i * a.nwave + w];
    }
    for (int q = 0; q < nl; q++) {
        const int l = l0 + q;
        const double temp = a.temp[l];
        const int64_t at = (int64_t)l * a.nwave + w;
        double ec = a.ec[at];
#pragma unroll
        for (int m = 0; m < kContRank1; m++)
            if (m < a.nrank1)
                ec += csv[m] * a.f[(int64_t)m * a.nlayers + l];
        for (int m = kContRank1; m < a.nrank1; m++)
            ec += a.cs[(int64_t)m * a.nwave + w] * a.f[(int64_t)m * a.nlayers + l];
        for (int c = 0; c < a.ncia; c++) {
            if (w < a.cia_lo[c] || w >= a.cia_hi[c])
                continue;
            const int idx = s_idx[q][c];
            double cs;
            if (idx < 0) {
                cs = NAN;                   // temperature off the table (the reference raises)
            } else {
                const double y0 = a.cia_tab[c][(int64_t)idx * a.nwave + w];
                cs = y0;
                if (s_inv[q][c] != 0.0) {
                    const double y1 = a.cia_tab[c][(int64_t)(idx + 1) * a.nwave + w];
                    cs = y0 + s_dt[q][c] * ((y1 - y0) / s_inv[q][c]);
                }
            }
            ec += cs * a.cia_f[(int64_t)c * a.nlayers + l];
        }
        if (a.hm_sigma_bf) {
            const double alpha = kPcH * kPcC / kPcK;
            const double bf = s_bfpre[q] * (1.0 - exp(-wn * alpha / temp)) * sig;
            // short-wavelength branch uses beta[0..3], the long one beta[1..5]; the unused
            // rows of hm_ff are zero.  Sum in ascending power like np.sum over that axis.
            double ff = 0.0;
#pragma unroll
            for (int i = 0; i < 6; i++)
                if (ffv[i] != 0.0)
                    ff += s_beta[q][i] * ffv[i];
            ff *= s_ffpost[q];
            ec += (bf + ff) * a.hm_f[l];
        }
        a.ec[at] = ec;
    }
}

struct AlkaliArgs {
    double *ec;
    const double *pressure, *wn, *temp, *voigt_det, *density;
    double detuning, lpar, part_func, cutoff;
    double wn0[kMaxAlkaliLines], gf[kMaxAlkaliLines];
    int nlines, nlayers, nwave;
};

// src_c/_alkali.c:58-104: thread = (layer, sample); lines in order; power-law wing outside
// the detuning distance, Lorentz core inside, hard cutoff.
__global__ __launch_bounds__(kBlock) void k_alkali(AlkaliArgs a)
{
    const double kAtm = 1010000.0, kC2 = 1.4387768775039338, kC3 = 8.852821681767784e-13;
    const int w = blockIdx.x * kBlock + threadIdx.x;
    const int l = blockIdx.y;
    const double temp = a.temp[l];
    // the layer's Lorentz width and detuning distance: two pow() per LAYER, evaluated once per
    // workgroup (every thread evaluating them was most of the pass on grids far from the lines)
    __shared__ double s_lorentz, s_dsigma;
    if (threadIdx.x == 0) {
        s_lorentz = a.lpar * pow(temp / 2000.0, -0.7) * a.pressure[l] / kAtm;
        s_dsigma = a.detuning * pow(temp / 500.0, 0.6);
    }
    __syncthreads();
    if (w >= a.nwave)
        return;
    const double lorentz = s_lorentz, dsigma = s_dsigma;
    const double wn = a.wn[w];
    double acc = 0.0;
    for (int j = 0; j < a.nlines; j++) {
        const double dwn = wn - a.wn0[j];
        const double abs_dwn = fabs(dwn);
        if (dwn < -a.cutoff || dwn > a.cutoff)
            continue;
        if (abs_dwn >= dsigma)
            acc += a.voigt_det[(int64_t)l * a.nlines + j] * pow(abs_dwn / dsigma, -1.5) * kC3 *
                   a.gf[j] / a.part_func * exp(-kC2 * (abs_dwn - dsigma) / temp);
        else
            acc += lorentz / pb::kPi / (pow(lorentz, 2.0) + pow(dwn, 2.0)) * kC3 * a.gf[j] /
                   a.part_func;
    }
    if (acc != 0.0) {
        const int64_t at = (int64_t)l * a.nwave + w;
        a.ec[at] += a.density ? acc * a.density[l] : acc;
    }
}

}  // namespace

extern "C" {

int pb_continuum(double *ec_d, const double *wn_d, const double *temp_d, int nlayers, int nwave,
                 int nrank1, const double *cs_d, const double *f_d, int ncia,
                 const double *const *cia_tab_d, const double *const *cia_temps_d,
                 const int32_t *cia_ntemp, const int32_t *cia_lo, const int32_t *cia_hi,
                 const double *cia_f_d, const double *hm_sigma_bf_d, const double *hm_ff_d,
                 const double *hm_f_d, void *stream)
{
    PB_REQUIRE(nlayers >= 0 && nwave >= 0, "pb_continuum: bad shape");
    PB_REQUIRE(nrank1 >= 0 && ncia >= 0 && ncia <= kMaxCia,
               "pb_continuum: at most %d CIA terms per call", kMaxCia);
    if (nlayers == 0 || nwave == 0)
        return PB_OK;
    PB_REQUIRE(ec_d && temp_d, "pb_continuum: null pointer");
    PB_REQUIRE(nrank1 == 0 || (cs_d && f_d), "pb_continuum: rank-1 terms without arrays");
    PB_REQUIRE(ncia == 0 || (cia_tab_d && cia_temps_d && cia_ntemp && cia_lo && cia_hi && cia_f_d),
               "pb_continuum: CIA terms without arrays");
    PB_REQUIRE(!hm_sigma_bf_d || (hm_ff_d && hm_f_d && wn_d), "pb_continuum: H- arrays missing");
    ContArgs a{};
    a.ec = ec_d;
    a.wn = wn_d;
    a.temp = temp_d;
    a.nlayers = nlayers;
    a.nwave = nwave;
    a.nrank1 = nrank1;
    a.cs = cs_d;
    a.f = f_d;
    a.ncia = ncia;
    for (int c = 0; c < ncia; c++) {
        PB_REQUIRE(cia_tab_d[c] && cia_temps_d[c] && cia_ntemp[c] >= 2,
                   "pb_continuum: CIA table %d is empty", c);
        PB_REQUIRE(cia_lo[c] >= 0 && cia_lo[c] <= cia_hi[c] && cia_hi[c] <= nwave,
                   "pb_continuum: CIA table %d column range", c);
        a.cia_tab[c] = cia_tab_d[c];
        a.cia_temps[c] = cia_temps_d[c];
        a.cia_ntemp[c] = cia_ntemp[c];
        a.cia_lo[c] = cia_lo[c];
        a.cia_hi[c] = cia_hi[c];
    }
    a.cia_f = cia_f_d;
    a.hm_sigma_bf = hm_sigma_bf_d;
    a.hm_ff = hm_ff_d;
    a.hm_f = hm_f_d;
    dim3 grid((unsigned)pb::div_up(nwave, kBlock), (unsigned)pb::div_up(nlayers, kContLayers));
    k_continuum<<<grid, kBlock, 0, pb::as_stream(stream)>>>(a);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_alkali_cross_section(double *ec_d, const double *pressure_d, const double *wn_d,
                            const double *temp_d, const double *voigt_det_d, double detuning,
                            double mass, double lorentz_par, double part_func, double cutoff,
                            const double *wn0_h, const double *gf_h, int nlines,
                            const double *density_d, int nlayers, int nwave, void *stream)
{
    (void)mass;          // the reference computes a Doppler width and never uses it (:66-76)
    PB_REQUIRE(nlayers >= 0 && nwave >= 0, "pb_alkali_cross_section: bad shape");
    PB_REQUIRE(nlines >= 0 && nlines <= kMaxAlkaliLines,
               "pb_alkali_cross_section: at most %d lines", kMaxAlkaliLines);
    if (nlayers == 0 || nwave == 0 || nlines == 0)
        return PB_OK;
    PB_REQUIRE(ec_d && pressure_d && wn_d && temp_d && voigt_det_d && wn0_h && gf_h,
               "pb_alkali_cross_section: null pointer");
    AlkaliArgs a{};
    a.ec = ec_d;
    a.pressure = pressure_d;
    a.wn = wn_d;
    a.temp = temp_d;
    a.voigt_det = voigt_det_d;
    a.density = density_d;
    a.detuning = detuning;
    a.lpar = lorentz_par;
    a.part_func = part_func;
    a.cutoff = cutoff;
    for (int j = 0; j < nlines; j++) {
        a.wn0[j] = wn0_h[j];
        a.gf[j] = gf_h[j];
    }
    a.nlines = nlines;
    a.nlayers = nlayers;
    a.nwave = nwave;
    dim3 grid((unsigned)pb::div_up(nwave, kBlock), (unsigned)nlayers);
    k_alkali<<<grid, kBlock, 0, pb::as_stream(stream)>>>(a);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // extern "C"
