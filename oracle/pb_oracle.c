/* TEST INFRASTRUCTURE ONLY -- see pb_oracle.h.
 *
 * Plain-C restatement of the algorithms of pcubillos/pyratbay v2.0.1 src_c/ for the
 * line-by-line opacity + radiative-transfer hot path.  Same operation order and the
 * same (legacy CODATA) constants as the reference, so that it agrees with the
 * compiled reference to rounding level.  Not a product component.
 */
#include "pb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Constants, values as in src_c/include/constants.h:5-38 */
static const double K_PI = 3.141592653589793;
static const double K_SQRTLN2 = 0.83255461115769775635;
static const double K_TWOOSQRTPI = 1.12837916709551257389;
static const double K_SQRTLN2PI = 0.46971863934982566689;
static const double K_LS = 2.99792458e10;
static const double K_KB = 1.380658e-16;
static const double K_AMU = 1.66053886e-24;
static const double K_H = 6.6260755e-27;
static const double K_EC = 4.8032068e-10;
static const double K_ME = 9.1093897e-28;

/* =====================================================================
 * helpers: include/utils.h:44-89
 * ===================================================================== */

/* Index of the element of array[lo..hi] closest to value; bisection keeps the
 * upper half when array[mid] <= value; ties go to the lower index
 * (utils.h:75-89). */
int orc_nearest(const double *array, double value, int lo, int hi)
{
    while (hi - lo > 1) {
        int mid = (hi + lo) / 2;
        if (array[mid] > value)
            hi = mid;
        else
            lo = mid;
    }
    if (fabs(array[hi] - value) < fabs(array[lo] - value))
        return hi;
    return lo;
}

/* Galloping search upward from lo, then bisection, then closest of the two
 * brackets (utils.h:44-72). */
int orc_pyramid(const double *array, double value, int lo, int hi)
{
    int step = 1, top;
    if (value < array[lo])
        return lo;
    if (array[hi] < value)
        return hi;
    top = (lo + step < hi) ? lo + step : hi;
    while (array[top] < value) {
        step *= 2;
        top = (lo + step < hi) ? lo + step : hi;
    }
    while (top - lo > 1) {
        int mid = (top + lo) / 2;
        if (array[mid] < value)
            lo = mid;
        else
            top = mid;
    }
    if (fabs(array[top] - value) < fabs(array[lo] - value))
        return top;
    return lo;
}

void orc_ediff(double *out, const double *arr, int n)
{
    /* cutils.c:27-42 */
    for (int i = 0; i + 1 < n; i++)
        out[i] = arr[i + 1] - arr[i];
}

void orc_arrbinsearch(int32_t *out, const double *values, int nvalues,
                      const double *array, int n)
{
    /* cutils.c:63-80.  The reference passes hi=n (one past the end, :77); the
     * restatement searches [0, n-1], which is what an in-bounds run returns. */
    for (int i = 0; i < nvalues; i++)
        out[i] = orc_nearest(array, values[i], 0, n - 1);
}

int orc_ifirst(const int32_t *data, int n, int default_ret)
{
    /* _indices.c:42-59 */
    for (int i = 0; i < n; i++)
        if (data[i] == 1)
            return i;
    return default_ret;
}

int orc_ilast(const int32_t *data, int n, int default_ret)
{
    /* _indices.c:92-109 */
    for (int i = n - 1; i >= 0; i--)
        if (data[i] == 1)
            return i;
    return default_ret;
}

/* =====================================================================
 * Voigt profile: include/voigt.h:147-359, vprofile.c:42-114
 * ===================================================================== */

/* 1/(n! (2n+1)), the tabulated ferf[] of voigt.h:60-123 */
static long double g_ferf[64];
static int g_ferf_ready = 0;

static void ferf_init(void)
{
    long double fact = 1.0L;
    g_ferf[0] = 1.0;
    for (int n = 1; n < 64; n++) {
        fact *= (long double)n;
        /* table entries are doubles in the reference */
        g_ferf[n] = (long double)(double)(1.0L / (fact * (long double)(2 * n + 1)));
    }
    g_ferf_ready = 1;
}

/* Re[w(x+iy)] * sqrt(ln2/pi)/alphaD in the three regions of voigt.h:147-217 */
double orc_voigt_point(double x, double y, double alphaD)
{
    const double A1 = 0.46131350, A2 = 0.19016350, A3 = 0.09999216,
                 A4 = 1.78449270, A5 = 0.002883894, A6 = 5.52534370;
    const double B1 = 0.51242424, B2 = 0.27525510, B3 = 0.05176536,
                 B4 = 2.72474500;
    const long double x2y2 = x * x - y * y;
    const long double xy2 = 2 * x * y;

    if (!g_ferf_ready)
        ferf_init();

    if (x < 3 && y < 1.8) {
        /* Region I: fixed number of series terms (eps<0 path, voigt.h:127,165) */
        const int nterms = (x < 1 ? 15 : (int)(6.842 * x + 8.0)) + 1;
        const long double c = cosl(xy2), s = sinl(xy2);
        long double pr = y, pi = -x;   /* previous power, real/imag  */
        long double sr = y, si = -x;   /* running series, real/imag  */
        for (int i = 1; i <= nterms; i++) {
            long double qi = pr * xy2 + pi * x2y2;
            long double qr = pr * x2y2 - pi * xy2;
            si += qi * g_ferf[i];
            sr += qr * g_ferf[i];
            pi = qi;
            pr = qr;
        }
        return (double)(K_SQRTLN2PI / alphaD * exp((double)(-x2y2)) *
                        (c * (1 - sr * K_TWOOSQRTPI) - s * si * K_TWOOSQRTPI));
    }
    {
        const long double d2 = xy2 * xy2;
        const long double nx = xy2 * x;
        if (x < 5 && y < 5) {
            /* Region II (voigt.h:199-208) */
            long double t1 = x2y2 - A2, t2 = x2y2 - A4, t3 = x2y2 - A6;
            return (double)(K_SQRTLN2PI / alphaD *
                            (A1 * ((nx - t1 * y) / (t1 * t1 + d2)) +
                             A3 * ((nx - t2 * y) / (t2 * t2 + d2)) +
                             A5 * ((nx - t3 * y) / (t3 * t3 + d2))));
        }
        /* Region III (voigt.h:209-216) */
        long double t1 = x2y2 - B2, t2 = x2y2 - B4;
        return (double)(K_SQRTLN2PI / alphaD *
                        (B1 * ((nx - t1 * y) / (t1 * t1 + d2)) +
                         B3 * ((nx - t2 * y) / (t2 * t2 + d2))));
    }
}

/* One profile of nwn samples spanning [-half, +half] (voigt.h:222-295).
 * Three regimes: quick point sampling; two-point mean when the step already
 * resolves the Doppler core; Simpson bin means on an oversampled grid. */
int orc_voigt_profile(int nwn, double half, double alphaL, double alphaD,
                      double *vpro, int quick)
{
    const double y = K_SQRTLN2 * alphaL / alphaD;
    const double step = 2.0 * half / (nwn - 1);
    double fine = alphaD / (50 - 1);
    int over, nfine;
    double *buf;

    if (step < fine || quick) {
        over = 1;
        fine = step;
        nfine = nwn + 1;
    } else {
        over = (int)(step / fine) + 1;
        if (over & 1)
            over++;
        nfine = nwn * over + 1;
        fine = 2.0 * half / (nfine - 1);
    }
    buf = (double *)calloc((size_t)nfine, sizeof(double));
    if (!buf)
        return 0;
    for (int i = 0; i < nfine; i++) {
        double x = K_SQRTLN2 * fabs(fine * i - half) / alphaD;
        buf[i] = orc_voigt_point(x, y, alphaD);
    }
    if (quick) {
        memcpy(vpro, buf, (size_t)nwn * sizeof(double));
    } else if (((over + 1) & 1) != 0) {
        /* Simpson mean of each bin of `over` sub-intervals (voigt.h:300-331) */
        const double *in = buf;
        for (int o = 0; o < nwn; o++, in += over) {
            double acc = 0;
            for (int i = 1; i < over; i += 2)
                acc += in[i];
            acc *= 2;
            for (int i = 2; i < over; i += 2)
                acc += in[i];
            acc *= 2;
            acc += in[0] + in[over];
            vpro[o] = acc / (over * 3.0);
        }
    } else {
        /* trapezoid mean (voigt.h:336-359); over==1 -> mean of two points */
        const double *in = buf;
        for (int o = 0; o < nwn; o++, in += over) {
            double acc = 0;
            for (int i = 1; i < over; i++)
                acc += in[i];
            vpro[o] = (acc + (in[0] + in[over]) / 2.0) / (double)over;
        }
    }
    free(buf);
    return 1;
}

/* Concatenated table over the (nlor x ndop) width grid (vprofile.c:42-114).
 * psize==0 cells alias the previous Doppler column. */
int orc_voigt_grid(double *profile, int64_t nprofile, int32_t *psize, int32_t *pindex,
                   const double *lorentz, int nlor, const double *doppler, int ndop,
                   double dwn)
{
    int64_t idx = 0;
    for (int m = 0; m < nlor; m++) {
        for (int n = 0; n < ndop; n++) {
            int32_t half = psize[m * ndop + n];
            if (half != 0) {
                int nw = 2 * half + 1;
                if (idx + nw > nprofile)
                    return -1;
                if (!orc_voigt_profile(nw, dwn * (long)(nw / 2), lorentz[m],
                                       doppler[n], profile + idx, nw > 99999))
                    return 0;
                pindex[m * ndop + n] = (int32_t)idx;
                idx += nw;
            } else {
                if (n == 0)
                    return -2;   /* reference would read index[m,-1] */
                pindex[m * ndop + n] = pindex[m * ndop + n - 1];
                psize[m * ndop + n] = psize[m * ndop + n - 1];
            }
        }
    }
    return 1;
}

/* =====================================================================
 * Line-by-line extinction: _extcoeff.c:87-345
 * ===================================================================== */
int orc_extinction(
    double *ext, int nextinct_rows, int nwave,
    const double *profile, const int32_t *psize, const int32_t *pindex,
    const double *lorentz, int nlor, const double *doppler, int ndop,
    const double *wn, const double *own, int64_t onwn,
    const int32_t *divisors, int ndivs,
    const double *moldensity, const double *molrad, const double *molmass, int nmol,
    const int32_t *isoimol, const double *isomass, const double *isoratio,
    const double *isoz, const int32_t *isoiext, int niso,
    const double *lwn, const double *elow, const double *gf, const int32_t *lid,
    int64_t nlines,
    double cutoff, double ethresh, double temp, int add, int resolution,
    orc_ext_stats *stats)
{
    const double sigcte = K_PI * K_EC * K_EC / K_LS / K_LS / K_ME;
    const double expcte = K_H * K_LS / K_KB;
    const int nrows = add ? 1 : nextinct_rows;
    const double fdop = sqrt(2 * K_KB * temp / K_AMU) * K_SQRTLN2 / K_LS;
    const double flor = sqrt(2 * K_KB * temp / K_PI / K_AMU) / K_LS;
    double minwidth = 1e5;
    int nadd = 0, nskip = 0, neval = 0, d;

    double *alphal = (double *)malloc((size_t)niso * sizeof(double));
    double *alphad = (double *)malloc((size_t)niso * sizeof(double));
    int *idop = (int *)malloc((size_t)niso * sizeof(int));
    int *ilor = (int *)malloc((size_t)niso * sizeof(int));
    double *kprop = (double *)calloc((size_t)(nlines > 0 ? nlines : 1), sizeof(double));
    double *kmax = (double *)calloc((size_t)nrows, sizeof(double));
    /* one spare element: linterp reads sample ilo+1 (utils.h:159-160) */
    double *ktmp = (double *)calloc((size_t)nrows * (size_t)(onwn + 1), sizeof(double));
    const int64_t kstride = onwn + 1;

    /* widths per isotope (_extcoeff.c:151-183) */
    for (int i = 0; i < niso; i++) {
        int imol = isoimol[i];
        double acc = 0.0;
        for (int j = 0; j < nmol; j++) {
            double dia = molrad[imol] + molrad[j];
            acc += moldensity[j] * dia * dia * sqrt(1 / isomass[i] + 1 / molmass[j]);
        }
        alphal[i] = acc * flor;
        alphad[i] = fdop / sqrt(isomass[i]);
        {
            double vw = 0.5346 * alphal[i] +
                        sqrt(pow(alphal[i], 2) * 0.2166 + pow(alphad[i] * own[0], 2));
            minwidth = fmin(minwidth, vw);
        }
        idop[i] = orc_nearest(doppler, alphad[i] * own[0], 0, ndop - 1);
        ilor[i] = orc_nearest(lorentz, alphal[i], 0, nlor - 1);
    }

    const double wnstep = wn[1] - wn[0];
    const double ownstep = own[1] - own[0];
    /* dynamic sampling: largest divisor with >= 2 samples per min width (:185-195) */
    for (d = 1; d < ndivs; d++)
        if (divisors[d] * ownstep >= 0.5 * minwidth)
            break;
    const int ofactor = divisors[d - 1];
    const double dstep = ownstep * ofactor;
    const int64_t dnwn = 1 + (onwn - 1) / ofactor;
    const double wn_lo = own[0], wn_hi = own[onwn - 1];

    /* pass 1: line strengths and per-species maximum (:203-226) */
    for (int64_t ln = 0; ln < nlines; ln++) {
        int i = lid[ln];
        int iext = isoiext[i];
        double v = lwn[ln], k;
        if (iext < 0)
            continue;
        if (add)
            iext = 0;
        if (v < wn_lo || v > wn_hi)
            continue;
        k = sigcte * isoratio[i] * gf[ln] * exp(-expcte * elow[ln] / temp) *
            (1 - exp(-expcte * v / temp)) / isoz[i];
        kprop[ln] = k;
        kmax[iext] = fmax(kmax[iext], k);
    }

    /* pass 2: co-add, threshold, spread over the dynamic grid (:229-309) */
    for (int64_t ln = 0; ln < nlines; ln++) {
        int i = lid[ln];
        int iext = isoiext[i];
        double v = lwn[ln], k;
        int iown, idwn, subw, offset, half;
        long minj, maxj;
        if (iext < 0)
            continue;
        if (add)
            iext = 0;
        if (v < wn_lo || v > wn_hi)
            continue;

        iown = (int)((v - wn_lo) / ownstep);
        if (iown + 1 < onwn && fabs(v - own[iown + 1]) < fabs(v - own[iown]))
            iown++;

        k = kprop[ln];
        while (ln + 1 != nlines && lid[ln + 1] == i && lwn[ln + 1] <= wn_hi) {
            if (fabs(lwn[ln + 1] - own[iown]) < ownstep) {
                nadd++;
                ln++;
                k += kprop[ln];
            } else
                break;
        }
        if (k < ethresh * kmax[iext]) {
            nskip++;
            continue;
        }
        if (add)
            k *= moldensity[isoimol[i]];

        idwn = (int)((v - wn_lo) / dstep);
        idop[i] = orc_pyramid(doppler, alphad[i] * v, idop[i], ndop - 1);
        half = psize[ilor[i] * ndop + idop[i]];
        subw = iown - idwn * ofactor;
        offset = ofactor * idwn - half + subw;
        minj = idwn - (half - subw) / ofactor;
        maxj = idwn + (half + subw) / ofactor;
        if (minj < 0)
            minj = 0;
        if (maxj > dnwn)
            maxj = dnwn;
        if (cutoff > 0.0) {
            int mincut = (int)(idwn - cutoff / dstep);
            int maxcut = (int)(idwn + cutoff / dstep);
            if (mincut > minj)
                minj = mincut;
            if (maxcut < maxj)
                maxj = maxcut;
        }
        {
            const int32_t base = pindex[ilor[i] * ndop + idop[i]];
            int64_t jj = (int64_t)base + (int64_t)ofactor * minj - offset;
            double *dst = ktmp + (int64_t)iext * kstride;
            for (long j = minj; j < maxj; j++, jj += ofactor) {
                /* jj outside [base, base+2*half] is an out-of-profile read in the
                 * reference (possible only when half < ofactor); contributes 0 here */
                if (jj >= base && jj <= (int64_t)base + 2 * half)
                    dst[j] += k * profile[jj];
            }
        }
        neval++;
    }

    if (resolution == 1) {
        /* linear interpolation onto wn[] (utils.h:139-163); accumulates into ext */
        for (int r = 0; r < nrows; r++) {
            const double *src = ktmp + (int64_t)r * kstride;
            for (int i = 0; i < nwave; i++) {
                int ilo = (int)((wn[i] - wn[0]) / dstep);
                double wlo = wn[0] + dstep * ilo;
                ext[(int64_t)r * nwave + i] +=
                    (src[ilo] * (wlo + dstep - wn[i]) + src[ilo + 1] * (wn[i] - wlo)) /
                    dstep;
            }
        }
    } else {
        /* keep every scale-th dynamic sample (utils.h:119-135) */
        const int scale = (int)round(wnstep / ownstep / ofactor);
        const int64_t m = 1 + (dnwn - 1) / scale;
        for (int r = 0; r < nrows; r++) {
            const double *src = ktmp + (int64_t)r * kstride;
            for (int64_t j = 0; j < m && j < nwave; j++)
                ext[(int64_t)r * nwave + j] = src[(int64_t)scale * j];
        }
    }
    if (stats) {
        stats->ofactor = ofactor;
        stats->nadd = nadd;
        stats->nskip = nskip;
        stats->neval = neval;
    }
    free(alphal);
    free(alphad);
    free(idop);
    free(ilor);
    free(kprop);
    free(kmax);
    free(ktmp);
    return 1;
}

/* =====================================================================
 * Cross-section table interpolation: _extcoeff.c:367-472
 * etable[nmol, ntemp, nlayers, nwave]; density[nlayers, nmol];
 * extinction[nlayers, nwave] or (per_mol) [nmol, nlayers, nwave]; accumulates.
 * ===================================================================== */
int orc_interp_ec(double *extinction, const double *etable, const double *ttable,
                  const double *temperatures, const double *density,
                  int nmol, int ntemp, int nlayers, int nwave,
                  int lay1, int lay2, int per_mol)
{
    if (lay2 > nlayers)
        lay2 = nlayers;
    for (int k = lay1; k < lay2; k++) {
        double t = temperatures[k];
        int tlo = orc_nearest(ttable, t, 0, ntemp - 1), thi;
        double w_lo, w_hi, span;
        if (t < ttable[tlo] || tlo == ntemp - 1)
            tlo--;
        thi = tlo + 1;
        span = ttable[thi] - ttable[tlo];
        w_lo = (ttable[thi] - t) / span;
        w_hi = (t - ttable[tlo]) / span;
        for (int j = 0; j < nmol; j++) {
            double a = w_lo * density[k * nmol + j];
            double b = w_hi * density[k * nmol + j];
            const double *lo = etable + (((int64_t)j * ntemp + tlo) * nlayers + k) * nwave;
            const double *hi = etable + (((int64_t)j * ntemp + thi) * nlayers + k) * nwave;
            double *dst = per_mol ? extinction + ((int64_t)j * nlayers + k) * nwave
                                  : extinction + (int64_t)k * nwave;
            for (int i = 0; i < nwave; i++)
                dst[i] += lo[i] * a + hi[i] * b;
        }
    }
    return 1;
}

/* =====================================================================
 * Trapezoid family: _trapezoid.c
 * ===================================================================== */
double orc_trapezoid(const double *data, const double *h, int nint)
{
    /* _trapezoid.c:28-48 */
    double acc = 0;
    for (int i = 0; i < nint; i++)
        acc += h[i] * (data[i + 1] + data[i]);
    return 0.5 * acc;
}

void orc_trapezoid2D(double *out, const double *data, const double *h,
                     const int32_t *nint, int nwave)
{
    /* _trapezoid.c:70-90; data[nrows, nwave] */
    for (int j = 0; j < nwave; j++) {
        double acc = 0.0;
        for (int i = 0; i < nint[j]; i++)
            acc += h[i] * (data[(int64_t)i * nwave + j] + data[(int64_t)(i + 1) * nwave + j]);
        out[j] = acc * 0.5;
    }
}

int orc_cumulative_sum(double *out, const double *data, const double *h, int nint,
                       double threshold)
{
    /* _trapezoid.c:114-147 */
    out[0] = 0.0;
    if (nint < 1)
        return 0;
    for (int i = 0; i < nint; i++) {
        out[i + 1] = out[i] + 0.5 * h[i] * (data[i + 1] + data[i]);
        if (out[i + 1] >= threshold)
            return i + 1;
    }
    return nint;
}

void orc_plane_parallel_optical_depth(double *depth, int32_t *ideep, const double *ec,
                                      const double *h, double maxdepth, int itop,
                                      int ibottom, int nlayers, int nwave)
{
    /* _trapezoid.c:175-213: running trapezoid per column from itop downwards;
     * rows below the stopping layer are left as passed in. */
    for (int i = 0; i < nwave; i++) {
        double acc = 0.0;
        int k;
        for (k = 0; k < nlayers; k++) {
            if (k <= itop) {
                depth[(int64_t)k * nwave + i] = 0.0;
                continue;
            }
            acc += 0.5 * h[k - 1] *
                   (ec[(int64_t)k * nwave + i] + ec[(int64_t)(k - 1) * nwave + i]);
            depth[(int64_t)k * nwave + i] = acc;
            if (acc >= maxdepth || k == ibottom || k == nlayers - 1)
                break;
        }
        ideep[i] = k;
    }
}

void orc_optdepth(double *tau, const double *data, const double *h, int nint,
                  double taumax, int32_t *ideep, int ilay, int nwave)
{
    /* _trapezoid.c:238-276: twice the trapezoid (both halves of the chord);
     * columns already marked deep return 0. */
    for (int j = 0; j < nwave; j++) {
        double acc = 0.0;
        if (ideep[j] < 0) {
            for (int i = 0; i < nint; i++)
                acc += h[i] * (data[(int64_t)(i + 1) * nwave + j] + data[(int64_t)i * nwave + j]);
            if (acc > taumax)
                ideep[j] = ilay;
        }
        tau[j] = acc;
    }
}

void orc_intensity(double *out, const double *tau, const int32_t *ideep,
                   const double *bbody, const double *mu, int nmu, int rtop,
                   int nlayers, int nwave)
{
    /* _trapezoid.c:304-341 with tdiff/itrapezoid of utils.h:6-41 */
    double *dt = (double *)malloc((size_t)(nlayers > 0 ? nlayers : 1) * sizeof(double));
    for (int j = 0; j < nwave; j++) {
        int last = ideep[j];
        double taumax = tau[(int64_t)last * nwave + j];
        for (int k = 0; k < nmu; k++) {
            if (last - rtop == 1) {
                out[(int64_t)k * nwave + j] = bbody[(int64_t)last * nwave + j];
                continue;
            }
            for (int i = 0; i < last - rtop; i++)
                dt[i] = exp(-tau[(int64_t)(rtop + i + 1) * nwave + j] / mu[k]) -
                        exp(-tau[(int64_t)(rtop + i) * nwave + j] / mu[k]);
            {
                double acc = 0.0;
                for (int i = 0; i < last - rtop; i++)
                    acc += dt[i] * (bbody[(int64_t)(rtop + i + 1) * nwave + j] +
                                    bbody[(int64_t)(rtop + i) * nwave + j]);
                out[(int64_t)k * nwave + j] =
                    bbody[(int64_t)last * nwave + j] * exp(-taumax / mu[k]) - 0.5 * acc;
            }
        }
    }
    free(dt);
}

/* =====================================================================
 * Planck function: _blackbody.c:35-130
 * ===================================================================== */
void orc_blackbody_wn_2D(double *B, const double *wn, int nwave, const double *temp,
                         int nlayers, const int32_t *last)
{
    for (int i = 0; i < nwave; i++) {
        int ilast = last ? last[i] : nlayers - 1;
        double factor = 2 * K_H * K_LS * K_LS * pow(wn[i], 3);
        for (int j = 0; j <= ilast; j++)
            B[(int64_t)j * nwave + i] =
                factor / (exp(K_H * K_LS * wn[i] / (K_KB * temp[j])) - 1.0);
    }
}

void orc_blackbody_wn(double *B, const double *wn, int nwave, double temp)
{
    for (int i = 0; i < nwave; i++) {
        double factor = 2 * K_H * K_LS * K_LS * pow(wn[i], 3);
        B[i] = factor / (exp(K_H * K_LS * wn[i] / (K_KB * temp)) - 1.0);
    }
}

/* =====================================================================
 * Two-stream fluxes: pyratbay/pyrat/spectrum.py:454-522 (Heng et al. 2014, Eqs. B5-B6).
 * exp1 = scipy.special.exp1 (SciPy 1.15.3, scipy/special/xsf/expint.h:22-52: the E1XB
 * routine of Zhang & Jin, "Computation of Special Functions", 1996).
 * ===================================================================== */
double orc_exp1(double x)
{
    const double ga = 0.5772156649015328606065120900824024;     /* cephes SCIPY_EULER */
    if (x == 0.0)
        return INFINITY;
    if (x <= 1.0) {
        double e1 = 1.0, r = 1.0;
        for (int k = 1; k < 26; k++) {
            r = -r * k * x / pow(k + 1.0, 2);
            e1 += r;
            if (fabs(r) <= fabs(e1) * 1e-15)
                break;
        }
        return -ga - log(x) + x * e1;
    }
    int m = 20 + (int)(80.0 / x);
    double t0 = 0.0;
    for (int k = m; k > 0; k--)
        t0 = k / (1.0 + k / (x + t0));
    return exp(-x) * (1.0 / (x + t0));
}

/* Internal flux spectrum (spectrum.py:475-478): Planck at tint scaled so that its
 * trapezoid integral over wn equals sigma*tint^4; sigma = constants sigma (cgs). */
void orc_internal_flux(double *f_int, const double *wn, int nwave, double tint)
{
    const double sigma = 5.6703744191844314e-05;     /* astrophysical_constants.py:71 */
    orc_blackbody_wn(f_int, wn, nwave, tint);
    double total = 0.0;
    for (int i = 0; i + 1 < nwave; i++)
        total += (wn[i + 1] - wn[i]) * (f_int[i + 1] + f_int[i]) / 2.0;
    if (total > 0)
        for (int i = 0; i < nwave; i++)
            f_int[i] *= sigma * pow(tint, 4) / total;
}

/* depth[L,W], B[L,W] (Planck of every layer), f_int[W], flux_top[W] or NULL (the
 * irradiation written into row rtop before the downward sweep, :498-500; the sweep then
 * overwrites rows 1..L-1 exactly as the reference does) -> flux_down, flux_up [L,W]. */
void orc_two_stream(double *flux_down, double *flux_up, const double *depth, const double *B,
                    const double *f_int, const double *flux_top, int rtop, int nlayers,
                    int nwave)
{
    const double pi = 3.141592653589793;
    for (int j = 0; j < nwave; j++) {
        for (int i = 0; i < nlayers; i++) {
            flux_down[(int64_t)i * nwave + j] = 0.0;
            flux_up[(int64_t)i * nwave + j] = 0.0;
        }
        if (flux_top)
            flux_down[(int64_t)rtop * nwave + j] = flux_top[j];
        for (int i = 0; i < nlayers - 1; i++) {
            double dtau0 = depth[(int64_t)(i + 1) * nwave + j] - depth[(int64_t)i * nwave + j];
            double trans = (1 - dtau0) * exp(-dtau0) + dtau0 * dtau0 * orc_exp1(dtau0);
            double Bp = (B[(int64_t)(i + 1) * nwave + j] - B[(int64_t)i * nwave + j]) / dtau0;
            flux_down[(int64_t)(i + 1) * nwave + j] =
                trans * flux_down[(int64_t)i * nwave + j] +
                pi * B[(int64_t)i * nwave + j] * (1 - trans) +
                pi * Bp * (-2.0 / 3 * (1 - exp(-dtau0)) + dtau0 * (1 - trans / 3));
        }
        flux_up[(int64_t)(nlayers - 1) * nwave + j] =
            flux_down[(int64_t)(nlayers - 1) * nwave + j] + f_int[j];
        for (int i = nlayers - 2; i >= 0; i--) {
            double dtau0 = depth[(int64_t)(i + 1) * nwave + j] - depth[(int64_t)i * nwave + j];
            double trans = (1 - dtau0) * exp(-dtau0) + dtau0 * dtau0 * orc_exp1(dtau0);
            double Bp = (B[(int64_t)(i + 1) * nwave + j] - B[(int64_t)i * nwave + j]) / dtau0;
            flux_up[(int64_t)i * nwave + j] =
                trans * flux_up[(int64_t)(i + 1) * nwave + j] +
                pi * B[(int64_t)(i + 1) * nwave + j] * (1 - trans) +
                pi * Bp * (2.0 / 3 * (1 - exp(-dtau0)) - dtau0 * (1 - trans / 3));
        }
    }
}

/* =====================================================================
 * Continuum opacity terms with native arithmetic in the reference:
 * alkali doublets (src_c/_alkali.c:30-106) and the CIA interpolation helpers
 * (src_c/_spline.c:25-74 second_deriv, :95-131 splinterp_1D with include/spline.h:6-35,
 * :219-260 lin_interp_2D).
 * ===================================================================== */
void orc_alkali_cross_section(double *ec, const double *pressure, const double *wn,
                              const double *temp, const double *voigt_det, double detuning_wn,
                              double mass, double lorentz_par, double part_func, double cutoff,
                              const double *wn0, const double *gf, int nlines, int nlayers,
                              int nwave)
{
    const double K_ATM = 1010000.0, K_C2 = 1.4387768775039338, K_C3 = 8.852821681767784e-13;
    int flip = signbit(wn[1] - wn[0]);
    for (int j = 0; j < nlines; j++)
        for (int i = 0; i < nlayers; i++) {
            double lorentz = lorentz_par * pow(temp[i] / 2000.0, -0.7) * pressure[i] / K_ATM;
            double dsigma = detuning_wn * pow(temp[i] / 500.0, 0.6);
            for (int k = 0; k < nwave; k++) {
                int t = flip ? nwave - k - 1 : k;
                double dwn = wn[t] - wn0[j];
                double abs_dwn = fabs(dwn);
                if (dwn < -cutoff)
                    continue;
                else if (dwn > cutoff)
                    break;
                if (abs_dwn >= dsigma)
                    ec[(int64_t)i * nwave + t] +=
                        voigt_det[(int64_t)i * nlines + j] * pow(abs_dwn / dsigma, -1.5) * K_C3 *
                        gf[j] / part_func * exp(-K_C2 * (abs_dwn - dsigma) / temp[i]);
                else
                    ec[(int64_t)i * nwave + t] +=
                        lorentz / K_PI / (pow(lorentz, 2.0) + pow(dwn, 2.0)) * K_C3 * gf[j] /
                        part_func;
            }
        }
    (void)mass;     /* the Doppler width is computed but unused by the reference (:66-76) */
}

/* second derivatives for the natural cubic spline; NB the reference divides by
 * (xin[i+1] - YIN[i-1]) (_spline.c:52-53), kept as is */
void orc_second_deriv(double *y2nd, const double *yin, const double *xin, int nin)
{
    int n = nin - 1;
    double *u = calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    y2nd[0] = y2nd[n] = 0.0;
    u[0] = 0.0;
    for (int i = 1; i < n; i++) {
        double sig = (xin[i] - xin[i - 1]) / (xin[i + 1] - yin[i - 1]);
        double p = sig * y2nd[i - 1] + 2.0;
        y2nd[i] = (sig - 1.0) / p;
        u[i] = (yin[i + 1] - yin[i]) / (xin[i + 1] - xin[i]) -
               (yin[i] - yin[i - 1]) / (xin[i] - xin[i - 1]);
        u[i] = (6.0 * u[i] / (xin[i + 1] - xin[i - 1]) - sig * u[i - 1]) / p;
    }
    for (int i = n - 1; i >= 0; i--)
        y2nd[i] = y2nd[i] * y2nd[i + 1] + u[i];
    free(u);
}

void orc_splinterp_1D(double *yout, const double *yin, const double *xin, const double *y2nd,
                      int nin, const double *xout, int nout, double extrap)
{
    int lo = 0, hi = nout - 1;
    while (lo < nout && xout[lo] < xin[0])
        yout[lo++] = extrap;
    while (hi >= 0 && xout[hi] > xin[nin - 1])
        yout[hi--] = extrap;
    int i = 0;
    for (int n = lo; n <= hi; n++) {
        i = orc_nearest(xin, xout[n], i, nin - 1);
        if (i == nin - 1 || xout[n] < xin[i])
            i--;
        double dx = xin[i + 1] - xin[i];
        double a = (xin[i + 1] - xout[n]) / dx;
        double b = (xout[n] - xin[i]) / dx;
        yout[n] = a * yin[i] + b * yin[i + 1] +
                  ((a * a * a - a) * y2nd[i] + (b * b * b - b) * y2nd[i + 1]) * dx * dx / 6.0;
    }
}

/* yout[nout,nin2] rows lo..hi-1 of the second axis; returns 1 when an xout is off the table */
int orc_lin_interp_2D(double *yout, const double *yin, const double *xin, const double *dy_dx,
                      int nin, int nin2, const double *xout, int nout, int lo, int hi)
{
    for (int i = 0; i < nout; i++) {
        if (xout[i] < xin[0] || xout[i] > xin[nin - 1])
            return 1;
        int index = orc_nearest(xin, xout[i], 0, nin - 1);
        if (index == nin - 1 || xout[i] < xin[index])
            index--;
        if (xin[index] == xout[i]) {
            for (int j = lo; j < hi; j++)
                yout[(int64_t)i * nin2 + j] = yin[(int64_t)index * nin2 + j];
            continue;
        }
        double deltax = xout[i] - xin[index];
        for (int j = lo; j < hi; j++)
            yout[(int64_t)i * nin2 + j] =
                yin[(int64_t)index * nin2 + j] + deltax * dy_dx[(int64_t)index * nin2 + j];
    }
    return 0;
}

/* =====================================================================
 * Simpson family: _simpson.c:36-203, include/simpson.h:8-47
 * ===================================================================== */
void orc_geth(const double *h, int n, double *hsum, double *hratio, double *hfactor)
{
    /* pairs start at index n%2 (skip the first interval when n is odd) */
    int shift = n % 2;
    for (int i = 0; i < n / 2; i++) {
        int j = 2 * i + shift;
        hsum[i] = h[j] + h[j + 1];
        hratio[i] = h[j] / h[j + 1];
        hfactor[i] = hsum[i] * hsum[i] / (h[j] * h[j + 1]);
    }
}

static double simpson_core(const double *y, int64_t ystride, int n, const double *hsum,
                           const double *hratio, const double *hfactor)
{
    double acc = 0.0;
    for (int i = 0; i < (n - 1) / 2; i++) {
        int j = 2 * i;
        acc += (y[(int64_t)j * ystride] * (2.0 - 1.0 / hratio[i]) +
                y[(int64_t)(j + 1) * ystride] * hfactor[i] +
                y[(int64_t)(j + 2) * ystride] * (2.0 - hratio[i])) * hsum[i];
    }
    return acc / 6.0;
}

double orc_simps(const double *y, int n, const double *h, const double *hsum,
                 const double *hratio, const double *hfactor)
{
    double r;
    if (n < 2)
        return 0.0;
    if (n == 2)
        return h[0] * 0.5 * (y[0] + y[1]);
    r = simpson_core(y, 1, n, hsum, hratio, hfactor);
    if (n % 2 == 0)
        r += h[n - 2] * 0.5 * (y[n - 2] + y[n - 1]);
    return r;
}

void orc_simps2D(double *out, const double *y, int nwave, const double *h,
                 const int32_t *nint, const double *hsum, const double *hratio,
                 const double *hfactor)
{
    /* y[ny, nwave], integrate the first nint[i] rows of column i */
    for (int i = 0; i < nwave; i++) {
        int n = nint[i];
        if (n < 2)
            out[i] = 0.0;
        else if (n == 2)
            out[i] = h[0] * 0.5 * (y[i] + y[(int64_t)nwave + i]);
        else {
            double r = simpson_core(y + i, nwave, n, hsum, hratio, hfactor);
            if (n % 2 == 0)
                r += h[n - 2] * 0.5 *
                     (y[(int64_t)(n - 2) * nwave + i] + y[(int64_t)(n - 1) * nwave + i]);
            out[i] = r;
        }
    }
}
