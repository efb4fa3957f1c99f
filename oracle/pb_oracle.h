/* TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's hot-path algorithms (pcubillos/pyratbay
 * v2.0.1, src_c/).  This is the parity oracle: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (pyratbay_amd/)
 * never links, imports or calls anything declared here.
 *
 * Parity pin: every function is checked in tests/test_oracle_vs_golden.py against
 * golden vectors produced by the UNMODIFIED reference C (oracle/_ref, built by
 * oracle/Makefile from /root/reference/src_c) -- see tests/golden/make_golden.py.
 *
 * Conventions: all arrays C-contiguous; doubles are IEEE binary64; integer arrays
 * are int32 (the reference reads/writes C `int` through the NumPy stride,
 * src_c/include/ind.h:31-37).  Citations are file:line under /root/reference/.
 */
#ifndef PB_ORACLE_H
#define PB_ORACLE_H

#include <stdint.h>

/* ---- Voigt profile table: src_c/vprofile.c:42-114, include/voigt.h:147-359 ---- */
double orc_voigt_point(double x, double y, double alphaD);
int orc_voigt_profile(int nwn, double half, double alphaL, double alphaD,
                      double *vpro, int quick);
int orc_voigt_grid(double *profile, int64_t nprofile, int32_t *psize, int32_t *pindex,
                   const double *lorentz, int nlor, const double *doppler, int ndop,
                   double dwn);

/* ---- Line-by-line extinction: src_c/_extcoeff.c:87-345 ---- */
typedef struct {
    int32_t ofactor;   /* dynamic-sampling factor chosen for the layer */
    int32_t nadd;      /* co-added lines      (_extcoeff.c:256)        */
    int32_t nskip;     /* skipped groups      (_extcoeff.c:266)        */
    int32_t neval;     /* evaluated profiles  (_extcoeff.c:308)        */
} orc_ext_stats;

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
    orc_ext_stats *stats);

/* ---- Cross-section table interpolation: src_c/_extcoeff.c:367-472 ---- */
int orc_interp_ec(double *extinction, const double *etable, const double *ttable,
                  const double *temperatures, const double *density,
                  int nmol, int ntemp, int nlayers, int nwave,
                  int lay1, int lay2, int per_mol);

/* ---- Trapezoid family: src_c/_trapezoid.c ---- */
double orc_trapezoid(const double *data, const double *h, int nint);
void orc_trapezoid2D(double *out, const double *data, const double *h,
                     const int32_t *nint, int nwave);
int orc_cumulative_sum(double *out, const double *data, const double *h, int nint,
                       double threshold);
void orc_plane_parallel_optical_depth(double *depth, int32_t *ideep, const double *ec,
                                      const double *h, double maxdepth, int itop,
                                      int ibottom, int nlayers, int nwave);
void orc_optdepth(double *tau, const double *data, const double *h, int nint,
                  double taumax, int32_t *ideep, int ilay, int nwave);
void orc_intensity(double *out, const double *tau, const int32_t *ideep,
                   const double *bbody, const double *mu, int nmu, int rtop,
                   int nlayers, int nwave);

/* ---- Planck function: src_c/_blackbody.c:35-130 ---- */
void orc_blackbody_wn_2D(double *B, const double *wn, int nwave, const double *temp,
                         int nlayers, const int32_t *last /* may be NULL */);
void orc_blackbody_wn(double *B, const double *wn, int nwave, double temp);

/* continuum terms with native arithmetic in the reference (_alkali.c, _spline.c) */
void orc_alkali_cross_section(double *ec, const double *pressure, const double *wn,
                              const double *temp, const double *voigt_det, double detuning_wn,
                              double mass, double lorentz_par, double part_func, double cutoff,
                              const double *wn0, const double *gf, int nlines, int nlayers,
                              int nwave);
void orc_second_deriv(double *y2nd, const double *yin, const double *xin, int nin);
void orc_splinterp_1D(double *yout, const double *yin, const double *xin, const double *y2nd,
                      int nin, const double *xout, int nout, double extrap);
int orc_lin_interp_2D(double *yout, const double *yin, const double *xin, const double *dy_dx,
                      int nin, int nin2, const double *xout, int nout, int lo, int hi);

/* two-stream fluxes (pyrat/spectrum.py:454-522) and scipy.special.exp1 */
double orc_exp1(double x);
void orc_internal_flux(double *f_int, const double *wn, int nwave, double tint);
void orc_two_stream(double *flux_down, double *flux_up, const double *depth, const double *B,
                    const double *f_int, const double *flux_top, int rtop, int nlayers,
                    int nwave);

/* ---- Simpson family: src_c/_simpson.c, include/simpson.h ---- */
void orc_geth(const double *h, int n, double *hsum, double *hratio, double *hfactor);
double orc_simps(const double *y, int n, const double *h, const double *hsum,
                 const double *hratio, const double *hfactor);
void orc_simps2D(double *out, const double *y, int nwave, const double *h,
                 const int32_t *nint, const double *hsum, const double *hratio,
                 const double *hfactor);

/* ---- helpers: src_c/cutils.c, _indices.c, include/utils.h ---- */
void orc_ediff(double *out, const double *arr, int n);
int orc_nearest(const double *array, double value, int lo, int hi);
int orc_pyramid(const double *array, double value, int lo, int hi);
void orc_arrbinsearch(int32_t *out, const double *values, int nvalues,
                      const double *array, int n);
int orc_ifirst(const int32_t *data, int n, int default_ret);
int orc_ilast(const int32_t *data, int n, int default_ret);

#endif
