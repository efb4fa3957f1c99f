/* pbhip.h -- C ABI of libpbhip.so: MI355X (gfx950) kernels for the Pyrat Bay
 * line-by-line opacity + radiative-transfer hot path.
 *
 * Every entry point replaces one native function of the reference (pyratbay v2.0.1,
 * CPython extension modules under src_c/, file:line cited per function) or one
 * per-layer / per-impact-parameter Python loop around it.  The reference has no
 * C-level API of its own (each function parses a PyObject* tuple); this header is
 * the interface a binding (ctypes stub in INTEGRATION.md) attaches to.
 *
 * Conventions
 *   - `_d` pointers are DEVICE memory (hipMalloc / torch.cuda), `_h` pointers are
 *     HOST memory.  All arrays are C-contiguous; doubles are binary64; integer
 *     arrays are int32 (the reference reads/writes C `int` through the NumPy stride,
 *     src_c/include/ind.h:31-37).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *     asynchronous with respect to the host unless stated otherwise.
 *   - Return value: PB_OK (0) or a negative PB_ERR_* code; pb_last_error() returns a
 *     thread-local message.  The library never calls exit() and has no CPU fallback:
 *     without a usable GPU every compute entry fails with PB_ERR_HIP.
 */
#ifndef PBHIP_H
#define PBHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PB_OK 0
#define PB_ERR_ARG (-1)         /* invalid argument / shape                      */
#define PB_ERR_HIP (-2)         /* HIP runtime error (message has the detail)    */
#define PB_ERR_UNSUPPORTED (-3) /* valid in the reference, not supported here    */
#define PB_ERR_NOMEM (-4)

const char *pb_last_error(void);
int pb_version(void);
int pb_device_count(int *count);
int pb_set_device(int device);

/* =========================================================================
 * Stage timers and profiler ranges  (Pyrat.timestamps['extinction'|'odepth'|'spectrum'],
 * pyratbay/pyrat/pyrat_obj.py:203-214 with the Timer of pyratbay/tools/tools.py:832-843)
 * ========================================================================= */
typedef struct pb_timer pb_timer;

/* A timer holds max_stages + 1 HIP events.  pb_timer_start records the first one on `stream`
 * (and opens a rocTX range `first_stage` when not NULL); pb_timer_mark records the END of the
 * stage `name` and opens the range of `next_stage` (NULL: none).  Entry points that fuse two of
 * the reference's stages (pb_transit_spectrum*: optical depth, then transmission) mark the
 * boundary themselves on the timer most recently started by the calling thread, under the
 * reference's stage name ("odepth").  Nothing synchronises until pb_timer_read, which waits for
 * the stage's closing event and returns its duration in seconds. */
int pb_timer_create(pb_timer **out, int max_stages);
int pb_timer_start(pb_timer *t, const char *first_stage, void *stream);
int pb_timer_mark(pb_timer *t, const char *name, const char *next_stage, void *stream);
int pb_timer_count(const pb_timer *t, int *nstages);
int pb_timer_read(pb_timer *t, int stage, char *name_out, int name_cap, double *seconds);
void pb_timer_destroy(pb_timer *t);
/* rocTX ranges for rocprofv3 --marker-trace (no-ops when the marker library is absent) */
int pb_range_push(const char *name);
int pb_range_pop(void);
int pb_roctx_available(void);

/* =========================================================================
 * Voigt profile table  (vprofile.grid, src_c/vprofile.c:42-114;
 *                       voigtn/voigtxy, src_c/include/voigt.h:147-359)
 * ========================================================================= */
typedef struct pb_voigt pb_voigt;

/* Build the table on the device from the width grids and the requested half-sizes
 * (psize_h[nlor*ndop]; 0 = alias the previous Doppler column, vprofile.c:100-104).
 * `dwn` is the fine-grid step (spec.ownstep), `osamp` the oversampling factor
 * (spec.wnosamp) that fixes the phase-major device layout used by the extinction
 * kernel.  keep_flat: 0 = phase-major layout only (constant-step plans), 1 = also the
 * reference layout (vprofile.grid's `profile`), 2 = the reference layout ONLY: what a plan of
 * the `resolution` / `wlstep` mode reads (utils.h:139-163) -- no second copy of the table.
 * Synchronous. */
int pb_voigt_create(pb_voigt **out, const double *lorentz_h, int nlor,
                    const double *doppler_h, int ndop, const int32_t *psize_h,
                    double dwn, int osamp, int keep_flat, void *stream);
/* Wrap an existing reference-layout table (what vprofile.grid returned) held on the
 * host: uploads it and derives the phase-major layout.  psize_h/pindex_h are the
 * arrays as left by vprofile.grid. */
int pb_voigt_from_flat(pb_voigt **out, const double *profile_h, int64_t nprofile,
                       const double *lorentz_h, int nlor, const double *doppler_h,
                       int ndop, const int32_t *psize_h, const int32_t *pindex_h,
                       int osamp, int keep_flat, void *stream);
/* Outputs of vprofile.grid: final half-sizes, start indices, number of samples. */
int pb_voigt_meta(const pb_voigt *v, int32_t *psize_h, int32_t *pindex_h,
                  int64_t *nprofile);
/* Reference-layout table (concatenated profiles) copied to host memory. */
int pb_voigt_flat_to_host(pb_voigt *v, double *profile_h, int64_t nprofile);
int64_t pb_voigt_device_bytes(const pb_voigt *v);
void pb_voigt_destroy(pb_voigt *v);

/* =========================================================================
 * Line list  (arrays read by _extcoeff.extinction, src_c/_extcoeff.c:87-123;
 *             TLI reader pyratbay/pyrat/line_by_line.py:298-482)
 * ========================================================================= */
typedef struct pb_lines pb_lines;

/* Upload a line list and pre-compute what depends only on (lwn, lID, own): the
 * in-range mask, the nearest fine-grid index of each line (_extcoeff.c:243-245) and
 * the greedy co-add groups (_extcoeff.c:248-262).  own_h may be NULL, in which case
 * own[i] = own0 + i*ownstep is generated exactly as NumPy does
 * (pyratbay/pyrat/spectrum.py:222). */
int pb_lines_create(pb_lines **out, const double *lwn_h, const double *elow_h,
                    const double *gf_h, const int32_t *lid_h, int64_t nlines, int niso,
                    const double *own_h, int64_t onwn, double own0, double ownstep);
/* stats[0]=lines in range, [1]=groups, [2]=co-added lines (nadd of _extcoeff.c:256) */
int pb_lines_stats(const pb_lines *l, int64_t stats[3]);
/* flag = 1 when the co-add groups were built on the device (lists in TLI order: isotope, then
 * wavenumber), 0 when the host loop did it (any other order, or PB_LINES_HOST=1). */
int pb_lines_grouped_on_device(const pb_lines *l, int *flag);
/* The co-add groups in (isotope, fine index) order: first line, number of lines, nearest
 * fine-grid index of the leader [ngroups each], and the first group of every isotope
 * [niso + 1].  Any pointer may be NULL. */
int pb_lines_groups(const pb_lines *l, int32_t *first_h, int32_t *count_h, int32_t *iown_h,
                    int64_t *iso_gstart_h);
void pb_lines_destroy(pb_lines *l);

/* =========================================================================
 * Line-by-line extinction for ALL layers in one call
 * (replaces the per-layer loop pyratbay/pyrat/extinction.py:170-213 around
 *  _extcoeff.extinction, src_c/_extcoeff.c:87-345)
 * ========================================================================= */
typedef struct pb_lbl pb_lbl;

int pb_lbl_create(pb_lbl **out, pb_voigt *voigt, pb_lines *lines,
                  const double *wn_h, int nwave,
                  const int32_t *divisors_h, int ndivs,
                  const double *molrad_h, const double *molmass_h, int nmol,
                  const int32_t *isoimol_h, const double *isomass_h,
                  const double *isoratio_h, const int32_t *isoiext_h, int niso,
                  double cutoff, double ethresh, int resolution, int max_layers);
/* Change the species-selection flags (isoiext < 0 = neglect; extinction.py:164-167) */
int pb_lbl_set_isoiext(pb_lbl *p, const int32_t *isoiext_h);
int pb_lbl_set_ethresh(pb_lbl *p, double ethresh);
/* Gather-kernel selection for constant-step grids.
 *   0 = automatic: layers whose phase-major profile blocks fit in LDS (narrow profiles) run
 *       the resident-profile kernel; the others run the LDS-staged kernel when several lines
 *       share a phase row of a tile, else the global gather;
 *   1 = global gather only, 2 = LDS-staged only (falls back to 1 when a phase row does not
 *       fit in LDS), 3 = resident-profile kernel where it applies + global gather,
 *   4, 5, 7 = measured dead ends (scatter, rounds, wave: see "Experiments" at the end of this
 *       header): refused by libpbhip.so, selectable in libpbhip_exp.so.
 * All sum the same terms; only the order differs (global, resident: isotope, position;
 * staged: isotope, phase, position).  In mode 0 the choice depends on the size of the launch
 * (layers x samples); with a fixed mode every tiling and sharding adds the same terms in the
 * same order, so shards concatenate bit-exactly.
 * Plans of the `resolution` mode (pb_lbl_create(resolution = 1)): any mode but 6 = the direct
 * gather (two reference-layout table reads per (line, output): no set-up, what a one-off call
 * wants); 6 = per-layer dynamic grids: the grid a layer's lines are summed on in the reference
 * (step = ofactor fine samples, _extcoeff.c:185-195, 281-307) is a constant-step grid, so one
 * constant-step sub-plan per factor in use (created on first use and kept, with the Lorentz rows
 * of the Voigt table its layers read re-cut into phase rows modulo the factor -- all factors
 * together about one more copy of the table) computes it with the staged kernels and the outputs
 * are interpolated from it as utils.h:139-163 does.  Same terms, summed in the staged order (~1e-16
 * of the direct gather).  Reads the layers' factors back: one stream synchronisation per call
 * (such a call cannot be captured into a HIP graph) and deals the runs of layers to side streams
 * of its own that join the caller's stream before the call returns.
 * Two-phase shard calls of such a plan use the direct gather.
 * last_gather_mode reports what the last call ran: 1 global, 2 staged, 3 resolution mode (direct
 * gather), 6 resolution mode through dynamic grids, plus 8 when the resident-profile kernel ran
 * as well. */
int pb_lbl_set_gather_mode(pb_lbl *p, int mode);
int pb_lbl_last_gather_mode(const pb_lbl *p, int *mode);
/* Hint: the caller keeps n independent calls in flight on n streams (the reference's callers
 * with independent spectra: the temperature loop of pyrat/extinction.py:100-122, the walkers of
 * a retrieval; here also the pipelined shards of a multi-GPU rank).  Changes only how a small
 * launch is tiled (fewer, longer workgroups: the other streams fill the chip), never a term. */
int pb_lbl_set_concurrency(pb_lbl *p, int n);
/* Out-of-core line lists.  The reference walks any number of lines sequentially
 * (src_c/_extcoeff.c:203-309; the "computer-intensive" opacity-table run,
 * pyratbay/pyrat/extinction.py:100-122, reads whole line databases).  Here a call keeps one
 * 16-byte record per (layer, co-add group); when those exceed `bytes` (default 96 GiB) the call
 * is made in chunks of the line list: one pass for the per-species maxima of all lines
 * (_extcoeff.c:203-226), then per chunk the records and a gather that continues the running
 * sums -- the same terms in the same order as an unchunked call with one workgroup per tile
 * (PB_STAGE_SPLIT=1), hence bit-identical to it.  last_chunks: chunks of the last call (0 = none). */
int pb_lbl_set_record_budget(pb_lbl *p, int64_t bytes);
int pb_lbl_last_chunks(const pb_lbl *p, int *chunks);
/* ext_d[nlayers, nrows, wcount] with nrows = 1 if add else (max isoiext)+1, for the
 * output samples [wbegin, wbegin+wcount) of the global grid (wavenumber shard).
 * temp_d[nlayers]; dens_d[nlayers, nmol]; isoz_d element (i,l) at
 * isoz_d[i*z_iso_stride + l*z_layer_stride].  Resample mode assigns, resolution
 * (linterp) mode accumulates into ext_d, like the reference. */
int pb_lbl_extinction(pb_lbl *p, double *ext_d, int64_t wbegin, int64_t wcount,
                      const double *temp_d, const double *dens_d, const double *isoz_d,
                      int64_t z_iso_stride, int64_t z_layer_stride, int nlayers, int add,
                      void *stream);
/* Diagnostics of the last call, copied to host (synchronises the stream):
 * ofactor_h[nlayers] (dynamic-sampling factor, _extcoeff.c:195) and
 * kmax_h[nlayers*nrows] (per-species maximum line strength, :225). */
int pb_lbl_last_state(pb_lbl *p, int32_t *ofactor_h, double *kmax_h, int nlayers,
                      int nrows, void *stream);
/* Which layers of the last call the resident-profile kernel computed (resident_h[nlayers],
 * 0/1; all 0 when that kernel was off) and the size in doubles of the largest phase-major
 * profile block each layer can select (block_h[nlayers]).  Synchronises the stream. */
int pb_lbl_last_layer_kinds(pb_lbl *p, int32_t *resident_h, int32_t *block_h, int nlayers,
                            void *stream);
/* The same call in two halves, for wavenumber shards on several GPUs.  _begin derives the layer
 * state and the records of the groups within reach of the shard, with the per-row maxima
 * (_extcoeff.c:203-226) over THOSE groups only -- 1/N of the exp() work; the caller then
 * all-reduces (MAX) the buffer pb_lbl_kmax_buffer returns over the ranks -- count 64-bit words
 * holding the bit patterns of non-negative doubles, which order like integers -- and _end runs
 * the gather with the global maxima (ethresh * kmax is then the same threshold on every rank). */
int pb_lbl_extinction_begin(pb_lbl *p, double *ext_d, int64_t wbegin, int64_t wcount,
                            const double *temp_d, const double *dens_d, const double *isoz_d,
                            int64_t z_iso_stride, int64_t z_layer_stride, int nlayers, int add,
                            void *stream);
int pb_lbl_kmax_buffer(pb_lbl *p, void **kmax_d, int64_t *count);
int pb_lbl_extinction_end(pb_lbl *p, void *stream);
/* Work of the last pb_lbl_extinction call, counted on the device from its per-(layer, group)
 * records: work[0] = profile samples multiplied (the FMAs _extcoeff.c:302-307 keeps after
 * resampling), work[1] = lanes the LDS-staged kernels issue for them (256-sample spans),
 * work[2] = live records.  All -1 when the last launch kept no packed records. */
int pb_lbl_last_work(pb_lbl *p, int64_t work[3], void *stream);
/* Diagnostics: distinct Voigt-table samples the live records of the last extinction call select
 * (per (layer, isotope, Doppler column, phase) row the longest window taken from it) -- what any
 * evaluation of _extcoeff.c:302-307 must read of `profile` at least once.  -1 when the last
 * launch kept no packed one-piece records. */
int pb_lbl_last_table_samples(pb_lbl *p, int64_t *samples, void *stream);
/* Per-launch timing of the gather kernel with HIP events on the call's stream:
 * begin() arms up to max_launches start/stop pairs, every following
 * pb_lbl_extinction records one pair around its gather launch, end() returns the summed
 * kernel time and the number of launches (used by bench.py for the roofline figure). */
int pb_lbl_timing_begin(pb_lbl *p, int max_launches);
int pb_lbl_timing_end(pb_lbl *p, double *total_ms, int *launches);
void pb_lbl_destroy(pb_lbl *p);

/* =========================================================================
 * Cross-section table interpolation (_extcoeff.interp_ec / interp_ec_per_mol,
 * src_c/_extcoeff.c:367-472).  etable_d[nmol,ntemp,nlayers,nwave]; ttable_d[ntemp];
 * temperatures_d[nlayers];
 * density_d[nlayers,nmol]; accumulates into extinction_d[nlayers,nwave]
 * (per_mol: [nmol,nlayers,nwave]) for layers lay1 <= k < min(lay2,nlayers).
 * ========================================================================= */
int pb_interp_ec(double *extinction_d, const double *etable_d, const double *ttable_d,
                 const double *temperatures_d, const double *density_d, int nmol,
                 int ntemp, int nlayers, int nwave, int lay1, int lay2, int per_mol,
                 void *stream);
/* Same, but the rows lay1 <= k < min(lay2,nlayers) are ASSIGNED (= 0 + the sum) instead of
 * accumulated into: for callers that would zero the array first (the retrieval loop,
 * opacity/line_sampling.py:438-456 after `self.ec = np.zeros(...)`): saves that pass and
 * the read of `extinction`. */
int pb_interp_ec_set(double *extinction_d, const double *etable_d, const double *ttable_d,
                     const double *temperatures_d, const double *density_d, int nmol,
                     int ntemp, int nlayers, int nwave, int lay1, int lay2, int per_mol,
                     void *stream);

/* =========================================================================
 * Optical depth
 * ========================================================================= */
/* One impact parameter: _trapezoid.optdepth (src_c/_trapezoid.c:238-276).
 * data_d[(nint+1), nwave] rows row_stride elements apart. */
int pb_optdepth(double *tau_d, const double *data_d, int64_t row_stride,
                const double *intervals_d, int nint, double taumax, int32_t *ideep_d,
                int ilay, int nwave, void *stream);
/* All impact parameters in one launch: the loop of
 * pyratbay/opacity/optic_depth.py:103-112 including the final
 * `ideep[ideep<0] = r`.  raypath_d is the packed lower triangle of
 * atmosphere.transit_path(radius, itop): row r (itop<=r<nlayers) has r-itop
 * entries starting at ((r-itop)*(r-itop-1))/2.  depth_d[nlayers,nwave] is fully
 * written (rows outside [itop,ibottom) and below ideep are zero). */
int pb_optical_depth_transit(double *depth_d, int32_t *ideep_d, const double *ec_d,
                             const double *raypath_d, int itop, int ibottom,
                             double maxdepth, int nlayers, int nwave, void *stream);
/* Same loop fused with radiative_transfer.transmission (no cloud deck,
 * pyratbay/spectrum/radiative_transfer.py:57-71): depth_d and ideep_d as above, plus
 * spectrum_d[nwave] = (r_top^2 + 2*integral)/rstar^2 computed in the pass that resolves
 * the early exit. */
int pb_transit_spectrum(double *spectrum_d, double *depth_d, int32_t *ideep_d,
                        const double *ec_d, const double *raypath_d, const double *radius_d,
                        double rstar, int itop, int ibottom, double maxdepth, int nlayers,
                        int nwave, void *stream);
/* _trapezoid.plane_parallel_optical_depth (src_c/_trapezoid.c:175-213); rows below
 * the stopping layer are left as passed in, like the reference. */
int pb_plane_parallel_optical_depth(double *depth_d, int32_t *ideep_d,
                                    const double *ec_d, const double *intervals_d,
                                    double maxdepth, int itop, int ibottom, int nlayers,
                                    int nwave, void *stream);

/* =========================================================================
 * Spectrum integrals
 * ========================================================================= */
/* _trapezoid.trapezoid2D (src_c/_trapezoid.c:70-90): data_d[nrows,nwave] */
int pb_trapezoid2D(double *out_d, const double *data_d, const double *intervals_d,
                   const int32_t *nint_d, int nrows, int nwave, void *stream);
/* Fused transmission spectrum = radiative_transfer.transmission without cloud deck
 * (pyratbay/spectrum/radiative_transfer.py:57-71): exp(-depth)*r, trapezoid2D over
 * ideep-itop intervals, (r_top^2 + 2*integral)/rstar^2. */
int pb_transmission(double *spectrum_d, const double *depth_d, const int32_t *ideep_d,
                    const double *radius_d, int itop, double rstar, int nlayers,
                    int nwave, void *stream);
/* _blackbody.blackbody_wn_2D / blackbody_wn (src_c/_blackbody.c:35-130);
 * last_d may be NULL. */
int pb_blackbody_wn_2D(double *B_d, const double *wn_d, int nwave, const double *temp_d,
                       int nlayers, const int32_t *last_d, void *stream);
int pb_blackbody_wn(double *B_d, const double *wn_d, int nwave, double temp, void *stream);
/* _trapezoid.intensity (src_c/_trapezoid.c:304-341): out_d[nmu,nwave] */
int pb_intensity(double *out_d, const double *tau_d, const int32_t *ideep_d,
                 const double *bbody_d, const double *mu_d, int nmu, int rtop,
                 int nlayers, int nwave, void *stream);
/* Fused emission: Planck evaluated in registers (never stored), intensity per mu and
 * flux = sum_k I_k * w_k (pyratbay/pyrat/spectrum.py:366-377).  intensity_d may be
 * NULL. */
int pb_emission_flux(double *flux_d, double *intensity_d, const double *tau_d,
                     const int32_t *ideep_d, const double *wn_d, const double *temp_d,
                     const double *mu_d, const double *weights_d, int nmu, int rtop,
                     int nlayers, int nwave, void *stream);
/* Radiative transfer with an opaque cloud deck (clouds/gray.py:92-150 Deck; the optical depth
 * is then computed with ibottom = deck_itop + 1, pyrat_obj.py:134-138):
 *  - transit (radiative_transfer.py:57-71): the interval that ends at layer deck_itop ends at
 *    the cloud top instead, h = deck_rsurf - radius[deck_itop-1], with the integrand
 *    interpolated linearly in radius; deck_itop <= itop or < 0 = no deck;
 *  - emission (radiative_transfer.py:121-131): the caller passes temp_d with the cloud-top
 *    temperature at index cloud_itop (Planck of that row) and ideep is clipped to cloud_itop;
 *    cloud_itop < 0 = no deck. */
int pb_transmission_deck(double *spectrum_d, const double *depth_d, const int32_t *ideep_d,
                         const double *radius_d, int itop, double rstar, int deck_itop,
                         double deck_rsurf, int nlayers, int nwave, void *stream);
int pb_transit_spectrum_deck(double *spectrum_d, double *depth_d, int32_t *ideep_d,
                             const double *ec_d, const double *raypath_d,
                             const double *radius_d, double rstar, int itop, int ibottom,
                             double maxdepth, int deck_itop, double deck_rsurf, int nlayers,
                             int nwave, void *stream);
int pb_emission_flux_deck(double *flux_d, double *intensity_d, const double *tau_d,
                          const int32_t *ideep_d, const double *wn_d, const double *temp_d,
                          const double *mu_d, const double *weights_d, int nmu, int rtop,
                          int cloud_itop, int nlayers, int nwave, void *stream);

/* Continuum opacity terms, accumulated into ec_d[nlayers,nwave] in ONE pass:
 *  - nrank1 terms cs_d[m][nwave] * f_d[m][nlayers]: Rayleigh (rayleigh.py:85-107, f = number
 *    density of the scatterer), Lecavelier haze (lecavelier.py:73-100, f = p/kT) and
 *    constant-cross-section gray clouds (gray.py:63-75, cs = 1, f = cs[l]*p/kT);
 *  - ncia (<= 4) collision-induced-absorption tables cia_tab_d[c] -> [ntemp_c][nwave] on the
 *    model grid, already per (molecule cm-3)^2, linear in temperature between the nodes
 *    cia_temps_d[c] with the bracket rule of _spline.c:219-260 (lin_interp_2D), columns
 *    [cia_lo[c], cia_hi[c]) only, times cia_f_d[c][nlayers] = product of the pair's
 *    densities (cia.py:119-215).  A temperature off a table gives NaN there (the reference
 *    raises ValueError; the Python front-end checks before the call).  cia_tab_d and
 *    cia_temps_d are HOST arrays of device pointers;
 *  - H- bound-free + free-free (hydrogen_ion.py:157-276, John 1988) when hm_sigma_bf_d is not
 *    NULL: hm_sigma_bf_d[nwave] (Eq. 4), hm_ff_d[6][nwave] = 1e-29 * the wavelength
 *    polynomials of Eq. 6 as rows multiplying sqrt(5040/T)^(i+2) (zero rows where a branch
 *    does not use a power), hm_f_d[nlayers] = n_H * n_e. */
int pb_continuum(double *ec_d, const double *wn_d, const double *temp_d, int nlayers, int nwave,
                 int nrank1, const double *cs_d, const double *f_d, int ncia,
                 const double *const *cia_tab_d, const double *const *cia_temps_d,
                 const int32_t *cia_ntemp, const int32_t *cia_lo, const int32_t *cia_hi,
                 const double *cia_f_d, const double *hm_sigma_bf_d, const double *hm_ff_d,
                 const double *hm_f_d, void *stream);
/* Alkali resonance doublets, src_c/_alkali.c:30-106 (alkali_cross_section): adds the cross
 * section (cm2 molecule-1) of nlines (<= 8) lines to ec_d[nlayers,nwave] -- times
 * density_d[nlayers] when that is not NULL (alkali.py:239-262).  pressure_d in barye,
 * voigt_det_d[nlayers,nlines] = Voigt value at the detuning distance (alkali.py:56-89, host). */
int pb_alkali_cross_section(double *ec_d, const double *pressure_d, const double *wn_d,
                            const double *temp_d, const double *voigt_det_d, double detuning,
                            double mass, double lorentz_par, double part_func, double cutoff,
                            const double *wn0_h, const double *gf_h, int nlines,
                            const double *density_d, int nlayers, int nwave, void *stream);

/* Gaussian log-likelihood of band-integrated models (tools/retrieval_tools.py:98-104):
 * loglike_d[w] = -0.5*sum_b ((data_d[b] - bandflux_d[w,b]) / uncert_d[b])^2
 *                -0.5*sum_b log(2*pi*uncert_d[b]^2), or -1e98 when that is not finite
 * (the reference's reject value, retrieval_tools.py:101-103). */
int pb_loglike(double *loglike_d, const double *bandflux_d, const double *data_d,
               const double *uncert_d, int nwalkers, int nbands, void *stream);

/* Two-stream fluxes (pyratbay/pyrat/spectrum.py:454-522, Heng et al. 2014 Eqs. B5-B6) from
 * the plane-parallel optical depth depth_d[nlayers,nwave] (computed with maxdepth = inf,
 * opacity/optic_depth.py:124-125): flux_down_d, flux_up_d [nlayers,nwave]; the emission
 * spectrum is flux_up_d row 0.  f_int_d[nwave] = internal flux added at the bottom (NULL =
 * none), flux_top_d[nwave] = beta_irr*(rstar/smaxis)^2*starflux written into row rtop
 * before the downward sweep, exactly like the reference (NULL = no irradiation).  exp1 is
 * scipy.special.exp1 (SciPy 1.15.3 xsf/expint.h:22-52). */
int pb_two_stream(double *flux_down_d, double *flux_up_d, const double *depth_d,
                  const double *wn_d, const double *temp_d, const double *f_int_d,
                  const double *flux_top_d, int rtop, int nlayers, int nwave, void *stream);
/* f_int of spectrum.py:475-478: Planck at tint scaled to a bolometric sigma*tint^4. */
int pb_internal_flux(double *f_int_d, const double *wn_d, double tint, int nwave, void *stream);

/* _simpson.simps2D (src_c/_simpson.c:167-203): y_d[ny,nwave] */
int pb_simps2D(double *out_d, const double *y_d, int ny, int nwave, const double *h_d,
               const int32_t *nint_d, const double *hsum_d, const double *hratio_d,
               const double *hfactor_d, void *stream);
/* cutils.ediff (src_c/cutils.c:27-42) */
int pb_ediff(double *out_d, const double *arr_d, int n, void *stream);

/* =========================================================================
 * Band integration (first "next" row after the path):
 * PassBand.integrate (pyratbay/spectrum/spec_tools.py:193-233) for nbands pass bands:
 * bandflux_d[b] = sum over the pairs (i,i+1) of band b's contiguous index range
 * [band_start, band_start+band_count) of 0.5*(wn[i+1]-wn[i])*(y_i+y_{i+1}),
 * y_i = spectrum[i]*response_b[i-band_start].  Only pairs whose left sample lies in
 * [wbegin, wbegin+wcount) are summed, so wavenumber shards can be all-reduced;
 * spectrum_d and wn_d are indexed on the global grid.  The caller applies the band's
 * `height` (and the wl factor of photon counting through the response).
 * ========================================================================= */
int pb_band_integrate(double *bandflux_d, const double *spectrum_d, const double *wn_d,
                      const int32_t *band_start_d, const int32_t *band_count_d,
                      const double *response_d, const int64_t *response_offset_d,
                      int nbands, int64_t wbegin, int64_t wcount, void *stream);

/* =========================================================================
 * Walker-batched retrieval path (BASELINE config 5): one launch per stage for nwalkers models,
 * no per-walker host work.  Reference inner loop: pyratbay/pyrat/pyrat_obj.py:225-385.
 * ========================================================================= */
/* atmosphere.transit_path (pyratbay/atmosphere/atmosphere.py:782-802) on the device:
 * radius_d[nwalkers,nlayers] -> raypath_d[nwalkers, n(n-1)/2], n = nlayers - itop, the packed
 * lower triangle pb_optical_depth_transit takes. */
int pb_transit_path(double *raypath_d, const double *radius_d, int itop, int nlayers,
                    int nwalkers, void *stream);
/* Partition functions of the isotopes of ONE TLI database at ntemp temperatures (the layers of
 * one atmosphere, or of a batch of walkers): what Line_By_Line.__init__ prepares with
 * scipy.interpolate.interp1d(db.temp, db.iso_pf[j], kind='slinear')
 * (pyratbay/pyrat/line_by_line.py:156-158) and calc_extinction_coefficient evaluates at the
 * temperature profile on EVERY call (:219-222).  ttab_d[ntab] ascending, pf_d[niso, ntab];
 * z_d[i*z_iso_stride + t*z_t_stride] (the isoz operand of pb_lbl_extinction takes the same two
 * strides).  Bit-equal to SciPy's first-order spline (its de Boor recurrence, restated).  A
 * temperature outside [ttab[0], ttab[ntab-1]] makes interp1d raise; here NaN is written and,
 * when nbad_d is not NULL, counted there (int32, zeroed by the caller). */
int pb_iso_partition(double *z_d, int64_t z_iso_stride, int64_t z_t_stride,
                     const double *temp_d, int64_t ntemp, const double *ttab_d, int ntab,
                     const double *pf_d, int niso, int32_t *nbad_d, void *stream);
/* Loader of sampled cross sections: one species' table of one opacity file
 * (in_d[ntemp_in, nlay_in, nwave_in], cm2 molecule-1) brought onto the run's grid
 * (out_d[ntemp_out, nlay_out, nwave_out]) -- tools.interpolate_opacity
 * (pyratbay/tools/tools.py:1026-1107) as Line_Sample.__init__ calls it
 * (pyratbay/opacity/line_sampling.py:245-275).  wsel_d[nwave_out]: the kept wavenumber samples
 * of the file (window + thinning).  Per output temperature / pressure the lower bracket node
 * (tlo_d, plo_d) and the weight of the upper one (tweight_d, pweight_d; 0 = on or beyond a
 * node: constant extrapolation), prepared on the host.  resample = 0: the file's own grid,
 * values pass untouched (brackets = the nodes, weights ignored); else linear in log(cs), zeros
 * entering as exp(-230).  accumulate != 0 adds to out_d (a species spread over several files). */
int pb_resample_cross_section(double *out_d, const double *in_d, const int32_t *wsel_d,
                              const int32_t *tlo_d, const double *tweight_d,
                              const int32_t *plo_d, const double *pweight_d, int ntemp_in,
                              int nlay_in, int nwave_in, int ntemp_out, int nlay_out,
                              int nwave_out, int resample, int accumulate, void *stream);
/* interp_ec (src_c/_extcoeff.c:367-418), assigning form, for a batch: temps_d[nwalkers,nlayers],
 * density_d[nwalkers,nlayers,nmol] -> ec_d[nwalkers,nlayers,nwave].  The table is read once per
 * chunk of walkers.  work_d: nwalkers*nlayers*136 bytes of device scratch.  nmol <= 8. */
int pb_interp_ec_batch(double *ec_d, const double *etable_d, const double *ttable_d,
                       const double *temps_d, const double *density_d, void *work_d, int nmol,
                       int ntemp, int nlayers, int nwave, int nwalkers, void *stream);
/* optic_depth.py:103-112 + radiative_transfer.py:57-71 (no cloud deck) for a batch:
 * ec_d[nwalkers,nlayers,nwave], raypath_d[nwalkers, n(n-1)/2], radius_d[nwalkers,nlayers] ->
 * spectrum_d[nwalkers,nwave]; depth_d[nwalkers,nlayers,nwave] and ideep_d[nwalkers,nwave] are
 * optional (NULL: not stored).  work_d: pb_transit_work_doubles(...) doubles of device scratch
 * (the ray paths re-laid for scalar loads), or NULL (ray paths staged in LDS: slower). */
int64_t pb_transit_work_doubles(int nlayers, int itop, int ibottom, int nwave, int nwalkers);
int pb_transit_spectrum_batch(double *spectrum_d, double *depth_d, int32_t *ideep_d,
                              const double *ec_d, const double *raypath_d,
                              const double *radius_d, double rstar, int itop, int ibottom,
                              double maxdepth, int nlayers, int nwave, int nwalkers,
                              void *work_d, void *stream);
/* The same batch with its columns in an order of the caller's choosing (retrieval batches:
 * columns sorted by the row at which a typical model crosses maxdepth).  ec_d[nwalkers,nlayers,
 * nwave] holds the columns IN THAT ORDER; column_d[nwave] gives the grid index of each, and
 * spectrum_d[nwalkers,nwave] is written in grid order.  The optical depths are formed row tile by
 * row tile and a wavefront stops at the tile in which its 32 columns have all crossed maxdepth --
 * the reference's per-column exit (_trapezoid.c:259-273) at tile granularity: neither the
 * remaining products nor the deeper layers of ec are touched.  Per column the arithmetic does not
 * depend on the order: spectra equal those of pb_transit_spectrum_batch bit for bit.  2 ... 128
 * impact parameters, nwave >= 2, no cloud deck; work_d as above (required). */
int pb_transit_spectrum_ordered(double *spectrum_d, const double *ec_d, const double *raypath_d,
                                const double *radius_d, const int32_t *column_d, double rstar,
                                int itop, int ibottom, double maxdepth, int nlayers, int nwave,
                                int nwalkers, void *work_d, void *stream);
/* The same two passes without the layers nobody reads.  With the columns in the depth order of a
 * base model, tile_limit_d[ceil(nwave / 256)] names for every block of 256 (ordered) columns the
 * last ROW TILE (16 impact parameters; tile m = layers itop + 16 m ... itop + 16 m + 15) any of
 * its columns is expected to need (the base model's deepest first crossing of maxdepth in the
 * block, _trapezoid.c:259-273, plus a margin):
 *   pb_interp_ec_batch_limited writes ec only for the layers row0 ... row0 + 16 (tile + 1) - 1 of
 *     a block (row0 = itop; the other elements of ec_d are left as they are);
 *   pb_transit_spectrum_limited stops a wavefront whose columns are still open beyond their
 *     block's tile, leaves their spectrum samples unwritten and sets flags_d[walker] = 1 and
 *     flags_d[nwalkers] = 1 (int32[nwalkers + 1], zeroed by the caller).
 * Repair without a host round trip: the same two calls again with tile_limit_d = NULL and
 *   gate_d = flags_d + nwalkers (interpolation: every workgroup returns at once unless *gate_d != 0;
 *     the per-walker weights in work_d are those of the first call) resp.
 *   gate_d = flags_d (transit: walker w is computed only if gate_d[w] != 0; the Q blocks in work_d
 *     are those of the first call).
 * The spectra are then bit-identical to pb_interp_ec_batch + pb_transit_spectrum_ordered whatever
 * the limits were (tests/test_gpu_batch.py::test_tile_limited_batch). */
int pb_interp_ec_batch_limited(double *ec_d, const double *etable_d, const double *ttable_d,
                               const double *temps_d, const double *density_d, void *work_d,
                               int nmol, int ntemp, int nlayers, int nwave, int nwalkers,
                               const int32_t *tile_limit_d, int row0, const int32_t *gate_d,
                               void *stream);
int pb_transit_spectrum_limited(double *spectrum_d, const double *ec_d, const double *raypath_d,
                                const double *radius_d, const int32_t *column_d, double rstar,
                                int itop, int ibottom, double maxdepth, int nlayers, int nwave,
                                int nwalkers, void *work_d, const int32_t *tile_limit_d,
                                int32_t *flags_d, const int32_t *gate_d, void *stream);
/* The emission counterpart (pb_emission_flux_ordered with the same limits: a column stops at the
 * first layer whose plane-parallel depth reaches maxdepth, _trapezoid.c:199-209; one still open at
 * layer itop + 16 (tile + 1) of its block is left unwritten and its walker flagged; gate_d as for
 * the transit call). */
int pb_emission_flux_limited(double *flux_d, const double *ec_d, const double *intervals_d,
                             const double *wn_d, const double *temp_d, const double *mu_d,
                             const double *weights_d, const int32_t *column_d, int nmu,
                             double maxdepth, int itop, int ibottom, int nlayers, int nwave,
                             int nwalkers, const int32_t *tile_limit_d, int32_t *flags_d,
                             const int32_t *gate_d, void *stream);
/* Emission geometry for a batch: plane_parallel_optical_depth (src_c/_trapezoid.c:175-213) +
 * blackbody + intensity + quadrature sum (pyrat/spectrum.py:366-377) in one pass, no cloud deck:
 * ec_d[nwalkers,nlayers,nwave], intervals_d[nwalkers,nlayers-1], temp_d[nwalkers,nlayers] ->
 * flux_d[nwalkers,nwave]. */
int pb_emission_flux_batch(double *flux_d, const double *ec_d, const double *intervals_d,
                           const double *wn_d, const double *temp_d, const double *mu_d,
                           const double *weights_d, int nmu, double maxdepth, int itop,
                           int ibottom, int nlayers, int nwave, int nwalkers, void *stream);
/* The same batch with its columns in an order of the caller's choosing (columns that become
 * optically thick at similar layers side by side: a wavefront walks the layers until its LAST lane
 * has reached maxdepth, _trapezoid.c:190-200).  ec_d and wn_d hold the columns in that order,
 * column_d[nwave] the grid index of each; flux_d[nwalkers,nwave] is written in grid order.  Per
 * column the arithmetic does not depend on the order: same bits as pb_emission_flux_batch. */
int pb_emission_flux_ordered(double *flux_d, const double *ec_d, const double *intervals_d,
                             const double *wn_d, const double *temp_d, const double *mu_d,
                             const double *weights_d, const int32_t *column_d, int nmu,
                             double maxdepth, int itop, int ibottom, int nlayers, int nwave,
                             int nwalkers, void *stream);
/* PassBand.integrate for a batch of full-grid spectra: bandflux_d[nwalkers,nbands]
 * (x heights_d[b] when given). */
int pb_band_integrate_batch(double *bandflux_d, const double *spectrum_d, const double *wn_d,
                            const int32_t *band_start_d, const int32_t *band_count_d,
                            const double *response_d, const int64_t *response_offset_d,
                            const double *heights_d, int nbands, int nwave, int nwalkers,
                            void *stream);
/* What the reference makes of a plane-parallel flux after the radiative transfer
 * (pyrat/spectrum.py:394-405; eval()'s f_lambda conversion, pyrat_obj.py:323-329), per sample:
 *   fplanet = flux_d [* f_dilution when dilute != 0]
 *   mode 0 emission: spectrum = fplanet
 *   mode 1 eclipse : spectrum = fplanet * (1/starflux_d * scale), scale = (rplanet/rstar)^2
 *   mode 2 f_lambda: spectrum = 10 * fplanet * (scale * wn_d * 1e-4)^2, scale = rplanet/distance
 * -- the NumPy expressions' products in their order (bit-equal).  fplanet_d may be NULL; spectrum_d
 * and fplanet_d may alias flux_d. */
int pb_emission_observables(double *spectrum_d, double *fplanet_d, const double *flux_d,
                            const double *starflux_d, const double *wn_d, int64_t nwave, int mode,
                            int dilute, double f_dilution, double scale, void *stream);
/* bandflux_d[w,b] *= walker_scale_d[w] (f_dilution of walker w, pyrat_obj.py:296-297) and
 * *= band_scale_d[b] (eclipse: rprs^2 / bandflux_star, pyrat_obj.py:662-665), in that order;
 * either pointer may be NULL. */
int pb_band_scale(double *bandflux_d, const double *band_scale_d, const double *walker_scale_d,
                  int nbands, int nwalkers, void *stream);
/* eval()'s reject path (pyrat_obj.py:302-320, 378-380): walkers with a temperature outside
 * [tmin, tmax] get bandflux = +inf. */
int pb_reject_walkers(double *bandflux_d, const double *temps_d, double tmin, double tmax,
                      int nlayers, int nbands, int nwalkers, void *stream);

/* =========================================================================
 * Experiments -- NOT in libpbhip.so.  `make -C pyratbay_amd/csrc EXPERIMENTS=1` builds
 * libpbhip_exp.so (compiled with -DPB_EXPERIMENTS) = the product library + the variants that were
 * built, verified and measured SLOWER than the default path (profiles/notes_r0*.md): gather modes
 * 4 (scatter), 5 (rounds), 7 (wave) of pb_lbl_set_gather_mode, the layers-outer and the one-pass
 * matrix transit kernels, the LDS-tile transit kernel, predicted run plans of the `resolution`
 * mode.  Their entry points:
 * ========================================================================= */
#ifdef PB_EXPERIMENTS
/* Which layers of the last call the wave-autonomous kernel computed (wave_h[nlayers], 0/1: the
 * layers whose longest phase row is at most 384 samples; all 0 when that kernel was off).
 * Synchronises the stream. */
int pb_lbl_last_wave_layers(pb_lbl *p, int32_t *wave_h, int nlayers, void *stream);
/* `resolution` plans in gather mode 6 (per-layer dynamic grids, _extcoeff.c:185-195, 320-326): the
 * run plan of a call -- which layers share an oversampling factor, which Lorentz rows of the re-cut
 * tables they read -- needs the layers' factors on the host: one stream synchronisation per call.
 * pb_lbl_set_dyn_predict(plan, 1) removes it: from the second call on the plan is taken from the
 * last read-back (asynchronous), a device check marks the layers it fits and the direct gather
 * computes the others, so the result is right whatever the prediction (to the direct gather's
 * 1e-12 of the dynamic grids; bit for bit while the prediction fits, i.e. always for a steady
 * atmosphere), and such calls can be captured into a HIP graph.  A read-back that contradicts its
 * prediction makes the next 8 calls synchronous.  Default off (PB_RES_DYN_PREDICT=1: on).
 * stats = {calls planned from a prediction, synchronous calls, read-backs that contradicted their
 * prediction}. */
int pb_lbl_set_dyn_predict(pb_lbl *p, int on);
int pb_lbl_dyn_stats(pb_lbl *p, int64_t stats[3]);
/* interp_ec + optical depth + transmission of a batch in ONE pass (the retrieval inner loop,
 * pyrat_obj.py:225-385: opacity/line_sampling.py:394-463 -> _extcoeff.c:367-418, then
 * opacity/optic_depth.py:103-112 and spectrum/radiative_transfer.py:57-71, no cloud deck):
 * etable_d[nmol,ntemp,nlayers,nwave], temps_d[nwalkers,nlayers], density_d[nwalkers,nlayers,nmol],
 * raypath_d[nwalkers, n(n-1)/2], radius_d[nwalkers,nlayers] -> spectrum_d[nwalkers,nwave].
 * The interpolated extinction is formed in registers as the operand of the FP64 matrix products
 * and never stored.  From two walkers on, neighbours in an order of the walkers by their table
 * brackets share a wavefront and one set of table loads (decided on the device; a batch whose
 * pairs bracket different temperatures in more than a fifth of the layers runs one walker per
 * wavefront).  pb_table_transit_supported(...) != 0 says whether the shape has this form
 * (2..128 impact parameters, nwave >= 2, one species' block of the table below 4 GiB); work_d: pb_table_transit_work_doubles(...) doubles. */
int pb_table_transit_supported(int nmol, int ntemp, int nlayers, int itop, int ibottom,
                               int nwave);
int64_t pb_table_transit_work_doubles(int nmol, int nlayers, int itop, int ibottom, int nwalkers);
int pb_table_transit_batch(double *spectrum_d, const double *etable_d, const double *ttable_d,
                           const double *temps_d, const double *density_d,
                           const double *raypath_d, const double *radius_d, double rstar,
                           int itop, int ibottom, double maxdepth, int nmol, int ntemp,
                           int nlayers, int nwave, int nwalkers, void *work_d, void *stream);
#endif /* PB_EXPERIMENTS */

#ifdef __cplusplus
}
#endif
#endif /* PBHIP_H */
