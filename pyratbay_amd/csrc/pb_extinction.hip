// Line-by-line extinction for all layers of an atmosphere in one launch sequence.
//
// Restates _extcoeff.extinction (src_c/_extcoeff.c:87-345) as a GATHER:
//
//   reference : per layer, per line group: scatter k*profile[...] over a "dynamic"
//               fine grid (ktmp, up to W*osamp doubles), then keep every
//               (osamp/ofactor)-th sample (resample, utils.h:119-135).
//   here      : the kept samples are computed directly,
//                 ext[jo] = sum_groups k * profile_c[half + osamp*jo - iown],
//               restricted to the reference's window [minj,maxj) of that group
//               (_extcoeff.c:281-299), so ktmp never exists.  The result is the same
//               sum, term for term.
//
// Launch sequence per call (one stream, no host synchronisation):
//   1. k_layer_state : per layer/isotope Lorentz+Doppler widths, width-grid indices,
//                      dynamic-sampling factor (_extcoeff.c:138-200)
//   2. k_kmax        : per layer/species maximum line strength (_extcoeff.c:203-226)
//   3. k_ext_resample / k_ext_linterp : the gather.
//
// Gather kernel design (MI355X): a workgroup owns (layer, row, tile of 1024 output
// samples); a wavefront owns 4 consecutive 64-sample chunks and keeps their sums in
// registers, so every output is written exactly once, coalesced, without atomics and
// in a fixed order (bitwise reproducible).  Candidate groups of the tile are found by
// binary search in the (isotope, fine-index)-sorted group list; 256 of them at a time
// are turned -- one per lane -- into {strength, window, table offset} records in LDS
// (2 exp per group instead of per sample), then each wavefront walks the records that
// intersect its range (wave-level ballot culling).  The Voigt table is read through
// the phase-major layout of pb_voigt.hip: 64 lanes x 8 B contiguous per line chunk.
// Workgroups are numbered so that the 8 XCDs take different layers (the part of the
// table a layer touches -- one Lorentz row, a few Doppler columns -- then stays in
// that XCD's 4 MiB L2), heaviest (deepest) layers first.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <new>
#include <type_traits>
#include <vector>

#include "pb_common.h"
#include "pb_internal.h"
#include "pb_ext_args.h"

// `make EXPERIMENTS=1` (libpbhip_exp.so) keeps the measured dead ends selectable: gather modes 4
// (scatter), 5 (rounds: pb_rounds.hip) and 7 (wave: pb_wave.hip), the predicted run plans of the
// `resolution` mode.  The default library compiles none of them.
#ifdef PB_EXPERIMENTS
constexpr bool kExp = true;
#else
constexpr bool kExp = false;
#endif

using namespace pbx;

namespace {

constexpr int kBlock = 256;
constexpr int kChunks = 2;                       // 128-sample chunks per wavefront
constexpr int kLaneSamples = 2;                  // consecutive samples per lane (16-B loads)
constexpr int kChunk = 64 * kLaneSamples;        // samples per chunk
constexpr int kWaveSpan = kChunks * kChunk;      // samples per wavefront
constexpr int kTile = 4 * kWaveSpan;             // output samples per workgroup
constexpr int kRecs = 4;                         // records in flight per wavefront trip
static_assert(kTile <= kPmPad, "table padding must cover one tile");
static_assert(kTile < 65536, "window coordinates are packed in 16 bits");

// Line strength divided by the abundance (_extcoeff.c:219-224), same operation order; the three
// divisions by per-layer values through pb::quot(), the exponentials through pb::exp_s (the device
// library's exp arithmetic with its coefficients in scalar registers: same bits).  k_records
// evaluates this once per (layer, line): 2 exp + 3 divisions were 0.11 ms of every C2 spectrum.
__device__ inline double line_strength(double ratio, double gf, double elow, double wavn,
                                       double temp, double inv_temp, double z, double inv_z)
{
    const double k = pb::quot_fast(pb::kSigCte * ratio * gf *
                                       pb::exp_s(pb::quot_fast(-pb::kExpCte * elow, temp, inv_temp)) *
                                       (1 - pb::exp_s(pb::quot_fast(-pb::kExpCte * wavn, temp, inv_temp))),
                                   z, inv_z);
    // one test of the END result instead of one per quotient: a special value in any of the three
    // (temp or z zero / infinite, an infinite numerator) ends as NaN or 0 here -- then, and when
    // the strength has really underflowed, the reference's own divisions decide
    if (__builtin_expect(!(fabs(k) > 0.0), 0))
        return pb::kSigCte * ratio * gf * pb::exp_s(-pb::kExpCte * elow / temp) *
               (1 - pb::exp_s(-pb::kExpCte * wavn / temp)) / z;
    return k;
}

// ---------------------------------------------------------------------------
// 1. per-layer state: one workgroup (64 lanes) per layer, lanes over isotopes
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_layer_state(LblArgs a)
{
    __shared__ unsigned long long s_minwidth;
    __shared__ int s_block;          // largest phase-major profile block any isotope can use
    __shared__ int s_rowmax;         // longest phase row any isotope can select
    // the width grids and the divisors are searched serially by one or a few lanes: one
    // parallel copy into LDS first turns ~40 dependent global loads into LDS reads (the
    // kernel is on the critical path of every spectrum)
    extern __shared__ double s_grid[];                 // lorentz[nlor] | doppler[ndop] | divisors
    double *s_lor = s_grid, *s_dopp = s_grid + a.nlor;
    int *s_div = reinterpret_cast<int *>(s_dopp + a.ndop);
    for (int i = threadIdx.x; i < a.nlor; i += 64)
        s_lor[i] = a.lorentz[i];
    for (int i = threadIdx.x; i < a.ndop; i += 64)
        s_dopp[i] = a.doppler[i];
    for (int i = threadIdx.x; i < a.ndivs; i += 64)
        s_div[i] = a.divisors[i];
    const int layer = blockIdx.x;
    const double temp = a.temp[layer];
    const double fdop = sqrt(2 * pb::kKB * temp / pb::kAMU) * pb::kSqrtLn2 / pb::kLS;
    const double flor = sqrt(2 * pb::kKB * temp / pb::kPi / pb::kAMU) / pb::kLS;
    if (threadIdx.x == 0) {
        s_minwidth = __double_as_longlong(1e5);
        s_block = 0;
        s_rowmax = 0;
    }
    for (int r = threadIdx.x; r < a.nrows; r += 64)
        a.kmax_bits[(int64_t)layer * a.nrows + r] = 0ull;
    __syncthreads();
    const double *dens = a.dens + (int64_t)layer * a.nmol;
    for (int i = threadIdx.x; i < a.niso; i += 64) {
        const int imol = a.isoimol[i];
        double acc = 0.0;
        for (int j = 0; j < a.nmol; j++) {
            double dia = a.molrad[imol] + a.molrad[j];
            acc += dens[j] * dia * dia * sqrt(1 / a.isomass[i] + 1 / a.molmass[j]);
        }
        const double alphal = acc * flor;
        const double alphad = fdop / sqrt(a.isomass[i]);
        const double dw = alphad * a.own0;
        const double vw = 0.5346 * alphal + sqrt(alphal * alphal * 0.2166 + dw * dw);
        atomicMin(&s_minwidth, (unsigned long long)__double_as_longlong(vw));
        const int ilor = pb::nearest_index(s_lor, alphal, 0, a.nlor - 1);
        int hmax = 0;
        for (int d = 0; d < a.ndop; d++)
            hmax = max(hmax, a.psize[ilor * a.ndop + d]);
        // the resident kernel stages whole cells: only the Doppler columns that lines on
        // this grid can select matter (one column of margin on both sides)
        {
            const int dlo = max(0, pb::nearest_index(s_dopp, alphad * a.own0, 0, a.ndop - 1) - 1);
            const int dhi = min(a.ndop - 1,
                                pb::nearest_index(s_dopp, alphad * a.own_last, 0, a.ndop - 1) + 1);
            int used = 0, hlo = INT_MAX, hhi = 0;
            for (int d = dlo; d <= dhi; d++) {
                used = max(used, a.pm_stride[ilor * a.ndop + d]);
                hlo = min(hlo, a.psize[ilor * a.ndop + d]);
                hhi = max(hhi, a.psize[ilor * a.ndop + d]);
            }
            atomicMax(&s_block, used * a.osamp);
            atomicMax(&s_rowmax, used);
            a.li_rowmax[(int64_t)layer * a.niso + i] = used;
            a.li_hlo[(int64_t)layer * a.niso + i] = hlo;
            a.li_hhi[(int64_t)layer * a.niso + i] = hhi;
        }
        const int64_t k = (int64_t)layer * a.niso + i;
        a.li_alphad[k] = alphad;
        a.li_ilor[k] = ilor;
        a.li_hmax[k] = hmax;
        a.li_dens[k] = dens[imol];
        a.li_z[k] = a.isoz[i * a.z_iso_stride + layer * a.z_layer_stride];
        a.li_invz[k] = 1.0 / a.li_z[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double minwidth = __longlong_as_double((long long)s_minwidth);
        int d;
        for (d = 1; d < a.ndivs; d++)
            if (s_div[d] * a.ownstep >= 0.5 * minwidth)
                break;
        const int ofactor = s_div[d - 1];
        a.ls_ofactor[layer] = ofactor;
        a.ls_dwnstep[layer] = a.ownstep * ofactor;
        a.ls_cutsteps[layer] = a.cutoff / (a.ownstep * ofactor);
        a.ls_inv_ofactor[layer] = 1.0 / (double)ofactor;
        a.ls_inv_temp[layer] = 1.0 / temp;
        a.ls_inv_scale[layer] = 1.0 / (double)(int)round(a.wnstep / a.ownstep / ofactor);
        a.ls_dnwn[layer] = 1 + (a.onwn - 1) / ofactor;
        a.ls_scale[layer] = (int)round(a.wnstep / a.ownstep / ofactor);
        a.ls_resident[layer] = a.res_cap > 0 && s_block <= a.res_cap;
        a.ls_block[layer] = s_block;
        // (a resident layer stays the resident kernel's)
        a.ls_wave[layer] = a.wave_cap > 0 && !(a.res_cap > 0 && s_block <= a.res_cap) &&
                           s_rowmax <= a.wave_cap;
    }
}

// a wave-uniform int written by an earlier kernel, by a scalar load
__device__ __forceinline__ int uniform_load_i32(const int32_t *p, int64_t i)
{
    typedef const int32_t __attribute__((address_space(4))) *cptr;
    return ((cptr)(unsigned long long)p)[i];
}

// ---------------------------------------------------------------------------
// 2. per layer / output row maximum line strength over all in-range lines
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_kmax(LblArgs a, int lines_per_block)
{
    extern __shared__ unsigned long long s_max[];
    const int layer = blockIdx.y;
    if (a.lskip && uniform_load_i32(a.lskip, layer))
        return;
    for (int r = threadIdx.x; r < a.nrows; r += kBlock)
        s_max[r] = 0ull;
    __syncthreads();
    const double temp = a.temp[layer], inv_temp = a.ls_inv_temp[layer];
    const int64_t begin = (int64_t)blockIdx.x * lines_per_block;
    const int64_t end = min(begin + lines_per_block, a.nlines);
    int cur_row = -1;
    double cur_max = 0.0;
    for (int64_t ln = begin + threadIdx.x; ln < end; ln += kBlock) {
        const int i = a.lid[ln];
        int row = a.isoiext[i];
        if (row < 0)
            continue;
        if (a.add)
            row = 0;
        const double v = a.lwn[ln];
        if (v < a.own0 || v > a.own_last)
            continue;
        const int64_t li = (int64_t)layer * a.niso + i;
        const double k = line_strength(a.isoratio[i], a.gf[ln], a.elow[ln], v, temp, inv_temp,
                                       a.li_z[li], a.li_invz[li]);
        if (row != cur_row) {
            if (cur_row >= 0)
                atomicMax(&s_max[cur_row], (unsigned long long)__double_as_longlong(cur_max));
            cur_row = row;
            cur_max = 0.0;
        }
        cur_max = fmax(cur_max, k);
    }
    if (cur_row >= 0)
        atomicMax(&s_max[cur_row], (unsigned long long)__double_as_longlong(cur_max));
    __syncthreads();
    for (int r = threadIdx.x; r < a.nrows; r += kBlock)
        if (s_max[r] != 0ull)
            atomicMax(&a.kmax_bits[(int64_t)layer * a.nrows + r], s_max[r]);
}

// ---------------------------------------------------------------------------
// helpers for the gather kernels
// ---------------------------------------------------------------------------
__device__ inline double bcast(double v, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// Window of one group on the dynamic grid, exactly as _extcoeff.c:274-299.
struct Window {
    long minj, maxj;
    int half, cell;
};

// a / d truncated toward zero like C's integer division, for |a| < 2^31, 0 < d < 2^20, inv = 1.0/d
__device__ inline int trunc_div_inv(int a, double inv)
{
    return a >= 0 ? floor_div_inv(a, inv) : -floor_div_inv(-a, inv);
}

// cutsteps = cutoff / dwnstep and inv_ofactor = 1.0 / ofactor are per-layer values prepared by
// k_layer_state (the same quotients the reference forms per line, _extcoeff.c:281-299)
__device__ inline Window group_window(const LblArgs &a, double wavn, int iown, int ilor,
                                      double alphad, int ofactor, double dwnstep,
                                      int64_t dnwn, int idop_lo, int idop_hi,
                                      const double *doppler, double cutsteps,
                                      double inv_ofactor, bool clip_lo = true,
                                      bool clip_hi = true)
{
    // [idop_lo, idop_hi] brackets the answer (nearest index is monotonic in wavn), which
    // turns the bisection over the whole Doppler grid into 0-2 steps; `doppler` may point
    // to an LDS copy of the grid
    Window w;
    const int idwn = (int)((wavn - a.own0) / dwnstep);
    const int idop = idop_lo == idop_hi
                         ? idop_lo
                         : pb::nearest_index(doppler ? doppler : a.doppler, alphad * wavn,
                                             idop_lo, idop_hi);
    w.cell = ilor * a.ndop + idop;
    w.half = a.psize[w.cell];
    const int subw = iown - idwn * ofactor;
    w.minj = idwn - trunc_div_inv(w.half - subw, inv_ofactor);
    w.maxj = idwn + trunc_div_inv(w.half + subw, inv_ofactor);
    // (the packed records of the staged gathers keep a window that leaves the grid unclipped:
    // see k_records)
    if (clip_lo && w.minj < 0)
        w.minj = 0;
    if (clip_hi && w.maxj > dnwn)
        w.maxj = dnwn;
    if (a.cutoff > 0.0) {
        const int mincut = (int)(idwn - cutsteps);
        const int maxcut = (int)(idwn + cutsteps);
        if (mincut > w.minj)
            w.minj = mincut;
        if (maxcut < w.maxj)
            w.maxj = maxcut;
    }
    return w;
}

// Co-added strength of a group (left-to-right sum of its members, _extcoeff.c:248-262)
__device__ inline double group_strength(const LblArgs &a, int first, int count, double ratio,
                                        double temp, double inv_temp, double z, double inv_z)
{
    double k = line_strength(ratio, a.gf[first], a.elow[first], a.lwn[first], temp, inv_temp, z,
                             inv_z);
    for (int m = 1; m < count; m++)
        k += line_strength(ratio, a.gf[first + m], a.elow[first + m], a.lwn[first + m],
                           temp, inv_temp, z, inv_z);
    return k;
}

__device__ inline void decode_block(const LblArgs &a, int &tile, int &layer)
{
    // blocks b and b+8 share an XCD: give each XCD its own layers, deepest first
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int k = id >> 3;
    tile = k % a.ntiles;
    const int grp = k / a.ntiles;
    // dealt to the XCDs in snake order (0..7, 7..0, ...): the cost of a layer falls with height,
    // and workgroup i always runs on XCD i % 8 -- dealt 0..7 every time, XCD 0 gets the heaviest
    // layer of every group of eight and finishes last (C3: 44.0 ms of work against 38.4 on XCD 6)
    const int rank = grp * 8 + ((grp & 1) ? 7 - xcd : xcd);
    layer = a.nlayers - 1 - rank;      // < 0 for the padding blocks
}

// ---------------------------------------------------------------------------
// 3a. gather, constant-step output grid (resample mode)
//
// Record of one candidate group (16 B in LDS): strength k, 32-bit table offset relative
// to a per-(workgroup, isotope) base pointer, window [lo, hi) packed as two 16-bit tile
// coordinates.  A lane owns the two consecutive samples c0 + 2*lane + {0,1} of each
// 128-sample chunk, so its byte offset from a record's table pointer is a per-lane
// CONSTANT: the load is `global_load_dwordx4 v, v_lane_off, s[tab]` with no per-record
// address arithmetic.  Lanes outside a record's window read whatever lies there (the
// table is padded by kTile samples on both sides, all finite); whether a chunk is
// fully inside (fma with the scalar k), partial (k or 0 selected per sample) or outside
// (skipped) is decided by scalar compares.
// ---------------------------------------------------------------------------
// RS wavefronts share one 256-sample range and split its records between them (each walks
// the records of every RS-th 64-record round); their partial sums are added in wavefront
// order through LDS.  RS > 1 shortens the critical path of a workgroup RS-fold -- used
// when the launch has too few workgroups to fill the chip (multi-GPU shards).
template <int RS, int NW>
__global__ __launch_bounds__(NW * 64) void k_ext_resample(LblArgs a)
{
    constexpr int NT = NW * 64;                  // threads = records per batch
    constexpr int kTileRS = (NW / RS) * kWaveSpan;   // output samples per workgroup
    static_assert(NW % RS == 0, "wavefronts must divide evenly into record shares");
    __shared__ double s_k[NT];
    __shared__ unsigned s_off[NT];
    __shared__ unsigned s_win[NT];               // lo | hi << 16

    int tile, layer;
    decode_block(a, tile, layer);
    if (layer < 0 || (a.res_cap > 0 && a.ls_resident[layer]))
        return;
    // launched beside the staged kernel (uneven line density): only the tiles it leaves out
    if (a.tsplit && a.tsplit[((int64_t)tile * kTileRS) / a.ts_tile] != 0)
        return;
    const int row = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const int64_t t0 = a.wbegin + (int64_t)tile * kTileRS;         // global sample index
    const int64_t tend = min(t0 + kTileRS, a.wbegin + a.wcount);
    const int part = wave % RS;                                    // my share of the records
    const int rlo = (wave / RS) * kWaveSpan;                       // tile coordinates
    const int64_t recbase = (int64_t)layer * a.ngroups;
    const int rhi = (int)min((int64_t)rlo + kWaveSpan, tend - t0);

    const int ofactor = a.ls_ofactor[layer];
    const double kthresh =
        a.ethresh * __longlong_as_double((long long)a.kmax_bits[(int64_t)layer * a.nrows + row]);
    const int osamp = a.osamp;

    double acc[kChunks][kLaneSamples];
#pragma unroll
    for (int s = 0; s < kChunks; s++)
        acc[s][0] = acc[s][1] = 0.0;

    for (int iso = 0; iso < a.niso; iso++) {
        const int iext = a.isoiext[iso];
        if (iext < 0 || (a.add ? 0 : iext) != row)
            continue;
        const int64_t li = (int64_t)layer * a.niso + iso;
        const int ilor = a.li_ilor[li];
        const double alphad = a.li_alphad[li];
        const double dens = a.li_dens[li];
        int64_t reach = a.li_hmax[li];
        if (a.cutoff > 0.0)
            reach = min(reach, (int64_t)(a.cutoff / a.ownstep) + 2 * (int64_t)ofactor + 2);
        reach += osamp + ofactor;
        // groups whose window can touch [t0, tend)
        const int64_t seg0 = a.iso_gstart[iso], seg1 = a.iso_gstart[iso + 1];
        const int64_t g0 = lower_bound_i32(a.giown, seg0, seg1, t0 * osamp - reach);
        const int64_t g1 = lower_bound_i32(a.giown, seg0, seg1, (tend - 1) * osamp + reach + 1);
        if (g0 >= g1)
            continue;
        // Doppler-grid indices that the candidates can take: the nearest-index map is
        // monotonic in the line position, so it is bracketed by the images of the first
        // and last candidate positions (a leader lies within one fine step of own[iown]).
        const double vmin = a.own0 + ((double)a.giown[g0] - 1.0) * a.ownstep;
        // lowest Doppler column the candidates can use (their cells lie at or after it in
        // the table, so 32-bit offsets from its start are non-negative)
        const int idop_lo = pb::nearest_index(a.doppler, alphad * vmin, 0, a.ndop - 1);
        const int64_t cell_lo = (int64_t)ilor * a.ndop + idop_lo;
        const int64_t base_idx = a.pm_base[cell_lo] - kTile;       // inside the front pad
        const double *base = a.pm + base_idx;

        for (int64_t gb = g0; gb < g1; gb += NT) {
            __syncthreads();
            // ---- one record per lane: (layer, group) records from k_records ----
            {
                const int64_t g = gb + threadIdx.x;
                double k = 0.0;
                unsigned off = 0, win = 0;
                if (g < g1) {
                    int ulo, uhi, cell, phi, q;
                    if (a.tsplit) {
                        // the staged kernel's packed records (phase order): window end, row
                        // offset and phase follow from the cell's half-width and the position
                        const Rec16 r = a.rec16[(int64_t)layer * a.rec_pitch + a.pos2ph[g]];
                        k = r.k;
                        ulo = r.ulo;
                        uhi = ulo + (int)(r.lc & 0xfffu);
                        cell = (int)(r.lc >> 12);
                        const int d = a.psize[cell] - a.giown[g];          // half - iown
                        q = floor_div_inv(d, a.inv_osamp);
                        phi = d - q * osamp;
                    } else {
                        const int64_t idx = recbase + g;
                        k = a.rec_k[idx];
                        ulo = a.rec_ulo[idx];
                        uhi = a.rec_uhi[idx];
                        cell = a.rec_cell[idx];
                        phi = a.rec_phi[idx];
                        q = a.rec_q[idx];
                    }
                    const int64_t lo = max((int64_t)ulo, t0);
                    const int64_t hi = min((int64_t)uhi, tend);
                    if (!(k < kthresh) && lo < hi) {
                        if (a.add)
                            k *= dens;
                        win = (unsigned)(lo - t0) | ((unsigned)(hi - t0) << 16);
                        // tile sample j reads base[off + j]
                        off = (unsigned)(a.pm_base[cell] + (int64_t)phi * a.pm_stride[cell] + q +
                                         t0 - base_idx);
                        if (a.experiment == 1)
                            off = (unsigned)(kTile + (lo - t0));
                    } else {
                        k = 0.0;
                    }
                }
                s_k[threadIdx.x] = k;
                s_off[threadIdx.x] = off;
                s_win[threadIdx.x] = win;
            }
            __syncthreads();
            // ---- every wavefront walks the records that reach its samples ----
            const int nrec = a.experiment == 2 ? 0 : (int)min((int64_t)NT, g1 - gb);
            for (int b = 0; b < nrec; b += 64) {
                const int e = (b + lane) & (NT - 1);
                const unsigned my_win = s_win[e];
                // record g belongs to wavefront share (g / 64) % RS: a property of the
                // group, not of the tiling, so any tiling adds the same partial sums
                const bool mine = RS == 1 || (int)(((gb + b + lane) >> 6) & (RS - 1)) == part;
                const bool hit = mine && b + lane < nrec && (int)(my_win & 0xffff) < rhi &&
                                 (int)(my_win >> 16) > rlo;
                unsigned long long mask = __ballot(hit);
                const double my_k = s_k[e];
                const unsigned my_off = s_off[e];
                // kRecs records per trip: all loads are issued before the first fma
                while (mask) {
                    int src[kRecs];
#pragma unroll
                    for (int r = 0; r < kRecs; r++) {
                        src[r] = mask ? __builtin_amdgcn_readfirstlane(__builtin_ctzll(mask))
                                      : -1;
                        mask &= mask - 1;        // no-op once mask is 0
                    }
                    double2 v[kRecs][kChunks];
                    double kk[kRecs];
                    int lo[kRecs], hi[kRecs];
#pragma unroll
                    for (int r = 0; r < kRecs; r++) {
                        const int sl = src[r] < 0 ? src[0] : src[r];
                        kk[r] = bcast(my_k, sl);
                        const unsigned win = (unsigned)__builtin_amdgcn_readlane((int)my_win, sl);
                        const unsigned off = (unsigned)__builtin_amdgcn_readlane((int)my_off, sl);
                        lo[r] = (int)(win & 0xffff);
                        hi[r] = (int)(win >> 16);
                        if (src[r] < 0)
                            kk[r] = 0.0;                                // padding slot
                        const double *tab = base + off;
#pragma unroll
                        for (int s = 0; s < kChunks; s++) {
                            // pair address clamped to [lo-1, hi-1]: in-window samples stay
                            // in their own slot, lanes outside re-read an edge line
                            const int j = rlo + s * kChunk + 2 * lane;
                            const int jc = min(max(j, lo[r] - 1), hi[r] - 1);
                            v[r][s] = *reinterpret_cast<const double2 *>(tab + jc);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < kRecs; r++) {
                        const unsigned span = (unsigned)(hi[r] - lo[r]);
#pragma unroll
                        for (int s = 0; s < kChunks; s++) {
                            const int j = rlo + s * kChunk + 2 * lane;
                            acc[s][0] = fma((unsigned)(j - lo[r]) < span ? kk[r] : 0.0,
                                            v[r][s].x, acc[s][0]);
                            acc[s][1] = fma((unsigned)(j + 1 - lo[r]) < span ? kk[r] : 0.0,
                                            v[r][s].y, acc[s][1]);
                        }
                    }
                }
            }
        }
    }

    if (RS > 1) {
        // partial sums of the wavefronts that share a range, added in wavefront order
        __shared__ double s_acc[NT * kChunks * kLaneSamples];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < kChunks; s++) {
            s_acc[(threadIdx.x * kChunks + s) * kLaneSamples + 0] = acc[s][0];
            s_acc[(threadIdx.x * kChunks + s) * kLaneSamples + 1] = acc[s][1];
        }
        __syncthreads();
        if (part != 0)
            return;
#pragma unroll
        for (int s = 0; s < kChunks; s++) {
            for (int q = 1; q < RS; q++) {
                const int t = threadIdx.x + q * 64;
                acc[s][0] += s_acc[(t * kChunks + s) * kLaneSamples + 0];
                acc[s][1] += s_acc[(t * kChunks + s) * kLaneSamples + 1];
            }
        }
    }
    double *dst = a.ext + ((int64_t)layer * a.nrows + row) * a.wcount + (t0 - a.wbegin);
#pragma unroll
    for (int s = 0; s < kChunks; s++) {
        const int j = rlo + s * kChunk + 2 * lane;
        if (j < rhi)
            dst[j] = acc[s][0];
        if (j + 1 < rhi)
            dst[j + 1] = acc[s][1];
    }
}

// ---------------------------------------------------------------------------
// 2'. Records for the gather kernels: everything about a (layer, group) pair that does
// not depend on the output tile -- co-added strength, table cell and phase row, window on
// the global grid -- is computed ONCE here (coalesced, no workgroup synchronisation) and
// streamed by the gather kernel; the per-row maximum strength (k_kmax) is fused in.
// ---------------------------------------------------------------------------
constexpr int kLongLenBits = 14;   // window length in a long-row record (rows of up to 16 x 1024 samples)
constexpr int kChunkRow = 1024;      // samples per chunk of a long phase row (= kStageRowMax)
constexpr int kRecLayers = 4;        // layers per thread of k_records (group data loaded once);
                                     // 1 for launches of few layers (multi-GPU ranks)

// kFmt = where the records of the layers walked in phase order go: 0 the six SoA arrays, 1 the
// packed 16-byte records, 2 packed records per (group, chunk of a long row); 3 = every layer
// in position order into 32-byte records (scatter kernel).  Layers of the resident-profile
// kernel are in position order and always use the SoA arrays.  The two orders are two passes
// with their own pointer sets (one body instantiated twice): with both sets and every record
// format live at once the kernel held a third of its scalar state in spilled registers.
// A wave-uniform element of a device array written by an EARLIER kernel: read through the constant
// address space, i.e. by a scalar load (a plain load of a uniform address is a vector load that
// every lane waits for).
template <class T>
__device__ __forceinline__ T uniform_load(const T *p, int64_t i)
{
    typedef const T __attribute__((address_space(4))) *cptr;
    return ((cptr)(unsigned long long)p)[i];
}

template <int kFmt, int kRecLayers>
__global__ __launch_bounds__(kBlock) void k_records(LblArgs a)
{
    extern __shared__ unsigned long long s_max[];                 // [kRecLayers][nrows]
    double *s_dop = reinterpret_cast<double *>(s_max + kRecLayers * a.nrows);   // [ndop]
    const int layer0 = blockIdx.y * kRecLayers;
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (int r = threadIdx.x; r < kRecLayers * a.nrows; r += kBlock)
        s_max[r] = 0ull;
    for (int d = threadIdx.x; d < a.ndop; d += kBlock)
        s_dop[d] = a.doppler[d];
    // the state of this block's (layer, isotope) pairs, once into LDS: the per-record reads of
    // it were ~12 same-address global loads per layer in a dependent chain
    double *s_alphad = s_dop + a.ndop;                            // [kRecLayers][niso]
    double *s_z = s_alphad + kRecLayers * a.niso;
    double *s_invz = s_z + kRecLayers * a.niso;
    double *s_ratio = s_invz + kRecLayers * a.niso;               // [niso]
    int *s_ilor = reinterpret_cast<int *>(s_ratio + a.niso);      // [kRecLayers][niso]
    int *s_iext = s_ilor + kRecLayers * a.niso;                   // [niso]
    for (int e = threadIdx.x; e < kRecLayers * a.niso; e += kBlock) {
        const int layer = layer0 + e / a.niso;
        if (layer < a.nlayers) {
            const int64_t li = (int64_t)layer * a.niso + e % a.niso;
            s_alphad[e] = a.li_alphad[li];
            s_z[e] = a.li_z[li];
            s_invz[e] = a.li_invz[li];
            s_ilor[e] = a.li_ilor[li];
        }
    }
    for (int e = threadIdx.x; e < a.niso; e += kBlock) {
        s_ratio[e] = a.isoratio[e];
        s_iext[e] = a.isoiext[e];
    }
    int32_t *s_wm = reinterpret_cast<int32_t *>(s_iext + a.niso);  // [wm_n[0] + 1] run offsets
    if (a.wm_off[0] && a.wm_lds)
        for (int e = threadIdx.x; e <= a.wm_n[0]; e += kBlock)
            s_wm[e] = a.wm_off[0][e];
    __syncthreads();
    auto pass = [&](auto posc) {
        constexpr bool kPos = decltype(posc)::value;
        // the group's static data, in the order this pass's gather kernel walks the groups
        int iso = 0, first = 0, count = 0, iown = 0;
        double wavn = 0.0, elow = 0.0, gf = 0.0;
        // A wavenumber shard needs the records of the groups within reach of it only
        // [rec_flo, rec_fhi]; the others still count for the per-row maximum unless the caller
        // all-reduces the maxima of the shards (kmax_local: they are skipped altogether).
        int64_t g = kPos ? t : t + a.grp_lo;                    // (chunked calls: no kPos pass)
        if (a.wm_off[kPos ? 1 : 0]) {
            // run of the window map that holds thread t (the offsets of the phase-order map are
            // in LDS; the position-order map has one run per isotope)
            const int m = kPos ? 1 : 0;
            g = a.ngroups;
            if (t < a.wm_total[m]) {
                const int32_t *off = (kPos || !a.wm_lds) ? a.wm_off[m] : s_wm;
                int lo = 0, hi = a.wm_n[m];                    // last run with off[run] <= t
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (off[mid] <= t)
                        lo = mid;
                    else
                        hi = mid;
                }
                g = (int64_t)a.wm_lo[m][lo] + (t - off[lo]);
            }
        }
        bool have = g < (kPos ? a.ngroups : a.grp_hi), inwin = false;
        if (have) {
            iown = (kPos ? a.giown : a.rk_iown)[g];
            inwin = iown >= a.rec_flo && iown <= a.rec_fhi;
            have = inwin || !a.kmax_local;
        }
        if (have) {
            iso = (kPos ? a.giso : a.rk_iso)[g];
            first = (kPos ? a.gfirst : a.rk_first)[g];
            count = (kPos ? a.gcount : a.rk_count)[g];
            wavn = (kPos ? a.g_lead : a.rk_lwn)[g];                // leader's record
            elow = (kPos ? a.g_lead + a.ngroups : a.rk_elow)[g];
            gf = (kPos ? a.g_lead + 2 * a.ngroups : a.rk_gf)[g];
        }
        for (int i = 0; i < kRecLayers; i++) {
            const int layer = layer0 + i;
            if (layer >= a.nlayers)
                break;
            const bool pos = kFmt == 3 || (a.res_cap > 0 && uniform_load(a.ls_resident, layer));
            if (pos != kPos)                                          // wave-uniform
                continue;
            double k = 0.0, lmax = 0.0;
            int ulo = 0, uhi = 0, q = 0, cell = 0, phi = 0, row = -1;
            if (have) {
                row = s_iext[iso];
                if (row >= 0 && a.add)
                    row = 0;
                if (row >= 0) {
                    const int e = i * a.niso + iso;
                    const double temp = uniform_load(a.temp, layer);
                    const double inv_temp = uniform_load(a.ls_inv_temp, layer);
                    const double ratio = s_ratio[iso];
                    const double z = s_z[e], inv_z = s_invz[e];
                    k = line_strength(ratio, gf, elow, wavn, temp, inv_temp, z, inv_z);
                    lmax = k;
                    for (int m = 1; m < count; m++) {
                        const double kp = line_strength(ratio, a.gf[first + m],
                                                        a.elow[first + m], a.lwn[first + m],
                                                        temp, inv_temp, z, inv_z);
                        k += kp;
                        lmax = fmax(lmax, kp);
                    }
                    if (!inwin) {
                        // outside the shard's reach: the strength for the maximum, no record
                        atomicMax(&s_max[i * a.nrows + row],
                                  (unsigned long long)__double_as_longlong(lmax));
                        continue;
                    }
                    const int ofactor = uniform_load(a.ls_ofactor, layer);
                    const int64_t dnwn = uniform_load(a.ls_dnwn, layer);
                    const double inv_scale = uniform_load(a.ls_inv_scale, layer);
                    // The packed records of the staged gathers keep the window of a group that
                    // leaves the grid UNCLIPPED: their consumers clamp every window to the tile
                    // (hence to the grid) anyway, and the samples below 0 / at or beyond nwave
                    // that the reference's clips `minj = 0`, `maxj = dnwn` remove do not exist
                    // (the upper one only while ceil(dnwn / scale) reaches nwave: checked).
                    // Clipped, every such group had a row window of its own -- one staged row,
                    // one barrier step per RECORD: the first and the last tile of a layer ran
                    // twice as long as the others and ended the launch (904 of 1003 us at C2).
                    constexpr bool kPacked = (kFmt == 1 || kFmt == 2) && !kPos;
                    const bool hi_free =
                        kPacked && -floor_div_inv(-(int)dnwn, inv_scale) >= a.nwave;
                    const Window w = group_window(a, wavn, iown, s_ilor[e], s_alphad[e],
                                                  ofactor, uniform_load(a.ls_dwnstep, layer), dnwn,
                                                  0, a.ndop - 1, s_dop, uniform_load(a.ls_cutsteps, layer),
                                                  uniform_load(a.ls_inv_ofactor, layer), !kPacked,
                                                  !hi_free);
                    // kept samples: minj <= scale*jo < maxj, inside the profile and the grid
                    ulo = -floor_div_inv(-(int)w.minj, inv_scale);
                    uhi = -floor_div_inv(-(int)w.maxj, inv_scale);
                    ulo = max(ulo, -floor_div_inv(w.half - iown, a.inv_osamp));
                    uhi = min(uhi, floor_div_inv(iown + w.half, a.inv_osamp) + 1);
                    if (!hi_free)
                        uhi = min(uhi, a.nwave);
                    q = floor_div_inv(w.half - iown, a.inv_osamp);
                    phi = (w.half - iown) - q * a.osamp;
                    cell = w.cell;
                    if (uhi < ulo)
                        uhi = ulo;
                }
                const int64_t idx = (kFmt == 1 || kFmt == 2) && !kPos
                                        ? (int64_t)layer * a.rec_pitch + (g - a.grp_lo)
                                        : (int64_t)layer * a.ngroups + g;
                if constexpr (kFmt == 3) {
                    Rec32 r;
                    r.k = k;
                    r.off = a.pm_base[cell] + (long long)phi * a.pm_stride[cell] + q;
                    r.ulo = ulo;
                    r.uhi = uhi;
                    r.pad[0] = r.pad[1] = 0;
                    a.rec32[idx] = r;
                } else if constexpr (kFmt == 2 && !kPos) {
                    // long rows (staged in chunks of kChunkRow samples): ONE record per group with
                    // its whole window (14-bit length); the gather clips it to the chunk it
                    // stages.  (Round 2 wrote one record per (group, chunk): 12.8 GB per C3
                    // spectrum, 77 GB per C4 spectrum, written here and streamed back by the gather.)
                    Rec16 r;
                    r.k = k;
                    r.ulo = ulo;
                    r.lc = (uint32_t)(uhi - ulo) | ((uint32_t)cell << kLongLenBits);
                    a.rec16[idx] = r;
                } else if constexpr (kFmt == 1 && !kPos) {
                    Rec16 r;
                    r.k = k;
                    r.ulo = ulo;
                    r.lc = (uint32_t)(uhi - ulo) | ((uint32_t)cell << 12);
                    a.rec16[idx] = r;
                } else {
                    a.rec_k[idx] = k;
                    a.rec_ulo[idx] = ulo;
                    a.rec_uhi[idx] = uhi;
                    a.rec_q[idx] = q;
                    a.rec_cell[idx] = cell;
                    a.rec_phi[idx] = phi;
                }
            }
            if (row >= 0)
                atomicMax(&s_max[i * a.nrows + row],
                          (unsigned long long)__double_as_longlong(lmax));
        }
    };
    if (kFmt != 3)
        pass(std::false_type());
    if (kFmt == 3 || a.res_cap > 0)
        pass(std::true_type());
    __syncthreads();
    for (int r = threadIdx.x; r < kRecLayers * a.nrows; r += kBlock) {
        const int layer = layer0 + r / a.nrows;
        if (layer < a.nlayers && s_max[r] != 0ull)
            atomicMax(&a.kmax_bits[(int64_t)layer * a.nrows + r % a.nrows], s_max[r]);
    }
}

// ---------------------------------------------------------------------------
// 3a'. LDS-staged gather (constant-step grid), for line lists dense enough that several
// lines of a tile share one phase row of the Voigt table.
//
// Groups are visited in (isotope, phase key = iown mod osamp, iown) order.  All records
// with one (cell, phase, row window) key form a SEGMENT; the phase row is copied once
// into LDS -- zero outside the window, zero pads of kStagePad samples on both sides --
// and every record of the segment then adds k * row[j + q] to the samples it reaches
// with `ds_read_b64` + `v_fma_f64` and NO per-lane predicate: lanes beyond the window
// read zeros.  Compared with the global gather this moves each table sample through the
// texture path once per (tile, layer) instead of once per line, reads the operands at
// the LDS rate and executes ~1 VALU instruction per 64 samples.
// A lane owns the samples rlo + 64*c + lane (c < 4*G); a wavefront tests a record's
// window against groups of 4 chunks with scalar compares.
// ---------------------------------------------------------------------------

// NW wavefronts per workgroup, S sub-tiles of NW*256 samples each (tile = S*NW*256); one
// record per thread per batch.
// kProbe (timing experiments, results INVALID, never selected unless PB_STAGE_PROBE is set):
// 1 = "contiguous ownership": a lane owns 4 consecutive samples of a span and a visit is two
//     16-byte LDS reads at the 16-byte-aligned address at or below the one it needs (what a
//     second, 8-byte-shifted image of the row would make legal) instead of four 8-byte reads;
// 2 = the same with 2 consecutive samples per lane in each half of the span;
// 3 = (valid results) the remainder records of a visit loop software-pipelined;
// 4 = (valid results) the CYCLE ACCOUNT of a segment step: every
//     wavefront stamps s_memtime around the parts of its work and adds the differences, per layer,
//     to a.probe[layer * 24 + c]:  c = 0 candidate search and scans of an isotope, 1 record
//     fetch + decode of a batch (with its barriers), 2 segment detection + segment table (three
//     barriers), 3 find_hits (two bisections for 64 segments at once), 4 issue of a row's LDS-DMA
//     (the one wavefront whose turn it is), 5 the walk (hit decode, record broadcasts, row reads,
//     FMAs -- up to the last LDS value consumed), 6 s_waitcnt vmcnt(0) for the DMA, 7 the barrier
//     that ends the step, 8 pipeline fill of a batch (first row + barrier), 9 the wavefront's
//     lifetime, 10 segment steps, 11 batches, 12 (record, sub-tile) visits of this wavefront,
//     13 steps in which it visited anything, 14 accumulator read / write-back; calibration:
//     15 the lifetime on the 100-MHz wall clock (s_memrealtime), 16 two back-to-back stamp
//     intervals with an empty LDS queue, 17 wavefronts, 18 the interval of category 4 in the
//     wavefronts that did NOT request a row (a stamp right after a barrier release), 19 rows
//     requested, 20 the part of category 4 up to the first load (descriptor read + decode).
//     A stamp is s_memtime + s_waitcnt lgkmcnt(0): it also drains the LDS queue, which is why
//     the probe is not the product kernel (profiles/r05_step_account.md gives both times).
// kIssue (kDma only): who requests a row's LDS-DMA.  0 = ONE wavefront per segment, the wavefronts
// taking turns, descriptor read and decoded at the time of the request (rounds 2-4); 1 = the same
// with the wavefront's NEXT descriptor fetched from LDS one turn (8 steps) ahead; 2 = EVERY
// wavefront requests its own 128-sample slice of the row (one load each), from a descriptor it
// fetched one step ahead.  Why: the cycle account (profiles/r05_step_account.md) shows the
// requesting wavefront ~700 cycles behind its peers in every step, who wait for it at the barrier.
template <int NW, int S, bool kDma, int kProbe = 0, int kIssue = 0>
__global__ __launch_bounds__(NW * 64, 8) void k_ext_staged(LblArgs a)
{
    constexpr int kThreads = NW * 64;
    constexpr int kSub = NW * kStageSpan;         // samples per sub-tile
    constexpr int kT = S * kSub;                  // samples per workgroup
    constexpr int kRowRegs = (kStageRowMax + kThreads - 1) / kThreads;
    static_assert(kT < 65536, "window coordinates are packed in 16 bits");
    extern __shared__ __align__(16) unsigned char smem[];
    const int osamp = a.osamp;
    // row buffers: [pad][row 0][pad][row 1][pad], pads of kStagePad zero samples shared
    double *s_row = reinterpret_cast<double *>(smem);
    struct __align__(16) Rec {
        double k;
        int qoff;                 // byte offset of tile sample 0 from the row buffer start
        unsigned win;             // lo | hi << 16 (tile coordinates)
    };
    const int rowspan = a.rowlds + kStagePad;                            // buffer pitch
    constexpr int kNB = 2;                        // row buffers
    Rec *s_rec = reinterpret_cast<Rec *>(s_row + kNB * rowspan + kStagePad);   // [kThreads]
    long long *s_src = reinterpret_cast<long long *>(s_rec + kThreads);  // row start, -1 empty
    unsigned long long *s_desc = reinterpret_cast<unsigned long long *>(s_src);   // per segment
    unsigned long long *s_segmask =
        reinterpret_cast<unsigned long long *>(s_src + kThreads);        // [NW]
    unsigned *s_m = reinterpret_cast<unsigned *>(s_segmask + NW);        // mlo | mhi << 16
    // the segment table takes the place of the window keys, which nothing reads once the
    // segment starts are known (a barrier lies between the last read and the first write)
    unsigned *s_seg = s_m;                                               // i0 | i1 << 16
    // the table cells of this (layer, isotope): one Lorentz row, ndop Doppler columns
    long long *s_cbase = reinterpret_cast<long long *>(s_m + kThreads);  // [ndop] pm_base
    int *s_csize = reinterpret_cast<int *>(s_cbase + a.ndop);            // [ndop] psize
    int *s_cstride = s_csize + a.ndop;                                   // [ndop] pm_stride
    int *s_part = s_cstride + a.ndop;                                    // [NW] scan scratch
    const int vmax = osamp * a.nch_max;                                  // (phase, chunk) pairs
    int *s_cum = s_part + NW;                                            // [vmax+1]
    int *s_phs = s_cum + (vmax + 1);                                     // [vmax]

    // Blocks b and b+8 share an XCD.  The unit handed to an XCD is a (layer, phase split)
    // pair, deepest layer first: the splits of one layer read different phase rows, so
    // spreading them over the XCDs costs no L2 sharing, and a launch of few layers (a
    // multi-GPU rank: 10 layers on 8 XCDs) still loads every XCD alike.
    int tile, layer, zsplit;
    {
        const int id = blockIdx.x;
        const int k = id >> 3;
        tile = k % a.ntiles;
        const int grp = k / a.ntiles;                  // snake order over the XCDs: decode_block
        const int unit = grp * 8 + ((grp & 1) ? 7 - (id & 7) : (id & 7));
        if (a.unit_tab) {
            // per-layer split: the unit table lists the (layer, piece) pairs, deepest layer first
            const int e = unit < a.nunits ? a.unit_tab[unit] : -1;
            layer = e < 0 ? -1 : e >> 8;
            zsplit = e & 0xff;
        } else {
            layer = a.nlayers - 1 - unit / a.nsplit;   // < 0 for the padding blocks
            zsplit = unit % a.nsplit;
        }
    }
    if (layer < 0 || (a.res_cap > 0 && a.ls_resident[layer]) || (a.wave_cap > 0 && a.ls_wave[layer]))
        return;
    // pieces of this tile: per layer, per tile (uneven line density) or one number for the launch
    const int nsp = a.tsplit ? a.tsplit[tile] : a.lsplit ? a.lsplit[layer] : a.nsplit;
    if (zsplit >= nsp)
        return;
    const int row = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long pc[21] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&]() -> unsigned long long {
        if constexpr (kProbe == 4)
            return (unsigned long long)__builtin_amdgcn_s_memtime();
        return 0ull;
    };
    unsigned long long real_birth = 0;
    if constexpr (kProbe == 4) {
        // calibration: the cost of a stamp with an empty LDS queue (16 = three stamps in a row,
        // i.e. two intervals; 17 = 1) and the 100-MHz wall clock beside s_memtime (15)
        real_birth = (unsigned long long)__builtin_amdgcn_s_memrealtime();
        const unsigned long long c0 = tick();
        const unsigned long long c1 = tick();
        const unsigned long long c2 = tick();
        pc[16] = (c2 - c0) + 0 * c1;
        pc[17] = 1;
    }
    const unsigned long long t_birth = tick();

    const int64_t t0 = a.wbegin + (int64_t)tile * kT;
    const int64_t tend = min(t0 + kT, a.wbegin + a.wcount);
    const int tlen = (int)(tend - t0);
    const int rlo = wave * kStageSpan;            // first sample of sub-tile 0 (tile coords)

    const double kthresh =
        a.ethresh * __longlong_as_double((long long)a.kmax_bits[(int64_t)layer * a.nrows + row]);
    // packed records of this layer, indexed by the position in the phase-sorted group list
    const int64_t recbase = a.rec16 ? (int64_t)layer * a.rec_pitch - a.grp_lo
                                    : (int64_t)layer * a.ngroups;
    double *const out = zsplit == 0
                            ? a.ext
                            : a.part + (int64_t)(zsplit - 1) * a.nlayers * a.nrows * a.wcount;
    double *const dst = out + ((int64_t)layer * a.nrows + row) * a.wcount + (t0 - a.wbegin);

    double acc[S][4];
#pragma unroll
    for (int u = 0; u < S; u++)
        acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.0;
    if (a.accumulate) {
        // a later chunk of the line list: go on from the running sums of the earlier ones
#pragma unroll
        for (int u = 0; u < S; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int j = rlo + u * kSub + c * 64 + lane;
                if (j < tlen)
                    acc[u][c] = dst[j];
            }
    }
    for (int i = tid; i < kNB * rowspan + kStagePad; i += kThreads)
        s_row[i] = 0.0;                            // the pads stay zero for good

    // Row of one segment -> registers (issued early), registers -> LDS (after the walk).
    // The registers form a ring of D sets of R doubles per lane (D*R = 2*kRowRegs): rows
    // that fit one register per lane (<= kThreads samples) are fetched 4 segments ahead.
    double ring[kDma ? 1 : 2 * kRowRegs];
    // Rows of this (layer, isotope) are at most `rowlim` samples long: wavefronts whose
    // lanes lie beyond it do not store (the buffers are zeroed per isotope).
    int rowlim = a.rowlds;
    // kDma: the row of a segment goes global -> LDS directly (`buffer_load_dwordx4 ... lds`,
    // 1 KiB = 128 samples per wave-instruction), issued by ONE wavefront per segment (the
    // wavefronts take turns), one step ahead of the walk.  No row registers, no LDS stores
    // and one descriptor chain per workgroup and segment instead of eight.  Lanes outside
    // the window read 0 through the buffer range check (checked per dword at the upper end;
    // a lane below the window is out of range as a whole, so the LDS image is shifted by the
    // parity e of the window start -- row sample m sits at position m - e, which the
    // records' offsets account for -- and no lane straddles the lower edge).  The
    // instruction is issued from inline asm: the compiler would otherwise drain it
    // (vmcnt(0)) before the next LDS read; the wait before the barrier is explicit.
    // LDS byte address of row buffer 0 (taken once: the flat -> LDS cast of a pointer the compiler
    // cannot prove non-null costs a compare per use, and trips a register-class bug of this
    // compiler when the lambda below is instantiated for several call sites)
    const unsigned row_lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)(
        reinterpret_cast<char *>(s_row + kStagePad));
    // (tref, probe 4: the stamp that opened the step -- the descriptor chain is timed against it)
    auto dma_desc = [&](unsigned long long d, int buf, int c_lo, int c_hi,
                        unsigned long long tref = 0) {
        const unsigned dlo = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
        const unsigned dhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(d >> 32));
        const long long first = ((long long)(dhi & 0xffu) << 32) | dlo;
        const int len = (int)((dhi >> 8) & 0xfffu), mlo = (int)(dhi >> 20);
        const unsigned long long base = (unsigned long long)(a.pm + first);
        typedef int v4i __attribute__((ext_vector_type(4)));
        v4i rsrc;
        rsrc.x = __builtin_amdgcn_readfirstlane((int)base);
        rsrc.y = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffffu));
        rsrc.z = __builtin_amdgcn_readfirstlane(len * 8);
        rsrc.w = 0x00020000;
        int voff = (2 * lane - (mlo & ~1)) * 8;
        unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(row_lds0 + buf * rowspan * 8));
        if constexpr (kProbe == 4) {
            if (tref)
                pc[20] += tick() - tref;
        }
        voff += 1024 * c_lo;
        dst += 1024 * (unsigned)c_lo;
        for (int c = c_lo; c < c_hi && c * 128 < rowlim; c++) {
            unsigned keep;
            asm volatile("s_nop 4\n\t"
                         "s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %1\n\t"
                         "s_nop 0\n\t"
                         "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "s"(dst), "v"(voff), "s"(rsrc)
                         : "memory");
            voff += 1024;
            dst += 1024;
        }
    };
    auto dma_row = [&](int sg, int buf, unsigned long long tref = 0) {
        dma_desc(s_desc[sg], buf, 0, 1 << 20, tref);
    };
    // out-of-window lanes fall outside the buffer descriptor and read 0 (no predicate)
    auto load_row = [&](int sg, auto Rc, double *reg) {
        constexpr int R = decltype(Rc)::value;
        const unsigned long long d = s_desc[sg];
        const unsigned dlo = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
        const unsigned dhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(d >> 32));
        const long long first = ((long long)(dhi & 0xffu) << 32) | dlo;
        const int len = (int)((dhi >> 8) & 0xfffu), mlo = (int)(dhi >> 20);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.pm + first), 0, len * 8, 0x00020000);
        typedef int v2i __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int r = 0; r < R; r++) {
            const v2i v = __builtin_amdgcn_raw_buffer_load_b64(
                rsrc, (tid + r * kThreads - mlo) * 8, 0, 0);
            reg[r] = __hiloint2double(v.y, v.x);
        }
    };
    auto store_row = [&](int buf, auto Rc, const double *reg, auto Dc) {
        constexpr int R = decltype(Rc)::value;
        // The (D-1)*R loads issued after this set stay in flight.  The wait is explicit and
        // unconditional: left to the conditional stores below, the path that skips them
        // reaches the loop header with the set still pending and the compiler drains the
        // whole prefetch queue (vmcnt(0)) at the top of every other segment.
        constexpr int kInFlight = (decltype(Dc)::value - 1) * R;
        __builtin_amdgcn_s_waitcnt((kInFlight & 0xf) | (0x7 << 4) | (0xf << 8) |
                                   ((kInFlight >> 4) << 14));
        double *dst = s_row + kStagePad + buf * rowspan;
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (wave * 64 + r * kThreads >= rowlim)
                continue;
            const int mm = tid + r * kThreads;
            if (mm < a.rowlds)
                dst[mm] = reg[r];
        }
    };

    pc[14] += tick() - t_birth;
    for (int iso = 0; iso < a.niso; iso++) {
        const int iext = a.isoiext[iso];
        if (iext < 0 || (a.add ? 0 : iext) != row)
            continue;
        if ((iso + 1) * osamp <= a.key_lo || iso * osamp >= a.key_hi)
            continue;                              // no key of this isotope in the chunk
        const unsigned long long t_iso = tick();
        const int64_t li = (int64_t)layer * a.niso + iso;
        const double dens = a.li_dens[li];
        int64_t reach = a.li_hmax[li];
        if (a.cutoff > 0.0)
            reach = min(reach, (int64_t)(a.cutoff / a.ownstep) + 2 * (int64_t)a.ls_ofactor[layer] + 2);
        reach += osamp + a.ls_ofactor[layer];
        const int64_t flo = t0 * osamp - reach, fhi = (tend - 1) * osamp + reach;

        // candidates of every phase key: [s_phs[p], s_phs[p] + count) in the phase list,
        // then an exclusive scan of the counts (thread t owns a run of `per` phases)
        __syncthreads();
        const int cell0 = a.li_ilor[li] * a.ndop;      // first cell of the isotope's Lorentz row
        for (int d = tid; d < a.ndop; d += kThreads) {
            s_cbase[d] = a.pm_base[cell0 + d];
            s_csize[d] = a.psize[cell0 + d];
            s_cstride[d] = a.pm_stride[cell0 + d];
        }
        {
            const int lim = min(a.rowlds, a.li_rowmax[li]);
            if (lim != rowlim) {                   // wave-uniform
                for (int i = tid; i < kNB * rowspan + kStagePad; i += kThreads)
                    s_row[i] = 0.0;
                rowlim = lim;
            }
        }
        // "virtual phases": (phase p, chunk c) pairs, pv = p*nch + c; nch = 1 unless the rows of
        // this (layer, isotope) are longer than kStageRowMax
        const int nch = a.nch_max > 1
                            ? (min(a.rowcap, a.li_rowmax[li]) + kChunkRow - 1) / kChunkRow
                            : 1;
        const int nvirt = osamp * nch;
        const int per = (nvirt + kThreads - 1) / kThreads;
        int mine = 0;
        for (int r = 0; r < per; r++) {
            const int pv = tid * per + r;
            const int p = pv / nch, c = pv - p * nch;
            if (pv < nvirt && ((nsp > 1 && p * nsp / osamp != zsplit) ||
                               iso * osamp + p < a.key_lo || iso * osamp + p >= a.key_hi)) {
                s_phs[pv] = 0;                     // another workgroup's, or another chunk's, phase
                s_cum[pv] = 0;
            } else if (pv < nvirt) {
                // two table lookups bracket each bound to within one bin (a fraction of a
                // record per phase), then a short bisection makes it exact
                const int32_t *bin = a.ph_bin + ((int64_t)iso * osamp + p) * (a.ph_nbins + 1);
                const int64_t binw = (int64_t)kBinSamples * osamp;
                // a chunk of a long row covers the samples [c0 - q, c0 + 1024 - q) with
                // q = floor((half - iown)/osamp): only groups at fine positions within
                // [t0*osamp + half - (c0+1024)*osamp, tend*osamp + half - c0*osamp] (+- one
                // sample, half between the smallest and largest one of the isotope) reach the tile
                int64_t clo = flo, chi = fhi;
                if (a.nch_max > 1) {
                    const int64_t c0 = (int64_t)c * kChunkRow;
                    clo = max(clo, (t0 - c0 - kChunkRow - 1) * osamp + a.li_hlo[li]);
                    chi = min(chi, (tend - c0 + 1) * osamp + a.li_hhi[li]);
                    if (chi < clo)
                        chi = clo - 1;
                }
                const int b0 = (int)min((int64_t)a.ph_nbins - 1, max((int64_t)0, clo) / binw);
                const int b1 = (int)min((int64_t)a.ph_nbins - 1, max((int64_t)0, chi + 1) / binw);
                const int32_t l0 = bin[b0], h0 = bin[b0 + 1], l1 = bin[b1], h1 = bin[b1 + 1];
                int64_t s0, s1;
                lower_bound2_i32(a.ph_iown, l0, h0, clo, l1, h1, chi + 1, s0, s1);
                s_phs[pv] = (int)s0;               // entry of the layer's record array
                s_cum[pv] = (int)(s1 - s0);
                mine += (int)(s1 - s0);
            }
        }
        int incl = mine;                           // inclusive scan inside the wavefront
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d)
                incl += up;
        }
        if (lane == 63)
            s_part[wave] = incl;
        __syncthreads();
        int base = 0, total = 0;
        for (int w = 0; w < NW; w++) {
            const int pw = s_part[w];
            if (w < wave)
                base += pw;
            total += pw;
        }
        int run = base + incl - mine;
        for (int r = 0; r < per; r++) {
            const int pv = tid * per + r;
            if (pv < nvirt) {
                const int c = s_cum[pv];
                s_cum[pv] = run;
                run += c;
            }
        }
        if (tid == 0)
            s_cum[nvirt] = total;
        __syncthreads();
        pc[0] += tick() - t_iso;

        for (int x0 = 0; x0 < total; x0 += kThreads) {
            const int nrec = min(kThreads, total - x0);
            const unsigned long long t_batch = tick();
            pc[11] += 1;
            __syncthreads();
            // ---- one record per lane, in (phase, iown) order, from k_records ----
            long long src = -1;                        // phase row of my record (table element)
            unsigned mwin = 0;                         // its window in row coordinates
            {
                double k = 0.0;
                unsigned win = 0;
                int qoff = 0;
                const int x = x0 + tid;
                if (x < total) {
                    int plo = 0, pup = nvirt;           // largest pv with s_cum[pv] <= x
                    while (pup - plo > 1) {
                        const int mid = (plo + pup) >> 1;
                        if (s_cum[mid] <= x)
                            plo = mid;
                        else
                            pup = mid;
                    }
                    const int64_t entry = s_phs[plo] + (x - s_cum[plo]);
                    const int64_t gidx = entry;         // index in the phase-sorted group list
                    int c0 = 0;                         // first row sample of my chunk
                    if (a.nch_max > 1)
                        c0 = (plo % nch) * kChunkRow;
                    const int64_t idx = recbase + entry;
                    int ulo, uhi, q, cell, phi;
                    if (a.rec16) {
                        const Rec16 r = a.rec16[idx];
                        k = r.k;
                        ulo = r.ulo;
                        const int lbits = a.nch_max > 1 ? kLongLenBits : 12;
                        uhi = ulo + (int)(r.lc & ((1u << lbits) - 1u));
                        cell = (int)(r.lc >> lbits);
                        const int d = s_csize[cell - cell0] - a.ph_iown[gidx];   // half - iown
                        q = floor_div_inv(d, a.inv_osamp);
                        phi = d - q * osamp;
                        if (a.nch_max > 1) {
                            // the part of the window whose row coordinates u = sample + q fall
                            // into my chunk [c0, c0 + kChunkRow)
                            const int wlo = max(ulo + q, c0), whi = min(uhi + q, c0 + kChunkRow);
                            if (whi > wlo) {
                                ulo = wlo - q;
                                uhi = whi - q;
                            } else {
                                uhi = ulo;
                            }
                        }
                        q -= c0;                        // row index relative to the chunk
                    } else {
                        k = a.rec_k[idx];
                        ulo = a.rec_ulo[idx];
                        uhi = a.rec_uhi[idx];
                        q = a.rec_q[idx];
                        cell = a.rec_cell[idx];
                        phi = a.rec_phi[idx];
                    }
                    const int lo = (int)(max((int64_t)ulo, t0) - t0);
                    const int hi = (int)(min((int64_t)uhi, tend) - t0);
                    if (!(k < kthresh) && lo < hi) {
                        if (a.add)
                            k *= dens;
                        win = (unsigned)lo | ((unsigned)hi << 16);
                        qoff = (int)(q + t0) * 8;       // tile sample j reads row[j + q + t0]
                        if (kDma)
                            qoff -= ((ulo + q) & 1) * 8;  // the image starts at an even sample
                        src = s_cbase[cell - cell0] + (long long)phi * s_cstride[cell - cell0] + c0;
                        mwin = (unsigned)(ulo + q) | ((unsigned)(uhi + q) << 16);
                    } else {
                        k = 0.0;
                    }
                }
                s_rec[tid].k = k;
                s_rec[tid].qoff = qoff;
                s_rec[tid].win = win;
                s_src[tid] = src;
                s_m[tid] = mwin;
            }
            __syncthreads();
            const unsigned long long t_seg = tick();
            pc[1] += t_seg - t_batch;
            // ---- segments: runs of equal (cell, phase, row window); live ones are listed ----
            bool start = false;
            {
                if (tid < nrec)
                    start = tid == 0 || s_src[tid] != s_src[tid - 1] || s_m[tid] != s_m[tid - 1];
                const unsigned long long m = __ballot(start);
                if (lane == 0)
                    s_segmask[wave] = m;
            }
            const bool live = start && s_src[tid] >= 0;
            const unsigned long long livemask = __ballot(live);
            if (lane == 0)
                s_part[wave] = __builtin_popcountll(livemask);
            __syncthreads();
            int nseg = 0;
            {
                int before = 0;
                for (int w = 0; w < NW; w++) {
                    const int c = s_part[w];
                    if (w < wave)
                        before += c;
                    nseg += c;
                }
                if (live) {
                    // end of my segment = next start after me
                    int w = tid >> 6;
                    unsigned long long m = s_segmask[w] & ~((2ull << (tid & 63)) - 1ull);
                    int end = nrec;
                    for (;;) {
                        if (m) {
                            end = min(nrec, w * 64 + (int)__builtin_ctzll(m));
                            break;
                        }
                        if (++w >= NW)
                            break;
                        m = s_segmask[w];
                    }
                    const int pos = before + __builtin_popcountll(livemask & ((1ull << lane) - 1ull));
                    s_seg[pos] = (unsigned)tid | ((unsigned)end << 16);
                    // the row descriptor of the segment, packed: first element of the window
                    // (40 bits) | window length << 40 (12 bits) | window start << 52 (12 bits).
                    // It overwrites s_src, which nothing reads after the barrier above.
                    const unsigned long long mlo = mwin & 0xffffu, mhi = mwin >> 16;
                    s_desc[pos] = (unsigned long long)(src + (long long)mlo) | ((mhi - mlo) << 40) |
                                  (mlo << 52);
                }
            }
            __syncthreads();
            pc[2] += tick() - t_seg;
            if (nseg == 0)
                continue;
            // ---- rows are double-buffered in LDS and fetched two segments ahead: segment
            // sg+2's row is already in flight (in registers) while sg is walked, and is
            // written to the free LDS buffer one segment later ----
            // Lane l of every wavefront keeps, for segment (sg & ~63) + l, the records that
            // reach each of the wavefront's sub-tiles: first | count << 16.  The records of a
            // segment have equal window lengths and ascending positions, so those are
            // consecutive and two bisections find them; 64 segments are done at once.
            unsigned hits[S];
            auto find_hits = [&](int sg0) {
                const int sgl = sg0 + lane;
                const unsigned sd = sgl < nseg ? s_seg[sgl] : 0u;
                const int i0 = (int)(sd & 0xffff), i1 = (int)(sd >> 16);
#pragma unroll
                for (int u = 0; u < S; u++) {
                    const int slo = rlo + u * kSub;
                    int lo = i0, hi = i1;              // first record with window end > slo
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if ((int)(s_rec[mid].win >> 16) > slo)
                            hi = mid;
                        else
                            lo = mid + 1;
                    }
                    const int first = lo;
                    hi = i1;                           // first record starting at or after the end
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if ((int)(s_rec[mid].win & 0xffff) >= slo + kStageSpan)
                            hi = mid;
                        else
                            lo = mid + 1;
                    }
                    hits[u] = slo < tlen ? (unsigned)first | ((unsigned)(lo - first) << 16) : 0u;
                }
            };
            auto walk = [&](int sg, int buf) {
                if ((sg & 63) == 0) {
                    const unsigned long long tf = tick();
                    find_hits(sg);
                    pc[3] += tick() - tf;
                }
                bool any = false;
                // byte address of this lane's first sample in the staged row
                const char *rowp = reinterpret_cast<const char *>(
                    s_row + kStagePad + buf * rowspan + rlo + lane);
#pragma unroll
                for (int u = 0; u < S; u++) {
                    const unsigned h = (unsigned)__builtin_amdgcn_readlane((int)hits[u], sg & 63);
                    if (h < 0x10000u)
                        continue;
                    if constexpr (kProbe == 4) {
                        pc[12] += h >> 16;
                        any = true;
                    }
                    const int first = (int)(h & 0xffff);
                    const int last = first + (int)(h >> 16);
                    // per record: ONE 16-byte broadcast read of {k, offset, window}, four
                    // 64-sample row reads, four fma; four records per trip
                    const char *rs = rowp + (size_t)u * kSub * 8;
                    const double2 *recs = reinterpret_cast<const double2 *>(s_rec);
                    auto visit = [&](const double2 raw) {
                        const double k = raw.x;
                        const int qoff = __double2loint(raw.y);
                        if constexpr (kProbe == 1 || kProbe == 2) {
                            // 1: 4 consecutive samples per lane (lane pitch 32 B: two-way bank
                            // conflicts); 2: 2 consecutive samples in each half of the span (lane
                            // pitch 16 B, conflict-free)
                            // (offsets from the LDS base: a pointer rebuilt from an integer
                            // would be a flat pointer and the reads flat loads)
                            const char *base = reinterpret_cast<const char *>(s_row);
                            const int off = ((int)(rs - base) + (kProbe == 1 ? 3 : 1) * 8 * lane + qoff) & ~15;
                            const double2 v0 = *reinterpret_cast<const double2 *>(base + off);
                            const double2 v1 = *reinterpret_cast<const double2 *>(
                                base + off + (kProbe == 1 ? 16 : 1024));
                            acc[u][0] = fma(k, v0.x, acc[u][0]);
                            acc[u][1] = fma(k, v0.y, acc[u][1]);
                            acc[u][2] = fma(k, v1.x, acc[u][2]);
                            acc[u][3] = fma(k, v1.y, acc[u][3]);
                            return;
                        }
                        const double *p0 = reinterpret_cast<const double *>(rs + qoff);
                        const double a0 = p0[0], a1 = p0[64], a2 = p0[128], a3 = p0[192];
                        acc[u][0] = fma(k, a0, acc[u][0]);
                        acc[u][1] = fma(k, a1, acc[u][1]);
                        acc[u][2] = fma(k, a2, acc[u][2]);
                        acc[u][3] = fma(k, a3, acc[u][3]);
                    };
                    int r = first;
                    for (; r + 3 < last; r += 4) {
                        const double2 w0 = recs[r], w1 = recs[r + 1], w2 = recs[r + 2],
                                      w3 = recs[r + 3];
                        visit(w0);
                        visit(w1);
                        visit(w2);
                        visit(w3);
                    }
                    if constexpr (kProbe == 3) {
                        // remainder records software-pipelined: the next record's {k, offset} is
                        // requested before the current record's row reads are consumed
                        if (r < last) {
                            double2 cur = recs[r];
                            for (; r + 1 < last; r++) {
                                const double2 nxt = recs[r + 1];
                                visit(cur);
                                cur = nxt;
                            }
                            visit(cur);
                        }
                    } else {
                        for (; r < last; r++)
                            visit(recs[r]);
                    }
                }
                if constexpr (kProbe == 4)
                    pc[13] += any ? 1 : 0;
            };
            // Software pipeline over the segments.  Segment j lives in register set j % D;
            // its loads are issued D steps before its row is written to LDS buffer j % 2
            // (one step before it is walked).  Every step issues exactly R loads (the
            // segment index is clamped, a surplus row is loaded and dropped), so the number
            // of loads in flight behind the row about to be stored is a compile-time
            // constant and the wait before the store is `s_waitcnt vmcnt((D-1)*R)`.
            auto run = [&](auto Dc, auto Rc) {
                constexpr int D = decltype(Dc)::value, R = decltype(Rc)::value;
                static_assert(D == 2, "two register sets, two LDS buffers");
                load_row(0, Rc, ring);
                load_row(min(1, nseg - 1), Rc, ring + R);
                store_row(0, Rc, ring, Dc);
                __syncthreads();
                // Two steps per trip so that register sets and LDS buffers are compile-time
                // names, and NO exit inside the trip: the structurised CFG routes an exit
                // through the back edge, which leaves a static path on which the header is
                // entered with the youngest loads pending -- the compiler then drains the
                // queue (vmcnt(0)) at the top of every trip.  The last one or two segments
                // are walked after the loop.
                int sg = 0;
                for (; sg + 2 < nseg; sg += 2) {
                    load_row(sg + 2, Rc, ring);
                    walk(sg, 0);
                    store_row(1, Rc, ring + R, Dc);
                    __syncthreads();
                    load_row(min(sg + 3, nseg - 1), Rc, ring + R);
                    walk(sg + 1, 1);
                    store_row(0, Rc, ring, Dc);
                    __syncthreads();
                }
                walk(sg, 0);
                if (sg + 1 < nseg) {
                    store_row(1, Rc, ring + R, std::integral_constant<int, 1>());
                    __syncthreads();
                    walk(sg + 1, 1);
                }
            };
            // (a ring of 4 single-register sets for rows of <= 512 samples was measured:
            // it spills at the 64-VGPR budget of 8 waves/SIMD and runs 30 % slower)
            if (kDma) {
                // the row of segment sg+1 is requested by one wavefront (they take turns)
                // while sg is walked, and has landed before the barrier that ends the step.
                // (Three buffers and a request two steps ahead measured slower, 1.22 vs 1.14 ms
                // at C2: 48 KB of LDS leave three workgroups per CU.)
                if constexpr (kIssue == 1) {
                    // the turn-taking requester with its next descriptor fetched a turn ahead
                    unsigned long long dn = s_desc[min(nseg - 1, wave == 0 ? NW : wave)];
                    if (wave == 0)
                        dma_row(0, 0);
                    __builtin_amdgcn_s_waitcnt(0x0f70);
                    __syncthreads();
                    for (int sg = 0; sg < nseg; sg++) {
                        if (sg + 1 < nseg && wave == ((sg + 1) & (NW - 1))) {
                            dma_desc(dn, (sg + 1) & 1, 0, 1 << 20);
                            dn = s_desc[min(nseg - 1, sg + 1 + NW)];
                        }
                        walk(sg, sg & 1);
                        __builtin_amdgcn_s_waitcnt(0x0f70);
                        __syncthreads();
                    }
                    continue;
                }
                if constexpr (kIssue == 2) {
                    // every wavefront requests its own slices of the row (8 slices of 128 samples
                    // over NW wavefronts) from a descriptor it fetched one step ahead
                    constexpr int kPer = (8 + NW - 1) / NW;
                    unsigned long long dn = s_desc[min(nseg - 1, 1)];
                    dma_desc(s_desc[0], 0, wave * kPer, wave * kPer + kPer);
                    __builtin_amdgcn_s_waitcnt(0x0f70);
                    __syncthreads();
                    for (int sg = 0; sg < nseg; sg++) {
                        const unsigned long long dcur = dn;
                        dn = s_desc[min(nseg - 1, sg + 2)];
                        if (sg + 1 < nseg)
                            dma_desc(dcur, (sg + 1) & 1, wave * kPer, wave * kPer + kPer);
                        walk(sg, sg & 1);
                        __builtin_amdgcn_s_waitcnt(0x0f70);
                        __syncthreads();
                    }
                    continue;
                }
                unsigned long long ta = tick();
                if (wave == 0)
                    dma_row(0, 0);
                __builtin_amdgcn_s_waitcnt(0x0f70);           // vmcnt(0)
                __syncthreads();
                if constexpr (kProbe == 4) {
                    const unsigned long long tb = tick();
                    pc[8] += tb - ta;
                    ta = tb;
                }
                for (int sg = 0; sg < nseg; sg++) {
                    if (sg + 1 < nseg && wave == ((sg + 1) & (NW - 1)))
                        dma_row(sg + 1, (sg + 1) & 1, ta);
                    if constexpr (kProbe == 4) {
                        const unsigned long long tb = tick();
                        // (4: the wavefront whose turn it was to request the row; 18: the others,
                        // i.e. the cost of the stamp itself right after a barrier release)
                        if (sg + 1 < nseg && wave == ((sg + 1) & (NW - 1))) {
                            pc[4] += tb - ta;
                            pc[19] += 1;
                        } else {
                            pc[18] += tb - ta;
                        }
                        ta = tb;
                        const unsigned long long f0 = pc[3];
                        walk(sg, sg & 1);
                        // (every LDS value of the walk has been consumed by its FMA: lgkmcnt(0))
                        const unsigned long long tc = tick();
                        pc[5] += (tc - ta) - (pc[3] - f0);
                        __builtin_amdgcn_s_waitcnt(0x0f70);
                        const unsigned long long td = tick();
                        pc[6] += td - tc;
                        __syncthreads();
                        ta = tick();
                        pc[7] += ta - td;
                        pc[10] += 1;
                        continue;
                    }
                    walk(sg, sg & 1);
                    __builtin_amdgcn_s_waitcnt(0x0f70);       // the row of sg+1 has landed
                    __syncthreads();
                }
            } else if (nseg > 0) {
                run(std::integral_constant<int, 2>(), std::integral_constant<int, kRowRegs>());
            }
        }
    }

    const unsigned long long t_out = tick();
#pragma unroll
    for (int u = 0; u < S; u++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int j = rlo + u * kSub + c * 64 + lane;
            if (j < tlen)
                dst[j] = acc[u][c];
        }
    }
    if constexpr (kProbe == 4) {
        const unsigned long long t_end = tick();
        pc[14] += t_end - t_out;
        pc[9] = t_end - t_birth;
        pc[15] = (unsigned long long)__builtin_amdgcn_s_memrealtime() - real_birth;
        if (lane == 0 && a.probe)
            for (int i = 0; i < 21; i++)
                atomicAdd(&a.probe[(int64_t)layer * 24 + i], pc[i]);
    }
}

// ---------------------------------------------------------------------------
// Work of the last launch, counted from its packed records (bench.py's roofline.binding):
// out[0] = profile samples multiplied (sum of the live records' windows inside the shard),
// out[1] = lanes the staged kernels issue for them (every 256-sample span a window touches,
// spans aligned to the shard start), out[2] = live records.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_work_stats(LblArgs a, unsigned long long *out)
{
    const int64_t per_layer = (int64_t)a.ngroups;
    const int64_t n = (int64_t)a.nlayers * per_layer;
    const int lbits = a.nch_max > 1 ? kLongLenBits : 12;
    unsigned long long useful = 0, issued = 0, live = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kBlock) {
        const int layer = (int)(i / per_layer);
        const int64_t g = i - (int64_t)layer * a.ngroups;
        const int iext = a.isoiext[a.ph_iso[g]];
        if (iext < 0)
            continue;
        const int row = a.add ? 0 : iext;
        const double kthresh =
            a.ethresh * __longlong_as_double((long long)a.kmax_bits[(int64_t)layer * a.nrows + row]);
        const Rec16 r = a.rec16[i];
        const int64_t ulo = r.ulo, uhi = ulo + (int64_t)(r.lc & ((1u << lbits) - 1u));
        const int64_t lo = max(ulo, a.wbegin) - a.wbegin;
        const int64_t hi = min(uhi, a.wbegin + a.wcount) - a.wbegin;
        if (r.k < kthresh || hi <= lo)
            continue;
        useful += (unsigned long long)(hi - lo);
        if (a.nch_max == 1) {
            issued += (unsigned long long)(((hi - 1) / kStageSpan - lo / kStageSpan + 1) * kStageSpan);
        } else {
            // a long row is visited chunk by chunk: the spans every chunk's part of the window touches
            const int cell = (int)(r.lc >> lbits);
            const int q = floor_div_inv(a.psize[cell] - a.ph_iown[g], a.inv_osamp);
            for (int64_t c0 = 0; c0 < a.rowcap; c0 += kChunkRow) {
                const int64_t clo = max(max(ulo + q, c0) - q, a.wbegin) - a.wbegin;
                const int64_t chi = min(min(uhi + q, c0 + kChunkRow) - q, a.wbegin + a.wcount) - a.wbegin;
                if (chi > clo)
                    issued += (unsigned long long)(((chi - 1) / kStageSpan - clo / kStageSpan + 1) *
                                                   kStageSpan);
            }
        }
        live++;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        useful += __shfl_down(useful, d);
        issued += __shfl_down(issued, d);
        live += __shfl_down(live, d);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], useful);
        atomicAdd(&out[1], issued);
        atomicAdd(&out[2], live);
    }
}

// Test aid (PB_POISON_RECORDS=1): new record buffers are filled with LIVE records of an enormous
// strength that select the whole grid from table cell 0, instead of zeros (dead records).  A
// gather kernel that examines a record k_records did not write in this call then shows up as a
// result of ~1e300, not as silence -- tests/test_gpu_extinction.py::test_shards_never_read_unwritten_records.
__global__ __launch_bounds__(kBlock) void k_poison_records(Rec16 *rec16, int64_t n16, Rec32 *rec32,
                                                          int64_t n32, double *rec_k,
                                                          int32_t *rec_i32, int64_t nsoa)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride) {
        Rec16 r;
        r.k = 1e300;
        r.ulo = 0;
        r.lc = 0xfffu;                      // 4095 samples of cell 0
        rec16[i] = r;
    }
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n32; i += stride) {
        Rec32 r;
        r.k = 1e300;
        r.off = 0;
        r.ulo = 0;
        r.uhi = INT_MAX;
        r.pad[0] = r.pad[1] = 0;
        rec32[i] = r;
    }
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nsoa; i += stride) {
        rec_k[i] = 1e300;
        rec_i32[i] = 0;                     // ulo
        rec_i32[nsoa + i] = INT_MAX;        // uhi
        rec_i32[2 * nsoa + i] = 0;          // q
        rec_i32[3 * nsoa + i] = 0;          // cell
        rec_i32[4 * nsoa + i] = 0;          // phi
    }
}

// Distinct Voigt-table samples the live records of the last launch select: per (layer, isotope,
// Doppler column, phase) row the longest window any record takes from it (maxlen, zeroed by the
// caller); the sum of those lengths is what ANY gather must read of the table at least once --
// the operand SURVEY 8(d)'s byte count leaves out.
__global__ __launch_bounds__(kBlock) void k_table_rows(LblArgs a, int32_t *maxlen)
{
    const int64_t n = (int64_t)a.nlayers * a.ngroups;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kBlock) {
        const int layer = (int)(i / a.ngroups);
        const int64_t g = i - (int64_t)layer * a.ngroups;
        const int iso = a.ph_iso[g];
        const int iext = a.isoiext[iso];
        if (iext < 0)
            continue;
        const int row = a.add ? 0 : iext;
        const double kthresh =
            a.ethresh * __longlong_as_double((long long)a.kmax_bits[(int64_t)layer * a.nrows + row]);
        const Rec16 r = a.rec16[i];
        const int len = (int)(r.lc & 0xfffu);
        const int64_t lo = max((int64_t)r.ulo, a.wbegin);
        const int64_t hi = min((int64_t)r.ulo + len, a.wbegin + a.wcount);
        if (r.k < kthresh || hi <= lo)
            continue;
        const int cell = (int)(r.lc >> 12);
        const int idop = cell - a.li_ilor[(int64_t)layer * a.niso + iso] * a.ndop;
        const int d = a.psize[cell] - a.ph_iown[g];
        const int q = floor_div_inv(d, a.inv_osamp);
        const int phi = d - q * a.osamp;
        atomicMax(&maxlen[(((int64_t)layer * a.niso + iso) * a.ndop + idop) * a.osamp + phi], len);
    }
}

__global__ __launch_bounds__(kBlock) void k_sum_i32(const int32_t *v, int64_t n,
                                                    unsigned long long *out)
{
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kBlock)
        acc += (unsigned long long)v[i];
    for (int d = 32; d >= 1; d >>= 1)
        acc += __shfl_down(acc, d);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(out, acc);
}

// ---------------------------------------------------------------------------
// 3a''. Resident-profile gather (constant-step grid), for layers whose profiles are narrow:
// the WHOLE phase-major block of a table cell (osamp rows of a few tens of samples,
// <= res_cap doubles) is copied into LDS once per (workgroup, isotope, cell), and then
// every lane owns ONE output sample and walks, in position order, the records whose window
// reaches its wavefront's 64 samples: one 16-byte broadcast read of the record, one
// `ds_read_b64` of row[phi][sample + q] and one fma per (record, 64 samples), with a
// per-lane window predicate that redirects outside lanes to a zero slot.
// With ~20-sample rows a record costs 64 lanes of work instead of the 256 (staged) or 512
// (global) of the kernels above, and there are no per-phase segments to synchronise on.
// The terms of a sample are added in (isotope, position) order -- the order of the global
// gather -- whatever the tiling.
// ---------------------------------------------------------------------------
constexpr int kResWaves = 8;
constexpr int kResThreads = kResWaves * 64;
constexpr int kResStrips = 2;                              // 64-sample strips per wavefront
constexpr int kResTile = kResThreads * kResStrips;         // output samples per workgroup
constexpr int kResCapDefault = 8192;                       // doubles of LDS for one profile block
static_assert(kResTile < 65536, "window coordinates are packed in 16 bits");

__global__ __launch_bounds__(kResThreads) void k_ext_resident(LblArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    struct __align__(16) Rec {
        double k;
        int off;                  // LDS byte offset of tile sample 0 in the record's phase row
        unsigned win;             // lo | length << 16 (tile coordinates)
    };
    double *s_prof = reinterpret_cast<double *>(smem);                   // [res_cap] + zero slot
    const int zslot = a.res_cap;
    Rec *s_rec = reinterpret_cast<Rec *>(s_prof + a.res_cap + 2);        // [kResThreads]
    int *s_cell = reinterpret_cast<int *>(s_rec + kResThreads);          // [kResThreads]
    int *s_part = s_cell + kResThreads;                                  // [kResWaves]

    int tile, layer;
    decode_block(a, tile, layer);
    if (layer < 0 || !a.ls_resident[layer])
        return;
    const int row = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int64_t t0 = a.wbegin + (int64_t)tile * kResTile;
    const int64_t tend = min(t0 + kResTile, a.wbegin + a.wcount);
    const int tlen = (int)(tend - t0);
    const double kthresh =
        a.ethresh * __longlong_as_double((long long)a.kmax_bits[(int64_t)layer * a.nrows + row]);
    const int64_t recbase = (int64_t)layer * a.ngroups;
    const int osamp = a.osamp;
    const int ofactor = a.ls_ofactor[layer];

    double acc[kResStrips];
#pragma unroll
    for (int u = 0; u < kResStrips; u++)
        acc[u] = 0.0;
    if (tid == 0)
        s_prof[zslot] = 0.0;

    for (int iso = 0; iso < a.niso; iso++) {
        const int iext = a.isoiext[iso];
        if (iext < 0 || (a.add ? 0 : iext) != row)
            continue;
        const int64_t li = (int64_t)layer * a.niso + iso;
        const double dens = a.li_dens[li];
        int64_t reach = a.li_hmax[li];
        if (a.cutoff > 0.0)
            reach = min(reach, (int64_t)(a.cutoff / a.ownstep) + 2 * (int64_t)ofactor + 2);
        reach += osamp + ofactor;
        // candidates: groups at fine positions [t0*osamp - reach, (tend-1)*osamp + reach],
        // bracketed through the per-sample index of the position-sorted list
        const int64_t rs = reach / osamp + 1;
        const int32_t *gs = a.gs_start + (int64_t)iso * (a.nwave + 1);
        const int64_t g0 = gs[max((int64_t)0, t0 - rs)];
        const int64_t g1 = gs[min((int64_t)a.nwave, tend + rs)];

        for (int64_t gb = g0; gb < g1; gb += kResThreads) {
            const int nrec = (int)min((int64_t)kResThreads, g1 - gb);
            __syncthreads();               // the previous walk has finished with s_rec / s_prof
            {
                double k = 0.0;
                unsigned win = 0;
                int off = 0, cell = -1;
                if (tid < nrec) {
                    const int64_t idx = recbase + gb + tid;
                    k = a.rec_k[idx];
                    const int ulo = a.rec_ulo[idx], uhi = a.rec_uhi[idx];
                    const int lo = (int)(max((int64_t)ulo, t0) - t0);
                    const int hi = (int)(min((int64_t)uhi, tend) - t0);
                    if (!(k < kthresh) && lo < hi) {
                        if (a.add)
                            k *= dens;
                        cell = a.rec_cell[idx];
                        win = (unsigned)lo | ((unsigned)(hi - lo) << 16);
                        // tile sample j reads s_prof[phi*stride + (j + t0 + q)]
                        off = 8 * (a.rec_phi[idx] * a.pm_stride[cell] + (int)(a.rec_q[idx] + t0));
                    } else {
                        k = 0.0;
                    }
                }
                s_rec[tid].k = k;
                s_rec[tid].off = off;
                s_rec[tid].win = win;
                s_cell[tid] = cell;
            }
            // the cells of a batch, in ascending order (the Doppler index grows with position)
            int cur = -1;
            for (;;) {
                __syncthreads();
                int c = s_cell[tid];
                c = c > cur ? c : INT_MAX;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1)
                    c = min(c, __shfl_xor(c, d));
                if (lane == 0)
                    s_part[wave] = c;
                __syncthreads();
                c = INT_MAX;
                for (int w = 0; w < kResWaves; w++)
                    c = min(c, s_part[w]);
                if (c == INT_MAX)
                    break;
                cur = c;
                // ---- the cell's phase-major block -> LDS ----
                {
                    const int n = a.pm_stride[cur] * osamp;
                    const double *src = a.pm + a.pm_base[cur];
                    for (int i = tid; i < n; i += kResThreads)
                        s_prof[i] = src[i];
                }
                __syncthreads();
                // ---- every wavefront walks the records that reach its strips ----
                const double2 *recs = reinterpret_cast<const double2 *>(s_rec);
#pragma unroll
                for (int u = 0; u < kResStrips; u++) {
                    const int slo = (wave + u * kResWaves) * 64;
                    if (slo >= tlen)
                        continue;
                    int first = INT_MAX, last = -1;
                    for (int b = 0; b < nrec; b += 64) {
                        const int e = b + lane;            // < kResThreads
                        const unsigned w = s_rec[e].win;
                        const int wlo = (int)(w & 0xffff);
                        const bool hit = s_cell[e] == cur && wlo < slo + 64 &&
                                         wlo + (int)(w >> 16) > slo;
                        const unsigned long long m = __ballot(hit);
                        if (m) {
                            first = min(first, b + (int)__builtin_ctzll(m));
                            last = max(last, b + 63 - (int)__builtin_clzll(m));
                        }
                    }
                    first = __builtin_amdgcn_readfirstlane(first);
                    last = __builtin_amdgcn_readfirstlane(last);
                    const unsigned j = (unsigned)(slo + lane);
                    typedef __attribute__((address_space(3))) const double lds_double;
                    // LDS byte addresses: the lane's sample in row 0, and the zero slot
                    const unsigned prof0 = (unsigned)(uintptr_t)(
                        (__attribute__((address_space(3))) const char *)s_prof);
                    const unsigned j8 = prof0 + j * 8u, z8 = prof0 + (unsigned)zslot * 8u;
                    double sum = acc[u];
#pragma unroll 4
                    for (int r = first; r <= last; r++) {
                        const double2 raw = recs[r];
                        const unsigned w = (unsigned)__double2hiint(raw.y);
                        const bool in = j - (w & 0xffffu) < (w >> 16);
                        const unsigned at = in ? j8 + (unsigned)__double2loint(raw.y) : z8;
                        sum = fma(raw.x, *(lds_double *)(uintptr_t)at, sum);
                    }
                    acc[u] = sum;
                }
            }
        }
    }

    double *dst = a.ext + ((int64_t)layer * a.nrows + row) * a.wcount + (t0 - a.wbegin);
#pragma unroll
    for (int u = 0; u < kResStrips; u++) {
        const int j = (wave + u * kResWaves) * 64 + lane;
        if (j < tlen)
            dst[j] = acc[u];
    }
}

#ifdef PB_EXPERIMENTS   // gather mode 4 (4x slower than the staged kernel)
// ---------------------------------------------------------------------------
// 3s. Scatter kernel (constant-step grid): ONE wavefront owns a tile of T output samples
// held in LDS and walks, in (isotope, position) order, the records whose window reaches the
// tile.  Per record and 64 samples: one coalesced `buffer_load_dwordx2` straight from the
// phase-major table (the window is the buffer, lanes past its end read 0), one v_mul_f64,
// one `ds_add_f64` into the tile -- products and sums rounded separately like the
// reference's `ktmp[j] += k * profile[...]`.  No staging, no phase sorting, no workgroup
// barrier; the table traffic is exactly the window lengths.  Record fields are wave-uniform
// and arrive by scalar loads.  The tile belongs to one wavefront and LDS executes a
// wavefront's operations in order, so every sample is summed in record order whatever the
// tiling: results are bitwise reproducible and shards concatenate exactly.
// ---------------------------------------------------------------------------
constexpr int kScatterPad = 64;       // lanes past a window's end add zeros there

// The walk of one tile with C chunks (64 samples each) per record in flight.
template <int C>
__device__ inline void scatter_walk(const LblArgs &a, double *s_tile, int layer, int row,
                                    int64_t t0, int64_t tend, double kthresh)
{
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x;
    const int osamp = a.osamp;
    const int ofactor = a.ls_ofactor[layer];
    const Rec32 *recs = a.rec32 + (int64_t)layer * a.ngroups;

    for (int iso = 0; iso < a.niso; iso++) {
        const int iext = a.isoiext[iso];
        if (iext < 0 || (a.add ? 0 : iext) != row)
            continue;
        const int64_t li = (int64_t)layer * a.niso + iso;
        const double dens = a.add ? a.li_dens[li] : 1.0;
        int64_t reach = a.li_hmax[li];
        if (a.cutoff > 0.0)
            reach = min(reach, (int64_t)(a.cutoff / a.ownstep) + 2 * (int64_t)ofactor + 2);
        reach += osamp + ofactor;
        const int64_t rs = reach / osamp + 1;
        const int32_t *gs = a.gs_start + (int64_t)iso * (a.nwave + 1);
        const int g0 = gs[max((int64_t)0, t0 - rs)];
        const int g1 = gs[min((int64_t)a.nwave, tend + rs)];

        // 64 records per batch, one per lane; the next batch is requested before this one
        // is walked
        v4i w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
        auto fetch = [&](int g) {
            const v4i *p = reinterpret_cast<const v4i *>(recs + min(g + lane, g1 - 1));
            w0 = p[0];
            w1 = p[1];
        };
        if (g0 < g1)
            fetch(g0);
        for (int gb = g0; gb < g1; gb += 64) {
            const double kk = __hiloint2double(w0.y, w0.x);
            const long long off = ((long long)w0.w << 32) | (unsigned)w0.z;
            const int lo = max(w1.x, (int)t0), hi = min(w1.y, (int)tend);
            const bool ok = gb + lane < g1 && !(kk < kthresh) && lo < hi;
            const double b_k = kk * dens;
            const long long b_src = off + lo;
            const int b_lo = lo - (int)t0, b_n = hi - lo;
            unsigned long long live = __ballot(ok);
            if (gb + 64 < g1)
                fetch(gb + 64);

            // the live records, two in flight: the C loads of the next one are issued before
            // the current one is multiplied and added
            struct Cur {
                int dst, nch;
                double k;
            };
            auto start = [&](double (&v)[C], Cur &c) {
                const int sl = (int)__builtin_ctzll(live);
                live &= live - 1;
                c.k = bcast(b_k, sl);
                const int s_lo = __builtin_amdgcn_readlane((int)b_src, sl);
                const int s_hi = __builtin_amdgcn_readlane((int)(b_src >> 32), sl);
                const int n = __builtin_amdgcn_readlane(b_n, sl);
                c.dst = __builtin_amdgcn_readlane(b_lo, sl);
                c.nch = (n + 63) >> 6;
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(a.pm + (((long long)s_hi << 32) | (unsigned)s_lo)), 0, n * 8,
                    0x00020000);
#pragma unroll
                for (int q = 0; q < C; q++) {
                    const v2i w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane * 8, q * 512, 0);
                    v[q] = __hiloint2double(w.y, w.x);
                }
            };
            auto finish = [&](const double (&v)[C], const Cur &c) {
                double *dst = s_tile + c.dst + lane;
#pragma unroll
                for (int q = 0; q < C; q++)
                    if (q < c.nch)
                        __hip_atomic_fetch_add(dst + q * 64, c.k * v[q], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            double va[C], vb[C];
            Cur ca, cb;
            if (!live)
                continue;
            start(va, ca);
            for (;;) {
                if (!live) {
                    finish(va, ca);
                    break;
                }
                start(vb, cb);
                finish(va, ca);
                if (!live) {
                    finish(vb, cb);
                    break;
                }
                start(va, ca);
                finish(vb, cb);
            }
        }
    }
}

template <int T>
__global__ __launch_bounds__(64) void k_ext_scatter(LblArgs a)
{
    __shared__ double s_tile[T + kScatterPad];
    int tile, layer;
    decode_block(a, tile, layer);
    if (layer < 0)
        return;
    const int row = blockIdx.y;
    const int lane = threadIdx.x;
    const int64_t t0 = a.wbegin + (int64_t)tile * T;
    const int64_t tend = min(t0 + T, a.wbegin + a.wcount);
    const int tlen = (int)(tend - t0);
    const double kthresh =
        a.ethresh * __longlong_as_double((long long)a.kmax_bits[(int64_t)layer * a.nrows + row]);
    for (int i = lane; i < T + kScatterPad; i += 64)
        s_tile[i] = 0.0;
    // chunks per record: the longest phase row this layer can select
    const int rowmax = a.ls_block[layer] / a.osamp;
    const int chunks = (min(rowmax, T) + 63) >> 6;
    if (chunks <= 4)
        scatter_walk<4>(a, s_tile, layer, row, t0, tend, kthresh);
    else if (chunks <= 6)
        scatter_walk<6>(a, s_tile, layer, row, t0, tend, kthresh);
    else if (chunks <= 8)
        scatter_walk<8>(a, s_tile, layer, row, t0, tend, kthresh);
    else if (chunks <= 12)
        scatter_walk<12>(a, s_tile, layer, row, t0, tend, kthresh);
    else if (chunks <= 16)
        scatter_walk<16>(a, s_tile, layer, row, t0, tend, kthresh);
    else
        scatter_walk<32>(a, s_tile, layer, row, t0, tend, kthresh);
    __builtin_amdgcn_s_waitcnt(0);     // the tile is complete (same wavefront: in order)
    double *out = a.ext + ((int64_t)layer * a.nrows + row) * a.wcount + (t0 - a.wbegin);
    for (int i = lane; i < tlen; i += 64)
        out[i] = s_tile[i];
}

#endif  // PB_EXPERIMENTS

// ext += part[0] + part[1] + ... in that order (the phase splits of a small staged launch).
// skip[layer] != 0: a layer another kernel computed whole (the resident-profile kernel): the
// staged kernel wrote NO partial sums for it, its planes hold whatever an earlier call left there
// (found by tools/fuzz_r4.py: a plan used in `staged` mode and then in automatic mode with resident
// layers added the earlier call's pieces to the resident kernel's result; on a fresh plan the
// planes are fresh zero pages, which is why no test saw it).
__global__ __launch_bounds__(kBlock) void k_combine_parts(double *ext, const double *part,
                                                         int nparts, int64_t n,
                                                         const int32_t *skip, int64_t per_layer)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    if (skip && skip[i / per_layer])
        return;
    double v = ext[i];
    for (int p = 0; p < nparts; p++)
        v += part[(int64_t)p * n + i];
    ext[i] = v;
}

// the same with a per-tile number of pieces (LblArgs::tsplit): sample w of a row belongs to tile
// w / tile and has tsplit[tile] - 1 partial planes
__global__ __launch_bounds__(kBlock) void k_combine_tile_parts(double *ext, const double *part,
                                                              const int32_t *tsplit, int tile,
                                                              int64_t wcount, int64_t n,
                                                              const int32_t *skip, int64_t per_layer)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    const int np = tsplit[(i % wcount) / tile] - 1;
    if (np <= 0 || (skip && skip[i / per_layer]))
        return;
    double v = ext[i];
    for (int p = 0; p < np; p++)
        v += part[(int64_t)p * n + i];
    ext[i] = v;
}

// the same with a per-layer number of pieces: grid.y = layer, layers in one piece are skipped
__global__ __launch_bounds__(kBlock) void k_combine_layer_parts(double *ext, const double *part,
                                                               const int32_t *lsplit,
                                                               int64_t per_layer, int64_t n,
                                                               const int32_t *skip)
{
    const int layer = blockIdx.y;
    const int nparts = lsplit[layer] - 1;
    if (nparts <= 0 || (skip && skip[layer]))
        return;
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= per_layer)
        return;
    const int64_t i = (int64_t)layer * per_layer + j;
    double v = ext[i];
    for (int p = 0; p < nparts; p++)
        v += part[(int64_t)p * n + i];
    ext[i] = v;
}

// ---------------------------------------------------------------------------
// 3b. gather, arbitrary output grid (resolution / wlstep mode): every output needs the
// two dynamic-grid samples that bracket it (linterp, utils.h:139-163).  Uses the
// reference-layout table (the stride between consecutive outputs is not constant).
// One output sample per lane, 256 per workgroup.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_ext_linterp(LblArgs a)
{
    // one record of the batch: two 16-byte LDS broadcast reads per (record, wavefront)
    struct __align__(16) RecA {
        double k;
        int64_t start;                                  // start of the profile in flat[]
    };
    struct __align__(16) RecB {
        int inoff, half2, mn, mx;                       // half - iown, 2*half, window [mn, mx)
    };
    __shared__ RecA s_ra[kBlock + 4];
    __shared__ RecB s_rb[kBlock + 4];

    int tile, layer;
    decode_block(a, tile, layer);
    if (layer < 0)
        return;
    if (a.lskip && uniform_load_i32(a.lskip, layer))
        return;
    const int row = blockIdx.y;

    const int64_t t0 = a.wbegin + (int64_t)tile * kBlock;
    const int64_t tend = min(t0 + kBlock, a.wbegin + a.wcount);
    const int64_t jo = t0 + threadIdx.x;
    const bool live = jo < tend;

    const int ofactor = a.ls_ofactor[layer];
    const int64_t dnwn = a.ls_dnwn[layer];
    const double dwnstep = a.ls_dwnstep[layer];
    const double temp = a.temp[layer];
    const double kthresh =
        a.ethresh * __longlong_as_double((long long)a.kmax_bits[(int64_t)layer * a.nrows + row]);

    // bracketing dynamic-grid sample of this output and of the tile's ends
    const double wn_i = live ? a.wn[jo] : a.wn[tend - 1];
    const int ilo = (int)((wn_i - a.wn0) / dwnstep);
    const int tile_jmin = (int)((a.wn[t0] - a.wn0) / dwnstep);
    const int tile_jmax = (int)((a.wn[tend - 1] - a.wn0) / dwnstep) + 1;

    double acc0 = 0.0, acc1 = 0.0;
    for (int iso = 0; iso < a.niso; iso++) {
        const int iext = a.isoiext[iso];
        if (iext < 0 || (a.add ? 0 : iext) != row)
            continue;
        const int64_t li = (int64_t)layer * a.niso + iso;
        const int ilor = a.li_ilor[li];
        const double alphad = a.li_alphad[li];
        const double ratio = a.isoratio[iso];
        const double z = a.li_z[li], inv_z = a.li_invz[li];
        const double inv_temp = a.ls_inv_temp[layer];
        const double dens = a.li_dens[li];
        int64_t reach = a.li_hmax[li];
        if (a.cutoff > 0.0)
            reach = min(reach, (int64_t)(a.cutoff / a.ownstep) + 2 * (int64_t)ofactor + 2);
        reach += 2 * (int64_t)ofactor;
        const int64_t seg0 = a.iso_gstart[iso], seg1 = a.iso_gstart[iso + 1];
        const int64_t g0 =
            lower_bound_i32(a.giown, seg0, seg1, (int64_t)tile_jmin * ofactor - reach);
        const int64_t g1 =
            lower_bound_i32(a.giown, seg0, seg1, (int64_t)tile_jmax * ofactor + reach + 1);
        for (int64_t gb = g0; gb < g1; gb += kBlock) {
            __syncthreads();
            {
                const int64_t g = gb + threadIdx.x;
                double k = 0.0;
                int64_t start = 0;
                int mn = 0, mx = 0, half2 = 0, inoff = 0;
                if (g < g1) {
                    const int first = a.gfirst[g];
                    const int iown = a.giown[g];
                    k = group_strength(a, first, a.gcount[g], ratio, temp, inv_temp, z, inv_z);
                    if (!(k < kthresh)) {
                        if (a.add)
                            k *= dens;
                        const Window w = group_window(a, a.lwn[first], iown, ilor, alphad,
                                                      ofactor, dwnstep, dnwn, 0, a.ndop - 1,
                                                      nullptr, a.ls_cutsteps[layer],
                                                      a.ls_inv_ofactor[layer]);
                        // dynamic sample j reads flat[pindex + half + ofactor*j - iown]
                        mn = (int)w.minj;
                        mx = (int)w.maxj;
                        half2 = 2 * w.half;
                        inoff = w.half - iown;
                        start = a.pindex[w.cell];
                    }
                }
                // (a dead record keeps an empty window, k = 0 and the start of the table)
                s_ra[threadIdx.x] = RecA{k, start};
                s_rb[threadIdx.x] = RecB{inoff, half2, mn, mx};
                if (threadIdx.x < 4) {                  // the padding of the last trip of four
                    s_ra[kBlock + threadIdx.x] = RecA{0.0, 0};
                    s_rb[kBlock + threadIdx.x] = RecB{0, 0, 0, 0};
                }
            }
            __syncthreads();
            const int nrec = (int)min((int64_t)kBlock, g1 - gb);
            // Four records per trip and NO branch around the table reads: a lane outside a
            // record's window reads element 0 of that profile with a zero strength instead.
            // Behind per-record branches every pair of reads was drained (s_waitcnt vmcnt(0))
            // before the next record's were issued; now eight gathers are in flight per lane.
            // The sums see the same terms in the same order (+ k * 0-weight terms that are
            // exactly zero: profile samples are finite).
            for (int e = 0; e < nrec; e += 4) {
                double k0[4], k1[4], v0[4], v1[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const RecA ra = s_ra[e + u];        // (entries beyond nrec: k = 0 records of
                    const RecB rb = s_rb[e + u];        // this or an earlier batch, or the padding)
                    const bool have = e + u < nrec;
                    const int64_t f0 = rb.inoff + (int64_t)ofactor * ilo;
                    const int64_t f1 = f0 + ofactor;
                    const bool in0 = have && live && ilo >= rb.mn && ilo < rb.mx && f0 >= 0 &&
                                     f0 <= rb.half2;
                    const bool in1 = have && live && ilo + 1 >= rb.mn && ilo + 1 < rb.mx && f1 >= 0 &&
                                     f1 <= rb.half2;
                    const double *tab = a.flat + ra.start;
                    v0[u] = tab[in0 ? f0 : 0];
                    v1[u] = tab[in1 ? f1 : 0];
                    k0[u] = in0 ? ra.k : 0.0;
                    k1[u] = in1 ? ra.k : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    acc0 = fma(k0[u], v0[u], acc0);
                    acc1 = fma(k1[u], v1[u], acc1);
                }
            }
        }
    }
    if (live) {
        const double wnlo = a.wn0 + dwnstep * ilo;
        a.ext[((int64_t)layer * a.nrows + row) * a.wcount + (jo - a.wbegin)] +=
            (acc0 * (wnlo + dwnstep - wn_i) + acc1 * (wn_i - wnlo)) / dwnstep;
    }
}

// 3b'. `resolution` mode through the dynamic grids: the sums of a run of layers on their
// dynamic grid (ktmp[layer][row][d0 .. d0+dcount), computed by a constant-step plan of step
// ofactor) interpolated onto the output grid exactly as utils.h:139-163 does, accumulated into ext.
// grid (output blocks, layers of the run x rows)
__global__ __launch_bounds__(kBlock) void k_dyn_interp(double *ext, const double *ktmp,
                                                      const double *wn, double wn0,
                                                      const double *dwnstep, int64_t d0,
                                                      int64_t dcount, int64_t wbegin,
                                                      int64_t wcount, int nrows,
                                                      const int32_t *ok)
{
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= wcount)
        return;
    const int lr = blockIdx.y;
    if (!uniform_load_i32(ok, lr / nrows))               // (a layer the run plan did not fit)
        return;
    const double step = dwnstep[lr / nrows];
    const double wn_i = wn[wbegin + j];
    const int64_t ilo = (int)((wn_i - wn0) / step);
    const double *src = ktmp + (int64_t)lr * dcount - d0;
    const double v0 = ilo >= d0 && ilo < d0 + dcount ? src[ilo] : 0.0;
    const double v1 = ilo + 1 >= d0 && ilo + 1 < d0 + dcount ? src[ilo + 1] : 0.0;
    const double wnlo = wn0 + step * ilo;
    ext[(int64_t)lr * wcount + j] += (v0 * (wnlo + step - wn_i) + v1 * (wn_i - wnlo)) / step;
}

// Which layers does the run plan of a host-free `resolution` call fit?  The plan was made from
// the factors and Lorentz rows the layers had when they were last read back; a layer is computed
// by its run iff its factor is the predicted one and every Lorentz row its isotopes select is
// filled in the re-cut table of that factor (unfilled rows read as zeros: wrong, never a fault).
__global__ void k_dyn_check(int32_t *ok, const int32_t *ofactor, const int32_t *ilor,
                            const int32_t *pred_f, const uint8_t *const *pred_mask, int nlayers,
                            int niso)
{
    const int layer = blockIdx.x * blockDim.x + threadIdx.x;
    if (layer >= nlayers)
        return;
    bool good = ofactor[layer] == pred_f[layer];
    const uint8_t *mask = pred_mask[layer];
    for (int i = 0; i < niso; i++)
        good = good && mask[ilor[(int64_t)layer * niso + i]] != 0;
    ok[layer] = good ? 1 : 0;
}

// the per-row maxima of a run's layers -> the plan's (only the layers the run computed)
__global__ void k_dyn_kmax(unsigned long long *dst, const unsigned long long *src,
                           const int32_t *ok, int nl, int nrows)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nl * nrows && ok[e / nrows])
        dst[e] = src[e];
}

// ---------------------------------------------------------------------------
// _extcoeff.interp_ec / interp_ec_per_mol (src_c/_extcoeff.c:367-472)
// grid (wavenumber blocks, layers, per_mol ? nmol : 1)
// ---------------------------------------------------------------------------
// kAssign: ext = sum instead of ext += sum (a caller that would zero ext first saves that
// pass and the read: 128 MB of 768 at the C5 shape)
template <bool kAssign>
__global__ __launch_bounds__(kBlock) void k_interp_ec(
    double *ext, const double *etable, const double *ttable, const double *temps,
    const double *density, int nmol, int ntemp, int nlayers, int nwave, int lay1,
    int per_mol)
{
    const int k = lay1 + blockIdx.y;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nwave)
        return;
    const double t = temps[k];
    int tlo = pb::nearest_index(ttable, t, 0, ntemp - 1);
    if (t < ttable[tlo] || tlo == ntemp - 1)
        tlo--;
    // A temperature below the table makes the reference read ttable[-1] and etable at a
    // negative offset (_extcoeff.c:394-398; its callers reject such models first,
    // line_sampling.py:426-427).  Here the bracket is clamped -- in-range results are
    // unchanged, an out-of-range layer extrapolates from the first interval -- so a
    // caller that forgot the check gets numbers instead of a GPU fault.
    tlo = max(tlo, 0);
    const int thi = tlo + 1;
    const double span = ttable[thi] - ttable[tlo];
    const double w_lo = (ttable[thi] - t) / span;
    const double w_hi = (t - ttable[tlo]) / span;
    if (per_mol) {
        const int j = blockIdx.z;
        const double d = density[(int64_t)k * nmol + j];
        const double lo = etable[(((int64_t)j * ntemp + tlo) * nlayers + k) * nwave + i];
        const double hi = etable[(((int64_t)j * ntemp + thi) * nlayers + k) * nwave + i];
        double *e = ext + ((int64_t)j * nlayers + k) * nwave + i;
        *e = (kAssign ? 0.0 : *e) + (lo * (w_lo * d) + hi * (w_hi * d));
    } else {
        double acc = kAssign ? 0.0 : ext[(int64_t)k * nwave + i];
        for (int j = 0; j < nmol; j++) {
            const double d = density[(int64_t)k * nmol + j];
            const double lo = etable[(((int64_t)j * ntemp + tlo) * nlayers + k) * nwave + i];
            const double hi = etable[(((int64_t)j * ntemp + thi) * nlayers + k) * nwave + i];
            acc += lo * (w_lo * d) + hi * (w_hi * d);
        }
        ext[(int64_t)k * nwave + i] = acc;
    }
}

template <typename T>
int upload(T **dst, const T *src, size_t n)
{
    PB_HIP(hipMalloc(dst, std::max<size_t>(n, 1) * sizeof(T)));
    if (n)
        PB_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PB_OK;
}

}  // namespace

// ===========================================================================
// handles
// ===========================================================================
struct pb_lbl {
    pb_voigt *voigt = nullptr;
    pb_lines *lines = nullptr;
    int nwave = 0, nmol = 0, niso = 0, ndivs = 0, max_layers = 0, resolution = 0;
    int nrows_sep = 1;       // rows when add == 0
    double cutoff = 0, ethresh = 0, wnstep = 0, wn0 = 0;
    std::vector<int32_t> isoiext;
    double *d_wn = nullptr, *d_molrad = nullptr, *d_molmass = nullptr, *d_isomass = nullptr,
           *d_isoratio = nullptr;
    int32_t *d_divisors = nullptr, *d_isoimol = nullptr, *d_isoiext = nullptr;
    // workspace
    int32_t *ls_ofactor = nullptr, *ls_scale = nullptr, *li_ilor = nullptr, *li_hmax = nullptr;
    int32_t *li_rowmax = nullptr, *li_hlo = nullptr, *li_hhi = nullptr;
    int64_t *ls_dnwn = nullptr;
    double *ls_dwnstep = nullptr, *li_alphad = nullptr, *li_dens = nullptr, *li_z = nullptr;
    double *ls_quot = nullptr;        // [4][max_layers]: cutsteps, 1/ofactor, 1/scale, 1/temp
    double *li_invz = nullptr;        // [max_layers][niso] 1 / partition function
    unsigned long long *kmax_bits = nullptr;
    int kmax_rows = 0;
    // phase-sorted copy of the groups for the LDS-staged kernel
    int32_t *ph_first = nullptr, *ph_count = nullptr, *ph_iown = nullptr;
    int64_t *ph_start = nullptr;
    int32_t *ph_iso = nullptr;
    int32_t *ph_bin = nullptr;
    int ph_nbins = 0;
    double *ph_lead = nullptr;       // leader lwn, elow, gf of the phase-sorted groups [3][G]
    double *g_lead = nullptr;        // same for the position-sorted groups
    double *rec_k = nullptr;
    int32_t *rec_i32 = nullptr;      // 5 arrays of max_layers*ngroups
    int rowcap = 0;
    int32_t *ls_resident = nullptr;   // [max_layers]
    int32_t *ls_block = nullptr;      // [max_layers]
    int32_t *ls_wave = nullptr;       // [max_layers] layers of the wave-autonomous kernel
    Rec32 *rec32 = nullptr;           // [max_layers][ngroups], scatter kernel
    Rec16 *rec16 = nullptr;           // [layers of the largest call][ngroups][nch_max], staged kernel
    size_t rec16_alloc = 0;
    double *part = nullptr;           // partial sums of a phase-split staged launch
    size_t part_bytes = 0;
    // window map of two-phase shard calls (LblArgs::wm_*): host copies of the phase-sorted group
    // positions, the cached map and the window / order it was built for
    std::vector<int32_t> h_ph_iown;
    std::vector<int64_t> h_ph_start;
    std::vector<int32_t> h_wm;
    int32_t *d_wm = nullptr;
    size_t wm_cap = 0;
    int64_t wm_flo = 0, wm_fhi = -1;
    int wm_staged = -1, wm_n0 = 0, wm_n1 = 0;
    int64_t wm_total0 = 0, wm_total1 = 0;
    int32_t *gs_start = nullptr;      // [niso][nwave+1]
    int res_cap = 0;                  // LDS doubles of one resident profile block (0 = none fits)
    // Which layers are resident is decided on the device, per call; a plan none of whose layers
    // ever qualifies (C2: the smallest block a layer selects is 53 820 doubles) still paid an
    // empty launch of the resident kernel on every spectrum (7.5 us).  The host looks at the
    // decision of the first automatic call and of every 256th one (one small synchronous copy
    // each): while no layer qualified the resident kernel is left out (res_cap = 0 for the whole
    // call: the staged / global kernel computes every layer).
    int res_seen = -1;                // -1 not looked yet, 0 no resident layer, 1 some
    uint64_t res_calls = 0;
    bool res_on_pending = false;      // decision of a two-phase call's first half
    bool res_look_pending = false;
    // packed (layer, group) records above this many bytes are produced and consumed in chunks of
    // the line list (pb_lbl_set_record_budget; PB_RECORD_BUDGET overrides)
    size_t record_budget = (size_t)96 << 30;
    int last_chunks = 0;     // chunks of the last call (0 = records of every group at once)
    // per-layer phase split of the staged kernel: device tables and the configuration they hold
    int32_t *d_unit_tab = nullptr, *d_lsplit = nullptr;
    // per-tile phase split (uneven line density): device table and what it was made for
    int32_t *d_tsplit = nullptr;
    int32_t *pos2ph = nullptr;                        // [ngroups] (LblArgs::pos2ph)
    bool ts_sparse = false;                           // some tile of the table is the global gather's
    int64_t ts_key[4] = {-1, -1, -1, -1};            // wbegin, wcount, tile, base split
    int ts_max = 0, ts_tiles = 0;
    int ut_key[4] = {-1, -1, -1, -1};                 // nlayers, base split, deep layers, deep split
    int ut_units = 0;
    int concurrency = 1;     // independent calls the caller keeps in flight beside this plan's
    int gather_mode = 0;     // 0 = choose, 1 = global gather, 2 = LDS-staged, 3 = resident+global
    int last_gather = 0;     // last call: 1 global, 2 staged, 3 linterp; +8 = resident kernel too
    double stage_threshold = 8.0;   // groups per (2048-sample tile, phase) to go staged
    // optional per-launch timing of the gather kernel (bench.py's roofline figure)
    std::vector<hipEvent_t> ev;      // start/stop pairs
    int ev_used = 0;
    // round-staged gather (pb_rounds.hip): per-unit capacities (cached per launch geometry)
    // and the visit-record / segment / header lists
    int64_t *unit_cap = nullptr;
    int64_t cap_key[5] = {-1, -1, -1, -1, -1};   // wbegin, wcount, tile, nsplit, total
    VRec *vrec = nullptr;
    VSeg *vseg = nullptr;
    int32_t *vrnd = nullptr;
    size_t vrec_alloc = 0;            // entries
    UnitHdr *uhdr = nullptr;
    size_t uhdr_alloc = 0;
    struct Pending {                 // call begun with pb_lbl_extinction_begin
        double *ext;
        int64_t wbegin, wcount;
        const double *temp, *dens, *isoz;
        int64_t zs0, zs1;
        int nlayers, add;
        bool open;
    } pending = {nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, 0, 0, false};
    LblArgs last_args;               // arguments of the last launch (pb_lbl_last_work)
    bool last_packed = false;        // ... whose records are packed, one per (layer, group)
    // `resolution` plans, gather mode 6: one constant-step plan per oversampling factor in use
    // (the layer's dynamic grid IS a constant-step grid of step ofactor fine samples), with the
    // Voigt table cut into phase rows modulo that factor (pb_voigt_rephase: kept by the table)
    struct DynSub {
        int f;
        pb_voigt *voigt;
        pb_lbl *plan;
        double *ktmp;                // dynamic-grid sums of one run of layers
        size_t ktmp_bytes;
        uint64_t call;               // last call that used it, and on which side stream
        int lane;
    };
    std::vector<DynSub> dyn;
    // Host-free calls (opt-in, pb_lbl_set_dyn_predict): the run plan comes from the factors /
    // Lorentz rows the layers had when they were last read back (pred_*), a device check marks the
    // layers it fits (d_ok), the direct gather computes the others; this call's state is read back
    // asynchronously (rb_*) and adopted by a later call.  A read-back that contradicts the
    // prediction makes the next dyn_hold calls synchronous (one stream synchronisation each, the
    // default form): atmospheres that change from call to call are not worth predicting.  Opt-in
    // because a layer that takes the direct gather differs from the same layer on its dynamic grid
    // in the last bits (the same terms in another order): with the prediction on, a result can
    // depend on the plan's history at the 1e-13 level; with a steady atmosphere it never does.
    std::vector<int32_t> pred_f, pred_ilor, used_f;
    int pred_layers = 0, pred_cap = 0, dyn_hold = 0;
    bool pred_dirty = false, rb_pending = false, dyn_fallback = false;
    int dyn_predict = 0;             // pb_lbl_set_dyn_predict
    int32_t *d_pred_f = nullptr, *d_ok = nullptr, *rb_host = nullptr;
    const uint8_t **d_pred_mask = nullptr;
    size_t rb_cap = 0;
    int rb_layers = 0;
    hipEvent_t rb_ev = nullptr;
    int64_t dyn_spec_calls = 0, dyn_sync_calls = 0, dyn_mispredicted = 0;
    uint64_t dyn_call = 0;
    int dyn_runs = 0;                // runs of equal-factor layers of the last call
    // the runs of a call are independent until ext: dealt to side streams (deep layers have
    // short dynamic grids and factors of their own: launches of one layer that leave the chip idle)
    std::vector<hipStream_t> dyn_streams;
    std::vector<hipEvent_t> dyn_join;
    hipEvent_t dyn_fork = nullptr;
    std::vector<int32_t> h_ofactor, h_ilor, h_divisors, h_isoimol, h_isoiext0;   // (isoiext at creation)
    std::vector<double> h_wn, h_molrad, h_molmass, h_isomass, h_isoratio;
};

extern "C" {

// ---------------------------------------------------------------------------
// line list
// ---------------------------------------------------------------------------
int pb_lines_create(pb_lines **out, const double *lwn_h, const double *elow_h,
                    const double *gf_h, const int32_t *lid_h, int64_t nlines, int niso,
                    const double *own_h, int64_t onwn, double own0, double ownstep)
{
    PB_REQUIRE(out, "pb_lines_create: null out");
    *out = nullptr;
    PB_REQUIRE(nlines >= 0 && niso > 0 && onwn >= 2, "pb_lines_create: bad sizes");
    PB_REQUIRE(nlines == 0 || (lwn_h && elow_h && gf_h && lid_h),
               "pb_lines_create: null line arrays");
    PB_REQUIRE(nlines < 2147483647LL && onwn < 2147483647LL,
               "pb_lines_create: sizes exceed the reference's 32-bit indices");
    // own[] generated exactly like NumPy (wnlow + arange*ownstep: multiply, then add)
    auto own_at = [&](int64_t i) -> double {
        if (own_h)
            return own_h[i];
        volatile double prod = (double)i * ownstep;
        return own0 + prod;
    };
    pb_lines *l = new (std::nothrow) pb_lines();
    if (!l)
        return PB_ERR_NOMEM;
    l->nlines = nlines;
    l->niso = niso;
    l->onwn = onwn;
    l->own0 = own_at(0);
    l->own_last = own_at(onwn - 1);
    l->ownstep = own_at(1) - own_at(0);
    const double lo = l->own0, hi = l->own_last, step = l->ownstep;

    l->h_lwn.assign(lwn_h, lwn_h + nlines);
    l->h_elow.assign(elow_h, elow_h + nlines);
    l->h_gf.assign(gf_h, gf_h + nlines);
    int rc = PB_OK;
    if (rc == PB_OK) rc = upload(&l->d_lwn, lwn_h, (size_t)nlines);
    if (rc == PB_OK) rc = upload(&l->d_elow, elow_h, (size_t)nlines);
    if (rc == PB_OK) rc = upload(&l->d_gf, gf_h, (size_t)nlines);
    if (rc == PB_OK) rc = upload(&l->d_lid, lid_h, (size_t)nlines);
    if (rc != PB_OK) {
        pb_lines_destroy(l);
        return rc;
    }
    // Grouping on the device (pb_lines.hip) when the list has the TLI order (isotope, then
    // wavenumber) and valid isotope ids; PB_LINES_HOST=1 or any other order: the host loop below
    bool grouped = false;
    if (nlines > 0 && !getenv("PB_LINES_HOST") && lid_h[0] >= 0 && lid_h[nlines - 1] < niso) {
        const int grc = pb_lines_group_device(l, lwn_h, lid_h, own_h);
        if (grc == PB_OK) {
            grouped = true;
            l->grouped_on_device = 1;
        } else if (grc != 1) {
            pb_lines_destroy(l);
            return grc;
        }
    }
    if (!grouped) {
    struct Group {
        int32_t first, count, iown, iso;
    };
    std::vector<Group> groups;
    groups.reserve((size_t)nlines);
    // The lines of ONE isotope must come in ascending wavenumber order (isotopes may interleave).
    // The reference's Doppler-width index is a one-way search from the isotope's previous line
    // (_extcoeff.c:278, utils.h:45-72: `if (value < array[lo]) return lo`): on a list that steps
    // back within an isotope it keeps a stale index, a result that depends on the list order and
    // on which lines the layer's threshold skipped.  The kernels evaluate the nearest index
    // statelessly -- the same thing on an ordered list, NOT on such a one: refuse it loudly.
    // (Every TLI reader output is ordered; the reference itself produces an unordered list only
    // from a TLI FILE holding several databases, whose isotope ids it confuses:
    // line_by_line.py:114-119, fixture G16 `onefile`.)
    std::vector<double> last_wn((size_t)niso, -HUGE_VAL);
    for (int64_t ln = 0; ln < nlines; ln++) {
        const int i = lid_h[ln];
        if (i < 0 || i >= niso) {
            pb::set_error("pb_lines_create: line %lld has isotope id %d outside [0,%d)",
                          (long long)ln, i, niso);
            pb_lines_destroy(l);
            return PB_ERR_ARG;
        }
        const double v = lwn_h[ln];
        if (v < lo || v > hi)
            continue;
        if (v < last_wn[(size_t)i]) {
            pb::set_error("pb_lines_create: line %lld (%.6f cm-1) of isotope %d comes after a "
                          "line at %.6f cm-1: the lines of an isotope must be in ascending "
                          "wavenumber order (the reference's Doppler-index search is one-way, "
                          "_extcoeff.c:278; its result on such a list is order-dependent)",
                          (long long)ln, v, i, last_wn[(size_t)i]);
            pb_lines_destroy(l);
            return PB_ERR_ARG;
        }
        last_wn[(size_t)i] = v;
        l->ninrange++;
        // nearest fine-grid index (_extcoeff.c:243-245)
        int64_t iown = (int64_t)((v - lo) / step);
        if (iown + 1 < onwn && fabs(v - own_at(iown + 1)) < fabs(v - own_at(iown)))
            iown++;
        Group g{(int32_t)ln, 1, (int32_t)iown, i};
        const double centre = own_at(iown);
        // greedy co-adding of the following lines of the same isotope (:248-262)
        while (ln + 1 != nlines && lid_h[ln + 1] == i && lwn_h[ln + 1] <= hi) {
            if (fabs(lwn_h[ln + 1] - centre) < step) {
                ln++;
                last_wn[(size_t)i] = std::max(last_wn[(size_t)i], lwn_h[ln]);
                g.count++;
                l->nadd++;
                l->ninrange++;
            } else
                break;
        }
        groups.push_back(g);
    }
    // (isotope, fine index) order; stable, so a sorted TLI keeps its file order
    std::stable_sort(groups.begin(), groups.end(), [](const Group &x, const Group &y) {
        return x.iso != y.iso ? x.iso < y.iso : x.iown < y.iown;
    });
    l->ngroups = (int64_t)groups.size();
    l->iso_gstart.assign((size_t)niso + 1, 0);
    for (const Group &g : groups)
        l->iso_gstart[(size_t)g.iso + 1]++;
    for (int i = 0; i < niso; i++)
        l->iso_gstart[(size_t)i + 1] += l->iso_gstart[(size_t)i];
    std::vector<int32_t> &gfirst = l->h_gfirst, &gcount = l->h_gcount, &giown = l->h_giown;
    gfirst.resize(groups.size());
    gcount.resize(groups.size());
    giown.resize(groups.size());
    std::vector<int32_t> giso(groups.size());
    for (size_t k = 0; k < groups.size(); k++) {
        gfirst[k] = groups[k].first;
        gcount[k] = groups[k].count;
        giown[k] = groups[k].iown;
        giso[k] = groups[k].iso;
    }
    if (rc == PB_OK) rc = upload(&l->d_gfirst, gfirst.data(), gfirst.size());
    if (rc == PB_OK) rc = upload(&l->d_gcount, gcount.data(), gcount.size());
    if (rc == PB_OK) rc = upload(&l->d_giown, giown.data(), giown.size());
    if (rc == PB_OK) rc = upload(&l->d_giso, giso.data(), giso.size());
    }
    if (rc == PB_OK) rc = upload(&l->d_iso_gstart, l->iso_gstart.data(), l->iso_gstart.size());
    if (rc != PB_OK) {
        pb_lines_destroy(l);
        return rc;
    }
    *out = l;
    return PB_OK;
}

int pb_lines_grouped_on_device(const pb_lines *l, int *flag)
{
    PB_REQUIRE(l && flag, "pb_lines_grouped_on_device: null pointer");
    *flag = l->grouped_on_device;
    return PB_OK;
}

int pb_lines_groups(const pb_lines *l, int32_t *first_h, int32_t *count_h, int32_t *iown_h,
                    int64_t *iso_gstart_h)
{
    PB_REQUIRE(l, "pb_lines_groups: null handle");
    const size_t n = (size_t)l->ngroups;
    if (first_h)
        std::copy(l->h_gfirst.begin(), l->h_gfirst.begin() + n, first_h);
    if (count_h)
        std::copy(l->h_gcount.begin(), l->h_gcount.begin() + n, count_h);
    if (iown_h)
        std::copy(l->h_giown.begin(), l->h_giown.begin() + n, iown_h);
    if (iso_gstart_h)
        std::copy(l->iso_gstart.begin(), l->iso_gstart.end(), iso_gstart_h);
    return PB_OK;
}

int pb_lines_stats(const pb_lines *l, int64_t stats[3])
{
    PB_REQUIRE(l && stats, "pb_lines_stats: null pointer");
    stats[0] = l->ninrange;
    stats[1] = l->ngroups;
    stats[2] = l->nadd;
    return PB_OK;
}

void pb_lines_destroy(pb_lines *l)
{
    if (!l)
        return;
    (void)hipFree(l->d_lwn);
    (void)hipFree(l->d_elow);
    (void)hipFree(l->d_gf);
    (void)hipFree(l->d_lid);
    (void)hipFree(l->d_gfirst);
    (void)hipFree(l->d_gcount);
    (void)hipFree(l->d_giown);
    (void)hipFree(l->d_giso);
    (void)hipFree(l->d_iso_gstart);
    delete l;
}

// ---------------------------------------------------------------------------
// LBL plan
// ---------------------------------------------------------------------------
int pb_lbl_create(pb_lbl **out, pb_voigt *voigt, pb_lines *lines, const double *wn_h,
                  int nwave, const int32_t *divisors_h, int ndivs, const double *molrad_h,
                  const double *molmass_h, int nmol, const int32_t *isoimol_h,
                  const double *isomass_h, const double *isoratio_h,
                  const int32_t *isoiext_h, int niso, double cutoff, double ethresh,
                  int resolution, int max_layers)
{
    PB_REQUIRE(out, "pb_lbl_create: null out");
    *out = nullptr;
    PB_REQUIRE(voigt && lines && wn_h && divisors_h && molrad_h && molmass_h && isoimol_h &&
                   isomass_h && isoratio_h && isoiext_h,
               "pb_lbl_create: null pointer");
    PB_REQUIRE(nwave >= 2 && ndivs >= 1 && nmol >= 1 && niso >= 1 && max_layers >= 1,
               "pb_lbl_create: bad sizes");
    PB_REQUIRE(lines->niso == niso, "pb_lbl_create: line list has %d isotopes, got %d",
               lines->niso, niso);
    for (int i = 0; i < niso; i++)
        PB_REQUIRE(isoimol_h[i] >= 0 && isoimol_h[i] < nmol,
                   "pb_lbl_create: isoimol[%d] out of range", i);
    PB_REQUIRE(divisors_h[0] >= 1, "pb_lbl_create: divisors must start at >= 1");
    PB_REQUIRE(resolution || voigt->d_pm || voigt->lazy_parent,
               "pb_lbl_create: this Voigt table keeps the reference layout only (keep_flat = 2); "
               "constant-step plans need the phase-major layout");
    const double wnstep = wn_h[1] - wn_h[0];
    if (!resolution) {
        // the kept samples of every admissible dynamic grid must be osamp apart
        for (int d = 0; d < ndivs; d++) {
            const int scale = (int)round(wnstep / lines->ownstep / divisors_h[d]);
            if ((int64_t)scale * divisors_h[d] != voigt->osamp) {
                pb::set_error("pb_lbl_create: wn step %.9g is not osamp=%d fine steps of "
                              "%.9g for divisor %d",
                              wnstep, voigt->osamp, lines->ownstep, divisors_h[d]);
                return PB_ERR_UNSUPPORTED;
            }
        }
    }
    if (!resolution && lines->onwn >= (1LL << 30)) {
        pb::set_error("pb_lbl_create: fine grid of %lld samples exceeds 2^30", (long long)lines->onwn);
        return PB_ERR_UNSUPPORTED;
    }
    // the gather kernel addresses one Lorentz row of the table with 32-bit offsets
    for (int m = 0; m < voigt->nlor; m++) {
        const size_t k0 = (size_t)m * voigt->ndop, k1 = k0 + voigt->ndop - 1;
        const int64_t span = voigt->pm_base[k1] + (int64_t)voigt->pm_stride[k1] * voigt->osamp -
                             voigt->pm_base[k0] + 4 * (int64_t)kPmPad;
        if (span >= 4294967296LL) {
            pb::set_error("pb_lbl_create: Lorentz row %d of the Voigt table spans %lld "
                          "samples (> 2^32)", m, (long long)span);
            return PB_ERR_UNSUPPORTED;
        }
    }
    pb_lbl *p = new (std::nothrow) pb_lbl();
    if (!p)
        return PB_ERR_NOMEM;
    p->voigt = voigt;
    p->lines = lines;
    p->nwave = nwave;
    p->nmol = nmol;
    p->niso = niso;
    p->ndivs = ndivs;
    p->max_layers = max_layers;
    p->resolution = resolution ? 1 : 0;
#ifdef PB_EXPERIMENTS
    if (const char *e = getenv("PB_RES_DYN_PREDICT"))
        p->dyn_predict = resolution && atoi(e) == 1 ? 1 : 0;
#endif
    p->cutoff = cutoff;
    p->ethresh = ethresh;
    p->wnstep = wnstep;
    p->wn0 = wn_h[0];
    p->isoiext.assign(isoiext_h, isoiext_h + niso);
    if (resolution) {
        p->h_wn.assign(wn_h, wn_h + nwave);
        p->h_divisors.assign(divisors_h, divisors_h + ndivs);
        p->h_molrad.assign(molrad_h, molrad_h + nmol);
        p->h_molmass.assign(molmass_h, molmass_h + nmol);
        p->h_isoimol.assign(isoimol_h, isoimol_h + niso);
        p->h_isoiext0.assign(isoiext_h, isoiext_h + niso);
        p->h_isomass.assign(isomass_h, isomass_h + niso);
        p->h_isoratio.assign(isoratio_h, isoratio_h + niso);
    }
    int rows = 1;
    for (int i = 0; i < niso; i++)
        rows = std::max(rows, isoiext_h[i] + 1);
    p->nrows_sep = rows;
    p->kmax_rows = rows;
    const size_t L = (size_t)max_layers, LI = L * (size_t)niso;
    int rc = PB_OK;
    if (rc == PB_OK) rc = upload(&p->d_wn, wn_h, (size_t)nwave);
    if (rc == PB_OK) rc = upload(&p->d_divisors, divisors_h, (size_t)ndivs);
    if (rc == PB_OK) rc = upload(&p->d_molrad, molrad_h, (size_t)nmol);
    if (rc == PB_OK) rc = upload(&p->d_molmass, molmass_h, (size_t)nmol);
    if (rc == PB_OK) rc = upload(&p->d_isoimol, isoimol_h, (size_t)niso);
    if (rc == PB_OK) rc = upload(&p->d_isomass, isomass_h, (size_t)niso);
    if (rc == PB_OK) rc = upload(&p->d_isoratio, isoratio_h, (size_t)niso);
    if (rc == PB_OK) rc = upload(&p->d_isoiext, isoiext_h, (size_t)niso);
    auto alloc = [&](void **ptr, size_t bytes) {
        if (rc == PB_OK && hipMalloc(ptr, bytes) != hipSuccess) {
            pb::set_error("pb_lbl_create: workspace allocation failed");
            rc = PB_ERR_NOMEM;
        }
    };
    alloc((void **)&p->ls_ofactor, L * 4);
    alloc((void **)&p->ls_scale, L * 4);
    alloc((void **)&p->ls_dnwn, L * 8);
    alloc((void **)&p->ls_dwnstep, L * 8);
    alloc((void **)&p->ls_quot, 4 * L * 8);
    alloc((void **)&p->li_invz, LI * 8);
    alloc((void **)&p->li_alphad, LI * 8);
    alloc((void **)&p->li_dens, LI * 8);
    alloc((void **)&p->li_z, LI * 8);
    alloc((void **)&p->li_ilor, LI * 4);
    alloc((void **)&p->li_hmax, LI * 4);
    alloc((void **)&p->li_rowmax, LI * 4);
    alloc((void **)&p->li_hlo, LI * 4);
    alloc((void **)&p->li_hhi, LI * 4);
    alloc((void **)&p->kmax_bits, L * (size_t)rows * 8);
    alloc((void **)&p->ls_resident, L * 4);
    alloc((void **)&p->ls_block, L * 4);
    alloc((void **)&p->ls_wave, L * 4);
    // the whole buffer is what a multi-GPU run all-reduces (pb_lbl_kmax_buffer): slots beyond the
    // rows of a call must not hold whatever the allocation did
    if (rc == PB_OK && hipMemset(p->kmax_bits, 0, L * (size_t)rows * 8) != hipSuccess)
        rc = PB_ERR_HIP;
    if (rc == PB_OK && !resolution) {
        // groups re-sorted by (isotope, iown mod osamp, iown): all lines that read the same
        // phase row of a profile become neighbours (k_ext_staged)
        const int osamp = voigt->osamp;
        const size_t ng = lines->h_giown.size();
        // counting sort by phase inside every isotope (stable: positions stay ascending); a
        // comparison sort with a modulo in its comparator took 1.3 s at 1e7 lines
        std::vector<int32_t> order(ng);
        std::vector<int64_t> start((size_t)niso * (osamp + 1) + 1, 0);
        for (int i = 0; i < niso; i++) {
            const int64_t s0 = lines->iso_gstart[i], s1 = lines->iso_gstart[i + 1];
            std::vector<int64_t> cnt((size_t)osamp + 1, 0);
            for (int64_t k = s0; k < s1; k++)
                cnt[(size_t)(lines->h_giown[k] % osamp) + 1]++;
            int64_t run = s0;
            for (int ph = 0; ph <= osamp; ph++) {
                run += cnt[ph];
                start[(size_t)i * (osamp + 1) + ph] = run;
            }
            // start[ph] = first slot AFTER phase ph-1 ... re-derive the first slot of each phase
            std::vector<int64_t> slot((size_t)osamp, 0);
            int64_t at = s0;
            for (int ph = 0; ph < osamp; ph++) {
                slot[ph] = at;
                at += cnt[(size_t)ph + 1];
            }
            for (int64_t k = s0; k < s1; k++)
                order[(size_t)slot[(size_t)(lines->h_giown[k] % osamp)]++] = (int32_t)k;
        }
        start[(size_t)niso * (osamp + 1)] = (int64_t)ng;
        std::vector<int32_t> f(ng), c(ng), w(ng);
        for (size_t k = 0; k < ng; k++) {
            f[k] = lines->h_gfirst[order[k]];
            c[k] = lines->h_gcount[order[k]];
            w[k] = lines->h_giown[order[k]];
        }
        {
            std::vector<int32_t> inv(ng);                 // position-sorted group -> phase-sorted slot
            for (size_t k = 0; k < ng; k++)
                inv[(size_t)order[k]] = (int32_t)k;
            if (rc == PB_OK) rc = upload(&p->pos2ph, inv.data(), ng);
        }
        if (rc == PB_OK) rc = upload(&p->ph_first, f.data(), ng);
        if (rc == PB_OK) rc = upload(&p->ph_count, c.data(), ng);
        if (rc == PB_OK) rc = upload(&p->ph_iown, w.data(), ng);
        if (rc == PB_OK) rc = upload(&p->ph_start, start.data(), start.size());
        p->h_ph_iown = w;
        p->h_ph_start = start;
        {
            // position index of every (isotope, phase) run, one entry per kBinSamples samples
            const int nbins = (int)pb::div_up((int64_t)nwave, (int64_t)kBinSamples) + 1;
            const int64_t binw = (int64_t)kBinSamples * osamp;
            std::vector<int32_t> bins((size_t)niso * osamp * ((size_t)nbins + 1));
            for (int i = 0; i < niso; i++)
                for (int ph = 0; ph < osamp; ph++) {
                    int64_t k = start[(size_t)i * (osamp + 1) + ph];
                    const int64_t kend = start[(size_t)i * (osamp + 1) + ph + 1];
                    int32_t *row = bins.data() + ((size_t)i * osamp + ph) * ((size_t)nbins + 1);
                    for (int b = 0; b < nbins; b++) {
                        while (k < kend && (int64_t)w[(size_t)k] < b * binw)
                            k++;
                        row[b] = (int32_t)k;
                    }
                    row[nbins] = (int32_t)kend;
                }
            p->ph_nbins = nbins;
            if (rc == PB_OK) rc = upload(&p->ph_bin, bins.data(), bins.size());
        }
        {
            std::vector<int32_t> iso_of(ng);
            for (int i = 0; i < niso; i++)
                for (int64_t k = lines->iso_gstart[i]; k < lines->iso_gstart[i + 1]; k++)
                    iso_of[(size_t)k] = i;
            if (rc == PB_OK) rc = upload(&p->ph_iso, iso_of.data(), ng);
        }
        {
            // leader line of every group in both walk orders (coalesced reads in k_records)
            const std::vector<double> &lwn_all = lines->h_lwn, &elow_all = lines->h_elow,
                                      &gf_all = lines->h_gf;
            std::vector<double> lead(3 * ng), glead(3 * ng);
            for (size_t k = 0; k < ng; k++) {
                const int32_t lf = f[k], gf_ = lines->h_gfirst[k];
                lead[k] = lwn_all[lf];
                lead[ng + k] = elow_all[lf];
                lead[2 * ng + k] = gf_all[lf];
                glead[k] = lwn_all[gf_];
                glead[ng + k] = elow_all[gf_];
                glead[2 * ng + k] = gf_all[gf_];
            }
            if (rc == PB_OK) rc = upload(&p->ph_lead, lead.data(), lead.size());
            if (rc == PB_OK) rc = upload(&p->g_lead, glead.data(), glead.size());
        }
        // per (layer, group) records of k_records: allocated on first use
        int cap = 0, smallest = INT_MAX;
        for (int32_t st : voigt->pm_stride) {
            cap = std::max(cap, st);
            smallest = std::min(smallest, st);
        }
        p->rowcap = cap;
        // resident-profile kernel: 64 KiB of LDS for one cell's phase-major block
        if (ng > 0 && (int64_t)smallest * osamp <= kResCapDefault)
            p->res_cap = kResCapDefault;
        if (const char *e = getenv("PB_RESIDENT_CAP"))
            p->res_cap = std::max(0, std::min(atoi(e), 19000));
        if (p->res_cap > 0) {
            // first position-sorted group of every isotope at or after each output sample
            std::vector<int32_t> gs((size_t)niso * ((size_t)nwave + 1));
            for (int i = 0; i < niso; i++) {
                int64_t g = lines->iso_gstart[i];
                const int64_t gend = lines->iso_gstart[i + 1];
                for (int64_t w = 0; w <= nwave; w++) {
                    while (g < gend && (int64_t)lines->h_giown[(size_t)g] < w * osamp)
                        g++;
                    gs[(size_t)i * ((size_t)nwave + 1) + (size_t)w] = (int32_t)g;
                }
            }
            if (rc == PB_OK) rc = upload(&p->gs_start, gs.data(), gs.size());
        }
    }
    {
        const char *e = getenv("PB_GATHER");
        if (e && !strcmp(e, "global"))
            p->gather_mode = 1;
        else if (e && !strcmp(e, "staged"))
            p->gather_mode = 2;
        else if (e && !strcmp(e, "resident"))
            p->gather_mode = 3;
#ifdef PB_EXPERIMENTS
        else if (e && !strcmp(e, "scatter"))
            p->gather_mode = 4;
        else if (e && !strcmp(e, "rounds"))
            p->gather_mode = 5;
#endif
        const char *t = getenv("PB_STAGE_THRESHOLD");
        if (t)
            p->stage_threshold = atof(t);
    }
    if (rc != PB_OK) {
        pb_lbl_destroy(p);
        return rc;
    }
    *out = p;
    return PB_OK;
}

int pb_lbl_set_isoiext(pb_lbl *p, const int32_t *isoiext_h)
{
    PB_REQUIRE(p && isoiext_h, "pb_lbl_set_isoiext: null pointer");
    for (int i = 0; i < p->niso; i++)
        PB_REQUIRE(isoiext_h[i] < p->kmax_rows,
                   "pb_lbl_set_isoiext: row %d exceeds the %d rows of the plan",
                   isoiext_h[i], p->kmax_rows);
    p->isoiext.assign(isoiext_h, isoiext_h + p->niso);
    PB_HIP(hipMemcpy(p->d_isoiext, isoiext_h, (size_t)p->niso * 4, hipMemcpyHostToDevice));
    return PB_OK;
}

int pb_lbl_set_gather_mode(pb_lbl *p, int mode)
{
    PB_REQUIRE(p && mode >= 0 && mode <= 7, "pb_lbl_set_gather_mode: mode must be 0..7");
#ifndef PB_EXPERIMENTS
    PB_REQUIRE(mode != 4 && mode != 5 && mode != 7,
               "pb_lbl_set_gather_mode: mode %d (scatter / rounds / wave) is a measured dead end "
               "kept out of libpbhip.so: `make -C pyratbay_amd/csrc EXPERIMENTS=1` builds "
               "libpbhip_exp.so with it", mode);
#endif
    PB_REQUIRE(mode != 6 || p->resolution,
               "pb_lbl_set_gather_mode: mode 6 (per-layer dynamic grids) is for `resolution` plans");
    p->gather_mode = mode;
    return PB_OK;
}

int pb_lbl_set_record_budget(pb_lbl *p, int64_t bytes)
{
    PB_REQUIRE(p && bytes >= (int64_t)sizeof(Rec16), "pb_lbl_set_record_budget: bad budget");
    p->record_budget = (size_t)bytes;
    return PB_OK;
}

int pb_lbl_last_chunks(const pb_lbl *p, int *chunks)
{
    PB_REQUIRE(p && chunks, "pb_lbl_last_chunks: null pointer");
    *chunks = p->last_chunks;
    return PB_OK;
}

int pb_lbl_set_concurrency(pb_lbl *p, int n)
{
    PB_REQUIRE(p && n >= 1, "pb_lbl_set_concurrency: n must be >= 1");
    p->concurrency = n;
    return PB_OK;
}

int pb_lbl_last_gather_mode(const pb_lbl *p, int *mode)
{
    PB_REQUIRE(p && mode, "pb_lbl_last_gather_mode: null pointer");
    *mode = p->last_gather;
    return PB_OK;
}

int pb_lbl_set_ethresh(pb_lbl *p, double ethresh)
{
    PB_REQUIRE(p, "pb_lbl_set_ethresh: null handle");
    p->ethresh = ethresh;
    return PB_OK;
}

static int lbl_resolution_dyn(pb_lbl *p, LblArgs &a, double *ext_d, int64_t wbegin, int64_t wcount,
                              const double *temp_d, const double *dens_d, const double *isoz_d,
                              int64_t z_iso_stride, int64_t z_layer_stride, int nlayers, int add,
                              hipStream_t s);

// phase 0: the whole call; 1: up to and including the records, per-row maxima over the shard's
// own groups only (the caller all-reduces them); 2: the gather of the call begun with phase 1
static int lbl_extinction(pb_lbl *p, double *ext_d, int64_t wbegin, int64_t wcount,
                          const double *temp_d, const double *dens_d, const double *isoz_d,
                          int64_t z_iso_stride, int64_t z_layer_stride, int nlayers, int add,
                          void *stream, int phase)
{
    PB_REQUIRE(p, "pb_lbl_extinction: null handle");
    PB_REQUIRE(wcount == 0 || (ext_d && temp_d && dens_d && isoz_d),
               "pb_lbl_extinction: null pointer");
    PB_REQUIRE(nlayers >= 1 && nlayers <= p->max_layers,
               "pb_lbl_extinction: nlayers=%d outside [1,%d]", nlayers, p->max_layers);
    PB_REQUIRE(wbegin >= 0 && wcount >= 0 && wbegin + wcount <= p->nwave,
               "pb_lbl_extinction: shard [%lld,+%lld) outside the %d-sample grid",
               (long long)wbegin, (long long)wcount, p->nwave);
    if (wcount == 0) {
        // an empty shard of a two-phase call still takes part in the all-reduce(MAX) of the
        // per-row maxima: it must contribute zeros, not what its previous call left behind
        if (phase == 1)
            PB_HIP(hipMemsetAsync(p->kmax_bits, 0, (size_t)p->max_layers * p->kmax_rows * 8,
                                  pb::as_stream(stream)));
        return PB_OK;
    }
    const pb_voigt *v = p->voigt;
    const pb_lines *l = p->lines;
    hipStream_t s = pb::as_stream(stream);
    if (p->resolution) {
        int rc = pb_voigt_ensure_flat(p->voigt, s);
        if (rc)
            return rc;
    }
    LblArgs a;
    memset(&a, 0, sizeof(a));
    a.pm = v->d_pm;
    a.flat = v->d_flat;
    a.pm_base = v->d_pm_base;
    a.pm_stride = v->d_pm_stride;
    a.psize = v->d_psize;
    a.pindex = v->d_pindex;
    a.doppler = v->d_doppler;
    a.lorentz = v->d_lorentz;
    a.ndop = v->ndop;
    a.nlor = v->nlor;
    a.osamp = v->osamp;
    a.lwn = l->d_lwn;
    a.elow = l->d_elow;
    a.gf = l->d_gf;
    a.lid = l->d_lid;
    a.gfirst = l->d_gfirst;
    a.gcount = l->d_gcount;
    a.giown = l->d_giown;
    a.iso_gstart = l->d_iso_gstart;
    a.nlines = l->nlines;
    a.ph_first = p->ph_first;
    a.ph_count = p->ph_count;
    a.ph_iown = p->ph_iown;
    a.ph_start = p->ph_start;
    a.rowcap = p->rowcap;
    a.ph_iso = p->ph_iso;
    a.ph_bin = p->ph_bin;
    a.ph_nbins = p->ph_nbins;
    a.g_lead = p->g_lead;
    a.ls_resident = p->ls_resident;
    a.ls_block = p->ls_block;
    a.ls_wave = p->ls_wave;
    a.wave_cap = 0;
    a.gs_start = p->gs_start;
    a.giso = l->d_giso;
    a.ngroups = l->ngroups;
    a.inv_osamp = 1.0 / (double)v->osamp;
    a.molrad = p->d_molrad;
    a.molmass = p->d_molmass;
    a.isoimol = p->d_isoimol;
    a.isoiext = p->d_isoiext;
    a.isomass = p->d_isomass;
    a.isoratio = p->d_isoratio;
    a.divisors = p->d_divisors;
    a.nmol = p->nmol;
    a.niso = p->niso;
    a.ndivs = p->ndivs;
    a.temp = temp_d;
    a.dens = dens_d;
    a.isoz = isoz_d;
    a.z_iso_stride = z_iso_stride;
    a.z_layer_stride = z_layer_stride;
    a.ls_ofactor = p->ls_ofactor;
    a.ls_scale = p->ls_scale;
    a.ls_dnwn = p->ls_dnwn;
    a.ls_dwnstep = p->ls_dwnstep;
    a.ls_cutsteps = p->ls_quot;
    a.ls_inv_ofactor = p->ls_quot + p->max_layers;
    a.ls_inv_scale = p->ls_quot + 2 * (size_t)p->max_layers;
    a.ls_inv_temp = p->ls_quot + 3 * (size_t)p->max_layers;
    a.li_invz = p->li_invz;
    a.li_alphad = p->li_alphad;
    a.li_dens = p->li_dens;
    a.li_z = p->li_z;
    a.li_ilor = p->li_ilor;
    a.li_hmax = p->li_hmax;
    a.li_rowmax = p->li_rowmax;
    a.li_hlo = p->li_hlo;
    a.li_hhi = p->li_hhi;
    a.kmax_bits = p->kmax_bits;
    a.wn = p->d_wn;
    a.own0 = l->own0;
    a.own_last = l->own_last;
    a.ownstep = l->ownstep;
    a.wnstep = p->wnstep;
    a.wn0 = p->wn0;
    a.onwn = l->onwn;
    a.cutoff = p->cutoff;
    a.ethresh = p->ethresh;
    a.add = add ? 1 : 0;
    a.nrows = add ? 1 : p->nrows_sep;
    a.nlayers = nlayers;
    a.nwave = p->nwave;
    a.wbegin = wbegin;
    a.wcount = wcount;
    a.ext = ext_d;
    {
        // records are needed for the groups within reach of the shard only.  The window must hold
        // every group ANY gather kernel may examine: the staged / global / round kernels bracket
        // their candidates by fine position (within min(hmax, cutoff) + osamp + ofactor of a tile),
        // the resident and scatter kernels through the per-sample index gs_start, which examines
        // up to two more output samples' worth of groups on either side -- hence 6 (not 2) x osamp
        // of margin.  A group examined but never written would be whatever the allocation held
        // (round 2: garbage records of a long-lived process sent the resident kernel out of
        // bounds; found by tools/fuzz_pipeline.py); the buffers are also zeroed when allocated, so
        // a record never written is a dead record.
        int64_t hmax_all = 0;
        for (int32_t h : v->psize)
            hmax_all = std::max<int64_t>(hmax_all, h);
        int64_t reach = hmax_all;
        if (a.cutoff > 0.0)
            reach = std::min(reach, (int64_t)(a.cutoff / a.ownstep) + 2 * (int64_t)v->osamp + 2);
        reach += 6 * (int64_t)v->osamp;
        const bool whole = wbegin == 0 && wcount == p->nwave;
        a.rec_flo = whole ? INT64_MIN : wbegin * (int64_t)v->osamp - reach;
        a.rec_fhi = whole ? INT64_MAX : (wbegin + wcount - 1) * (int64_t)v->osamp + reach;
        a.kmax_local = phase != 0 ? 1 : 0;
    }
    {
        const char *e = getenv("PB_EXPERIMENT");
        a.experiment = e ? atoi(e) : 0;
        a.probe = nullptr;
    }

    // Kernel choice (constant-step grids): the LDS-staged kernel when several groups share
    // a (tile, phase) row, else the global gather.  Every kernel adds the terms of a sample in
    // one fixed order, so a call is bitwise reproducible and shards of ONE configuration
    // concatenate exactly; the choice of kernel and the phase split below do depend on the size
    // of the call (shard width, layers), and a different choice changes the association of the
    // per-sample sums: results of different configurations agree to ~1e-13, not bit for bit.
    constexpr int kStagedWaves = 8;
    constexpr int kStagedThreads = kStagedWaves * 64;
    // rows longer than kStageRowMax samples are staged in chunks: (phase, chunk) pairs act as
    // phases; k_records writes one packed record per group, clipped to the chunk by the gather
    const int nch_max = (int)pb::div_up((int64_t)a.rowcap, (int64_t)kChunkRow);
    a.nch_max = std::max(1, nch_max);
    a.rowlds = (std::min(a.rowcap, kStageRowMax) + 1) & ~1;     // even: 16-byte aligned buffers
    const size_t lds_fixed = (size_t)kStagedThreads * (16 + 8 + 4) + kStagedWaves * 12 +
                             (size_t)a.ndop * 16 +
                             (size_t)(2 * v->osamp * a.nch_max + 1) * 4 + 64;
    bool dma = true;                     // rows by LDS-DMA (k_ext_staged); PB_STAGE_DMA=0: via registers
    if (const char *e = getenv("PB_STAGE_DMA"))
        dma = atoi(e) != 0;
    const size_t lds = (2 * ((size_t)a.rowlds + kStagePad) + kStagePad) * 8 + lds_fixed;
    const double per_phase = (double)l->ngroups / std::max(1, p->nwave) * 2048.0 / v->osamp;
    // table cell and window length share 32 bits of a packed record: 20 + 12, for long rows
    // 18 + kLongLenBits (windows of up to 16 383 samples)
    const bool packable = v->nlor * v->ndop < (a.nch_max > 1 ? (1 << (32 - kLongLenBits)) : (1 << 20)) &&
                          a.rowcap < (1 << kLongLenBits);
    // sized for the layers of this call (a layer shard of a multi-GPU run holds few of them)
    const size_t rec16_bytes = (size_t)nlayers * (size_t)l->ngroups * sizeof(Rec16);
    const bool can_stage = !p->resolution && lds <= 160 * 1024 && l->ngroups > 0 &&
                           (a.nch_max == 1 ||
                            (a.nch_max <= 16 && packable && !getenv("PB_NO_LONG_ROWS")));
    // the staged kernel needs enough workgroups to hide its per-segment latency; launches that
    // stay below 750 even when split eight ways go to the global gather with record splitting.
    // Tiling of the staged kernel: S = 2 sub-tiles of 2048 samples per workgroup (1 measured
    // slower at every launch size once the splits are spread over the XCDs; 4 spills: 32
    // accumulators at the 64-register budget) and nsplit workgroups per tile, each with its
    // share of the phases.  A launch ends when its slowest workgroup does, so small launches
    // and launches of long-running workgroups are split further:
    //  * light tiles (C2: ~8 records per phase row): aim for ~2000 workgroups.  Layer shards
    //    of C2 (a sweep of round 2): 40 layers       0.70 ms unsplit / 0.65 in two; 20 layers
    //    0.40 in two / 0.37 in four; 10 layers 0.23 in four or eight; all 80 layers 1.19
    //    unsplit / 1.20 in two.
    //  * heavy tiles (>= 64 groups per phase row and 2048 samples: 1e6 lines on 1e5 samples;
    //    a workgroup then runs for milliseconds): aim for ~8000.  1e6 lines, 80 layers:
    //    8.13 ms unsplit, 7.33 in four; a 10-layer shard 2.03 -> 1.13 in eight.
    // The partial sums stay below 1 GB.
    const int64_t sub = kStagedWaves * kStageSpan;
    const int64_t blocks2 = pb::div_up(wcount, 2 * sub) * (int64_t)nlayers;
    // With other spectra in flight on other streams (pb_lbl_set_concurrency: the walkers of a
    // retrieval, the pipelined shards of a multi-GPU rank) a launch need not fill the chip by
    // itself: one round of workgroups (~1000) amortises the per-workgroup candidate search
    // better.  A 1/8 wavenumber shard of C2, two-phase, per spectrum: one at a time 0.269-0.277
    // ms whatever the split; two in flight 0.211 (3 workgroups per 4096-sample tile = 960, one round; 4 per tile: 0.232) against
    // 0.240 (7 per 2048-sample tile, the one-at-a-time rule); three in flight 0.190 against 0.236.
    const bool shared_chip = p->concurrency > 1 && per_phase < 64.0;
    int S = 2;
    int nsplit = (int)std::min<int64_t>(
        8, pb::div_up((int64_t)(per_phase >= 64.0 ? 8000 : 2000), std::max<int64_t>(1, blocks2)));
    if (shared_chip) {
        // the ~2000 workgroups of the rule above, counted over ALL the spectra in flight.  (Until
        // the rank loop stopped being host-bound -- second session of round 3 -- one round of 1024
        // per launch measured best; with the host out of the way, a 1/8 shard of C2 with three in
        // flight: 1 / 2 / 3 / 4 pieces per tile 0.211 / 0.163 / 0.179 / 0.181 ms per spectrum, a
        // 1/4 shard 0.276 / 0.279 / 0.291, a 1/2 shard 0.467 / 0.502.)
        const int64_t inflight = blocks2 * std::max(1, p->concurrency);
        nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(8, (2000 + inflight / 2) / std::max<int64_t>(1, inflight)));
    }
    {
        const int64_t plane = (int64_t)nlayers * a.nrows * wcount * 8;
        while (nsplit > 1 && (nsplit - 1) * plane > ((int64_t)1 << 30))
            nsplit--;
    }
    {
        // tile quantisation of a narrow shard: 12 500 samples are 3.05 tiles of 4096 (31 % of
        // the workgroups' rows staged for nothing) but 6.1 of 2048 (15 %); one span per
        // workgroup when that saves more than 12 % (N = 8 shards of C2: 0.305 -> 0.297 ms, of the
        // 1e6-line list 1.47 -> 1.28 ms; N = 4: the two-span tile stays faster).  The sums do not
        // depend on S (tests/test_gpu_extinction.py::test_staged_variants_agree).
        const double w2 = (double)pb::div_up(wcount, 2 * sub) * 2 * sub / (double)wcount;
        const double w1 = (double)pb::div_up(wcount, sub) * sub / (double)wcount;
        if (w2 - w1 > 0.12 && !shared_chip)
            S = 1;
    }
    if (const char *e = getenv("PB_STAGE_S"))
        S = atoi(e) >= 4 ? 4 : atoi(e) >= 2 ? 2 : 1;
    if (const char *e = getenv("PB_STAGE_SPLIT"))
        nsplit = std::max(1, std::min(8, atoi(e)));
    // (with other spectra in flight the launches of all of them count towards filling the chip)
    const bool enough_blocks = pb::div_up(wcount, S * sub) * (int64_t)nlayers * nsplit *
                                   (shared_chip ? p->concurrency : 1) >= 750;
    // round-staged kernel (pb_rounds.hip): rows of one piece (<= 1024 samples), packed records
    const bool rounds = kExp && can_stage && a.nch_max == 1 && packable && p->gather_mode == 5;
    const bool staged = can_stage && (p->gather_mode == 2 || p->gather_mode == 7 || rounds ||
                                      (p->gather_mode == 0 && enough_blocks &&
                                       per_phase >= p->stage_threshold));
    const bool use_records = !p->resolution && l->ngroups > 0;
    // layers with narrow profiles go to the resident-profile kernel (decided per layer on
    // the device, from the layer alone); the kernel chosen above computes the others
    const bool scatter = kExp && use_records && p->gs_start && p->gather_mode == 4;
    bool resident = use_records && !scatter && p->res_cap > 0 && p->gs_start &&
                    (p->gather_mode == 0 || p->gather_mode == 3);
    bool res_look = false;            // read the layers' decision back after this call
    if (resident && p->gather_mode == 0) {
        if (phase == 2) {
            resident = p->res_on_pending;
            res_look = p->res_look_pending;
        } else {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(s, &cs) == hipSuccess &&
                                   cs != hipStreamCaptureStatusNone;
            const bool probe = !capturing && (p->res_seen < 0 || (p->res_calls++ & 255) == 255);
            if (!probe && p->res_seen == 0)
                resident = false;
            res_look = probe && resident;
            p->res_on_pending = resident;
            p->res_look_pending = res_look;
        }
    }
    a.res_cap = resident ? p->res_cap : 0;
    a.rec32 = nullptr;
    a.rec16 = nullptr;
    const bool poison = getenv("PB_POISON_RECORDS") && atoi(getenv("PB_POISON_RECORDS")) != 0;
    if (!staged || scatter)
        a.nch_max = 1;
    a.grp_lo = 0;
    a.grp_hi = l->ngroups;
    a.rec_pitch = l->ngroups;
    a.key_lo = 0;
    a.key_hi = a.niso * v->osamp;
    a.accumulate = 0;
    // Out-of-core line lists (the reference walks any number of lines one after the other,
    // _extcoeff.c:203-309): when the packed records of all groups exceed the record budget, the
    // phase-sorted group list is cut into chunks of consecutive (isotope, phase) keys that fit.
    size_t budget = p->record_budget;
    if (const char *e = getenv("PB_RECORD_BUDGET"))
        budget = (size_t)atoll(e);
    struct Chunk {
        int key_lo, key_hi;
        int64_t g_lo, g_hi;
    };
    std::vector<Chunk> chunks;
    size_t rec16_need = rec16_bytes;
    if (staged && !scatter && rec16_bytes > budget) {
        if (!packable || rounds || phase != 0 || getenv("PB_REC_SOA")) {
            pb::set_error("pb_lbl_extinction: %zu B of line records exceed the record budget of "
                          "%zu B and this call cannot be chunked (%s)", rec16_bytes, budget,
                          phase != 0 ? "two-phase shard call"
                                     : rounds ? "round gather" : "records are not packable");
            return PB_ERR_NOMEM;
        }
        const int osamp = v->osamp;
        const int64_t gmax = (int64_t)(budget / ((size_t)nlayers * sizeof(Rec16)));
        Chunk c{0, 0, 0, 0};
        int64_t biggest = 0;
        for (int i = 0; i < a.niso; i++)
            for (int ph = 0; ph < osamp; ph++) {
                const int64_t g0 = p->h_ph_start[(size_t)i * (osamp + 1) + ph];
                const int64_t g1 = p->h_ph_start[(size_t)i * (osamp + 1) + ph + 1];
                if (g1 - g0 > gmax) {
                    pb::set_error("pb_lbl_extinction: the %lld groups of one (isotope, phase) key "
                                  "need more than the record budget of %zu B",
                                  (long long)(g1 - g0), budget);
                    return PB_ERR_NOMEM;
                }
                if (g1 - c.g_lo > gmax) {              // close the chunk before this key
                    chunks.push_back(c);
                    biggest = std::max(biggest, c.g_hi - c.g_lo);
                    c.key_lo = i * osamp + ph;
                    c.g_lo = g0;
                }
                c.key_hi = i * osamp + ph + 1;
                c.g_hi = g1;
            }
        chunks.push_back(c);
        biggest = std::max(biggest, c.g_hi - c.g_lo);
        rec16_need = (size_t)std::max<int64_t>(1, biggest) * nlayers * sizeof(Rec16);
    }
    const bool chunked = !chunks.empty();
    p->last_chunks = (int)chunks.size();
    if (staged && !scatter && packable && (a.nch_max > 1 || chunked || !getenv("PB_REC_SOA"))) {
        const size_t rec16_bytes = rec16_need;       // (shadows the whole-list size)
        if (rec16_bytes > p->rec16_alloc) {
            if (p->rec16) {
                PB_HIP(hipStreamSynchronize(s));       // an earlier call may still read it
                (void)hipFree(p->rec16);
                p->rec16 = nullptr;
                p->rec16_alloc = 0;
            }
            if (hipMalloc(&p->rec16, rec16_bytes) != hipSuccess) {
                pb::set_error("pb_lbl_extinction: cannot allocate %zu B of line records",
                              rec16_bytes);
                return PB_ERR_NOMEM;
            }
            p->rec16_alloc = rec16_bytes;
            PB_HIP(hipMemsetAsync(p->rec16, 0, rec16_bytes, s));
            if (poison)
                k_poison_records<<<1024, kBlock, 0, s>>>(p->rec16, (int64_t)(rec16_bytes / sizeof(Rec16)),
                                                        nullptr, 0, nullptr, nullptr, 0);
        }
        a.rec16 = p->rec16;
    }
    // Layers of short phase rows (<= kWvRowMax samples: the Doppler-core layers) go to the
    // wave-autonomous kernel (pb_wave.hip), decided per layer on the device by k_layer_state; the
    // staged kernel computes the others.  Mode 7 selects the pair; modes 0 and 2 keep to the staged
    // kernel alone.  Chunked line lists continue running sums in the staged kernel's order and
    // keep to it.
#ifdef PB_EXPERIMENTS
    {
        // (measured at C2, round 4: the pair takes 1.20 ms per extinction against 1.05 ms for the
        // staged kernel alone -- profiles/r04_gather_wave.md -- so mode 0 does not use it unless
        // PB_WAVE=1 asks for it)
        bool wave_on = staged && !rounds && !scatter && !chunked && a.rec16 != nullptr &&
                       p->gather_mode == 7;
        if (const char *e = getenv("PB_WAVE"))
            wave_on = staged && !rounds && !scatter && !chunked && a.rec16 != nullptr &&
                      (p->gather_mode == 0 || p->gather_mode == 7) && atoi(e) != 0;
        if (wave_on && wave_lds(a) <= 160 * 1024)
            a.wave_cap = kWvRowMax;
    }
#endif  // PB_EXPERIMENTS
    a.nsplit = 1;
    a.part = nullptr;
    a.wm_lo[0] = a.wm_lo[1] = a.wm_off[0] = a.wm_off[1] = nullptr;
    a.wm_n[0] = a.wm_n[1] = 0;
    a.wm_total[0] = a.wm_total[1] = 0;
#ifdef PB_EXPERIMENTS
    if (scatter) {
        if (!p->rec32) {
            const size_t n = (size_t)p->max_layers * (size_t)l->ngroups;
            if (hipMalloc(&p->rec32, n * sizeof(Rec32)) != hipSuccess) {
                pb::set_error("pb_lbl_extinction: cannot allocate %zu B of line records",
                              n * sizeof(Rec32));
                return PB_ERR_NOMEM;
            }
            PB_HIP(hipMemsetAsync(p->rec32, 0, n * sizeof(Rec32), s));
            if (poison)
                k_poison_records<<<1024, kBlock, 0, s>>>(nullptr, 0, p->rec32, (int64_t)n, nullptr,
                                                        nullptr, 0);
        }
        a.rec32 = p->rec32;
    }
#endif  // PB_EXPERIMENTS
    if (chunked)
        a.res_cap = 0;                   // every layer through the staged gather
    // the SoA records serve the global gather, the resident layers and PB_REC_SOA
    const bool need_soa = use_records && !scatter && (a.rec16 == nullptr || a.res_cap > 0);
    if (need_soa && !p->rec_k) {
        const size_t n = (size_t)p->max_layers * (size_t)l->ngroups;
        if (hipMalloc(&p->rec_k, n * 8) != hipSuccess ||
            hipMalloc(&p->rec_i32, n * 4 * 5) != hipSuccess) {
            pb::set_error("pb_lbl_extinction: cannot allocate %zu B of line records", n * 28);
            return PB_ERR_NOMEM;
        }
        PB_HIP(hipMemsetAsync(p->rec_k, 0, n * 8, s));
        PB_HIP(hipMemsetAsync(p->rec_i32, 0, n * 4 * 5, s));
        if (poison)
            k_poison_records<<<1024, kBlock, 0, s>>>(nullptr, 0, nullptr, 0, p->rec_k, p->rec_i32,
                                                    (int64_t)n);
    }
    if (use_records) {
        const size_t n = (size_t)p->max_layers * (size_t)l->ngroups;
        a.rec_k = p->rec_k;
        a.rec_ulo = p->rec_i32;
        a.rec_uhi = p->rec_i32 + n;
        a.rec_q = p->rec_i32 + 2 * n;
        a.rec_cell = p->rec_i32 + 3 * n;
        a.rec_phi = p->rec_i32 + 4 * n;
        // records are laid out in the order the chosen gather kernel walks the groups
        a.rk_first = staged ? p->ph_first : l->d_gfirst;
        a.rk_count = staged ? p->ph_count : l->d_gcount;
        a.rk_iown = staged ? p->ph_iown : l->d_giown;
        a.rk_iso = staged ? p->ph_iso : l->d_giso;
        const double *lead = staged ? p->ph_lead : p->g_lead;
        a.rk_lwn = lead;
        a.rk_elow = lead + l->ngroups;
        a.rk_gf = lead + 2 * l->ngroups;
    }
    a.use_records = use_records ? 1 : 0;

    bool dyn_fall = false;
    // (a fine grid shorter than two steps of the coarsest dynamic grid has no constant-step form)
    if (p->resolution && p->gather_mode == 6 && phase == 0 && l->ngroups > 0 &&
        l->onwn > 2 * (int64_t)v->osamp &&
        !(getenv("PB_RES_DYN") && atoi(getenv("PB_RES_DYN")) == 0)) {
        const int rc = lbl_resolution_dyn(p, a, ext_d, wbegin, wcount, temp_d, dens_d, isoz_d,
                                          z_iso_stride, z_layer_stride, nlayers, add, s);
        // a re-cut table row that cannot be addressed (pb_voigt_ensure_rows) before any run has
        // added to ext: the direct gather below computes the call instead
        if (!(rc == PB_ERR_UNSUPPORTED && p->dyn_runs == 0)) {
            if (rc != PB_OK || !p->dyn_fallback)
                return rc;
            // a call planned from a prediction: the layers the plan did not fit (none, as a
            // rule: the launches below then end at once) go through the direct gather
            dyn_fall = true;
            a.lskip = p->d_ok;
        }
    }
    if (phase != 2 && !dyn_fall) {
    k_layer_state<<<nlayers, 64, ((size_t)a.nlor + a.ndop) * 8 + (size_t)a.ndivs * 4, s>>>(a);
    PB_LAUNCH_CHECK();
    }
    if (chunked) {
        // pass 1: the per-row maxima over ALL lines (the threshold of every chunk's gather);
        // pass 2: per chunk, the records of its groups, then the gather, which continues the
        // running sums of the earlier chunks.  One workgroup per tile (no phase split): the sums
        // of a sample are then exactly those of a single launch over every key.
        {
            const int lines_per_block = 4096;
            dim3 grid(pb::div_up(l->nlines, lines_per_block), nlayers);
            k_kmax<<<grid, kBlock, (size_t)a.nrows * 8, s>>>(a, lines_per_block);
            PB_LAUNCH_CHECK();
        }
        const bool timed = p->ev_used + 2 <= (int)p->ev.size();
        if (timed)
            PB_HIP(hipEventRecord(p->ev[p->ev_used], s));
        const int per = kRecLayers;
        const size_t rlds = (size_t)per * a.nrows * 8 + (size_t)a.ndop * 8 +
                            (size_t)per * a.niso * (8 + 8 + 8 + 4) + (size_t)a.niso * (8 + 4) + 16 + 8;
        PB_REQUIRE(rlds <= 64 * 1024, "pb_lbl_extinction: %zu B of LDS for the record kernel", rlds);
        a.wm_lds = 0;
        a.nsplit = 1;
        a.part = nullptr;
        a.ntiles = pb::div_up(wcount, S * sub);
        const int unit_groups = (nlayers + 7) / 8;
        dim3 ggrid((unsigned)(8 * a.ntiles * unit_groups), a.nrows);
        void (*kern)(LblArgs) =
            dma ? (S == 4   ? k_ext_staged<kStagedWaves, 4, true>
                   : S == 2 ? k_ext_staged<kStagedWaves, 2, true>
                            : k_ext_staged<kStagedWaves, 1, true>)
                : (S == 4   ? k_ext_staged<kStagedWaves, 4, false>
                   : S == 2 ? k_ext_staged<kStagedWaves, 2, false>
                            : k_ext_staged<kStagedWaves, 1, false>);
        if (lds > 64 * 1024)
            PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (size_t c = 0; c < chunks.size(); c++) {
            a.grp_lo = chunks[c].g_lo;
            a.grp_hi = chunks[c].g_hi;
            a.rec_pitch = std::max<int64_t>(1, a.grp_hi - a.grp_lo);
            a.key_lo = chunks[c].key_lo;
            a.key_hi = chunks[c].key_hi;
            a.accumulate = c > 0 ? 1 : 0;
            if (a.grp_hi > a.grp_lo) {
                dim3 rgrid(pb::div_up(a.grp_hi - a.grp_lo, kBlock), pb::div_up(nlayers, per));
                if (a.nch_max > 1)
                    k_records<2, kRecLayers><<<rgrid, kBlock, rlds, s>>>(a);
                else
                    k_records<1, kRecLayers><<<rgrid, kBlock, rlds, s>>>(a);
                PB_LAUNCH_CHECK();
            }
            kern<<<ggrid, kStagedThreads, lds, s>>>(a);
            PB_LAUNCH_CHECK();
        }
        p->last_gather = 2;
        p->last_args = a;
        p->last_packed = false;
        if (timed) {
            PB_HIP(hipEventRecord(p->ev[p->ev_used + 1], s));
            p->ev_used += 2;
        }
        return PB_OK;
    }
    if (phase == 2) {
        // records and maxima are in place
    } else if (use_records) {
        // (one layer per thread for launches of few layers measured slower: 10 layers of C2
        // 49 us against 27 us with four; PB_REC_LAYERS=1 selects it)
        const int per = getenv("PB_REC_LAYERS") && atoi(getenv("PB_REC_LAYERS")) == 1 ? 1 : kRecLayers;
        int64_t rec_threads = l->ngroups;
        if (a.kmax_local && a.rec_flo != INT64_MIN && !p->h_ph_iown.empty() &&
            !getenv("PB_NO_WINDOW_MAP")) {
            // two-phase shard call: only the groups within reach of the shard get a thread
            const int osamp = v->osamp, niso = a.niso;
            if (p->wm_flo != a.rec_flo || p->wm_fhi != a.rec_fhi || p->wm_staged != (int)staged) {
                const int n0 = staged ? niso * osamp : niso, n1 = niso;
                std::vector<int32_t> &h = p->h_wm;
                h.assign((size_t)2 * n0 + 1 + 2 * n1 + 1, 0);
                int32_t *lo0 = h.data(), *off0 = lo0 + n0, *lo1 = off0 + n0 + 1, *off1 = lo1 + n1;
                const int64_t flo = std::max<int64_t>(a.rec_flo, INT32_MIN);
                const int64_t fhi = std::min<int64_t>(a.rec_fhi, INT32_MAX);
                auto run = [&](const std::vector<int32_t> &pos, int64_t b, int64_t e, int32_t *lo,
                               int32_t *off, int r) {
                    const auto first = pos.begin() + b, last = pos.begin() + e;
                    const auto x0 = std::lower_bound(first, last, (int32_t)flo);
                    const auto x1 = std::upper_bound(x0, last, (int32_t)fhi);
                    lo[r] = (int32_t)(x0 - pos.begin());
                    off[r + 1] = off[r] + (int32_t)(x1 - x0);
                };
                for (int i = 0; i < niso; i++) {
                    run(l->h_giown, l->iso_gstart[(size_t)i], l->iso_gstart[(size_t)i + 1], lo1, off1, i);
                    if (staged)
                        for (int ph = 0; ph < osamp; ph++)
                            run(p->h_ph_iown, p->h_ph_start[(size_t)i * (osamp + 1) + ph],
                                p->h_ph_start[(size_t)i * (osamp + 1) + ph + 1], lo0, off0,
                                i * osamp + ph);
                    else
                        run(l->h_giown, l->iso_gstart[(size_t)i], l->iso_gstart[(size_t)i + 1], lo0, off0, i);
                }
                if (h.size() > p->wm_cap) {
                    (void)hipFree(p->d_wm);
                    p->d_wm = nullptr;
                    p->wm_cap = 0;
                    if (hipMalloc(&p->d_wm, h.size() * 4) != hipSuccess) {
                        pb::set_error("pb_lbl_extinction: cannot allocate the window map");
                        return PB_ERR_NOMEM;
                    }
                    p->wm_cap = h.size();
                }
                PB_HIP(hipMemcpyAsync(p->d_wm, h.data(), h.size() * 4, hipMemcpyHostToDevice, s));
                p->wm_flo = a.rec_flo;
                p->wm_fhi = a.rec_fhi;
                p->wm_staged = (int)staged;
                p->wm_n0 = n0;
                p->wm_n1 = n1;
                p->wm_total0 = off0[n0];
                p->wm_total1 = off1[n1];
            }
            a.wm_n[0] = p->wm_n0;
            a.wm_n[1] = p->wm_n1;
            a.wm_lo[0] = p->d_wm;
            a.wm_off[0] = p->d_wm + p->wm_n0;
            a.wm_lo[1] = p->d_wm + 2 * p->wm_n0 + 1;
            a.wm_off[1] = a.wm_lo[1] + p->wm_n1;
            a.wm_total[0] = p->wm_total0;
            a.wm_total[1] = p->wm_total1;
            rec_threads = std::max<int64_t>(1, std::max(p->wm_total0, p->wm_total1));
        }
        dim3 grid(pb::div_up(rec_threads, kBlock), pb::div_up(nlayers, per));
        // the run offsets of the phase-order window map go to LDS while they fit beside the rest
        // in 48 KiB (niso * osamp + 1 words: the reference's default wnosamp of 2160 with 8
        // isotopes is already 69 KiB); larger maps are bisected in global memory
        const size_t rlds0 = (size_t)per * a.nrows * 8 + (size_t)a.ndop * 8 +
                             (size_t)per * a.niso * (8 + 8 + 8 + 4) + (size_t)a.niso * (8 + 4) + 16;
        const size_t wm_bytes = ((size_t)a.wm_n[0] + 2) * 4;
        size_t wm_cap_lds = 48 * 1024;
        if (const char *e = getenv("PB_WM_LDS_CAP"))
            wm_cap_lds = (size_t)atol(e);
        a.wm_lds = a.wm_off[0] && rlds0 + wm_bytes <= wm_cap_lds ? 1 : 0;
        const size_t rlds = rlds0 + (a.wm_lds ? wm_bytes : 8);
        PB_REQUIRE(rlds <= 64 * 1024, "pb_lbl_extinction: %zu B of LDS for the record kernel "
                   "(too many isotopes / output rows)", rlds);
        const int fmt = a.rec32 ? 3 : (a.rec16 && a.nch_max > 1) ? 2 : a.rec16 ? 1 : 0;
        void (*krec)(LblArgs) =
            per == 1 ? (fmt == 3   ? k_records<3, 1>
                        : fmt == 2 ? k_records<2, 1>
                        : fmt == 1 ? k_records<1, 1>
                                   : k_records<0, 1>)
                     : (fmt == 3   ? k_records<3, kRecLayers>
                        : fmt == 2 ? k_records<2, kRecLayers>
                        : fmt == 1 ? k_records<1, kRecLayers>
                                   : k_records<0, kRecLayers>);
        krec<<<grid, kBlock, rlds, s>>>(a);
        PB_LAUNCH_CHECK();
    } else if (l->nlines > 0) {
        const int lines_per_block = 4096;
        dim3 grid(pb::div_up(l->nlines, lines_per_block), nlayers);
        k_kmax<<<grid, kBlock, (size_t)a.nrows * 8, s>>>(a, lines_per_block);
        PB_LAUNCH_CHECK();
    }
    if (phase == 1)
        return PB_OK;
    const int layer_groups = (nlayers + 7) / 8;
    const bool timed = !dyn_fall && p->ev_used + 2 <= (int)p->ev.size();
    if (timed)
        PB_HIP(hipEventRecord(p->ev[p->ev_used], s));
    p->last_gather = dyn_fall ? 6 : scatter ? 4
                             : (p->resolution ? 3 : rounds ? 5 : staged ? 2 : 1) +
                                   (resident ? 8 : 0) + (a.wave_cap > 0 ? 16 : 0);
#ifdef PB_EXPERIMENTS
    if (scatter) {
        int T = 512;
        if (const char *e = getenv("PB_SCATTER_T"))
            T = atoi(e);
        T = T >= 2048 ? 2048 : T >= 1024 ? 1024 : 512;
        a.ntiles = pb::div_up(wcount, T);
        dim3 grid((unsigned)(8 * a.ntiles * layer_groups), a.nrows);
        if (T == 2048)
            k_ext_scatter<2048><<<grid, 64, 0, s>>>(a);
        else if (T == 1024)
            k_ext_scatter<1024><<<grid, 64, 0, s>>>(a);
        else
            k_ext_scatter<512><<<grid, 64, 0, s>>>(a);
        PB_LAUNCH_CHECK();
    } else
#endif  // PB_EXPERIMENTS
    if (resident) {
        a.ntiles = pb::div_up(wcount, kResTile);
        dim3 grid((unsigned)(8 * a.ntiles * layer_groups), a.nrows);
        const size_t rlds = ((size_t)a.res_cap + 2) * 8 + (size_t)kResThreads * (16 + 4) +
                            kResWaves * 4 + 64;
        if (rlds > 64 * 1024)
            PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ext_resident),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
        k_ext_resident<<<grid, kResThreads, rlds, s>>>(a);
        PB_LAUNCH_CHECK();
    }
    if (scatter) {
        // the scatter kernel computed every layer
    } else if (p->resolution) {
        a.ntiles = pb::div_up(wcount, kBlock);
        dim3 grid((unsigned)(8 * a.ntiles * layer_groups), a.nrows);
        k_ext_linterp<<<grid, kBlock, 0, s>>>(a);
    }
#ifdef PB_EXPERIMENTS
    else if (rounds) {
        // geometry: 16 wavefronts x 2 spans of 256 samples (tile 8192, two LDS buffers of 8192
        // samples, one workgroup per CU) or 8 x 2 (tile 4096, buffers of 4096, two per CU)
        int geom = 2;
        if (const char *e = getenv("PB_ROUNDS_GEOM"))
            geom = std::max(0, std::min(7, atoi(e)));
        int T = 0;
        rounds_geometry(geom, &T, &a.rbuf);
        a.rtile = T;
        a.ntiles = pb::div_up(wcount, T);
        int rsplit = (int)std::min<int64_t>(
            8, pb::div_up((int64_t)(per_phase >= 64.0 ? 4000 : 1000),
                          std::max<int64_t>(1, (int64_t)a.ntiles * nlayers)));
        if (const char *e = getenv("PB_STAGE_SPLIT"))
            rsplit = std::max(1, std::min(8, atoi(e)));
        {
            const int64_t plane = (int64_t)nlayers * a.nrows * wcount * 8;
            while (rsplit > 1 && (rsplit - 1) * plane > ((int64_t)1 << 30))
                rsplit--;
        }
        a.nsplit = rsplit;
        if (rsplit > 1) {
            const size_t need = (size_t)(rsplit - 1) * nlayers * a.nrows * wcount * 8;
            if (need > p->part_bytes) {
                (void)hipFree(p->part);
                p->part = nullptr;
                p->part_bytes = 0;
                if (hipMalloc(&p->part, need) != hipSuccess) {
                    pb::set_error("pb_lbl_extinction: cannot allocate %zu B of partial sums", need);
                    return PB_ERR_NOMEM;
                }
                p->part_bytes = need;
            }
            a.part = p->part;
        }
        // the largest distance from which a group can reach a tile, over all layers
        int64_t hmax_all = 0;
        for (int32_t h : v->psize)
            hmax_all = std::max<int64_t>(hmax_all, h);
        int64_t reachmax = hmax_all;
        if (a.cutoff > 0.0)
            reachmax = std::min(reachmax, (int64_t)(a.cutoff / a.ownstep) + 2 * (int64_t)v->osamp + 2);
        reachmax += 2 * (int64_t)v->osamp;
        a.reachmax = reachmax;
        const int nunits = a.ntiles * rsplit;
        if (p->cap_key[0] != wbegin || p->cap_key[1] != wcount || p->cap_key[2] != T ||
            p->cap_key[3] != rsplit) {
            // capacity of every unit: the groups (any isotope) within reach of its tile
            std::vector<int64_t> cap((size_t)nunits + 1, 0);
            for (int t = 0; t < a.ntiles; t++) {
                const int64_t t0 = wbegin + (int64_t)t * T;
                const int64_t tend = std::min(t0 + T, wbegin + wcount);
                const int64_t flo = t0 * v->osamp - reachmax, fhi = (tend - 1) * v->osamp + reachmax;
                int64_t n = 0;
                for (int i = 0; i < p->niso; i++) {
                    const int32_t *b = l->h_giown.data() + l->iso_gstart[i];
                    const int32_t *e = l->h_giown.data() + l->iso_gstart[i + 1];
                    n += std::upper_bound(b, e, (int32_t)std::min<int64_t>(fhi, INT_MAX)) -
                         std::lower_bound(b, e, (int32_t)std::max<int64_t>(flo, INT_MIN));
                }
                n = (n + 3) & ~(int64_t)3;
                for (int z = 0; z < rsplit; z++)
                    cap[(size_t)t * rsplit + z + 1] = n;
            }
            for (int u = 0; u < nunits; u++)
                cap[(size_t)u + 1] += cap[(size_t)u];
            PB_HIP(hipStreamSynchronize(s));       // an earlier call may still read the lists
            (void)hipFree(p->unit_cap);
            p->unit_cap = nullptr;
            PB_HIP(hipMalloc(&p->unit_cap, cap.size() * 8));
            PB_HIP(hipMemcpy(p->unit_cap, cap.data(), cap.size() * 8, hipMemcpyHostToDevice));
            p->cap_key[0] = wbegin;
            p->cap_key[1] = wcount;
            p->cap_key[2] = T;
            p->cap_key[3] = rsplit;
            p->cap_key[4] = cap.back();
        }
        const size_t nent = (size_t)nlayers * a.nrows * (size_t)p->cap_key[4] + 4;
        if (nent > p->vrec_alloc) {
            PB_HIP(hipStreamSynchronize(s));
            (void)hipFree(p->vrec);
            (void)hipFree(p->vseg);
            (void)hipFree(p->vrnd);
            p->vrec = nullptr;
            p->vseg = nullptr;
            p->vrnd = nullptr;
            p->vrec_alloc = 0;
            if (hipMalloc(&p->vrec, nent * 16) != hipSuccess ||
                hipMalloc(&p->vseg, nent * 16) != hipSuccess ||
                hipMalloc(&p->vrnd, nent * 4) != hipSuccess) {
                pb::set_error("pb_lbl_extinction: cannot allocate %zu B of visit records", nent * 32);
                return PB_ERR_NOMEM;
            }
            p->vrec_alloc = nent;
        }
        const size_t nhdr = (size_t)nlayers * a.nrows * nunits;
        if (nhdr > p->uhdr_alloc) {
            PB_HIP(hipStreamSynchronize(s));
            (void)hipFree(p->uhdr);
            p->uhdr = nullptr;
            PB_HIP(hipMalloc(&p->uhdr, nhdr * 16));
            p->uhdr_alloc = nhdr;
        }
        a.unit_cap = p->unit_cap;
        a.vrec = p->vrec;
        a.vseg = p->vseg;
        a.vrnd = p->vrnd;
        a.uhdr = p->uhdr;
        int rc = rounds_launch(a, geom, s);
        if (rc != PB_OK)
            return rc;
        if (rsplit > 1) {
            const int64_t n = (int64_t)nlayers * a.nrows * wcount;
            k_combine_parts<<<(unsigned)pb::div_up(n, kBlock), kBlock, 0, s>>>(
                ext_d, p->part, rsplit - 1, n, nullptr, (int64_t)a.nrows * wcount);
        }
    }
#endif  // PB_EXPERIMENTS
    else if (staged) {
        // Per-layer split.  A launch ends when its slowest workgroup does, and the slowest are the
        // tiles of the deepest layers (cutoff-limited windows of ~1000 samples against 200-300
        // higher up: 0.7-0.9 ms of a 0.88-ms C2 launch, profiles/r02_gather_ab.md), which the
        // dispatch order puts first.  The deepest `deep` layers are cut into more pieces than the
        // others, so that no single workgroup spans the launch.  Pieces of a tile add their sums in
        // a fixed order (k_combine_layer_parts): bitwise reproducible; against an unsplit launch
        // the association of a sample's terms differs (~1e-16).  PB_STAGE_SPLIT pins one split for
        // every layer (the exactness tests), PB_STAGE_DEEP=frac[,factor] tunes the rule.
        // Measured (profiles/r03_gather_ab.md): C3 44.1 -> 42.3 ms (-4 %), the 1e6-line list -1 %,
        // C2 +4 % (one spectrum at a time) / +3 % (pipelined): the second prologue and the combine
        // pass cost a light launch more than its tail does.  On by default only for long rows.
        int deep = 0, deep_split = nsplit;
        if (!getenv("PB_STAGE_SPLIT") && nsplit < 8 && nlayers >= 8) {
            double frac = a.nch_max > 1 ? 0.3 : 0.0;
            int factor = 2;
            if (const char *e = getenv("PB_STAGE_DEEP")) {
                frac = atof(e);
                if (const char *c = strchr(e, ','))
                    factor = std::max(1, atoi(c + 1));
            }
            deep = (int)(frac * nlayers + 0.5);
            deep_split = std::min(8, nsplit * factor);
            const int64_t plane = (int64_t)nlayers * a.nrows * wcount * 8;
            while (deep_split > nsplit && (deep_split - 1) * plane > ((int64_t)1 << 30))
                deep_split--;
            if (deep <= 0 || deep_split <= nsplit)
                deep = 0, deep_split = nsplit;
        }
        a.unit_tab = nullptr;
        a.lsplit = nullptr;
        a.nunits = 0;
        a.tsplit = nullptr;
        // Per-tile split.  A uniform line list gives every tile the same number of records; a
        // real one has band heads (10^2-10^3 x the line density of the gaps): the few tiles under
        // a head run 10 x as long as the others and end the launch alone (the C2 grid with 8 band
        // heads per isotope at 300 x contrast: gather 1.90 ms unsplit, 1.34 ms with every tile in
        // four pieces -- profiles/r04_bands.md).  From the groups within reach of every tile (host
        // copy of the group positions, cached per tiling) the tiles above 1.5 x the median count
        // get more pieces, up to 8; the others keep the launch's.  Automatic mode only: a forced
        // mode adds the terms of a sample in one order whatever the tiling (pbhip.h).
        if (deep == 0 && a.nch_max == 1 && p->gather_mode == 0 && a.wave_cap == 0 &&
            !getenv("PB_STAGE_SPLIT") && !(getenv("PB_TILE_SPLIT") && atoi(getenv("PB_TILE_SPLIT")) == 0) &&
            !l->h_giown.empty()) {
            const int tile = S * (int)sub;
            const int nt = pb::div_up(wcount, tile);
            if (p->ts_key[0] != wbegin || p->ts_key[1] != wcount || p->ts_key[2] != tile ||
                p->ts_key[3] != nsplit) {
                int64_t hmax_all = 0;
                for (int32_t h : v->psize)
                    hmax_all = std::max<int64_t>(hmax_all, h);
                int64_t reach = hmax_all;
                if (a.cutoff > 0.0)
                    reach = std::min(reach, (int64_t)(a.cutoff / a.ownstep) + 2 * (int64_t)v->osamp + 2);
                std::vector<int64_t> cnt((size_t)nt, 0);
                for (int t = 0; t < nt; t++) {
                    const int64_t t0 = wbegin + (int64_t)t * tile;
                    const int64_t tend = std::min<int64_t>(t0 + tile, wbegin + wcount);
                    const int64_t flo = t0 * v->osamp - reach, fhi = (tend - 1) * v->osamp + reach;
                    for (int i = 0; i < p->niso; i++) {
                        const int32_t *b = l->h_giown.data() + l->iso_gstart[(size_t)i];
                        const int32_t *e = l->h_giown.data() + l->iso_gstart[(size_t)i + 1];
                        cnt[(size_t)t] += std::upper_bound(b, e, (int32_t)std::min<int64_t>(fhi, INT_MAX)) -
                                          std::lower_bound(b, e, (int32_t)std::max<int64_t>(flo, INT_MIN));
                    }
                }
                std::vector<int64_t> sorted(cnt);
                std::nth_element(sorted.begin(), sorted.begin() + nt / 2, sorted.end());
                const double median = (double)std::max<int64_t>(1, sorted[(size_t)nt / 2]);
                std::vector<int32_t> ts((size_t)nt);
                int tmax = nsplit;
                const int tmin = getenv("PB_TILE_MIN") ? atoi(getenv("PB_TILE_MIN")) : 1;
                if (getenv("PB_TILE_DEBUG")) {
                    fprintf(stderr, "tile counts (median %.0f):", median);
                    for (int t = 0; t < nt; t++)
                        fprintf(stderr, " %lld", (long long)cnt[(size_t)t]);
                    fprintf(stderr, "\n");
                }
                // groups per (2048 samples, phase row) of a tile against the threshold that sends a
                // whole launch to the global gather: a tile below it is visit-starved in the staged
                // kernel (~300 barrier steps of one or two records: as long as a full tile) -- once
                // SOME tile is dense enough to be split, the sparse ones go to the global gather
                bool sparse = false, dense = false;
                for (int t = 0; t < nt; t++) {
                    const int k = std::max(tmin, (int)std::ceil((double)cnt[(size_t)t] / (1.5 * median)));
                    ts[(size_t)t] = std::max(nsplit, std::min(8, nsplit * std::max(1, k)));
                    tmax = std::max(tmax, ts[(size_t)t]);
                    dense = dense || ts[(size_t)t] > nsplit;
                }
                if (dense && packable && p->pos2ph &&
                    !(getenv("PB_TILE_GLOBAL") && atoi(getenv("PB_TILE_GLOBAL")) == 0))
                    for (int t = 0; t < nt; t++) {
                        const double per =
                            (double)cnt[(size_t)t] * 2048.0 / (double)tile / (double)v->osamp;
                        if (per < p->stage_threshold) {
                            ts[(size_t)t] = 0;
                            sparse = true;
                        }
                    }
                p->ts_sparse = sparse;
                PB_HIP(hipStreamSynchronize(s));       // an earlier call may still read the table
                if (nt > p->ts_tiles) {
                    (void)hipFree(p->d_tsplit);
                    p->d_tsplit = nullptr;
                    PB_HIP(hipMalloc(&p->d_tsplit, (size_t)nt * 4));
                    p->ts_tiles = nt;
                }
                PB_HIP(hipMemcpy(p->d_tsplit, ts.data(), (size_t)nt * 4, hipMemcpyHostToDevice));
                p->ts_key[0] = wbegin;
                p->ts_key[1] = wcount;
                p->ts_key[2] = tile;
                p->ts_key[3] = nsplit;
                p->ts_max = tmax;
            }
            if (p->ts_max > nsplit) {
                const int64_t plane = (int64_t)nlayers * a.nrows * wcount * 8;
                if ((int64_t)(p->ts_max - 1) * plane <= ((int64_t)1 << 30)) {
                    a.tsplit = p->d_tsplit;
                    a.ts_tile = tile;
                    a.pos2ph = p->pos2ph;
                    nsplit = deep_split = p->ts_max;
                }
            }
        }
        if (deep > 0) {
            if (p->ut_key[0] != nlayers || p->ut_key[1] != nsplit || p->ut_key[2] != deep ||
                p->ut_key[3] != deep_split) {
                std::vector<int32_t> tab, ls((size_t)nlayers);
                for (int layer = nlayers - 1; layer >= 0; layer--) {
                    const int n = layer >= nlayers - deep ? deep_split : nsplit;
                    ls[(size_t)layer] = n;
                    for (int z = 0; z < n; z++)
                        tab.push_back((layer << 8) | z);
                }
                PB_HIP(hipStreamSynchronize(s));       // an earlier call may still read the tables
                (void)hipFree(p->d_unit_tab);
                (void)hipFree(p->d_lsplit);
                p->d_unit_tab = p->d_lsplit = nullptr;
                PB_HIP(hipMalloc(&p->d_unit_tab, tab.size() * 4));
                PB_HIP(hipMalloc(&p->d_lsplit, ls.size() * 4));
                PB_HIP(hipMemcpy(p->d_unit_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
                PB_HIP(hipMemcpy(p->d_lsplit, ls.data(), ls.size() * 4, hipMemcpyHostToDevice));
                p->ut_key[0] = nlayers;
                p->ut_key[1] = nsplit;
                p->ut_key[2] = deep;
                p->ut_key[3] = deep_split;
                p->ut_units = (int)tab.size();
            }
            a.unit_tab = p->d_unit_tab;
            a.lsplit = p->d_lsplit;
            a.nunits = p->ut_units;
        }
        const int nunits = deep > 0 ? p->ut_units : nlayers * nsplit;
        const int nplanes = deep > 0 ? deep_split : nsplit;
        a.nsplit = nsplit;
        if (nplanes > 1) {
            const size_t need = (size_t)(nplanes - 1) * nlayers * a.nrows * wcount * 8;
            if (need > p->part_bytes) {
                (void)hipFree(p->part);
                p->part = nullptr;
                p->part_bytes = 0;
                if (hipMalloc(&p->part, need) != hipSuccess) {
                    pb::set_error("pb_lbl_extinction: cannot allocate %zu B of partial sums", need);
                    return PB_ERR_NOMEM;
                }
                p->part_bytes = need;
            }
            a.part = p->part;
        }
        a.ntiles = pb::div_up(wcount, S * sub);
        const int unit_groups = (nunits + 7) / 8;               // (layer, split) units per XCD
        dim3 grid((unsigned)(8 * a.ntiles * unit_groups), a.nrows);
        void (*kern)(LblArgs) =
            dma ? (S == 4   ? k_ext_staged<kStagedWaves, 4, true>
                   : S == 2 ? k_ext_staged<kStagedWaves, 2, true>
                            : k_ext_staged<kStagedWaves, 1, true>)
                : (S == 4   ? k_ext_staged<kStagedWaves, 4, false>
                   : S == 2 ? k_ext_staged<kStagedWaves, 2, false>
                            : k_ext_staged<kStagedWaves, 1, false>);
#ifdef PB_EXPERIMENTS
        unsigned long long *probe_d = nullptr;
        if (const char *e = getenv("PB_STAGE_PROBE")) {
            if (atoi(e) >= 1 && atoi(e) <= 3 && S == 2 && dma)    // 1, 2: timing probes, wrong sums
                kern = atoi(e) == 1   ? k_ext_staged<kStagedWaves, 2, true, 1>
                       : atoi(e) == 2 ? k_ext_staged<kStagedWaves, 2, true, 2>
                                      : k_ext_staged<kStagedWaves, 2, true, 3>;
            if (atoi(e) >= 11 && atoi(e) <= 12 && S == 2 && dma)  // 11, 12: who requests the row DMA
                kern = atoi(e) == 11 ? k_ext_staged<kStagedWaves, 2, true, 0, 1>
                                     : k_ext_staged<kStagedWaves, 2, true, 0, 2>;
            if (atoi(e) == 4 && S == 2 && dma) {                  // the cycle account (valid sums)
                kern = k_ext_staged<kStagedWaves, 2, true, 4>;
                PB_HIP(hipMalloc(&probe_d, (size_t)nlayers * 24 * 8));
                PB_HIP(hipMemsetAsync(probe_d, 0, (size_t)nlayers * 24 * 8, s));
                a.probe = probe_d;
            }
        }
#endif
        if (lds > 64 * 1024)
            PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#ifdef PB_EXPERIMENTS
        if (a.wave_cap > 0) {
            // the short-row layers first: many short workgroups, then the staged kernel's long ones
            const int rc = wave_launch(a, nunits, s);
            if (rc != PB_OK)
                return rc;
        }
#endif
        kern<<<grid, kStagedThreads, lds, s>>>(a);
#ifdef PB_EXPERIMENTS
        if (probe_d) {
            // one line per layer on stderr: layer, rowmax of its first isotope, then the 15 sums
            std::vector<unsigned long long> h((size_t)nlayers * 24);
            PB_HIP(hipMemcpyAsync(h.data(), probe_d, h.size() * 8, hipMemcpyDeviceToHost, s));
            std::vector<int32_t> rowmax((size_t)nlayers * a.niso);
            PB_HIP(hipMemcpyAsync(rowmax.data(), a.li_rowmax, rowmax.size() * 4,
                                  hipMemcpyDeviceToHost, s));
            PB_HIP(hipStreamSynchronize(s));
            (void)hipFree(probe_d);
            a.probe = nullptr;
            for (int layer = 0; layer < nlayers; layer++) {
                fprintf(stderr, "STAGE_PROBE layer %d rowmax %d :", layer,
                        rowmax[(size_t)layer * a.niso]);
                for (int i = 0; i < 21; i++)
                    fprintf(stderr, " %llu", h[(size_t)layer * 24 + i]);
                fprintf(stderr, "\n");
            }
        }
#endif
        if (deep > 0) {
            PB_LAUNCH_CHECK();
            const int64_t per_layer = (int64_t)a.nrows * wcount;
            dim3 cgrid((unsigned)pb::div_up(per_layer, kBlock), nlayers);
            k_combine_layer_parts<<<cgrid, kBlock, 0, s>>>(ext_d, p->part, p->d_lsplit, per_layer,
                                                          (int64_t)nlayers * per_layer,
                                                          a.res_cap > 0 ? a.ls_resident : nullptr);
        } else if (a.tsplit) {
            PB_LAUNCH_CHECK();
            if (p->ts_sparse) {
                // the sparse tiles through the global gather (RS = 1: its own 1024-sample tiles;
                // the tiles of the staged kernel end at once there, and the other way round)
                LblArgs g = a;
                g.ntiles = pb::div_up(wcount, kTile);
                dim3 ggrid((unsigned)(8 * g.ntiles * layer_groups), a.nrows);
                k_ext_resample<1, 4><<<ggrid, kBlock, 0, s>>>(g);
                PB_LAUNCH_CHECK();
            }
            const int64_t n = (int64_t)nlayers * a.nrows * wcount;
            k_combine_tile_parts<<<(unsigned)pb::div_up(n, kBlock), kBlock, 0, s>>>(
                ext_d, p->part, a.tsplit, S * (int)sub, wcount, n,
                a.res_cap > 0 ? a.ls_resident : nullptr, (int64_t)a.nrows * wcount);
        } else if (nsplit > 1) {
            PB_LAUNCH_CHECK();
            const int64_t n = (int64_t)nlayers * a.nrows * wcount;
            k_combine_parts<<<(unsigned)pb::div_up(n, kBlock), kBlock, 0, s>>>(
                ext_d, p->part, nsplit - 1, n, a.res_cap > 0 ? a.ls_resident : nullptr,
                (int64_t)a.nrows * wcount);
        }
    } else {
        // record splitting when the launch would not fill the chip
        // (measured at C2: RS=2 beats RS=1 until the launch has ~16k workgroups; a 16-wave
        // workgroup with a 16-way split, selectable with PB_RSPLIT=16, measured slower than
        // RS=4: every wavefront still scans every 64-record round for its share)
        const int64_t tiles1 = pb::div_up(wcount, kTile) * (int64_t)nlayers * a.nrows;
        int RS = tiles1 >= 16000 ? 1 : tiles1 >= 1500 ? 2 : 4;
        if (const char *e = getenv("PB_RSPLIT")) {
            const int v_ = atoi(e);
            RS = v_ >= 16 ? 16 : v_ >= 4 ? 4 : v_ >= 2 ? 2 : 1;
        }
        const int tile = RS == 16 ? kWaveSpan : kTile / RS;
        a.ntiles = pb::div_up(wcount, tile);
        dim3 grid((unsigned)(8 * a.ntiles * layer_groups), a.nrows);
        if (RS == 16)
            k_ext_resample<16, 16><<<grid, 1024, 0, s>>>(a);
        else if (RS == 4)
            k_ext_resample<4, 4><<<grid, kBlock, 0, s>>>(a);
        else if (RS == 2)
            k_ext_resample<2, 4><<<grid, kBlock, 0, s>>>(a);
        else
            k_ext_resample<1, 4><<<grid, kBlock, 0, s>>>(a);
    }
    PB_LAUNCH_CHECK();
    if (res_look) {
        std::vector<int32_t> h((size_t)nlayers);
        PB_HIP(hipMemcpyAsync(h.data(), p->ls_resident, (size_t)nlayers * 4, hipMemcpyDeviceToHost, s));
        PB_HIP(hipStreamSynchronize(s));
        p->res_seen = std::any_of(h.begin(), h.end(), [](int32_t v) { return v != 0; }) ? 1 : 0;
    }
    p->last_args = a;
    // (chunked records are counted only in the one-row form with every isotope kept: the entries
    // of a skipped isotope are never written)
    p->last_packed = a.rec16 != nullptr &&
                     (a.nch_max == 1 ||
                      (a.nrows == 1 && std::all_of(p->isoiext.begin(), p->isoiext.end(),
                                                   [](int32_t v) { return v >= 0; })));
    if (timed) {
        PB_HIP(hipEventRecord(p->ev[p->ev_used + 1], s));
        p->ev_used += 2;
    }
    return PB_OK;
}

// `resolution` plans, gather mode 6.  The reference accumulates every line of a layer on the
// layer's dynamic grid -- constant step, ofactor fine samples (_extcoeff.c:185-195, 281-307) -- and
// interpolates the outputs from it (:320-326).  That grid is a constant-step output grid with
// oversampling factor ofactor and nothing to resample, so a constant-step plan per factor computes
// it with the staged kernels (phase rows modulo the factor, shared in LDS by every line of a
// phase), and k_dyn_interp finishes.  Layers are walked in runs of equal factor.  The factors are
// read back from the device (one stream synchronisation per call).
static int dyn_subplan(pb_lbl *p, int f, hipStream_t s, pb_lbl::DynSub **out)
{
    for (pb_lbl::DynSub &d : p->dyn)
        if (d.f == f) {
            *out = &d;
            return PB_OK;
        }
    const pb_lines *l = p->lines;
    pb_lbl::DynSub d{f, nullptr, nullptr, nullptr, 0, 0, 0};
    int rc = pb_voigt_rephase(&d.voigt, p->voigt, f, s);
    if (rc)
        return rc;
    std::vector<int32_t> divs;
    for (int32_t x : p->h_divisors)
        if (x <= f && f % x == 0)
            divs.push_back(x);
    const int64_t dn = 1 + (l->onwn - 1) / f;
    std::vector<double> wn((size_t)dn);
    for (int64_t i = 0; i < dn; i++)
        wn[(size_t)i] = l->own0 + (double)(i * f) * l->ownstep;
    rc = pb_lbl_create(&d.plan, d.voigt, p->lines, wn.data(), (int)dn, divs.data(),
                       (int)divs.size(), p->h_molrad.data(), p->h_molmass.data(), p->nmol,
                       p->h_isoimol.data(), p->h_isomass.data(), p->h_isoratio.data(),
                       p->h_isoiext0.data(), p->niso, p->cutoff, p->ethresh, 0, p->max_layers);
    if (rc)
        return rc;                                     // (the table stays with p->voigt)
    // runs of one or two deep layers are small launches: the automatic choice would send them
    // to the global gather (c2-res: 180-580 us per layer against 55-200 staged)
    if (!getenv("PB_GATHER"))
        d.plan->gather_mode = 2;
    p->dyn.push_back(d);
    *out = &p->dyn.back();
    return PB_OK;
}

static int lbl_resolution_dyn(pb_lbl *p, LblArgs &a, double *ext_d, int64_t wbegin, int64_t wcount,
                              const double *temp_d, const double *dens_d, const double *isoz_d,
                              int64_t z_iso_stride, int64_t z_layer_stride, int nlayers, int add,
                              hipStream_t s)
{
    const pb_lines *l = p->lines;
    PB_REQUIRE(l->onwn < (1LL << 30), "pb_lbl_extinction: fine grid of %lld samples exceeds 2^30",
               (long long)l->onwn);
    k_layer_state<<<nlayers, 64, ((size_t)a.nlor + a.ndop) * 8 + (size_t)a.ndivs * 4, s>>>(a);
    PB_LAUNCH_CHECK();
    const size_t nstate = (size_t)nlayers * (1 + p->niso);
    // a finished read-back of an earlier call: adopt it as the prediction; if it contradicts the
    // prediction that call was planned with, the atmosphere is moving -- synchronise for a while
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capturing) != hipSuccess)
        (void)hipGetLastError();
    if (capturing != hipStreamCaptureStatusNone) {
        // (no event query while a graph is being captured)
    } else if (p->rb_pending && hipEventQuery(p->rb_ev) == hipSuccess) {
        p->rb_pending = false;
        if (p->rb_layers == nlayers && p->pred_layers == nlayers) {
            const int32_t *f = p->rb_host, *il = p->rb_host + nlayers;
            const bool same_f = std::equal(f, f + nlayers, p->used_f.begin());
            if (!same_f || !std::equal(il, il + (size_t)nlayers * p->niso, p->pred_ilor.begin())) {
                p->pred_f.assign(f, f + nlayers);
                p->pred_ilor.assign(il, il + (size_t)nlayers * p->niso);
                p->pred_dirty = true;
            }
            if (!same_f) {
                p->dyn_mispredicted++;
                p->dyn_hold = 8;
            }
        }
    } else if (p->rb_pending) {
        (void)hipGetLastError();                          // (hipErrorNotReady is not an error)
    }
    // (a captured call cannot synchronise: it is planned from the prediction or not at all)
    // (the default library has no switch for it: pb_lbl_set_dyn_predict exists in the experiments
    // build only -- measured no faster)
    const bool spec = kExp && p->dyn_predict && p->pred_layers == nlayers &&
                      (capturing != hipStreamCaptureStatusNone || p->dyn_hold == 0);
    PB_REQUIRE(spec || capturing == hipStreamCaptureStatusNone,
               "pb_lbl_extinction: a `resolution` plan in gather mode 6 can be captured into a "
               "graph only with pb_lbl_set_dyn_predict(plan, 1) and after one spectrum of this "
               "many layers");
    p->h_ofactor.resize((size_t)nlayers);
    p->h_ilor.resize((size_t)nlayers * p->niso);
    if (!spec) {
        PB_HIP(hipMemcpyAsync(p->h_ofactor.data(), p->ls_ofactor, (size_t)nlayers * 4,
                              hipMemcpyDeviceToHost, s));
        PB_HIP(hipMemcpyAsync(p->h_ilor.data(), p->li_ilor, (size_t)nlayers * p->niso * 4,
                              hipMemcpyDeviceToHost, s));
        PB_HIP(hipStreamSynchronize(s));
        if (p->pred_layers != nlayers || p->pred_f != p->h_ofactor || p->pred_ilor != p->h_ilor) {
            p->pred_f = p->h_ofactor;
            p->pred_ilor = p->h_ilor;
            p->pred_layers = nlayers;
            p->pred_dirty = true;
        }
        if (p->dyn_hold > 0)
            p->dyn_hold--;
        p->dyn_sync_calls++;
    } else {
        p->h_ofactor = p->pred_f;
        p->h_ilor = p->pred_ilor;
        p->dyn_spec_calls++;
        if (!p->rb_pending && capturing == hipStreamCaptureStatusNone) {
            if (nstate > p->rb_cap) {
                if (p->rb_host)
                    (void)hipHostFree(p->rb_host);
                p->rb_host = nullptr;
                p->rb_cap = 0;
                PB_HIP(hipHostMalloc((void **)&p->rb_host, nstate * 4, hipHostMallocDefault));
                p->rb_cap = nstate;
            }
            if (!p->rb_ev)
                PB_HIP(hipEventCreateWithFlags(&p->rb_ev, hipEventDisableTiming));
            PB_HIP(hipMemcpyAsync(p->rb_host, p->ls_ofactor, (size_t)nlayers * 4,
                                  hipMemcpyDeviceToHost, s));
            PB_HIP(hipMemcpyAsync(p->rb_host + nlayers, p->li_ilor, (size_t)nlayers * p->niso * 4,
                                  hipMemcpyDeviceToHost, s));
            PB_HIP(hipEventRecord(p->rb_ev, s));
            p->rb_pending = true;
            p->rb_layers = nlayers;
        }
    }
    p->used_f = p->h_ofactor;                             // (what this call is planned with)
    p->dyn_fallback = spec;
    // every sub-plan and Lorentz row of the plan exists before the device check runs; then the
    // prediction (factor and row mask of the factor's table, per layer) goes to the device
    for (int l0 = 0; l0 < nlayers;) {
        const int f = p->h_ofactor[(size_t)l0];
        int l1 = l0 + 1;
        while (l1 < nlayers && p->h_ofactor[(size_t)l1] == f)
            l1++;
        pb_lbl::DynSub *sub = nullptr;
        int rc = dyn_subplan(p, f, s, &sub);
        if (rc)
            return rc;
        std::vector<int> rows(p->h_ilor.begin() + (size_t)l0 * p->niso,
                              p->h_ilor.begin() + (size_t)l1 * p->niso);
        std::sort(rows.begin(), rows.end());
        rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
        rc = pb_voigt_ensure_rows(sub->voigt, rows.data(), (int)rows.size(), s);
        if (rc)
            return rc;
        l0 = l1;
    }
    if (nlayers > p->pred_cap) {
        PB_HIP(hipStreamSynchronize(s));
        (void)hipFree(p->d_pred_f);
        (void)hipFree(p->d_ok);
        (void)hipFree((void *)p->d_pred_mask);
        p->d_pred_f = p->d_ok = nullptr;
        p->d_pred_mask = nullptr;
        p->pred_cap = 0;
        PB_HIP(hipMalloc(&p->d_pred_f, (size_t)nlayers * 4));
        PB_HIP(hipMalloc(&p->d_ok, (size_t)nlayers * 4));
        PB_HIP(hipMalloc((void **)&p->d_pred_mask, (size_t)nlayers * sizeof(void *)));
        p->pred_cap = nlayers;
        p->pred_dirty = true;
    }
    if (p->pred_dirty) {
        PB_REQUIRE(capturing == hipStreamCaptureStatusNone,
                   "pb_lbl_extinction: the run plan of a `resolution` call changed while a graph "
                   "was being captured");
        std::vector<const uint8_t *> masks((size_t)nlayers);
        for (int layer = 0; layer < nlayers; layer++) {
            pb_lbl::DynSub *sub = nullptr;
            const int rc = dyn_subplan(p, p->h_ofactor[(size_t)layer], s, &sub);
            if (rc)
                return rc;
            masks[(size_t)layer] = sub->voigt->d_rowmask;
        }
        PB_HIP(hipMemcpyAsync(p->d_pred_f, p->h_ofactor.data(), (size_t)nlayers * 4,
                              hipMemcpyHostToDevice, s));
        PB_HIP(hipMemcpyAsync((void *)p->d_pred_mask, masks.data(), (size_t)nlayers * sizeof(void *),
                              hipMemcpyHostToDevice, s));
        PB_HIP(hipStreamSynchronize(s));                  // (`masks` is a local; rare)
        p->pred_dirty = false;
    }
    k_dyn_check<<<pb::div_up(nlayers, 64), 64, 0, s>>>(p->d_ok, p->ls_ofactor, p->li_ilor,
                                                      p->d_pred_f, p->d_pred_mask, nlayers, p->niso);
    PB_LAUNCH_CHECK();
    int lanes = 4;
    if (const char *e = getenv("PB_RES_DYN_STREAMS"))
        lanes = std::max(1, std::min(8, atoi(e)));
    if (lanes > 1 && p->dyn_streams.empty()) {
        PB_HIP(hipEventCreateWithFlags(&p->dyn_fork, hipEventDisableTiming));
        for (int k = 0; k < 8; k++) {
            hipStream_t t;
            hipEvent_t e;
            PB_HIP(hipStreamCreateWithFlags(&t, hipStreamNonBlocking));
            p->dyn_streams.push_back(t);
            PB_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            p->dyn_join.push_back(e);
        }
    }
    const bool timed = p->ev_used + 2 <= (int)p->ev.size();
    if (timed)
        PB_HIP(hipEventRecord(p->ev[p->ev_used], s));
    if (lanes > 1) {
        PB_HIP(hipEventRecord(p->dyn_fork, s));        // ext as the caller left it (zeroed, or sums so far)
        for (int k = 0; k < lanes; k++)
            PB_HIP(hipStreamWaitEvent(p->dyn_streams[(size_t)k], p->dyn_fork, 0));
    }
    p->last_gather = 6;
    p->dyn_runs = 0;
    p->dyn_call++;
    const double w_lo = p->h_wn[(size_t)wbegin], w_hi = p->h_wn[(size_t)(wbegin + wcount - 1)];
    int rc = PB_OK;
    // Dynamic sampling keeps the samples per line and layer about constant, so a run costs about
    // its layers (c2-res: 43 us per layer in runs of 14-18, 55-200 us for a run of one) plus its
    // small launches.  Runs that fill the chip by themselves queue on side stream 0; the others
    // (deep layers: one or two layers on a short grid) are dealt to the remaining side streams,
    // least work first, and run in the shadow of the large ones.
    double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int nbig = 1;
    if (const char *e = getenv("PB_RES_DYN_BIG"))
        nbig = std::max(1, std::min(7, atoi(e)));
    for (int l0 = 0; l0 < nlayers && rc == PB_OK;) {
        const int f = p->h_ofactor[(size_t)l0];
        // (a run's dynamic-grid sums stay below 1 GiB: fine factors on long fine grids)
        const int64_t run_max = std::max<int64_t>(
            1, ((int64_t)1 << 27) / ((int64_t)a.nrows * (1 + (l->onwn - 1) / f)));
        int l1 = l0 + 1;
        while (l1 < nlayers && l1 - l0 < run_max && p->h_ofactor[(size_t)l1] == f)
            l1++;
        pb_lbl::DynSub *sub = nullptr;
        rc = dyn_subplan(p, f, s, &sub);
        if (rc)
            break;
        // a factor that comes back later in the same call (a temperature inversion) shares the
        // sub-plan's workspaces with its first run: same side stream, hence in order
        int lane = 0;
        const int64_t groups = (int64_t)(l1 - l0) * pb::div_up((int64_t)sub->plan->nwave, (int64_t)4096);
        const int lo = groups < 512 ? std::min(nbig, lanes - 1) : 0;
        const int hi = groups < 512 ? lanes : std::min(nbig, lanes);
        lane = lo;
        for (int k = lo + 1; k < hi; k++)
            if (load[k] < load[lane])
                lane = k;
        if (sub->call == p->dyn_call)
            lane = sub->lane;
        load[lane] += 60.0 + 45.0 * (l1 - l0);
        sub->call = p->dyn_call;
        sub->lane = lane;
        hipStream_t t = lanes > 1 ? p->dyn_streams[(size_t)lane] : s;
        pb_lbl *q = sub->plan;
        {
            // the Lorentz rows of the re-cut table that the layers of this run read
            std::vector<int> rows(p->h_ilor.begin() + (size_t)l0 * p->niso,
                                  p->h_ilor.begin() + (size_t)l1 * p->niso);
            std::sort(rows.begin(), rows.end());
            rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
            rc = pb_voigt_ensure_rows(sub->voigt, rows.data(), (int)rows.size(), t);
            if (rc)
                break;
        }
        q->ethresh = p->ethresh;
        q->concurrency = std::max(p->concurrency, 1);
        if (q->isoiext != p->isoiext) {
            rc = pb_lbl_set_isoiext(q, p->isoiext.data());
            if (rc)
                break;
        }
        // the dynamic samples the outputs of this call read (one of margin on either side)
        const double step = l->ownstep * f;
        const int64_t dn = q->nwave;
        int64_t d0 = (int64_t)((w_lo - p->wn0) / step) - 1;
        int64_t d1 = (int64_t)((w_hi - p->wn0) / step) + 3;
        d0 = std::max<int64_t>(0, std::min(d0, dn - 1));
        d1 = std::max(d0 + 1, std::min(d1, dn));
        const int nl = l1 - l0;
        const size_t need = (size_t)nl * a.nrows * (size_t)(d1 - d0) * 8;
        if (need > sub->ktmp_bytes) {
            if (hipStreamSynchronize(t) != hipSuccess) {
                // (not PB_HIP: the side streams below must be joined on every path)
                pb::set_error("pb_lbl_extinction: side stream failed: %s",
                              hipGetErrorString(hipGetLastError()));
                rc = PB_ERR_HIP;
                break;
            }
            (void)hipFree(sub->ktmp);
            sub->ktmp = nullptr;
            sub->ktmp_bytes = 0;
            if (hipMalloc(&sub->ktmp, need) != hipSuccess) {
                pb::set_error("pb_lbl_extinction: cannot allocate %zu B of dynamic-grid sums", need);
                rc = PB_ERR_NOMEM;
                break;
            }
            sub->ktmp_bytes = need;
        }
        rc = lbl_extinction(q, sub->ktmp, d0, d1 - d0, temp_d + l0, dens_d + (int64_t)l0 * p->nmol,
                            isoz_d + (int64_t)l0 * z_layer_stride, z_iso_stride, z_layer_stride,
                            nl, add, t, 0);
        if (rc)
            break;
        // (pb_lbl_last_state / pb_lbl_kmax_buffer of this plan report the run's maxima)
        k_dyn_kmax<<<pb::div_up(nl * a.nrows, 64), 64, 0, t>>>(
            p->kmax_bits + (size_t)l0 * a.nrows, q->kmax_bits, p->d_ok + l0, nl, a.nrows);
        if (hipGetLastError() != hipSuccess) {
            pb::set_error("pb_lbl_extinction: copy of the per-row maxima failed");
            rc = PB_ERR_HIP;
            break;
        }
        dim3 grid((unsigned)pb::div_up(wcount, (int64_t)kBlock), (unsigned)(nl * a.nrows));
        k_dyn_interp<<<grid, kBlock, 0, t>>>(ext_d + (int64_t)l0 * a.nrows * wcount, sub->ktmp,
                                            p->d_wn, p->wn0, p->ls_dwnstep + l0, d0, d1 - d0,
                                            wbegin, wcount, a.nrows, p->d_ok + l0);
        if (hipGetLastError() != hipSuccess) {
            pb::set_error("pb_lbl_extinction: k_dyn_interp launch failed");
            rc = PB_ERR_HIP;
            break;
        }
        p->dyn_runs++;
        l0 = l1;
    }
    // (joined on every path: the caller's stream must not run ahead of a side stream)
    // Best effort, lane by lane: a failing record / wait must not leave the other lanes unjoined
    // (their kernels still write ext_d and the sub-plans' sums); a lane that cannot be joined
    // through its event is waited for on the host.
    if (lanes > 1)
        for (int k = 0; k < lanes; k++) {
            hipStream_t t = p->dyn_streams[(size_t)k];
            if (hipEventRecord(p->dyn_join[(size_t)k], t) != hipSuccess ||
                hipStreamWaitEvent(s, p->dyn_join[(size_t)k], 0) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipStreamSynchronize(t);
                if (rc == PB_OK) {
                    pb::set_error("pb_lbl_extinction: joining side stream %d failed", k);
                    rc = PB_ERR_HIP;
                }
            }
        }
    if (rc)
        return rc;
    if (timed) {
        PB_HIP(hipEventRecord(p->ev[p->ev_used + 1], s));
        p->ev_used += 2;
    }
    return PB_OK;
}

int pb_lbl_extinction(pb_lbl *p, double *ext_d, int64_t wbegin, int64_t wcount,
                      const double *temp_d, const double *dens_d, const double *isoz_d,
                      int64_t z_iso_stride, int64_t z_layer_stride, int nlayers, int add,
                      void *stream)
{
    return lbl_extinction(p, ext_d, wbegin, wcount, temp_d, dens_d, isoz_d, z_iso_stride,
                          z_layer_stride, nlayers, add, stream, 0);
}

int pb_lbl_extinction_begin(pb_lbl *p, double *ext_d, int64_t wbegin, int64_t wcount,
                            const double *temp_d, const double *dens_d, const double *isoz_d,
                            int64_t z_iso_stride, int64_t z_layer_stride, int nlayers, int add,
                            void *stream)
{
    PB_REQUIRE(p, "pb_lbl_extinction_begin: null handle");
    p->pending = {ext_d, wbegin, wcount, temp_d, dens_d, isoz_d, z_iso_stride, z_layer_stride,
                  nlayers, add, true};
    return lbl_extinction(p, ext_d, wbegin, wcount, temp_d, dens_d, isoz_d, z_iso_stride,
                          z_layer_stride, nlayers, add, stream, 1);
}

int pb_lbl_extinction_end(pb_lbl *p, void *stream)
{
    PB_REQUIRE(p && p->pending.open, "pb_lbl_extinction_end: no call was begun");
    const auto c = p->pending;
    p->pending.open = false;
    return lbl_extinction(p, c.ext, c.wbegin, c.wcount, c.temp, c.dens, c.isoz, c.zs0, c.zs1,
                          c.nlayers, c.add, stream, 2);
}

int pb_lbl_kmax_buffer(pb_lbl *p, void **kmax_d, int64_t *count)
{
    PB_REQUIRE(p && kmax_d && count, "pb_lbl_kmax_buffer: null pointer");
    *kmax_d = p->kmax_bits;
    *count = (int64_t)p->max_layers * p->kmax_rows;
    return PB_OK;
}

int pb_lbl_timing_begin(pb_lbl *p, int max_launches)
{
    PB_REQUIRE(p && max_launches >= 0, "pb_lbl_timing_begin: bad argument");
    while ((int)p->ev.size() < 2 * max_launches) {
        hipEvent_t e;
        PB_HIP(hipEventCreate(&e));
        p->ev.push_back(e);
    }
    while ((int)p->ev.size() > 2 * max_launches) {
        (void)hipEventDestroy(p->ev.back());
        p->ev.pop_back();
    }
    p->ev_used = 0;
    return PB_OK;
}

int pb_lbl_timing_end(pb_lbl *p, double *total_ms, int *launches)
{
    PB_REQUIRE(p && total_ms && launches, "pb_lbl_timing_end: null pointer");
    double sum = 0.0;
    for (int i = 0; i + 1 < p->ev_used; i += 2) {
        PB_HIP(hipEventSynchronize(p->ev[i + 1]));
        float ms = 0.f;
        PB_HIP(hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = p->ev_used / 2;
    p->ev_used = 0;
    for (hipEvent_t e : p->ev)
        (void)hipEventDestroy(e);
    p->ev.clear();
    return PB_OK;
}

int pb_lbl_last_state(pb_lbl *p, int32_t *ofactor_h, double *kmax_h, int nlayers, int nrows,
                      void *stream)
{
    PB_REQUIRE(p, "pb_lbl_last_state: null handle");
    PB_REQUIRE(nlayers >= 1 && nlayers <= p->max_layers && nrows >= 1 && nrows <= p->kmax_rows,
               "pb_lbl_last_state: bad sizes");
    PB_HIP(hipStreamSynchronize(pb::as_stream(stream)));
    if (ofactor_h)
        PB_HIP(hipMemcpy(ofactor_h, p->ls_ofactor, (size_t)nlayers * 4, hipMemcpyDeviceToHost));
    if (kmax_h)
        PB_HIP(hipMemcpy(kmax_h, p->kmax_bits, (size_t)nlayers * nrows * 8,
                         hipMemcpyDeviceToHost));
    return PB_OK;
}

int pb_lbl_last_layer_kinds(pb_lbl *p, int32_t *resident_h, int32_t *block_h, int nlayers,
                            void *stream)
{
    PB_REQUIRE(p, "pb_lbl_last_layer_kinds: null handle");
    PB_REQUIRE(nlayers >= 1 && nlayers <= p->max_layers, "pb_lbl_last_layer_kinds: bad sizes");
    PB_HIP(hipStreamSynchronize(pb::as_stream(stream)));
    if (resident_h)
        PB_HIP(hipMemcpy(resident_h, p->ls_resident, (size_t)nlayers * 4, hipMemcpyDeviceToHost));
    if (block_h)
        PB_HIP(hipMemcpy(block_h, p->ls_block, (size_t)nlayers * 4, hipMemcpyDeviceToHost));
    return PB_OK;
}

#ifdef PB_EXPERIMENTS
int pb_lbl_last_wave_layers(pb_lbl *p, int32_t *wave_h, int nlayers, void *stream)
{
    PB_REQUIRE(p && wave_h, "pb_lbl_last_wave_layers: null pointer");
    PB_REQUIRE(nlayers >= 1 && nlayers <= p->max_layers, "pb_lbl_last_wave_layers: bad sizes");
    PB_HIP(hipStreamSynchronize(pb::as_stream(stream)));
    if (p->last_args.wave_cap > 0)
        PB_HIP(hipMemcpy(wave_h, p->ls_wave, (size_t)nlayers * 4, hipMemcpyDeviceToHost));
    else
        memset(wave_h, 0, (size_t)nlayers * 4);
    return PB_OK;
}

int pb_lbl_set_dyn_predict(pb_lbl *p, int on)
{
    PB_REQUIRE(p, "pb_lbl_set_dyn_predict: null handle");
    PB_REQUIRE(p->resolution, "pb_lbl_set_dyn_predict: the plan is not a `resolution` plan");
    p->dyn_predict = on ? 1 : 0;
    return PB_OK;
}

int pb_lbl_dyn_stats(pb_lbl *p, int64_t stats[3])
{
    PB_REQUIRE(p && stats, "pb_lbl_dyn_stats: null pointer");
    stats[0] = p->dyn_spec_calls;
    stats[1] = p->dyn_sync_calls;
    stats[2] = p->dyn_mispredicted;
    return PB_OK;
}
#endif  // PB_EXPERIMENTS

int pb_lbl_last_work(pb_lbl *p, int64_t work[3], void *stream)
{
    PB_REQUIRE(p && work, "pb_lbl_last_work: null pointer");
    work[0] = work[1] = work[2] = -1;
    if (!p->last_packed)
        return PB_OK;                 // not counted for this kernel / record format
    hipStream_t s = pb::as_stream(stream);
    unsigned long long *d = nullptr;
    PB_HIP(hipMalloc(&d, 3 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 3 * sizeof(unsigned long long), s);
    if (e == hipSuccess) {
        k_work_stats<<<1024, kBlock, 0, s>>>(p->last_args, d);
        e = hipGetLastError();
    }
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess)
        e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    (void)hipFree(d);
    PB_HIP(e);
    for (int i = 0; i < 3; i++)
        work[i] = (int64_t)h[i];
    return PB_OK;
}

int pb_lbl_last_table_samples(pb_lbl *p, int64_t *samples, void *stream)
{
    PB_REQUIRE(p && samples, "pb_lbl_last_table_samples: null pointer");
    *samples = -1;
    const LblArgs &a = p->last_args;
    if (!p->last_packed || a.nch_max != 1)
        return PB_OK;                 // not counted for this kernel / record format
    hipStream_t s = pb::as_stream(stream);
    const int64_t n = (int64_t)a.nlayers * a.niso * a.ndop * a.osamp;
    int32_t *maxlen = nullptr;
    unsigned long long *d = nullptr;
    PB_HIP(hipMalloc(&maxlen, (size_t)n * sizeof(int32_t)));
    hipError_t e = hipMalloc(&d, sizeof(unsigned long long));
    if (e == hipSuccess)
        e = hipMemsetAsync(maxlen, 0, (size_t)n * sizeof(int32_t), s);
    if (e == hipSuccess)
        e = hipMemsetAsync(d, 0, sizeof(unsigned long long), s);
    unsigned long long h = 0;
    if (e == hipSuccess) {
        k_table_rows<<<1024, kBlock, 0, s>>>(a, maxlen);
        k_sum_i32<<<256, kBlock, 0, s>>>(maxlen, n, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    (void)hipFree(maxlen);
    (void)hipFree(d);
    PB_HIP(e);
    *samples = (int64_t)h;
    return PB_OK;
}

void pb_lbl_destroy(pb_lbl *p)
{
    if (!p)
        return;
    (void)hipFree(p->d_wn);
    (void)hipFree(p->d_molrad);
    (void)hipFree(p->d_molmass);
    (void)hipFree(p->d_isomass);
    (void)hipFree(p->d_isoratio);
    (void)hipFree(p->d_divisors);
    (void)hipFree(p->d_isoimol);
    (void)hipFree(p->d_isoiext);
    (void)hipFree(p->ls_ofactor);
    (void)hipFree(p->ls_scale);
    (void)hipFree(p->ls_resident);
    (void)hipFree(p->ls_block);
    (void)hipFree(p->ls_wave);
    (void)hipFree(p->d_tsplit);
    (void)hipFree(p->pos2ph);
    (void)hipFree(p->rec32);
    (void)hipFree(p->rec16);
    (void)hipFree(p->part);
    (void)hipFree(p->d_wm);
    (void)hipFree(p->d_unit_tab);
    (void)hipFree(p->d_lsplit);
    (void)hipFree(p->gs_start);
    (void)hipFree(p->ls_dnwn);
    (void)hipFree(p->ls_dwnstep);
    (void)hipFree(p->ls_quot);
    (void)hipFree(p->li_alphad);
    (void)hipFree(p->li_dens);
    (void)hipFree(p->li_z);
    (void)hipFree(p->li_invz);
    (void)hipFree(p->li_ilor);
    (void)hipFree(p->li_hmax);
    (void)hipFree(p->li_rowmax);
    (void)hipFree(p->li_hlo);
    (void)hipFree(p->li_hhi);
    (void)hipFree(p->kmax_bits);
    (void)hipFree(p->unit_cap);
    (void)hipFree(p->vrec);
    (void)hipFree(p->vseg);
    (void)hipFree(p->vrnd);
    (void)hipFree(p->uhdr);
    (void)hipFree(p->ph_first);
    (void)hipFree(p->ph_count);
    (void)hipFree(p->ph_iown);
    (void)hipFree(p->ph_start);
    (void)hipFree(p->ph_iso);
    (void)hipFree(p->ph_bin);
    (void)hipFree(p->ph_lead);
    (void)hipFree(p->g_lead);
    (void)hipFree(p->rec_k);
    (void)hipFree(p->rec_i32);
    for (hipEvent_t e : p->ev)
        (void)hipEventDestroy(e);
    for (pb_lbl::DynSub &d : p->dyn) {
        pb_lbl_destroy(d.plan);                        // (their tables belong to p->voigt)
        (void)hipFree(d.ktmp);
    }
    for (hipStream_t t : p->dyn_streams)
        (void)hipStreamDestroy(t);
    for (hipEvent_t e : p->dyn_join)
        (void)hipEventDestroy(e);
    if (p->dyn_fork)
        (void)hipEventDestroy(p->dyn_fork);
    if (p->rb_ev)
        (void)hipEventDestroy(p->rb_ev);
    (void)hipFree(p->d_pred_f);
    (void)hipFree(p->d_ok);
    (void)hipFree((void *)p->d_pred_mask);
    if (p->rb_host)
        (void)hipHostFree(p->rb_host);
    delete p;
}

static int interp_ec_launch(bool assign, double *extinction_d, const double *etable_d,
                            const double *ttable_d, const double *temperatures_d,
                            const double *density_d, int nmol, int ntemp, int nlayers,
                            int nwave, int lay1, int lay2, int per_mol, void *stream)
{
    PB_REQUIRE(nmol >= 1 && ntemp >= 2 && nlayers >= 1 && nwave >= 0,
               "pb_interp_ec: bad shape (needs >= 2 table temperatures)");
    PB_REQUIRE(lay1 >= 0, "pb_interp_ec: lay1 < 0");
    if (lay2 > nlayers)
        lay2 = nlayers;
    if (lay2 <= lay1 || nwave == 0)
        return PB_OK;
    PB_REQUIRE(extinction_d && etable_d && ttable_d && temperatures_d && density_d,
               "pb_interp_ec: null pointer");
    dim3 grid(pb::div_up(nwave, kBlock), lay2 - lay1, per_mol ? nmol : 1);
    if (assign)
        k_interp_ec<true><<<grid, kBlock, 0, pb::as_stream(stream)>>>(
            extinction_d, etable_d, ttable_d, temperatures_d, density_d, nmol, ntemp, nlayers,
            nwave, lay1, per_mol ? 1 : 0);
    else
        k_interp_ec<false><<<grid, kBlock, 0, pb::as_stream(stream)>>>(
            extinction_d, etable_d, ttable_d, temperatures_d, density_d, nmol, ntemp, nlayers,
            nwave, lay1, per_mol ? 1 : 0);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

int pb_interp_ec(double *extinction_d, const double *etable_d, const double *ttable_d,
                 const double *temperatures_d, const double *density_d, int nmol, int ntemp,
                 int nlayers, int nwave, int lay1, int lay2, int per_mol, void *stream)
{
    return interp_ec_launch(false, extinction_d, etable_d, ttable_d, temperatures_d, density_d,
                            nmol, ntemp, nlayers, nwave, lay1, lay2, per_mol, stream);
}

int pb_interp_ec_set(double *extinction_d, const double *etable_d, const double *ttable_d,
                     const double *temperatures_d, const double *density_d, int nmol, int ntemp,
                     int nlayers, int nwave, int lay1, int lay2, int per_mol, void *stream)
{
    return interp_ec_launch(true, extinction_d, etable_d, ttable_d, temperatures_d, density_d,
                            nmol, ntemp, nlayers, nwave, lay1, lay2, per_mol, stream);
}

}  // extern "C"
