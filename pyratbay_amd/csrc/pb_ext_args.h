// Argument block and record formats shared by the translation units of the line-by-line
// extinction (pb_extinction.hip: layer state, records, the gather kernels and the C ABI;
// pb_rounds.hip: the round-staged gather).
#pragma once

#include <cstdint>

#include "pb_common.h"
#include "pb_internal.h"

namespace pbx {

constexpr int kBinSamples = 256;     // output samples per bin of the phase-list position index
constexpr int kStagePad = 256;       // zero samples on either side of a staged row
constexpr int kStageSpan = 256;      // samples per wavefront and sub-tile (4 chunks of 64)
constexpr int kStageRowMax = 1024;   // longest phase row the staged kernels keep in one piece

// wave-autonomous gather (pb_wave.hip): output samples per workgroup (every wavefront keeps all
// of them) and the longest phase row / window it stages (three LDS-DMA pieces of 128 samples)
constexpr int kWvTile = 2048;
constexpr int kWvRowMax = 384;

struct VRec;
struct VSeg;
struct UnitHdr;

// Record of the scatter kernel: everything one wavefront needs about a (layer, group) pair,
// read with ONE scalar load.
struct __attribute__((aligned(32))) Rec32 {
    double k;            // co-added strength (before threshold / density)
    long long off;       // table element read by output sample 0 (row start + q)
    int ulo, uhi;        // window on the global output grid
    int pad[2];
};

// Record of the staged kernel: 16 bytes per (layer, phase-sorted group).  The window end, the
// row offset q and the phase follow from ulo, len, the cell's half-width and the group's
// fine index (ph_iown), so they are not stored.
struct __attribute__((aligned(16))) Rec16 {
    double k;            // co-added strength (before threshold / density)
    int32_t ulo;         // window start on the global output grid
    uint32_t lc;         // window length (12 bits) | table cell << 12
};

struct LblArgs {
    // Voigt table
    const double *pm;
    const double *flat;
    const int64_t *pm_base;
    const int32_t *pm_stride;
    const int32_t *psize;
    const int32_t *pindex;
    const double *doppler;
    const double *lorentz;
    int ndop, nlor, osamp;
    // lines and groups
    const double *lwn, *elow, *gf;
    const int32_t *lid;
    const int32_t *gfirst, *gcount, *giown;
    const int64_t *iso_gstart;
    // the same groups sorted by (isotope, iown mod osamp, iown) for the staged kernel
    const int32_t *ph_first, *ph_count, *ph_iown;
    const int64_t *ph_start;          // [niso*(osamp+1)+1]
    // coarse position index of the phase-sorted list: ph_bin[(iso*osamp + p)*(nbins+1) + b] =
    // first entry of (iso, p) at or after fine position b * kBinSamples * osamp
    const int32_t *ph_bin;
    int ph_nbins;
    int rowcap;                       // longest phase row of the table (samples)
    int rowlds;                       // longest row (or row chunk) a staged LDS buffer holds
    const int32_t *ph_iso;            // isotope of every phase-sorted group
    const int32_t *giso;              // isotope of every position-sorted group
    // group list that k_records walks (phase-sorted or position-sorted) and whether the
    // gather kernel reads records (1) or derives them itself (0: resolution mode)
    const int32_t *rk_first, *rk_count, *rk_iown, *rk_iso;
    const double *rk_lwn, *rk_elow, *rk_gf;   // the leader line of every group, same order
    const double *g_lead;             // leader lines in position order [3][ngroups]
    int use_records;
    int64_t ngroups;
    // fine-index window of the groups whose records this call needs (a shard +- the largest
    // reach), and whether groups outside it are left out of the per-row maxima too
    int64_t rec_flo, rec_fhi;
    int kmax_local;
    // two-phase shard calls (kmax_local): the groups within [rec_flo, rec_fhi] are runs of the
    // group order k_records walks (one per (isotope, phase) in phase order, one per isotope in
    // position order); thread t of the launch takes group wm_lo[r] + t - wm_off[r] of the run r
    // that holds t, and the launch has wm_total threads per layer block instead of ngroups.
    // Index 0: the order of the phase-walking pass, 1: position order.  Null: every group.
    const int32_t *wm_lo[2];
    const int32_t *wm_off[2];
    int wm_n[2];
    int wm_lds;                       // the offsets of map 0 are copied to LDS (else bisected in global memory)
    int64_t wm_total[2];
    // resident-profile kernel: which layers it computes, its LDS capacity (doubles, 0 = off)
    // and, per isotope, the first position-sorted group at or after every output sample
    int32_t *ls_resident;
    int32_t *ls_block;                // largest phase-major profile block of the layer (doubles)
    int res_cap;
    // wave-autonomous kernel (pb_wave.hip): layers whose longest phase row is at most wave_cap
    // samples are its (ls_wave, decided by k_layer_state; 0 = kernel off) and skipped by the staged one
    int32_t *ls_wave;
    int wave_cap;
    const int32_t *gs_start;          // [niso][nwave+1]
    // scatter kernel: one 32-byte record per (layer, position-sorted group)
    struct Rec32 *rec32;
    // staged kernel: packed records of the layers it computes (null: SoA records)
    Rec16 *rec16;
    // long phase rows (> kStageRowMax samples) are cut into nch_max chunks of kStageRowMax
    // samples; every (group, chunk) then has its own packed record and the gather kernel
    // treats (phase, chunk) as a phase of its own.  Layout of a layer's records:
    // [phase p][chunk k][position] = ps*nch_max + k*cnt_p + (g - ps).  1 = no chunking.
    int nch_max;
    // line lists whose packed records exceed the plan's record budget are walked in CHUNKS of
    // consecutive (isotope, phase) keys of the phase-sorted group list: k_records and the staged
    // gather then see the groups [grp_lo, grp_hi) only (records at rec16[layer*rec_pitch + g -
    // grp_lo]), the gather walks the keys iso*osamp + phase in [key_lo, key_hi) and, from the
    // second chunk on (accumulate), continues the running sums it finds in ext -- the same terms
    // in the same order as one launch over every key.  Unchunked: grp = [0, ngroups),
    // rec_pitch = ngroups, key = [0, niso*osamp), accumulate = 0.
    int64_t grp_lo, grp_hi, rec_pitch;
    int key_lo, key_hi, accumulate;
    // staged kernel, small launches: the phases of a tile are split between nsplit workgroups
    // (see the kernel's block decoding); split 0 writes ext, the others part[split-1][layer][row][sample]
    int nsplit;
    double *part;
    // per-layer phase split of the staged kernel (null: every layer is split nsplit ways): the
    // (layer, split) units of the launch in dispatch order, unit_tab[u] = layer << 8 | split, and
    // the number of pieces of every layer; nsplit is then the largest of them (plane count)
    const int32_t *unit_tab;
    const int32_t *lsplit;
    int nunits;
    // per-tile phase split of the staged kernel (null: off): tile t of every layer is computed by
    // tsplit[t] <= nsplit workgroups; the (tile, split) workgroups beyond it end at once.  For line
    // lists of uneven density (band heads): the tiles under a head hold 10-100 x the records of
    // the others and would end the launch alone.
    const int32_t *tsplit;
    // layers a launch leaves alone (lskip[layer] != 0): the direct gather of a host-free
    // `resolution` call computes only the layers its predicted run plan did not fit
    const int32_t *lskip;
    // tsplit[t] == 0: the tile is too sparse for the staged kernel (a few records per phase row:
    // its time is its ~300 barrier steps whatever they hold) and is computed by the global gather,
    // launched beside it over the tiles so marked; ts_tile = samples per tile of that table;
    // pos2ph[g] = index of position-sorted group g in the phase-sorted list (the packed records'
    // order), for the global gather's record fetch
    int ts_tile;
    const int32_t *pos2ph;
    // per (layer, phase-sorted group) records written by k_records [nlayers][ngroups]
    double *rec_k;                    // co-added strength (before threshold / density)
    int32_t *rec_ulo, *rec_uhi;       // window on the global output grid
    int32_t *rec_q;                   // row index = output sample + q
    int32_t *rec_cell, *rec_phi;      // table cell and phase row
    double inv_osamp;
    int64_t nlines;
    // static species data
    const double *molrad, *molmass;
    const int32_t *isoimol, *isoiext;
    const double *isomass, *isoratio;
    const int32_t *divisors;
    int nmol, niso, ndivs;
    // per-call inputs
    const double *temp, *dens, *isoz;
    int64_t z_iso_stride, z_layer_stride;
    // layer state (workspace)
    int32_t *ls_ofactor, *ls_scale;
    int64_t *ls_dnwn;
    double *ls_dwnstep;
    double *ls_cutsteps, *ls_inv_ofactor, *ls_inv_scale;   // per-layer quotients for k_records
    double *ls_inv_temp;              // 1 / temperature of the layer
    double *li_invz;                  // 1 / partition function of the (layer, isotope)
    double *li_alphad, *li_dens, *li_z;
    int32_t *li_ilor, *li_hmax;
    int32_t *li_rowmax;               // longest phase row the (layer, isotope) can select
    int32_t *li_hlo, *li_hhi;         // smallest / largest profile half-width it can select
    unsigned long long *kmax_bits;
    // grid
    const double *wn;
    double own0, own_last, ownstep, wnstep, wn0;
    int64_t onwn;
    double cutoff, ethresh;
    int add, nrows, nlayers, nwave;
    int64_t wbegin, wcount;
    int ntiles;
    int experiment;      // diagnostics only (PB_EXPERIMENT): 1 = every record reads one slice
    // cycle account of the staged kernel (experiments build, PB_STAGE_PROBE=4): s_memtime ticks per
    // category and layer, probe[layer * 24 + category] (see k_ext_staged); null otherwise
    unsigned long long *probe;
    double *ext;
    // round-staged gather (pb_rounds.hip): tile and LDS row buffer (samples), the largest
    // distance (fine samples) from which a group can reach a tile, per-unit capacity offsets
    // of the visit-record / segment lists [ntiles*nsplit + 1] and the lists themselves
    int rtile, rbuf;
    int64_t reachmax;
    const int64_t *unit_cap;
    VRec *vrec;
    VSeg *vseg;
    int32_t *vrnd;     // first visit record of every round (+ one past the last)
    UnitHdr *uhdr;
};

size_t rounds_prep_lds(const LblArgs &a);
void rounds_geometry(int geom, int *tile, int *rbuf);
int rounds_launch(const LblArgs &a, int geom, hipStream_t s);
size_t wave_lds(const LblArgs &a);
int wave_launch(LblArgs a, int nunits, hipStream_t s);


// first index in [lo,hi) with a[idx] >= v
__device__ inline int64_t lower_bound_i32(const int32_t *a, int64_t lo, int64_t hi, int64_t v)
{
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// two lower bounds in lock step: the loads of the two bisections are independent, so every trip
// has both in flight (the candidate search of the staged gather is a chain of dependent global
// loads: its two bounds one after the other were 21 % of a rank-size launch's slot-time)
__device__ inline void lower_bound2_i32(const int32_t *a, int64_t lo0, int64_t hi0, int64_t v0,
                                        int64_t lo1, int64_t hi1, int64_t v1, int64_t &r0,
                                        int64_t &r1)
{
    while (lo0 < hi0 || lo1 < hi1) {
        const bool go0 = lo0 < hi0, go1 = lo1 < hi1;
        const int64_t mid0 = go0 ? (lo0 + hi0) >> 1 : lo0 - (lo0 > 0);
        const int64_t mid1 = go1 ? (lo1 + hi1) >> 1 : mid0;
        const int32_t x0 = a[mid0], x1 = a[mid1];
        if (go0) {
            if (x0 < v0)
                lo0 = mid0 + 1;
            else
                hi0 = mid0;
        }
        if (go1) {
            if (x1 < v1)
                lo1 = mid1 + 1;
            else
                hi1 = mid1;
        }
    }
    r0 = lo0;
    r1 = lo1;
}

// floor(a / d) for |a| < 2^31 and 0 < d < 2^20, with inv = 1.0/d: (a + 0.5)/d is never an
// integer, so the product cannot round across one.
__device__ inline int floor_div_inv(int a, double inv)
{
    return (int)floor(((double)a + 0.5) * inv);
}

}  // namespace pbx
