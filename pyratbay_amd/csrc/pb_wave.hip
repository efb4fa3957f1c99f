// Wave-autonomous extinction gather for SHORT phase rows (constant-step grids).
//
// Replaces the profile accumulation + resample of _extcoeff.extinction
// (src_c/_extcoeff.c:281-332) for the layers whose phase rows are at most kWvRowMax samples
// long -- the Doppler-core layers of an atmosphere: 49 of the 80 layers of BASELINE config 2,
// where a line adds ~130-300 samples -- like k_ext_staged (pb_extinction.hip) does for the rest.
//
// Why another kernel.  k_ext_staged shares every staged row between the eight wavefronts of a
// workgroup: one `s_barrier` per phase row, between two barriers a wavefront has one or two
// (record, 256-sample span) visits, and every step waits for its busiest wavefront
// (profiles/r02_gather_ab.md: 64 % of the wave-cycles parked, LDS 33 % busy).  Here NOTHING is
// shared inside the segment loop:
//
//   * the workgroup's kWvWaves wavefronts all own the WHOLE tile of 2048 output samples (32
//     accumulators per lane) and split its PHASES between them: wavefront w takes the phase
//     rows p with p mod kWvWaves == w.  The assignment depends on the phase alone, not on the
//     tiling, so wavenumber shards still concatenate bit for bit.
//   * each wavefront finds its own candidates, decodes its own records 64 at a time, detects
//     its own segments (runs of records that read one row window) with wave-level ballots, and
//     stages its own rows by LDS-DMA (`buffer_load_dwordx4 ... lds`) into a PRIVATE pair of row
//     slots, one segment ahead, waiting only for its own `s_waitcnt vmcnt(N)`.  No workgroup
//     barrier between the per-isotope set-up and the final sum.
//   * a visit covers the 64-sample CHUNKS the window reaches and no others: the record carries
//     its first chunk c0 (wave-uniform), a `switch` over c0 enters straight-line code with the
//     accumulators as compile-time register names; 3 chunks always, 2 + 2 more when the window
//     reaches them.  (k_ext_staged visits 256-sample spans: 45 % useful lanes on these layers,
//     here ~75 %.)  A row slot is [64 zeros][window image][zeros]: lanes outside the window
//     read zeros, no predicate.
//   * at the end the wavefronts' partial sums are added in wavefront order through LDS.
//
// Every sample's terms are added in one fixed order -- per wavefront (phase, position) order,
// then ((w0 + w1) + w2) + w3 -- so a call is bitwise reproducible; against k_ext_staged the
// association of a sample's terms differs (~1e-16 relative).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "pb_ext_args.h"

using namespace pbx;

namespace {

constexpr int kWvWaves = 4;                      // wavefronts per workgroup
constexpr int kWvThreads = kWvWaves * 64;
constexpr int kWvBatch = 32;                      // records per batch (two buffers per wavefront)
constexpr int kWvChunks = kWvTile / 64;          // accumulators per lane
constexpr int kWvVisit = 7;                      // chunks a visit can reach (window <= 385 samples)
constexpr int kWvPad = 64;                       // zero samples in front of / behind an image
constexpr int kWvImage = kWvRowMax;              // 3 LDS-DMA pieces of 128 samples
constexpr int kWvSlot = kWvPad + kWvImage;       // slot pitch; the pad behind is the next one's front
constexpr int kWvRowArea = 2 * kWvSlot + kWvPad; // doubles per wavefront: [Z][A][Z][B][Z]
static_assert(kWvImage % 128 == 0, "whole LDS-DMA pieces");
static_assert(64 * (kWvVisit - 1) + 1 >= kWvRowMax, "a visit must cover the longest window");
static_assert(kWvVisit * 64 <= kWvImage + kWvPad, "the last chunk of a visit stays inside the slot");
static_assert(kWvChunks % (2 * kWvWaves) == 0, "final sum: 8 chunks per round, 2 per wavefront");

struct __align__(16) WRec {
    double k;                 // strength (x density when add); never below the threshold
    int qoffb;                // -8 * (window start in tile coordinates): byte offset of tile sample 0
    int cinfo;                // byte offset of the visit's code block from its dispatch (pb_wave_visit.inc)
};

#include "pb_wave_visit.inc"

// kProbe (PB_WV_PROBE=1, diagnostics): per-wavefront cycle counts of the stages, summed over the
// launch and printed by wave_launch.
template <bool kProbe>
__global__ __launch_bounds__(kWvThreads, 4) void k_ext_wave(LblArgs a, unsigned long long *probe)
{
    // search, decode (issue + finish), dma issue, dma wait, walk, batches, segments, records
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = [&]() -> unsigned long long { return kProbe ? (unsigned long long)clock64() : 0ull; };
    extern __shared__ __align__(16) unsigned char smem[];
    const int osamp = a.osamp;
    double *s_row = reinterpret_cast<double *>(smem);                         // [NW][kWvRowArea]
    WRec *s_rec = reinterpret_cast<WRec *>(s_row + kWvWaves * kWvRowArea);    // [NW][2][kWvBatch]
    unsigned long long *s_desc = reinterpret_cast<unsigned long long *>(s_rec + kWvThreads);
    unsigned *s_seg = reinterpret_cast<unsigned *>(s_desc + kWvThreads);      // i0 | i1 << 16
    int *s_wcum = reinterpret_cast<int *>(s_seg + kWvThreads);                // [NW][64]
    int *s_wphs = s_wcum + kWvThreads;                                        // [NW][64]
    long long *s_cbase = reinterpret_cast<long long *>(s_wphs + kWvThreads);  // [ndop]
    int *s_csize = reinterpret_cast<int *>(s_cbase + a.ndop);                 // [ndop]
    int *s_cstride = s_csize + a.ndop;                                        // [ndop]

    // workgroup -> (tile, layer, phase split): the decoding of k_ext_staged (XCD snake)
    int tile, layer, zsplit;
    {
        const int id = blockIdx.x;
        const int k = id >> 3;
        tile = k % a.ntiles;
        const int grp = k / a.ntiles;
        const int unit = grp * 8 + ((grp & 1) ? 7 - (id & 7) : (id & 7));
        if (a.unit_tab) {
            const int e = unit < a.nunits ? a.unit_tab[unit] : -1;
            layer = e < 0 ? -1 : e >> 8;
            zsplit = e & 0xff;
        } else {
            layer = a.nlayers - 1 - unit / a.nsplit;   // < 0 for the padding blocks
            zsplit = unit % a.nsplit;
        }
    }
    if (layer < 0 || !a.ls_wave[layer])
        return;
    const int nsp = a.lsplit ? a.lsplit[layer] : a.nsplit;    // pieces of this layer's tiles
    const int row = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int64_t t0 = a.wbegin + (int64_t)tile * kWvTile;
    const int64_t tend = min(t0 + kWvTile, a.wbegin + a.wcount);
    const int tlen = (int)(tend - t0);
    const double kthresh =
        a.ethresh * __longlong_as_double((long long)a.kmax_bits[(int64_t)layer * a.nrows + row]);
    const int64_t recbase = (int64_t)layer * a.rec_pitch - a.grp_lo;
    double *const out = zsplit == 0
                            ? a.ext
                            : a.part + (int64_t)(zsplit - 1) * a.nlayers * a.nrows * a.wcount;
    double *const dst = out + ((int64_t)layer * a.nrows + row) * a.wcount + (t0 - a.wbegin);
    const int lbits = a.nch_max > 1 ? 14 : 12;

    double acc[kWvChunks];
#pragma unroll
    for (int c = 0; c < kWvChunks; c++)
        acc[c] = 0.0;

    // this wavefront's private areas
    double *const w_row = s_row + wave * kWvRowArea;
    WRec *const w_rec = s_rec + wave * 64;                   // two buffers of kWvBatch records
    unsigned long long *const w_desc = s_desc + wave * 64;
    unsigned *const w_seg = s_seg + wave * 64;
    int *const w_cum = s_wcum + wave * 64;
    int *const w_phs = s_wphs + wave * 64;
    for (int i = lane; i < kWvRowArea; i += 64)
        w_row[i] = 0.0;                            // pads (and unused image tails) stay zero
    int pieces_prev = 0;
    const unsigned w_row_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(
        __attribute__((address_space(3))) char *)(reinterpret_cast<char *>(w_row)));

    // The row window of a segment -> a slot, by LDS-DMA: piece c brings image samples
    // [128 c, 128 c + 128), lane l two of them (16 bytes), all pieces from ONE address pair (the
    // instruction offset moves the source and the LDS destination together).  The buffer
    // descriptor IS the window: samples beyond its length arrive as zeros.  Inline asm: left to
    // the builtin the compiler drains every DMA (vmcnt(0)) before the next LDS read; M0 is
    // saved and restored.
    const int ablate = a.experiment;               // PB_EXPERIMENT (timing only): 16 = no DMA, 32 = no walk
    auto dma_row = [&](int ent, int slot, int pieces) {
        if (ablate & 16)
            return;
        const unsigned long long d = w_desc[ent];
        const unsigned dlo = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
        const unsigned dhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(d >> 32));
        const long long first = ((long long)(dhi & 0xffu) << 32) | dlo;
        const int len = (int)(dhi >> 8);
        const unsigned long long base = (unsigned long long)(a.pm + first);
        typedef int v4i __attribute__((ext_vector_type(4)));
        v4i rsrc;
        rsrc.x = __builtin_amdgcn_readfirstlane((int)base);
        rsrc.y = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffffu));
        rsrc.z = __builtin_amdgcn_readfirstlane(len * 8);
        rsrc.w = 0x00020000;
        const int voff = lane * 16;
        const unsigned dstb = w_row_lds + (unsigned)((slot * kWvSlot + kWvPad) * 8);
        unsigned keep;
        if (pieces == 1)
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(dstb), "v"(voff), "s"(rsrc) : "memory");
        else if (pieces == 2)
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                         "buffer_load_dwordx4 %2, %3, 0 offen offset:1024 lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(dstb), "v"(voff), "s"(rsrc) : "memory");
        else
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                         "buffer_load_dwordx4 %2, %3, 0 offen offset:1024 lds\n\t"
                         "buffer_load_dwordx4 %2, %3, 0 offen offset:2048 lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(dstb), "v"(voff), "s"(rsrc) : "memory");
    };
    // wait until all but the `young` most recent vector-memory operations have completed
    auto wait_vm = [&](int young) {
        switch (young) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        }
    };

    for (int iso = 0; iso < a.niso; iso++) {
        const int iext = a.isoiext[iso];
        if (iext < 0 || (a.add ? 0 : iext) != row)
            continue;
        const int64_t li = (int64_t)layer * a.niso + iso;
        const double dens = a.li_dens[li];
        int64_t reach = a.li_hmax[li];
        if (a.cutoff > 0.0)
            reach = min(reach, (int64_t)(a.cutoff / a.ownstep) + 2 * (int64_t)a.ls_ofactor[layer] + 2);
        reach += osamp + a.ls_ofactor[layer];
        const int64_t flo = t0 * osamp - reach, fhi = (tend - 1) * osamp + reach;

        __syncthreads();                           // every wavefront is done with the last isotope's cells
        const int cell0 = a.li_ilor[li] * a.ndop;  // first cell of the isotope's Lorentz row
        for (int d = tid; d < a.ndop; d += kWvThreads) {
            s_cbase[d] = a.pm_base[cell0 + d];
            s_csize[d] = a.psize[cell0 + d];
            s_cstride[d] = a.pm_stride[cell0 + d];
        }
        __syncthreads();
        // LDS-DMA pieces per row of this (layer, isotope); a shorter image than the last
        // isotope's leaves stale samples behind it: zero the area again
        const int pieces = (min(a.li_rowmax[li], kWvRowMax) + 127) / 128;
        if (pieces < pieces_prev) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA of mine is still landing
            for (int i = lane; i < kWvRowArea; i += 64)
                w_row[i] = 0.0;
        }
        pieces_prev = pieces;

        // phases of this wavefront: p = (64 pr + lane) kWvWaves + wave
        for (int pr = 0; pr * 64 * kWvWaves < osamp; pr++) {
            const unsigned long long ts0 = tick();
            const int p = (pr * 64 + lane) * kWvWaves + wave;
            int first = 0, cnt = 0;
            if (p < osamp && (nsp == 1 || p * nsp / osamp == zsplit)) {
                // two table lookups bracket each bound to within one bin, then a short bisection
                const int32_t *bin = a.ph_bin + ((int64_t)iso * osamp + p) * (a.ph_nbins + 1);
                const int64_t binw = (int64_t)kBinSamples * osamp;
                const int b0 = (int)min((int64_t)a.ph_nbins - 1, max((int64_t)0, flo) / binw);
                const int b1 = (int)min((int64_t)a.ph_nbins - 1, max((int64_t)0, fhi + 1) / binw);
                const int32_t l0 = bin[b0], h0 = bin[b0 + 1], l1 = bin[b1], h1 = bin[b1 + 1];
                int64_t s0, s1;
                lower_bound2_i32(a.ph_iown, l0, h0, flo, l1, h1, fhi + 1, s0, s1);
                first = (int)s0;
                cnt = (int)(s1 - s0);
            }
            int incl = cnt;                        // inclusive scan over the wavefront's phases
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d)
                    incl += up;
            }
            const int total = __builtin_amdgcn_readlane(incl, 63);
            w_phs[lane] = first;
            w_cum[lane] = incl - cnt;
            pc[0] += tick() - ts0;
            if (total == 0)
                continue;

            // ---- records in batches of kWvBatch, one per lane of the lower half, in (phase,
            // position) order.  A batch is decoded in two steps: `issue` finds the lane's record
            // and requests it (two loads, clamped instead of predicated), `finish` -- one batch of
            // walking later -- turns it into a visit record, detects the SEGMENTS (runs of
            // records that read one row window) and lists the live ones.
            struct Raw {
                Rec16 r;
                int iown;
                int ok;
            };
            auto issue = [&](int b) {
                Raw w;
                const int x = b * kWvBatch + lane;
                w.ok = lane < kWvBatch && x < total;
                const int xc = min(x, total - 1);
                int plo = 0, pup = 64;             // largest l with w_cum[l] <= xc
                while (pup - plo > 1) {
                    const int mid = (plo + pup) >> 1;
                    if (w_cum[mid] <= xc)
                        plo = mid;
                    else
                        pup = mid;
                }
                const int64_t entry = w_phs[plo] + (xc - w_cum[plo]);
                w.r = a.rec16[recbase + entry];
                w.iown = a.ph_iown[entry];
                return w;
            };
            auto finish = [&](const Raw &w, int buf) -> int {
                double k = w.r.k;
                int qoffb = 0, cinfo = 0;
                unsigned long long key = ~0ull;    // first table element | window length << 40
                const int ulo = w.r.ulo;
                const int len = (int)(w.r.lc & ((1u << lbits) - 1u));
                const int cell = (int)(w.r.lc >> lbits);
                const int dd = s_csize[cell - cell0] - w.iown;             // half - iown
                const int q = floor_div_inv(dd, a.inv_osamp);
                const int phi = dd - q * osamp;
                const int wlo = (int)((int64_t)ulo - t0), whi = wlo + len;
                if (w.ok && !(k < kthresh) && max(wlo, 0) < min(whi, tlen) && len <= kWvRowMax) {
                    if (a.add)
                        k *= dens;
                    const int c0 = wlo >> 6;                   // floor: may be negative
                    const int c1 = (min(whi, tlen) - 1) >> 6;
                    qoffb = -wlo * 8;
                    // block of the visit code: 3 (c0 + 6) + (0, 1, 2 for 3, 5, 7 chunks)
                    cinfo = 3 * (c0 + 6) + (c1 >= c0 + 5 ? 2 : c1 >= c0 + 3 ? 1 : 0);
                    cinfo = cinfo * 128 + 12;                  // byte offset from the dispatch
                    const long long src = s_cbase[cell - cell0] +
                                          (long long)phi * s_cstride[cell - cell0];
                    key = (unsigned long long)(src + (long long)(ulo + q)) |
                          ((unsigned long long)len << 40);
                } else {
                    k = 0.0;
                }
                if (lane < kWvBatch) {
                    WRec rec;
                    rec.k = k;
                    rec.qoffb = qoffb;
                    rec.cinfo = cinfo;
                    w_rec[buf * kWvBatch + lane] = rec;
                }
                const unsigned klo = (unsigned)key, khi = (unsigned)(key >> 32);
                const unsigned plo = (unsigned)__shfl_up((int)klo, 1);
                const unsigned phi2 = (unsigned)__shfl_up((int)khi, 1);
                const bool start = w.ok && (lane == 0 || plo != klo || phi2 != khi);
                const bool live = start && key != ~0ull;
                const unsigned long long startmask = __ballot(start);
                const unsigned long long okmask = __ballot(w.ok != 0);
                const unsigned long long livemask = __ballot(live);
                if (live) {
                    const unsigned long long m = startmask & ~((2ull << lane) - 1ull);
                    const int end = m ? (int)__builtin_ctzll(m) : __builtin_popcountll(okmask);
                    const int pos = __builtin_popcountll(livemask & ((1ull << lane) - 1ull));
                    w_seg[buf * kWvBatch + pos] =
                        (unsigned)(buf * kWvBatch + lane) | ((unsigned)(buf * kWvBatch + end) << 16);
                    w_desc[buf * kWvBatch + pos] = key;
                }
                return __builtin_popcountll(livemask);
            };
            // one segment: every record of it against the tile
            auto walk = [&](int ent, int slot) {
                if (ablate & 32)
                    return;
                const unsigned sd = (unsigned)__builtin_amdgcn_readfirstlane((int)w_seg[ent]);
                const int i0 = (int)(sd & 0xffffu), i1 = (int)(sd >> 16);
                // LDS byte address of this lane's sample of chunk 0 when the window starts at
                // tile sample 0 (the record's qoffb moves it)
                const unsigned rowa = w_row_lds + (unsigned)((slot * kWvSlot + kWvPad + lane) * 8);
                const double2 *recs = reinterpret_cast<const double2 *>(w_rec);
                double2 nxt = recs[i0];
                for (int i = i0; i < i1; i++) {
                    const double2 cur = nxt;
                    if (i + 1 < i1)
                        nxt = recs[i + 1];             // the next record's read overlaps this visit
                    wave_visit(acc, rowa + (unsigned)__double2loint(cur.y), cur.x,
                               __builtin_amdgcn_readfirstlane(__double2hiint(cur.y)));
                }
                pc[7] += i1 - i0;
            };

            const int nb = (total + kWvBatch - 1) / kWvBatch;
            unsigned long long td = tick();
            Raw raw = issue(0);
            int nseg = finish(raw, 0);
            pc[1] += tick() - td;
            int s = 0;                             // segments walked so far: segment s uses slot s & 1
            if (nseg > 0)
                dma_row(0, 0, pieces);
            for (int b = 0; b < nb; b++) {
                const int buf = b & 1;
                const bool more = b + 1 < nb;
                td = tick();
                if (more)
                    raw = issue(b + 1);            // in flight while this batch is walked
                pc[1] += tick() - td;
                pc[5] += 1;
                pc[6] += nseg;
                int nseg_next = 0;
                bool next_ready = !more;
                for (int sg = 0; sg < nseg; sg++, s++) {
                    const unsigned long long t0s = tick();
                    int young;                     // vector-memory operations younger than this segment's DMA
                    if (sg + 1 < nseg) {
                        dma_row(buf * kWvBatch + sg + 1, (s + 1) & 1, pieces);
                        young = pieces + (sg == 0 && more ? 2 : 0);
                    } else {
                        // the last segment of the batch: the next batch becomes ready and its
                        // first row is requested before this one is walked
                        young = 0;
                        if (more) {
                            const unsigned long long tf = tick();
                            nseg_next = finish(raw, buf ^ 1);
                            pc[1] += tick() - tf;
                            next_ready = true;
                            if (nseg_next > 0) {
                                dma_row((buf ^ 1) * kWvBatch, (s + 1) & 1, pieces);
                                young = pieces;
                            }
                        }
                    }
                    const unsigned long long t1s = tick();
                    pc[2] += t1s - t0s;
                    wait_vm(young);
                    const unsigned long long t2s = tick();
                    pc[3] += t2s - t1s;
                    walk(buf * kWvBatch + sg, s & 1);
                    pc[4] += tick() - t2s;
                }
                if (!next_ready) {                 // (a batch without live records)
                    td = tick();
                    nseg_next = finish(raw, buf ^ 1);
                    pc[1] += tick() - td;
                    if (nseg_next > 0)
                        dma_row((buf ^ 1) * kWvBatch, s & 1, pieces);
                }
                nseg = nseg_next;
            }
        }
    }

    if (kProbe && lane == 0)
        for (int i = 0; i < 8; i++)
            atomicAdd(&probe[i], pc[i]);
    // ---- the wavefronts' partial sums, added in wavefront order; 8 chunks per round ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double *red = s_row;                           // [kWvWaves][8][64]
#pragma unroll
    for (int r = 0; r < kWvChunks / 8; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            red[(wave * 8 + i) * 64 + lane] = acc[8 * r + i];
        __syncthreads();
#pragma unroll
        for (int i2 = 0; i2 < 8 / kWvWaves; i2++) {
            const int i = wave * (8 / kWvWaves) + i2;
            double sum = red[i * 64 + lane];
#pragma unroll
            for (int v = 1; v < kWvWaves; v++)
                sum += red[(v * 8 + i) * 64 + lane];
            const int j = (8 * r + i) * 64 + lane;
            if (j < tlen)
                dst[j] = sum;
        }
        __syncthreads();
    }
}

}  // namespace

namespace pbx {

size_t wave_lds(const LblArgs &a)
{
    return (size_t)kWvWaves * kWvRowArea * 8 + (size_t)kWvThreads * (16 + 8 + 4 + 4 + 4) +
           (size_t)a.ndop * 16 + 16;
}

// One workgroup per (tile of kWvTile samples, layer, phase split); layers that are not the wave
// kernel's (LblArgs::ls_wave, decided by k_layer_state) end at once.  `a` is the argument block
// of the staged launch (its phase split and unit table are shared: both kernels write planes of
// one set of partial sums); only the tiling is this kernel's own.
int wave_launch(LblArgs a, int nunits, hipStream_t s)
{
    a.ntiles = pb::div_up(a.wcount, kWvTile);
    const int unit_groups = (nunits + 7) / 8;
    const size_t lds = wave_lds(a);
    const bool probe = getenv("PB_WV_PROBE") && atoi(getenv("PB_WV_PROBE")) != 0;
    void (*kern)(LblArgs, unsigned long long *) = probe ? k_ext_wave<true> : k_ext_wave<false>;
    if (lds > 64 * 1024)
        PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)(8 * a.ntiles * unit_groups), a.nrows);
    unsigned long long *d = nullptr;
    if (probe) {
        PB_HIP(hipMalloc(&d, 8 * sizeof(unsigned long long)));
        PB_HIP(hipMemsetAsync(d, 0, 8 * sizeof(unsigned long long), s));
    }
    kern<<<grid, kWvThreads, lds, s>>>(a, d);
    PB_LAUNCH_CHECK();
    if (probe) {
        unsigned long long h[8];
        PB_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s));
        PB_HIP(hipStreamSynchronize(s));
        (void)hipFree(d);
        fprintf(stderr, "k_ext_wave probe (wavefront cycles, 100 MHz ticks x ?): search %llu decode %llu "
                "dma-issue %llu dma-wait %llu walk %llu | batches %llu segments %llu records %llu\n",
                h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
    return PB_OK;
}

}  // namespace pbx
