// Round-staged extinction gather (constant-step grids, phase rows of <= 1024 samples).
//
// Replaces the profile accumulation + resample of _extcoeff.extinction
// (src_c/_extcoeff.c:300-332), like k_ext_staged in pb_extinction.hip, with the per-segment
// machinery of that kernel taken apart into two launches:
//
//   k_rounds      one 256-thread workgroup per (layer, tile, phase split).  Walks the tile's
//                 candidate (layer, group) records of k_records in (isotope, phase, position)
//                 order, drops the dead ones (below ethresh * kmax, or outside the tile),
//                 finds the SEGMENTS (runs of records that read the same phase row through
//                 the same row window) and deals them to ROUNDS of R rows -- as many rows as
//                 one LDS buffer of the gather kernel holds.  Output per unit: compacted
//                 16-byte visit records {k, LDS byte offset of tile sample 0, tile window},
//                 one 16-byte row descriptor per segment, and a small header.
//   k_ext_rounds  one workgroup per (layer, tile, split), NW wavefronts (geometries in
//                 rounds_launch).  Per round: the R rows AND the records of the NEXT round are
//                 requested by LDS-DMA (`buffer_load ... lds`, 1-KiB pieces dealt to the
//                 wavefronts round-robin), the records of THIS round are tested 64 at a time
//                 (lane l: record l of the LDS record buffer), every wavefront ballots the
//                 records that reach each of its 256-sample spans and walks the set bits: one
//                 16-byte broadcast read of {k, row offset}, 4 `ds_read_b64`, 4 `v_fma_f64`.
//                 ONE barrier per round of R rows instead of one per row; no segment table, no
//                 bisections, nothing fetched from global memory inside a round.
//
// The sums are accumulated in registers in (isotope, phase, position) order -- the order of
// k_ext_staged -- whatever the tile size, so the two kernels agree bit for bit at equal
// phase split.
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <vector>

#include "pb_ext_args.h"

using namespace pbx;

namespace pbx {

struct __attribute__((aligned(16))) VRec {
    double k;          // co-added strength (x density when add), never below the threshold
    int32_t off;       // byte offset, from the start of a row buffer, of the row position
                       // that tile sample 0 reads
    uint32_t win;      // lo | hi << 16, tile coordinates
};

struct __attribute__((aligned(16))) VSeg {
    // the row window as the first three words of a buffer resource (base address of its first
    // element, 48 bits; byte length) + the even sample the LDS image starts at
    uint32_t base_lo, base_hi, bytes;
    int32_t mlo2;              // window start & ~1
};

struct __attribute__((aligned(16))) UnitHdr {
    int32_t nrec, nseg, pitch, rows;   // records, segments, slot pitch (samples), rows per round
};

}  // namespace pbx

namespace {

constexpr int kRT = 256;              // threads of k_rounds

// ---------------------------------------------------------------------------
// k_rounds
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kRT) void k_rounds(LblArgs a)
{
    constexpr int NWv = kRT / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    const int osamp = a.osamp;
    // 8-byte arrays first, then the 4-byte ones
    long long *s_cbase = reinterpret_cast<long long *>(smem);    // [ndop] pm_base
    double *s_k = reinterpret_cast<double *>(s_cbase + a.ndop);  // compacted live records ...
    long long *s_src = reinterpret_cast<long long *>(s_k + kRT);
    unsigned *s_m = reinterpret_cast<unsigned *>(s_src + kRT);
    unsigned *s_win = s_m + kRT;
    int *s_q = reinterpret_cast<int *>(s_win + kRT);             // ... of one batch
    int *s_csize = s_q + kRT;                                    // [ndop] psize
    int *s_cstride = s_csize + a.ndop;                           // [ndop] pm_stride
    int *s_part = s_cstride + a.ndop;                            // [NWv] + [NWv]
    int *s_cum = s_part + 2 * NWv;                               // [osamp+1]
    int *s_phs = s_cum + (osamp + 1);                            // [osamp]
    __shared__ int s_rowmax;

    const int T = a.rtile;
    int tile, layer, zsplit;
    {
        const int id = blockIdx.x;
        const int k = id >> 3;
        tile = k % a.ntiles;
        const int grp = k / a.ntiles;                  // snake order over the XCDs
        const int unit = grp * 8 + ((grp & 1) ? 7 - (id & 7) : (id & 7));
        layer = a.nlayers - 1 - unit / a.nsplit;
        zsplit = unit % a.nsplit;
    }
    if (layer < 0)
        return;
    const int row = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t t0 = a.wbegin + (int64_t)tile * T;
    const int64_t tend = min(t0 + T, a.wbegin + a.wcount);
    const int64_t uidx = ((int64_t)(layer * a.nrows + row) * a.ntiles + tile) * a.nsplit + zsplit;
    const int64_t cap0 = a.unit_cap[tile * a.nsplit + zsplit];
    const int64_t lstride = a.unit_cap[a.ntiles * a.nsplit];
    const int64_t ubase = (int64_t)(layer * a.nrows + row) * lstride + cap0;
    VRec *vrec = a.vrec + ubase;
    VSeg *vseg = a.vseg + ubase;
    int32_t *vrnd = a.vrnd + ubase;

    const double kthresh =
        a.ethresh * __longlong_as_double((long long)a.kmax_bits[(int64_t)layer * a.nrows + row]);
    const int64_t recbase = (int64_t)layer * a.ngroups;

    // slot pitch of this unit: the longest phase row any isotope of the row can select on
    // the tile's wavenumber range (one Doppler column of margin on both sides)
    if (tid == 0)
        s_rowmax = 0;
    __syncthreads();
    for (int iso = tid; iso < a.niso; iso += kRT) {
        const int iext = a.isoiext[iso];
        if (iext < 0 || (a.add ? 0 : iext) != row)
            continue;
        const int64_t li = (int64_t)layer * a.niso + iso;
        const double wlo = a.own0 + (double)max((int64_t)0, t0 * osamp - a.reachmax) * a.ownstep;
        const double whi = a.own0 + (double)min(a.onwn - 1, tend * osamp + a.reachmax) * a.ownstep;
        const int dlo = max(0, pb::nearest_index(a.doppler, a.li_alphad[li] * wlo, 0, a.ndop - 1) - 1);
        const int dhi = min(a.ndop - 1,
                            pb::nearest_index(a.doppler, a.li_alphad[li] * whi, 0, a.ndop - 1) + 1);
        int m = 0;
        for (int d = dlo; d <= dhi; d++)
            m = max(m, a.pm_stride[a.li_ilor[li] * a.ndop + d]);
        atomicMax(&s_rowmax, min(m, a.rowlds));
    }
    __syncthreads();
    const int rowmax128 = (max(s_rowmax, 1) + 127) & ~127;
    const int pitch = rowmax128 + kStagePad;
    const int rows = max(1, min(32, (a.rbuf - kStagePad) / pitch));

    // carried over the batches
    int nrec_out = 0, nseg_out = 0;
    long long last_src = -1;
    unsigned last_m = 0xffffffffu;

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

        __syncthreads();
        const int cell0 = a.li_ilor[li] * a.ndop;
        for (int d = tid; d < a.ndop; d += kRT) {
            s_cbase[d] = a.pm_base[cell0 + d];
            s_csize[d] = a.psize[cell0 + d];
            s_cstride[d] = a.pm_stride[cell0 + d];
        }
        // candidates of every phase: [s_phs[p], s_phs[p] + count) in the phase list, then an
        // exclusive scan of the counts (thread t owns a run of `per` phases)
        const int per = (osamp + kRT - 1) / kRT;
        int mine = 0;
        for (int r = 0; r < per; r++) {
            const int p = tid * per + r;
            if (p < osamp && a.nsplit > 1 && p * a.nsplit / osamp != zsplit) {
                s_phs[p] = 0;
                s_cum[p] = 0;
            } else if (p < osamp) {
                const int32_t *bin = a.ph_bin + ((int64_t)iso * osamp + p) * (a.ph_nbins + 1);
                const int64_t binw = (int64_t)kBinSamples * osamp;
                const int b0 = (int)min((int64_t)a.ph_nbins - 1, max((int64_t)0, flo) / binw);
                const int b1 = (int)min((int64_t)a.ph_nbins - 1, max((int64_t)0, fhi + 1) / binw);
                const int64_t s0 = lower_bound_i32(a.ph_iown, bin[b0], bin[b0 + 1], flo);
                const int64_t s1 = lower_bound_i32(a.ph_iown, bin[b1], bin[b1 + 1], fhi + 1);
                s_phs[p] = (int)s0;
                s_cum[p] = (int)(s1 - s0);
                mine += (int)(s1 - s0);
            }
        }
        int incl = mine;
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
        for (int w = 0; w < NWv; w++) {
            const int pw = s_part[w];
            if (w < wave)
                base += pw;
            total += pw;
        }
        int run = base + incl - mine;
        for (int r = 0; r < per; r++) {
            const int p = tid * per + r;
            if (p < osamp) {
                const int c = s_cum[p];
                s_cum[p] = run;
                run += c;
            }
        }
        if (tid == 0)
            s_cum[osamp] = total;
        __syncthreads();

        for (int x0 = 0; x0 < total; x0 += kRT) {
            // ---- phase A: one candidate per thread ----
            double k = 0.0;
            unsigned win = 0, mwin = 0;
            int qpos = 0;
            long long src = -1;
            bool live = false;
            const int x = x0 + tid;
            if (x < total) {
                int plo = 0, pup = osamp;              // largest p with s_cum[p] <= x
                while (pup - plo > 1) {
                    const int mid = (plo + pup) >> 1;
                    if (s_cum[mid] <= x)
                        plo = mid;
                    else
                        pup = mid;
                }
                const int64_t gidx = s_phs[plo] + (x - s_cum[plo]);
                const Rec16 r = a.rec16[recbase + gidx];
                k = r.k;
                const int ulo = r.ulo;
                const int uhi = ulo + (int)(r.lc & 0xfffu);
                const int cell = (int)(r.lc >> 12);
                const int d = s_csize[cell - cell0] - a.ph_iown[gidx];     // half - iown
                const int q = floor_div_inv(d, a.inv_osamp);
                const int phi = d - q * osamp;
                const int lo = (int)(max((int64_t)ulo, t0) - t0);
                const int hi = (int)(min((int64_t)uhi, tend) - t0);
                if (!(k < kthresh) && lo < hi) {
                    live = true;
                    if (a.add)
                        k *= dens;
                    win = (unsigned)lo | ((unsigned)hi << 16);
                    // tile sample j reads row sample j + t0 + q, stored at LDS position
                    // (row sample) - e, e = parity of the window start (see k_ext_rounds)
                    qpos = (int)(q + t0) - ((ulo + q) & 1);
                    src = s_cbase[cell - cell0] + (long long)phi * s_cstride[cell - cell0];
                    mwin = (unsigned)(ulo + q) | ((unsigned)(uhi + q) << 16);
                }
            }
            // compaction of the live records, order kept
            const unsigned long long lm = __ballot(live);
            if (lane == 0)
                s_part[wave] = __builtin_popcountll(lm);
            __syncthreads();
            int before = 0, nlive = 0;
            for (int w = 0; w < NWv; w++) {
                const int c = s_part[w];
                if (w < wave)
                    before += c;
                nlive += c;
            }
            if (live) {
                const int ci = before + __builtin_popcountll(lm & ((1ull << lane) - 1ull));
                s_k[ci] = k;
                s_src[ci] = src;
                s_m[ci] = mwin;
                s_win[ci] = win;
                s_q[ci] = qpos;
            }
            __syncthreads();
            // ---- phase B: one live record per thread ----
            bool start = false;
            if (tid < nlive) {
                const long long psrc = tid == 0 ? last_src : s_src[tid - 1];
                const unsigned pm = tid == 0 ? last_m : s_m[tid - 1];
                start = s_src[tid] != psrc || s_m[tid] != pm;
            }
            const unsigned long long sm = __ballot(start);
            if (lane == 0)
                s_part[NWv + wave] = __builtin_popcountll(sm);
            __syncthreads();
            int sbefore = 0, nstart = 0;
            for (int w = 0; w < NWv; w++) {
                const int c = s_part[NWv + w];
                if (w < wave)
                    sbefore += c;
                nstart += c;
            }
            if (tid < nlive) {
                // ordinal of my segment among the unit's segments
                const int ord = nseg_out + sbefore +
                                __builtin_popcountll(sm & ((2ull << lane) - 1ull)) - 1;
                const int slot = ord % rows;
                VRec v;
                v.k = s_k[tid];
                v.off = (kStagePad + slot * pitch + s_q[tid]) * 8;
                v.win = s_win[tid];
                vrec[nrec_out + tid] = v;
                if (start) {
                    const unsigned long long mlo = s_m[tid] & 0xffffu, mhi = s_m[tid] >> 16;
                    const unsigned long long base =
                        (unsigned long long)(a.pm + (s_src[tid] + (long long)mlo));
                    VSeg sg;
                    sg.base_lo = (uint32_t)base;
                    sg.base_hi = (uint32_t)((base >> 32) & 0xffffu);
                    sg.bytes = (uint32_t)((mhi - mlo) * 8);
                    sg.mlo2 = (int32_t)(mlo & ~1ull);
                    vseg[ord] = sg;
                    if (slot == 0)
                        vrnd[ord / rows] = nrec_out + tid;     // first record of a round
                }
            }
            if (nlive > 0) {
                last_src = s_src[nlive - 1];
                last_m = s_m[nlive - 1];
            }
            nrec_out += nlive;
            nseg_out += nstart;
            __syncthreads();
        }
    }
    if (tid == 0) {
        vrnd[(nseg_out + rows - 1) / rows] = nrec_out;
        UnitHdr h;
        h.nrec = nrec_out;
        h.nseg = nseg_out;
        h.pitch = pitch;
        h.rows = rows;
        a.uhdr[uidx] = h;
    }
}

// ---------------------------------------------------------------------------
// k_ext_rounds
// ---------------------------------------------------------------------------
constexpr int kRecMax = 256;      // records of a round kept in LDS (the rest: global, on demand)

template <int NW, int S, int OCC>
__global__ __launch_bounds__(NW * 64, OCC) void k_ext_rounds(LblArgs a)
{
    constexpr int kThreads = NW * 64;
    constexpr int kSub = NW * kStageSpan;         // samples per sub-tile
    constexpr int kT = S * kSub;                  // samples per workgroup
    static_assert(kT < 65536, "window coordinates are packed in 16 bits");
    extern __shared__ __align__(16) unsigned char smem[];
    double *s_row = reinterpret_cast<double *>(smem);     // two buffers of a.rbuf samples
    VRec *s_rec = reinterpret_cast<VRec *>(s_row + 2 * a.rbuf);   // two buffers of kRecMax

    int tile, layer, zsplit;
    {
        const int id = blockIdx.x;
        const int k = id >> 3;
        tile = k % a.ntiles;
        const int grp = k / a.ntiles;                  // snake order over the XCDs
        const int unit = grp * 8 + ((grp & 1) ? 7 - (id & 7) : (id & 7));
        layer = a.nlayers - 1 - unit / a.nsplit;       // < 0 for the padding blocks
        zsplit = unit % a.nsplit;
    }
    if (layer < 0)
        return;
    const int row = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int64_t t0 = a.wbegin + (int64_t)tile * kT;
    const int64_t tend = min(t0 + kT, a.wbegin + a.wcount);
    const int tlen = (int)(tend - t0);

    const int64_t uidx = ((int64_t)(layer * a.nrows + row) * a.ntiles + tile) * a.nsplit + zsplit;
    const int64_t cap0 = a.unit_cap[tile * a.nsplit + zsplit];
    const int64_t lstride = a.unit_cap[a.ntiles * a.nsplit];
    const int64_t ubase = (int64_t)(layer * a.nrows + row) * lstride + cap0;
    const VRec *__restrict__ vrec = a.vrec + ubase;
    const VSeg *__restrict__ vseg = a.vseg + ubase;
    const int32_t *__restrict__ vrnd = a.vrnd + ubase;
    const UnitHdr hdr = a.uhdr[uidx];
    const int nseg = __builtin_amdgcn_readfirstlane(hdr.nseg);
    const int pitch = __builtin_amdgcn_readfirstlane(hdr.pitch);
    const int rows = __builtin_amdgcn_readfirstlane(hdr.rows);     // <= 32
    const int npiece = (pitch - kStagePad) / 128;      // 1 KiB DMA pieces per row
    const int nrounds = (nseg + rows - 1) / rows;

    double acc[S][4];
#pragma unroll
    for (int u = 0; u < S; u++)
        acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.0;

    if (nseg > 0) {
        for (int i = tid; i < 2 * a.rbuf; i += kThreads)
            s_row[i] = 0.0;                        // the pads stay zero for good

        // Everything a round needs is requested one round ahead: its rows and its first
        // kRecMax records go global -> LDS by DMA, its row descriptors (lane l: segment l of
        // the round) into `dnext`; lane l of `rst` holds the first record of round rst0 + l.
        int rst = 0, rst0 = 0;
        auto load_rst = [&](int r0) {
            rst0 = r0;
            rst = vrnd[min(r0 + lane, nrounds)];
        };
        auto round_rec = [&](int r) -> int {       // rst0 <= r < rst0 + 64
            return __builtin_amdgcn_readlane(rst, r - rst0);
        };
        const VSeg noseg = {0u, 0u, 0u, 0};
        auto load_desc = [&](int r) -> VSeg {
            const int sg = r * rows + lane;
            return (lane < rows && sg < nseg) ? vseg[sg] : noseg;
        };
        // one 1-KiB piece: 64 lanes x 16 B from buffer `rsrc` at byte offset voff -> LDS dst.
        // Lanes outside the descriptor's range write zeros.  Issued from inline asm (M0 saved
        // and restored in the same statement); the wait before the barrier is explicit.
        typedef int v4i __attribute__((ext_vector_type(4)));
        auto piece = [&](v4i rsrc, int voff, unsigned dst) {
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
        };
        const unsigned lds_row0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)(
            reinterpret_cast<char *>(s_row));
        const unsigned lds_rec0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)(
            reinterpret_cast<char *>(s_rec));
        // Rows and records of round r -> buffers r & 1.  The pieces (rows x npiece row pieces,
        // then the record pieces) are dealt to the wavefronts round-robin.  A row's descriptor
        // IS its window; the upper bound is checked per dword, a lane below the window is out
        // of range as a whole, so the LDS image is shifted by the parity e of the window start
        // (row sample m sits at position m - e, which the records' offsets account for) and no
        // 16-byte lane straddles the lower edge.
        // One step of requests, dealt to the wavefronts round-robin in 1-KiB pieces: the rows of
        // round `r` (nrow > 0) -> slots of row buffer r & 1, and the records [qb, qe) -> record
        // buffer `rbuf`.  A row's descriptor IS its window; the upper bound is checked per dword,
        // a lane below the window is out of range as a whole, so the LDS image is shifted by the
        // parity e of the window start (row sample m sits at position m - e, which the records'
        // offsets account for) and no 16-byte lane straddles the lower edge.
        auto dma_step = [&](int r, int nrow, const VSeg &dall, int qb, int qe, int rbuf) {
            const int nrp = nrow * npiece;
            const int total = nrp + ((qe - qb + 63) >> 6);
            for (int q = wave; q < total; q += NW) {
                v4i rsrc;
                int voff;
                unsigned dst;
                if (q < nrp) {
                    const int sl = q / npiece, c = q - sl * npiece;
                    rsrc.x = __builtin_amdgcn_readlane((int)dall.base_lo, sl);
                    rsrc.y = __builtin_amdgcn_readlane((int)dall.base_hi, sl);
                    rsrc.z = __builtin_amdgcn_readlane((int)dall.bytes, sl);
                    voff = (2 * lane - __builtin_amdgcn_readlane(dall.mlo2, sl)) * 8 + c * 1024;
                    dst = lds_row0 + (unsigned)(((r & 1) * a.rbuf + kStagePad + sl * pitch) * 8 +
                                                c * 1024);
                } else {
                    const int c = q - nrp;
                    const unsigned long long base = (unsigned long long)(vrec + qb);
                    rsrc.x = (int)base;
                    rsrc.y = (int)((base >> 32) & 0xffffu);
                    rsrc.z = (qe - qb) * 16;
                    voff = lane * 16 + c * 1024;
                    dst = lds_rec0 + (unsigned)(rbuf * kRecMax * 16 + c * 1024);
                }
                rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x);
                rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y);
                rsrc.z = __builtin_amdgcn_readfirstlane(rsrc.z);
                rsrc.w = 0x00020000;
                piece(rsrc, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)dst));
            }
        };

        // One batch of 64 records of the LDS record buffer (lane l tests record l): every span of
        // this wavefront ballots the records that reach it and walks the set bits.  The strength
        // and the row offset of a hit come from a 16-byte BROADCAST read of its record (all
        // lanes the same address): no cross-lane VALU, the fma takes k from a VGPR.
        const char *lbase[S];
        auto pop_bit = [&](unsigned long long &m) -> int {     // lowest set bit, cleared
            const int i = (int)__builtin_ctzll(m);
            asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(i));
            return i;
        };
        auto process = [&](const VRec *batch) {         // LDS, record 0 of the batch
            const unsigned win = batch[lane].win;
            const int lo = (int)(win & 0xffffu), hi = (int)(win >> 16);
#pragma unroll
            for (int u = 0; u < S; u++) {
                const int slo = u * kSub + wave * kStageSpan;
                unsigned long long m = __ballot(lo < slo + kStageSpan && hi > slo);
                const double2 *bq = reinterpret_cast<const double2 *>(batch);
                auto rowp = [&](const double2 &raw) {
                    return reinterpret_cast<const double *>(lbase[u] + __double2loint(raw.y));
                };
                // four visits per trip while four bits are left
                while (__builtin_popcountll(m) >= 4) {
                    const int i0 = pop_bit(m), i1 = pop_bit(m), i2 = pop_bit(m), i3 = pop_bit(m);
                    const double2 w0 = bq[i0], w1 = bq[i1], w2 = bq[i2], w3 = bq[i3];
                    const double *p0 = rowp(w0), *p1 = rowp(w1), *p2 = rowp(w2), *p3 = rowp(w3);
                    const double a00 = p0[0], a01 = p0[64], a02 = p0[128], a03 = p0[192];
                    const double a10 = p1[0], a11 = p1[64], a12 = p1[128], a13 = p1[192];
                    const double a20 = p2[0], a21 = p2[64], a22 = p2[128], a23 = p2[192];
                    const double a30 = p3[0], a31 = p3[64], a32 = p3[128], a33 = p3[192];
                    acc[u][0] = fma(w0.x, a00, acc[u][0]);
                    acc[u][1] = fma(w0.x, a01, acc[u][1]);
                    acc[u][2] = fma(w0.x, a02, acc[u][2]);
                    acc[u][3] = fma(w0.x, a03, acc[u][3]);
                    acc[u][0] = fma(w1.x, a10, acc[u][0]);
                    acc[u][1] = fma(w1.x, a11, acc[u][1]);
                    acc[u][2] = fma(w1.x, a12, acc[u][2]);
                    acc[u][3] = fma(w1.x, a13, acc[u][3]);
                    acc[u][0] = fma(w2.x, a20, acc[u][0]);
                    acc[u][1] = fma(w2.x, a21, acc[u][1]);
                    acc[u][2] = fma(w2.x, a22, acc[u][2]);
                    acc[u][3] = fma(w2.x, a23, acc[u][3]);
                    acc[u][0] = fma(w3.x, a30, acc[u][0]);
                    acc[u][1] = fma(w3.x, a31, acc[u][1]);
                    acc[u][2] = fma(w3.x, a32, acc[u][2]);
                    acc[u][3] = fma(w3.x, a33, acc[u][3]);
                }
                // then pairs; a missing second visit repeats the first with k = 0 (adds
                // exactly nothing)
                while (m) {
                    const int i0 = pop_bit(m);
                    const bool h1 = m != 0;
                    const int i1 = h1 ? pop_bit(m) : i0;
                    const double2 w0 = bq[i0];
                    double2 w1 = bq[i1];
                    if (!h1)
                        w1.x = 0.0;
                    const double *p0 = rowp(w0), *p1 = rowp(w1);
                    const double a00 = p0[0], a01 = p0[64], a02 = p0[128], a03 = p0[192];
                    const double a10 = p1[0], a11 = p1[64], a12 = p1[128], a13 = p1[192];
                    acc[u][0] = fma(w0.x, a00, acc[u][0]);
                    acc[u][1] = fma(w0.x, a01, acc[u][1]);
                    acc[u][2] = fma(w0.x, a02, acc[u][2]);
                    acc[u][3] = fma(w0.x, a03, acc[u][3]);
                    acc[u][0] = fma(w1.x, a10, acc[u][0]);
                    acc[u][1] = fma(w1.x, a11, acc[u][1]);
                    acc[u][2] = fma(w1.x, a12, acc[u][2]);
                    acc[u][3] = fma(w1.x, a13, acc[u][3]);
                }
            }
        };

        load_rst(0);
        VSeg dnext = load_desc(0);
        __syncthreads();                                   // buffers zeroed
        dma_step(0, min(rows, nseg), dnext, 0, min(round_rec(1), kRecMax), 0);
        dnext = load_desc(1);
        __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0)
        __syncthreads();
        int gc = 0;                                        // chunks so far: record buffer gc & 1
        for (int r = 0; r < nrounds; r++) {
            if (r - rst0 >= 32)
                load_rst(r);                       // (a wait, once per 32 rounds)
            const int rb = round_rec(r), re = round_rec(r + 1), re2 = round_rec(r + 2);
#pragma unroll
            for (int u = 0; u < S; u++)
                lbase[u] = reinterpret_cast<const char *>(s_row + (r & 1) * a.rbuf +
                                                          u * kSub + wave * kStageSpan + lane);
            // the records of a round pass through the LDS record buffers in chunks of kRecMax
            // (one chunk unless the line list is dense), a barrier after every chunk
            for (int cb = rb; cb == rb || cb < re; cb += kRecMax, gc++) {
                const int ce = min(cb + kRecMax, re);
                // requests of this step: the rows of the next round (with the first chunk) and
                // the next chunk of records
                int qb = ce, qe = min(ce + kRecMax, re), nrow = 0;
                if (ce >= re) {
                    qb = re;
                    qe = min(re + kRecMax, re2);
                }
                if (cb == rb && r + 1 < nrounds)
                    nrow = min(rows, nseg - (r + 1) * rows);
                if (!(a.experiment & 2))
                    dma_step(r + 1, nrow, dnext, qb, qe, (gc + 1) & 1);
                if (cb == rb && r + 1 < nrounds)
                    dnext = load_desc(r + 2);
                if (!(a.experiment & 1)) {
                    const VRec *lrec = s_rec + (gc & 1) * kRecMax;
                    for (int j = 0; j < ce - cb; j += 64)
                        process(lrec + j);
                }
                __builtin_amdgcn_s_waitcnt(0x0f70);       // the requests of this step are in
                __syncthreads();
            }
        }
    }

    double *out = zsplit == 0
                      ? a.ext
                      : a.part + (int64_t)(zsplit - 1) * a.nlayers * a.nrows * a.wcount;
    double *dst = out + ((int64_t)layer * a.nrows + row) * a.wcount + (t0 - a.wbegin);
#pragma unroll
    for (int u = 0; u < S; u++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int j = u * kSub + wave * kStageSpan + c * 64 + lane;
            if (j < tlen)
                dst[j] = acc[u][c];
        }
    }
}

}  // namespace

namespace pbx {

size_t rounds_prep_lds(const LblArgs &a)
{
    return (size_t)a.ndop * 16 + (size_t)kRT * (8 + 8 + 4 + 4 + 4) + 2 * (kRT / 64) * 4 +
           (2 * (size_t)a.osamp + 1) * 4 + 16;
}

// Launches k_rounds + k_ext_rounds for the argument block `a` (ntiles, nsplit, rtile, rbuf,
// unit_cap, vrec, vseg, vrnd, uhdr set by the caller).  Geometries (wavefronts x 256-sample
// spans per wavefront = tile; samples per LDS row buffer; workgroups per CU):
struct Geom {
    int nw, spans, rbuf, per_cu;
    void (*kern)(LblArgs);
};
static const Geom kGeoms[] = {
    {16, 2, 4096, 2, k_ext_rounds<16, 2, 8>},      // 0: tile 8192
    {16, 2, 8192, 1, k_ext_rounds<16, 2, 4>},      // 1
    {8, 2, 4096, 2, k_ext_rounds<8, 2, 4>},        // 2: tile 4096
    {8, 2, 2048, 4, k_ext_rounds<8, 2, 8>},        // 3
    {16, 4, 4096, 2, k_ext_rounds<16, 4, 8>},      // 4: tile 16384
    {16, 4, 8192, 1, k_ext_rounds<16, 4, 4>},      // 5
    {8, 4, 4096, 2, k_ext_rounds<8, 4, 4>},        // 6: tile 8192
    {8, 4, 2048, 4, k_ext_rounds<8, 4, 8>},        // 7
};
constexpr int kNGeoms = (int)(sizeof(kGeoms) / sizeof(kGeoms[0]));

void rounds_geometry(int geom, int *tile, int *rbuf)
{
    const Geom &g = kGeoms[std::max(0, std::min(kNGeoms - 1, geom))];
    *tile = g.nw * g.spans * kStageSpan;
    *rbuf = g.rbuf;
}

int rounds_launch(const LblArgs &a, int geom, hipStream_t s)
{
    const Geom &g = kGeoms[std::max(0, std::min(kNGeoms - 1, geom))];
    const int unit_groups = (a.nlayers * a.nsplit + 7) / 8;
    dim3 grid((unsigned)(8 * a.ntiles * unit_groups), a.nrows);
    k_rounds<<<grid, kRT, rounds_prep_lds(a), s>>>(a);
    PB_LAUNCH_CHECK();
    const size_t lds = (size_t)2 * a.rbuf * 8 + (size_t)2 * kRecMax * 16;
    if (lds > 64 * 1024)
        PB_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g.kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    g.kern<<<grid, g.nw * 64, lds, s>>>(a);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

}  // namespace pbx
