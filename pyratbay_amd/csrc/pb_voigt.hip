// Voigt profile table on the device.
//
// Restates vprofile.grid (src_c/vprofile.c:42-114) with voigtn / voigtxy
// (src_c/include/voigt.h:147-359): for every (Lorentz, Doppler) cell with a non-zero
// half-size, an area-normalised profile of 2*half+1 samples spaced `dwn`.
//
// MI355X layout.  The extinction kernel reads, for one line, the samples
//     profile[start + half + osamp*jo - iown],   jo = consecutive output samples,
// i.e. a stride-`osamp` walk through the reference's concatenated table -- one cache
// line per lane.  The table is therefore stored PHASE-MAJOR: cell c holds `osamp`
// sub-lattices, sub-lattice phi = { profile_c[phi + osamp*m] : m = 0.. } contiguous in
// m, so the same walk becomes a unit-stride, fully coalesced read.  It is a pure
// permutation of the reference table (identical values).  The reference layout
// ("flat") is produced on demand for the drop-in vprofile.grid and for the
// resolution-mode kernel.
//
// The reference evaluates Region I of voigtxy with x87 `long double` accumulators;
// gfx950 has no 80-bit type, the kernel uses binary64 (measured effect <= 4e-15
// relative).
#include <cmath>
#include <new>
#include <vector>

#include "pb_common.h"
#include "pb_internal.h"

namespace {

constexpr int kBlock = 256;

// 1/(n!(2n+1)), n = 0..31 (the ferf[] table of voigt.h:60-123); only n <= 29 is used.
__constant__ double c_ferf[32];
bool g_ferf_uploaded[16] = {false};

int upload_ferf()
{
    int dev = 0;
    PB_HIP(hipGetDevice(&dev));
    if (dev < 16 && g_ferf_uploaded[dev])
        return PB_OK;
    double table[32];
    long double fact = 1.0L;
    table[0] = 1.0;
    for (int n = 1; n < 32; n++) {
        fact *= (long double)n;
        table[n] = (double)(1.0L / (fact * (long double)(2 * n + 1)));
    }
    PB_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_ferf), table, sizeof(table)));
    if (dev < 16)
        g_ferf_uploaded[dev] = true;
    return PB_OK;
}

// Re[w(x+iy)] * sqrt(ln2/pi)/alphaD, three regions (voigt.h:147-217)
__device__ double voigt_point(double x, double y, double alphaD)
{
    const double A1 = 0.46131350, A2 = 0.19016350, A3 = 0.09999216, A4 = 1.78449270,
                 A5 = 0.002883894, A6 = 5.52534370;
    const double B1 = 0.51242424, B2 = 0.27525510, B3 = 0.05176536, B4 = 2.72474500;
    const double x2y2 = x * x - y * y;
    const double xy2 = 2 * x * y;
    if (x < 3 && y < 1.8) {
        const int nterms = (x < 1 ? 15 : (int)(6.842 * x + 8.0)) + 1;
        const double c = cos(xy2), s = sin(xy2);
        double pr = y, pi = -x, sr = y, si = -x;
        for (int i = 1; i <= nterms; i++) {
            double qi = pr * xy2 + pi * x2y2;
            double qr = pr * x2y2 - pi * xy2;
            si += qi * c_ferf[i];
            sr += qr * c_ferf[i];
            pi = qi;
            pr = qr;
        }
        return pb::kSqrtLn2Pi / alphaD * exp(-x2y2) *
               (c * (1 - sr * pb::kTwoOSqrtPi) - s * si * pb::kTwoOSqrtPi);
    }
    const double d2 = xy2 * xy2;
    const double nx = xy2 * x;
    if (x < 5 && y < 5) {
        double t1 = x2y2 - A2, t2 = x2y2 - A4, t3 = x2y2 - A6;
        return pb::kSqrtLn2Pi / alphaD *
               (A1 * ((nx - t1 * y) / (t1 * t1 + d2)) + A3 * ((nx - t2 * y) / (t2 * t2 + d2)) +
                A5 * ((nx - t3 * y) / (t3 * t3 + d2)));
    }
    double t1 = x2y2 - B2, t2 = x2y2 - B4;
    return pb::kSqrtLn2Pi / alphaD *
           (B1 * ((nx - t1 * y) / (t1 * t1 + d2)) + B3 * ((nx - t2 * y) / (t2 * t2 + d2)));
}

// Per computed cell: everything voigtn (voigt.h:222-262) derives before its loops.
struct Cell {
    int64_t flat_base;   // start in the reference layout
    int64_t pm_base;     // start in the phase-major layout
    double alphaL, alphaD;
    double halfwidth;    // dwn * half
    double fine;         // spacing of the evaluated points
    int32_t nwn;         // 2*half+1
    int32_t over;        // sub-intervals per output bin
    int32_t mode;        // 0 = point samples (QUICK), 1 = mean over `over` (trapezoid),
                         // 2 = Simpson mean
    int32_t pm_stride;   // samples per phase
};

__device__ inline double cell_point(const Cell &c, double y, int64_t idx)
{
    double x = pb::kSqrtLn2 * fabs(c.fine * idx - c.halfwidth) / c.alphaD;
    return voigt_point(x, y, c.alphaD);
}

// One output sample (voigt.h:271-290 with meanintegSimp :300-331 / meanintegTrap :336-359)
__device__ double cell_sample(const Cell &c, int i)
{
    const double y = pb::kSqrtLn2 * c.alphaL / c.alphaD;
    if (c.mode == 0)
        return cell_point(c, y, i);
    const int over = c.over;
    const int64_t b = (int64_t)i * over;
    if (c.mode == 2) {
        double acc = 0;
        for (int t = 1; t < over; t += 2)
            acc += cell_point(c, y, b + t);
        acc *= 2;
        for (int t = 2; t < over; t += 2)
            acc += cell_point(c, y, b + t);
        acc *= 2;
        acc += cell_point(c, y, b) + cell_point(c, y, b + over);
        return acc / (over * 3.0);
    }
    double acc = 0;
    for (int t = 1; t < over; t++)
        acc += cell_point(c, y, b + t);
    return (acc + (cell_point(c, y, b) + cell_point(c, y, b + over)) / 2.0) / (double)over;
}

__device__ inline int find_cell(const int64_t *bases, int ncell, int64_t pos)
{
    // last cell whose base <= pos
    int lo = 0, hi = ncell;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (bases[mid] <= pos)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// Evaluate the table directly in the phase-major layout (coalesced stores).
__global__ __launch_bounds__(kBlock) void k_voigt_pm(double *pm, const Cell *cells,
                                                     const int64_t *pm_bases, int ncell,
                                                     int64_t npm, int osamp)
{
    int64_t pos = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (pos >= npm)
        return;
    int ci = find_cell(pm_bases, ncell, pos);
    const Cell c = cells[ci];
    int64_t r = pos - c.pm_base;
    int phi = (int)(r / c.pm_stride);
    int m = (int)(r - (int64_t)phi * c.pm_stride);
    int64_t i = (int64_t)phi + (int64_t)osamp * m;
    pm[pos] = (i < c.nwn) ? cell_sample(c, (int)i) : 0.0;
}

// Permutations between the two layouts.
// (positions [pos0, npm) of the layout; pm points to position 0)
__global__ void k_flat_to_pm(double *pm, const double *flat, const Cell *cells,
                             const int64_t *pm_bases, int ncell, int64_t npm, int osamp,
                             int64_t pos0 = 0)
{
    int64_t pos = pos0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (pos >= npm)
        return;
    int ci = find_cell(pm_bases, ncell, pos);
    const Cell c = cells[ci];
    int64_t r = pos - c.pm_base;
    int phi = (int)(r / c.pm_stride);
    int m = (int)(r - (int64_t)phi * c.pm_stride);
    int64_t i = (int64_t)phi + (int64_t)osamp * m;
    pm[pos] = (i < c.nwn) ? flat[c.flat_base + i] : 0.0;
}

__global__ void k_pm_to_flat(double *flat, const double *pm, const Cell *cells,
                             const int64_t *flat_bases, int ncell, int64_t nflat,
                             int osamp)
{
    int64_t pos = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (pos >= nflat)
        return;
    int ci = find_cell(flat_bases, ncell, pos);
    const Cell c = cells[ci];
    int64_t i = pos - c.flat_base;
    if (i >= c.nwn) {
        flat[pos] = 0.0;
        return;
    }
    int phi = (int)(i % osamp);
    int64_t m = i / osamp;
    flat[pos] = pm[c.pm_base + (int64_t)phi * c.pm_stride + m];
}

// Host: the decisions voigtn takes before evaluating (voigt.h:235-262), in binary64
// exactly as the reference does.
void plan_cell(Cell &c, int half, double dwn, double alphaL, double alphaD, int osamp)
{
    const int nwn = 2 * half + 1;
    const double halfwidth = dwn * (long)(nwn / 2);
    const double step = 2.0 * halfwidth / (nwn - 1);
    double fine = alphaD / (50 - 1);
    const bool quick = nwn > 99999;
    int over;
    if (step < fine || quick) {
        over = 1;
        fine = step;
    } else {
        over = (int)(step / fine) + 1;
        if (over & 1)
            over++;
        int64_t nfine = (int64_t)nwn * over + 1;
        fine = 2.0 * halfwidth / (double)(nfine - 1);
    }
    c.alphaL = alphaL;
    c.alphaD = alphaD;
    c.halfwidth = halfwidth;
    c.fine = fine;
    c.nwn = nwn;
    c.over = over;
    c.mode = quick ? 0 : (((over + 1) & 1) ? 2 : 1);
    c.pm_stride = (nwn + osamp - 1) / osamp;
}

// Common set-up of both constructors: resolve aliases, lay out both tables.
int plan_table(pb_voigt *v, const double *lorentz_h, int nlor, const double *doppler_h,
               int ndop, const int32_t *psize_in, const int32_t *pindex_in, double dwn,
               int osamp, std::vector<Cell> &cells)
{
    v->nlor = nlor;
    v->ndop = ndop;
    v->osamp = osamp;
    v->dwn = dwn;
    v->lorentz.assign(lorentz_h, lorentz_h + nlor);
    v->doppler.assign(doppler_h, doppler_h + ndop);
    v->psize.assign((size_t)nlor * ndop, 0);
    v->pindex.assign((size_t)nlor * ndop, 0);
    v->pm_base.assign((size_t)nlor * ndop, 0);
    v->pm_stride.assign((size_t)nlor * ndop, 0);
    int64_t idx = 0, pidx = 0;
    int max_half = 0;
    for (int m = 0; m < nlor; m++) {
        for (int n = 0; n < ndop; n++) {
            const size_t k = (size_t)m * ndop + n;
            int half = psize_in[k];
            bool alias = pindex_in ? (n > 0 && pindex_in[k] == pindex_in[k - 1] &&
                                      psize_in[k] == psize_in[k - 1])
                                   : (half == 0);
            if (half < 0) {
                pb::set_error("voigt: negative half-size at cell (%d,%d)", m, n);
                return PB_ERR_ARG;
            }
            if (!alias) {
                if (half == 0) {
                    pb::set_error("voigt: zero half-size at cell (%d,%d)", m, n);
                    return PB_ERR_ARG;
                }
                Cell c;
                plan_cell(c, half, dwn, lorentz_h[m], doppler_h[n], osamp);
                c.flat_base = pindex_in ? (int64_t)pindex_in[k] : idx;
                c.pm_base = pidx;
                if (c.flat_base + c.nwn > 2147483647LL) {
                    pb::set_error("voigt: table exceeds the reference's 32-bit index");
                    return PB_ERR_UNSUPPORTED;
                }
                cells.push_back(c);
                v->psize[k] = half;
                v->pindex[k] = (int32_t)c.flat_base;
                v->pm_base[k] = c.pm_base;
                v->pm_stride[k] = c.pm_stride;
                idx = c.flat_base + c.nwn;
                pidx += (int64_t)c.pm_stride * osamp;
                if (half > max_half)
                    max_half = half;
            } else {
                if (n == 0) {
                    pb::set_error("voigt: first Doppler column of row %d has size 0", m);
                    return PB_ERR_ARG;
                }
                v->psize[k] = v->psize[k - 1];
                v->pindex[k] = v->pindex[k - 1];
                v->pm_base[k] = v->pm_base[k - 1];
                v->pm_stride[k] = v->pm_stride[k - 1];
            }
        }
    }
    v->nflat = idx;
    v->npm = pidx;
    v->max_half = max_half;
    v->ncell = (int)cells.size();
    return PB_OK;
}

int upload_meta(pb_voigt *v, const std::vector<Cell> &cells)
{
    const size_t ncell = cells.size();
    std::vector<int64_t> fb(ncell), pb_(ncell);
    for (size_t i = 0; i < ncell; i++) {
        fb[i] = cells[i].flat_base;
        pb_[i] = cells[i].pm_base;
    }
    const size_t n2 = (size_t)v->nlor * v->ndop;
    PB_HIP(hipMalloc(&v->d_cells, ncell * sizeof(Cell)));
    PB_HIP(hipMalloc(&v->d_flat_bases, ncell * sizeof(int64_t)));
    PB_HIP(hipMalloc(&v->d_pm_bases, ncell * sizeof(int64_t)));
    PB_HIP(hipMalloc(&v->d_psize, n2 * sizeof(int32_t)));
    PB_HIP(hipMalloc(&v->d_pindex, n2 * sizeof(int32_t)));
    PB_HIP(hipMalloc(&v->d_pm_base, n2 * sizeof(int64_t)));
    PB_HIP(hipMalloc(&v->d_pm_stride, n2 * sizeof(int32_t)));
    PB_HIP(hipMalloc(&v->d_lorentz, v->nlor * sizeof(double)));
    PB_HIP(hipMalloc(&v->d_doppler, v->ndop * sizeof(double)));
    PB_HIP(hipMemcpy(v->d_cells, cells.data(), ncell * sizeof(Cell), hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(v->d_flat_bases, fb.data(), ncell * 8, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(v->d_pm_bases, pb_.data(), ncell * 8, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(v->d_psize, v->psize.data(), n2 * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(v->d_pindex, v->pindex.data(), n2 * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(v->d_pm_base, v->pm_base.data(), n2 * 8, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(v->d_pm_stride, v->pm_stride.data(), n2 * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(v->d_lorentz, v->lorentz.data(), v->nlor * 8, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(v->d_doppler, v->doppler.data(), v->ndop * 8, hipMemcpyHostToDevice));
    return PB_OK;
}

int grid_for(int64_t n, unsigned *out)
{
    int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 2147483647LL) {
        pb::set_error("voigt: table too large for one launch");
        return PB_ERR_UNSUPPORTED;
    }
    *out = (unsigned)blocks;
    return PB_OK;
}

// Allocate the phase-major table with kPmPad zeroed samples on either side.
int alloc_pm(pb_voigt *v, hipStream_t s)
{
    const size_t n = (size_t)v->npm + 2 * kPmPad;
    if (hipMalloc(&v->d_pm_alloc, n * sizeof(double)) != hipSuccess) {
        pb::set_error("voigt: cannot allocate %zu B for the table", n * 8);
        return PB_ERR_NOMEM;
    }
    v->d_pm = v->d_pm_alloc + kPmPad;
    PB_HIP(hipMemsetAsync(v->d_pm_alloc, 0, kPmPad * sizeof(double), s));
    PB_HIP(hipMemsetAsync(v->d_pm + v->npm, 0, kPmPad * sizeof(double), s));
    return PB_OK;
}

}  // namespace

int pb_voigt_ensure_flat(pb_voigt *v, hipStream_t stream)
{
    if (v->d_flat)
        return PB_OK;
    unsigned g;
    int rc = grid_for(v->nflat, &g);
    if (rc)
        return rc;
    PB_HIP(hipMalloc(&v->d_flat, (size_t)v->nflat * sizeof(double)));
    k_pm_to_flat<<<g, kBlock, 0, stream>>>(v->d_flat, v->d_pm, (const Cell *)v->d_cells,
                                           v->d_flat_bases, v->ncell, v->nflat, v->osamp);
    PB_LAUNCH_CHECK();
    return PB_OK;
}

// keep_flat == 2: only the reference layout stays on the device (plans of the `resolution` /
// `wlstep` mode read nothing else); the phase-major table is released once the flat one exists
static int release_phase_major(pb_voigt *v, hipStream_t s)
{
    PB_HIP(hipStreamSynchronize(s));
    (void)hipFree(v->d_pm_alloc);
    v->d_pm_alloc = nullptr;
    v->d_pm = nullptr;
    return PB_OK;
}

// The same table with its phase-major layout cut for another oversampling factor (the per-layer
// dynamic grids of the `resolution` mode: sample d of a grid of step f reads element
// half + f*d - iown of a profile, i.e. phase rows modulo f).  The new handle BELONGS to src (kept
// for every plan that asks for the same factor, destroyed with src) and is filled LAZILY, one
// Lorentz row of the table at a time (pb_voigt_ensure_rows): a layer reads the cells of its own
// Lorentz row only, and the layers that meet a given factor share a few neighbouring rows, so all
// the factors of a plan together hold about one copy of the table, not one each.
int pb_voigt_rephase(pb_voigt **out, pb_voigt *src, int osamp, hipStream_t s)
{
    PB_REQUIRE(out && src && osamp > 0, "pb_voigt_rephase: bad argument");
    *out = nullptr;
    for (pb_voigt *r : src->rephased)
        if (r->osamp == osamp) {
            *out = r;
            return PB_OK;
        }
    int rc = pb_voigt_ensure_flat(src, s);
    if (rc)
        return rc;
    pb_voigt *v = new (std::nothrow) pb_voigt();
    if (!v)
        return PB_ERR_NOMEM;
    std::vector<Cell> cells;
    rc = plan_table(v, src->lorentz.data(), src->nlor, src->doppler.data(), src->ndop,
                    src->psize.data(), src->pindex.data(), src->dwn, osamp, cells);
    if (rc == PB_OK)
        rc = upload_meta(v, cells);
    if (rc != PB_OK) {
        pb_voigt_destroy(v);
        return rc;
    }
    // the positions of the contiguous layout that every Lorentz row would occupy; the rows
    // themselves are allocated when a layer first needs them
    v->lazy_parent = src;
    v->row_pos.assign((size_t)v->nlor + 1, v->npm);
    for (int m = 0; m < v->nlor; m++)
        v->row_pos[(size_t)m] = v->pm_base[(size_t)m * v->ndop];
    v->row_data.assign((size_t)v->nlor, nullptr);
    // the block of zeros that every row not yet filled reads (see pb_internal.h), the anchor of
    // the table's offsets (as in pb_voigt_ensure_rows) and the cells' offsets into the block;
    // the host mirror pm_base keeps the layout positions of unfilled rows (ensure_rows needs them)
    {
        int64_t longest = 0;
        for (int m = 0; m < v->nlor; m++)
            longest = std::max(longest, v->row_pos[(size_t)m + 1] - v->row_pos[(size_t)m]);
        const size_t count = (size_t)longest + 2 * kPmPad;
        auto fail = [&](int code, const char *what) {
            pb::set_error("pb_voigt_rephase: %s", what);
            pb_voigt_destroy(v);
            return code;
        };
        if (hipMalloc(&v->zero_block, count * sizeof(double)) != hipSuccess)
            return fail(PB_ERR_NOMEM, "cannot allocate the block of zeros of unfilled rows");
        if (hipMalloc(&v->d_rowmask, (size_t)std::max(v->nlor, 1)) != hipSuccess)
            return fail(PB_ERR_NOMEM, "cannot allocate the row mask");
        if (hipMemsetAsync(v->zero_block, 0, count * sizeof(double), s) != hipSuccess ||
            hipMemsetAsync(v->d_rowmask, 0, (size_t)std::max(v->nlor, 1), s) != hipSuccess)
            return fail(PB_ERR_HIP, "zeroing failed");
        v->lazy_bytes += (int64_t)count * 8;              // (reported with the table's bytes)
        double *zdata = v->zero_block + kPmPad;
        const uintptr_t span = ((uintptr_t)1 << 39) * sizeof(double);
        const uintptr_t at = reinterpret_cast<uintptr_t>(zdata);
        v->d_pm = reinterpret_cast<double *>(at > span ? at - span : (uintptr_t)0);
        const int64_t zoff = (int64_t)((at - reinterpret_cast<uintptr_t>(v->d_pm)) / sizeof(double));
        if (zoff - kPmPad < 0 || zoff + longest + kPmPad >= ((int64_t)1 << 40))
            return fail(PB_ERR_UNSUPPORTED, "the block of zeros cannot be addressed with 40-bit offsets");
        std::vector<int64_t> based(v->pm_base.size());
        for (int m = 0; m < v->nlor; m++)
            for (int d = 0; d < v->ndop; d++) {
                const size_t k = (size_t)m * v->ndop + d;
                based[k] = zoff + (v->pm_base[k] - v->row_pos[(size_t)m]);
            }
        if (hipMemcpyAsync(v->d_pm_base, based.data(), based.size() * 8, hipMemcpyHostToDevice,
                           s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)           // (`based` is a local)
            return fail(PB_ERR_HIP, "uploading the cell offsets failed");
    }
    src->rephased.push_back(v);
    *out = v;
    return PB_OK;
}

// Fill the Lorentz rows rows[0..n) of a lazily filled table (no-op for the rows already there and
// for tables that are not lazy).  Each row gets its own allocation with kPmPad zero samples on
// either side, and the offsets of its cells (host mirror and device copy) are re-based to it.
// The kernels address a cell as d_pm + offset with offsets that the staged gather packs into 40
// unsigned bits: d_pm of a lazy table is not an allocation but an ANCHOR 2^39 samples (4 TiB)
// below the first row allocated, so every allocation within 4 TiB of that one has an offset in
// range; one beyond (never seen: a process's device allocations lie within ~1 TiB) is refused.
// Rows in use by kernels in flight are never touched: a row is filled before the first launch
// that reads it.
int pb_voigt_ensure_rows(pb_voigt *v, const int *rows, int n, hipStream_t s)
{
    PB_REQUIRE(v && (rows || n == 0), "pb_voigt_ensure_rows: null pointer");
    if (!v->lazy_parent)
        return PB_OK;
    pb_voigt *src = v->lazy_parent;
    for (int i = 0; i < n; i++) {
        const int m = rows[i];
        PB_REQUIRE(m >= 0 && m < v->nlor, "pb_voigt_ensure_rows: row %d outside [0,%d)", m, v->nlor);
        if (v->row_data[(size_t)m])
            continue;
        const int64_t p0 = v->row_pos[(size_t)m], p1 = v->row_pos[(size_t)m + 1];
        const size_t count = (size_t)(p1 - p0) + 2 * kPmPad;
        double *block = nullptr;
        if (hipMalloc(&block, count * sizeof(double)) != hipSuccess) {
            pb::set_error("pb_voigt_ensure_rows: cannot allocate %zu B for Lorentz row %d",
                          count * 8, m);
            return PB_ERR_NOMEM;
        }
        double *data = block + kPmPad;               // position p0 of the layout
        if (!v->d_pm) {
            // (clamped: a first row below 4 TiB anchors the table at address 0 -- the
            // subtraction would wrap and every row then fail the range check for good)
            const uintptr_t span = ((uintptr_t)1 << 39) * sizeof(double);
            const uintptr_t at = reinterpret_cast<uintptr_t>(data);
            v->d_pm = reinterpret_cast<double *>(at > span ? at - span : (uintptr_t)0);
        }
        // offset of layout position 0 of this row's frame from the anchor, in samples
        const int64_t shift =
            (int64_t)((reinterpret_cast<uintptr_t>(data) - reinterpret_cast<uintptr_t>(v->d_pm)) /
                      sizeof(double)) - p0;
        if (reinterpret_cast<uintptr_t>(data) < reinterpret_cast<uintptr_t>(v->d_pm) ||
            shift + p0 - kPmPad < 0 || shift + p1 + kPmPad >= ((int64_t)1 << 40)) {
            (void)hipFree(block);
            pb::set_error("pb_voigt_ensure_rows: an allocation %lld samples from the table's "
                          "anchor cannot be addressed with 40-bit offsets",
                          (long long)(shift + p0));
            return PB_ERR_UNSUPPORTED;
        }
        // (every failure from here on gives the row's allocation back)
        auto fail = [&](const char *what) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(s);
            (void)hipFree(block);
            pb::set_error("pb_voigt_ensure_rows: %s failed for Lorentz row %d", what, m);
            return PB_ERR_HIP;
        };
        if (hipMemsetAsync(block, 0, kPmPad * sizeof(double), s) != hipSuccess ||
            hipMemsetAsync(data + (p1 - p0), 0, kPmPad * sizeof(double), s) != hipSuccess)
            return fail("zeroing the pads");
        unsigned g = 0;
        int rc = grid_for(p1 - p0, &g);
        if (rc) {
            (void)hipFree(block);
            return rc;
        }
        // (the kernel works in layout positions: d_pm_bases / d_cells, not the re-based offsets)
        k_flat_to_pm<<<g, kBlock, 0, s>>>(data - p0, src->d_flat, (const Cell *)v->d_cells,
                                         v->d_pm_bases, v->ncell, p1, v->osamp, p0);
        if (hipGetLastError() != hipSuccess) {
            (void)hipFree(block);
            pb::set_error("pb_voigt_ensure_rows: permutation kernel failed");
            return PB_ERR_HIP;
        }
        const size_t k0 = (size_t)m * v->ndop;
        std::vector<int64_t> based((size_t)v->ndop);
        for (int d = 0; d < v->ndop; d++)           // (aliased cells repeat a base)
            based[(size_t)d] = v->pm_base[k0 + d] + shift;
        if (hipMemcpyAsync(v->d_pm_base + k0, based.data(), (size_t)v->ndop * 8,
                           hipMemcpyHostToDevice, s) != hipSuccess)
            return fail("uploading the cell offsets");
        if (v->d_rowmask && hipMemsetAsync(v->d_rowmask + m, 1, 1, s) != hipSuccess)
            return fail("marking the row");
        if (hipStreamSynchronize(s) != hipSuccess)   // `based` is a local
            return fail("the stream");
        for (int d = 0; d < v->ndop; d++)
            v->pm_base[k0 + d] = based[(size_t)d];
        v->row_data[(size_t)m] = block;
        v->lazy_bytes += (int64_t)count * 8;
    }
    return PB_OK;
}

extern "C" {

int pb_voigt_create(pb_voigt **out, const double *lorentz_h, int nlor,
                    const double *doppler_h, int ndop, const int32_t *psize_h,
                    double dwn, int osamp, int keep_flat, void *stream)
{
    PB_REQUIRE(out && lorentz_h && doppler_h && psize_h, "pb_voigt_create: null pointer");
    PB_REQUIRE(nlor > 0 && ndop > 0 && osamp > 0 && dwn > 0, "pb_voigt_create: bad sizes");
    *out = nullptr;
    pb_voigt *v = new (std::nothrow) pb_voigt();
    if (!v)
        return PB_ERR_NOMEM;
    std::vector<Cell> cells;
    int rc = plan_table(v, lorentz_h, nlor, doppler_h, ndop, psize_h, nullptr, dwn, osamp,
                        cells);
    if (rc == PB_OK)
        rc = upload_ferf();
    if (rc == PB_OK)
        rc = upload_meta(v, cells);
    hipStream_t s = pb::as_stream(stream);
    unsigned g = 0;
    if (rc == PB_OK)
        rc = grid_for(v->npm, &g);
    if (rc == PB_OK)
        rc = alloc_pm(v, s);
    if (rc == PB_OK) {
        k_voigt_pm<<<g, kBlock, 0, s>>>(v->d_pm, (const Cell *)v->d_cells, v->d_pm_bases,
                                       v->ncell, v->npm, osamp);
        if (hipGetLastError() != hipSuccess) {
            pb::set_error("pb_voigt_create: launch failed");
            rc = PB_ERR_HIP;
        }
    }
    if (rc == PB_OK && keep_flat)
        rc = pb_voigt_ensure_flat(v, s);
    if (rc == PB_OK && keep_flat == 2)
        rc = release_phase_major(v, s);
    if (rc == PB_OK && hipStreamSynchronize(s) != hipSuccess) {
        pb::set_error("pb_voigt_create: kernel failed: %s",
                      hipGetErrorString(hipGetLastError()));
        rc = PB_ERR_HIP;
    }
    if (rc != PB_OK) {
        pb_voigt_destroy(v);
        return rc;
    }
    *out = v;
    return PB_OK;
}

int pb_voigt_from_flat(pb_voigt **out, const double *profile_h, int64_t nprofile,
                       const double *lorentz_h, int nlor, const double *doppler_h,
                       int ndop, const int32_t *psize_h, const int32_t *pindex_h,
                       int osamp, int keep_flat, void *stream)
{
    PB_REQUIRE(out && profile_h && lorentz_h && doppler_h && psize_h && pindex_h,
               "pb_voigt_from_flat: null pointer");
    PB_REQUIRE(nlor > 0 && ndop > 0 && osamp > 0, "pb_voigt_from_flat: bad sizes");
    *out = nullptr;
    pb_voigt *v = new (std::nothrow) pb_voigt();
    if (!v)
        return PB_ERR_NOMEM;
    std::vector<Cell> cells;
    int rc = plan_table(v, lorentz_h, nlor, doppler_h, ndop, psize_h, pindex_h, 1.0, osamp,
                        cells);
    if (rc == PB_OK && v->nflat > nprofile) {
        pb::set_error("pb_voigt_from_flat: index/size need %lld samples, profile has %lld",
                      (long long)v->nflat, (long long)nprofile);
        rc = PB_ERR_ARG;
    }
    if (rc == PB_OK)
        rc = upload_meta(v, cells);
    hipStream_t s = pb::as_stream(stream);
    unsigned g = 0;
    if (rc == PB_OK)
        rc = grid_for(v->npm, &g);
    if (rc == PB_OK) {
        if (hipMalloc(&v->d_flat, (size_t)v->nflat * 8) != hipSuccess) {
            pb::set_error("pb_voigt_from_flat: cannot allocate the table");
            rc = PB_ERR_NOMEM;
        } else
            rc = alloc_pm(v, s);
    }
    if (rc == PB_OK) {
        if (hipMemcpyAsync(v->d_flat, profile_h, (size_t)v->nflat * 8, hipMemcpyHostToDevice,
                           s) != hipSuccess) {
            pb::set_error("pb_voigt_from_flat: upload failed");
            rc = PB_ERR_HIP;
        }
    }
    if (rc == PB_OK) {
        k_flat_to_pm<<<g, kBlock, 0, s>>>(v->d_pm, v->d_flat, (const Cell *)v->d_cells,
                                         v->d_pm_bases, v->ncell, v->npm, osamp);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            pb::set_error("pb_voigt_from_flat: permutation kernel failed");
            rc = PB_ERR_HIP;
        }
    }
    if (rc == PB_OK && !keep_flat) {
        (void)hipFree(v->d_flat);
        v->d_flat = nullptr;
    }
    if (rc != PB_OK) {
        pb_voigt_destroy(v);
        return rc;
    }
    *out = v;
    return PB_OK;
}

int pb_voigt_meta(const pb_voigt *v, int32_t *psize_h, int32_t *pindex_h, int64_t *nprofile)
{
    PB_REQUIRE(v, "pb_voigt_meta: null handle");
    const size_t n2 = (size_t)v->nlor * v->ndop;
    if (psize_h)
        memcpy(psize_h, v->psize.data(), n2 * sizeof(int32_t));
    if (pindex_h)
        memcpy(pindex_h, v->pindex.data(), n2 * sizeof(int32_t));
    if (nprofile)
        *nprofile = v->nflat;
    return PB_OK;
}

int pb_voigt_flat_to_host(pb_voigt *v, double *profile_h, int64_t nprofile)
{
    PB_REQUIRE(v && profile_h, "pb_voigt_flat_to_host: null pointer");
    PB_REQUIRE(nprofile >= v->nflat, "pb_voigt_flat_to_host: buffer too small");
    const bool had = v->d_flat != nullptr;
    int rc = pb_voigt_ensure_flat(v, nullptr);
    if (rc)
        return rc;
    PB_HIP(hipDeviceSynchronize());
    PB_HIP(hipMemcpy(profile_h, v->d_flat, (size_t)v->nflat * 8, hipMemcpyDeviceToHost));
    if (!had) {
        (void)hipFree(v->d_flat);
        v->d_flat = nullptr;
    }
    return PB_OK;
}

int64_t pb_voigt_device_bytes(const pb_voigt *v)
{
    if (!v)
        return 0;
    int64_t n = v->lazy_parent
                    ? v->lazy_bytes
                    : ((v->d_pm_alloc ? v->npm + 2 * kPmPad : 0) + (v->d_flat ? v->nflat : 0)) * 8;
    for (const pb_voigt *r : v->rephased)
        n += pb_voigt_device_bytes(r);
    return n;
}

void pb_voigt_destroy(pb_voigt *v)
{
    if (!v)
        return;
    for (pb_voigt *r : v->rephased)
        pb_voigt_destroy(r);
    for (double *block : v->row_data)
        (void)hipFree(block);
    (void)hipFree(v->zero_block);
    (void)hipFree(v->d_rowmask);
    (void)hipFree(v->d_pm_alloc);
    (void)hipFree(v->d_flat);
    (void)hipFree(v->d_cells);
    (void)hipFree(v->d_flat_bases);
    (void)hipFree(v->d_pm_bases);
    (void)hipFree(v->d_psize);
    (void)hipFree(v->d_pindex);
    (void)hipFree(v->d_pm_base);
    (void)hipFree(v->d_pm_stride);
    (void)hipFree(v->d_lorentz);
    (void)hipFree(v->d_doppler);
    delete v;
}

}  // extern "C"
