// Handle layouts shared between the translation units of libpbhip.
#pragma once

#include <cstdint>
#include <vector>

#include <hip/hip_runtime.h>

#include "pbhip.h"

constexpr int kPmPad = 1024;   // >= the gather kernel's tile (pb_extinction.hip kTile)

struct pb_voigt {
    int nlor = 0, ndop = 0, osamp = 0, ncell = 0, max_half = 0;
    double dwn = 0.0;
    int64_t nflat = 0;   // samples in the reference layout (sum of 2*half+1)
    int64_t npm = 0;     // samples in the phase-major layout (padded)
    // host mirrors, [nlor*ndop]; aliased cells repeat the previous Doppler column
    std::vector<double> lorentz, doppler;
    std::vector<int32_t> psize, pindex, pm_stride;
    std::vector<int64_t> pm_base;
    // device
    double *d_pm = nullptr;      // phase-major table (= d_pm_alloc + kPmPad)
    double *d_pm_alloc = nullptr; // allocation: kPmPad zero samples before and after; the
                                  // gather kernel's lanes outside a line's window read
                                  // (and discard) up to one tile beyond it
    double *d_flat = nullptr;    // reference-layout table (optional)
    void *d_cells = nullptr;     // per computed cell descriptors
    int64_t *d_flat_bases = nullptr, *d_pm_bases = nullptr;  // [ncell]
    int32_t *d_psize = nullptr, *d_pindex = nullptr, *d_pm_stride = nullptr;  // [nlor*ndop]
    int64_t *d_pm_base = nullptr;
    double *d_lorentz = nullptr, *d_doppler = nullptr;
    std::vector<pb_voigt *> rephased;   // this table cut for other factors (pb_voigt_rephase)
    // a re-cut table is filled one Lorentz row at a time (pb_voigt_ensure_rows): the table it was
    // cut from, the positions of the rows in the contiguous layout [nlor+1], the rows' allocations
    pb_voigt *lazy_parent = nullptr;
    std::vector<int64_t> row_pos;
    std::vector<double *> row_data;
    int64_t lazy_bytes = 0;
    // Rows not filled yet read as ZEROS: the device offsets of their cells point into one block
    // of zeros as long as the longest row (+ pads), so a launch that was planned from a
    // prediction of the rows its layers need (the host-free `resolution` path) can never read
    // outside an allocation -- a wrong prediction costs a recomputed layer, not a fault.
    // d_rowmask[m] = 1 once Lorentz row m is filled (the device-side validity check reads it).
    double *zero_block = nullptr;
    uint8_t *d_rowmask = nullptr;
};

int pb_voigt_ensure_flat(pb_voigt *v, hipStream_t stream);
int pb_voigt_rephase(pb_voigt **out, pb_voigt *src, int osamp, hipStream_t stream);
int pb_voigt_ensure_rows(pb_voigt *v, const int *rows, int n, hipStream_t stream);

struct pb_lines;
int pb_lines_group_device(pb_lines *l, const double *lwn_h, const int32_t *lid_h,
                          const double *own_h);

struct pb_lines {
    int64_t nlines = 0, ngroups = 0, nadd = 0, ninrange = 0;
    int niso = 0;
    int grouped_on_device = 0;                          // co-add groups built by pb_lines.hip
    int64_t onwn = 0;
    double own0 = 0.0, own_last = 0.0, ownstep = 0.0;   // from the own[] array
    std::vector<int64_t> iso_gstart;                    // [niso+1] group segments
    std::vector<int32_t> h_gfirst, h_gcount, h_giown;   // host copies of the groups
    std::vector<double> h_lwn, h_elow, h_gf;            // host copies of the line records
    // device: line records
    double *d_lwn = nullptr, *d_elow = nullptr, *d_gf = nullptr;
    int32_t *d_lid = nullptr;
    // device: co-add groups, sorted by (isotope, iown)
    int32_t *d_gfirst = nullptr, *d_gcount = nullptr, *d_giown = nullptr, *d_giso = nullptr;
    int64_t *d_iso_gstart = nullptr;
};
