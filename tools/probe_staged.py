"""Instrumentation of k_ext_staged for A/B builds -- NOT part of the product build.

Patches pyratbay_amd/csrc/pb_extinction.hip + pb_ext_args.h IN PLACE (restore them with
`git checkout pyratbay_amd/csrc` afterwards) so that `hipcc -DPB_PROBE=<bits>` builds
    bit 0  the kernel without its row DMA         (timing only, results are wrong)
    bit 1  the kernel without the walk            (timing only, results are wrong)
    bit 3  the walk stops after the hit search (find_hits)       (timing only)
    bit 8  start and end time of every workgroup (wall_clock64, 100 MHz) written to
           $PB_PROBE_TIMES (binary, grid x 2 uint64) after every launch: the tail of the launch
    bit 9  (with bit 8) per workgroup also the time spent in the candidate search, in the batch
           set-up (record decode, segment detection) and in the segment steps: $PB_PROBE_TIMES
           then holds grid x 6 uint64 (start, end, search, set-up, steps, batches)
    bit 10 the batch set-up (record decode, segment detection) runs TWICE per batch (idempotent,
           results stay valid): the time it adds is what the set-up costs in throughput
    bit 2  counters printed to stderr after every launch: segment steps, visits, the sum over
           the steps of the busiest wavefront's visits (what a barrier waits for), batches,
           records, and the shader clock the kernel ran at (clock64 against wall_clock64)
Build the object with the flags of csrc/Makefile, link it with the other objects into
build_ab/libpbhip_p<bits>.so and run `PB_PROBE_LIB=build_ab/libpbhip_p<bits>.so python
tools/bench_stages.py c2` (tools/wg_probe_run.py does both for the workgroup time stamps).  Findings of round 2: profiles/r02_gather_ab.md."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

p = os.path.join(ROOT, 'pyratbay_amd/csrc/pb_extinction.hip')
s=open(p).read()

# --- bit 10: batch set-up twice ---
old = """            __syncthreads();
            // ---- one record per lane, in (phase, iown) order, from k_records ----
            long long src = -1;"""
new = """            __syncthreads();
            int nseg = 0;
            for (int rep = 0; rep < ((PB_PROBE & 1024) ? 2 : 1); rep++) {
            if (rep)
                __syncthreads();
            // ---- one record per lane, in (phase, iown) order, from k_records ----
            long long src = -1;"""
assert s.count(old) == 1
s = s.replace(old, new)
old = """            int nseg = 0;
            {
                int before = 0;"""
new = """            nseg = 0;
            {
                int before = 0;"""
assert s.count(old) == 1
s = s.replace(old, new)
old = """            __syncthreads();
            if (nseg == 0)
                continue;"""
new = """            __syncthreads();
            }
            if (nseg == 0)
                continue;"""
assert s.count(old) == 1
s = s.replace(old, new)

h = os.path.join(ROOT, 'pyratbay_amd/csrc/pb_ext_args.h')
t=open(h).read()
t=t.replace("    int experiment;","    unsigned long long *probe;\n    int experiment;",1)
open(h,'w').write(t)
s=s.replace('template <int NW, int S, bool kDma>\n__global__ __launch_bounds__(NW * 64, 8) void k_ext_staged','#ifndef PB_PROBE\n#define PB_PROBE 0\n#endif\ntemplate <int NW, int S, bool kDma>\n__global__ __launch_bounds__(NW * 64, 8) void k_ext_staged',1)
old='''    auto dma_row = [&](int sg, int buf) {
        const unsigned long long d = s_desc[sg];'''
new='''    auto dma_row = [&](int sg, int buf) {
#if PB_PROBE & 1
        return;
#endif
        const unsigned long long d = s_desc[sg];'''
assert old in s
s=s.replace(old,new)
old='''            auto walk = [&](int sg, int buf) {
'''
new='''            auto walk = [&](int sg, int buf) {
#if PB_PROBE & 4
                if ((sg & 63) == 0) find_hits(sg);
                {
                    int cnt = 0;
                    for (int u = 0; u < S; u++)
                        cnt += (int)((unsigned)__builtin_amdgcn_readlane((int)hits[u], sg & 63) >> 16);
                    if (lane == 0) {
                        atomicAdd(&a.probe[1], (unsigned long long)cnt);
                        atomicMax(&s_part[0], cnt);
                        if (wave == 0) atomicAdd(&a.probe[0], 1ull);
                    }
                }
#endif
#if PB_PROBE & 2
                return;
#endif
'''
assert s.count(old)==1
s=s.replace(old,new)
old='''                    walk(sg, sg & 1);
                    __builtin_amdgcn_s_waitcnt(0x0f70);       // the row of sg+1 has landed
                    __syncthreads();'''
new='''                    walk(sg, sg & 1);
                    __builtin_amdgcn_s_waitcnt(0x0f70);       // the row of sg+1 has landed
                    __syncthreads();
#if PB_PROBE & 4
                    if (tid == 0) { atomicAdd(&a.probe[2], (unsigned long long)s_part[0]); s_part[0] = 0; }
                    __syncthreads();
#endif'''
assert old in s
s=s.replace(old,new)
old='''            if (nseg == 0)
                continue;'''
new='''#if PB_PROBE & 4
            if (tid == 0) { atomicAdd(&a.probe[3], 1ull); atomicAdd(&a.probe[4], (unsigned long long)nrec); s_part[0] = 0; }
            __syncthreads();
#endif
            if (nseg == 0)
                continue;'''
assert s.count(old)==1
s=s.replace(old,new)
old='''        kern<<<grid, kStagedThreads, lds, s>>>(a);'''
new='''#if PB_PROBE & 4
        static unsigned long long *probe_d = nullptr;
        if (!probe_d) hipMalloc(&probe_d, 64);
        hipMemsetAsync(probe_d, 0, 64, s);
        a.probe = probe_d;
#endif
        kern<<<grid, kStagedThreads, lds, s>>>(a);
#if PB_PROBE & 4
        { unsigned long long hh[8]; hipMemcpyAsync(hh, probe_d, 64, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
          fprintf(stderr, "probe: segsteps %llu visits %llu sum_max_visits %llu batches %llu recs %llu  grid %u nsplit %d S %d\\n", hh[0], hh[1], hh[2], hh[3], hh[4], grid.x, a.nsplit, S); }
#endif'''
assert s.count(old)==1
s=s.replace(old,new)
i = s.index('void k_ext_staged')
old2 = """    const int row = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
"""
j = s.index(old2, i)
s = s[:j] + old2 + """#if PB_PROBE & 4
    const long long probe_c0 = clock64(), probe_w0 = wall_clock64();
#endif
""" + s[j + len(old2):]
old3 = """    double *out = zsplit == 0
                      ? a.ext
                      : a.part + (int64_t)(zsplit - 1) * a.nlayers * a.nrows * a.wcount;"""
j = s.index(old3, i)
s = s[:j] + """#if PB_PROBE & 4
    if (tid == 0) {
        atomicAdd(&a.probe[5], (unsigned long long)(clock64() - probe_c0));
        atomicAdd(&a.probe[6], (unsigned long long)(wall_clock64() - probe_w0));
    }
#endif
""" + s[j:]
old_fh = """                if ((sg & 63) == 0)
                    find_hits(sg);
                // byte address of this lane's first sample in the staged row"""
assert old_fh in s
s = s.replace(old_fh, """                if ((sg & 63) == 0)
                    find_hits(sg);
#if PB_PROBE & 8
                return;                                   // bit 3: the hit search alone
#endif
                // byte address of this lane's first sample in the staged row""", 1)
s = s.replace('fprintf(stderr, "probe: segsteps %llu', 'fprintf(stderr, "probe: MHz %.0f segsteps %llu', 1)
s = s.replace('hh[0], hh[1], hh[2], hh[3], hh[4], grid.x',
              'hh[6] ? 100.0 * (double)hh[5] / (double)hh[6] : 0.0, hh[0], hh[1], hh[2], hh[3], '
              'hh[4], grid.x', 1)

# --- bit 8: workgroup start / end times ---
old = """    if (layer < 0 || (a.res_cap > 0 && a.ls_resident[layer]))
        return;
    const int row = blockIdx.y;"""
new = """#if PB_PROBE & 256
    if (threadIdx.x == 0 && blockIdx.y == 0) {
        a.probe[2 * blockIdx.x] = (unsigned long long)wall_clock64();
        a.probe[2 * blockIdx.x + 1] = 0ull;
    }
#endif
    if (layer < 0 || (a.res_cap > 0 && a.ls_resident[layer]))
        return;
    const int row = blockIdx.y;"""
i = s.index('void k_ext_staged')
assert s.count(old) >= 1
j = s.index(old, i)
s = s[:j] + new + s[j + len(old):]
old3 = """    double *out = zsplit == 0
                      ? a.ext
                      : a.part + (int64_t)(zsplit - 1) * a.nlayers * a.nrows * a.wcount;"""
j = s.index(old3, i)
s = s[:j] + """#if PB_PROBE & 256
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.y == 0)
        a.probe[2 * blockIdx.x + 1] = (unsigned long long)wall_clock64();
#endif
""" + s[j:]
old = """        kern<<<grid, kStagedThreads, lds, s>>>(a);"""
new = """#if PB_PROBE & 256
        static unsigned long long *probe_t = nullptr;
        static size_t probe_n = 0;
        if (probe_n < 2 * (size_t)grid.x) {
            if (probe_t) hipFree(probe_t);
            probe_n = 2 * (size_t)grid.x;
            hipMalloc(&probe_t, probe_n * 8);
        }
        hipMemsetAsync(probe_t, 0, probe_n * 8, s);
        a.probe = probe_t;
#endif
        kern<<<grid, kStagedThreads, lds, s>>>(a);
#if PB_PROBE & 256
        if (const char *f = getenv("PB_PROBE_TIMES")) {
            std::vector<unsigned long long> hh(2 * (size_t)grid.x);
            hipMemcpyAsync(hh.data(), probe_t, hh.size() * 8, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            if (FILE *fp = fopen(f, "wb")) { fwrite(hh.data(), 8, hh.size(), fp); fclose(fp); }
        }
#endif"""
k = s.index(old)   # first occurrence is already patched by the bit-2 code; patch that site
assert s.count(old) == 1
s = s.replace(old, new)

# --- bit 9: phase timers inside a workgroup (thread 0) ---
s = s.replace("a.probe[2 * blockIdx.x]", "a.probe[PB_PROBE_W * blockIdx.x]")
s = s.replace("a.probe[2 * blockIdx.x + 1]", "a.probe[PB_PROBE_W * blockIdx.x + 1]")
s = s.replace("probe_n < 2 * (size_t)grid.x", "probe_n < PB_PROBE_W * (size_t)grid.x")
s = s.replace("probe_n = 2 * (size_t)grid.x;", "probe_n = PB_PROBE_W * (size_t)grid.x;")
s = s.replace("std::vector<unsigned long long> hh(2 * (size_t)grid.x);", "std::vector<unsigned long long> hh(PB_PROBE_W * (size_t)grid.x);")
s = s.replace("#ifndef PB_PROBE\n#define PB_PROBE 0\n#endif\n", "#ifndef PB_PROBE\n#define PB_PROBE 0\n#endif\n#define PB_PROBE_W ((PB_PROBE & 512) ? 6 : 2)\n", 1)
i = s.index('void k_ext_staged')
old = """    for (int iso = 0; iso < a.niso; iso++) {
        const int iext = a.isoiext[iso];
        if (iext < 0 || (a.add ? 0 : iext) != row)
            continue;"""
j = s.index(old, i)
s = s[:j] + """#if PB_PROBE & 512
    __shared__ long long s_pt[6];           // search, set-up, steps, batches, mark
    if (threadIdx.x == 0) { s_pt[0] = s_pt[1] = s_pt[2] = s_pt[3] = 0; s_pt[4] = wall_clock64(); }
#endif
""" + s[j:]
old = """        for (int x0 = 0; x0 < total; x0 += kThreads) {
            const int nrec = min(kThreads, total - x0);
            __syncthreads();"""
j = s.index(old, i)
s = s[:j] + """#if PB_PROBE & 512
        if (threadIdx.x == 0) { const long long now = wall_clock64(); s_pt[0] += now - s_pt[4]; s_pt[4] = now; }
#endif
""" + s[j:]
old = """            if (nseg == 0)
                continue;
            // ---- rows are double-buffered"""
j = s.index(old, i)
s = s[:j] + """#if PB_PROBE & 512
            if (threadIdx.x == 0) { const long long now = wall_clock64(); s_pt[1] += now - s_pt[4]; s_pt[4] = now; s_pt[3]++; }
#endif
""" + s[j:]
# end of the batch body: after the kDma/else run block -> find the closing of the x0 loop
old = """            } else if (nseg > 0) {
                run(std::integral_constant<int, 2>(), std::integral_constant<int, kRowRegs>());
            }
        }
    }
"""
j = s.index(old, i)
new = """            } else if (nseg > 0) {
                run(std::integral_constant<int, 2>(), std::integral_constant<int, kRowRegs>());
            }
#if PB_PROBE & 512
            if (threadIdx.x == 0) { const long long now = wall_clock64(); s_pt[2] += now - s_pt[4]; s_pt[4] = now; }
#endif
        }
#if PB_PROBE & 512
        if (threadIdx.x == 0) { const long long now = wall_clock64(); s_pt[0] += now - s_pt[4]; s_pt[4] = now; }
#endif
    }
#if PB_PROBE & 512
    if (threadIdx.x == 0 && blockIdx.y == 0) {
        a.probe[PB_PROBE_W * blockIdx.x + 2] = (unsigned long long)s_pt[0];
        a.probe[PB_PROBE_W * blockIdx.x + 3] = (unsigned long long)s_pt[1];
        a.probe[PB_PROBE_W * blockIdx.x + 4] = (unsigned long long)s_pt[2];
        a.probe[PB_PROBE_W * blockIdx.x + 5] = (unsigned long long)s_pt[3];
    }
#endif
"""
s = s[:j] + new + s[j + len(old):]

open(p,'w').write(s)
