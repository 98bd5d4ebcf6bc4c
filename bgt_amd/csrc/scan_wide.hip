// Wide cohorts: the row index kernel and the team-mode instantiations with many columns per thread
// (512 threads = two waves per SIMD, up to 256 VGPRs per lane).
#include "scan_device.inc.h"

namespace bgth {

// ----------------------------------------------------------------------------------------------------
// Row index: one wave walks one RLE string (the same 4-bytes-per-lane decode as the scan kernel) and
// records where every 256-byte chunk starts in the row, and the carries at every 8192-position boundary.
// Built once per file; lets the team kernels turn chunks into toggles and run directory trips
// independently of each other.
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowindex_kernel(const uint64_t *__restrict__ rowdesc, const uint8_t *__restrict__ rle,
                                                       int64_t n_str, int m_, int S8, uint32_t *chunkinfo, uint32_t *segc)
{
    const int lane = threadIdx.x & 63;
    const int64_t sidx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sidx >= n_str) return;
    const uint32_t m = (uint32_t)m_;
    const uint64_t desc = rowdesc[sidx];
    const uint64_t off = desc & kDescOffMask;
    const uint32_t len = (uint32_t)(desc >> kDescLenShift);
    const uint32_t *q4 = reinterpret_cast<const uint32_t*>(rle + off);
    uint32_t *sc = segc + (size_t)sidx * (size_t)(S8 + 1);
    uint32_t pos = 0, prevbit = 0, ones = 0;
    bool stop = false;
    if (lane == 0) sc[0] = 0u;

    // boundaries P = 8192 s with start < P <= end lie in (or at the end of) a run [start,end) of bit b
    auto mark = [&](uint32_t start, uint32_t end, uint32_t b, uint32_t ones_before) {
        for (uint32_t P = ((start >> 13) + 1u) << 13; P <= end && P < m; P += 8192u)
            sc[P >> 13] = (ones_before + (b ? P - start : 0u)) | (b << 31);
    };

    for (uint32_t base = 0; base < len; base += 256) {
        if (lane == 0) chunkinfo[((off + base) >> 8) + (uint64_t)sidx] = stop ? kChunkDead : (pos | (prevbit << 31));
        if (stop) continue;
        const uint32_t k0 = base + 4u * (uint32_t)lane;
        const uint32_t w = k0 < len ? q4[(base >> 2) + lane] : 0u;
        const ChunkDecode d = decode_chunk(w, k0, len, lane);
        uint32_t lane_ones = 0;
        const uint32_t incl = wave_incl_add(d.run);
        const uint32_t lane_start = pos + incl - d.run;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t st = lane_start + d.before[i];
            if (d.valid[i] && d.bit[i] && st < m) lane_ones += (st + d.l[i] < m ? d.l[i] : m - st);
        }
        const uint32_t incl1 = wave_incl_add(lane_ones);
        uint32_t ob = ones + incl1 - lane_ones;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t st = lane_start + d.before[i];
            if (d.valid[i] && d.l[i] && st < m) {
                const uint32_t en = st + d.l[i] < m ? st + d.l[i] : m;
                mark(st, en, d.bit[i], ob);
                if (d.bit[i]) ob += en - st;
            }
        }
        const uint32_t lb = chunk_last_bit(d, lane);
        const uint64_t anyvalid = __ballot(d.valid[0]);
        if (anyvalid) prevbit = lb;
        pos += lane63(incl);
        ones += lane63(incl1);
        stop = d.stop;
    }
    // a string that ends before position m leaves the rest of the row at its last bit (the toggle
    // representation of the scan kernels does the same)
    if (lane == 0) {
        if (pos < m) {
            mark(pos, m, prevbit, ones);
            if (prevbit) ones += m - pos;
        }
        sc[S8] = ones;
    }
}

hipError_t launch_rowindex(const uint64_t *rowdesc, const uint8_t *rle, int64_t n_str, int m, int S8,
                           uint32_t *chunkinfo, uint32_t *segc, hipStream_t s)
{
    if (n_str <= 0) return hipSuccess;
    hipLaunchKernelGGL(rowindex_kernel, dim3((unsigned)((n_str + 3) / 4)), dim3(256), 0, s,
                       rowdesc, rle, n_str, m, S8, chunkinfo, segc);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------------
template <int CPT, bool MULTI, bool GT, bool ZP, bool SNAP = false>
static hipError_t launch_one(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
    auto fn = scan_kernel<512, CPT, MULTI, GT, true, ZP, SNAP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(g.workgroups), dim3(512), g.lds_bytes, s, a, a.rowdesc, a.rle, a.chunkinfo, a.segc);
    return hipGetLastError();
}

template <int CPT, bool ZP>
static hipError_t launch_variant(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
    if (g.wpp <= 1) return hipErrorInvalidConfiguration;          // team mode only
    if (a.snap) return (a.G > 1 || a.h0) ? hipErrorInvalidConfiguration : launch_one<CPT, false, false, ZP, true>(a, g, s);   // the image-open pass
    switch ((a.G > 1 ? 2 : 0) | (a.h0 ? 1 : 0)) {
    case 0: return launch_one<CPT, false, false, ZP>(a, g, s);
    case 1: return launch_one<CPT, false, true, ZP>(a, g, s);
    case 2: return launch_one<CPT, true, false, ZP>(a, g, s);
    default: return launch_one<CPT, true, true, ZP>(a, g, s);
    }
}

hipError_t launch_scan_wide(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
#define X(CPT_) if (g.cpt == CPT_) return a.zp ? launch_variant<CPT_, true>(a, g, s) : launch_variant<CPT_, false>(a, g, s);
    BGTH_CPT_512_WIDE(X)
#undef X
    return hipErrorInvalidConfiguration;
}

}  // namespace bgth
