// pbf_encoder.hip -- the write half of the codec seam on the device: rows of 2-bit codes -> the PBWT run-length
// strings, checkpoints and footer of a .pbf, byte for byte what the reference writer produces
// (pbf_open_w / pbf_write / pbf_close, reference pbwt.c:199-219, :288-311, :264-277; row encoder pbc_enc_core,
// pbwt.c:57-66; run-length bytes pbwt.c:24-36).
//
// The encoder is the mirror image of the scan kernel.  One workgroup per bit plane keeps the RANK of every column
// in registers, complemented as in the scan (the reference moves the permutation array instead).  Per row:
//   scatter    every column with a 1 drops it at its rank into an LDS rank directory {32 bits, ones before}  (ds_or)
//   directory  ones before every 32-bit word (block scan); where runs end (w ^ (w >> 1 | next << 31))
//   emit       the last position of every run goes into one list for the row (index: block scan of the run counts);
//              thread k encodes run k -- its bit is the row's first bit ^ (k & 1) -- at the offset a third scan gives
//   step       rank <- bit ? zeros + ones_before(rank) : rank - ones_before(rank)   (the stable partition; the scan's
//              8-instruction step on the same directory layout)
// Rows are sequential only in appearance -- row r is written in the order rows 0..r-1 left -- and planes are
// independent.  The order at a row is a SORT: columns ordered by their bits read backwards from that row, ties in
// the order they had before.  A call with many rows is therefore cut into UNITS of 4096 rows and runs as
//   A  every unit and plane at once, from the identity order and without emitting: the order L the unit's own rows
//      produce; then which neighbours in L are identical over the whole unit (classes), on a bit-transposed copy
//   B  unit after unit (cheap): the columns in the order before unit k, sorted stably by their class in L_k, are the
//      order after it -- ONE launch for the whole chain (order_chain_kernel: a workgroup per plane, the order resident in
//      LDS, a block radix sort per unit); above 32768 columns the order does not fit and a rocPRIM sort per unit does it;
//      all orders turned into ranks by one kernel at the end
//   C  every unit and plane at once, from its true start order: scatter / directory / emit / step as above.
// A call with one unit is phase C alone.  Above 32768 columns (encode_wide_kernel) the ranks live in memory.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cstring>
#include <chrono>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "../../include/bgt_hip.h"

namespace {

constexpr int kThreads = 1024;                 // one word of the bit-vector per thread: m <= 32768
constexpr int kMaxM = kThreads * 32;             // registers hold the ranks up to here
constexpr int kMaxWideM = kThreads * 32 * 8;     // beyond: ranks in memory, up to 8 directory words per thread (directory in the LDS)
constexpr int kMaxHugeM = kThreads * 32 * 64;    // beyond kMaxWideM: the directory in memory too (encode_huge_kernel), 2,097,152 columns
static_assert(kMaxM < kMaxWideM && kMaxWideM < kMaxHugeM, "three regimes: ranks in registers | ranks in memory | directory in memory too");

#ifdef BGTH_ABLATE
#define ENC_ABLATE(a, bits) ((a).debug & (bits))
#else
#define ENC_ABLATE(a, bits) 0
#endif

struct EncodeArgs {
    const uint8_t *codes;      // [n_rows][CPT * 1024]  bit k of a byte = plane k, zero beyond column m
    int64_t n_rows, row0;      // row0 = file row of codes[0]
    int32_t m, mask;           // mask = (1 << shift) - 1
    int32_t g, unit_rows;      // workgroup (plane, unit) handles rows [unit * unit_rows, ...)
    int32_t stride;            // bytes between rows of codes (a multiple of 4)
    const int32_t *rank_in;    // [n_units][g][m] order before each unit, or NULL = identity
    int32_t *rank_out;         // [n_units][g][m] order after it
    int32_t *perm_out;         // [n_units][g][m] the same as position -> column (phase A), or NULL
    uint8_t *out;              // [n_units][g][cap]
    int64_t cap;
    int64_t *out_len;          // [n_units][g]
    int32_t *row_len;          // [g][n_rows]
    int32_t *snap;             // [g][n_snap][m]   permutation before every row with (row & mask) == 0
    const int32_t *snap_base;  // [n_units] index of the first of them in each unit
    int32_t n_snap;
    uint2 *gdir;               // encode_huge_kernel: [n_units][g][2][1024 WPT + 1] directory entries in memory
    int32_t *status;           // != 0: output capacity exceeded
    int32_t debug;             // profiling build only (make ABLATE=1): BGTH_ENC_DEBUG ablation switches for timing (1 no byte
                               // stores, 2 no run passes, 4 no run list); the shipped encoder compiles them out (ENC_ABLATE = 0)
};

// inclusive prefix sum / prefix maximum over the 64 lanes in the VALU (DPP row shifts + the two row broadcasts);
// lanes without a source read 0, the identity of both
__device__ __forceinline__ uint32_t dpp(uint32_t v, int ctrl, int row_mask)
{
    switch (ctrl) {
    case 0x111: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    case 0x112: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    case 0x114: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    case 0x118: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    case 0x142: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    default:    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    }
}
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v)
{
    v += dpp(v, 0x111, 0xf); v += dpp(v, 0x112, 0xf); v += dpp(v, 0x114, 0xf); v += dpp(v, 0x118, 0xf);
    v += dpp(v, 0x142, 0xa); v += dpp(v, 0x143, 0xc);
    return v;
}
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v)
{
    v = umax(v, dpp(v, 0x111, 0xf)); v = umax(v, dpp(v, 0x112, 0xf)); v = umax(v, dpp(v, 0x114, 0xf));
    v = umax(v, dpp(v, 0x118, 0xf)); v = umax(v, dpp(v, 0x142, 0xa)); v = umax(v, dpp(v, 0x143, 0xc));
    return v;
}
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v)      // value of the lane below, 0 into lane 0
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}
// Workgroup barrier for data exchanged through LDS only: __syncthreads() would also wait for the global loads of the
// next row (prefetched a row ahead) and for the stores of the run-length bytes.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

// bytes of one run (ref pbwt.c:24-36): one below 16, else one per non-zero hex digit
__device__ __forceinline__ uint32_t run_bytes(uint32_t len)
{
    if (len < 16u) return 1u;
    const uint32_t nz = (len | len >> 1 | len >> 2 | len >> 3) & 0x11111111u;
    return (uint32_t)__popc(nz);
}

__device__ __forceinline__ uint32_t put_run(uint8_t *dst, uint32_t len, uint32_t bit)
{
    if (len < 16u) { dst[0] = (uint8_t)(len << 1 | bit); return 1u; }
    uint32_t n = 0;
    for (uint32_t nz = (len | len >> 1 | len >> 2 | len >> 3) & 0x11111111u; nz;) {     // non-zero hex digits, top first
        const uint32_t hb = 31u - (uint32_t)__builtin_clz(nz), digit = hb >> 2;
        nz &= ~(1u << hb);
        dst[n++] = (uint8_t)(((digit << 4) | ((len >> (4u * digit)) & 15u)) << 1 | bit);
    }
    return n;
}

// CPT columns per thread (a multiple of 4); rows of `codes` are CPT * 1024 bytes apart, zero beyond column m, so a
// thread fetches its columns as whole words and the padding columns never set a bit.
template <int CPT, bool EMIT>
__global__ __launch_bounds__(kThreads) void encode_kernel(EncodeArgs a)
{
    // the row in PBWT order as a rank directory, double-buffered over rows: {32 bits, ones before them} per entry, so
    // that one 8-byte read answers a lookup (the layout of the scan kernel)
    __shared__ uint2 dir2[2][kThreads + 1];
    __shared__ uint32_t agg[4][16];             // per wave: ones, runs that end in it, bytes; [3][0]: the row's first bit
    __shared__ uint16_t run_end[EMIT ? kMaxM : 2];   // last position of every run of the row
    const int tid = threadIdx.x, lane = tid & 63, plane = blockIdx.x, unit = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, nw = (m + 31) >> 5;
    constexpr int stride = CPT * kThreads;
    const size_t up = (size_t)unit * a.g + plane;
    const int64_t r_beg = (int64_t)unit * a.unit_rows;
    const int64_t r_end = r_beg + a.unit_rows < a.n_rows ? r_beg + a.unit_rows : a.n_rows;
    uint8_t *out = a.out + up * a.cap;
    const int col0 = tid * CPT;                 // this thread's columns: col0 .. col0 + CPT - 1
    int32_t Q[CPT];                             // COMPLEMENTED ranks, q = ~r (saves an instruction per lookup, see the scan)
    uint32_t nxt[CPT / 4];
#pragma unroll
    for (int i = 0; i < CPT; ++i) Q[i] = ~(col0 + i < m ? (a.rank_in ? a.rank_in[up * m + col0 + i] : col0 + i) : 0);
#pragma unroll
    for (int q = 0; q < CPT / 4; ++q)
        nxt[q] = r_beg < r_end ? reinterpret_cast<const uint32_t*>(a.codes + (size_t)r_beg * stride + col0)[q] : 0u;
    dir2[0][tid] = make_uint2(0u, 0u); dir2[1][tid] = make_uint2(0u, 0u);
    if (tid == 0) { dir2[0][kThreads] = make_uint2(0u, 0u); dir2[1][kThreads] = make_uint2(0u, 0u); }
    lds_barrier();
    const uint32_t valid = tid < nw ? ((tid == nw - 1 && (m & 31)) ? (1u << (m & 31)) - 1u : 0xffffffffu) : 0u;
    const uint32_t last_bit = (tid == nw - 1) ? 1u << ((m - 1) & 31) : 0u;      // the row ends here
    int64_t off = 0;
    int snap_i = EMIT ? a.snap_base[unit] : 0;
    for (int64_t r = r_beg; r < r_end; ++r) {
        uint2 *dir = dir2[r & 1];
        char *dir_m8 = reinterpret_cast<char*>(dir) - 8;   // entry of rank r: dir_m8 - 8 * (q >> 5)
        if (EMIT && ((a.row0 + r) & a.mask) == 0) {         // the 'S' record: S[rank] = column (ref pbwt.c:292-301)
            int32_t *S = a.snap + ((size_t)plane * a.n_snap + snap_i) * m;
#pragma unroll
            for (int i = 0; i < CPT; ++i) if (col0 + i < m) S[~Q[i]] = col0 + i;
            ++snap_i;
        }
        // ---- scatter this row, fetch the next one
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const uint32_t b = (nxt[i >> 2] >> (8 * (i & 3) + plane)) & 1u;
            if (b) atomicOr(reinterpret_cast<uint32_t*>(dir_m8 - 8 * (Q[i] >> 5)), 0x80000000u >> (Q[i] & 31));   // bit r & 31
        }
        lds_barrier();                                    // (1) the bit-vector is complete
        dir2[(r & 1) ^ 1][tid].x = 0u;                      // everyone is done with the previous row's
        const uint32_t w = dir[tid].x;
        const uint32_t wn = dir[tid + 1].x;
        // a run ends at bit i of this word if the next position holds the other bit, or the row ends there
        const uint32_t ends = EMIT ? ((((w ^ (w >> 1 | wn << 31)) & valid) & ~last_bit) | last_bit) : 0u;
        const uint32_t pc = (uint32_t)__popc(w), ne = (uint32_t)__popc(ends);
        const uint32_t incl = wave_incl_add(pc);
        const uint32_t incl_e = EMIT ? wave_incl_add(ne) : 0u;
        if (lane == 63) { agg[0][wave] = incl; agg[1][wave] = incl_e; }
        if (EMIT && tid == 0) agg[3][0] = w & 1u;           // runs alternate: the bit of run k is this one ^ (k & 1)
        lds_barrier();                                    // (2)
        uint32_t ones, n_runs = 0, first_bit = 0;
        {
            const uint32_t va = lane < 16 ? agg[0][lane] : 0u;
            const uint32_t sa = wave_incl_add(va);
            ones = lane_value(sa, 15);
            dir[tid].y = (wave ? lane_value(sa, wave - 1) : 0u) + incl - pc;
            if (EMIT) {                                     // the last position of every run, in order, one list for the row
                const uint32_t ve = lane < 16 ? agg[1][lane] : 0u;
                const uint32_t se = wave_incl_add(ve);
                n_runs = lane_value(se, 15);
                uint32_t k = (wave ? lane_value(se, wave - 1) : 0u) + incl_e - ne;
                if (!ENC_ABLATE(a, 4)) for (uint32_t x = ends; x; x &= x - 1u) run_end[k++] = (uint16_t)(tid * 32 + __builtin_ctz(x));
                first_bit = agg[3][0];
            }
        }
        lds_barrier();                                    // (3) `before` and the run list are visible
        // the runs dealt evenly to the threads (a row of C2 has ~165: three waves emit, thirteen only take part in the scan)
        const uint32_t rpt = (n_runs + kThreads - 1) >> 10, k0 = (uint32_t)tid * rpt, k1 = k0 + rpt < n_runs ? k0 + rpt : n_runs;
        uint32_t nb = 0;
        if (EMIT && !ENC_ABLATE(a, 2))
            for (uint32_t k = k0; k < k1; ++k) nb += run_bytes((uint32_t)run_end[k] - (k ? (uint32_t)run_end[k - 1] : 0xffffffffu));
        uint32_t bbase = 0, total = 0, incl2 = 0;
        if (EMIT) {
            incl2 = wave_incl_add(nb);
            if (lane == 63) agg[2][wave] = incl2;
            lds_barrier();                                // (4)
            const uint32_t vb = lane < 16 ? agg[2][lane] : 0u;
            const uint32_t sb = wave_incl_add(vb);
            total = lane_value(sb, 15);
            bbase = wave ? lane_value(sb, wave - 1) : 0u;
        }
        if (EMIT && off + (int64_t)total > a.cap) { if (tid == 0) *a.status = 1; break; }  // uniform
        if (EMIT && !ENC_ABLATE(a, 3)) {
            uint8_t *dst = out + off + bbase + incl2 - nb;
            for (uint32_t k = k0; k < k1; ++k)
                dst += put_run(dst, (uint32_t)run_end[k] - (k ? (uint32_t)run_end[k - 1] : 0xffffffffu), first_bit ^ (k & 1u));
        }
        if (EMIT && tid == 0) a.row_len[(size_t)plane * a.n_rows + r] = (int32_t)total;
        off += total;
        // the next row's codes, fetched behind the stores of this row's bytes (the memory counter retires in order: a
        // fetch issued before them would make the next scatter wait for the stores as well) and under the rank step
        if (r + 1 < r_end) {
            const uint32_t *src = reinterpret_cast<const uint32_t*>(a.codes + (size_t)(r + 1) * stride + col0);
#pragma unroll
            for (int q = 0; q < CPT / 4; ++q) nxt[q] = src[q];
        }
        // ---- the stable partition, on ranks (ref pbwt.c:57-66 moves S instead)
        // t = word << (q & 31): bit r & 31 in the sign, the bits above it gone; oi = ones up to and including r;
        // bit ? r = n0 + oi - 1 : r = r - oi, i.e. q = bit ? -n0 - oi : q + oi
        const int32_t neg_n0 = (int32_t)ones - m;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const uint2 e = *reinterpret_cast<const uint2*>(dir_m8 - 8 * (Q[i] >> 5));
            const uint32_t t = e.x << (Q[i] & 31);
            const int32_t oi = (int32_t)(e.y + (uint32_t)__popc(t));
            Q[i] = (int32_t)t < 0 ? neg_n0 - oi : Q[i] + oi;
        }
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        if (col0 + i < m) {
            a.rank_out[up * m + col0 + i] = ~Q[i];
            if (a.perm_out) a.perm_out[up * m + ~Q[i]] = col0 + i;
        }
    if (EMIT && tid == 0) a.out_len[up] = off;
}

// ---- wide cohorts (more than 32768 columns): the ranks do not fit the registers of one workgroup, so they live in
// memory (800 KB per plane at 100,000 samples: L2) and every thread owns WPT words of the directory.  Same phases,
// same barriers.
template <int WPT, bool EMIT>
__global__ __launch_bounds__(kThreads) void encode_wide_kernel(EncodeArgs a)
{
    extern __shared__ uint2 wide_dir[];         // [2][kThreads * WPT + 1]
    __shared__ uint32_t agg[3][16];
    constexpr int NWP = kThreads * WPT + 1;
    const int tid = threadIdx.x, lane = tid & 63, plane = blockIdx.x, unit = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, nw = (m + 31) >> 5;
    const size_t up = (size_t)unit * a.g + plane;
    const int64_t r_beg = (int64_t)unit * a.unit_rows;
    const int64_t r_end = r_beg + a.unit_rows < a.n_rows ? r_beg + a.unit_rows : a.n_rows;
    uint8_t *out = a.out + up * a.cap;
    int32_t *Qg = a.rank_out + up * m;          // complemented ranks; column c belongs to thread c % 1024 throughout
    for (int c = tid; c < m; c += kThreads) Qg[c] = ~(a.rank_in ? a.rank_in[up * m + c] : c);
    for (int i = tid; i < 2 * NWP; i += kThreads) wide_dir[i] = make_uint2(0u, 0u);
    lds_barrier();
    const int w0 = tid * WPT;
    int64_t off = 0;
    int snap_i = EMIT ? a.snap_base[unit] : 0;
    if (r_beg < r_end) {                                    // the first row's bits; later rows are scattered by the step before them
        const uint8_t *src = a.codes + (size_t)r_beg * a.stride;
        char *dir_m8 = reinterpret_cast<char*>(wide_dir + (r_beg & 1) * NWP) - 8;
        for (int c = tid; c < m; c += kThreads) {
            const int32_t q = Qg[c];
            if ((src[c] >> plane) & 1) atomicOr(reinterpret_cast<uint32_t*>(dir_m8 - 8 * (q >> 5)), 0x80000000u >> (q & 31));
        }
    }
    for (int64_t r = r_beg; r < r_end; ++r) {
        uint2 *dir = wide_dir + (r & 1) * NWP, *other = wide_dir + ((r & 1) ^ 1) * NWP;
        char *dir_m8 = reinterpret_cast<char*>(dir) - 8;
        if (EMIT && ((a.row0 + r) & a.mask) == 0) {
            int32_t *S = a.snap + ((size_t)plane * a.n_snap + snap_i) * m;
            for (int c = tid; c < m; c += kThreads) S[~Qg[c]] = c;
            ++snap_i;
        }
        lds_barrier();                                      // (1) the bit-vector of this row is complete
        uint32_t w[WPT], ends[WPT], pc = 0, le = 0;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const int wi = w0 + k;
            other[wi].x = 0u;
            w[k] = dir[wi].x;
            const uint32_t wn = dir[wi + 1].x;
            const uint32_t valid = wi < nw ? ((wi == nw - 1 && (m & 31)) ? (1u << (m & 31)) - 1u : 0xffffffffu) : 0u;
            const uint32_t last_bit = wi == nw - 1 ? 1u << ((m - 1) & 31) : 0u;
            ends[k] = EMIT ? ((((w[k] ^ (w[k] >> 1 | wn << 31)) & valid) & ~last_bit) | last_bit) : 0u;
            pc += (uint32_t)__popc(w[k]);
            if (ends[k]) le = (uint32_t)(wi * 32 + 32 - __builtin_clz(ends[k]));
        }
        const uint32_t incl = wave_incl_add(pc);
        const uint32_t lmax = EMIT ? wave_incl_max(le) : 0u;
        if (lane == 63) { agg[0][wave] = incl; agg[1][wave] = lmax; }
        lds_barrier();                                      // (2)
        uint32_t ones, start = 0;
        {
            const uint32_t va = lane < 16 ? agg[0][lane] : 0u;
            const uint32_t sa = wave_incl_add(va);
            ones = lane_value(sa, 15);
            uint32_t base = (wave ? lane_value(sa, wave - 1) : 0u) + incl - pc;
#pragma unroll
            for (int k = 0; k < WPT; ++k) { dir[w0 + k].y = base; base += (uint32_t)__popc(w[k]); }
            if (EMIT) {
                const uint32_t vm = lane < 16 ? agg[1][lane] : 0u;
                const uint32_t sx = wave_incl_max(vm);
                start = umax(wave ? lane_value(sx, wave - 1) : 0u, wave_shr1(lmax));
            }
        }
        uint32_t nb = 0;
        if (EMIT) {
            uint32_t st = start;
#pragma unroll
            for (int k = 0; k < WPT; ++k)
                for (uint32_t x = ends[k]; x;) {
                    const uint32_t e = (uint32_t)((w0 + k) * 32 + __builtin_ctz(x) + 1);
                    x &= x - 1u;
                    nb += run_bytes(e - st);
                    st = e;
                }
        }
        const uint32_t incl2 = EMIT ? wave_incl_add(nb) : 0u;
        if (EMIT && lane == 63) agg[2][wave] = incl2;
        lds_barrier();                                      // (3)
        uint32_t total = 0;
        if (EMIT) {
            const uint32_t vb = lane < 16 ? agg[2][lane] : 0u;
            const uint32_t sb = wave_incl_add(vb);
            total = lane_value(sb, 15);
            const uint32_t bbase = wave ? lane_value(sb, wave - 1) : 0u;
            if (off + (int64_t)total > a.cap) { if (tid == 0) *a.status = 1; break; }      // uniform
            uint8_t *dst = out + off + bbase + incl2 - nb;
            uint32_t st = start;
#pragma unroll
            for (int k = 0; k < WPT; ++k)
                for (uint32_t x = ends[k]; x;) {
                    const uint32_t i = (uint32_t)__builtin_ctz(x);
                    const uint32_t e = (uint32_t)((w0 + k) * 32) + i + 1u;
                    x &= x - 1u;
                    dst += put_run(dst, e - st, (w[k] >> i) & 1u);
                    st = e;
                }
            if (tid == 0) a.row_len[(size_t)plane * a.n_rows + r] = (int32_t)total;
            off += total;
        }
        // the step of this row fused with the scatter of the next one (one pass over the ranks per row), eight
        // columns at a time so that their loads are in flight together
        const int32_t neg_n0 = (int32_t)ones - m;
        const bool more = r + 1 < r_end;
        const uint8_t *nsrc = a.codes + (size_t)(more ? r + 1 : r) * a.stride;
        char *other_m8 = reinterpret_cast<char*>(other) - 8;
        for (int c0 = tid; c0 < m; c0 += 8 * kThreads) {
            int32_t q[8];
            uint32_t nb8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + j * kThreads;
                q[j] = c < m ? Qg[c] : -1;
                nb8[j] = (c < m && more) ? nsrc[c] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + j * kThreads;
                const uint2 e = *reinterpret_cast<const uint2*>(dir_m8 - 8 * (q[j] >> 5));
                const uint32_t t = e.x << (q[j] & 31);
                const int32_t oi = (int32_t)(e.y + (uint32_t)__popc(t));
                const int32_t qn = (int32_t)t < 0 ? neg_n0 - oi : q[j] + oi;
                if (c < m) {
                    Qg[c] = qn;
                    if ((nb8[j] >> plane) & 1u) atomicOr(reinterpret_cast<uint32_t*>(other_m8 - 8 * (qn >> 5)), 0x80000000u >> (qn & 31));
                }
            }
        }
    }
    for (int c = tid; c < m; c += kThreads) {
        const int32_t rk = ~Qg[c];
        Qg[c] = rk;
        if (a.perm_out) a.perm_out[up * m + rk] = c;
    }
    if (EMIT && tid == 0) a.out_len[up] = off;
}

// ---- cohorts beyond 262,144 columns: the two row directories (16 bytes per 32 columns) no longer fit the LDS either and live
// in memory, a region per (unit, plane) workgroup (a.gdir: 0.5 MB per plane at 1,000,000 haplotypes -- L2).  encode_wide_kernel's
// phases word for word; what changes is how the workgroup's waves see each other's directory words: the scatter is an atomic
// OR performed in the L2, so every access to the directory is an agent-scope atomic (loads and stores that go past the CU's
// vector cache, which an L2 atomic does not update), and the barriers also wait for memory (vmcnt), not only for the LDS.
// The reference writer takes any int32 m (pbwt.c:199-219); this path is what lifts the encoder's limit -- speed is not its point.
__device__ __forceinline__ uint32_t gload(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gstore(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mem_barrier() { __threadfence(); __syncthreads(); }

template <int WPT, bool EMIT>
__global__ __launch_bounds__(kThreads) void encode_huge_kernel(EncodeArgs a)
{
    __shared__ uint32_t agg[3][16];
    constexpr int NWP = kThreads * WPT + 1;
    const int tid = threadIdx.x, lane = tid & 63, plane = blockIdx.x, unit = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, nw = (m + 31) >> 5;
    const size_t up = (size_t)unit * a.g + plane;
    const int64_t r_beg = (int64_t)unit * a.unit_rows;
    const int64_t r_end = r_beg + a.unit_rows < a.n_rows ? r_beg + a.unit_rows : a.n_rows;
    uint8_t *out = a.out + up * a.cap;
    uint32_t *gd = reinterpret_cast<uint32_t*>(a.gdir + up * 2 * (size_t)NWP);    // entry i of directory d: words 2 (d NWP + i) (.x bits), + 1 (.y ones before)
    int32_t *Qg = a.rank_out + up * m;          // complemented ranks; column c belongs to thread c % 1024 throughout
    for (int c = tid; c < m; c += kThreads) Qg[c] = ~(a.rank_in ? a.rank_in[up * m + c] : c);
    for (int i = tid; i < 4 * NWP; i += kThreads) gstore(gd + i, 0u);
    mem_barrier();
    const int w0 = tid * WPT;
    int64_t off = 0;
    int snap_i = EMIT ? a.snap_base[unit] : 0;
    // (entry index of rank r = NWP_base + (r >> 5); with the complement q = ~r: r >> 5 = -(q >> 5) - 1)
    auto entry = [&](int d, int32_t q) -> uint32_t* { return gd + 2 * ((size_t)d * NWP + (size_t)(-(q >> 5) - 1)); };
    if (r_beg < r_end) {                                    // the first row's bits; later rows are scattered by the step before them
        const uint8_t *src = a.codes + (size_t)r_beg * a.stride;
        for (int c = tid; c < m; c += kThreads) {
            const int32_t q = Qg[c];
            if ((src[c] >> plane) & 1) atomicOr(entry((int)(r_beg & 1), q), 0x80000000u >> (q & 31));
        }
    }
    for (int64_t r = r_beg; r < r_end; ++r) {
        const int cur = (int)(r & 1), oth = cur ^ 1;
        uint32_t *dir = gd + 2 * (size_t)cur * NWP, *other = gd + 2 * (size_t)oth * NWP;
        if (EMIT && ((a.row0 + r) & a.mask) == 0) {
            int32_t *S = a.snap + ((size_t)plane * a.n_snap + snap_i) * m;
            for (int c = tid; c < m; c += kThreads) S[~Qg[c]] = c;
            ++snap_i;
        }
        mem_barrier();                                      // (1) the bit-vector of this row is complete
        uint32_t pc = 0, le = 0;
        // (the words are not kept in registers -- WPT is up to 64 --: they are read again where the runs are counted and written)
        for (int k = 0; k < WPT; ++k) {
            const int wi = w0 + k;
            gstore(other + 2 * wi, 0u);
            const uint32_t w = gload(dir + 2 * wi);
            pc += (uint32_t)__popc(w);
            if (EMIT) {
                const uint32_t wn = gload(dir + 2 * (wi + 1));
                const uint32_t valid = wi < nw ? ((wi == nw - 1 && (m & 31)) ? (1u << (m & 31)) - 1u : 0xffffffffu) : 0u;
                const uint32_t last_bit = wi == nw - 1 ? 1u << ((m - 1) & 31) : 0u;
                const uint32_t ends = (((w ^ (w >> 1 | wn << 31)) & valid) & ~last_bit) | last_bit;
                if (ends) le = (uint32_t)(wi * 32 + 32 - __builtin_clz(ends));
            }
        }
        const uint32_t incl = wave_incl_add(pc);
        const uint32_t lmax = EMIT ? wave_incl_max(le) : 0u;
        if (lane == 63) { agg[0][wave] = incl; agg[1][wave] = lmax; }
        mem_barrier();                                      // (2)
        uint32_t ones, start = 0;
        {
            const uint32_t va = lane < 16 ? agg[0][lane] : 0u;
            const uint32_t sa = wave_incl_add(va);
            ones = lane_value(sa, 15);
            uint32_t base = (wave ? lane_value(sa, wave - 1) : 0u) + incl - pc;
            for (int k = 0; k < WPT; ++k) { gstore(dir + 2 * (w0 + k) + 1, base); base += (uint32_t)__popc(gload(dir + 2 * (w0 + k))); }
            if (EMIT) {
                const uint32_t vm = lane < 16 ? agg[1][lane] : 0u;
                const uint32_t sx = wave_incl_max(vm);
                start = umax(wave ? lane_value(sx, wave - 1) : 0u, wave_shr1(lmax));
            }
        }
        auto ends_of = [&](int wi, uint32_t &w) -> uint32_t {   // where runs end inside word wi (the row's last position always ends one)
            w = gload(dir + 2 * wi);
            const uint32_t wn = gload(dir + 2 * (wi + 1));
            const uint32_t valid = wi < nw ? ((wi == nw - 1 && (m & 31)) ? (1u << (m & 31)) - 1u : 0xffffffffu) : 0u;
            const uint32_t last_bit = wi == nw - 1 ? 1u << ((m - 1) & 31) : 0u;
            return (((w ^ (w >> 1 | wn << 31)) & valid) & ~last_bit) | last_bit;
        };
        uint32_t nb = 0;
        if (EMIT) {
            uint32_t st = start;
            for (int k = 0; k < WPT; ++k) {
                uint32_t w;
                for (uint32_t x = ends_of(w0 + k, w); x;) {
                    const uint32_t e = (uint32_t)((w0 + k) * 32 + __builtin_ctz(x) + 1);
                    x &= x - 1u;
                    nb += run_bytes(e - st);
                    st = e;
                }
            }
        }
        const uint32_t incl2 = EMIT ? wave_incl_add(nb) : 0u;
        if (EMIT && lane == 63) agg[2][wave] = incl2;
        mem_barrier();                                      // (3) (also: every .y of this row is stored before the step reads it)
        uint32_t total = 0;
        if (EMIT) {
            const uint32_t vb = lane < 16 ? agg[2][lane] : 0u;
            const uint32_t sb = wave_incl_add(vb);
            total = lane_value(sb, 15);
            const uint32_t bbase = wave ? lane_value(sb, wave - 1) : 0u;
            if (off + (int64_t)total > a.cap) { if (tid == 0) *a.status = 1; break; }      // uniform
            uint8_t *dst = out + off + bbase + incl2 - nb;
            uint32_t st = start;
            for (int k = 0; k < WPT; ++k) {
                uint32_t w;
                for (uint32_t x = ends_of(w0 + k, w); x;) {
                    const uint32_t i = (uint32_t)__builtin_ctz(x);
                    const uint32_t e = (uint32_t)((w0 + k) * 32) + i + 1u;
                    x &= x - 1u;
                    dst += put_run(dst, e - st, (w >> i) & 1u);
                    st = e;
                }
            }
            if (tid == 0) a.row_len[(size_t)plane * a.n_rows + r] = (int32_t)total;
            off += total;
        }
        // the step of this row fused with the scatter of the next one, as in encode_wide_kernel
        const int32_t neg_n0 = (int32_t)ones - m;
        const bool more = r + 1 < r_end;
        const uint8_t *nsrc = a.codes + (size_t)(more ? r + 1 : r) * a.stride;
        for (int c0 = tid; c0 < m; c0 += 8 * kThreads) {
            int32_t q[8];
            uint32_t nb8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + j * kThreads;
                q[j] = c < m ? Qg[c] : -1;
                nb8[j] = (c < m && more) ? nsrc[c] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + j * kThreads;
                const uint32_t *e = entry(cur, q[j]);
                const uint32_t ex = gload(e), ey = gload(e + 1);
                const uint32_t t = ex << (q[j] & 31);
                const int32_t oi = (int32_t)(ey + (uint32_t)__popc(t));
                const int32_t qn = (int32_t)t < 0 ? neg_n0 - oi : q[j] + oi;
                if (c < m) {
                    Qg[c] = qn;
                    if ((nb8[j] >> plane) & 1u) atomicOr(entry(oth, qn), 0x80000000u >> (qn & 31));
                }
            }
        }
    }
    for (int c = tid; c < m; c += kThreads) {
        const int32_t rk = ~Qg[c];
        Qg[c] = rk;
        if (a.perm_out) a.perm_out[up * m + rk] = c;
    }
    if (EMIT && tid == 0) a.out_len[up] = off;
}

// ---- phase B helpers ------------------------------------------------------------------------------------------
// The columns of a unit as bits: colbits[(unit * wpu + j) * 2 + plane][col] = bits of the column in rows 32j..32j+31 of the
// unit (row-block major, so that both this kernel's loads and stores are coalesced).  Planes 0 and 1 only (the parallel
// path is used for g <= 2).
__global__ __launch_bounds__(256) void column_bits_kernel(const uint8_t *codes, int stride, int64_t n_rows, int unit_rows, int wpu,
                                                          int m, uint32_t *colbits)
{
    const int col = 4 * (blockIdx.x * 256 + threadIdx.x), blk = blockIdx.y;   // four columns per thread; blk = unit * wpu + j
    if (col >= m) return;
    const int unit = blk / wpu, j = blk - unit * wpu;
    const int64_t u_end = (int64_t)(unit + 1) * unit_rows < n_rows ? (int64_t)(unit + 1) * unit_rows : n_rows;
    const int64_t r0 = (int64_t)unit * unit_rows + 32 * j;
    uint32_t w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0};
#pragma unroll 8
    for (int i = 0; i < 32; ++i)
        if (r0 + i < u_end) {
            const uint32_t c4 = *reinterpret_cast<const uint32_t*>(codes + (size_t)(r0 + i) * stride + col);   // stride % 4 == 0
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                w0[k] |= ((c4 >> (8 * k)) & 1u) << i;
                w1[k] |= ((c4 >> (8 * k + 1)) & 1u) << i;
            }
        }
    uint32_t *d0 = colbits + ((size_t)blk * 2 + 0) * m + col, *d1 = colbits + ((size_t)blk * 2 + 1) * m + col;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (col + k < m) { d0[k] = w0[k]; d1[k] = w1[k]; }
}

// flag[p] = 1 if the column at position p of a unit's own order differs from its left neighbour in any row of the unit
__global__ __launch_bounds__(256) void class_flags_kernel(const uint32_t *colbits, int wpu, int m, int g, const int32_t *perm, uint8_t *flag)
{
    const int p = blockIdx.x * 256 + threadIdx.x, plane = blockIdx.y, unit = blockIdx.z;
    if (p >= m) return;
    const size_t up = (size_t)unit * g + plane;
    uint32_t diff = 1u;
    if (p > 0) {
        const int ca = perm[up * m + p], cb = perm[up * m + p - 1];
        diff = 0u;
        for (int j = 0; j < wpu && !diff; ++j) {
            const uint32_t *w = colbits + (((size_t)unit * wpu + j) * 2 + plane) * m;
            diff = w[ca] ^ w[cb];
        }
    }
    flag[up * m + p] = diff ? 1 : 0;
}

// cid[p] = number of class starts in positions 1..p (one workgroup per unit and plane)
__global__ __launch_bounds__(kThreads) void class_ids_kernel(int m, const uint8_t *flag, int32_t *cid)
{
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t up = blockIdx.x;
    const int per = (m + kThreads - 1) / kThreads, p0 = tid * per;
    uint32_t mine = 0;
    for (int i = 0; i < per; ++i) if (p0 + i < m && p0 + i > 0) mine += flag[up * m + p0 + i];
    const uint32_t incl = wave_incl_add(mine);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = incl - mine;
    for (int k = 0; k < wave; ++k) base += wsum[k];
    for (int i = 0; i < per; ++i)
        if (p0 + i < m) { if (p0 + i > 0) base += flag[up * m + p0 + i]; cid[up * m + p0 + i] = (int32_t)base; }
}

// Phase B walks the columns IN THE ORDER BEFORE THE UNIT (perm) and sorts them stably by the class they have in the
// unit's own order: key = plane | class.  The sorted columns are the order after the unit; ties keep the order before.
__global__ __launch_bounds__(256) void class_keys_kernel(int m, int g, int bits, const int32_t *cid_u, const int32_t *local_u,
                                                         const int32_t *perm_u, uint32_t *key)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= g * m) return;
    const int plane = i / m;
    const int col = perm_u[i];
    key[i] = (uint32_t)plane << bits | (uint32_t)cid_u[(size_t)plane * m + local_u[(size_t)plane * m + col]];
}

// ---- phase B on one workgroup per plane (m <= 32768) --------------------------------------------------------------
// ccls[u][plane][col] = class of the column in unit u's own order, for every unit at once (16 bits: a class number is
// below m <= 32768)
__global__ __launch_bounds__(256) void column_class_kernel(int m, int64_t n_total, const int32_t *cid, const int32_t *local, uint16_t *ccls)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    const int64_t base = i / m * m;
    ccls[i] = (uint16_t)cid[base + local[i]];
}

// The chain of phase B in ONE launch: a workgroup per plane keeps the order (position -> column) in LDS and, unit after
// unit, sorts it stably by the columns' class in that unit -- an LSD radix sort over the 15 bits of a class number,
// DIGIT bits a pass, the elements {class << 16 | column} blocked EPT to a thread: digits counted in registers (packed
// bytes), one scan over the [bucket][thread] counts, scatter in place.  Every unit's order goes out as it is found
// (perms[u + 1]).  Positions m .. 1024 EPT - 1 hold sentinels that sort last in every pass.  (128 units at m = 20,000:
// 128 x 7 launches of a library sort, 5.1 ms, become one launch of 4.5 ms -- 35 us a unit, four passes of five barriers
// each on one CU; BGTH_ENC_LIBSORT=1 takes the library path for comparison.)
template <int EPT, int DIGIT>
__global__ __launch_bounds__(kThreads) void order_chain_kernel(int m, int g, int n_units, const uint16_t *ccls, int32_t *perms)
{
    constexpr int NB = 1 << DIGIT, PASSES = (15 + DIGIT - 1) / DIGIT;
    extern __shared__ uint32_t chain_lds[];                // [1024 EPT] elements, then [NB][1024] 16-bit counts
    __shared__ uint32_t agg[16];
    uint32_t *data = chain_lds;
    uint16_t *cnt = reinterpret_cast<uint16_t*>(chain_lds + kThreads * EPT);
    const int tid = threadIdx.x, lane = tid & 63, plane = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t gm = (size_t)g * m;
    uint32_t el[EPT];
    {
        const int32_t *src = perms + (size_t)plane * m;
#pragma unroll
        for (int i = 0; i < EPT; ++i) { const int p = tid * EPT + i; el[i] = p < m ? (uint32_t)src[p] : 0xffffffffu; }
    }
    for (int u = 0; u < n_units; ++u) {
        const uint16_t *cls = ccls + (size_t)u * gm + (size_t)plane * m;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const bool real = el[i] != 0xffffffffu;
            const uint32_t col = real ? el[i] & 0xffffu : 0u, c = cls[col];
            el[i] = real ? c << 16 | col : 0xffffffffu;
        }
#pragma unroll 1
        for (int pass = 0; pass < PASSES; ++pass) {
            const int sh = 16 + pass * DIGIT;
            // this thread's count per digit, in its own column of the [bucket][thread] table; an element's number among its
            // thread's elements of the same digit is the count it finds (a byte each in rk[])
            uint32_t rk[EPT / 4];
#pragma unroll
            for (int i = 0; i < EPT / 4; ++i) rk[i] = 0u;
#pragma unroll
            for (int b = 0; b < NB; ++b) cnt[b * kThreads + tid] = 0;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const uint32_t d = (el[i] >> sh) & (NB - 1);
                uint16_t *c = cnt + d * kThreads + tid;
                const uint32_t r = *c;
                *c = (uint16_t)(r + 1u);
                rk[i >> 2] |= r << (8 * (i & 3));
            }
            lds_barrier();
            // exclusive scan over the counts in [bucket][thread] order: this thread's NB consecutive entries
            uint32_t *cw = reinterpret_cast<uint32_t*>(cnt + tid * NB);
            uint32_t total = 0;
#pragma unroll
            for (int b = 0; b < NB; b += 2) { const uint32_t w = cw[b >> 1]; total += (w & 0xffffu) + (w >> 16); }
            const uint32_t incl = wave_incl_add(total);
            if (lane == 63) agg[wave] = incl;
            lds_barrier();
            {
                const uint32_t va = lane < 16 ? agg[lane] : 0u;
                const uint32_t sa = wave_incl_add(va);
                uint32_t base = (wave ? lane_value(sa, wave - 1) : 0u) + incl - total;
#pragma unroll
                for (int b = 0; b < NB; b += 2) {                // (read again rather than kept across the barrier: registers)
                    const uint32_t w = cw[b >> 1], b0 = base, b1 = base + (w & 0xffffu);
                    base = b1 + (w >> 16);
                    cw[b >> 1] = b0 | b1 << 16;
                }
            }
            lds_barrier();
            int sh2 = sh;
            asm volatile("" : "+s"(sh2));                       // (the digits are computed again, not kept in registers since the count)
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const uint32_t d = (el[i] >> sh2) & (NB - 1);
                data[(uint32_t)cnt[d * kThreads + tid] + ((rk[i >> 2] >> (8 * (i & 3))) & 255u)] = el[i];
            }
            lds_barrier();
#pragma unroll
            for (int i = 0; i < EPT; ++i) el[i] = data[tid * EPT + i];
            lds_barrier();                                  // (the next pass's scatter, or the next unit's, must not overtake these reads)
        }
        int32_t *dst = perms + (size_t)(u + 1) * gm + (size_t)plane * m;
        for (int p = tid; p < m; p += kThreads) dst[p] = (int32_t)(data[p] & 0xffffu);
    }
}

// position -> column  <->  column -> position, for n consecutive (unit, plane) arrays of m entries
__global__ __launch_bounds__(256) void invert_kernel(int m, int64_t n_total, const int32_t *src, int32_t *dst)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    const int64_t base = i / m * m;
    dst[base + src[i]] = (int32_t)(i - base);
}

// rows of 2-bit codes, four columns to a byte (column c in bits 2 (c & 3) of byte c >> 2: the layout the reader's
// genotype rows have), spread to one byte per column with zero padding
__global__ __launch_bounds__(256) void unpack_codes_kernel(const uint8_t *packed, int64_t n_rows, int m, int stride, uint8_t *codes)
{
    const int pb = (m + 3) >> 2;
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= pb) return;
    const int left = m - 4 * b;                              // columns of this byte that exist
    const uint32_t keep = left < 4 ? (1u << (8 * left)) - 1u : 0xffffffffu;
    for (int64_t r = blockIdx.y; r < n_rows; r += gridDim.y) {
        const uint32_t x = packed[(size_t)r * pb + b];
        const uint32_t w = (x & 3u) | ((x >> 2) & 3u) << 8 | ((x >> 4) & 3u) << 16 | ((x >> 6) & 3u) << 24;
        *reinterpret_cast<uint32_t*>(codes + (size_t)r * stride + 4 * b) = w & keep;
    }
}

// the used part of every (unit, plane) output region, one after the other: one download instead of hundreds
__global__ __launch_bounds__(kThreads) void compact_kernel(const uint8_t *out, int64_t cap, const int64_t *base, uint8_t *packed)
{
    const int64_t b0 = base[blockIdx.x], n = base[blockIdx.x + 1] - b0;
    const uint8_t *src = out + (size_t)blockIdx.x * cap;
    for (int64_t i = threadIdx.x; i < n; i += kThreads) packed[b0 + i] = src[i];
}

thread_local char g_enc_err[256] = "";
void enc_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_enc_err, sizeof(g_enc_err), fmt, ap);
    va_end(ap);
}

#define ENC_TRY(expr, onfail)                                                                     \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) { enc_err("[E::%s] %s: %s", __func__, #expr, hipGetErrorString(e_)); onfail; } \
    } while (0)

}  // namespace

struct bgth_encoder_s {
    int32_t m = 0, g = 0, shift = 0, device = 0;
    int64_t n = 0;                                   // rows written
    bool finished = false;
    uint64_t taken = 0;                              // bytes of the file already handed out by bgth_encoder_take
    bool broken = false;                             // a pass failed half way: the order on the device is ahead of the image
    int32_t cpt = 0, stride = 0;                     // device rows of codes are stride = cpt * 1024 bytes apart
    int32_t unit_rows = 4096;                        // most rows per parallel unit (BGTH_ENC_UNIT_SHIFT fixes the size)
    bool unit_fixed = false;
    int64_t batch_rows = 0;                          // most rows per device pass
    int64_t rows_cap = 0;                            // what the buffers below hold
    int32_t units_cap = 0, snap_cap = 0;
    hipStream_t stream = nullptr;
    uint8_t *d_codes = nullptr, *d_out = nullptr, *d_flag = nullptr;
    uint8_t *d_packed_in = nullptr;                  // bgth_encoder_write_packed: the rows as they came
    size_t packed_in_cap = 0;
    uint8_t *d_packed = nullptr;                     // the run-length bytes of a pass, regions closed up
    int64_t *d_base = nullptr;
    size_t packed_cap = 0, base_cap = 0;
    uint32_t *d_colbits = nullptr;                   // [units * ceil(unit_rows / 32)][2][m] the columns of every unit as bits
    size_t colbits_cap = 0;
    int32_t *d_state = nullptr;                      // [g][m] order after the last row written
    int32_t *d_true = nullptr, *d_local = nullptr, *d_perm = nullptr, *d_cid = nullptr;   // [units(+1)][g][m]
    int32_t *d_row_len = nullptr, *d_snap = nullptr, *d_snap_base = nullptr, *d_status = nullptr;
    int32_t *d_perms = nullptr;                      // [units + 1][g][m] true order before every unit as position -> column
    uint32_t *d_key[2] = {nullptr, nullptr};
    uint16_t *d_ccls = nullptr;                      // [units][g][m] class of every column in every unit's own order (narrow cohorts)
    int32_t wpt = 0;                                 // > 0: the wide kernel with this many directory words per thread (> 8: encode_huge_kernel)
    uint2 *d_gdir = nullptr;                         // encode_huge_kernel's directories: [units][g][2][1024 wpt + 1]
    size_t gdir_cap = 0;
    void *d_temp = nullptr;
    size_t temp_bytes = 0;
    int64_t *d_out_len = nullptr;
    std::vector<uint8_t> image;
    std::vector<uint64_t> idx;
    uint8_t *h_out = nullptr;                        // pinned: the packed run-length bytes of a pass
    size_t h_out_cap = 0;
    std::vector<int32_t> h_row_len, h_snap, h_snap_base;
    std::vector<int64_t> h_out_len;
    double kernel_ms = 0.0;
};

extern "C" const char *bgth_encoder_last_error(void) { return g_enc_err; }

static void free_batch_buffers(bgth_encoder_t *e)
{
    hipFree(e->d_codes); hipFree(e->d_out); hipFree(e->d_flag); hipFree(e->d_true); hipFree(e->d_perms); hipFree(e->d_local); hipFree(e->d_perm);
    hipFree(e->d_cid); hipFree(e->d_row_len); hipFree(e->d_snap); hipFree(e->d_snap_base); hipFree(e->d_out_len); hipFree(e->d_ccls);
    e->d_codes = e->d_out = e->d_flag = nullptr;
    e->d_ccls = nullptr;
    e->d_perms = nullptr;
    e->d_true = e->d_local = e->d_perm = e->d_cid = e->d_row_len = e->d_snap = e->d_snap_base = nullptr;
    e->d_out_len = nullptr;
    e->rows_cap = 0; e->units_cap = 0; e->snap_cap = 0;
}

extern "C" void bgth_encoder_close(bgth_encoder_t *e)
{
    if (!e) return;
    hipSetDevice(e->device);
    free_batch_buffers(e);
    hipFree(e->d_gdir);
    hipFree(e->d_state); hipFree(e->d_status); hipFree(e->d_temp); hipFree(e->d_colbits); hipFree(e->d_packed); hipFree(e->d_packed_in); hipFree(e->d_base); hipHostFree(e->h_out);
    for (int i = 0; i < 2; ++i) hipFree(e->d_key[i]);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

extern "C" bgth_encoder_t *bgth_encoder_open(int32_t m, int32_t g, int32_t shift, int device)
{
    if (m < 1 || m > kMaxHugeM) { enc_err("[E::%s] %d columns: this encoder holds 1..%d", __func__, m, kMaxHugeM); return nullptr; }
    if (g < 1 || g > 8 || shift < 0 || shift > 30) { enc_err("[E::%s] bad plane count or checkpoint shift", __func__); return nullptr; }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
        enc_err("[E::%s] no usable HIP device %d (the encoder has no CPU path)", __func__, device);
        return nullptr;
    }
    bgth_encoder_t *e = new bgth_encoder_t;
    e->m = m; e->g = g; e->shift = shift; e->device = device;
    e->cpt = m <= 4 * kThreads ? 4 : m <= 8 * kThreads ? 8 : m <= 20 * kThreads ? 20 : 32;
    e->stride = e->cpt * kThreads;
    if (m > kMaxM) {
        const int nw = (m + 31) / 32;
        e->wpt = nw <= 2 * kThreads ? 2 : nw <= 4 * kThreads ? 4 : nw <= 8 * kThreads ? 8 : nw <= 16 * kThreads ? 16 : nw <= 32 * kThreads ? 32 : 64;
        e->stride = (m + 3) & ~3;
    }
    if (e->wpt) e->unit_rows = 1024;                 // wide rows are slow and large: smaller units, more of them at once
    if (e->wpt > 8) e->unit_rows = 512;              // (beyond 262,144 columns: 0.5 GB of codes per unit at a million)
    if (const char *u = getenv("BGTH_ENC_UNIT_SHIFT")) {
        const int us = atoi(u);
        if (us >= 1 && us <= 20) { e->unit_rows = 1 << us; e->unit_fixed = true; }
    }
    // one device pass: up to 12 GB of codes = 150 units of 4096 rows at 10,000 samples, two workgroups each (the
    // run-length output is sized for the worst case, one byte per bit: 36 GB of the 288 in all)
    e->batch_rows = ((int64_t)12 << 30) / e->stride / e->unit_rows * e->unit_rows;
    if (e->batch_rows < e->unit_rows) e->batch_rows = e->unit_rows;
    ENC_TRY(hipSetDevice(device), { delete e; return nullptr; });
    ENC_TRY(hipStreamCreate(&e->stream), { delete e; return nullptr; });
    ENC_TRY(hipMalloc(&e->d_state, (size_t)g * m * 4), { bgth_encoder_close(e); return nullptr; });
    ENC_TRY(hipMalloc(&e->d_status, 4), { bgth_encoder_close(e); return nullptr; });
    for (int i = 0; i < 2; ++i) {
        ENC_TRY(hipMalloc(&e->d_key[i], (size_t)g * m * 4), { bgth_encoder_close(e); return nullptr; });
    }
    ENC_TRY(rocprim::radix_sort_pairs(nullptr, e->temp_bytes, e->d_key[0], e->d_key[1], (int32_t*)nullptr, (int32_t*)nullptr, (unsigned)(g * m), 0, 22,
                                      e->stream), { bgth_encoder_close(e); return nullptr; });
    ENC_TRY(hipMalloc(&e->d_temp, e->temp_bytes ? e->temp_bytes : 16), { bgth_encoder_close(e); return nullptr; });
    std::vector<int32_t> ident((size_t)g * m);
    for (int k = 0; k < g; ++k) for (int j = 0; j < m; ++j) ident[(size_t)k * m + j] = j;     // identity start (ref pbwt.c:92-105)
    ENC_TRY(hipMemcpy(e->d_state, ident.data(), ident.size() * 4, hipMemcpyHostToDevice), { bgth_encoder_close(e); return nullptr; });
    e->image.reserve(1 << 20);
    const int32_t hdr[3] = {m, g, shift};                                                       // ref pbwt.c:199-219
    e->image.insert(e->image.end(), (const uint8_t*)"PBF\1", (const uint8_t*)"PBF\1" + 4);
    e->image.insert(e->image.end(), (const uint8_t*)hdr, (const uint8_t*)hdr + 12);
    return e;
}

static int ensure_capacity(bgth_encoder_t *e, int64_t rows, int32_t n_units, int32_t n_snap)
{
    // `rows` covers whole units: every (unit, plane) has an output region of unit_rows * m bytes
    if (rows <= e->rows_cap && n_units <= e->units_cap && n_snap <= e->snap_cap) return 0;
    const int m = e->m, g = e->g;
    if (rows < e->rows_cap) rows = e->rows_cap;
    if (n_units < e->units_cap) n_units = e->units_cap;
    if (n_snap < e->snap_cap) n_snap = e->snap_cap;
    free_batch_buffers(e);
    const size_t ugm = (size_t)n_units * g * m;
    ENC_TRY(hipMalloc(&e->d_codes, (size_t)rows * e->stride), return -1);
    ENC_TRY(hipMemset(e->d_codes, 0, (size_t)rows * e->stride), return -1);                     // the padding stays 0
    ENC_TRY(hipMalloc(&e->d_out, (size_t)g * rows * m), return -1);                             // a row of m bits: at most m bytes
    ENC_TRY(hipMalloc(&e->d_flag, ugm), return -1);
    ENC_TRY(hipMalloc(&e->d_true, (ugm + (size_t)g * m) * 4), return -1);
    ENC_TRY(hipMalloc(&e->d_perms, (ugm + (size_t)g * m) * 4), return -1);
    ENC_TRY(hipMalloc(&e->d_local, ugm * 4), return -1);
    ENC_TRY(hipMalloc(&e->d_perm, ugm * 4), return -1);
    ENC_TRY(hipMalloc(&e->d_cid, ugm * 4), return -1);
    if (!e->wpt) ENC_TRY(hipMalloc(&e->d_ccls, ugm * 2), return -1);
    ENC_TRY(hipMalloc(&e->d_row_len, (size_t)g * rows * 4), return -1);
    ENC_TRY(hipMalloc(&e->d_snap, ((size_t)g * n_snap * m + 1) * 4), return -1);
    ENC_TRY(hipMalloc(&e->d_snap_base, (size_t)n_units * 4), return -1);
    ENC_TRY(hipMalloc(&e->d_out_len, (size_t)n_units * g * 8), return -1);
    e->rows_cap = rows; e->units_cap = n_units; e->snap_cap = n_snap;
    return 0;
}

template <bool EMIT>
static void launch_encode(const bgth_encoder_t *e, const EncodeArgs &a, int n_units)
{
    const dim3 grid((unsigned)e->g, (unsigned)n_units), block(kThreads);
    if (e->wpt > 8) {                                   // directories in memory: any number of words per thread
        if (e->wpt == 16)      hipLaunchKernelGGL((encode_huge_kernel<16, EMIT>), grid, block, 0, e->stream, a);
        else if (e->wpt == 32) hipLaunchKernelGGL((encode_huge_kernel<32, EMIT>), grid, block, 0, e->stream, a);
        else                   hipLaunchKernelGGL((encode_huge_kernel<64, EMIT>), grid, block, 0, e->stream, a);
        return;
    }
    if (e->wpt) {
        const size_t lds = (size_t)2 * (kThreads * e->wpt + 1) * sizeof(uint2);
        if (e->wpt == 2) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_wide_kernel<2, EMIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((encode_wide_kernel<2, EMIT>), grid, block, lds, e->stream, a);
        } else if (e->wpt == 4) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_wide_kernel<4, EMIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((encode_wide_kernel<4, EMIT>), grid, block, lds, e->stream, a);
        } else {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&encode_wide_kernel<8, EMIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((encode_wide_kernel<8, EMIT>), grid, block, lds, e->stream, a);
        }
        return;
    }
    if (e->cpt == 4)       hipLaunchKernelGGL((encode_kernel<4, EMIT>),  grid, block, 0, e->stream, a);
    else if (e->cpt == 8)  hipLaunchKernelGGL((encode_kernel<8, EMIT>),  grid, block, 0, e->stream, a);
    else if (e->cpt == 20) hipLaunchKernelGGL((encode_kernel<20, EMIT>), grid, block, 0, e->stream, a);
    else                   hipLaunchKernelGGL((encode_kernel<32, EMIT>), grid, block, 0, e->stream, a);
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int encode_batch(bgth_encoder_t *e, const uint8_t *codes, int64_t rows, bool packed)
{
    const bool trace = getenv("BGTH_TRACE") != nullptr;
    const double t_begin = now_ms();
    const int m = e->m, g = e->g;
    const int64_t mask = ((int64_t)1 << e->shift) - 1;
    // units: a call of few rows, or more than two planes (the sort key holds one plane bit), is one unit
    // The time of a pass is the time of ONE workgroup over its unit (they all run at once), so a call that does not
    // fill the chip with full-size units gets smaller ones: about one workgroup per CU.
    int64_t want = e->unit_rows;
    // (the chain of phase B grows with the number of units: below ~sqrt(9 rows) narrow / sqrt(1.4 rows) wide it costs more than it saves)
    // (beyond 262,144 columns a workgroup needs ~0.4 ms per row -- every directory access goes to the L2 --: the parallelism of many
    //  short units is worth more than the longer chain: 1,000,000 columns, 1,024 rows per call: 2 units of 512 -> 16 of 64)
    const int64_t least = e->wpt > 8 ? 64 : e->wpt ? 256 : 512;
    if (!e->unit_fixed) while (want > least && (rows + want - 1) / want * g < 256) want >>= 1;
    const bool parallel = rows > want && g <= 2;
    const int64_t unit_rows = parallel ? want : rows;
    const int32_t n_units = (int32_t)((rows + unit_rows - 1) / unit_rows);
    e->h_snap_base.assign((size_t)n_units, 0);
    int32_t n_snap = 0;
    for (int64_t r = 0; r < rows; ++r) {
        if (r % unit_rows == 0) e->h_snap_base[(size_t)(r / unit_rows)] = n_snap;
        if (((e->n + r) & mask) == 0) ++n_snap;
    }
    if (ensure_capacity(e, (int64_t)n_units * unit_rows, n_units, n_snap) < 0) return -1;
    if (e->wpt > 8) {
        const size_t need = (size_t)n_units * e->g * 2 * ((size_t)kThreads * e->wpt + 1);
        if (need > e->gdir_cap) {
            hipFree(e->d_gdir); e->d_gdir = nullptr; e->gdir_cap = 0;
            ENC_TRY(hipMalloc(&e->d_gdir, need * sizeof(uint2)), return -1);
            e->gdir_cap = need;
        }
    }
    const double t_alloc = now_ms();
    const size_t gm = (size_t)g * m;
    EncodeArgs a;
    a.codes = e->d_codes; a.n_rows = rows; a.row0 = e->n; a.m = m; a.mask = (int32_t)mask; a.g = g; a.unit_rows = (int32_t)unit_rows; a.stride = e->stride;
#ifdef BGTH_ABLATE
    a.debug = getenv("BGTH_ENC_DEBUG") ? atoi(getenv("BGTH_ENC_DEBUG")) : 0;
#else
    a.debug = 0;
#endif
    a.out = e->d_out; a.cap = unit_rows * (int64_t)m; a.out_len = e->d_out_len; a.row_len = e->d_row_len;
    a.snap = e->d_snap; a.snap_base = e->d_snap_base; a.n_snap = n_snap; a.status = e->d_status; a.gdir = e->d_gdir;
    if (packed) {                                           // a quarter of the bytes over PCIe, spread on the device
        const size_t nb = (size_t)rows * ((m + 3) >> 2);
        if (nb > e->packed_in_cap) {
            hipFree(e->d_packed_in); e->d_packed_in = nullptr; e->packed_in_cap = 0;
            ENC_TRY(hipMalloc(&e->d_packed_in, nb), return -1);
            e->packed_in_cap = nb;
        }
        ENC_TRY(hipMemcpyAsync(e->d_packed_in, codes, nb, hipMemcpyHostToDevice, e->stream), return -1);
        hipLaunchKernelGGL(unpack_codes_kernel, dim3((unsigned)((((m + 3) >> 2) + 255) / 256), (unsigned)(rows < 16384 ? rows : 16384)), dim3(256), 0, e->stream,
                           e->d_packed_in, rows, m, e->stride, e->d_codes);
    } else
        ENC_TRY(hipMemcpy2DAsync(e->d_codes, (size_t)e->stride, codes, (size_t)m, (size_t)m, (size_t)rows, hipMemcpyHostToDevice, e->stream), return -1);
    ENC_TRY(hipMemcpyAsync(e->d_snap_base, e->h_snap_base.data(), (size_t)n_units * 4, hipMemcpyHostToDevice, e->stream), return -1);
    ENC_TRY(hipMemsetAsync(e->d_status, 0, 4, e->stream), return -1);
    if (trace) hipStreamSynchronize(e->stream);
    const double t_up = now_ms();
    hipEvent_t ev0, ev1;
    ENC_TRY(hipEventCreate(&ev0), return -1);
    ENC_TRY(hipEventCreate(&ev1), return -1);
    hipEventRecord(ev0, e->stream);
    if (n_units > 1) {
        // A: every unit from the identity order -> its own order (ranks and permutation), then the classes of identical columns
        a.rank_in = nullptr; a.rank_out = e->d_local; a.perm_out = e->d_perm;
        launch_encode<false>(e, a, n_units);
        const int wpu = (int)((unit_rows + 31) / 32);
        const size_t cb_words = (size_t)n_units * wpu * 2 * m;
        if (cb_words > e->colbits_cap) {
            hipFree(e->d_colbits); e->d_colbits = nullptr; e->colbits_cap = 0;
            ENC_TRY(hipMalloc(&e->d_colbits, cb_words * 4), return -1);
            e->colbits_cap = cb_words;
        }
        hipLaunchKernelGGL(column_bits_kernel, dim3((unsigned)((m + 1023) / 1024), (unsigned)(n_units * wpu)), dim3(256), 0, e->stream,
                           e->d_codes, e->stride, rows, (int)unit_rows, wpu, m, e->d_colbits);
        hipLaunchKernelGGL(class_flags_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)g, (unsigned)n_units), dim3(256), 0, e->stream,
                           e->d_colbits, wpu, m, g, e->d_perm, e->d_flag);
        hipLaunchKernelGGL(class_ids_kernel, dim3((unsigned)(n_units * g)), dim3(kThreads), 0, e->stream, m, e->d_flag, e->d_cid);
        // B: the true order before every unit: one stable sort by class per unit, then all orders turned into ranks
        hipEvent_t evb0 = nullptr, evb1 = nullptr;
        if (trace && hipEventCreate(&evb0) == hipSuccess && hipEventCreate(&evb1) == hipSuccess) hipEventRecord(evb0, e->stream);
        const unsigned nk = (unsigned)gm, kb = (nk + 255) / 256;
        const int cbits = e->wpt > 8 ? 21 : e->wpt ? 18 : 15;   // bits of a class number (< m)
        hipLaunchKernelGGL(invert_kernel, dim3(kb), dim3(256), 0, e->stream, m, (int64_t)gm, e->d_state, e->d_perms);
        if (!e->wpt && !getenv("BGTH_ENC_LIBSORT")) {
            // one launch for the whole chain: a workgroup per plane, the order resident in LDS (order_chain_kernel)
            const int64_t n_cls = (int64_t)n_units * (int64_t)gm;
            hipLaunchKernelGGL(column_class_kernel, dim3((unsigned)((n_cls + 255) / 256)), dim3(256), 0, e->stream, m, n_cls, e->d_cid, e->d_local, e->d_ccls);
            const int digit = e->cpt == 32 ? 3 : 4;
            const size_t lds = (size_t)kThreads * e->cpt * 4 + ((size_t)2 << digit) * kThreads;
            const dim3 cgrid((unsigned)g), cblock(kThreads);
            if (e->cpt == 4) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&order_chain_kernel<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((order_chain_kernel<4, 4>), cgrid, cblock, lds, e->stream, m, g, n_units, e->d_ccls, e->d_perms);
            } else if (e->cpt == 8) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&order_chain_kernel<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((order_chain_kernel<8, 4>), cgrid, cblock, lds, e->stream, m, g, n_units, e->d_ccls, e->d_perms);
            } else if (e->cpt == 20) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&order_chain_kernel<20, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((order_chain_kernel<20, 4>), cgrid, cblock, lds, e->stream, m, g, n_units, e->d_ccls, e->d_perms);
            } else {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&order_chain_kernel<32, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((order_chain_kernel<32, 3>), cgrid, cblock, lds, e->stream, m, g, n_units, e->d_ccls, e->d_perms);
            }
        } else
        for (int32_t k = 0; k < n_units; ++k) {                 // wide cohorts (the order does not fit the LDS): a library sort per unit
            hipLaunchKernelGGL(class_keys_kernel, dim3(kb), dim3(256), 0, e->stream, m, g, cbits, e->d_cid + (size_t)k * gm,
                               e->d_local + (size_t)k * gm, e->d_perms + (size_t)k * gm, e->d_key[0]);
            ENC_TRY(rocprim::radix_sort_pairs(e->d_temp, e->temp_bytes, e->d_key[0], e->d_key[1], e->d_perms + (size_t)k * gm,
                                              e->d_perms + (size_t)(k + 1) * gm, nk, 0, (unsigned)(cbits + (g > 1 ? 1 : 0)), e->stream), return -1);
        }
        if (evb1) {
            hipEventRecord(evb1, e->stream);
            hipEventSynchronize(evb1);
            float msb = 0.f;
            hipEventElapsedTime(&msb, evb0, evb1);
            fprintf(stderr, "[bgth_encoder] phase B (the chain of %d units): %.2f ms\n", n_units, (double)msb);
            hipEventDestroy(evb0); hipEventDestroy(evb1);
        }
        {
            const int64_t n_total = (int64_t)(n_units + 1) * (int64_t)gm;
            hipLaunchKernelGGL(invert_kernel, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, e->stream, m, n_total, e->d_perms, e->d_true);
        }
        ENC_TRY(hipMemcpyAsync(e->d_state, e->d_true + (size_t)n_units * gm, gm * 4, hipMemcpyDeviceToDevice, e->stream), return -1);
        // C: every unit from its true start order
        a.rank_in = e->d_true; a.rank_out = e->d_local; a.perm_out = nullptr;
        launch_encode<true>(e, a, n_units);
    } else {
        a.rank_in = e->d_state; a.rank_out = e->d_state; a.perm_out = nullptr;
        launch_encode<true>(e, a, 1);
    }
    hipEventRecord(ev1, e->stream);
    ENC_TRY(hipGetLastError(), return -1);
    int32_t status = 0;
    e->h_out_len.resize((size_t)n_units * g);
    ENC_TRY(hipMemcpyAsync(&status, e->d_status, 4, hipMemcpyDeviceToHost, e->stream), return -1);
    ENC_TRY(hipMemcpyAsync(e->h_out_len.data(), e->d_out_len, e->h_out_len.size() * 8, hipMemcpyDeviceToHost, e->stream), return -1);
    e->h_row_len.resize((size_t)g * rows);
    ENC_TRY(hipMemcpyAsync(e->h_row_len.data(), e->d_row_len, (size_t)g * rows * 4, hipMemcpyDeviceToHost, e->stream), return -1);
    e->h_snap.resize((size_t)g * n_snap * m);
    if (n_snap) ENC_TRY(hipMemcpyAsync(e->h_snap.data(), e->d_snap, e->h_snap.size() * 4, hipMemcpyDeviceToHost, e->stream), return -1);
    ENC_TRY(hipStreamSynchronize(e->stream), return -1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev0, ev1);
    e->kernel_ms += ms;
    hipEventDestroy(ev0); hipEventDestroy(ev1);
    if (status != 0) { enc_err("[E::%s] run-length output exceeded its buffer", __func__); return -1; }
    // the run-length bytes of every (unit, plane), packed one after the other
    std::vector<int64_t> base((size_t)n_units * g + 1, 0);
    for (size_t i = 0; i < e->h_out_len.size(); ++i) base[i + 1] = base[i] + e->h_out_len[i];
    if (base.back() > 0) {
        if ((size_t)base.back() > e->h_out_cap) {
            hipHostFree(e->h_out); e->h_out = nullptr; e->h_out_cap = 0;
            ENC_TRY(hipHostMalloc(&e->h_out, (size_t)base.back(), hipHostMallocDefault), return -1);
            e->h_out_cap = (size_t)base.back();
        }
        if ((size_t)base.back() > e->packed_cap) {
            hipFree(e->d_packed); e->d_packed = nullptr; e->packed_cap = 0;
            ENC_TRY(hipMalloc(&e->d_packed, (size_t)base.back()), return -1);
            e->packed_cap = (size_t)base.back();
        }
        if (base.size() > e->base_cap) {
            hipFree(e->d_base); e->d_base = nullptr; e->base_cap = 0;
            ENC_TRY(hipMalloc(&e->d_base, base.size() * 8), return -1);
            e->base_cap = base.size();
        }
        ENC_TRY(hipMemcpyAsync(e->d_base, base.data(), base.size() * 8, hipMemcpyHostToDevice, e->stream), return -1);
        hipLaunchKernelGGL(compact_kernel, dim3((unsigned)(n_units * g)), dim3(kThreads), 0, e->stream, e->d_out, a.cap, e->d_base, e->d_packed);
        ENC_TRY(hipMemcpyAsync(e->h_out, e->d_packed, (size_t)base.back(), hipMemcpyDeviceToHost, e->stream), return -1);
        ENC_TRY(hipStreamSynchronize(e->stream), return -1);
    }
    const double t_down = now_ms();
    // ---- records in file order (ref pbwt.c:288-311): ['S' perms]  'B' { int32 len, bytes } per plane
    size_t grow = (size_t)rows * (1 + 4 * (size_t)g) + (size_t)base.back() + (size_t)n_snap * (1 + (size_t)g * m * 4);
    const size_t old_size = e->image.size();
    e->image.resize(old_size + grow);
    uint8_t *wp = e->image.data() + old_size;
    std::vector<size_t> at((size_t)g, 0);
    int32_t si = 0;
    for (int64_t r = 0; r < rows; ++r) {
        if (r % unit_rows == 0) for (int k = 0; k < g; ++k) at[(size_t)k] = (size_t)base[(size_t)(r / unit_rows) * g + k];
        if (((e->n + r) & mask) == 0) {
            e->idx.push_back(e->taken + (uint64_t)(wp - e->image.data()));
            *wp++ = 'S';
            for (int k = 0; k < g; ++k) {
                memcpy(wp, e->h_snap.data() + ((size_t)k * n_snap + si) * m, (size_t)m * 4);
                wp += (size_t)m * 4;
            }
            ++si;
        }
        *wp++ = 'B';
        for (int k = 0; k < g; ++k) {
            const int32_t l = e->h_row_len[(size_t)k * rows + r];
            memcpy(wp, &l, 4);
            memcpy(wp + 4, e->h_out + at[(size_t)k], (size_t)l);
            wp += 4 + (size_t)l;
            at[(size_t)k] += (size_t)l;
        }
    }
    if (wp != e->image.data() + e->image.size()) { enc_err("[E::%s] record sizes do not add up", __func__); return -1; }
    e->n += rows;
    if (trace)
        fprintf(stderr, "[bgth_encoder] %lld rows, %d units of %lld: buffers %.1f ms, upload %.1f, device + download %.1f (kernels %.1f), records %.1f\n",
                (long long)rows, n_units, (long long)unit_rows, t_alloc - t_begin, t_up - t_alloc, t_down - t_up, (double)ms, now_ms() - t_down);
    return 0;
}

static int write_rows(bgth_encoder_t *e, const uint8_t *codes, int64_t n_rows, bool packed, const char *who)
{
    if (!e || (!codes && n_rows > 0) || n_rows < 0) { enc_err("[E::%s] bad argument", who); return -1; }
    if (e->finished) { enc_err("[E::%s] the image is finished", who); return -1; }
    if (e->broken) { enc_err("[E::%s] an earlier write failed; close this encoder", who); return -1; }
    ENC_TRY(hipSetDevice(e->device), return -1);
    const size_t row_bytes = packed ? (size_t)((e->m + 3) >> 2) : (size_t)e->m;
    for (int64_t done = 0; done < n_rows;) {
        int64_t rows = n_rows - done;
        if (rows > e->batch_rows) rows = e->batch_rows;
        if (encode_batch(e, codes + (size_t)done * row_bytes, rows, packed) < 0) { e->broken = true; return -1; }
        done += rows;
    }
    return 0;
}

extern "C" int bgth_encoder_write(bgth_encoder_t *e, const uint8_t *codes, int64_t n_rows) { return write_rows(e, codes, n_rows, false, __func__); }
extern "C" int bgth_encoder_write_packed(bgth_encoder_t *e, const uint8_t *packed, int64_t n_rows) { return write_rows(e, packed, n_rows, true, __func__); }

extern "C" int64_t bgth_encoder_finish(bgth_encoder_t *e, uint8_t **image)
{
    if (!e || !image) { enc_err("[E::%s] bad argument", __func__); return -1; }
    if (e->finished) { enc_err("[E::%s] the image is finished", __func__); return -1; }
    if (e->broken) { enc_err("[E::%s] an earlier write failed; there is no complete image", __func__); return -1; }
    e->finished = true;
    const uint64_t off = e->taken + (uint64_t)e->image.size();                                  // ref pbwt.c:264-277
    const int64_t n = e->n;
    const int32_t n_idx = (int32_t)e->idx.size();
    e->image.push_back('I');
    e->image.insert(e->image.end(), (const uint8_t*)&n, (const uint8_t*)&n + 8);
    e->image.insert(e->image.end(), (const uint8_t*)&n_idx, (const uint8_t*)&n_idx + 4);
    e->image.insert(e->image.end(), (const uint8_t*)e->idx.data(), (const uint8_t*)e->idx.data() + (size_t)n_idx * 8);
    e->image.insert(e->image.end(), (const uint8_t*)&off, (const uint8_t*)&off + 8);
    uint8_t *p = (uint8_t*)malloc(e->image.size() ? e->image.size() : 1);
    if (!p) { enc_err("[E::%s] out of memory", __func__); return -1; }
    memcpy(p, e->image.data(), e->image.size());
    *image = p;
    return (int64_t)e->image.size();
}

extern "C" int64_t bgth_encoder_take(bgth_encoder_t *e, uint8_t **chunk)
{
    if (!e || !chunk) { enc_err("[E::%s] bad argument", __func__); return -1; }
    if (e->finished || e->broken) { enc_err("[E::%s] nothing to take from a finished or failed encoder", __func__); return -1; }
    const size_t n = e->image.size();
    uint8_t *p = (uint8_t*)malloc(n ? n : 1);
    if (!p) { enc_err("[E::%s] out of memory", __func__); return -1; }
    memcpy(p, e->image.data(), n);
    e->taken += n;
    e->image.clear();
    *chunk = p;
    return (int64_t)n;
}

extern "C" void bgth_encoder_free_image(uint8_t *image) { free(image); }
extern "C" double bgth_encoder_kernel_ms(const bgth_encoder_t *e) { return e ? e->kernel_ms : 0.0; }
