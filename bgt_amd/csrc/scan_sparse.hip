// The sparse-plane tracker (round 4): plane 1 of a BGT matrix without tracking a rank per column.
//
// Plane 1 is the high bit of the 2-bit code (reference import.c:96-99): it is set for missing calls and <M> alleles only, so
// its rows are almost empty -- 20 ones in 20,000 columns on the benchmark cohort, 4 % of that in the <M> sites.  The dense
// kernels nevertheless pay one lookup per tracked column and row for it (half of a scan), because a column's rank moves
// whenever ANY column before it carries a one (reference pbwt.c:79-88, 147-166).  Here the order itself is kept, as a set:
//
//     order of plane 1 at any row  =  [ epoch positions still alive, ascending ]  ++  [ tail slots still alive, ascending ]
//
// An epoch starts from a checkpoint (the ranks of the sub-block, inverted): e2s[position] = output slot of the column that
// sits there (-1: not selected).  A row with ones at positions p_1 < p_2 < ... (known from its run-length string alone) moves
// exactly those elements behind everything else, in that order -- the stable partition of pbwt.c:79-88 -- i.e. each is
// looked up (the p-th live element: a SELECT on a bitmap with block counts), deleted, and appended to the tail.  Nothing else
// of the order changes, so a row costs work proportional to its ones, not to the columns.  What it yields is what the bit
// plane H1 needs: the output slots of the columns whose bit is 1.  When the tail fills up the order is compacted into a new
// epoch.  One workgroup per sub-block (the rows depend on each other), four waves sharing the selects of a row.
//
// The dense kernels then run plane 0 alone (plane_kernel / scan_kernel with a.skip1) and the planes meet in count_planes_kernel
// exactly as they do for the plane-split kernels.  Chosen per scan when the image's plane 1 is sparse (bgt_hip.cpp: plane 1
// statistics); any image can still take the dense path, and the parity tests run both.
#include "scan_device.inc.h"

namespace bgth {

namespace {

constexpr int kSpNT = 256;                  // threads of a tracker workgroup

// a set of slots with select: bitmap + live counts per 8 words (c2) and per 128 words (c1), p1 = exclusive prefix of c1
struct LiveSet {
    uint32_t *bits, *c2, *c1, *p1;
    int nb1;                                // c1 entries
};

// slot of the k-th (0-based) live element; k must be below the live total (a malformed row cannot make this read outside the set)
__device__ __forceinline__ uint32_t live_select(const LiveSet &S, uint32_t k)
{
    int B = 0;
    for (int i = 1; i < S.nb1; ++i) B += S.p1[i] <= k ? 1 : 0;
    uint32_t rem = k - S.p1[B];
    const uint4 *q = reinterpret_cast<const uint4*>(S.c2 + 16 * B);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    const uint32_t cc[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    int i2 = 0;
    bool open = true;
#pragma unroll
    for (int i = 0; i < 15; ++i) {
        const bool skip = open && rem >= cc[i];
        rem -= skip ? cc[i] : 0u;
        i2 += skip ? 1 : 0;
        open = skip;
    }
    const int w0 = 8 * (16 * B + i2);
    const uint4 *wq = reinterpret_cast<const uint4*>(S.bits + w0);
    const uint4 e = wq[0], f = wq[1];
    const uint32_t ww[8] = {e.x, e.y, e.z, e.w, f.x, f.y, f.z, f.w};
    int iw = 0;
    uint32_t word = ww[0];
    open = true;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const uint32_t pc = (uint32_t)__popc(ww[i]);
        const bool skip = open && rem >= pc;
        rem -= skip ? pc : 0u;
        iw += skip ? 1 : 0;
        word = skip ? ww[i + 1] : word;
        open = skip;
    }
    // the rem-th set bit of word
    uint32_t base = 0;
#pragma unroll
    for (int width = 16; width >= 1; width >>= 1) {
        const uint32_t lowc = (uint32_t)__popc(word & ((1u << width) - 1u));
        const bool up = rem >= lowc;
        rem -= up ? lowc : 0u;
        base += up ? (uint32_t)width : 0u;
        word = up ? word >> width : word;
    }
    return 32u * (uint32_t)(w0 + iw) + base;
}

__device__ __forceinline__ void live_remove(const LiveSet &S, uint32_t slot)
{
    atomicAnd(&S.bits[slot >> 5], ~(1u << (slot & 31)));
    atomicSub(&S.c2[slot >> 8], 1u);
    atomicSub(&S.c1[slot >> 12], 1u);
}
__device__ __forceinline__ void live_insert(const LiveSet &S, uint32_t slot)
{
    atomicOr(&S.bits[slot >> 5], 1u << (slot & 31));
    atomicAdd(&S.c2[slot >> 8], 1u);
    atomicAdd(&S.c1[slot >> 12], 1u);
}
// p1 = exclusive prefix of c1 (one wave; p1[nb1] = total)
__device__ __forceinline__ void live_prefix(const LiveSet &S, int lane)
{
    uint32_t carry = 0;
    for (int i0 = 0; i0 < S.nb1; i0 += 64) {
        const int i = i0 + lane;
        const uint32_t v = i < S.nb1 ? S.c1[i] : 0u;
        const uint32_t incl = wave_incl_add(v);
        if (i < S.nb1) S.p1[i] = carry + incl - v;
        carry += lane63(incl);
    }
    if (lane == 0) S.p1[S.nb1] = carry;
}
// number of live elements in slots below word w (any thread)
__device__ __forceinline__ uint32_t live_before_word(const LiveSet &S, int w)
{
    const int B = w >> 7, b2 = w >> 3;
    uint32_t n = S.p1[B];
    for (int i = 16 * B; i < b2; ++i) n += S.c2[i];
    for (int i = 8 * b2; i < w; ++i) n += (uint32_t)__popc(S.bits[i]);
    return n;
}

__device__ __forceinline__ int32_t load_coherent(const int32_t *p)
{   // written by other waves of this workgroup during the kernel: not through this CU's vector L1
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

__global__ __launch_bounds__(kSpNT) void sparse_plane1_kernel(const ScanArgs a, const uint64_t *__restrict__ rowdesc,
                                                              const uint8_t *__restrict__ rle)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bl = blockIdx.x;
    if (bl >= a.n_blk) return;
    const int m = a.m;
    const int nwa = ((m + 31) / 32 + 127) & ~127;                       // words of the epoch set, whole L1 blocks
    const int nwt = a.sp_tcap / 32;                                     // words of the tail set (sp_tcap is a multiple of 4096)
    const int mpad = 32 * nwa;
    // LDS: the two sets, the positions of a row's ones, a few shared scalars
    uint32_t *lds = reinterpret_cast<uint32_t*>(smem);
    LiveSet A, T;
    A.bits = lds;                 A.c2 = A.bits + nwa;      A.nb1 = nwa / 128; A.c1 = A.c2 + nwa / 8; A.p1 = A.c1 + A.nb1;
    T.bits = A.p1 + A.nb1 + 1;    T.bits += (4 - ((T.bits - lds) & 3)) & 3;                         // 16-byte aligned
    T.c2 = T.bits + nwt;          T.nb1 = nwt / 128;        T.c1 = T.c2 + nwt / 8; T.p1 = T.c1 + T.nb1;
    uint32_t *shv = T.p1 + T.nb1 + 1;                                   // [0..1] ones of the row in either buffer, [4..5] pass counters
    uint32_t *ipos = shv + 8;                                           // [2][sp_icap] positions of a row's ones

    const int64_t blk = (int64_t)a.blk0 + bl;
    const int64_t blk_beg = blk << a.shift;
    int64_t blk_end = (blk + 1) << a.shift;
    if (blk_end > a.row1) blk_end = a.row1;

    int32_t *e2s = a.sp_e2s + (size_t)bl * 2 * (size_t)mpad;           // two epoch tables: current / next
    int32_t *tailref = a.sp_tail + (size_t)bl * (size_t)a.sp_tcap;
    int cur = 0;

    // ---- epoch 0: the order of the checkpoint.  e2s[rank of column c] = output slot of c
    {
        const int32_t *rk = a.rank0 + blk * a.rank0_blk_stride + m;    // plane 1
        for (int c = tid; c < m; c += kSpNT) e2s[rk[c]] = a.sp_slot_of_col[c];
        for (int i = tid; i < nwa; i += kSpNT) {
            const int lo = 32 * i;
            A.bits[i] = lo + 32 <= m ? 0xffffffffu : lo < m ? (1u << (m - lo)) - 1u : 0u;
        }
        for (int i = tid; i < nwt; i += kSpNT) T.bits[i] = 0u;
        for (int i = tid; i < nwt / 8; i += kSpNT) T.c2[i] = 0u;
        for (int i = tid; i < T.nb1; i += kSpNT) T.c1[i] = 0u;
        __syncthreads();
        for (int i = tid; i < nwa / 8; i += kSpNT) { uint32_t n = 0; for (int k = 0; k < 8; ++k) n += (uint32_t)__popc(A.bits[8 * i + k]); A.c2[i] = n; }
        __syncthreads();
        for (int i = tid; i < A.nb1; i += kSpNT) { uint32_t n = 0; for (int k = 0; k < 16; ++k) n += A.c2[16 * i + k]; A.c1[i] = n; }
        __threadfence();
        __syncthreads();
        if (wave == 0) { live_prefix(A, lane); live_prefix(T, lane); }
        __syncthreads();
    }
    uint32_t n_alive = (uint32_t)m, tail_len = 0;                       // (uniform: every thread keeps its copy)
    // output of the previous item of this thread: its slot is in flight while the next select runs
    bool pend = false;
    int32_t pend_slot = -1;
    int64_t pend_row = 0;
    auto flush_pending = [&]() {
        if (pend && pend_slot >= 0 && pend_row >= a.row0)
            atomicOr(reinterpret_cast<unsigned long long*>(a.h1 + (size_t)(pend_row - a.h_row0) * a.n_chunks + (pend_slot >> 6)), 1ull << (pend_slot & 63));
        pend = false;
    };
    // Roles: the last wave DECODES (the positions of the ones of row r + 1, from its string, while the others work on row r);
    // the other waves share the selects of a row, kSel items a pass.  Only LDS is exchanged inside a row, so the barriers wait
    // for LDS alone (lds_barrier): the global traffic -- output bits, tail references, epoch tables -- stays in flight.
    constexpr int kSel = kSpNT - 64;
    const int icap = a.sp_icap;
    const bool decoder = wave == kSpNT / 64 - 1;
    // d = the row's plane-1 descriptor, w0 = this lane's four bytes of the string's first chunk: both were fetched rows ahead
    auto decode_row = [&](int64_t row, int buf, uint64_t d, uint32_t w0) {      // one wave
        uint32_t *out = ipos + (size_t)buf * icap;
        uint32_t item = 0;
        if (row < blk_end) {
            const uint32_t slen = (uint32_t)(d >> kDescLenShift);
            const uint32_t *q4 = reinterpret_cast<const uint32_t*>(rle + (d & kDescOffMask));
            uint32_t pos = 0;
            for (uint32_t base = 0; base < slen; base += 256) {
                const uint32_t k0 = base + 4u * (uint32_t)lane;
                const uint32_t w = base == 0 ? w0 : k0 < slen ? q4[(base >> 2) + lane] : 0u;
                const ChunkDecode cd = decode_chunk(w, k0, slen, lane);
                const uint32_t incl = wave_incl_add(cd.run);
                const uint32_t lane_start = pos + incl - cd.run;
                uint32_t ones[4], st[4], mine = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    st[i] = lane_start + cd.before[i];
                    const uint32_t room = st[i] < (uint32_t)m ? (uint32_t)m - st[i] : 0u;
                    ones[i] = cd.valid[i] && cd.bit[i] ? (cd.l[i] < room ? cd.l[i] : room) : 0u;
                    mine += ones[i];
                }
                if (__ballot(mine != 0u)) {
                    const uint32_t inclo = wave_incl_add(mine);
                    uint32_t j = item + inclo - mine;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        for (uint32_t t = 0; t < ones[i]; ++t, ++j)
                            if (j < (uint32_t)icap) out[j] = st[i] + t;
                    item += lane63(inclo);
                }
                pos += lane63(incl);
                if (cd.stop) break;
            }
        }
        if (lane == 0) shv[buf] = item < (uint32_t)icap ? item : (uint32_t)icap;   // (icap >= the most ones of any row of the image)
    };
    // the decoder's fetches run ahead of its decoding: descriptor three rows, first chunk two rows (vector loads of uniform
    // addresses too: nothing here may wait on the scalar cache's counter, which the LDS barriers share)
    auto vidx = [](uint64_t i) { uint32_t lo = (uint32_t)i, hi = (uint32_t)(i >> 32); asm volatile("" : "+v"(lo), "+v"(hi)); return (uint64_t)hi << 32 | lo; };
    auto fetch_desc = [&](int64_t row) -> uint64_t { return row < blk_end ? rowdesc[vidx((uint64_t)(2 * row + 1))] : 0ull; };
    auto uniform64 = [](uint64_t v) { return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto fetch_w0 = [&](uint64_t d) -> uint32_t {
        const uint32_t slen = (uint32_t)(d >> kDescLenShift);
        return 4u * (uint32_t)lane < slen ? reinterpret_cast<const uint32_t*>(rle + (d & kDescOffMask))[lane] : 0u;
    };
    uint64_t dA = 0, dB = 0;                                            // descriptors of row + 1, row + 2 (decoder wave)
    uint32_t wA = 0;                                                    // first chunk of row + 1
    if (tid < 8) shv[tid] = 0u;
    lds_barrier();
    if (decoder) {
        const uint64_t d0 = uniform64(fetch_desc(blk_beg));
        dA = uniform64(fetch_desc(blk_beg + 1));
        dB = fetch_desc(blk_beg + 2);
        wA = fetch_w0(dA);
        decode_row(blk_beg, 0, d0, fetch_w0(d0));
    }
    lds_barrier();

    // (profiling build only: cycles of wave 0 in 0 select, 1 apply + prefix, 2 barrier; of the decoder in 3 decode, 4 barrier)
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = BGTH_TIMES(a) ? __builtin_amdgcn_s_memtime() : 0ull;
    for (int64_t row = blk_beg; row < blk_end; ++row) {
        const int buf = (int)(row - blk_beg) & 1;
        const uint32_t n1 = shv[buf];
        const uint32_t *items = ipos + (size_t)buf * icap;
        uint32_t delA = 0, delT = 0;                                     // deleted so far in this row: epoch set / tail
        uint32_t j0 = 0;
        bool first = true;
        if (n1 > 64) do {                                               // passes of kSel ones
            const uint32_t j = j0 + (uint32_t)tid;
            const bool valid = !decoder && j < n1 && tid < kSel;
            uint32_t p = 0, slot = 0;
            bool in_alive = false;
            if (decoder) {
                if (first) {
                    const uint64_t dN = uniform64(dB);                  // row + 2: its first chunk and the descriptor of row + 3 travel
                    const uint32_t wN = fetch_w0(dN);                   // behind this decode
                    dB = fetch_desc(row + 3);
                    decode_row(row + 1, buf ^ 1, dA, wA);
                    dA = dN; wA = wN;
                }
            }
            else if (valid) {
                p = items[j];
                in_alive = p < n_alive;
                slot = in_alive ? live_select(A, p - delA) : live_select(T, p - n_alive - delT);
                if (!in_alive) shv[6] = 1u;
            }
            first = false;
            lds_barrier();                                              // every select of the pass is done
            if (shv[6]) {
                // (rare) an element that was one in an earlier row is one again: its reference was stored to memory by another
                // thread then -- nothing inside a row waits for global stores, so make sure of it now
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_barrier();
            }
            uint32_t nv = n1 - j0; if (nv > (uint32_t)kSel) nv = (uint32_t)kSel;
            // the items are ascending: those inside the epoch set come first
            uint32_t nA = 0;
            {
                const uint64_t b = __ballot(valid && in_alive);
                if (lane == 0 && b) atomicAdd(&shv[4 + ((j0 / kSel) & 1)], (uint32_t)__popcll(b));
            }
            if (valid) {
                int32_t ref = (int32_t)slot;
                if (in_alive) live_remove(A, slot);
                else {
                    live_remove(T, slot);
                    // (rare: an element that was one before is one again)  its reference was stored by another thread rows ago
                    ref = load_coherent(tailref + slot);
                }
                const uint32_t ts = tail_len + j;                      // appended in the order of the positions
                live_insert(T, ts);
                __hip_atomic_store(tailref + ts, ref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                flush_pending();
                pend = true; pend_row = row; pend_slot = load_coherent(e2s + (size_t)cur * mpad + ref);
            }
            lds_barrier();
            nA = shv[4 + ((j0 / kSel) & 1)];
            if (wave == 0) { live_prefix(A, lane); live_prefix(T, lane); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (tail references stored in this pass are in memory behind it)
            lds_barrier();
            if (tid == 0) { shv[4 + ((j0 / kSel) & 1)] = 0u; shv[6] = 0u; }   // (their next use is behind the next barrier)
            delA += nA; delT += nv - nA;
            j0 += (uint32_t)kSel;
        } while (j0 < n1);
        else {
            // ---- the usual row: at most 64 ones.  Wave 0 selects, applies and renews the prefixes on its own -- its LDS operations
            // complete in order, so nothing inside the row needs a barrier -- while the decoder wave prepares the next row.
            if (decoder) {
                const uint64_t dN = uniform64(dB);
                const uint32_t wN = fetch_w0(dN);
                dB = fetch_desc(row + 3);
                decode_row(row + 1, buf ^ 1, dA, wA);
                dA = dN; wA = wN;
                BGTH_TICK(3);
            } else if (wave == 0) {
                const bool valid = (uint32_t)lane < n1;
                uint32_t slot = 0;
                bool in_alive = false;
                if (valid) {
                    const uint32_t p = items[lane];
                    in_alive = p < n_alive;
                    slot = in_alive ? live_select(A, p) : live_select(T, p - n_alive);
                }
                BGTH_TICK(0);
                const uint64_t ba = __ballot(valid && in_alive), bt = __ballot(valid && !in_alive);
                if (bt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (rare: a reference this wave stored rows ago is read back)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (valid) {
                    int32_t ref = (int32_t)slot;
                    if (in_alive) live_remove(A, slot);
                    else { live_remove(T, slot); ref = load_coherent(tailref + slot); }
                    const uint32_t ts = tail_len + (uint32_t)lane;
                    live_insert(T, ts);
                    __hip_atomic_store(tailref + ts, ref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    flush_pending();
                    pend = true; pend_row = row; pend_slot = load_coherent(e2s + (size_t)cur * mpad + ref);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (n1) { live_prefix(A, lane); live_prefix(T, lane); }
                if (lane == 0) shv[2 + buf] = (uint32_t)__popcll(ba);
                BGTH_TICK(1);
            }
            lds_barrier();
            if (wave == 0) BGTH_TICK(2); else if (decoder) BGTH_TICK(4);
            delA = shv[2 + buf];
        }
        n_alive -= delA; tail_len += n1;

        // ---- a full tail: compact the order into a new epoch (every element is live again at its new position)
        if (tail_len + (uint32_t)icap > (uint32_t)a.sp_tcap && row + 1 < blk_end) {
            flush_pending();
            __threadfence();
            __syncthreads();                                            // every reference and table entry written so far is in memory
            const int32_t *src = e2s + (size_t)cur * mpad;
            int32_t *dst = e2s + (size_t)(cur ^ 1) * mpad;
            for (int i = tid; i < mpad; i += kSpNT) {                   // one bit a thread: independent loads
                const uint32_t word = A.bits[i >> 5];
                if (word >> (i & 31) & 1u)
                    dst[live_before_word(A, i >> 5) + (uint32_t)__popc(word & ((1u << (i & 31)) - 1u))] = load_coherent(src + i);
            }
            for (int i = tid; i < a.sp_tcap; i += kSpNT) {
                const uint32_t word = T.bits[i >> 5];
                if (word >> (i & 31) & 1u)
                    dst[n_alive + live_before_word(T, i >> 5) + (uint32_t)__popc(word & ((1u << (i & 31)) - 1u))] =
                        load_coherent(src + load_coherent(tailref + i));
            }
            __threadfence();
            __syncthreads();
            for (int i = tid; i < nwa; i += kSpNT) {
                const int lo = 32 * i;
                A.bits[i] = lo + 32 <= m ? 0xffffffffu : lo < m ? (1u << (m - lo)) - 1u : 0u;
            }
            for (int i = tid; i < nwt; i += kSpNT) T.bits[i] = 0u;
            for (int i = tid; i < nwt / 8; i += kSpNT) T.c2[i] = 0u;
            for (int i = tid; i < T.nb1; i += kSpNT) T.c1[i] = 0u;
            __syncthreads();
            for (int i = tid; i < nwa / 8; i += kSpNT) { uint32_t n = 0; for (int k = 0; k < 8; ++k) n += (uint32_t)__popc(A.bits[8 * i + k]); A.c2[i] = n; }
            __syncthreads();
            for (int i = tid; i < A.nb1; i += kSpNT) { uint32_t n = 0; for (int k = 0; k < 16; ++k) n += A.c2[16 * i + k]; A.c1[i] = n; }
            __syncthreads();
            if (wave == 0) { live_prefix(A, lane); live_prefix(T, lane); }
            __syncthreads();
            cur ^= 1; n_alive = (uint32_t)m; tail_len = 0;
        }
    }
    flush_pending();
#ifdef BGTH_ABLATE
    if (BGTH_TIMES(a) && lane == 0 && (wave == 0 || decoder)) for (int i = 0; i < 8; ++i) atomicAdd(a.debug_times + i, tsum[i]);
#endif
}

// n1[row] = ones of plane 1 of row (one wave per string); stats[0] = max, stats[1..2] = sum (64 bit)
__global__ __launch_bounds__(256) void plane1_ones_kernel(const uint64_t *__restrict__ rowdesc, const uint8_t *__restrict__ rle, int64_t n_rows,
                                                          int m, int32_t *n1, unsigned long long *stats)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const uint64_t d = rowdesc[2 * row + 1];
    const uint32_t slen = (uint32_t)(d >> kDescLenShift);
    const uint32_t *q4 = reinterpret_cast<const uint32_t*>(rle + (d & kDescOffMask));
    uint32_t pos = 0, total = 0;
    for (uint32_t base = 0; base < slen; base += 256) {
        const uint32_t k0 = base + 4u * (uint32_t)lane;
        const uint32_t w = k0 < slen ? q4[(base >> 2) + lane] : 0u;
        const ChunkDecode cd = decode_chunk(w, k0, slen, lane);
        const uint32_t incl = wave_incl_add(cd.run);
        const uint32_t lane_start = pos + incl - cd.run;
        uint32_t mine = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t st = lane_start + cd.before[i];
            const uint32_t room = st < (uint32_t)m ? (uint32_t)m - st : 0u;
            mine += cd.valid[i] && cd.bit[i] ? (cd.l[i] < room ? cd.l[i] : room) : 0u;
        }
        total += lane63(wave_incl_add(mine));
        pos += lane63(incl);
        if (cd.stop) break;
    }
    if (lane == 0) {
        n1[row] = (int32_t)total;
        atomicMax(stats, (unsigned long long)total);
        atomicAdd(stats + 1, (unsigned long long)total);
    }
}

int sparse_lds_bytes(int m, int tcap, int icap)
{
    const int nwa = ((m + 31) / 32 + 127) & ~127, nwt = tcap / 32;
    return 4 * (nwa + nwa / 8 + 2 * (nwa / 128) + 1 + 4 + nwt + nwt / 8 + 2 * (nwt / 128) + 1 + 8 + 2 * icap) + 64;
}

hipError_t launch_sparse_plane1(const ScanArgs &a, hipStream_t s)
{
    const int lds = sparse_lds_bytes(a.m, a.sp_tcap, a.sp_icap);
    auto fn = sparse_plane1_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(a.n_blk), dim3(kSpNT), lds, s, a, a.rowdesc, a.rle);
    return hipGetLastError();
}

hipError_t launch_plane1_ones(const uint64_t *rowdesc, const uint8_t *rle, int64_t n_rows, int m, int32_t *n1, unsigned long long *stats,
                              hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(plane1_ones_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, rowdesc, rle, n_rows, m, n1, stats);
    return hipGetLastError();
}

}  // namespace bgth
