// pbf_encoder.hip -- the write half of the codec seam on the device: rows of 2-bit codes -> the PBWT run-length
// strings, checkpoints and footer of a .pbf, byte for byte what the reference writer produces
// (pbf_open_w / pbf_write / pbf_close, reference pbwt.c:199-219, :288-311, :264-277; row encoder pbc_enc_core,
// pbwt.c:57-66; run-length bytes pbwt.c:24-36).
//
// The encoder is the mirror image of the scan kernel.  One workgroup per bit plane keeps the RANK of every column
// in registers (the reference moves the permutation array instead).  Per row:
//   scatter    every column drops its bit at its rank into an LDS bit-vector              (a[S[j]] read backwards)
//   directory  ones before every 32-bit word (block scan) and the positions where a run starts
//   emit       every word-thread writes the run-length bytes of the runs that start in its word
//   step       rank <- bit ? zeros + ones_before(rank) : rank - ones_before(rank)         (the stable partition)
// Rows are sequential (row r is written in the order rows 0..r-1 left); planes are independent.  DESIGN.md section 7
// has the plan that removes the sequential pass (block-start orders by a device sort).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/bgt_hip.h"

namespace {

constexpr int kThreads = 1024;                 // one word of the bit-vector per thread: m <= 32768
constexpr int kMaxM = kThreads * 32;

struct EncodeArgs {
    const uint8_t *codes;      // [n_rows][CPT * 1024]  bit k of a byte = plane k, zero beyond column m
    int64_t n_rows, row0;      // row0 = file row of codes[0]
    int32_t m, mask;           // mask = (1 << shift) - 1
    int32_t *rank;             // [g][m] in/out
    uint8_t *out;              // [g][cap]
    int64_t cap;
    int64_t *out_len;          // [g]
    int32_t *row_len;          // [g][n_rows]
    int32_t *snap;             // [g][n_snap][m]   permutation before every row with (row & mask) == 0
    int32_t n_snap;
    int32_t *status;           // != 0: output capacity exceeded
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
    for (int digit = 7; digit >= 0; --digit) {
        const uint32_t d = (len >> (4 * digit)) & 15u;
        if (d) dst[n++] = (uint8_t)((((uint32_t)digit << 4) | d) << 1 | bit);
    }
    return n;
}

// CPT columns per thread (a multiple of 4); rows of `codes` are CPT * 1024 bytes apart, zero beyond column m, so a
// thread fetches its columns as whole words and the padding columns never set a bit.
template <int CPT>
__global__ __launch_bounds__(kThreads) void encode_kernel(EncodeArgs a)
{
    __shared__ uint32_t bits2[2][kThreads + 1]; // the row's bit-vector in PBWT order, double-buffered over rows
    __shared__ uint32_t before[kThreads];       // ones before each word
    __shared__ uint32_t agg[3][16];             // per wave: ones, end of the last run that ends in it, bytes
    const int tid = threadIdx.x, lane = tid & 63, plane = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, nw = (m + 31) >> 5;
    constexpr int stride = CPT * kThreads;
    int32_t *rank = a.rank + (size_t)plane * m;
    uint8_t *out = a.out + (size_t)plane * a.cap;
    const int col0 = tid * CPT;                 // this thread's columns: col0 .. col0 + CPT - 1
    int32_t R[CPT];
    uint32_t nxt[CPT / 4];
#pragma unroll
    for (int i = 0; i < CPT; ++i) R[i] = col0 + i < m ? rank[col0 + i] : 0;
#pragma unroll
    for (int q = 0; q < CPT / 4; ++q) nxt[q] = a.n_rows > 0 ? reinterpret_cast<const uint32_t*>(a.codes + col0)[q] : 0u;
    bits2[0][tid] = 0u; bits2[1][tid] = 0u;
    if (tid == 0) { bits2[0][kThreads] = 0u; bits2[1][kThreads] = 0u; }
    __syncthreads();
    const uint32_t valid = tid < nw ? ((tid == nw - 1 && (m & 31)) ? (1u << (m & 31)) - 1u : 0xffffffffu) : 0u;
    const uint32_t last_bit = (tid == nw - 1) ? 1u << ((m - 1) & 31) : 0u;      // the row ends here
    int64_t off = 0;
    int snap_i = 0;
    for (int64_t r = 0; r < a.n_rows; ++r) {
        uint32_t *bits = bits2[r & 1];
        if (((a.row0 + r) & a.mask) == 0) {                 // the 'S' record: S[rank] = column (ref pbwt.c:292-301)
            int32_t *S = a.snap + ((size_t)plane * a.n_snap + snap_i) * m;
#pragma unroll
            for (int i = 0; i < CPT; ++i) if (col0 + i < m) S[R[i]] = col0 + i;
            ++snap_i;
        }
        // ---- scatter this row, fetch the next one
        uint32_t mine = 0;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const uint32_t b = (nxt[i >> 2] >> (8 * (i & 3) + plane)) & 1u;
            mine |= b << i;
            if (b) atomicOr(&bits[R[i] >> 5], 1u << (R[i] & 31));
        }
        if (r + 1 < a.n_rows) {
            const uint32_t *src = reinterpret_cast<const uint32_t*>(a.codes + (size_t)(r + 1) * stride + col0);
#pragma unroll
            for (int q = 0; q < CPT / 4; ++q) nxt[q] = src[q];
        }
        __syncthreads();                                    // (1) the bit-vector is complete
        bits2[(r & 1) ^ 1][tid] = 0u;                       // everyone is done with the previous row's
        const uint32_t w = bits[tid];
        const uint32_t wn = bits[tid + 1];
        // a run ends at bit i of this word if the next position holds the other bit, or the row ends there
        const uint32_t ends = (((w ^ (w >> 1 | wn << 31)) & valid) & ~last_bit) | last_bit;
        const uint32_t pc = (uint32_t)__popc(w);
        const uint32_t incl = wave_incl_add(pc);
        const uint32_t le = ends ? (uint32_t)(tid * 32 + 32 - __builtin_clz(ends)) : 0u;   // position after the last end
        const uint32_t lmax = wave_incl_max(le);
        if (lane == 63) { agg[0][wave] = incl; agg[1][wave] = lmax; }
        __syncthreads();                                    // (2)
        uint32_t ones, wbase, start;
        {
            const uint32_t va = lane < 16 ? agg[0][lane] : 0u, vm = lane < 16 ? agg[1][lane] : 0u;
            const uint32_t sa = wave_incl_add(va), sx = wave_incl_max(vm);
            ones = lane_value(sa, 15);
            wbase = wave ? lane_value(sa, wave - 1) : 0u;
            start = umax(wave ? lane_value(sx, wave - 1) : 0u, wave_shr1(lmax));   // where the first run ending here began
        }
        before[tid] = wbase + incl - pc;
        uint32_t nb = 0;
        {
            uint32_t st = start;
            for (uint32_t x = ends; x;) {                   // the runs that end in this word
                const uint32_t e = (uint32_t)(tid * 32 + __builtin_ctz(x) + 1);
                x &= x - 1u;
                nb += run_bytes(e - st);
                st = e;
            }
        }
        const uint32_t incl2 = wave_incl_add(nb);
        if (lane == 63) agg[2][wave] = incl2;
        __syncthreads();                                    // (3) `before` and the byte counts are visible
        uint32_t bbase, total;
        {
            const uint32_t vb = lane < 16 ? agg[2][lane] : 0u;
            const uint32_t sb = wave_incl_add(vb);
            total = lane_value(sb, 15);
            bbase = wave ? lane_value(sb, wave - 1) : 0u;
        }
        if (off + (int64_t)total > a.cap) { if (tid == 0) *a.status = 1; break; }          // uniform
        {
            uint8_t *dst = out + off + bbase + incl2 - nb;
            uint32_t st = start;
            for (uint32_t x = ends; x;) {
                const uint32_t i = (uint32_t)__builtin_ctz(x);
                const uint32_t e = (uint32_t)(tid * 32) + i + 1u;
                x &= x - 1u;
                dst += put_run(dst, e - st, (w >> i) & 1u);
                st = e;
            }
        }
        if (tid == 0) a.row_len[(size_t)plane * a.n_rows + r] = (int32_t)total;
        off += total;
        // ---- the stable partition, on ranks (ref pbwt.c:57-66 moves S instead)
        const int32_t n0 = m - (int32_t)ones;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int wd = R[i] >> 5;
            const int32_t r1 = (int32_t)(before[wd] + (uint32_t)__popc(bits[wd] & ((1u << (R[i] & 31)) - 1u)));
            R[i] = ((mine >> i) & 1u) ? n0 + r1 : R[i] - r1;
        }
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i) if (col0 + i < m) rank[col0 + i] = R[i];
    if (tid == 0) a.out_len[plane] = off;
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
    int64_t batch_rows = 0, cap = 0;
    int32_t cpt = 0, stride = 0;                     // device rows of codes are stride = cpt * 1024 bytes apart
    int32_t max_snap = 0;
    hipStream_t stream = nullptr;
    uint8_t *d_codes = nullptr, *d_out = nullptr;
    int32_t *d_rank = nullptr, *d_row_len = nullptr, *d_snap = nullptr, *d_status = nullptr;
    int64_t *d_out_len = nullptr;
    std::vector<uint8_t> image;
    std::vector<uint64_t> idx;
    std::vector<uint8_t> h_out;
    std::vector<int32_t> h_row_len, h_snap;
    double kernel_ms = 0.0;
};

extern "C" const char *bgth_encoder_last_error(void) { return g_enc_err; }

extern "C" void bgth_encoder_close(bgth_encoder_t *e)
{
    if (!e) return;
    hipSetDevice(e->device);
    hipFree(e->d_codes); hipFree(e->d_out); hipFree(e->d_rank); hipFree(e->d_row_len); hipFree(e->d_snap);
    hipFree(e->d_status); hipFree(e->d_out_len);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

extern "C" bgth_encoder_t *bgth_encoder_open(int32_t m, int32_t g, int32_t shift, int device)
{
    if (m < 1 || m > kMaxM) { enc_err("[E::%s] %d columns: this encoder holds 1..%d", __func__, m, kMaxM); return nullptr; }
    if (g < 1 || g > 8 || shift < 0 || shift > 30) { enc_err("[E::%s] bad plane count or checkpoint shift", __func__); return nullptr; }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
        enc_err("[E::%s] no usable HIP device %d (the encoder has no CPU path)", __func__, device);
        return nullptr;
    }
    bgth_encoder_t *e = new bgth_encoder_t;
    e->m = m; e->g = g; e->shift = shift; e->device = device;
    // a batch: at most 8192 rows and about 256 MB of codes; a row of m bits never needs more than m bytes
    e->cpt = m <= 4 * kThreads ? 4 : m <= 8 * kThreads ? 8 : m <= 20 * kThreads ? 20 : 32;
    e->stride = e->cpt * kThreads;
    e->batch_rows = (int64_t)(256 << 20) / e->stride;
    if (e->batch_rows > 8192) e->batch_rows = 8192;
    if (e->batch_rows < 16) e->batch_rows = 16;
    e->cap = e->batch_rows * (int64_t)m;
    e->max_snap = (int32_t)((e->batch_rows >> shift) + 2);
    ENC_TRY(hipSetDevice(device), { delete e; return nullptr; });
    ENC_TRY(hipStreamCreate(&e->stream), { delete e; return nullptr; });
    ENC_TRY(hipMalloc(&e->d_codes, (size_t)e->batch_rows * e->stride), { bgth_encoder_close(e); return nullptr; });
    ENC_TRY(hipMemset(e->d_codes, 0, (size_t)e->batch_rows * e->stride), { bgth_encoder_close(e); return nullptr; });   // the padding stays 0
    ENC_TRY(hipMalloc(&e->d_out, (size_t)g * e->cap), { bgth_encoder_close(e); return nullptr; });
    ENC_TRY(hipMalloc(&e->d_rank, (size_t)g * m * 4), { bgth_encoder_close(e); return nullptr; });
    ENC_TRY(hipMalloc(&e->d_row_len, (size_t)g * e->batch_rows * 4), { bgth_encoder_close(e); return nullptr; });
    ENC_TRY(hipMalloc(&e->d_snap, (size_t)g * e->max_snap * m * 4), { bgth_encoder_close(e); return nullptr; });
    ENC_TRY(hipMalloc(&e->d_status, 4), { bgth_encoder_close(e); return nullptr; });
    ENC_TRY(hipMalloc(&e->d_out_len, (size_t)g * 8), { bgth_encoder_close(e); return nullptr; });
    std::vector<int32_t> ident((size_t)g * m);
    for (int k = 0; k < g; ++k) for (int j = 0; j < m; ++j) ident[(size_t)k * m + j] = j;     // identity start (ref pbwt.c:92-105)
    ENC_TRY(hipMemcpy(e->d_rank, ident.data(), ident.size() * 4, hipMemcpyHostToDevice), { bgth_encoder_close(e); return nullptr; });
    e->image.reserve(1 << 20);
    const int32_t hdr[3] = {m, g, shift};                                                       // ref pbwt.c:199-219
    e->image.insert(e->image.end(), (const uint8_t*)"PBF\1", (const uint8_t*)"PBF\1" + 4);
    e->image.insert(e->image.end(), (const uint8_t*)hdr, (const uint8_t*)hdr + 12);
    return e;
}

static int encode_batch(bgth_encoder_t *e, const uint8_t *codes, int64_t rows)
{
    const int m = e->m, g = e->g;
    const int64_t mask = ((int64_t)1 << e->shift) - 1;
    int32_t n_snap = 0;
    for (int64_t r = 0; r < rows; ++r) if (((e->n + r) & mask) == 0) ++n_snap;
    EncodeArgs a;
    a.codes = e->d_codes; a.n_rows = rows; a.row0 = e->n; a.m = m; a.mask = (int32_t)mask;
    a.rank = e->d_rank; a.out = e->d_out; a.cap = e->cap; a.out_len = e->d_out_len; a.row_len = e->d_row_len;
    a.snap = e->d_snap; a.n_snap = n_snap; a.status = e->d_status;
    ENC_TRY(hipMemcpy2DAsync(e->d_codes, (size_t)e->stride, codes, (size_t)m, (size_t)m, (size_t)rows, hipMemcpyHostToDevice, e->stream), return -1);
    ENC_TRY(hipMemsetAsync(e->d_status, 0, 4, e->stream), return -1);
    hipEvent_t ev0, ev1;
    ENC_TRY(hipEventCreate(&ev0), return -1);
    ENC_TRY(hipEventCreate(&ev1), return -1);
    hipEventRecord(ev0, e->stream);
    if (e->cpt == 4)       hipLaunchKernelGGL(encode_kernel<4>,  dim3(g), dim3(kThreads), 0, e->stream, a);
    else if (e->cpt == 8)  hipLaunchKernelGGL(encode_kernel<8>,  dim3(g), dim3(kThreads), 0, e->stream, a);
    else if (e->cpt == 20) hipLaunchKernelGGL(encode_kernel<20>, dim3(g), dim3(kThreads), 0, e->stream, a);
    else                   hipLaunchKernelGGL(encode_kernel<32>, dim3(g), dim3(kThreads), 0, e->stream, a);
    hipEventRecord(ev1, e->stream);
    ENC_TRY(hipGetLastError(), return -1);
    int32_t status = 0;
    std::vector<int64_t> out_len((size_t)g);
    ENC_TRY(hipMemcpyAsync(&status, e->d_status, 4, hipMemcpyDeviceToHost, e->stream), return -1);
    ENC_TRY(hipMemcpyAsync(out_len.data(), e->d_out_len, (size_t)g * 8, hipMemcpyDeviceToHost, e->stream), return -1);
    e->h_row_len.resize((size_t)g * rows);
    for (int k = 0; k < g; ++k)
        ENC_TRY(hipMemcpyAsync(e->h_row_len.data() + (size_t)k * rows, e->d_row_len + (size_t)k * rows, (size_t)rows * 4,
                               hipMemcpyDeviceToHost, e->stream), return -1);
    e->h_snap.resize((size_t)g * n_snap * m);
    if (n_snap) ENC_TRY(hipMemcpyAsync(e->h_snap.data(), e->d_snap, e->h_snap.size() * 4, hipMemcpyDeviceToHost, e->stream), return -1);
    ENC_TRY(hipStreamSynchronize(e->stream), return -1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev0, ev1);
    e->kernel_ms += ms;
    hipEventDestroy(ev0); hipEventDestroy(ev1);
    if (status != 0) { enc_err("[E::%s] run-length output exceeded its buffer", __func__); return -1; }
    int64_t worst = 0;
    for (int k = 0; k < g; ++k) if (out_len[k] > worst) worst = out_len[k];
    e->h_out.resize((size_t)g * worst);
    for (int k = 0; k < g; ++k)
        if (out_len[k]) ENC_TRY(hipMemcpy(e->h_out.data() + (size_t)k * worst, e->d_out + (size_t)k * e->cap, (size_t)out_len[k],
                                          hipMemcpyDeviceToHost), return -1);
    // ---- records in file order (ref pbwt.c:288-311): ['S' perms]  'B' { int32 len, bytes } per plane
    std::vector<int64_t> at((size_t)g, 0);
    int32_t si = 0;
    for (int64_t r = 0; r < rows; ++r) {
        if (((e->n + r) & mask) == 0) {
            e->idx.push_back((uint64_t)e->image.size());
            e->image.push_back('S');
            for (int k = 0; k < g; ++k) {
                const uint8_t *p = (const uint8_t*)(e->h_snap.data() + ((size_t)k * n_snap + si) * m);
                e->image.insert(e->image.end(), p, p + (size_t)m * 4);
            }
            ++si;
        }
        e->image.push_back('B');
        for (int k = 0; k < g; ++k) {
            const int32_t l = e->h_row_len[(size_t)k * rows + r];
            const uint8_t *p = e->h_out.data() + (size_t)k * worst + at[k];
            e->image.insert(e->image.end(), (const uint8_t*)&l, (const uint8_t*)&l + 4);
            e->image.insert(e->image.end(), p, p + l);
            at[k] += l;
        }
    }
    e->n += rows;
    return 0;
}

extern "C" int bgth_encoder_write(bgth_encoder_t *e, const uint8_t *codes, int64_t n_rows)
{
    if (!e || (!codes && n_rows > 0) || n_rows < 0) { enc_err("[E::%s] bad argument", __func__); return -1; }
    ENC_TRY(hipSetDevice(e->device), return -1);
    for (int64_t done = 0; done < n_rows;) {
        int64_t rows = n_rows - done;
        if (rows > e->batch_rows) rows = e->batch_rows;
        if (encode_batch(e, codes + (size_t)done * e->m, rows) < 0) return -1;
        done += rows;
    }
    return 0;
}

extern "C" int64_t bgth_encoder_finish(bgth_encoder_t *e, uint8_t **image)
{
    if (!e || !image) { enc_err("[E::%s] bad argument", __func__); return -1; }
    const uint64_t off = (uint64_t)e->image.size();                                             // ref pbwt.c:264-277
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

extern "C" void bgth_encoder_free_image(uint8_t *image) { free(image); }
extern "C" double bgth_encoder_kernel_ms(const bgth_encoder_t *e) { return e ? e->kernel_ms : 0.0; }
