// How fast does an MI355X take 16-byte stores in the two shapes the directory path's producer could use?  (GPU box:
// hipcc --offload-arch=gfx950 -O3 scripts/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern)
//   pattern 0: a wave instruction writes 64 x 16 B CONTIGUOUS (1 KB); the second instruction the next 1 KB
//   pattern 1: lane l writes 16 B at 32 l and, with the second instruction, 16 B at 32 l + 16 (what directory_trips_tog does:
//              a lane owns four consecutive 8-byte entries): each instruction half-fills sixteen 128-byte lines
//   pattern 2: as 1 with non-temporal stores;  pattern 3: as 0 with non-temporal stores
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int P>
__global__ __launch_bounds__(256) void store_kernel(u32x4 *dst, size_t n_blocks2k, unsigned v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * 4;
    for (size_t b = (size_t)blockIdx.x * 4 + wave; b < n_blocks2k; b += stride) {   // one 2 KB block per wave and trip
        u32x4 *p = dst + b * 128;
        const u32x4 x = {v, (unsigned)b, (unsigned)lane, 1u}, y = {v, (unsigned)b, (unsigned)lane, 2u};
        if (P == 0) { p[lane] = x; p[64 + lane] = y; }
        else if (P == 1) { p[2 * lane] = x; p[2 * lane + 1] = y; }
        else if (P == 2) { __builtin_nontemporal_store(x, p + 2 * lane); __builtin_nontemporal_store(y, p + 2 * lane + 1); }
        else { __builtin_nontemporal_store(x, p + lane); __builtin_nontemporal_store(y, p + 64 + lane); }
    }
}
int main()
{
    const size_t bytes = (size_t)8 << 30, nb = bytes / 2048;
    u32x4 *d = nullptr;
    if (hipMalloc((void**)&d, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192}) for (int pat = 0; pat < 4; ++pat) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, nullptr);
            if (pat == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(256), 0, nullptr, d, nb, (unsigned)rep);
            if (pat == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(256), 0, nullptr, d, nb, (unsigned)rep);
            if (pat == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(grid), dim3(256), 0, nullptr, d, nb, (unsigned)rep);
            if (pat == 3) hipLaunchKernelGGL(store_kernel<3>, dim3(grid), dim3(256), 0, nullptr, d, nb, (unsigned)rep);
            hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("grid %5d pattern %d: %.3f ms  %.2f TB/s\n", grid, pat, best, bytes / best / 1e9);
    }
    return 0;
}
