// Does s_setreg_b32 on HW_REG_STATUS.USER_PRIO (bits 4:3) take a wave priority from an SGPR (s_setprio only takes an immediate)?
// Answer (MI355X, 2026-09-30): no -- "after s_setprio 2: 2; after s_setreg 1, 3, 0: 2 2 2": the field is read-only to s_setreg, a wave-dependent
// priority needs a branch per s_setprio.
// Prints what s_getreg reads back after s_setprio 2, then after s_setreg with 1, 3, 0.   hipcc --offload-arch=gfx950 -o /tmp/p scripts/setreg_prio_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out, unsigned v1, unsigned v2, unsigned v3)
{
    unsigned a, b, c, d;
    asm volatile("s_setprio 2\n\ts_nop 4\n\ts_getreg_b32 %0, hwreg(HW_REG_STATUS, 3, 2)" : "=s"(a));
    asm volatile("s_setreg_b32 hwreg(HW_REG_STATUS, 3, 2), %1\n\ts_nop 4\n\ts_getreg_b32 %0, hwreg(HW_REG_STATUS, 3, 2)" : "=s"(b) : "s"(v1));
    asm volatile("s_setreg_b32 hwreg(HW_REG_STATUS, 3, 2), %1\n\ts_nop 4\n\ts_getreg_b32 %0, hwreg(HW_REG_STATUS, 3, 2)" : "=s"(c) : "s"(v2));
    asm volatile("s_setreg_b32 hwreg(HW_REG_STATUS, 3, 2), %1\n\ts_nop 4\n\ts_getreg_b32 %0, hwreg(HW_REG_STATUS, 3, 2)" : "=s"(d) : "s"(v3));
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; out[2] = c; out[3] = d; }
}
int main()
{
    unsigned *d, h[4];
    hipMalloc(&d, 16);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 1u, 3u, 0u);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("after s_setprio 2: %u; after s_setreg 1, 3, 0: %u %u %u\n", h[0], h[1], h[2], h[3]);
    return 0;
}
