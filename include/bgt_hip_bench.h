/* libbgt_hip_bench.so -- measurement tools that are NOT part of the product library: the issue-rate calibration
 * kernels behind the roofline that bench.py reports (bgt_amd/csrc/microbench.hip).  No reference interface is replaced
 * by these; they exist so that `peak` in the bench line is measured on the chip it ran on. */
#ifndef BGT_HIP_BENCH_H
#define BGT_HIP_BENCH_H
#ifdef __cplusplus
extern "C" {
#endif
/* Issue-rate calibration for the roofline of the scan kernel (which is bound by VALU issue and LDS gathers, not by
 * HBM): runs `iters` iterations of instruction mix `mix` (bgth_debug_issue_rate_name(mix) describes it; NULL past the
 * last one) on every CU with `waves_per_simd` (1..4) waves per SIMD.  out[0] = shader cycles of the slowest wave,
 * out[1] = milliseconds of the launch, out[2] / out[3] = VALU / LDS wave-instructions issued per wave. */
int         bgth_debug_issue_rate(int device, int mix, int waves_per_simd, int iters, double out[4]);
const char *bgth_debug_issue_rate_name(int mix);
/* The same for single VALU opcodes (instruction classes: which issue in 2 cycles per wave64, which in 4 or more):
 * out[0] = cycles, out[1] = ms, out[2] = instructions per wave. */
int         bgth_debug_op_rate(int device, int op, int waves_per_simd, int iters, double out[3]);
const char *bgth_debug_op_rate_name(int op);

#ifdef __cplusplus
}
#endif
#endif
