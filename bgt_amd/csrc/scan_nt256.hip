// Instantiations of the scan kernel for workgroups of 256 threads (one translation unit per workgroup
// size so that `make -j` compiles them in parallel).
#include "scan_device.inc.h"

namespace bgth {

static const int kLdsBytesLocal = 160 * 1024;

template <int NT, int CPT, bool MULTI, bool GT, bool TEAM, bool ZP, bool CC = false, bool SNAP = false>
static hipError_t launch_one(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
    auto fn = scan_kernel<NT, CPT, MULTI, GT, TEAM, ZP, CC, SNAP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
    if (e != hipSuccess) return e;
    (void)kLdsBytesLocal;
    hipLaunchKernelGGL(fn, dim3(g.workgroups), dim3(NT), g.lds_bytes, s, a, a.rowdesc, a.rle, a.chunkinfo, a.segc);
    return hipGetLastError();
}

template <int NT, int CPT, bool ZP>
static hipError_t launch_variant(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
    // (skip1: plane 0 alone, its ballots to h0 -- the counts come from count_planes, whatever the groups)
    const int v = a.skip1 ? 2 : (a.G > 1 ? 4 : 0) | (a.h0 ? 2 : 0) | (g.wpp > 1 ? 1 : 0);
#ifdef BGTH_CC_EXPERIMENT
    // EXPERIMENT builds only (make ccform N=..; profiles/r04_issue/): one group, counts only, every column of the cohort tracked,
    // pipelined narrow mode, no empty-plane shortcut -> the ballot-free instruction-major row step of scan_step_cc.inc.h.  It is
    // bit-exact and SLOWER in the kernel (11.4-12.3 ms against 10.9 ms on C2) although faster in isolation, so it does not ship.
    if constexpr (!ZP) if (v == 0 && a.cc_step && g.nbuf == 2) return launch_one<NT, CPT, false, false, false, false, true>(a, g, s);
#endif
    if (a.snap) {                                                  // the image-open pass (sub-checkpoints): counts only, one group
        if (v == 0) return launch_one<NT, CPT, false, false, false, ZP, false, true>(a, g, s);
        if (v == 1) return launch_one<NT, CPT, false, false, true, ZP, false, true>(a, g, s);
        return hipErrorInvalidConfiguration;
    }
    switch (v) {
    case 0: return launch_one<NT, CPT, false, false, false, ZP>(a, g, s);
    case 1: return launch_one<NT, CPT, false, false, true, ZP>(a, g, s);
    case 2: return launch_one<NT, CPT, false, true, false, ZP>(a, g, s);
    case 3: return launch_one<NT, CPT, false, true, true, ZP>(a, g, s);
    case 4: return launch_one<NT, CPT, true, false, false, ZP>(a, g, s);
    case 5: return launch_one<NT, CPT, true, false, true, ZP>(a, g, s);
    case 6: return launch_one<NT, CPT, true, true, false, ZP>(a, g, s);
    default: return launch_one<NT, CPT, true, true, true, ZP>(a, g, s);
    }
}

hipError_t launch_scan_nt256(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
#define X(CPT_) if (g.cpt == CPT_) return a.zp ? launch_variant<256, CPT_, true>(a, g, s) : launch_variant<256, CPT_, false>(a, g, s);
    BGTH_CPT_256(X)
#undef X
    return hipErrorInvalidConfiguration;
}

}  // namespace bgth
