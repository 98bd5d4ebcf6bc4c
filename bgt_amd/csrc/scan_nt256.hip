// Instantiations of the scan kernel for workgroups of 256 threads (one translation unit per workgroup
// size so that `make -j` compiles them in parallel).
#include "scan_device.inc.h"

namespace bgth {

static const int kLdsBytesLocal = 160 * 1024;

template <int NT, int CPT, bool MULTI, bool GT, bool TEAM, bool ZP, bool SNAP = false, bool WC = false>
static hipError_t launch_one(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
    auto fn = scan_kernel<NT, CPT, MULTI, GT, TEAM, ZP, SNAP, WC>;
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
    const int v = (a.G > 1 ? 4 : 0) | (a.h0 ? 2 : 0) | (g.wpp > 1 ? 1 : 0);
    if (a.snap) {                                                  // the image-open pass (sub-checkpoints): counts only, one group
        if (v == 0) return launch_one<NT, CPT, false, false, false, ZP, true>(a, g, s);
        if (v == 1) return launch_one<NT, CPT, false, false, true, ZP, true>(a, g, s);
        return hipErrorInvalidConfiguration;
    }
    // whole cohort, one group, counts only, pipelined narrow mode, no empty-plane shortcut: n(code 3) alone is counted (WC)
    if constexpr (!ZP) if (v == 0 && a.whole_counts) return launch_one<NT, CPT, false, false, false, false, false, true>(a, g, s);
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
