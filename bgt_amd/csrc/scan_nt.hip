// Instantiations of the scan kernel for ONE launch geometry: compiled once per (threads, columns per thread) of BGTH_CPT_256 / _512 /
// _1024 with -DBGTH_NT=<threads> -DBGTH_CPT=<columns> (Makefile) -- `make -j` compiles them side by side, and a process loads the
// code object of the geometry it launches (HIP loads a translation unit's kernels at the first launch of one of them: as one
// unit per workgroup size the 1024-thread kernels were 7.4 MB, ~11 ms per MB to load).
#include "scan_device.inc.h"

namespace bgth {

static const int kLdsBytesLocal = 160 * 1024;

template <int NT, int CPT, bool MULTI, bool GT, bool TEAM, bool ZP, bool SNAP = false, bool WC = false, bool PK = false>
static hipError_t launch_one(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
    auto fn = scan_kernel<NT, CPT, MULTI, GT, TEAM, ZP, SNAP, WC, PK>;
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
#ifdef BGTH_ABLATE      // (profiling build: the packed-rank statement of the round-6 A/B, profiles/r06_pk16 -- slower, not in the product)
    if constexpr (!ZP && CPT % 4 == 0) if (v == 0 && a.whole_counts && a.pk16) return launch_one<NT, CPT, false, false, false, false, false, true, true>(a, g, s);
#endif
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

#ifndef BGTH_NT
#error "compile with -DBGTH_NT=<threads> -DBGTH_CPT=<columns per thread>"
#endif
#define BGTH_CAT4(a, b, c, d) a##b##c##d
#define BGTH_LAUNCH_NAME(nt, cpt) BGTH_CAT4(launch_scan_nt, nt, _c, cpt)
hipError_t BGTH_LAUNCH_NAME(BGTH_NT, BGTH_CPT)(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
    return a.zp ? launch_variant<BGTH_NT, BGTH_CPT, true>(a, g, s) : launch_variant<BGTH_NT, BGTH_CPT, false>(a, g, s);
}

}  // namespace bgth
