// libbgt_hip.so -- C ABI (include/bgt_hip.h) over the gfx950 kernels of scan_kernels.hip.
// Host side only: .pbf parsing/packing, HBM residency, selection tables, launches, result staging.
// There is deliberately no CPU decode path in this file: every genotype comes out of the HIP kernels.
#include "../../include/bgt_hip.h"
#include "scan_kernels.h"

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

using namespace bgth;

// ----------------------------------------------------------------------------------------------------
// errors
// ----------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr, onfail)                                                            \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            set_err("[E::%s] %s: %s", __func__, #expr, hipGetErrorString(e_));           \
            onfail;                                                                      \
        }                                                                                \
    } while (0)

extern "C" const char *bgth_last_error(void) { return g_err; }

// Device memory of a process that ended a moment ago (the child `bgt view` leaves behind, view_cli.c: work_in_a_child) comes
// back asynchronously: an allocation that fails for lack of memory is tried again for up to BGTH_OOM_WAIT_MS (default 300 ms,
// 0 = fail at once) before it is an error.  Every hipMalloc of this file goes through here.
static hipError_t dev_malloc_retry(void **p, size_t n)
{
    hipError_t e = (hipMalloc)(p, n);
    if (e != hipErrorOutOfMemory) return e;
    static const int wait_ms = [] { const char *v = getenv("BGTH_OOM_WAIT_MS"); const int w = v ? atoi(v) : 300; return w < 0 ? 0 : (w > 10000 ? 10000 : w); }();
    for (int waited = 0; waited < wait_ms && e == hipErrorOutOfMemory; waited += 25) {
        (void)hipGetLastError();                                         // (the failed call's sticky status)
        std::this_thread::sleep_for(std::chrono::milliseconds(25));
        e = (hipMalloc)(p, n);
    }
    return e;
}
#define hipMalloc(p, n) dev_malloc_retry((void**)(p), (n))

// Nothing C++ may cross the C ABI: entry points that build host-side vectors run under this guard.  An image under
// construction is registered in t_building so that an exception does not leak its device memory.
static thread_local bgth_pbf_t *t_building = nullptr;
template <class F, class R>
static R guarded(const char *who, R on_fail, F body)
{
    try { return body(); }
    catch (const std::bad_alloc &) { set_err("[E::%s] out of host memory", who); }
    catch (const std::exception &e) { set_err("[E::%s] %s", who, e.what()); }
    if (t_building) { bgth_pbf_t *p = t_building; t_building = nullptr; bgth_pbf_close(p); }
    return on_fail;
}
extern "C" const char *bgth_version(void) { return "bgt-hip 0.1 (gfx950)"; }

// The HIP runtime takes 60-220 ms to start and the first launch loads the code objects: a process that knows it will
// need the device starts both on a thread of their own while it parses headers, sample tables and the site side-car.
// (Whoever touches the device first simply waits on the runtime's own initialisation lock.)
static std::mutex g_warm_lock;
static std::atomic<bool> g_runtime_ready{false};   // a HIP call has returned in this process: the runtime is up (nobody waits for it any more)
static std::thread *g_warm_thread = nullptr;    // (on the heap and never destroyed: a caller that does not wait must not die in a destructor)
extern "C" void bgth_runtime_warmup_async(int device)
{
    std::lock_guard<std::mutex> g(g_warm_lock);
    if (g_warm_thread) return;                                           // one warm-up per process
    try {
        g_warm_thread = new std::thread([device] {
            if (hipSetDevice(device) != hipSuccess) return;
            hipFree(nullptr);
            int32_t *d = nullptr;                                                           // first launch: code object load
            if (hipMalloc((void**)&d, 64) == hipSuccess) {
                hipMemset(d, 0, 64);
                launch_finalize(d, d + 8, d + 4, 1, 1, nullptr);
                hipDeviceSynchronize();
                hipFree(d);
            }
            g_runtime_ready = true;
        });
    } catch (...) {}
}
// A process that leaves before it ever touched the device (a usage error found after the warm-up was started) must not
// run its exit handlers under a runtime that is still coming up on another thread: it waits here first.
extern "C" void bgth_runtime_warmup_wait(void)
{
    std::lock_guard<std::mutex> g(g_warm_lock);
    if (g_warm_thread && g_warm_thread->joinable()) g_warm_thread->join();
}

extern "C" int bgth_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { set_err("[E::bgth_device_count] %s", hipGetErrorString(e)); return -1; }
    return n;
}

// ----------------------------------------------------------------------------------------------------
// RCCL (the gather of per-shard counts over xGMI).  The library is bound at first use, not at link time: a
// process that has torch loaded already holds torch's own copy of librccl, and a second RCCL runtime in the process is
// the same trap as a second HIP runtime (bgt_amd/hip.py) -- so a copy that is already mapped is preferred.
// ----------------------------------------------------------------------------------------------------
struct Rccl {
    void *h = nullptr;
    int (*CommInitAll)(void **comm, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) = nullptr;
    int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t s) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
static Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (int pass = 0; pass < 2 && !r.h; ++pass)
            for (const char *n : names) {
                r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (r.h) break;
            }
        if (!r.h) r.h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.h) return;
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.h, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
        r.Send = (decltype(r.Send))dlsym(r.h, "ncclSend");
        r.Recv = (decltype(r.Recv))dlsym(r.h, "ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.h, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.h, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.Send && r.Recv && r.GroupStart && r.GroupEnd && r.GetErrorString;
    });
    return r;
}
#define RCCL_TRY(expr, onfail)                                                           \
    do {                                                                                 \
        const int e_ = (expr);                                                           \
        if (e_ != 0) { set_err("[E::%s] %s: %s", __func__, #expr, rccl().GetErrorString(e_)); onfail; } \
    } while (0)

// ----------------------------------------------------------------------------------------------------
// objects
// ----------------------------------------------------------------------------------------------------
struct Selection {
    int width = 0, n_chunks = 0, G = 1;
    bool whole = false;                // every column of the cohort exactly once (no subset list)
    std::vector<int32_t> slot_col, slot_of_out, group_haps;
    std::vector<uint32_t> chunk_desc;
    int32_t *d_slot_col = nullptr, *d_slot_of_out = nullptr, *d_group_haps = nullptr;
    uint32_t *d_chunk_desc = nullptr;
    size_t cap[4] = {0, 0, 0, 0};      // bytes behind the four device tables: a reader that is selected again (a pooled
                                       // reader serves query after query) allocates only when a table grows
    void release()
    {
        if (d_slot_col) hipFree(d_slot_col);
        if (d_slot_of_out) hipFree(d_slot_of_out);
        if (d_group_haps) hipFree(d_group_haps);
        if (d_chunk_desc) hipFree(d_chunk_desc);
        d_slot_col = d_slot_of_out = d_group_haps = nullptr;
        d_chunk_desc = nullptr;
        cap[0] = cap[1] = cap[2] = cap[3] = 0;
    }
};

struct bgth_pbf_s {
    int device = 0;
    int32_t m = 0, g = 0, shift = 0;
    int32_t g_file = 0;               // planes of the file when that is 1 (prefix.pb1, `bgt import -1`): the image then carries an
                                      // empty second plane, see expand_one_plane(); 0 = as g
    int32_t sub_shift = 0;            // sub-checkpoints every 1 << sub_shift rows (<= shift), see derive_sub_checkpoints
    bool mem_plane = false;           // m > 650,000: not even one of them does -- toggles and entries in memory (the *_mem kernels)
    bool wide_plane = false;          // 327,000 < m: a row's two bit-vectors do not fit the LDS together; every scan
                                      // takes the directory path with one plane per workgroup (scan_plane.hip), no sub-checkpoints
    bool one_shot = false;            // opened with BGTH_OPEN_HINT=walk: no sub-checkpoints, arena passes of one round of workgroups
    int64_t n = 0, n_blk = 0;         // n_blk: file blocks of 1 << shift rows
    int64_t n_sub = 0;                // sub-blocks of 1 << sub_shift rows: the unit the kernels work on
    int arena_share = 1;              // shards of one sharded image on this device (the directory arena is cap / arena_share)
    // a partial image (bgth_pbf_open_rows) holds the file blocks that cover a row range: every internal index is
    // relative to row_off (a multiple of 1 << shift), the C ABI speaks file rows
    int64_t row_off = 0, n_total = 0;
    int64_t n_empty1 = 0;             // rows whose plane 1 is all zero (no run of ones in its string): see use_zp()
    int64_t rle_bytes = 0;            // RLE payload as in the file
    int64_t packed_bytes = 0;         // payload + padding of every string to 4 bytes
    uint8_t  *d_rle = nullptr;
    uint64_t *d_rowdesc = nullptr;
    int32_t  *d_rank0 = nullptr;      // [n_sub][2][m] ranks by column at every (sub-)checkpoint
    int64_t   rank_epoch = 0;         // counts the times d_rank0 was replaced (rebase): readers key their compact start-rank tables on it
    // [n_sub][m]: the plane-1 ranks of every (sub-)checkpoint in the order of its plane-0 ranks -- what a whole-cohort, one-group,
    // counts-only scan starts from (slots in plane-0 rank order: profiles/r05_lds); built at the first such scan, dropped when the
    // checkpoints change (rebase); order_failed: no HBM for it, the scans use the column order
    int32_t  *d_order = nullptr;
    bool      order_failed = false;
    int32_t  *d_final = nullptr;      // [2][m] ranks by column after the last row (images built by bgth_pbf_from_rle only)
    // row index (scan_kernels.h), built on the first wide-cohort (team-mode) launch
    uint32_t *d_chunkinfo = nullptr, *d_segc = nullptr;
    int32_t   S8 = 0;
    int64_t   rowindex_bytes = 0;
    std::mutex rowindex_lock;         // an image is shared by readers on different threads
    // A SHARDED image (bgth_pbf_open_sharded): the file's blocks dealt out as contiguous block ranges, one partial image
    // per shard, each on its own device.  The parent holds no device data; n = n_total, row_off = 0.
    std::vector<bgth_pbf_t*> shards;
    std::vector<uint8_t> file_image;  // ... and the file's own bytes, which bgth_pbf_save writes back (such an image cannot be re-based)
    std::vector<bgth_pbf_t*> pairs;   // a file of MORE than two bit planes (pbwt.c:211-213, 325-334 loop over any g): the planes are independent
                                      // PBWTs, held as images of planes (0,1), (2,3), ... -- the codec interface (select / seek / read) only
    // RCCL communicator over the DISTINCT devices of the shards (bgth_reader_scan_device on a sharded image gathers the
    // per-shard counts on shard 0's device): comm[i] belongs to comm_dev[i]; built at the first gather
    std::vector<void*> comm;
    std::vector<int> comm_dev;
    std::mutex comm_lock;
    // Readers given back by bgth_reader_destroy wait here (stream, events, device and pinned buffers intact) for the next
    // bgth_reader_create on this image: a resident process answers query after query without re-allocating any of it.
    std::mutex pool_lock;
    std::vector<bgth_reader_t*> pool;
};

// Kernel families that all give the SAME results can be forced (bgth_force_kernels, include/bgt_hip.h: BGTH_FORCE_*) so that tests
// run every family on shapes the CPU oracle decodes quickly; 0 = automatic.  The shipped library reads no environment variable
// for this; the profiling build (make ABLATE=1) also honours BGTH_VARIANT=<bits> and three experiment bits of its own.
enum { kVariantNoTog = BGTH_FORCE_NO_TOGGLE_ARRAY, kVariantNeverZP = BGTH_FORCE_NO_EMPTY_PLANE_SHORTCUT,
       kVariantAlwaysZP = BGTH_FORCE_EMPTY_PLANE_SHORTCUT,
       kVariantDirAlways = BGTH_FORCE_DIRECTORY_PATH, kVariantDirNever = BGTH_FORCE_NO_DIRECTORY_PATH,
       kVariantDirNoReuse = BGTH_FORCE_REBUILD_ROWS,
       kVariantSeqCheckpoints = BGTH_FORCE_SEQUENTIAL_CHECKPOINTS, kVariantRcclSelf = BGTH_FORCE_RCCL_TO_SELF,
       kVariantPlaneNever = BGTH_FORCE_NO_PLANE_SPLIT, kVariantPlaneAlways = BGTH_FORCE_PLANE_SPLIT,
       kVariantColumnOrder = BGTH_FORCE_COLUMN_ORDER,
       kVariantThreeBuffers = BGTH_FORCE_THREE_PLANE_BUFFERS,
       // profiling build only: no window prefetch in the pull interface | no L2 warming of the next row's plane 1 | no
       // progress-based wave priorities in the walk | whole-cohort counts with a column's two ranks packed in one register (the
       // round-6 A/B that lost: profiles/r06_pk16)
       kVariantNoPrefetch = 16, kVariantDirNoWarm = 256, kVariantNoWalkPrio = 16384, kVariantPackedRanks = 32768 };
static std::atomic<unsigned> g_forced{0};
extern "C" void bgth_force_kernels(unsigned flags) { g_forced.store(flags, std::memory_order_relaxed); }
static bool variant_flag(int bit)
{
    unsigned f = g_forced.load(std::memory_order_relaxed);
#ifdef BGTH_ABLATE
    if (const char *d = getenv("BGTH_VARIANT")) f |= (unsigned)atoi(d);
#else
    if (bit == kVariantNoPrefetch || bit == kVariantDirNoWarm || bit == kVariantNoWalkPrio || bit == kVariantPackedRanks) return false;
#endif
    return (f & (unsigned)bit) != 0;
}

// Builds the row index once per image; later calls only return it.  The build is enqueued on `s`, the
// stream of the scan that needs it, and waited for, so that other streams may use the index afterwards.
static bool ensure_rowindex(bgth_pbf_t *p, hipStream_t s)
{
    std::lock_guard<std::mutex> guard(p->rowindex_lock);
    if (p->d_chunkinfo) return true;
    const int64_t n_str = p->n * p->g;
    const int S8 = (p->m + 8191) >> 13;
    const size_t n_ci = (size_t)(p->packed_bytes >> 8) + (size_t)n_str + 2;
    const size_t n_sc = (size_t)n_str * (size_t)(S8 + 1) + 1;
    uint32_t *ci = nullptr, *sc = nullptr;
    HIP_TRY(hipMalloc((void**)&ci, n_ci * 4), return false);
    HIP_TRY(hipMalloc((void**)&sc, n_sc * 4), { hipFree(ci); return false; });
    HIP_TRY(launch_rowindex(p->d_rowdesc, p->d_rle, n_str, p->m, S8, ci, sc, s), { hipFree(ci); hipFree(sc); return false; });
    HIP_TRY(hipStreamSynchronize(s), { hipFree(ci); hipFree(sc); return false; });
    p->d_chunkinfo = ci; p->d_segc = sc; p->S8 = S8;
    p->rowindex_bytes = (int64_t)(n_ci + n_sc) * 4;
    return true;
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool reserve(size_t n)
    {
        if (n <= cap) return true;
        if (p) hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(&p, n) != hipSuccess) return false;
        cap = n;
        return true;
    }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
};

struct HostBuf {   // pinned
    void *p = nullptr;
    size_t cap = 0;
    bool reserve(size_t n)
    {
        if (n <= cap) return true;
        if (p) hipHostFree(p);
        p = nullptr; cap = 0;
        if (hipHostMalloc(&p, n, hipHostMallocPortable) != hipSuccess) return false;   // (pinned for every device: shards)
        cap = n;
        return true;
    }
    void release() { if (p) hipHostFree(p); p = nullptr; cap = 0; }
};

// Buffers of one window of the pull interface.  A reader has two: while the caller consumes the current window the next
// one is decoded and copied behind it on the reader's stream (single-device readers; the shards of a sharded reader fill the
// parent's host buffers from their own first window).
struct PullWindow {
    DevBuf fin, h0, h1, planes, gt8, gttext;
    HostBuf h_counts, h_planes, h_gt8, h_gttext;
    int64_t row0 = 0, row1 = 0;       // image rows it holds (device bit planes and host ring alike)
    int has = 0;                      // BGTH_WANT_* bits it was filled with
    bool valid = false;               // row0 / row1 / has describe its contents
    bool pending = false;             // enqueued on the stream and not waited for yet
    void release()
    {
        fin.release(); h0.release(); h1.release(); planes.release(); gt8.release(); gttext.release();
        h_counts.release(); h_planes.release(); h_gt8.release(); h_gttext.release();
        valid = pending = false;
    }
};


// One host thread per shard of a sharded reader, alive as long as the reader: bgth_reader_scan and every refill of the pull
// interface hand each shard's piece to its worker instead of creating and joining a thread per call (eight thread
// creations per window on an 8-GPU node).  The worker binds its device once.
struct ShardWorker {
    std::thread th;
    std::mutex lock;
    std::condition_variable cv;
    std::function<void()> job;
    bool busy = false, stop = false;
    explicit ShardWorker(int device)
    {
        th = std::thread([this, device] {
            hipSetDevice(device);
            std::unique_lock<std::mutex> g(lock);
            for (;;) {
                cv.wait(g, [this] { return stop || (busy && job); });
                if (stop) return;
                std::function<void()> f = std::move(job);
                job = nullptr;
                g.unlock();
                try { f(); } catch (...) {}                  // (jobs report through their own error slots)
                g.lock();
                busy = false;
                cv.notify_all();
            }
        });
    }
    void submit(std::function<void()> f)
    {
        std::unique_lock<std::mutex> g(lock);
        cv.wait(g, [this] { return !busy; });
        job = std::move(f); busy = true;
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> g(lock);
        cv.wait(g, [this] { return !busy; });
    }
    ~ShardWorker()
    {
        { std::lock_guard<std::mutex> g(lock); stop = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

struct bgth_reader_s {
    bgth_pbf_t *pbf = nullptr;
    Selection sel;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    DevBuf raw, fin, h0, h1, gt;      // scratch of every scan; results of bgth_reader_scan
    DevBuf ph0, ph1;                  // bit planes of the plane-split kernels when the caller wants counts only
    DevBuf start_tab;                 // plane-split kernels: this selection's start ranks at every sub-checkpoint, in slot order
    int64_t start_epoch = -1, start_n_sub = 0;   // (valid for the image's rank_epoch / n_sub; -1: to be gathered)
    int start_slots = 0;
    int plane_path = 0;               // the last scan ran the plane-split kernels
    // directory path: the arena of {bits, ones before} rows and their zero counts; [dir_lo, dir_hi) = image rows it holds
    // from the last producer pass (a later scan inside that range only walks), dir_passes/dir_built = what the last scan did
    DevBuf dir, dir_n0;
    DevBuf tog_mem;                   // cohorts beyond 650,000 haplotypes: the producer's toggle words (dirbuild_mem_kernel)
    int64_t dir_lo = 0, dir_hi = 0;
    hipStream_t dir_stream = nullptr; // the stream the arena was filled on (another stream would have to wait for it)
    int dir_passes = 0, dir_built = 0;
    float dir_build_ms = 0;
    hipEvent_t ev_dir[2] = {nullptr, nullptr};
    hipEvent_t ev_gather = nullptr;   // sharded scan_device: this shard's counts have left for the root device
    hipEvent_t ev_copied = nullptr;   // ... and the root stream's copy out of this shard's `fin` is done: the shard's next scan waits for it
    DevBuf carriers, hapsig;          // allele-set accumulators (bgth_reader_fold_last), zeroed when folds_live turns true
    bool folds_live = false;
    PullWindow win[2];                // pull interface: current window and the one being prefetched
    int cur = 0;
    float t_ms[3] = {0, 0, 0};
    bool t_pending = false;           // events recorded on a caller stream, not yet read back
    Geometry geom = {0, 0, 0, 0, 0, 0, 1, 1};
    int tune_threads = 0, tune_cpt = 0, tune_K = 0;
    // pull interface
    int64_t next = 0, ring0 = 0, ring1 = 0;   // ring0 / ring1 / ring_has: the current window
    int ring_has = 0;                 // BGTH_WANT_* bits the ring was filled with
    int want = BGTH_WANT_PLANES;      // pull interface: what a refill materialises besides the counts
    int64_t max_ahead = 0;            // rows per refill (0 = automatic)
    int64_t ahead = 0;                // automatic look-ahead of the next refill (grows on sequential reads)
    const uint8_t *ret[2] = {nullptr, nullptr};
    const int32_t *last_counts = nullptr;
    const int8_t *last_gt8 = nullptr;
    const char *last_gttext = nullptr;
    std::vector<bgth_reader_t*> subs; // reader of a sharded image: one reader per shard (own device, stream, buffers)
    std::vector<bgth_reader_t*> pair_readers;   // reader of an image of more than two planes: one reader per pair of planes
    std::vector<const uint8_t*> multi_ret;      // ... and what its bgth_reader_read returns: g plane pointers
    std::vector<ShardWorker*> workers; //   and one persistent host thread per shard
};

static bool use_device(int device)
{
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { set_err("[E::bgth] hipSetDevice(%d): %s", device, hipGetErrorString(e)); return false; }
    g_runtime_ready = true;
    return true;
}

// ----------------------------------------------------------------------------------------------------
// selection tables
// ----------------------------------------------------------------------------------------------------
// Output columns are regrouped into SLOTS: all columns of group 0, padded to a multiple of 64, then
// group 1, ...; inside a group the output order is kept.  (ref bgt.c:239-242: the subset list is
// {2*sample, 2*sample+1} in ascending sample order; ref bgt.c:154,612-621: one group id per sample.)
static bool build_selection(Selection &s, int m, int n_sub, const int32_t *sub, const uint32_t *group, int G)
{
    if (n_sub <= 0 || n_sub >= m || sub == nullptr) { n_sub = m; sub = nullptr; }   // ref pbwt.c:377
    if (G < 1 || G > 32) { set_err("[E::bgth_reader_select] n_groups %d out of 1..32", G); return false; }
    if (group == nullptr) G = 1;
    s.width = n_sub; s.G = G; s.whole = sub == nullptr;
    std::vector<std::vector<int32_t>> by_group(G);
    for (int i = 0; i < n_sub; ++i) {
        const int col = sub ? sub[i] : i;
        if (col < 0 || col >= m) { set_err("[E::bgth_reader_select] column %d out of range", col); return false; }
        int g = 0;
        if (group) {
            const uint32_t gi = group[i >> 1];
            if (gi < 1 || gi > (uint32_t)G) { set_err("[E::bgth_reader_select] group id %u out of 1..%d", gi, G); return false; }
            g = (int)gi - 1;
        }
        by_group[g].push_back(i);
    }
    s.slot_of_out.assign(n_sub, -1);
    s.group_haps.assign(G, 0);
    s.slot_col.clear(); s.chunk_desc.clear();
    // Groups start at a multiple of FOUR chunks (one row-step statement) where that costs no launch geometry -- every geometry
    // holds a multiple of 8 chunks, so as long as the padded total stays within the unpadded total rounded up to 8: a statement
    // across two groups is counted chunk by chunk with LDS atomics, and its wave is the one the workgroup waits for in every row
    // (C2 in quarters: 79-chunk groups, three such waves, 13.95 ms against 11.8 for halves or eighths).
    bool align4 = false;
    if (G > 1) {
        size_t plain = 0, padded = 0;
        for (int g = 0; g < G; ++g) {
            const size_t c = (by_group[g].size() + 63) / 64;
            plain += c;
            padded += g + 1 < G ? (c + 3) / 4 * 4 : c;
        }
        align4 = padded <= (plain + 7) / 8 * 8;
    }
    for (int g = 0; g < G; ++g) {
        const std::vector<int32_t> &v = by_group[g];
        s.group_haps[g] = (int32_t)v.size();
        for (size_t k = 0; k < v.size(); k += 64) {
            const int nv = (int)std::min<size_t>(64, v.size() - k);
            for (int l = 0; l < 64; ++l) {
                if (l < nv) {
                    const int out = v[k + l];
                    s.slot_of_out[out] = (int32_t)s.slot_col.size();
                    s.slot_col.push_back(sub ? sub[out] : out);
                } else s.slot_col.push_back(-1);
            }
            s.chunk_desc.push_back((uint32_t)g | (uint32_t)nv << 8);
        }
        if (align4 && g + 1 < G)
            while (s.chunk_desc.size() % 4) {                            // an empty chunk of this group: 64 padding slots
                s.slot_col.insert(s.slot_col.end(), 64, -1);
                s.chunk_desc.push_back((uint32_t)g);
            }
    }
    s.n_chunks = (int)s.chunk_desc.size();
    if (s.n_chunks == 0) { set_err("[E::bgth_reader_select] empty selection"); return false; }
    const size_t nslot = s.slot_col.size();
    auto grow = [&](void **ptr, size_t &have, size_t need) -> bool {
        if (need <= have) return true;
        if (*ptr) hipFree(*ptr);
        *ptr = nullptr; have = 0;
        HIP_TRY(hipMalloc(ptr, need), return false);
        have = need;
        return true;
    };
    if (!grow((void**)&s.d_slot_col, s.cap[0], nslot * 4) || !grow((void**)&s.d_slot_of_out, s.cap[1], (size_t)n_sub * 4) ||
        !grow((void**)&s.d_group_haps, s.cap[2], (size_t)G * 4) || !grow((void**)&s.d_chunk_desc, s.cap[3], (size_t)s.n_chunks * 4)) return false;
    HIP_TRY(hipMemcpy(s.d_slot_col, s.slot_col.data(), nslot * 4, hipMemcpyHostToDevice), return false);
    HIP_TRY(hipMemcpy(s.d_slot_of_out, s.slot_of_out.data(), (size_t)n_sub * 4, hipMemcpyHostToDevice), return false);
    HIP_TRY(hipMemcpy(s.d_group_haps, s.group_haps.data(), (size_t)G * 4, hipMemcpyHostToDevice), return false);
    HIP_TRY(hipMemcpy(s.d_chunk_desc, s.chunk_desc.data(), (size_t)s.n_chunks * 4, hipMemcpyHostToDevice), return false);
    return true;
}

// ----------------------------------------------------------------------------------------------------
// .pbf image -> HBM
// ----------------------------------------------------------------------------------------------------
static void set_rows(bgth_pbf_t *p, int64_t n)
{
    p->n = n;
    p->n_blk = (n + ((int64_t)1 << p->shift) - 1) >> p->shift;
    p->n_sub = (n + ((int64_t)1 << p->sub_shift) - 1) >> p->sub_shift;
}

// Sub-checkpoint spacing of an image of n rows.  A workgroup decodes one sub-block x column slice, so a SHORT image at the
// default spacing of 2048 rows leaves most of the 256 CUs idle (142,000 rows = 70 sub-blocks) or makes the kernels slice its
// columns, each slice repeating the row build.  The spacing that minimises
//        (1 + 10 / rows per sub-block)  x  (workgroups rounded up to whole rounds of what the chip holds) / workgroups
// -- ~10 rows' worth of start-up per workgroup against the idle tail of the last round -- is taken, the default unless a finer
// one is >= 2 % better: 1 M rows keep 2048 (scripts/subshift_ab.py: 10.9 ms at 2048, 11.2 ms at 256), the HRC-shaped 142,000 x
// 64,976 takes 128 (8.7 -> 6.5 ms), 50,000 x 5,008 takes 128 (2.3 -> 0.24 ms).  Costs rows * m / 2^(sub_shift-3) bytes.
static bool want_dir_path(const bgth_pbf_t *p, const Geometry &classic, bool tuned, int width);
static void fit_sub_shift(bgth_pbf_t *p, int64_t n)
{
    if (getenv("BGTH_SUB_SHIFT") || p->wide_plane || n <= 0) return;
    const int top = std::min(p->shift, 11), chunks = (p->m + 63) / 64;
    int slices = 1;                                      // column slices of a long whole-cohort scan
    int64_t slots = 256;                                 // workgroups the chip holds at a time
    bool plane = false;                                  // sample subsets of this width go to the plane-split kernels:
    Geometry g, w;                                       //   one workgroup per sub-block and plane, two per CU
    if (choose_geometry(p->m, chunks, 1, 4096, 0, 0, 0, &g)) {
        slices = g.slices;
        const int64_t n_was = p->n;
        p->n = n;                                        // (use_zp compares the rows whose plane 1 is empty with the image's rows)
        const bool dir = want_dir_path(p, g, false, p->m);
        p->n = n_was;
        if (dir && choose_walk_geometry(p->m, chunks, 1, 4096, 0, 0, &w)) { slices = w.slices; plane = true; }
        else slots = 256 * (int64_t)std::max(1, std::min(2048 / g.threads, (160 * 1024) / std::max(1, g.lds_bytes)));
    }
    // the finer checkpoints may not crowd the device: at most a sixteenth of the HBM that is free now (and never below 256 MB,
    // which every default spacing of a real file stays under); a failed allocation falls back to the default (rank0_alloc)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)4 << 30;
    const double cap_bytes = std::max<double>((double)((size_t)256 << 20), (double)free_b / 16.0);
    if (plane) {
        // Directory path (walk-only workgroups, one per CU): a workgroup's start-up is a few rows' worth (the start ranks come in
        // batches of loads), and many short workgroups balance the CUs better than the rounds alone say -- one C4 shard 147.3 / 145.6 /
        // 142.2 / 140.6 / 140.2 ms at 2048 / 1024 / 512 / 256 / 128 rows, 13,000 samples x 1 M 15.29 ... 14.69, C3's plane-split
        // kernels on the same image +-1 %.  The finest spacing of >= 128 rows that stays under ~80 rounds of workgroups and the cap.
        int pick = top;
        for (int s = top - 1; s >= std::min(top, 7); --s) {
            const int64_t subs = (n + ((int64_t)1 << s) - 1) >> s, wgs = subs * slices;
            if ((double)subs * 8.0 * (double)p->m > cap_bytes || (wgs + slots - 1) / slots > 80) break;
            pick = s;
        }
        p->sub_shift = pick;
        return;
    }
    double best = 1e30;
    int pick = top;
    for (int s = top; s >= std::min(top, 7); --s) {
        const int64_t subs = (n + ((int64_t)1 << s) - 1) >> s, wgs = subs * slices;
        if (s < top && (double)subs * 8.0 * (double)p->m > cap_bytes) break;
        const double start = 1.0 + 10.0 / (double)((int64_t)1 << s);
        double t = start * (double)((wgs + slots - 1) / slots * slots) / (double)wgs;
        if (t < best * 0.98) { best = t; pick = s; }
    }
    p->sub_shift = pick;
}

// d_rank0 for the image's sub-checkpoints; when the spacing fit_sub_shift chose does not fit the HBM that is left, once more at
// the default spacing (an image that opened before the spacing was fitted to the chip still opens)
static bool rank0_alloc(bgth_pbf_t *p)
{
    const size_t per = (size_t)2 * p->m;
    if (hipMalloc((void**)&p->d_rank0, (size_t)std::max<int64_t>(p->n_sub, 1) * per * 4) == hipSuccess) return true;
    (void)hipGetLastError();
    const int top = p->wide_plane ? p->shift : std::min(p->shift, 11);
    if (p->sub_shift < top && !getenv("BGTH_SUB_SHIFT")) {
        p->sub_shift = top;
        set_rows(p, p->n);
        if (hipMalloc((void**)&p->d_rank0, (size_t)std::max<int64_t>(p->n_sub, 1) * per * 4) == hipSuccess) return true;
    }
    p->d_rank0 = nullptr;
    set_err("[E::bgth_pbf] out of HBM for the checkpoints (%lld x %zu bytes): %s", (long long)p->n_sub, per * 4, hipGetErrorString(hipGetLastError()));
    return false;
}

static bgth_pbf_t *pbf_alloc(int device, int m, int g, int shift, int64_t n)
{
    if (g != 2) { set_err("[E::bgth_pbf] g=%d bit planes: an image holds BGT's two planes (import.c:68); whole files of one plane (.pb1) or of more than two open as bundles of such images (bgth_pbf_open / bgth_pbf_open_mem)", g); return nullptr; }
    if (m <= 0 || shift < 0 || shift > 30) { set_err("[E::bgth_pbf] bad header m=%d shift=%d", m, shift); return nullptr; }
    Geometry geo;
    bool wide_plane = false, mem_plane = false;
    if (!choose_geometry(m, (m + 63) / 64, 1, 1, 0, 0, 0, &geo)) {
        // both bit-vectors of a row (m / 2 bytes with their rank directories) do not fit the 160 KiB LDS: one plane per
        // workgroup does, up to m / 4 bytes = 160 KiB
        // ... and beyond 650,000 haplotypes not even that: the producer keeps its toggle words and the walk reads its entries in
        // memory (dirbuild_mem_kernel / walk_mem_kernel: any m the row index addresses -- 30 bits of position)
        if (!choose_walk_plane_geometry(m, (m + 63) / 64, 1, &geo)) {
            if (m > (1 << 30) || ((((m + 31) / 32) + 255) >> 8) > 63 * 16) {
                set_err("[E::bgth_pbf] m=%d columns: this build reads cohorts of up to 264,241,152 haplotypes", m);
                return nullptr;
            }
            mem_plane = true;
        }
        wide_plane = true;
    }
    bgth_pbf_t *p = new bgth_pbf_s();
    p->device = device; p->m = m; p->g = g; p->shift = shift; p->wide_plane = wide_plane; p->mem_plane = mem_plane;
    // Sub-checkpoints: the file carries the permutation every 1 << shift (8192) rows; the image keeps the rank
    // form every 1 << sub_shift rows, derived once on the device.  Finer units = more workgroups per launch (a
    // whole-cohort scan of few blocks fills the GPU without slicing columns, which would repeat the per-row
    // bit-vector build) and shorter pre-rolls for region queries.  Costs rows * m / 2^(sub_shift-3) bytes of HBM.
    p->sub_shift = std::min(shift, 11);
    if (const char *e = getenv("BGTH_SUB_SHIFT")) p->sub_shift = std::max(0, std::min(shift, atoi(e)));
    if (wide_plane) p->sub_shift = shift;                // (the pass that derives sub-checkpoints runs the two-plane kernels)
    fit_sub_shift(p, n);
    set_rows(p, n);
    return p;
}

static void reader_free(bgth_reader_t *r);
extern "C" void bgth_pbf_close(bgth_pbf_t *p)
{
    if (!p) return;
    for (void *c : p->comm) if (c) rccl().CommDestroy(c);
    p->comm.clear();
    for (bgth_pbf_t *sh : p->shards) bgth_pbf_close(sh);
    for (bgth_pbf_t *pp : p->pairs) bgth_pbf_close(pp);
    for (bgth_reader_t *r : p->pool) reader_free(r);
    p->pool.clear();
    hipSetDevice(p->device);
    if (p->d_rle) hipFree(p->d_rle);
    if (p->d_rowdesc) hipFree(p->d_rowdesc);
    if (p->d_rank0) hipFree(p->d_rank0);
    if (p->d_order) hipFree(p->d_order);
    if (p->d_final) hipFree(p->d_final);
    if (p->d_chunkinfo) hipFree(p->d_chunkinfo);
    if (p->d_segc) hipFree(p->d_segc);
    delete p;
}

static bool derive_sub_checkpoints(bgth_pbf_t *p);
static bool string_is_all_zero(const uint8_t *q, size_t l);
static bool run_block_pass(bgth_pbf_t *p, Selection &all, int64_t blk, int64_t n_blk, int32_t *d_final, hipStream_t s,
                           int64_t final_blk_stride);

// BGTH_TRACE=1: wall-clock of the image-open stages on stderr (tuning aid)
struct Trace {
    bool on; std::chrono::steady_clock::time_point t0;
    Trace() : on(getenv("BGTH_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void lap(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[bgth trace] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// Walks the record stream of an image (format: SURVEY.md App. A; ref pbwt.c:288-311 writer,
// :313-337 reader) and splits it into packed RLE bytes, row descriptors and checkpoint permutations.
// What parsing a .pbf image yields: the strings packed 4-byte aligned, a descriptor per string, the checkpoint permutations.
struct Parsed {
    uint8_t *rle = nullptr; size_t rle_n = 0;
    uint64_t *desc = nullptr; size_t n_desc = 0;
    int32_t *perms = nullptr; size_t n_perm = 0;
    int64_t payload = 0, n_empty1 = 0, rows = 0;
    ~Parsed() { free(rle); free(desc); free(perms); }
    void take(const std::vector<uint8_t> &r, const std::vector<uint64_t> &d, const std::vector<int32_t> &pm, int64_t pay, int64_t e1, int64_t n)
    {
        rle_n = r.size(); n_desc = d.size(); n_perm = pm.size(); payload = pay; n_empty1 = e1; rows = n;
        rle = (uint8_t*)malloc(std::max<size_t>(rle_n, 1)); desc = (uint64_t*)malloc(std::max<size_t>(n_desc, 1) * 8); perms = (int32_t*)malloc(std::max<size_t>(n_perm, 1) * 4);
        if (rle_n) memcpy(rle, r.data(), rle_n);
        if (n_desc) memcpy(desc, d.data(), n_desc * 8);
        if (n_perm) memcpy(perms, pm.data(), n_perm * 4);
    }
};

static bool string_is_all_zero(const uint8_t *q, size_t l);

// One file block ('S' record + up to 1 << shift 'B' records) at buf[beg,end).  dst_rle == nullptr: count only.
struct BlockScan { int64_t rows = 0, payload = 0, empty1 = 0; size_t packed = 0; bool ok = false; };
static BlockScan scan_block(const uint8_t *buf, size_t beg, size_t end, int m, int g, uint8_t *dst_rle, uint64_t *dst_desc,
                            uint64_t base_off, int32_t *dst_perm)
{
    BlockScan r;
    size_t pos = beg;
    const size_t sbytes = (size_t)g * m * 4;
    if (pos >= end || buf[pos] != 'S' || pos + 1 + sbytes > end) return r;
    if (dst_perm) memcpy(dst_perm, buf + pos + 1, sbytes);
    pos += 1 + sbytes;
    while (pos < end) {
        if (buf[pos] != 'B') return r;
        ++pos;
        for (int k = 0; k < g; ++k) {
            int32_t l;
            if (pos + 4 > end) return r;
            memcpy(&l, buf + pos, 4);
            pos += 4;
            if (l < 0 || l >= (1 << 24) || pos + (size_t)l > end) return r;
            const size_t pad = ((size_t)l + 3) & ~(size_t)3;
            if (dst_rle) {
                dst_desc[r.rows * g + k] = (base_off + r.packed) | (uint64_t)l << kDescLenShift;
                memcpy(dst_rle + r.packed, buf + pos, (size_t)l);
                for (size_t z = (size_t)l; z < pad; ++z) dst_rle[r.packed + z] = 0;
                if (k == 1 && string_is_all_zero(buf + pos, (size_t)l)) ++r.empty1;    // (the pass that reads the bytes anyway)
            }
            r.packed += pad; r.payload += l;
            pos += (size_t)l;
        }
        ++r.rows;
    }
    r.ok = pos == end;
    return r;
}

// The footer's block index makes the blocks of a file independent: sizes first, then every block copied to its place, on
// several host threads.  false = no usable index (the caller then walks the records in order and reports what is wrong).
// With an Upload the packed strings and the 'S' records never exist as one host copy: every host thread packs a block into
// its own buffer and copies it to its place in HBM, while the others pack (one C4 shard, 2.4 GB: parse 222 + upload 101 +
// freeing the host copy 257 ms become one stage).
struct Upload {
    int device = 0;
    uint8_t *d_rle = nullptr;          // rle_n + 256 bytes (kernels may read whole dwords past the last string)
    int32_t *d_perm = nullptr;         // the 'S' records, in file order
    bool failed = false;               // a HIP call failed (the error is set): do not try the sequential parser
};
static bool parse_blocks_parallel(const uint8_t *buf, size_t end, size_t len, int m, int g, int shift, int64_t n_rows, Parsed &out,
                                  Upload *up = nullptr)
{
    if (end + 13 > len || n_rows <= 0 || m <= 0 || g != 2) return false;
    int32_t n_idx;
    memcpy(&n_idx, buf + end + 9, 4);
    const int64_t blk_rows = (int64_t)1 << shift;
    if (n_idx <= 0 || (int64_t)n_idx != (n_rows + blk_rows - 1) / blk_rows || end + 13 + (size_t)n_idx * 8 + 8 > len) return false;
    std::vector<uint64_t> idx((size_t)n_idx + 1);
    memcpy(idx.data(), buf + end + 13, (size_t)n_idx * 8);
    idx[n_idx] = end;
    if (idx[0] != 16) return false;
    for (int i = 0; i < n_idx; ++i) if (idx[i] >= idx[i + 1]) return false;
    const int nt = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<BlockScan> info((size_t)n_idx);
    auto run = [&](auto fn) {
        std::vector<std::thread> th;
        std::atomic<int> next(0);
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { for (int b; (b = next.fetch_add(1)) < n_idx;) fn(b, t); });
        for (std::thread &t : th) t.join();
    };
    Trace tr;
    run([&](int b, int) { info[b] = scan_block(buf, idx[b], idx[b + 1], m, g, nullptr, nullptr, 0, nullptr); });
    tr.lap("  blocks sized");
    std::vector<uint64_t> off((size_t)n_idx + 1, 0);
    std::vector<int64_t> row0((size_t)n_idx + 1, 0);
    for (int b = 0; b < n_idx; ++b) {
        if (!info[b].ok || info[b].rows != (b + 1 < n_idx ? blk_rows : n_rows - (int64_t)b * blk_rows)) return false;
        off[b + 1] = off[b] + info[b].packed; row0[b + 1] = row0[b] + info[b].rows;
        out.payload += info[b].payload;
    }
    if (off[n_idx] >= ((uint64_t)1 << kDescLenShift)) return false;
    out.rows = n_rows;
    out.rle_n = off[n_idx]; out.n_desc = (size_t)n_rows * g; out.n_perm = (size_t)n_idx * g * m;
    out.desc = (uint64_t*)malloc(out.n_desc * 8);
    if (!out.desc) throw std::bad_alloc();
    std::atomic<int64_t> empty1(0);
    if (!up) {
        out.rle = (uint8_t*)malloc(std::max<size_t>(out.rle_n, 1));
        out.perms = (int32_t*)malloc(out.n_perm * 4);
        if (!out.rle || !out.perms) throw std::bad_alloc();
        run([&](int b, int) {
            empty1 += scan_block(buf, idx[b], idx[b + 1], m, g, out.rle + off[b], out.desc + (size_t)row0[b] * g, off[b], out.perms + (size_t)b * g * m).empty1;
        });
        out.n_empty1 = empty1;
        return true;
    }
    const size_t pad = 256, sbytes = (size_t)g * m * 4;
    size_t widest = 0;
    for (int b = 0; b < n_idx; ++b) widest = std::max(widest, info[b].packed);
    up->failed = true;
    if (!use_device(up->device)) return false;
    HIP_TRY(hipMalloc((void**)&up->d_rle, out.rle_n + pad), return false);
    HIP_TRY(hipMemset(up->d_rle + out.rle_n, 0, pad), return false);
    HIP_TRY(hipMalloc((void**)&up->d_perm, out.n_perm * 4), return false);
    std::vector<uint8_t*> mine((size_t)nt, nullptr);
    std::atomic<int> hip_err((int)hipSuccess), no_mem(0);
    run([&](int b, int t) {
        if (hip_err.load() != (int)hipSuccess || no_mem.load()) return;
        if (!mine[t]) {
            mine[t] = (uint8_t*)malloc(widest + sbytes);
            if (!mine[t]) { no_mem = 1; return; }
            hipError_t e = hipSetDevice(up->device);
            if (e != hipSuccess) { hip_err = (int)e; return; }
        }
        empty1 += scan_block(buf, idx[b], idx[b + 1], m, g, mine[t], out.desc + (size_t)row0[b] * g, off[b], (int32_t*)(mine[t] + widest)).empty1;
        hipError_t e = info[b].packed ? hipMemcpy(up->d_rle + off[b], mine[t], info[b].packed, hipMemcpyHostToDevice) : hipSuccess;
        if (e == hipSuccess) e = hipMemcpy(up->d_perm + (size_t)b * g * m, mine[t] + widest, sbytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) hip_err = (int)e;
    });
    for (uint8_t *q : mine) free(q);
    tr.lap("  blocks packed + copied");
    if (no_mem.load()) throw std::bad_alloc();
    if (hip_err.load() != (int)hipSuccess) { set_err("[E::bgth_pbf_open] copying the strings to the device: %s", hipGetErrorString((hipError_t)hip_err.load())); return false; }
    out.n_empty1 = empty1;
    up->failed = false;
    return true;
}

// The record stream walked in order (format: SURVEY.md App. A; ref pbwt.c:288-311 writer, :313-337 reader): the path of
// images without a usable block index, and the one that names what is wrong with a damaged file.
static bool parse_sequential(const uint8_t *buf, size_t end, int m, int g, int shift, Parsed &ps)
{
    int64_t n_empty1 = 0;
    std::vector<uint8_t> rle;
    std::vector<uint64_t> desc;
    std::vector<int32_t> perms;
    int64_t payload = 0;
    rle.reserve(end);
    size_t pos = 16;
    int64_t row = 0;
    const int64_t blk_rows = (int64_t)1 << shift;
    while (pos < end && buf[pos] != 'I') {
        if (buf[pos] == 'S') {
            if (row % blk_rows != 0) { set_err("[E::bgth_pbf_open] 'S' record at row %lld is not on a block boundary", (long long)row); return false; }
            if (pos + 1 + (size_t)g * m * 4 > end) { set_err("[E::bgth_pbf_open] truncated 'S' record"); return false; }
            const size_t at = perms.size();
            perms.resize(at + (size_t)g * m);
            memcpy(perms.data() + at, buf + pos + 1, (size_t)g * m * 4);
            pos += 1 + (size_t)g * m * 4;
        } else if (row % blk_rows == 0) { set_err("[E::bgth_pbf_open] missing 'S' record at row %lld", (long long)row); return false; }
        if (pos >= end || buf[pos] != 'B') { set_err("[E::bgth_pbf_open] bad record tag at offset %zu", pos); return false; }
        ++pos;
        for (int k = 0; k < g; ++k) {
            int32_t l;
            if (pos + 4 > end) { set_err("[E::bgth_pbf_open] truncated 'B' record"); return false; }
            memcpy(&l, buf + pos, 4);
            pos += 4;
            if (l < 0 || l >= (1 << 24) || pos + (size_t)l > end) { set_err("[E::bgth_pbf_open] bad RLE length %d at row %lld", l, (long long)row); return false; }
            while (rle.size() & 3) rle.push_back(0);          // kernels read whole aligned dwords
            desc.push_back((uint64_t)rle.size() | (uint64_t)l << kDescLenShift);
            rle.insert(rle.end(), buf + pos, buf + pos + l);
            if (k == 1 && string_is_all_zero(buf + pos, (size_t)l)) ++n_empty1;
            payload += l;
            pos += (size_t)l;
        }
        ++row;
    }
    ps.take(rle, desc, perms, payload, n_empty1, row);
    return true;
}

static bgth_pbf_t *open_mem_impl(const void *image, size_t len, int device);
extern "C" bgth_pbf_t *bgth_pbf_open_mem(const void *image, size_t len, int device)
{
    return guarded("bgth_pbf_open", (bgth_pbf_t*)nullptr, [&] { return open_mem_impl(image, len, device); });
}

// A ONE-plane file (g = 1: the prefix.pb1 that `import -1` writes, reference import.c:72-74) as the two-plane image the
// kernels hold: every 'S' record gets the identity order for a second plane, every 'B' record a second, empty string (a
// string that stops short leaves the rest of its row at its last bit, 0: a plane that is all zero, whose ranks never
// move and which the empty-plane kernels do not even look up).  The footer is rebuilt for the new offsets.
static bool expand_one_plane(const uint8_t *buf, size_t len, std::vector<uint8_t> &out)
{
    int32_t hdr[3];
    memcpy(hdr, buf + 4, 12);
    const int m = hdr[0];
    if (m <= 0 || m > (1 << 28)) { set_err("[E::bgth_pbf_open] bad header m=%d", m); return false; }
    out.clear();
    out.reserve(len + len / 2 + 64);
    hdr[1] = 2;
    out.insert(out.end(), buf, buf + 4);
    out.insert(out.end(), (const uint8_t*)hdr, (const uint8_t*)hdr + 12);
    std::vector<int32_t> ident((size_t)m);
    for (int j = 0; j < m; ++j) ident[j] = j;
    std::vector<uint64_t> idx;
    int64_t rows = 0;
    size_t pos = 16;
    const int32_t zero = 0;
    while (pos < len && buf[pos] != 'I') {
        if (buf[pos] == 'S') {
            if (pos + 1 + (size_t)m * 4 > len) { set_err("[E::bgth_pbf_open] truncated 'S' record"); return false; }
            idx.push_back((uint64_t)out.size());
            out.insert(out.end(), buf + pos, buf + pos + 1 + (size_t)m * 4);
            out.insert(out.end(), (const uint8_t*)ident.data(), (const uint8_t*)ident.data() + (size_t)m * 4);
            pos += 1 + (size_t)m * 4;
        }
        if (pos >= len || buf[pos] != 'B' || pos + 5 > len) { set_err("[E::bgth_pbf_open] malformed record at byte %zu", pos); return false; }
        int32_t l;
        memcpy(&l, buf + pos + 1, 4);
        if (l < 0 || pos + 5 + (size_t)l > len) { set_err("[E::bgth_pbf_open] truncated 'B' record"); return false; }
        out.insert(out.end(), buf + pos, buf + pos + 5 + (size_t)l);
        out.insert(out.end(), (const uint8_t*)&zero, (const uint8_t*)&zero + 4);
        pos += 5 + (size_t)l;
        ++rows;
    }
    if (pos >= len) { set_err("[E::bgth_pbf_open] no index footer: truncated or not a PBF image"); return false; }
    const uint64_t off = (uint64_t)out.size();
    const int32_t n_idx = (int32_t)idx.size();
    out.push_back('I');
    out.insert(out.end(), (const uint8_t*)&rows, (const uint8_t*)&rows + 8);
    out.insert(out.end(), (const uint8_t*)&n_idx, (const uint8_t*)&n_idx + 4);
    out.insert(out.end(), (const uint8_t*)idx.data(), (const uint8_t*)idx.data() + idx.size() * 8);
    out.insert(out.end(), (const uint8_t*)&off, (const uint8_t*)&off + 8);
    return true;
}

// Planes [k0, k0 + nk) of a file of g planes as a file of nk planes (nk = 1 or 2): 'S' records keep those planes' orders, 'B'
// records those planes' strings, the footer is rebuilt for the new offsets.  Every bit plane of a PBF is a PBWT of its own
// (pbwt.c:211-213: one pbc_t per plane), so the pieces decode independently.
static bool extract_planes(const uint8_t *buf, size_t len, int k0, int nk, std::vector<uint8_t> &out)
{
    int32_t hdr[3];
    memcpy(hdr, buf + 4, 12);
    const int m = hdr[0], g = hdr[1];
    if (m <= 0 || m > (1 << 28) || g < 1 || k0 < 0 || nk < 1 || k0 + nk > g) { set_err("[E::bgth_pbf_open] bad header m=%d g=%d", m, g); return false; }
    out.clear();
    out.reserve(len / (size_t)g * (size_t)nk + 64);
    hdr[1] = nk;
    out.insert(out.end(), buf, buf + 4);
    out.insert(out.end(), (const uint8_t*)hdr, (const uint8_t*)hdr + 12);
    std::vector<uint64_t> idx;
    int64_t rows = 0;
    size_t pos = 16;
    const size_t perm_bytes = (size_t)m * 4;
    while (pos < len && buf[pos] != 'I') {
        if (buf[pos] == 'S') {
            if (pos + 1 + (size_t)g * perm_bytes > len) { set_err("[E::bgth_pbf_open] truncated 'S' record"); return false; }
            idx.push_back((uint64_t)out.size());
            out.push_back('S');
            out.insert(out.end(), buf + pos + 1 + (size_t)k0 * perm_bytes, buf + pos + 1 + (size_t)(k0 + nk) * perm_bytes);
            pos += 1 + (size_t)g * perm_bytes;
        }
        if (pos >= len || buf[pos] != 'B') { set_err("[E::bgth_pbf_open] malformed record at byte %zu", pos); return false; }
        ++pos;
        out.push_back('B');
        for (int k = 0; k < g; ++k) {
            int32_t l;
            if (pos + 4 > len) { set_err("[E::bgth_pbf_open] truncated 'B' record"); return false; }
            memcpy(&l, buf + pos, 4);
            if (l < 0 || pos + 4 + (size_t)l > len) { set_err("[E::bgth_pbf_open] truncated 'B' record"); return false; }
            if (k >= k0 && k < k0 + nk) out.insert(out.end(), buf + pos, buf + pos + 4 + (size_t)l);
            pos += 4 + (size_t)l;
        }
        ++rows;
    }
    if (pos >= len) { set_err("[E::bgth_pbf_open] no index footer: truncated or not a PBF image"); return false; }
    const uint64_t off = (uint64_t)out.size();
    const int32_t n_idx = (int32_t)idx.size();
    out.push_back('I');
    out.insert(out.end(), (const uint8_t*)&rows, (const uint8_t*)&rows + 8);
    out.insert(out.end(), (const uint8_t*)&n_idx, (const uint8_t*)&n_idx + 4);
    out.insert(out.end(), (const uint8_t*)idx.data(), (const uint8_t*)idx.data() + idx.size() * 8);
    out.insert(out.end(), (const uint8_t*)&off, (const uint8_t*)&off + 8);
    return true;
}

static bgth_pbf_t *open_mem_impl(const void *image, size_t len, int device)
{
    const uint8_t *buf = (const uint8_t*)image;
    if (len < 16 || memcmp(buf, "PBF\1", 4) != 0) { set_err("[E::bgth_pbf_open] not a PBF image"); return nullptr; }
    int32_t hdr[3];
    memcpy(hdr, buf + 4, 12);
    if (hdr[1] > 2 && hdr[1] <= 64) {                            // more than two planes: an image per pair of planes (the last may be single)
        bgth_pbf_t *par = new bgth_pbf_s();
        par->device = device; par->m = hdr[0]; par->g = hdr[1]; par->shift = hdr[2];
        std::vector<uint8_t> piece;
        for (int k0 = 0; k0 < hdr[1]; k0 += 2) {
            bgth_pbf_t *pp = extract_planes(buf, len, k0, std::min(2, hdr[1] - k0), piece) ? open_mem_impl(piece.data(), piece.size(), device) : nullptr;
            if (!pp) { bgth_pbf_close(par); return nullptr; }
            par->pairs.push_back(pp);
        }
        par->file_image.assign(buf, buf + len);
        par->n = par->n_total = par->pairs[0]->n_total;
        for (const bgth_pbf_t *pp : par->pairs) { par->rle_bytes += pp->rle_bytes; par->packed_bytes += pp->packed_bytes; }
        return par;
    }
    if (hdr[1] == 1) {                                            // one plane: held as two, the second one empty
        std::vector<uint8_t> two;
        if (!expand_one_plane(buf, len, two)) return nullptr;
        bgth_pbf_t *p1 = open_mem_impl(two.data(), two.size(), device);
        if (p1) p1->g_file = 1;
        return p1;
    }
    const int m = hdr[0], g = hdr[1], shift = hdr[2];
    int64_t n_footer = -1;
    size_t end = len;
    if (len >= 16 + 21) {
        uint64_t off;
        memcpy(&off, buf + len - 8, 8);
        if (off >= 16 && off + 13 <= len && buf[off] == 'I') { memcpy(&n_footer, buf + off + 1, 8); end = (size_t)off; }
    }
    // the reference reader finds everything through the footer (pbwt.c:228-235); an image without one is truncated
    if (n_footer < 0) { set_err("[E::bgth_pbf_open] no index footer: truncated or not a PBF image"); return nullptr; }
    Trace tr;
    // A cold process is still starting the HIP runtime on its warm-up thread (60-220 ms): then everything the host can do
    // alone comes first -- both passes of the block parser, the strings packed into host memory and uploaded in one piece --
    // instead of waiting for the runtime and copying block by block (C2 database, cold `bgt view`: ~45 ms of 230).  With the
    // runtime up (a resident host, a second database) the blocks go to the device as they are packed.
    const bool beside_init = !g_runtime_ready.load() && !getenv("BGTH_OPEN_WAIT_FIRST");
    if (!beside_init) {
        if (!use_device(device)) return nullptr;
        tr.lap("device init");
    }
    bgth_pbf_t *p = pbf_alloc(device, m, g, shift, 0);
    if (!p) return nullptr;
    t_building = p;

    Parsed ps;
    Upload up;
    up.device = device;
    if (!parse_blocks_parallel(buf, end, len, m, g, shift, n_footer, ps, beside_init ? nullptr : &up)) {
        p->d_rle = up.d_rle;                                     // (freed with the image)
        if (up.d_perm) hipFree(up.d_perm);
        up.d_rle = nullptr; up.d_perm = nullptr;
        if (up.failed || !parse_sequential(buf, end, m, g, shift, ps)) goto fail;
        p->d_rle = nullptr;
    }
    if (beside_init) {
        tr.lap("parse records (beside the runtime's start)");
        if (!use_device(device)) goto fail;
        tr.lap("device init (what was left of it)");
    }
    p->n_empty1 = ps.n_empty1;
    {
    const int64_t row = ps.rows, payload = ps.payload;
    struct { const Parsed &q; size_t size() const { return q.rle_n; } bool empty() const { return q.rle_n == 0; } const uint8_t *data() const { return q.rle; } } rle = {ps};
    struct { const Parsed &q; size_t size() const { return q.n_desc; } bool empty() const { return q.n_desc == 0; } const uint64_t *data() const { return q.desc; } } desc = {ps};
    struct { const Parsed &q; size_t size() const { return q.n_perm; } const int32_t *data() const { return q.perms; } } perms = {ps};
    if (n_footer >= 0 && n_footer != row) { set_err("[E::bgth_pbf_open] footer says %lld rows, stream has %lld", (long long)n_footer, (long long)row); goto fail; }
    if (rle.size() >= ((size_t)1 << kDescLenShift)) { set_err("[E::bgth_pbf_open] RLE payload too large"); goto fail; }
    fit_sub_shift(p, row);
    set_rows(p, row);
    p->n_total = row;
    // BGTH_OPEN_HINT=walk (set by `bgt view` for a one-shot walk of the whole file): when the file blocks alone fill the
    // chip -- blocks x column slices >= two workgroups per CU -- the pass that derives sub-checkpoints costs as much as the
    // walk it is meant to spread, so it is skipped (one C4 shard: 319 of 1,460 ms).  A resident process wants them: they
    // cut the pre-roll of a region query from <= 8191 to <= 2047 rows.
    if (const char *hint = getenv("BGTH_OPEN_HINT")) {
        Geometry wg;
        if (strcmp(hint, "walk") == 0 && !p->wide_plane && p->sub_shift < p->shift &&
            choose_walk_geometry(m, (m + 63) / 64, 1, (int)p->n_blk, 0, 0, &wg) &&
            p->n_blk * wg.slices >= (getenv("BGTH_OPEN_HINT_MIN") ? atoll(getenv("BGTH_OPEN_HINT_MIN")) : 512)) {   // (the variable: tests)
            p->sub_shift = p->shift;
            p->one_shot = true;
            if (getenv("BGTH_TRACE")) fprintf(stderr, "[bgth trace] one-shot walk: %lld file blocks x %d column slices, no sub-checkpoints\n", (long long)p->n_blk, wg.slices);
            set_rows(p, row);
        }
    }
    tr.lap("parse records");
    p->rle_bytes = payload;
    p->packed_bytes = (int64_t)rle.size();
    {
        const size_t pad = 256;    // kernels may read whole dwords past the last string
        if (up.d_rle) p->d_rle = up.d_rle;                      // the parser put the strings in place
        else {
            HIP_TRY(hipMalloc((void**)&p->d_rle, rle.size() + pad), goto fail);
            HIP_TRY(hipMemset(p->d_rle + rle.size(), 0, pad), goto fail);
            // (ONE copy.  Measured and dropped: eight threads copying slices -- 48-97 against 35-40 ms for 331 MB, round 4 --, and a few
            //  threads each staging 8 MB chunks through a pinned buffer of their own + hipMemcpyAsync -- a cold C2 `bgt view` 261-356 ms
            //  against 225-305, this stage 100 ms against 70, round 6: profiles/r06_cold)
            if (!rle.empty()) HIP_TRY(hipMemcpy(p->d_rle, rle.data(), rle.size(), hipMemcpyHostToDevice), goto fail);
        }
        HIP_TRY(hipMalloc((void**)&p->d_rowdesc, std::max<size_t>(desc.size(), 1) * 8), goto fail);
        if (!desc.empty()) HIP_TRY(hipMemcpy(p->d_rowdesc, desc.data(), desc.size() * 8, hipMemcpyHostToDevice), goto fail);
        tr.lap("upload strings");
        // checkpoints: permutation (rank -> column) to rank form (column -> rank), on the device, each at the
        // sub-block index of its row; then one decode pass fills the sub-checkpoints in between
        const size_t np = perms.size(), per = (size_t)2 * m;
        if (np) {
            int32_t *d_perm = up.d_perm;
            int *d_bad = nullptr, bad = 0;
            up.d_perm = nullptr;
            if (!d_perm) HIP_TRY(hipMalloc((void**)&d_perm, np * 4), goto fail);
            HIP_TRY(hipMalloc((void**)&d_bad, 4), { hipFree(d_perm); goto fail; });
            HIP_TRY(hipMemset(d_bad, 0, 4), { hipFree(d_perm); hipFree(d_bad); goto fail; });
            if (!rank0_alloc(p)) { hipFree(d_perm); hipFree(d_bad); goto fail; }
            const int d = p->shift - p->sub_shift;                // (after the allocation: it may fall back to the default spacing)
            if (perms.data()) HIP_TRY(hipMemcpy(d_perm, perms.data(), np * 4, hipMemcpyHostToDevice), { hipFree(d_perm); hipFree(d_bad); goto fail; });
            for (size_t b = 0; b < np / per; ++b)                 // validated: the records come from a file
                HIP_TRY(launch_invert(d_perm + b * per, p->d_rank0 + ((size_t)b << d) * per, m, 2, nullptr, d_bad), { hipFree(d_perm); hipFree(d_bad); goto fail; });
            HIP_TRY(hipDeviceSynchronize(), { hipFree(d_perm); hipFree(d_bad); goto fail; });
            HIP_TRY(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost), { hipFree(d_perm); hipFree(d_bad); goto fail; });
            hipFree(d_perm); hipFree(d_bad);
            if (bad) {
                set_err("[E::bgth_pbf_open] corrupt 'S' record: %s", (bad & 1) ? "an entry is outside 0..m-1" : "not a permutation of the columns");
                goto fail;
            }
            tr.lap("checkpoints -> rank form");
            if (!derive_sub_checkpoints(p)) goto fail;
            tr.lap("sub-checkpoint pass");
        }
    }
    }
    t_building = nullptr;
    return p;
fail:
    t_building = nullptr;
    if (up.d_perm) hipFree(up.d_perm);
    bgth_pbf_close(p);
    return nullptr;
}

extern "C" bgth_pbf_t *bgth_pbf_open(const char *path, int device)
{
    // the file is mapped, not read: the parser copies every string once, from the page cache to its packed place
    const int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        if (fd >= 0) close(fd);
        set_err("[E::bgth_pbf_open] cannot open '%s'", path);
        return nullptr;
    }
    if (st.st_size == 0) { close(fd); set_err("[E::bgth_pbf_open] not a PBF image"); return nullptr; }
    void *map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) { set_err("[E::bgth_pbf_open] cannot map '%s'", path); return nullptr; }
    madvise(map, (size_t)st.st_size, MADV_SEQUENTIAL);       // (advice values are enumerators, not flags: one call each)
    madvise(map, (size_t)st.st_size, MADV_WILLNEED);
    bgth_pbf_t *p = bgth_pbf_open_mem(map, (size_t)st.st_size, device);
    Trace tr;
    munmap(map, (size_t)st.st_size);                          // (leaving it to the process's exit costs the same there: measured)
    tr.lap("unmap the file");
    return p;
}

// A string without a byte of bit 1 describes an all-zero row (the row starts at 0 and nothing toggles it).
static bool string_is_all_zero(const uint8_t *q, size_t l)
{
    uint8_t any = 0;
    for (size_t i = 0; i < l; ++i) any |= q[i];
    return !(any & 1);
}

// Kernels with the all-zero-plane-1 shortcut pay a taken scalar branch per statement on rows that cannot use it
// (2 % on the benchmark cohort, where every row has missing calls) and walk a row with an empty plane 1 in half
// the lookups (1.5x on a fully called panel): use them when at least one row in eight qualifies.
static bool use_zp(const bgth_pbf_t *p)
{
    if (variant_flag(kVariantNeverZP)) return false;
    if (variant_flag(kVariantAlwaysZP)) return true;
    return p->n_empty1 * 8 >= std::max<int64_t>(p->n, 1);
}

// The part of the kernel arguments that only depends on the image, the selection and the launch geometry.
static bool common_scan_args(ScanArgs &a, bgth_pbf_t *p, const Selection &sel, const Geometry &geo, hipStream_t s)
{
    memset(&a, 0, sizeof(a));
    a.rle = p->d_rle;
    a.rowdesc = p->d_rowdesc;
    a.rank0 = p->d_rank0;
    a.slot_col = sel.d_slot_col;
    a.chunk_desc = sel.d_chunk_desc;
    a.m = p->m;
    a.nw = (p->m + 31) / 32;
    a.n_chunks = sel.n_chunks;
    a.G = sel.G;
    a.K = geo.K;
    a.wpp = geo.wpp;
    a.nbuf = geo.nbuf;
    a.n_slices = geo.slices;
    a.zp = use_zp(p) ? 1 : 0;
    a.walk_prio = variant_flag(kVariantNoWalkPrio) ? 0 : 1;
    if (geo.wpp > 1) {                                   // team (wide-cohort) kernels read the row index
        if (!ensure_rowindex(p, s)) return false;
        a.chunkinfo = p->d_chunkinfo;
        a.segc = p->d_segc;
        a.S8 = p->S8;
        a.tog_off = geo.tog_off;
    }
    return true;
}

// Decode pass over the FILE blocks [blk, blk + n_blk) that emits nothing but ranks: the sub-checkpoints inside
// the blocks (always) and the ranks after the last row (d_final, optional: the next block's checkpoint when an
// image is built from bare RLE strings).
static bool run_block_pass(bgth_pbf_t *p, Selection &all, int64_t blk, int64_t n_blk, int32_t *d_final, hipStream_t s,
                           int64_t final_blk_stride)
{
    Geometry geo;
    if (!choose_geometry(p->m, all.n_chunks, 1, (int)n_blk, 0, 0, 0, &geo, !variant_flag(kVariantNoTog))) { set_err("[E::bgth] geometry"); return false; }
    ScanArgs a;
    if (!common_scan_args(a, p, all, geo, s)) return false;
    a.shift = p->shift;                                                                   // units = file blocks,
    a.rank0_blk_stride = ((int64_t)2 * p->m) << (p->shift - p->sub_shift);               // whose checkpoint sits at its sub index
    a.final_rank = d_final;
    a.final_blk_stride = final_blk_stride;               // != 0: one record of final ranks per block of the launch
    if (p->sub_shift < p->shift) { a.snap = p->d_rank0; a.snap_shift = p->sub_shift; }
    a.blk0 = (int32_t)blk;
    a.n_blk = (int32_t)n_blk;
    a.row1 = std::min<int64_t>(p->n, (blk + n_blk) << p->shift);
    a.row0 = a.row1;                      // nothing emitted: only ranks are wanted
    HIP_TRY(launch_scan(a, geo, s), return false);
    return true;
}

// Partial image: only the file blocks that cover rows [row0, row1) are read (through the footer's block index,
// pbwt.c:268-276), parsed and uploaded -- what a region query needs instead of the whole file.
static bgth_pbf_t *open_rows_impl(const char *path, int64_t row0, int64_t row1, int device);
extern "C" bgth_pbf_t *bgth_pbf_open_rows(const char *path, int64_t row0, int64_t row1, int device)
{
    return guarded("bgth_pbf_open_rows", (bgth_pbf_t*)nullptr, [&] { return open_rows_impl(path, row0, row1, device); });
}

static bgth_pbf_t *open_rows_impl(const char *path, int64_t row0, int64_t row1, int device)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) { set_err("[E::bgth_pbf_open_rows] cannot open '%s'", path); return nullptr; }
    uint8_t hdr[16], tail[8], rec[13];
    uint64_t off = 0;
    int64_t n_total = 0;
    int32_t n_idx = 0, shift;
    std::vector<uint64_t> idx;
    std::vector<uint8_t> img;
    bgth_pbf_t *p = nullptr;
    bool ok = fread(hdr, 1, 16, fp) == 16 && memcmp(hdr, "PBF\1", 4) == 0 && fseek(fp, -8, SEEK_END) == 0 && fread(tail, 1, 8, fp) == 8;
    if (ok) { memcpy(&off, tail, 8); ok = off >= 16 && fseek(fp, (long)off, SEEK_SET) == 0 && fread(rec, 1, 13, fp) == 13 && rec[0] == 'I'; }
    if (ok) {
        memcpy(&n_total, rec + 1, 8); memcpy(&n_idx, rec + 9, 4);
        ok = n_total >= 0 && n_idx >= 0;
        if (ok) { idx.resize((size_t)n_idx); ok = n_idx == 0 || fread(idx.data(), 8, (size_t)n_idx, fp) == (size_t)n_idx; }
    }
    if (!ok) { fclose(fp); set_err("[E::bgth_pbf_open_rows] '%s': no PBF header / index footer", path); return nullptr; }
    memcpy(&shift, hdr + 12, 4);
    {
        int32_t g_file;
        memcpy(&g_file, hdr + 8, 4);
        if (g_file > 2) { fclose(fp); set_err("[E::bgth_pbf_open_rows] '%s' has %d bit planes: partial and sharded images are for BGT's two (open the whole file)", path, g_file); return nullptr; }
    }
    if (shift < 0 || shift > 30 || row0 < 0 || row1 > n_total || row0 >= row1) {
        fclose(fp);
        set_err("[E::bgth_pbf_open_rows] rows [%lld,%lld) outside 0..%lld", (long long)row0, (long long)row1, (long long)n_total);
        return nullptr;
    }
    {
        const int64_t b0 = row0 >> shift, b1 = (row1 - 1) >> shift;
        if (b1 >= n_idx) { fclose(fp); set_err("[E::bgth_pbf_open_rows] block index shorter than the file's rows"); return nullptr; }
        const uint64_t beg = idx[(size_t)b0], end = b1 + 1 < n_idx ? idx[(size_t)b1 + 1] : off;
        const int64_t n_rows = std::min<int64_t>(n_total, (b1 + 1) << shift) - (b0 << shift);
        // a small image of its own: header + the blocks + a footer without block index
        img.resize(16 + (size_t)(end - beg) + 21);
        memcpy(img.data(), hdr, 16);
        ok = end >= beg && fseek(fp, (long)beg, SEEK_SET) == 0 && fread(img.data() + 16, 1, (size_t)(end - beg), fp) == (size_t)(end - beg);
        fclose(fp);
        if (!ok) { set_err("[E::bgth_pbf_open_rows] short read on '%s'", path); return nullptr; }
        uint8_t *f = img.data() + 16 + (size_t)(end - beg);
        const uint64_t foot = 16 + (end - beg);
        const int32_t zero = 0;
        f[0] = 'I'; memcpy(f + 1, &n_rows, 8); memcpy(f + 9, &zero, 4); memcpy(f + 13, &foot, 8);
        p = bgth_pbf_open_mem(img.data(), img.size(), device);
        if (p) { p->row_off = b0 << shift; p->n_total = n_total; }
    }
    return p;
}

// Block-aligned site-range shards: ceil(B / P) file blocks each, the last ones may be shorter or empty.
extern "C" void bgth_shard_ranges(int64_t n_rows, int shift, int n_shards, int64_t *ranges)
{
    const int64_t n_blk = (n_rows + ((int64_t)1 << shift) - 1) >> shift, per = (n_blk + n_shards - 1) / std::max(n_shards, 1);
    for (int i = 0; i < n_shards; ++i) {
        const int64_t b0 = std::min(n_blk, i * per), b1 = std::min(n_blk, (i + 1) * per);
        ranges[2 * i] = std::min(n_rows, b0 << shift); ranges[2 * i + 1] = std::min(n_rows, b1 << shift);
    }
}

static bool read_pbf_geometry(const char *path, int32_t hdr[3], int64_t *n_total)
{
    FILE *fp = fopen(path, "rb");
    uint8_t h[16], tail[8], rec[9];
    uint64_t off = 0;
    bool ok = fp && fread(h, 1, 16, fp) == 16 && memcmp(h, "PBF\1", 4) == 0 && fseek(fp, -8, SEEK_END) == 0 && fread(tail, 1, 8, fp) == 8;
    if (ok) { memcpy(&off, tail, 8); ok = off >= 16 && fseek(fp, (long)off, SEEK_SET) == 0 && fread(rec, 1, 9, fp) == 9 && rec[0] == 'I'; }
    if (fp) fclose(fp);
    if (!ok) return false;
    memcpy(hdr, h + 4, 12); memcpy(n_total, rec + 1, 8);
    return true;
}

extern "C" bgth_pbf_t *bgth_pbf_open_sharded(const char *path, int n_shards, const int *devices)
{
    return guarded("bgth_pbf_open_sharded", (bgth_pbf_t*)nullptr, [&]() -> bgth_pbf_t* {
        int32_t hdr[3]; int64_t n_total = 0;
        if (n_shards < 1 || n_shards > 64 || !devices) { set_err("[E::bgth_pbf_open_sharded] 1..64 shards, one device each"); return nullptr; }
        if (!read_pbf_geometry(path, hdr, &n_total) || n_total < 0 || hdr[2] < 0 || hdr[2] > 30) { set_err("[E::bgth_pbf_open_sharded] '%s': no PBF header / index footer", path); return nullptr; }
        {   // every device named must exist BEFORE anything is read or allocated: a clean refusal, not a half-opened image
            int n_dev = 0;
            if (hipGetDeviceCount(&n_dev) != hipSuccess) n_dev = 0;
            for (int i = 0; i < n_shards; ++i)
                if (devices[i] < 0 || devices[i] >= n_dev) {
                    set_err("[E::bgth_pbf_open_sharded] shard %d names device %d, this process sees %d device(s)", i, devices[i], n_dev);
                    return nullptr;
                }
        }
        if (!use_device(devices[0])) return nullptr;
        bgth_pbf_t *p = pbf_alloc(devices[0], hdr[0], hdr[1], hdr[2], n_total);
        if (!p) return nullptr;
        p->n_total = n_total;
        std::vector<int64_t> rg((size_t)2 * n_shards);
        bgth_shard_ranges(n_total, hdr[2], n_shards, rg.data());
        // every shard reads, uploads and checkpoints its own blocks, concurrently (different devices, or one device's queue)
        std::vector<bgth_pbf_t*> parts((size_t)n_shards, nullptr);
        std::vector<std::string> errs((size_t)n_shards);
        std::vector<std::thread> th;
        for (int i = 0; i < n_shards; ++i) {
            if (rg[2 * i] >= rg[2 * i + 1]) continue;
            th.emplace_back([&, i] {
                parts[i] = bgth_pbf_open_rows(path, rg[2 * i], rg[2 * i + 1], devices[i]);
                if (!parts[i]) errs[i] = g_err[0] ? g_err : "shard open failed";
            });
        }
        for (std::thread &t : th) t.join();
        bool ok = true;
        for (int i = 0; i < n_shards; ++i) {
            if (!errs[i].empty()) { set_err("%s", errs[i].c_str()); ok = false; }
            if (parts[i]) {
                // shards that share a device share its HBM: each one's directory arena is capped at its share of the free memory
                int same = 0;
                for (int j = 0; j < n_shards; ++j) same += parts[j] && devices[j] == devices[i] ? 1 : 0;
                parts[i]->arena_share = std::max(1, same);
                p->shards.push_back(parts[i]);
            }
        }
        if (!ok || (p->shards.empty() && n_total > 0)) { bgth_pbf_close(p); return nullptr; }
        // the parent describes its shards: their checkpoint spacing (each shard fitted its own to ITS row count; the pull
        // windows of a sharded reader are cut in the parent's units, which must be whole units of every shard) is the coarsest
        if (!p->shards.empty()) {
            int coarsest = 0;
            for (const bgth_pbf_t *sh : p->shards) coarsest = std::max(coarsest, sh->sub_shift);
            p->sub_shift = coarsest;
            set_rows(p, p->n);
        }
        return p;
    });
}
extern "C" int bgth_pbf_n_shards(const bgth_pbf_t *p) { return (int)p->shards.size(); }

extern "C" int64_t bgth_pbf_first_row(const bgth_pbf_t *p) { return p->row_off; }
extern "C" int64_t bgth_pbf_loaded_rows(const bgth_pbf_t *p) { return p->n; }

static bool derive_sub_checkpoints(bgth_pbf_t *p)
{
    if (p->sub_shift >= p->shift || p->n <= ((int64_t)1 << p->sub_shift)) return true;
    Selection all;
    bool ok = build_selection(all, p->m, 0, nullptr, nullptr, 1) && run_block_pass(p, all, 0, p->n_blk, nullptr, nullptr, 0);
    if (ok && hipDeviceSynchronize() != hipSuccess) { set_err("[E::bgth_pbf_open] sub-checkpoint pass failed"); ok = false; }
    all.release();
    return ok;
}

// Checkpoints of an image built from bare RLE strings, derived WITHOUT walking the file in order.  A row moves whatever
// sits at position R to LF(R) (pbwt.c:76-88), so what a block does to the order is a map of positions, F_b, and it does
// not depend on the order the block starts from:
//   1. every file block at once from the identity order (one launch, n_blk units): F_b = its final ranks, and the ranks at
//      its sub-checkpoints as maps of positions too;
//   2. true checkpoint of block b+1 = F_b o (true checkpoint of block b): n_blk small gathers, in order;
//   3. every sub-checkpoint re-based by one gather through its block's true checkpoint.
// (Before: one launch per block, each waiting for the one before -- 0.1 s per block at 100,000 samples.)
// Leaves the ranks after the last row in p->d_final.  BGTH_VARIANT 512 keeps the sequential form (tests compare the two).
static bool derive_all_checkpoints(bgth_pbf_t *p, Selection &all, const std::vector<int32_t> &ident)
{
    const size_t per = (size_t)2 * p->m;
    const int64_t spb = (int64_t)1 << (p->shift - p->sub_shift);           // sub-checkpoints per file block
    HIP_TRY(hipMalloc((void**)&p->d_final, per * 4), return false);
    if (p->n_blk == 0) { HIP_TRY(hipMemcpy(p->d_final, ident.data(), per * 4, hipMemcpyHostToDevice), return false); return true; }
    if (variant_flag(kVariantSeqCheckpoints) || p->n_blk == 1) {
        HIP_TRY(hipMemcpy(p->d_rank0, ident.data(), per * 4, hipMemcpyHostToDevice), return false);
        // block b's final ranks are block b+1's checkpoint: strictly sequential, one launch per block
        // (the pass over a block also leaves the block's sub-checkpoints)
        for (int64_t b = 0; b < p->n_blk; ++b)
            if (!run_block_pass(p, all, b, 1, b + 1 < p->n_blk ? p->d_rank0 + (size_t)((b + 1) * spb) * per : p->d_final, nullptr, 0)) return false;
        HIP_TRY(hipDeviceSynchronize(), return false);
        return true;
    }
    int32_t *fin = nullptr, *tru = nullptr, *rebased = nullptr;
    bool ok = false;
    do {
        HIP_TRY(hipMalloc((void**)&fin, (size_t)p->n_blk * per * 4), break);
        HIP_TRY(hipMalloc((void**)&tru, (size_t)(p->n_blk + 1) * per * 4), break);
        for (int64_t b = 0; b < p->n_blk; ++b)                             // 1. every block starts from the identity order
            HIP_TRY(hipMemcpyAsync(p->d_rank0 + (size_t)(b * spb) * per, b ? (const void*)p->d_rank0 : (const void*)ident.data(), per * 4,
                                   b ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, nullptr), goto done);
        if (!run_block_pass(p, all, 0, p->n_blk, fin, nullptr, (int64_t)per)) break;
        HIP_TRY(hipMemcpyAsync(tru, p->d_rank0, per * 4, hipMemcpyDeviceToDevice, nullptr), break);   // 2. (block 0: the identity)
        for (int64_t b = 0; b < p->n_blk; ++b)
            HIP_TRY(launch_compose(fin + (size_t)b * per, 0, tru + (size_t)b * per, 0, tru + (size_t)(b + 1) * per, 0, p->m, 1, nullptr), goto done);
        HIP_TRY(hipMemcpyAsync(p->d_final, tru + (size_t)p->n_blk * per, per * 4, hipMemcpyDeviceToDevice, nullptr), break);
        // 3. sub-checkpoint k of block b: virtual ranks through the block's true checkpoint (k = 0: the checkpoint itself)
        HIP_TRY(hipMalloc((void**)&rebased, (size_t)std::max<int64_t>(p->n_sub, 1) * per * 4), break);
        for (int64_t b = 0; b < p->n_blk; ++b) {
            const int64_t s0 = b * spb, ns = std::min<int64_t>(spb, p->n_sub - s0);
            HIP_TRY(hipMemcpyAsync(rebased + (size_t)s0 * per, tru + (size_t)b * per, per * 4, hipMemcpyDeviceToDevice, nullptr), goto done);
            if (ns > 1)
                HIP_TRY(launch_compose(p->d_rank0 + (size_t)(s0 + 1) * per, (int64_t)per, tru + (size_t)b * per, 0,
                                       rebased + (size_t)(s0 + 1) * per, (int64_t)per, p->m, ns - 1, nullptr), goto done);
        }
        HIP_TRY(hipDeviceSynchronize(), break);
        hipFree(p->d_rank0);
        p->d_rank0 = rebased; rebased = nullptr;
        ++p->rank_epoch;
        {
            std::lock_guard<std::mutex> guard(p->rowindex_lock);
            if (p->d_order) { hipFree(p->d_order); p->d_order = nullptr; }  // (derived from the checkpoints: built again on demand)
            p->order_failed = false;
        }
        ok = true;
    } while (0);
done:
    if (fin) hipFree(fin);
    if (tru) hipFree(tru);
    if (rebased) hipFree(rebased);
    return ok;
}

// The ranks (by column, [2][m]) after the last row of an image built by bgth_pbf_from_rle, and the re-basing of such an image
// onto another start order: how the shards of ONE database are opened side by side (include/bgt_hip.h).
// An image of more than two bit planes (bgth_pbf_s::pairs) serves the codec interface only; everything else says so.
static bool planes_image(const bgth_pbf_t *p, const char *who)
{
    if (!p || p->pairs.empty()) return false;
    set_err("[E::%s] an image of %d bit planes serves the codec interface (bgth_reader_select / seek / read): counts, genotype codes "
            "and checkpoints are defined for BGT's two planes", who, p->g);
    return true;
}

extern "C" int bgth_pbf_final_ranks(const bgth_pbf_t *p, int32_t *out)
{
    if (!p || !out) { set_err("[E::bgth_pbf_final_ranks] NULL argument"); return -1; }
    if (planes_image(p, "bgth_pbf_final_ranks")) return -1;
    if (!p->d_final) { set_err("[E::bgth_pbf_final_ranks] only images built by bgth_pbf_from_rle keep their final ranks"); return -1; }
    if (!use_device(p->device)) return -1;
    HIP_TRY(hipMemcpy(out, p->d_final, (size_t)2 * p->m * 4, hipMemcpyDeviceToHost), return -1);
    return 0;
}

extern "C" int bgth_pbf_ranks_at(const bgth_pbf_t *p, int64_t row, int32_t *out)
{
    if (!p || !out) { set_err("[E::bgth_pbf_ranks_at] NULL argument"); return -1; }
    if (planes_image(p, "bgth_pbf_ranks_at")) return -1;
    if (!p->shards.empty()) {
        for (const bgth_pbf_t *sh : p->shards) if (row >= sh->row_off && row < sh->row_off + sh->n) return bgth_pbf_ranks_at(sh, row, out);
        set_err("[E::bgth_pbf_ranks_at] row %lld is in no shard", (long long)row); return -1;
    }
    const int64_t r = row - p->row_off;
    if (r < 0 || r >= p->n || (r & (((int64_t)1 << p->sub_shift) - 1))) {
        set_err("[E::bgth_pbf_ranks_at] row %lld: the image keeps ranks every %lld rows of its rows [%lld,%lld)", (long long)row,
                (long long)1 << p->sub_shift, (long long)p->row_off, (long long)(p->row_off + p->n));
        return -1;
    }
    if (!use_device(p->device)) return -1;
    HIP_TRY(hipMemcpy(out, p->d_rank0 + (size_t)(r >> p->sub_shift) * 2 * p->m, (size_t)2 * p->m * 4, hipMemcpyDeviceToHost), return -1);
    return 0;
}

extern "C" int bgth_pbf_rebase(bgth_pbf_t *p, const int32_t *start_ranks)
{
    if (!p || !start_ranks) { set_err("[E::bgth_pbf_rebase] NULL argument"); return -1; }
    if (planes_image(p, "bgth_pbf_rebase")) return -1;
    if (!p->d_final) { set_err("[E::bgth_pbf_rebase] only images built by bgth_pbf_from_rle can be re-based"); return -1; }
    if (!use_device(p->device)) return -1;
    const size_t per = (size_t)2 * p->m;
    {   // Both planes must be PERMUTATIONS of 0 .. m-1: a rank out of range would address outside the tables, and a repeated one
        // leaves slots of the plane-1-by-plane-0 table unwritten (the rank-ordered start of whole-cohort scans scatters by rank:
        // the default path and BGTH_FORCE_COLUMN_ORDER would then silently disagree)
        std::vector<uint8_t> seen((size_t)p->m);
        for (int k = 0; k < 2; ++k) {
            std::fill(seen.begin(), seen.end(), 0);
            for (int j = 0; j < p->m; ++j) {
                const int32_t v = start_ranks[(size_t)k * p->m + j];
                if (v < 0 || v >= p->m) { set_err("[E::bgth_pbf_rebase] rank %d out of range", v); return -1; }
                if (seen[v]) { set_err("[E::bgth_pbf_rebase] plane %d: rank %d appears twice: not a permutation", k, v); return -1; }
                seen[v] = 1;
            }
        }
    }
    int32_t *via = nullptr, *out = nullptr;
    const int64_t n = std::max<int64_t>(p->n_sub, 1);
    int rc = -1;
    do {
        HIP_TRY(hipMalloc((void**)&via, per * 4), break);
        HIP_TRY(hipMalloc((void**)&out, (size_t)(n + 1) * per * 4), break);
        HIP_TRY(hipMemcpy(via, start_ranks, per * 4, hipMemcpyHostToDevice), break);
        HIP_TRY(launch_compose(p->d_rank0, (int64_t)per, via, 0, out, (int64_t)per, p->m, n, nullptr), break);
        HIP_TRY(launch_compose(p->d_final, 0, via, 0, out + (size_t)n * per, 0, p->m, 1, nullptr), break);
        HIP_TRY(hipMemcpy(p->d_final, out + (size_t)n * per, per * 4, hipMemcpyDeviceToDevice), break);
        HIP_TRY(hipDeviceSynchronize(), break);
        {   // readers of the image keep no ranks of their own; an arena a reader filled holds directory rows, which do not depend on the order
            std::lock_guard<std::mutex> guard(p->rowindex_lock);           // (d_order is built under this lock by the scans)
            hipFree(p->d_rank0);
            p->d_rank0 = out; out = nullptr;
            ++p->rank_epoch;
            if (p->d_order) { hipFree(p->d_order); p->d_order = nullptr; }
            p->order_failed = false;
        }
        rc = 0;
    } while (0);
    if (via) hipFree(via);
    if (out) hipFree(out);
    return rc;
}

static bgth_pbf_t *from_rle_impl(int m, int g, int shift, int64_t n_rows, const uint8_t *rle, const uint32_t *len, int device);
extern "C" bgth_pbf_t *bgth_pbf_from_rle(int m, int g, int shift, int64_t n_rows, const uint8_t *rle,
                                         const uint32_t *len, int device)
{
    return guarded("bgth_pbf_from_rle", (bgth_pbf_t*)nullptr, [&] { return from_rle_impl(m, g, shift, n_rows, rle, len, device); });
}

static bgth_pbf_t *from_rle_impl(int m, int g, int shift, int64_t n_rows, const uint8_t *rle, const uint32_t *len, int device)
{
    if (!use_device(device)) return nullptr;
    bgth_pbf_t *p = pbf_alloc(device, m, g, shift, n_rows);
    if (!p) return nullptr;
    if (p->wide_plane) {
        set_err("[E::bgth_pbf_from_rle] m=%d columns: checkpoints are derived by the two-plane kernels (up to 327,000 haplotypes); "
                "wider cohorts are read from files, which carry their 'S' records", m);
        bgth_pbf_close(p);
        return nullptr;
    }
    t_building = p;
    p->n_total = n_rows;
    std::vector<uint64_t> desc((size_t)n_rows * g);
    std::vector<uint8_t> packed;
    uint64_t off = 0, src = 0;
    {
        uint64_t total = 0;
        for (size_t i = 0; i < desc.size(); ++i) total += ((uint64_t)len[i] + 3) & ~(uint64_t)3;
        packed.resize(total);
    }
    for (size_t i = 0; i < desc.size(); ++i) {
        if (len[i] >= (1u << 24)) { set_err("[E::bgth_pbf_from_rle] string %zu too long", i); t_building = nullptr; bgth_pbf_close(p); return nullptr; }
        desc[i] = off | (uint64_t)len[i] << kDescLenShift;
        memcpy(packed.data() + off, rle + src, len[i]);
        if ((i & 1) && string_is_all_zero(rle + src, len[i])) ++p->n_empty1;
        src += len[i];
        off += ((uint64_t)len[i] + 3) & ~(uint64_t)3;           // kernels read whole aligned dwords
    }
    p->rle_bytes = (int64_t)src;
    p->packed_bytes = (int64_t)off;
    rle = packed.data();
    if (p->n_empty1) { fit_sub_shift(p, n_rows); set_rows(p, n_rows); }   // (the kernel family, and with it the spacing, depends on the empty plane-1 rows)
    Selection all;
    {
        const size_t pad = 256;
        HIP_TRY(hipMalloc((void**)&p->d_rle, off + pad), goto fail);
        HIP_TRY(hipMemset(p->d_rle + off, 0, pad), goto fail);
        if (off) HIP_TRY(hipMemcpy(p->d_rle, rle, off, hipMemcpyHostToDevice), goto fail);
        HIP_TRY(hipMalloc((void**)&p->d_rowdesc, std::max<size_t>(desc.size(), 1) * 8), goto fail);
        if (!desc.empty()) HIP_TRY(hipMemcpy(p->d_rowdesc, desc.data(), desc.size() * 8, hipMemcpyHostToDevice), goto fail);
        const size_t per = (size_t)2 * m;
        if (!rank0_alloc(p)) goto fail;
        std::vector<int32_t> ident(per);
        for (int k = 0; k < 2; ++k) for (int j = 0; j < m; ++j) ident[(size_t)k * m + j] = j;   // ref pbwt.c:103
        if (!build_selection(all, m, 0, nullptr, nullptr, 1)) goto fail;
        if (!derive_all_checkpoints(p, all, ident)) goto fail;
        HIP_TRY(hipDeviceSynchronize(), goto fail);
    }
    all.release();
    t_building = nullptr;
    return p;
fail:
    all.release();
    t_building = nullptr;
    bgth_pbf_close(p);
    return nullptr;
}

static int64_t save_impl(const bgth_pbf_t *p, const char *path);
extern "C" int64_t bgth_pbf_save(const bgth_pbf_t *p, const char *path)
{
    return guarded("bgth_pbf_save", (int64_t)-1, [&] { return save_impl(p, path); });
}

static int64_t save_impl(const bgth_pbf_t *p, const char *path)
{
    if (!p) return -1;
    if (p->row_off != 0 || p->n != p->n_total || !p->shards.empty()) { set_err("[E::bgth_pbf_save] a partial or sharded image cannot be saved"); return -1; }
    if (!p->pairs.empty()) {                                     // more than two planes: the file's own bytes (nothing can have changed them)
        FILE *fp = fopen(path, "wb");
        if (!fp) { set_err("[E::bgth_pbf_save] cannot create '%s'", path); return -1; }
        const bool ok = fwrite(p->file_image.data(), 1, p->file_image.size(), fp) == p->file_image.size();
        if (fclose(fp) != 0 || !ok) { set_err("[E::bgth_pbf_save] writing '%s' failed", path); return -1; }
        return (int64_t)p->file_image.size();
    }
    if (!use_device(p->device)) return -1;
    const int m = p->m;
    const size_t per = (size_t)2 * m;
    std::vector<uint8_t> rle((size_t)p->packed_bytes);
    std::vector<uint64_t> desc((size_t)p->n * 2);
    std::vector<int32_t> perm(per);
    int32_t *d_perm = nullptr;
    FILE *fp = fopen(path, "wb");
    if (!fp) { set_err("[E::bgth_pbf_save] cannot create '%s'", path); return -1; }
    std::vector<uint64_t> idx;
    int64_t written = -1;
    {
        if (!rle.empty()) HIP_TRY(hipMemcpy(rle.data(), p->d_rle, rle.size(), hipMemcpyDeviceToHost), goto done);
        if (!desc.empty()) HIP_TRY(hipMemcpy(desc.data(), p->d_rowdesc, desc.size() * 8, hipMemcpyDeviceToHost), goto done);
        HIP_TRY(hipMalloc((void**)&d_perm, per * 4), goto done);
        const int gw = p->g_file ? p->g_file : 2;             // planes written (a one-plane file gets its one plane back)
        int32_t hdr[3] = {p->m, gw, p->shift};
        fwrite("PBF\1", 1, 4, fp); fwrite(hdr, 4, 3, fp);
        for (int64_t r = 0; r < p->n; ++r) {
            if ((r & (((int64_t)1 << p->shift) - 1)) == 0) {
                const int64_t b = r >> p->sub_shift;                    // the sub-checkpoint at a file block's first row
                HIP_TRY(launch_invert(p->d_rank0 + (size_t)b * per, d_perm, m, 2, nullptr), goto done);
                HIP_TRY(hipMemcpy(perm.data(), d_perm, per * 4, hipMemcpyDeviceToHost), goto done);
                idx.push_back((uint64_t)ftell(fp));
                fputc('S', fp);
                fwrite(perm.data(), 4, (size_t)gw * m, fp);
            }
            fputc('B', fp);
            for (int k = 0; k < gw; ++k) {
                const uint64_t d = desc[(size_t)r * 2 + k];
                const int32_t l = (int32_t)(d >> kDescLenShift);
                fwrite(&l, 4, 1, fp);
                fwrite(rle.data() + (d & kDescOffMask), 1, (size_t)l, fp);
            }
        }
        const uint64_t off = (uint64_t)ftell(fp);
        const int32_t n_idx = (int32_t)idx.size();
        fputc('I', fp);
        fwrite(&p->n, 8, 1, fp); fwrite(&n_idx, 4, 1, fp);
        fwrite(idx.data(), 8, idx.size(), fp); fwrite(&off, 8, 1, fp);
        written = ftell(fp);
    }
done:
    if (d_perm) hipFree(d_perm);
    fclose(fp);
    return written;
}

extern "C" int bgth_pbf_get_m(const bgth_pbf_t *p) { return p->m; }
extern "C" int bgth_pbf_get_g(const bgth_pbf_t *p) { return p->g_file ? p->g_file : p->g; }
extern "C" int bgth_pbf_get_shift(const bgth_pbf_t *p) { return p->shift; }
extern "C" int64_t bgth_pbf_unit_rows(const bgth_pbf_t *p) { const bgth_pbf_t *q = !p->pairs.empty() ? p->pairs[0] : p->shards.empty() ? p : p->shards[0]; return (int64_t)1 << q->sub_shift; }
extern "C" int64_t bgth_pbf_get_n(const bgth_pbf_t *p) { return p->n_total; }
extern "C" int64_t bgth_pbf_rle_bytes(const bgth_pbf_t *p) { return p->rle_bytes; }
extern "C" int64_t bgth_pbf_hbm_bytes(const bgth_pbf_t *p)
{
    if (!p->shards.empty()) { int64_t t = 0; for (const bgth_pbf_t *sh : p->shards) t += bgth_pbf_hbm_bytes(sh); return t; }
    if (!p->pairs.empty()) { int64_t t = 0; for (const bgth_pbf_t *pp : p->pairs) t += bgth_pbf_hbm_bytes(pp); return t; }
    return p->packed_bytes + 256 + p->n * 2 * 8 + p->n_sub * 2 * (int64_t)p->m * 4 + p->rowindex_bytes +
           (p->d_order ? p->n_sub * (int64_t)p->m * 4 : 0);
}

// ----------------------------------------------------------------------------------------------------
// reader
// ----------------------------------------------------------------------------------------------------
// frees everything a reader owns (its shard readers are already gone)
static void reader_free(bgth_reader_t *r)
{
    hipSetDevice(r->pbf->device);
    if (r->stream) hipStreamSynchronize(r->stream);
    r->sel.release();
    r->raw.release(); r->fin.release(); r->h0.release(); r->h1.release(); r->gt.release();
    r->carriers.release(); r->hapsig.release();
    r->win[0].release(); r->win[1].release();
    r->dir.release(); r->dir_n0.release(); r->tog_mem.release(); r->start_tab.release();
    r->ph0.release(); r->ph1.release();
    for (int i = 0; i < 4; ++i) if (r->ev[i]) hipEventDestroy(r->ev[i]);
    for (int i = 0; i < 2; ++i) if (r->ev_dir[i]) hipEventDestroy(r->ev_dir[i]);
    if (r->ev_gather) hipEventDestroy(r->ev_gather);
    if (r->ev_copied) hipEventDestroy(r->ev_copied);
    if (r->stream) hipStreamDestroy(r->stream);
    delete r;
}

constexpr size_t kReaderPoolMax = 32;          // per image; more concurrent readers than that are created and freed as before

extern "C" bgth_reader_t *bgth_reader_create(bgth_pbf_t *p)
{
    if (!p) { set_err("[E::bgth_reader_create] NULL image"); return nullptr; }
    if (!use_device(p->device)) return nullptr;
    if (!p->pairs.empty()) {                                     // more than two planes: a reader per pair of planes behind one handle
        bgth_reader_t *w = new bgth_reader_s();
        w->pbf = p;
        w->sel.width = p->m;
        for (bgth_pbf_t *pp : p->pairs) {
            bgth_reader_t *sub = bgth_reader_create(pp);
            if (!sub) { bgth_reader_destroy(w); return nullptr; }
            w->pair_readers.push_back(sub);
        }
        w->multi_ret.assign((size_t)p->g, nullptr);
        return w;
    }
    bgth_reader_t *r = nullptr;
    {
        std::lock_guard<std::mutex> g(p->pool_lock);
        if (!p->pool.empty()) { r = p->pool.back(); p->pool.pop_back(); }
    }
    if (r) {
        // a pooled reader: stream, events, device and pinned buffers as they were; every piece of per-use state back to
        // what a new reader has
        HIP_TRY(hipStreamSynchronize(r->stream), { reader_free(r); return nullptr; });
        r->t_ms[0] = r->t_ms[1] = r->t_ms[2] = 0.f; r->t_pending = false;
        r->tune_threads = r->tune_cpt = r->tune_K = 0;
        r->next = r->ring0 = r->ring1 = 0; r->ring_has = 0;
        r->want = BGTH_WANT_PLANES; r->max_ahead = 0; r->ahead = 0;
        r->ret[0] = r->ret[1] = nullptr; r->last_counts = nullptr; r->last_gt8 = nullptr; r->last_gttext = nullptr;
        r->folds_live = false;
        r->cur = 0;
        r->dir_lo = r->dir_hi = 0;                  // (a selection change does not invalidate the arena, a new query may not rely on it)
        for (PullWindow &w : r->win) w.valid = w.pending = false;
    } else {
        r = new bgth_reader_s();
        r->pbf = p;
        HIP_TRY(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking), { reader_free(r); return nullptr; });
        for (int i = 0; i < 4; ++i) HIP_TRY(hipEventCreate(&r->ev[i]), { reader_free(r); return nullptr; });
        for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreate(&r->ev_dir[i]), { reader_free(r); return nullptr; });
        HIP_TRY(hipEventCreateWithFlags(&r->ev_gather, hipEventDisableTiming), { reader_free(r); return nullptr; });
        HIP_TRY(hipEventCreateWithFlags(&r->ev_copied, hipEventDisableTiming), { reader_free(r); return nullptr; });
    }
    if (!guarded("bgth_reader_create", false, [&] { return build_selection(r->sel, p->m, 0, nullptr, nullptr, 1); })) { reader_free(r); return nullptr; }
    for (bgth_pbf_t *sh : p->shards) {
        bgth_reader_t *sub = bgth_reader_create(sh);
        if (!sub) { bgth_reader_destroy(r); return nullptr; }
        r->subs.push_back(sub);
        r->workers.push_back(new ShardWorker(sh->device));
    }
    return r;
}

// A destroyed reader goes back to its image's pool (up to kReaderPoolMax of them) with everything it has allocated; the
// image frees the pool when it is closed.  Readers must be destroyed before their image, as before.
extern "C" void bgth_reader_destroy(bgth_reader_t *r)
{
    if (!r) return;
    if (!r->pbf->pairs.empty()) {                                // (the handle over the pair readers owns nothing on the device)
        for (bgth_reader_t *sub : r->pair_readers) bgth_reader_destroy(sub);
        delete r;
        return;
    }
    for (ShardWorker *w : r->workers) delete w;
    r->workers.clear();
    for (bgth_reader_t *sub : r->subs) bgth_reader_destroy(sub);
    r->subs.clear();
    {   // a reader that served a big streaming query does not keep its windows (up to 2 x 256 MiB pinned + as much HBM)
        size_t held = 0;
        for (const PullWindow &w : r->win) held += w.h_counts.cap + w.h_planes.cap + w.h_gt8.cap + w.h_gttext.cap;
        if (held > ((size_t)64 << 20) || r->dir.cap > ((size_t)256 << 20) || r->ph0.cap > ((size_t)256 << 20) || r->start_tab.cap > ((size_t)64 << 20)) {
            hipSetDevice(r->pbf->device);
            if (r->stream) hipStreamSynchronize(r->stream);
            if (r->start_tab.cap > ((size_t)64 << 20)) { r->start_tab.release(); r->start_epoch = -1; }   // (a selection's compact start ranks: C3 314 MB)
            if (held > ((size_t)64 << 20)) { r->win[0].release(); r->win[1].release(); }
            if (r->dir.cap > ((size_t)256 << 20)) { r->dir.release(); r->dir_n0.release(); r->dir_lo = r->dir_hi = 0; }
            if (r->ph0.cap > ((size_t)256 << 20)) { r->ph0.release(); r->ph1.release(); }
        }
    }
    {   // (not waiting for the stream: a window prefetched behind the last row read may still be on its way -- it lands in
        //  buffers the pooled reader keeps, and whoever takes the reader next drains the stream first)
        std::lock_guard<std::mutex> g(r->pbf->pool_lock);
        if (r->pbf->pool.size() < kReaderPoolMax) { r->pbf->pool.push_back(r); return; }
    }
    reader_free(r);
}

extern "C" int bgth_reader_select(bgth_reader_t *r, int n_sub, const int32_t *sub, const uint32_t *group,
                                  int n_groups)
{
    if (!r) return -1;
    if (!r->pair_readers.empty()) {
        for (bgth_reader_t *pr : r->pair_readers) if (bgth_reader_select(pr, n_sub, sub, group, n_groups) < 0) return -1;
        r->sel.width = r->pair_readers[0]->sel.width;
        return 0;
    }
    if (!use_device(r->pbf->device)) return -1;
    hipStreamSynchronize(r->stream);
    if (!guarded("bgth_reader_select", false, [&] { return build_selection(r->sel, r->pbf->m, n_sub, sub, group, n_groups); })) return -1;
    r->start_epoch = -1;              // (the compact start ranks belong to the selection before)
    r->ring0 = r->ring1 = 0;          // invalidate the pull ring (the stream is idle: nothing is pending any more)
    for (PullWindow &w : r->win) w.valid = w.pending = false;
    r->folds_live = false;
    for (bgth_reader_t *sr : r->subs) if (bgth_reader_select(sr, n_sub, sub, group, n_groups) < 0) return -1;
    return 0;
}

extern "C" int bgth_reader_width(const bgth_reader_t *r) { return r->sel.width; }
extern "C" int bgth_reader_slot_words(const bgth_reader_t *r) { return r->pair_readers.empty() ? r->sel.n_chunks : r->pair_readers[0]->sel.n_chunks; }
extern "C" int bgth_reader_slot_map(const bgth_reader_t *r, int32_t *out)
{
    if (!r->pair_readers.empty()) return bgth_reader_slot_map(r->pair_readers[0], out);
    memcpy(out, r->sel.slot_of_out.data(), (size_t)r->sel.width * 4);
    return r->sel.width;
}

extern "C" int bgth_reader_tune(bgth_reader_t *r, int threads, int cpt, int K)
{
    r->tune_threads = threads; r->tune_cpt = cpt; r->tune_K = K;
    for (bgth_reader_t *sub : r->subs) bgth_reader_tune(sub, threads, cpt, K);
    for (bgth_reader_t *pr : r->pair_readers) bgth_reader_tune(pr, threads, cpt, K);
    return 0;
}

static int gx_of(int G) { return G > 1 ? G : 0; }


// communicator of a sharded image over its distinct devices; false (error set) if RCCL cannot be used
static bool ensure_comm(bgth_pbf_t *p)
{
    std::lock_guard<std::mutex> g(p->comm_lock);
    if (!p->comm.empty()) return true;
    if (!rccl().ok) { set_err("[E::bgth] librccl.so could not be loaded: the shards of this image are on several devices and their counts are gathered over RCCL"); return false; }
    std::vector<int> devs;
    for (const bgth_pbf_t *sh : p->shards) if (std::find(devs.begin(), devs.end(), sh->device) == devs.end()) devs.push_back(sh->device);
    std::vector<void*> comm(devs.size(), nullptr);
    RCCL_TRY(rccl().CommInitAll(comm.data(), (int)devs.size(), devs.data()), return false);
    p->comm = comm; p->comm_dev = devs;
    return true;
}
static int comm_rank(const bgth_pbf_t *p, int device)
{
    for (size_t i = 0; i < p->comm_dev.size(); ++i) if (p->comm_dev[i] == device) return (int)i;
    return -1;
}

// ---- directory path (scan_dir.hip) ------------------------------------------------------------------
// Bytes the arena of a reader may take: BGTH_DIR_ARENA_MB, default 60 % of the HBM that is free when first asked.
static size_t dir_arena_cap(int device)
{
    if (const char *e = getenv("BGTH_DIR_ARENA_MB")) return (size_t)std::max(1ll, atoll(e)) << 20;
    static std::mutex lock;
    static std::vector<size_t> cap;
    std::lock_guard<std::mutex> g(lock);
    if ((size_t)device >= cap.size()) cap.resize(device + 1, 0);
    if (!cap[device]) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = (size_t)8 << 30;
        cap[device] = fr / 10 * 6;
    }
    return cap[device];
}

// Should this scan build its rows once into the arena and walk them from there?  Yes for a wide cohort whose columns
// span several workgroups (every slice of the team kernels rebuilds the row); BGTH_VARIANT 32 / 64 force / forbid it.
// (round 4: also for ONE column slice when most of the cohort is selected -- m = 34,000: 11.7 ms against the team kernels' 13.2 ms per
// 524,288 sites; the producer's arena is small there and the walk-only workgroup has no build phases to idle through)
static bool want_dir_path(const bgth_pbf_t *p, const Geometry &classic, bool tuned, int width)
{
    if (variant_flag(kVariantDirNever)) return false;
    if (variant_flag(kVariantDirAlways)) return true;
    if (use_zp(p)) return false;                         // (the walk-only kernel has no empty-plane shortcut: it looks plane 1 up)
    if (tuned) return false;                             // bgth_reader_tune names a classic geometry
    // (... and most of a cohort whose pipelined geometry needs two column slices, each repeating the row build: m = 26,000 as
    // 1024 x 16 x 2 runs at 2.83 T lookups/s, on the directory path at 3.16; m = 30,000: 3.03 / 3.14)
    // (more than 24,576 columns: the slices are not those a SHORT scan is cut into to fill the chip)
    if (classic.slices >= 2 && width > 24576 && 2 * (int64_t)width >= p->m) return true;
    // (round 5: one slice of 1024 threads -- four waves per SIMD -- with >= 4 rows per batch keeps the scan kernels: m = 32,768 as
    // 1024 x 32, K 4 runs at 4.20 T lookups/s, on the directory path at 3.70)
    if (classic.threads == 1024 && classic.slices == 1 && classic.K >= 4) return false;
    return classic.nbuf == 1 && classic.wpp > 1 && (classic.slices >= 2 || 2 * (int64_t)width >= p->m);
}


// Cohorts whose two bit-vectors do not fit the LDS together (bgth_pbf_s::wide_plane): producer + one walk-only workgroup
// per (sub-block, column slice, PLANE), the planes joined from their ballots (scan_plane.hip).  Passes over ranges of
// sub-blocks sized so that the directory arena and, when the caller takes counts only, the passes' own bit planes fit.
static int64_t enqueue_scan_wide_plane(bgth_reader_t *r, int64_t row0, int64_t row1, int32_t *d_fin, uint64_t *d_h0,
                                       uint64_t *d_h1, hipStream_t s, bool timed)
{
    bgth_pbf_t *p = r->pbf;
    const int64_t rows = row1 - row0;
    const int G = r->sel.G;
    const int64_t blk0 = row0 >> p->sub_shift, blk1 = (row1 - 1) >> p->sub_shift;
    Geometry wg;
    if (p->mem_plane) choose_walk_mem_geometry(p->m, r->sel.n_chunks, (int)(blk1 - blk0 + 1), &wg);
    else if (!choose_walk_plane_geometry(p->m, r->sel.n_chunks, (int)(blk1 - blk0 + 1), &wg)) { set_err("[E::bgth_reader_scan] no launch geometry for m=%d", p->m); return -1; }
    r->geom = wg; r->plane_path = 1; r->dir_passes = r->dir_built = 0;
    if (!r->raw.reserve((size_t)rows * G * 3 * 4)) { set_err("[E::bgth_reader_scan] out of HBM"); return -1; }
    ScanArgs a;
    { Geometry team = wg; team.wpp = 2; if (!common_scan_args(a, p, r->sel, team, s)) return -1; }
    a.wpp = wg.wpp; a.n_slices = wg.slices;
    a.shift = p->sub_shift;
    a.rank0_blk_stride = (int64_t)2 * p->m;
    a.row0 = row0; a.row1 = row1;
    const int nwp = ((p->m + 31) / 32 + 2) & ~1;
    const size_t plane_row = (size_t)r->sel.n_chunks * 8;
    const size_t row_bytes = (size_t)2 * nwp * 8 + (d_h0 ? 0 : 2 * plane_row);
    const int64_t sub_rows = (int64_t)1 << p->sub_shift;
    const int64_t fit = (int64_t)((dir_arena_cap(p->device) / (size_t)p->arena_share) / (row_bytes * (size_t)std::min<int64_t>(sub_rows, row1 - (blk0 << p->sub_shift))));
    if (fit < 1) { set_err("[E::bgth_reader_scan] the directory arena does not hold one block of m=%d", p->m); return -1; }
    int64_t per_pass = std::min<int64_t>(blk1 - blk0 + 1, fit >= 8 ? fit / 8 * 8 : fit);
    const int64_t pass_rows = std::min<int64_t>(per_pass * sub_rows, row1 - (blk0 << p->sub_shift));
    if (!r->dir.reserve((size_t)pass_rows * 2 * nwp * 8) || !r->dir_n0.reserve((size_t)pass_rows * 2 * 4) ||
        (!d_h0 && (!r->ph0.reserve((size_t)pass_rows * plane_row) || !r->ph1.reserve((size_t)pass_rows * plane_row)))) {
        set_err("[E::bgth_reader_scan] out of HBM (directory arena of %lld rows)", (long long)pass_rows);
        return -1;
    }
    r->dir_lo = r->dir_hi = 0;
    a.dir = (uint2*)r->dir.p; a.dir_n0 = (uint32_t*)r->dir_n0.p; a.dir_nwp = nwp; a.dir_stage = wg.dir_stage;
    a.tog_mem = nullptr;
    if (p->mem_plane) {                                  // the producer's toggle words live in memory: two arrays per workgroup
        if (!r->tog_mem.reserve((size_t)dirbuild_mem_workgroups(pass_rows) * 2 * (size_t)dirbuild_mem_words(p->m) * 4)) { set_err("[E::bgth_reader_scan] out of HBM (toggle words)"); return -1; }
        a.tog_mem = (uint32_t*)r->tog_mem.p;
    }
    if (timed) { HIP_TRY(hipEventRecord(r->ev[0], s), return -1); HIP_TRY(hipEventRecord(r->ev[1], s), return -1); }
    for (int64_t b = blk0; b <= blk1; b += per_pass) {
        const int64_t be = std::min(blk1 + 1, b + per_pass);
        const int64_t lo = b << p->sub_shift, hi = std::min(row1, be << p->sub_shift), e0 = std::max(lo, row0);
        Geometry pg = wg;
        pg.workgroups = (int)(((be - b) + 7) / 8 * 16 * wg.slices);
        a.blk0 = (int32_t)b; a.n_blk = (int32_t)(be - b); a.dir_row0 = lo;
        if (d_h0) { a.h0 = d_h0; a.h1 = d_h1; a.h_row0 = row0; }
        else { a.h0 = (uint64_t*)r->ph0.p; a.h1 = (uint64_t*)r->ph1.p; a.h_row0 = e0; }
        if (timed && b == blk0) HIP_TRY(hipEventRecord(r->ev_dir[0], s), return -1);
        HIP_TRY(p->mem_plane ? launch_dirbuild_mem(a, lo, hi, s) : launch_dirbuild(a, lo, hi, s), return -1);
        if (timed && b == blk0) HIP_TRY(hipEventRecord(r->ev_dir[1], s), return -1);
        ++r->dir_built;
        HIP_TRY(p->mem_plane ? launch_walk_mem(a, pg, s) : launch_walk_plane(a, pg, s), return -1);
        const size_t hoff = d_h0 ? (size_t)(e0 - row0) * r->sel.n_chunks : 0;
        HIP_TRY(launch_count_planes(a.h0 + hoff, a.h1 + hoff, r->sel.d_chunk_desc, (int32_t*)r->raw.p + (size_t)(e0 - row0) * G * 3,
                                    hi - e0, r->sel.n_chunks, G, s), return -1);
        ++r->dir_passes;
    }
    if (timed) HIP_TRY(hipEventRecord(r->ev[2], s), return -1);
    HIP_TRY(launch_finalize((const int32_t*)r->raw.p, d_fin, r->sel.d_group_haps, rows, G, s), return -1);
    if (timed) HIP_TRY(hipEventRecord(r->ev[3], s), return -1);
    return rows;
}

static void collect_timing(bgth_reader_t *r);
// enqueue decode+reduce of [row0,row1) on stream s; results in d_fin (+ optional planes)
static int64_t enqueue_scan(bgth_reader_t *r, int64_t row0, int64_t row1, int32_t *d_fin, uint64_t *d_h0,
                            uint64_t *d_h1, hipStream_t s, bool timed)
{
    bgth_pbf_t *p = r->pbf;
    if (!p->shards.empty()) { set_err("[E::bgth_reader_scan_device] a sharded image spans several devices: use bgth_reader_scan or the pull interface"); return -1; }
    if (row0 < 0 || row1 > p->n || row0 > row1) { set_err("[E::bgth_reader_scan] rows [%lld,%lld) outside 0..%lld", (long long)row0, (long long)row1, (long long)p->n); return -1; }
    const int64_t rows = row1 - row0;
    if (rows == 0) return 0;
    if (p->wide_plane) return enqueue_scan_wide_plane(r, row0, row1, d_fin, d_h0, d_h1, s, timed);
    const int G = r->sel.G;
    const int64_t blk0 = row0 >> p->sub_shift, blk1 = (row1 - 1) >> p->sub_shift;      // sub-blocks
    Geometry geo;
    if (!choose_geometry(p->m, r->sel.n_chunks, G, (int)(blk1 - blk0 + 1), r->tune_threads, r->tune_cpt, r->tune_K, &geo, !variant_flag(kVariantNoTog))) {
        set_err("[E::bgth_reader_scan] no launch geometry for m=%d (threads=%d cpt=%d)", p->m, r->tune_threads, r->tune_cpt);
        return -1;
    }
    bool dirpath = want_dir_path(p, geo, r->tune_threads || r->tune_cpt || r->tune_K, r->sel.width);
    Geometry wgeo;
    if (dirpath) {
        int wt = 0, wc = 0;
#ifdef BGTH_ABLATE
        if (const char *e = getenv("BGTH_WALK_GEOM")) sscanf(e, "%d,%d", &wt, &wc);   // threads,cols: tuning knob of the walk-only kernel
#endif
        if (!choose_walk_geometry(p->m, r->sel.n_chunks, G, (int)(blk1 - blk0 + 1), wt, wc, &wgeo)) dirpath = false;
        else if (variant_flag(kVariantThreeBuffers) && (wgeo.dir_stage & 4)) wgeo.dir_stage = 1;   // (test hook: the LDS of four holds three)
        // the team kernels also slice the columns of a SHORT scan to fill the chip; the directory path is for selections whose
        // columns do not fit one workgroup (every slice then repeats the build), not for those
        else if (wgeo.slices < 2 && 2 * (int64_t)r->sel.width < p->m && !variant_flag(kVariantDirAlways)) dirpath = false;
    }
    // Plane-split kernels: a selection of few columns of a WIDE cohort (the team kernels would run one workgroup per CU,
    // mostly building): one workgroup per plane, two per CU.  BGTH_VARIANT 2048 / 4096 forbid / force them.  Team kernels that
    // batch rows (K > 1: m = 40,000 ... 100,000) keep selections of more than a quarter of the cohort (scripts/subset_ab.py, with
    // three workgroups per CU: every 13th of 64,976 columns 1.15 against 1.90 ms, every 6th 1.95 against 2.13, every 4th 1.99
    // against 2.41; every 10th of 100,000 5.7 against 8.4 ms; every 4th of 40,000 4.91 against 5.05).
    Geometry pgeo;
    bool planepath = !dirpath && !variant_flag(kVariantPlaneNever) && !(r->tune_threads || r->tune_cpt || r->tune_K) &&
                     ((geo.nbuf == 1 && geo.wpp > 1 && (geo.K == 1 || 4 * (int64_t)r->sel.width <= p->m)) || variant_flag(kVariantPlaneAlways)) &&
                     choose_plane_geometry(p->m, r->sel.n_chunks, (int)(blk1 - blk0 + 1), &pgeo);
    uint64_t *p_h0 = d_h0, *p_h1 = d_h1;
    if (planepath && !d_h0) {
        const size_t pl = (size_t)rows * r->sel.n_chunks * 8;
        if (!r->ph0.reserve(pl) || !r->ph1.reserve(pl)) planepath = false;        // (no room for the planes: the team kernels count in place)
        else { p_h0 = (uint64_t*)r->ph0.p; p_h1 = (uint64_t*)r->ph1.p; }
    }
    r->geom = dirpath ? wgeo : planepath ? pgeo : geo;
    r->plane_path = planepath;
    r->dir_passes = r->dir_built = 0;
    if (!r->raw.reserve((size_t)rows * G * 3 * 4)) { set_err("[E::bgth_reader_scan] out of HBM"); return -1; }
    ScanArgs a;
    if (dirpath) { Geometry team = wgeo; team.wpp = 2; if (!common_scan_args(a, p, r->sel, team, s)) return -1; a.wpp = wgeo.wpp; }   // (wpp > 1: row index)
    else if (planepath) { if (!common_scan_args(a, p, r->sel, pgeo, s)) return -1; }
    else if (!common_scan_args(a, p, r->sel, geo, s)) return -1;
    a.shift = p->sub_shift;                              // units = sub-blocks
    a.rank0_blk_stride = (int64_t)2 * p->m;
    if (planepath && !r->sel.whole && !variant_flag(kVariantColumnOrder)) {
        // A sparse selection's start ranks, gathered once per selection into slot order (gather_start_ranks_kernel): a workgroup then
        // starts with one contiguous read instead of T gathers of 4 bytes per 64-byte line.  Up to 1/16 of the free HBM (C3: 314 MB).
        const int n_slots = r->sel.n_chunks * 64;
        const int64_t n_rec = std::max<int64_t>(p->n_sub, 1);
        if (r->start_epoch != p->rank_epoch || r->start_n_sub != n_rec || r->start_slots != n_slots) {
            size_t fr = 0, tot = 0;
            const size_t need = (size_t)n_rec * 2 * n_slots * 4;
            r->start_epoch = -1;
            if ((need <= r->start_tab.cap || (hipMemGetInfo(&fr, &tot) == hipSuccess && need <= fr / 16)) && r->start_tab.reserve(need)) {
                HIP_TRY(launch_gather_start_ranks(p->d_rank0, r->sel.d_slot_col, (int32_t*)r->start_tab.p, p->m, n_slots, n_rec, (int32_t)(32 * ((p->m + 31) / 32)), s), return -1);
                HIP_TRY(hipStreamSynchronize(s), return -1);                 // (a later scan may come on another stream; once per selection)
                r->start_epoch = p->rank_epoch; r->start_n_sub = n_rec; r->start_slots = n_slots;
            } else (void)hipGetLastError();
        }
        if (r->start_epoch == p->rank_epoch) { a.start_slots = (const int32_t*)r->start_tab.p; a.start_blk_stride = (int64_t)2 * n_slots; }
    }
    // Whole cohort, one group, counts only: the counts do not care which lane tracks which column, so slot s of a sub-block tracks
    // the column whose plane-0 rank at its checkpoint is s.  The 64 lanes of a wave then start on 64 consecutive ranks, PBWT order
    // keeps neighbours together for a while, and a wave's ds_read_b64 gather hits few distinct entries of the plane-0 row instead of
    // 64 random ones: LDS bank-conflict cycles -40 % at the HRC shape (sub-blocks of 128 rows), -20 % on one C4 shard, -3 % on C2
    // (2048 rows); kernel time -5.2 % / -2.3 % / -0.8 % (profiles/r05_lds).  BGTH_FORCE_COLUMN_ORDER keeps the slots in column order.
    if (r->sel.whole && G == 1 && !d_h0 && !planepath && !variant_flag(kVariantColumnOrder)) {
        a.whole_counts = 1;                              // ... and only n(code 3) is counted: the planes' ones are the rows' own (BGTH_COUNT3)
        a.pk16 = variant_flag(kVariantPackedRanks) && p->m <= 65504 ? 1 : 0;
        std::lock_guard<std::mutex> guard(p->rowindex_lock);
        if (!p->d_order && !p->order_failed) {
            // Built on the stream of this scan and PUBLISHED only once it is complete: another reader's stream may use it only after
            // this launch (the lock's holder synchronises), and a launch or sync that fails leaves no half-filled table behind for
            // later scans to start plane-1 ranks from -- they fall back to the column order.
            int32_t *order = nullptr;
            if (hipMalloc((void**)&order, (size_t)std::max<int64_t>(p->n_sub, 1) * p->m * 4) != hipSuccess) { (void)hipGetLastError(); p->order_failed = true; }
            else if (launch_plane1_by_plane0(p->d_rank0, order, p->m, p->n_sub, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                set_err("[E::bgth_reader_scan] building the plane-1-by-plane-0 rank table failed: %s", hipGetErrorString(hipGetLastError()));
                (void)hipFree(order);
                p->order_failed = true;
                return -1;
            }
            else p->d_order = order;
        }
        if (p->d_order) { a.order0 = p->d_order; a.order_blk_stride = (int64_t)p->m; }
    }
    a.raw_counts = (int32_t*)r->raw.p;
    a.h0 = d_h0;
    a.h1 = d_h1;
    a.h_row0 = row0;
    a.blk0 = (int32_t)blk0;
    a.n_blk = (int32_t)(blk1 - blk0 + 1);
    a.row0 = row0;
    a.row1 = row1;
    unsigned long long *d_times = nullptr;
#ifdef BGTH_ABLATE
    { const char *dbg = getenv("BGTH_DEBUG_SKIP"); a.debug_skip = dbg ? atoi(dbg) : 0; }
    if (getenv("BGTH_DEBUG_TIMES")) {                    // per-phase cycle sums over all waves
        HIP_TRY(hipMalloc((void**)&d_times, 64), return -1);
        HIP_TRY(hipMemsetAsync(d_times, 0, 64, s), return -1);
        a.debug_times = d_times;
    }
#endif
    if (timed) HIP_TRY(hipEventRecord(r->ev[0], s), return -1);
    if (!planepath && (G > 1 || r->geom.slices > 1))         // a single-group, single-slice launch stores its counts
        HIP_TRY(hipMemsetAsync(r->raw.p, 0, (size_t)rows * G * 3 * 4, s), return -1);
    if (timed) HIP_TRY(hipEventRecord(r->ev[1], s), return -1);
    if (planepath) {
        a.h0 = p_h0; a.h1 = p_h1;
        HIP_TRY(launch_plane_scan(a, pgeo, s), return -1);
        HIP_TRY(launch_count_planes(p_h0, p_h1, r->sel.d_chunk_desc, (int32_t*)r->raw.p, rows, r->sel.n_chunks, G, s), return -1);
    }
    else if (!dirpath) {
        HIP_TRY(launch_scan(a, geo, s), return -1);
    }
    else {
        // Passes over ranges of sub-blocks whose rows fit the arena: producer, then the walk-only kernel.  An arena that
        // already holds the rows of this scan (the previous scan of this reader covered them in one pass) is walked as is.
        const int nwp = ((p->m + 31) / 32 + 2) & ~1;
        const size_t row_bytes = (size_t)2 * nwp * 8;
        const int64_t sub_rows = (int64_t)1 << p->sub_shift;
        const int64_t first = blk0 << p->sub_shift;                                  // decoding starts at a sub-checkpoint
        const bool reuse = !variant_flag(kVariantDirNoReuse) && r->dir.p && r->dir_stream == s && r->dir_lo <= first && row1 <= r->dir_hi;
        int64_t per_pass = blk1 - blk0 + 1;
        if (!reuse) {
            const size_t cap = dir_arena_cap(p->device) / (size_t)p->arena_share;
            const int64_t blk_rows = std::min<int64_t>(sub_rows, row1 - first);     // (a short image has short sub-blocks)
            const int64_t fit = (int64_t)(cap / (row_bytes * (size_t)blk_rows));
            if (fit < 1) { set_err("[E::bgth_reader_scan] the directory arena (%zu MB) does not hold one sub-block of m=%d", cap >> 20, p->m); return -1; }
            if (p->one_shot) {
                // A walk that will not come back (BGTH_OPEN_HINT=walk) has nothing to gain from an arena of the whole file:
                // the walk-only kernel runs one workgroup per CU, so passes of one ROUND of workgroups each take the same
                // time as one launch of all of them -- with a third of the HBM (one C4 shard: 42 GB instead of 125 GB),
                // which the driver wipes when the process ends, at the expense of whoever allocates next.
                const int64_t rounds = ((blk1 - blk0 + 1) * wgeo.slices + 255) / 256;
                per_pass = std::min(per_pass, (blk1 - blk0 + rounds) / rounds);
            }
            if (per_pass > fit) per_pass = std::max<int64_t>(8, fit / 8 * 8);     // whole XCD rounds of workgroups
            if (per_pass > fit) per_pass = fit;
            const int64_t arena_rows = std::min<int64_t>(per_pass * sub_rows, row1 - first);
            Trace tr;
            struct Lap { Trace &t; ~Lap() { t.lap("directory arena (hipMalloc)"); } } lap_arena{tr};
            if (!r->dir.reserve((size_t)arena_rows * row_bytes) || !r->dir_n0.reserve((size_t)arena_rows * 2 * 4)) {
                r->dir_lo = r->dir_hi = 0;
                set_err("[E::bgth_reader_scan] out of HBM (directory arena of %lld rows)", (long long)arena_rows);
                return -1;
            }
            r->dir_lo = r->dir_hi = 0;
        }
        a.dir = (uint2*)r->dir.p;
        a.dir_n0 = (uint32_t*)r->dir_n0.p;
        a.dir_nwp = nwp;
        a.dir_stage = wgeo.dir_stage | (variant_flag(kVariantDirNoWarm) ? 0 : 2);   // bit 1: warm the L2 with the next row's plane 1
        if (variant_flag(kVariantNoWalkPrio)) a.dir_stage |= 8;                      // bit 3: no progress-based wave priorities in the walk
        for (int64_t b = blk0; b <= blk1; b += per_pass) {
            const int64_t be = std::min(blk1 + 1, b + per_pass);
            const int64_t lo = b << p->sub_shift, hi = std::min(row1, be << p->sub_shift);
            Geometry pg = wgeo;
            pg.workgroups = (int)(((be - b) + 7) / 8 * 8 * wgeo.slices);
            a.blk0 = (int32_t)b;
            a.n_blk = (int32_t)(be - b);
            a.dir_row0 = reuse ? r->dir_lo : lo;
            if (!reuse) {
                if (timed && b == blk0) HIP_TRY(hipEventRecord(r->ev_dir[0], s), return -1);
                HIP_TRY(launch_dirbuild(a, lo, hi, s), return -1);
                if (timed && b == blk0) HIP_TRY(hipEventRecord(r->ev_dir[1], s), return -1);
                ++r->dir_built;
            }
            HIP_TRY(launch_walk(a, pg, s), return -1);
            ++r->dir_passes;
        }
        if (!reuse && r->dir_passes == 1) { r->dir_lo = first; r->dir_hi = row1; r->dir_stream = s; }
    }
    if (timed) HIP_TRY(hipEventRecord(r->ev[2], s), return -1);
    if (d_times) {
        unsigned long long h[8];
        HIP_TRY(hipStreamSynchronize(s), return -1);
        HIP_TRY(hipMemcpy(h, d_times, 64, hipMemcpyDeviceToHost), return -1);
        hipFree(d_times);
        const Geometry &tg = dirpath ? wgeo : planepath ? pgeo : geo;   // (directory path: stage DMA | walk | wait | counts | plane-1 DMA | wait;
                                                                        //  plane-split: walk | toggles | barrier | directory | barrier)
        // a row (batch of K rows) is visited by `slices` workgroups (plane-split kernels: by two, one per plane)
        const double waves = (double)(planepath ? 2 : tg.slices) * ((a.debug_skip & 0x100) ? 1 : tg.threads / 64), nb = (double)rows / tg.K;
        fprintf(stderr, "[bgth debug] memtime ticks per wave and row batch, phases 0..7: %.0f | %.0f | %.0f | %.0f | %.0f | %.0f | %.0f | %.0f   (sum %.0f)\n",
                h[0] / waves / nb, h[1] / waves / nb, h[2] / waves / nb, h[3] / waves / nb, h[4] / waves / nb, h[5] / waves / nb, h[6] / waves / nb, h[7] / waves / nb,
                (double)(h[0] + h[1] + h[2] + h[3] + h[4] + h[5] + h[6] + h[7]) / waves / nb);
    }
    HIP_TRY(launch_finalize((const int32_t*)r->raw.p, d_fin, r->sel.d_group_haps, rows, G, s), return -1);
    if (timed) HIP_TRY(hipEventRecord(r->ev[3], s), return -1);
    return rows;
}

static void collect_timing(bgth_reader_t *r)
{
    float t[3];
    if (hipEventElapsedTime(&t[0], r->ev[1], r->ev[2]) != hipSuccess) return;   // not finished yet
    if (hipEventElapsedTime(&t[1], r->ev[2], r->ev[3]) != hipSuccess) return;
    if (hipEventElapsedTime(&t[2], r->ev[0], r->ev[3]) != hipSuccess) return;
    r->t_ms[0] = t[0]; r->t_ms[1] = t[1]; r->t_ms[2] = t[2];
    r->dir_build_ms = 0.f;
    if (r->dir_built) hipEventElapsedTime(&r->dir_build_ms, r->ev_dir[0], r->ev_dir[1]);   // (first pass)
    r->t_pending = false;
}

// file rows [row0,row1) -> rows of the (possibly partial) image; false if they are not all loaded
static bool to_image_rows(const bgth_pbf_t *p, int64_t &row0, int64_t &row1, const char *who)
{
    if (row0 < 0 || row1 > p->n_total || row0 > row1) {
        set_err("[E::%s] rows [%lld,%lld) outside 0..%lld", who, (long long)row0, (long long)row1, (long long)p->n_total);
        return false;
    }
    if (row0 < p->row_off || row1 > p->row_off + p->n) {
        set_err("[E::%s] rows [%lld,%lld) are not in this partial image, which holds [%lld,%lld)", who, (long long)row0,
                (long long)row1, (long long)p->row_off, (long long)(p->row_off + p->n));
        return false;
    }
    row0 -= p->row_off; row1 -= p->row_off;
    return true;
}


// bgth_reader_scan_device on a sharded image: every shard scans its part of [row0,row1) on its own device and stream, then
// the per-shard counts are GATHERED into d_counts on shard 0's device (the "root"), in shard = row order:
//   shards on another device      ncclSend on their communicator rank / ncclRecv on the root's, one group (RCCL over xGMI)
//   shards on the root's device   a device-to-device copy (virtual shards: several shards of one device)
// The caller's stream (on the root device; NULL = the root shard's stream, synchronised before returning) waits for all of
// it, so a consumer enqueued behind this call -- the device filter -- sees the gathered counts.  Genotype planes stay
// per shard (d_h0 / d_h1 must be NULL).
static int64_t scan_device_sharded(bgth_reader_t *r, int64_t row0, int64_t row1, void *d_counts, void *d_h0, void *d_h1, void *stream)
{
    bgth_pbf_t *p = r->pbf;
    if (d_h0 || d_h1) { set_err("[E::bgth_reader_scan_device] a sharded image gathers counts only: bit planes stay on the shard that decoded them"); return -1; }
    if (row0 < 0 || row1 > p->n_total || row0 > row1) { set_err("[E::bgth_reader_scan_device] rows [%lld,%lld) outside 0..%lld", (long long)row0, (long long)row1, (long long)p->n_total); return -1; }
    if (row0 == row1) return 0;
    const int root_dev = r->subs[0]->pbf->device;
    const bool self = variant_flag(kVariantRcclSelf);
    bool several = self;
    for (bgth_reader_t *sr : r->subs) several = several || sr->pbf->device != root_dev;
    if (several && !ensure_comm(p)) return -1;
    const int gx = gx_of(r->sel.G);
    const size_t cstride = (size_t)(1 + gx) * 3 * 4;                     // bytes per row
    if (!use_device(root_dev)) return -1;
    hipStream_t root_s = stream ? (hipStream_t)stream : r->subs[0]->stream;
    struct Piece { bgth_reader_t *sr; int64_t a0, a1; };
    std::vector<Piece> pieces;
    for (bgth_reader_t *sr : r->subs) {
        const int64_t a0 = std::max(row0, sr->pbf->row_off), a1 = std::min(row1, sr->pbf->row_off + sr->pbf->n);
        if (a0 < a1) pieces.push_back({sr, a0, a1});
    }
    // 1. every shard's scan, enqueued on its own stream (nothing here waits for the device)
    for (Piece &q : pieces) {
        bgth_reader_t *sr = q.sr;
        if (!use_device(sr->pbf->device)) return -1;
        if (!sr->fin.reserve((size_t)(q.a1 - q.a0) * cstride)) { set_err("[E::bgth_reader_scan_device] out of HBM"); return -1; }
        if (enqueue_scan(sr, q.a0 - sr->pbf->row_off, q.a1 - sr->pbf->row_off, (int32_t*)sr->fin.p, nullptr, nullptr, sr->stream, true) < 0) return -1;
        sr->t_pending = true;
    }
    // 2. the gather.  Pieces of the root's device: copies on the shard's stream, the root stream waits for them;
    //    pieces of other devices: send / receive pairs in one RCCL group, the receives on the root stream
    if (several) RCCL_TRY(rccl().GroupStart(), return -1);
    bool failed = false;
    for (Piece &q : pieces) {
        bgth_reader_t *sr = q.sr;
        char *dst = (char*)d_counts + (size_t)(q.a0 - row0) * cstride;
        const size_t bytes = (size_t)(q.a1 - q.a0) * cstride;
        if (sr->pbf->device == root_dev && !self) {
            // the copy runs on the CALLER's stream, behind the shard's scan (event): work the caller enqueued earlier that still
            // reads d_counts -- a filter over the previous scan's counts -- is then ordered before this overwrite
            if (hipSetDevice(root_dev) != hipSuccess) { failed = true; break; }
            if (sr->stream != root_s && (hipEventRecord(sr->ev_gather, sr->stream) != hipSuccess || hipStreamWaitEvent(root_s, sr->ev_gather, 0) != hipSuccess)) { failed = true; break; }
            if (hipMemcpyAsync(dst, sr->fin.p, bytes, hipMemcpyDeviceToDevice, root_s) != hipSuccess) { failed = true; break; }
            // ... and the shard's stream behind the copy: its NEXT scan (a caller that double-buffers d_counts enqueues it without a
            // host synchronisation in between) overwrites `fin`, which a backed-up root stream may not have read yet
            if (sr->stream != root_s && (hipEventRecord(sr->ev_copied, root_s) != hipSuccess || hipStreamWaitEvent(sr->stream, sr->ev_copied, 0) != hipSuccess)) { failed = true; break; }
        } else if (sr->pbf->device == root_dev) {                        // (test knob: the same bytes through RCCL, rank to itself)
            const int root = comm_rank(p, root_dev);
            if (hipSetDevice(root_dev) != hipSuccess || rccl().Send(sr->fin.p, bytes, 0, root, p->comm[root], sr->stream) != 0 ||
                rccl().Recv(dst, bytes, 0, root, p->comm[root], sr->stream) != 0) { failed = true; break; }
        } else {
            const int peer = comm_rank(p, sr->pbf->device), root = comm_rank(p, root_dev);
            if (hipSetDevice(sr->pbf->device) != hipSuccess || rccl().Send(sr->fin.p, bytes, 0 /* ncclInt8 */, root, p->comm[peer], sr->stream) != 0) { failed = true; break; }
            if (hipSetDevice(root_dev) != hipSuccess || rccl().Recv(dst, bytes, 0, peer, p->comm[root], root_s) != 0) { failed = true; break; }
        }
    }
    if (several) {
        const int e = rccl().GroupEnd();
        if (e != 0) { set_err("[E::bgth_reader_scan_device] ncclGroupEnd: %s", rccl().GetErrorString(e)); return -1; }
    }
    if (failed) { set_err("[E::bgth_reader_scan_device] gathering the shards' counts failed"); return -1; }
    for (Piece &q : pieces) {                                            // the root stream waits for what left on the shards' own streams
        bgth_reader_t *sr = q.sr;
        if (sr->pbf->device != root_dev || sr->stream == root_s) continue;
        if (hipSetDevice(root_dev) != hipSuccess || hipEventRecord(sr->ev_gather, sr->stream) != hipSuccess ||
            hipStreamWaitEvent(root_s, sr->ev_gather, 0) != hipSuccess) { set_err("[E::bgth_reader_scan_device] stream hand-over failed"); return -1; }
    }
    if (!stream) {
        if (!use_device(root_dev)) return -1;
        HIP_TRY(hipStreamSynchronize(root_s), return -1);
        for (Piece &q : pieces) { hipSetDevice(q.sr->pbf->device); collect_timing(q.sr); }
    }
    return row1 - row0;
}

extern "C" int64_t bgth_reader_scan_device(bgth_reader_t *r, int64_t row0, int64_t row1, void *d_counts,
                                           void *d_h0, void *d_h1, void *stream)
{
    if (!r || !d_counts) { set_err("[E::bgth_reader_scan_device] NULL argument"); return -1; }
    if (planes_image(r->pbf, "bgth_reader_scan_device")) return -1;
    if (!r->subs.empty()) return guarded("bgth_reader_scan_device", (int64_t)-1, [&] { return scan_device_sharded(r, row0, row1, d_counts, d_h0, d_h1, stream); });
    if (!use_device(r->pbf->device)) return -1;
    hipStream_t s = stream ? (hipStream_t)stream : r->stream;
    if (!to_image_rows(r->pbf, row0, row1, "bgth_reader_scan_device")) return -1;
    const int64_t n = enqueue_scan(r, row0, row1, (int32_t*)d_counts, (uint64_t*)d_h0, (uint64_t*)d_h1, s, true);
    if (n < 0) return n;
    r->t_pending = true;
    if (!stream) { HIP_TRY(hipStreamSynchronize(s), return -1); collect_timing(r); }
    return n;
}

extern "C" int64_t bgth_reader_scan(bgth_reader_t *r, int64_t row0, int64_t row1, int32_t *counts, uint8_t *gt)
{
    if (!r) return -1;
    bgth_pbf_t *p = r->pbf;
    if (planes_image(p, "bgth_reader_scan")) return -1;
    if (!use_device(p->device)) return -1;
    if (!to_image_rows(p, row0, row1, "bgth_reader_scan")) return -1;
    const int G = r->sel.G, gx = gx_of(G);
    const size_t cstride = (size_t)(1 + gx) * 3;
    const int nb = (r->sel.width + 3) / 4;
    if (!r->subs.empty()) return guarded("bgth_reader_scan", (int64_t)-1, [&]() -> int64_t {
        // every shard scans its part of the range on its own device, concurrently (one persistent host thread each), straight
        // into the caller's arrays at the shard's offset: the "gather in shard order" of SURVEY 8e is the address arithmetic
        std::vector<std::string> errs(r->subs.size());
        for (size_t i = 0; i < r->subs.size(); ++i) {
            bgth_reader_t *sr = r->subs[i];
            const int64_t a0 = std::max(row0, sr->pbf->row_off), a1 = std::min(row1, sr->pbf->row_off + sr->pbf->n);
            if (a0 >= a1) continue;
            r->workers[i]->submit([=, &errs] {
                if (bgth_reader_scan(sr, a0, a1, counts ? counts + (size_t)(a0 - row0) * cstride : nullptr,
                                     gt ? gt + (size_t)(a0 - row0) * nb : nullptr) < 0) errs[i] = g_err;
            });
        }
        for (ShardWorker *w : r->workers) w->wait();
        for (const std::string &e : errs) if (!e.empty()) { set_err("%s", e.c_str()); return -1; }
        return row1 - row0;
    });
    // with genotypes the planes are large: walk the range in pieces of whole blocks
    int64_t piece = row1 - row0;
    if (gt) {
        const int64_t per_row = (int64_t)r->sel.n_chunks * 16 + nb;
        int64_t max_rows = ((int64_t)1 << 30) / std::max<int64_t>(per_row, 1);
        const int64_t blk_rows = (int64_t)1 << p->sub_shift;
        max_rows = std::max<int64_t>(blk_rows, max_rows / blk_rows * blk_rows);
        piece = std::min(piece, max_rows);
    }
    for (int64_t a0 = row0; a0 < row1;) {
        // pieces end on block boundaries so that no block is decoded twice
        int64_t a1 = std::min(row1, a0 + piece);
        if (a1 < row1) a1 = std::max<int64_t>(a0 + 1, (a1 >> p->sub_shift) << p->sub_shift);
        const int64_t rows = a1 - a0;
        if (!r->fin.reserve((size_t)rows * cstride * 4)) { set_err("[E::bgth_reader_scan] out of HBM"); return -1; }
        uint64_t *d_h0 = nullptr, *d_h1 = nullptr;
        if (gt) {
            const size_t pl = (size_t)rows * r->sel.n_chunks * 8;
            if (!r->h0.reserve(pl) || !r->h1.reserve(pl) || !r->gt.reserve((size_t)rows * nb)) { set_err("[E::bgth_reader_scan] out of HBM"); return -1; }
            d_h0 = (uint64_t*)r->h0.p; d_h1 = (uint64_t*)r->h1.p;
        }
        if (enqueue_scan(r, a0, a1, (int32_t*)r->fin.p, d_h0, d_h1, r->stream, true) < 0) return -1;
        if (gt) HIP_TRY(launch_pack2(d_h0, d_h1, r->sel.d_slot_of_out, (uint8_t*)r->gt.p, rows, r->sel.n_chunks, r->sel.width, r->stream), return -1);
        if (counts) HIP_TRY(hipMemcpyAsync(counts + (size_t)(a0 - row0) * cstride, r->fin.p, (size_t)rows * cstride * 4, hipMemcpyDeviceToHost, r->stream), return -1);
        if (gt) HIP_TRY(hipMemcpyAsync(gt + (size_t)(a0 - row0) * nb, r->gt.p, (size_t)rows * nb, hipMemcpyDeviceToHost, r->stream), return -1);
        HIP_TRY(hipStreamSynchronize(r->stream), return -1);
        collect_timing(r);
        a0 = a1;
    }
    return row1 - row0;
}

extern "C" int bgth_reader_last_timing(const bgth_reader_t *rc, float out[3])
{
    bgth_reader_t *r = const_cast<bgth_reader_t*>(rc);
    if (!r->pair_readers.empty()) return bgth_reader_last_timing(r->pair_readers[0], out);
    if (!r->subs.empty()) {                                      // shards run concurrently: the slowest one
        out[0] = out[1] = out[2] = 0;
        for (bgth_reader_t *sr : r->subs) { float t[3]; bgth_reader_last_timing(sr, t); for (int k = 0; k < 3; ++k) out[k] = std::max(out[k], t[k]); }
        return 0;
    }
    if (r->t_pending) { hipSetDevice(r->pbf->device); collect_timing(r); }
    out[0] = r->t_ms[0]; out[1] = r->t_ms[1]; out[2] = r->t_ms[2];
    return 0;
}

// the same for ONE shard of a sharded reader (what bench.py's product_sharded record reports per device); -1 = no such shard
extern "C" int bgth_reader_shard_timing(const bgth_reader_t *rc, int shard, float out[3])
{
    if (!rc || shard < 0 || (size_t)shard >= rc->subs.size()) return -1;
    return bgth_reader_last_timing(rc->subs[(size_t)shard], out);
}

extern "C" int bgth_reader_last_path(const bgth_reader_t *rc, float out[4])
{
    bgth_reader_t *r = const_cast<bgth_reader_t*>(rc);
    if (!r->pair_readers.empty()) return bgth_reader_last_path(r->pair_readers[0], out);
    if (!r->subs.empty()) return bgth_reader_last_path(r->subs[0], out);
    if (r->t_pending) { hipSetDevice(r->pbf->device); collect_timing(r); }
    out[0] = (r->plane_path ? 2.f : r->geom.dir_stage >= 0 ? 1.f : 0.f); out[1] = (float)r->dir_passes; out[2] = (float)r->dir_built; out[3] = r->dir_build_ms;
    return 0;
}

extern "C" int bgth_reader_last_geometry(const bgth_reader_t *r, int out[6])
{
    if (!r->pair_readers.empty()) return bgth_reader_last_geometry(r->pair_readers[0], out);
    if (!r->subs.empty()) return bgth_reader_last_geometry(r->subs[0], out);
    out[0] = r->geom.threads; out[1] = r->geom.cpt; out[2] = r->geom.slices;
    out[3] = r->geom.K; out[4] = r->geom.lds_bytes; out[5] = r->geom.workgroups;
    return 0;
}

// ----------------------------------------------------------------------------------------------------
// pull interface: pbf_seek / pbf_read semantics served from batches decoded on the device
// ----------------------------------------------------------------------------------------------------
extern "C" int bgth_reader_seek(bgth_reader_t *r, int64_t row)
{
    if (!r) return -1;
    if (!r->pair_readers.empty()) {
        for (bgth_reader_t *pr : r->pair_readers) if (bgth_reader_seek(pr, row) < 0) return -1;
        return 0;
    }
    int64_t row1 = row + 1;
    if (!to_image_rows(r->pbf, row, row1, "bgth_reader_seek")) return -1;                 // ref pbwt.c:359
    r->next = row;
    return 0;
}

// Host destinations of one piece of a refill window (already offset to the piece's first row)
struct PieceDst { int32_t *counts; uint8_t *a0, *a1, *gt8, *gttext; };

// Enqueue on the reader's stream: decode image rows [row0,row1) of a single-device reader into the device buffers of `w`
// and copy what `want` asks for to the host destinations.  Nothing is waited for.
static bool enqueue_piece(bgth_reader_t *r, PullWindow &w, int want, int64_t row0, int64_t row1, const PieceDst &d)
{
    const int width = r->sel.width;
    const size_t cstride = (size_t)(1 + gx_of(r->sel.G)) * 3;
    const bool need_bits = want != 0;                            // any genotype output needs the bit planes H0/H1
    const int64_t rows = row1 - row0;
    const size_t pl = (size_t)rows * r->sel.n_chunks * 8, by = (size_t)rows * width;
    if (!w.fin.reserve((size_t)rows * cstride * 4) || (need_bits && (!w.h0.reserve(pl) || !w.h1.reserve(pl))) ||
        ((want & BGTH_WANT_PLANES) && !w.planes.reserve(2 * by)) || ((want & BGTH_WANT_GT8) && !w.gt8.reserve(by)) ||
        ((want & BGTH_WANT_GTTEXT) && !w.gttext.reserve(2 * by))) {
        set_err("[E::bgth_reader_read] out of HBM for a %lld-row batch", (long long)rows);
        return false;
    }
    uint64_t *d_h0 = need_bits ? (uint64_t*)w.h0.p : nullptr, *d_h1 = need_bits ? (uint64_t*)w.h1.p : nullptr;
    if (enqueue_scan(r, row0, row1, (int32_t*)w.fin.p, d_h0, d_h1, r->stream, true) < 0) return false;
    if (want & BGTH_WANT_PLANES) {
        uint8_t *d_a0 = (uint8_t*)w.planes.p, *d_a1 = d_a0 + by;
        HIP_TRY(launch_unpack_bytes(d_h0, d_h1, r->sel.d_slot_of_out, d_a0, d_a1, rows, r->sel.n_chunks, width, r->stream), return false);
        HIP_TRY(hipMemcpyAsync(d.a0, d_a0, by, hipMemcpyDeviceToHost, r->stream), return false);
        HIP_TRY(hipMemcpyAsync(d.a1, d_a1, by, hipMemcpyDeviceToHost, r->stream), return false);
    }
    if (want & (BGTH_WANT_GT8 | BGTH_WANT_GTTEXT)) {
        uint8_t *d_gt8 = (want & BGTH_WANT_GT8) ? (uint8_t*)w.gt8.p : nullptr;
        uint32_t *d_txt = (want & BGTH_WANT_GTTEXT) ? (uint32_t*)w.gttext.p : nullptr;
        HIP_TRY(launch_emit_gt(d_h0, d_h1, r->sel.d_slot_of_out, d_gt8, d_txt, rows, r->sel.n_chunks, width, r->stream), return false);
        if (d_gt8) HIP_TRY(hipMemcpyAsync(d.gt8, d_gt8, by, hipMemcpyDeviceToHost, r->stream), return false);
        if (d_txt) HIP_TRY(hipMemcpyAsync(d.gttext, d_txt, 2 * by, hipMemcpyDeviceToHost, r->stream), return false);
    }
    HIP_TRY(hipMemcpyAsync(d.counts, w.fin.p, (size_t)rows * cstride * 4, hipMemcpyDeviceToHost, r->stream), return false);
    w.row0 = row0; w.row1 = row1; w.has = want; w.valid = true;   // (rows of THIS reader's image: a shard's are local)
    return true;
}

// the same, and wait until the copies have landed (a shard's piece of the parent's window)
static bool decode_piece(bgth_reader_t *r, int want, int64_t row0, int64_t row1, const PieceDst &d)
{
    if (!enqueue_piece(r, r->win[0], want, row0, row1, d)) return false;
    HIP_TRY(hipStreamSynchronize(r->stream), return false);
    collect_timing(r);
    return true;
}

// host buffers of a window for `rows` rows
static bool window_host(bgth_reader_t *r, PullWindow &w, int want, int64_t rows, PieceDst &base)
{
    const int width = r->sel.width;
    const size_t cstride = (size_t)(1 + gx_of(r->sel.G)) * 3, by = (size_t)rows * width;
    if (!w.h_counts.reserve((size_t)rows * cstride * 4) || ((want & BGTH_WANT_PLANES) && !w.h_planes.reserve(2 * by)) ||
        ((want & BGTH_WANT_GT8) && !w.h_gt8.reserve(by)) || ((want & BGTH_WANT_GTTEXT) && !w.h_gttext.reserve(2 * by))) {
        set_err("[E::bgth_reader_read] out of memory for a %lld-row batch", (long long)rows);
        return false;
    }
    base = PieceDst{(int32_t*)w.h_counts.p, (uint8_t*)w.h_planes.p, (uint8_t*)w.h_planes.p + by, (uint8_t*)w.h_gt8.p, (uint8_t*)w.h_gttext.p};
    return true;
}

// rows a window may hold: with byte planes while they stay under 256 MiB, with counts only a wide window (the whole point
// of the device: one launch, many sites), always whole sub-blocks
static int64_t window_max_rows(const bgth_reader_t *r, int want)
{
    const int64_t blk_rows = (int64_t)1 << r->pbf->sub_shift;
    const int per_hap = (want & BGTH_WANT_PLANES ? 2 : 0) + (want & BGTH_WANT_GT8 ? 1 : 0) + (want & BGTH_WANT_GTTEXT ? 2 : 0);
    int64_t max_rows = want != 0 ? ((int64_t)256 << 20) / std::max(1, per_hap * r->sel.width) : (int64_t)1 << 22;
    if (want & BGTH_WANT_BITS) max_rows = std::min(max_rows, ((int64_t)512 << 20) / ((int64_t)r->sel.n_chunks * 16));   // H0 + H1 in HBM
    if (r->max_ahead > 0) max_rows = std::min(max_rows, r->max_ahead);
    return std::max<int64_t>(blk_rows, max_rows / blk_rows * blk_rows);
}

// A sequential walk: start the window behind the current one now, four times its size (up to the maximum); its scan and
// copies run while the caller consumes the current window.  Failures only mean there is no prefetched window.
static void prefetch_next(bgth_reader_t *r)
{
    bgth_pbf_t *p = r->pbf;
    PullWindow &nx = r->win[r->cur ^ 1];
    nx.valid = nx.pending = false;
    if (!r->subs.empty() || r->ring1 >= p->n || variant_flag(kVariantNoPrefetch)) return;
    const int want = r->ring_has;
    const int64_t blk_rows = (int64_t)1 << p->sub_shift, max_rows = window_max_rows(r, want);
    if (r->max_ahead <= 0) r->ahead = std::min(max_rows, std::max<int64_t>(blk_rows, r->ahead * 4));
    const int64_t row0 = r->ring1;
    const int64_t row1 = std::min<int64_t>(p->n, ((row0 >> p->sub_shift) << p->sub_shift) + (r->max_ahead > 0 ? max_rows : r->ahead));
    PieceDst base;
    if (!window_host(r, nx, want, row1 - row0, base) || !enqueue_piece(r, nx, want, row0, row1, base)) { nx.valid = false; return; }
    nx.pending = true;
}

static bool refill(bgth_reader_t *r)
{
    bgth_pbf_t *p = r->pbf;
    const int64_t blk_rows = (int64_t)1 << p->sub_shift;
    const int width = r->sel.width;
    const size_t cstride = (size_t)(1 + gx_of(r->sel.G)) * 3;
    Trace tr;
    const int want = r->want;
    const int64_t row0 = r->next;
    if ((want & (BGTH_WANT_GT8 | BGTH_WANT_GTTEXT)) && (width & 1)) {
        set_err("[E::bgth_reader_read] genotype vectors need whole samples (an even number of columns), have %d", width);
        return false;
    }
    // the prefetched window, if it is the one wanted
    PullWindow &nx = r->win[r->cur ^ 1];
    if (nx.pending) {
        HIP_TRY(hipStreamSynchronize(r->stream), { nx.pending = nx.valid = false; return false; });
        nx.pending = false;
        collect_timing(r);
    }
    if (nx.valid && row0 >= nx.row0 && row0 < nx.row1 && !(want & ~nx.has)) {
        r->cur ^= 1;
        r->ring0 = nx.row0; r->ring1 = nx.row1; r->ring_has = nx.has;
        tr.lap("refill: prefetched window");
        prefetch_next(r);
        return true;
    }
    nx.valid = false;
    int64_t max_rows = window_max_rows(r, want);
    // Adaptive look-ahead: a refill that continues where the last window ended is a sequential walk and gets four times
    // the previous window (up to max_rows); one that lands elsewhere (sparse access: BED / allele sets, merges with gaps)
    // starts again at one sub-block, so a visited site costs a pre-roll of at most one sub-block, not a 256 MiB window.
    const bool sequential = r->ring1 > r->ring0 && row0 == r->ring1 && r->ring_has == want;
    if (r->max_ahead <= 0) {
        r->ahead = sequential ? std::min(max_rows, std::max<int64_t>(blk_rows, r->ahead * 4)) : blk_rows;
        max_rows = std::min(max_rows, r->ahead);
    }
    const int64_t row1 = std::min<int64_t>(p->n, ((row0 >> p->sub_shift) << p->sub_shift) + max_rows);
    PullWindow &w = r->win[r->cur];
    w.valid = false;
    PieceDst base;
    if (!window_host(r, w, want, row1 - row0, base)) return false;
    tr.lap("refill: buffers");
    if (r->subs.empty()) {
        if (!enqueue_piece(r, w, want, row0, row1, base)) return false;
        HIP_TRY(hipStreamSynchronize(r->stream), { w.valid = false; return false; });
        collect_timing(r);
    } else {
        // the window is cut at the shard boundaries; every shard decodes its piece on its own device, concurrently, and
        // copies it to its place in the (portable, pinned) host ring
        std::vector<std::string> errs(r->subs.size());
        for (size_t i = 0; i < r->subs.size(); ++i) {
            bgth_reader_t *sr = r->subs[i];
            const int64_t a0 = std::max(row0, sr->pbf->row_off), a1 = std::min(row1, sr->pbf->row_off + sr->pbf->n);
            if (a0 >= a1) continue;
            const size_t k = (size_t)(a0 - row0);
            PieceDst d = {base.counts + k * cstride, base.a0 ? base.a0 + k * width : nullptr, base.a1 ? base.a1 + k * width : nullptr,
                          base.gt8 ? base.gt8 + k * width : nullptr, base.gttext ? base.gttext + 2 * k * width : nullptr};
            r->workers[i]->submit([=, &errs] {                             // (the shard's persistent thread: its device is bound)
                g_err[0] = 0;
                if (hipSetDevice(sr->pbf->device) != hipSuccess || !decode_piece(sr, want, a0 - sr->pbf->row_off, a1 - sr->pbf->row_off, d))
                    errs[i] = g_err[0] ? g_err : "shard failed";
            });
        }
        for (ShardWorker *w : r->workers) w->wait();
        for (const std::string &e : errs) if (!e.empty()) { set_err("%s", e.c_str()); return false; }
        w.row0 = row0; w.row1 = row1; w.has = want; w.valid = true;
    }
    tr.lap("refill: scan + copies");
    r->ring0 = row0; r->ring1 = row1; r->ring_has = want;
    if (sequential) prefetch_next(r);
    return true;
}

extern "C" const uint8_t **bgth_reader_read(bgth_reader_t *r)
{
    if (!r) return nullptr;
    if (!r->pair_readers.empty()) {                              // g planes: every pair's row, the plane pointers side by side (pbwt.c:325-334)
        size_t k = 0;
        for (bgth_reader_t *pr : r->pair_readers) {
            const uint8_t **a = bgth_reader_read(pr);
            if (!a) return nullptr;
            r->multi_ret[k++] = a[0];
            if (k < r->multi_ret.size()) r->multi_ret[k++] = a[1];
        }
        return r->multi_ret.data();
    }
    bgth_pbf_t *p = r->pbf;
    if (r->next >= p->n) {                                       // ref pbwt.c:336: no more 'B' records
        if (p->row_off + p->n < p->n_total) set_err("[E::bgth_reader_read] row %lld is beyond this partial image", (long long)(p->row_off + r->next));
        return nullptr;
    }
    if (!use_device(p->device)) return nullptr;
    if (r->next < r->ring0 || r->next >= r->ring1 || (r->want & ~r->ring_has))
        if (!guarded("bgth_reader_read", false, [&] { return refill(r); })) return nullptr;
    const PullWindow &w = r->win[r->cur];
    const int width = r->sel.width;
    const size_t by = (size_t)(r->ring1 - r->ring0) * width;
    const size_t k = (size_t)(r->next - r->ring0);
    if (r->want & BGTH_WANT_PLANES) {
        r->ret[0] = (const uint8_t*)w.h_planes.p + k * width;
        r->ret[1] = (const uint8_t*)w.h_planes.p + by + k * width;
    } else r->ret[0] = r->ret[1] = nullptr;
    r->last_gt8 = (r->want & BGTH_WANT_GT8) ? (const int8_t*)w.h_gt8.p + k * width : nullptr;
    r->last_gttext = (r->want & BGTH_WANT_GTTEXT) ? (const char*)w.h_gttext.p + 2 * k * width : nullptr;
    r->last_counts = (const int32_t*)w.h_counts.p + k * (size_t)(1 + gx_of(r->sel.G)) * 3;
    ++r->next;
    return r->ret;
}

extern "C" int bgth_reader_config(bgth_reader_t *r, int want_planes, int64_t max_rows_ahead)
{
    if (!r) return -1;
    if (want_planes & ~(BGTH_WANT_PLANES | BGTH_WANT_GT8 | BGTH_WANT_GTTEXT | BGTH_WANT_BITS)) { set_err("[E::bgth_reader_config] unknown output bits 0x%x", want_planes); return -1; }
    if (!r->pair_readers.empty()) {
        if (want_planes != BGTH_WANT_PLANES) { set_err("[E::bgth_reader_config] an image of %d bit planes hands out byte planes only (genotype codes are BGT's two planes)", r->pbf->g); return -1; }
        for (bgth_reader_t *pr : r->pair_readers) if (bgth_reader_config(pr, want_planes, max_rows_ahead) < 0) return -1;
        return 0;
    }
    r->want = want_planes;
    r->max_ahead = max_rows_ahead;
    return 0;
}

extern "C" const int32_t *bgth_reader_last_counts(const bgth_reader_t *r) { return r->last_counts; }
extern "C" const int8_t *bgth_reader_last_gt8(const bgth_reader_t *r) { return r->last_gt8; }
extern "C" const char *bgth_reader_last_gt_text(const bgth_reader_t *r) { return r->last_gttext; }

// ----------------------------------------------------------------------------------------------------
// allele-set reductions on the device (reference bgt.c:859-876): the row of the last bgth_reader_read, still in HBM as
// bit planes, is folded into per-reader accumulators; nothing of the row crosses PCIe
// ----------------------------------------------------------------------------------------------------
static bool folds_prepare(bgth_reader_t *r)
{
    if (r->folds_live) return true;
    const size_t nc = (size_t)(r->sel.width / 2) * 4, nh = (size_t)r->sel.width * 8;
    if (!r->carriers.reserve(nc ? nc : 4) || !r->hapsig.reserve(nh ? nh : 16)) { set_err("[E::bgth_reader_fold_last] out of HBM"); return false; }
    HIP_TRY(hipMemsetAsync(r->carriers.p, 0, nc, r->stream), return false);
    HIP_TRY(hipMemsetAsync(r->hapsig.p, 0, nh, r->stream), return false);
    r->folds_live = true;
    return true;
}

extern "C" int bgth_reader_fold_last(bgth_reader_t *r, int code, int bit)
{
    if (!r) return -1;
    if (code > 3 || bit > 63) { set_err("[E::bgth_reader_fold_last] code %d / bit %d out of range", code, bit); return -1; }
    if (planes_image(r->pbf, "bgth_reader_fold_last")) return -1;
    if (r->sel.width & 1) { set_err("[E::bgth_reader_fold_last] needs whole samples (an even number of columns), have %d", r->sel.width); return -1; }
    const int64_t row = r->next - 1;
    if (row < r->ring0 || row >= r->ring1 || !(r->ring_has & BGTH_WANT_BITS)) {
        set_err("[E::bgth_reader_fold_last] no row read, or the reader was not configured with BGTH_WANT_BITS");
        return -1;
    }
    bgth_reader_t *sr = r;
    int64_t lrow = row;
    for (bgth_reader_t *sub : r->subs)
        if (row >= sub->pbf->row_off && row < sub->pbf->row_off + sub->pbf->n) { sr = sub; lrow = row - sub->pbf->row_off; }
    const PullWindow &w = sr->win[sr->cur];                       // (a shard decodes into its first window, in its own rows)
    if (!w.valid || !(w.has & BGTH_WANT_BITS) || lrow < w.row0 || lrow >= w.row1) { set_err("[E::bgth_reader_fold_last] row %lld is not resident", (long long)row); return -1; }
    if (!use_device(sr->pbf->device) || !folds_prepare(sr)) return -1;
    const size_t off = (size_t)(lrow - w.row0) * sr->sel.n_chunks;
    HIP_TRY(launch_fold_alleles((const uint64_t*)w.h0.p + off, (const uint64_t*)w.h1.p + off, sr->sel.d_slot_of_out,
                                code >= 0 ? (int32_t*)sr->carriers.p : nullptr, bit >= 0 ? (uint64_t*)sr->hapsig.p : nullptr,
                                sr->sel.width, code, bit, sr->stream), return -1);
    return 0;
}

extern "C" int bgth_reader_take_folds(bgth_reader_t *r, int32_t *carriers, uint64_t *hap)
{
    if (!r) return -1;
    if (planes_image(r->pbf, "bgth_reader_take_folds")) return -1;
    return guarded("bgth_reader_take_folds", -1, [&]() -> int {
        const int width = r->sel.width, ns = width / 2;
        if (carriers) memset(carriers, 0, (size_t)ns * 4);
        if (hap) memset(hap, 0, (size_t)width * 8);
        std::vector<bgth_reader_t*> all(r->subs.begin(), r->subs.end());
        if (all.empty()) all.push_back(r);
        std::vector<int32_t> c((size_t)ns);
        std::vector<uint64_t> h((size_t)width);
        for (bgth_reader_t *sr : all) {                      // shards saw disjoint rows: their folds add / or together
            if (!sr->folds_live) continue;
            if (!use_device(sr->pbf->device)) return -1;
            HIP_TRY(hipMemcpyAsync(c.data(), sr->carriers.p, (size_t)ns * 4, hipMemcpyDeviceToHost, sr->stream), return -1);
            HIP_TRY(hipMemcpyAsync(h.data(), sr->hapsig.p, (size_t)width * 8, hipMemcpyDeviceToHost, sr->stream), return -1);
            HIP_TRY(hipStreamSynchronize(sr->stream), return -1);
            if (carriers) for (int i = 0; i < ns; ++i) carriers[i] += c[(size_t)i];
            if (hap) for (int i = 0; i < width; ++i) hap[i] |= h[(size_t)i];
            sr->folds_live = false;                          // the next fold starts from zero
        }
        return 0;
    });
}

// ----------------------------------------------------------------------------------------------------
// site filter on the device
// ----------------------------------------------------------------------------------------------------
struct bgth_filter_s { int device; FilterProgram prog; };

extern "C" bgth_filter_t *bgth_filter_create(int device, int n_items, const int32_t *op, const int64_t *ival,
                                             const double *rval, const int32_t *slot)
{
    if (n_items <= 0 || n_items > kFilterMaxItems) { set_err("[E::bgth_filter_create] %d program items (1..%d supported)", n_items, kFilterMaxItems); return nullptr; }
    bgth_filter_t *f = new bgth_filter_s();
    f->device = device; f->prog.n = n_items;
    for (int i = 0; i < n_items; ++i) {
        if (!(op[i] == 0 || op[i] == 1 || op[i] == 2 || (op[i] >= 17 && op[i] <= 40))) { set_err("[E::bgth_filter_create] item %d: unsupported op %d", i, op[i]); delete f; return nullptr; }
        f->prog.op[i] = op[i]; f->prog.slot[i] = slot ? slot[i] : -1; f->prog.ival[i] = ival[i]; f->prog.rval[i] = rval[i];
    }
    return f;
}

extern "C" void bgth_filter_destroy(bgth_filter_t *f) { delete f; }

extern "C" int bgth_filter_apply_device(const bgth_filter_t *f, const void *d_counts, int64_t n_rows, int ints_per_row,
                                        void *d_flags, void *d_n_pass, void *stream)
{
    if (!f || !d_counts || !d_flags || !d_n_pass) { set_err("[E::bgth_filter_apply_device] NULL argument"); return -1; }
    if (!use_device(f->device)) return -1;
    HIP_TRY(launch_filter(f->prog, (const int32_t*)d_counts, n_rows, ints_per_row, (uint8_t*)d_flags,
                          (unsigned long long*)d_n_pass, (hipStream_t)stream), return -1);
    return 0;
}

// ----------------------------------------------------------------------------------------------------
// diagnostics
// ----------------------------------------------------------------------------------------------------
extern "C" int bgth_debug_stream_read(int device, size_t bytes, int width, int repeats)
{
    if (!use_device(device)) return -1;
    void *buf = nullptr; uint32_t *sink = nullptr;
    HIP_TRY(hipMalloc(&buf, bytes), return -1);
    HIP_TRY(hipMalloc((void**)&sink, 4), { hipFree(buf); return -1; });
    HIP_TRY(hipMemset(buf, 1, bytes), { hipFree(buf); hipFree(sink); return -1; });
    for (int i = 0; i < repeats; ++i) HIP_TRY(launch_stream_read(buf, bytes, width, sink, nullptr), break);
    hipDeviceSynchronize();
    hipFree(buf); hipFree(sink);
    return 0;
}
