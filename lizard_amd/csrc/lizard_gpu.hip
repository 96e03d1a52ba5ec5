// lizard_gpu.hip — gfx950 kernels + the thin extern "C" shim the host C layer (lizard_host.c) calls.
//
// The kernels live in lz_kernels.h (one wavefront per Lizard API block, a persistent grid of one workgroup per CU).
// This file is the HIP side of the host layer: one context PER DEVICE (arenas, tables, streams), a per-thread device selection
// and error text, the launcher, the device-buffer entry points, the shim the host C files call (lizard_gpu_ctx.h:
// lizard_pipeline_host.c holds the pipelined host-buffer path, in C), and the single-process multi-device entry with an RCCL
// all-gather of the per-block sizes (lizard_shard.h).
#include <hip/hip_runtime.h>
#include <atomic>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/lizard_amd.h"
#include "lizard_gpu_shim.h"
#include "lizard_gpu_ctx.h"
#include "lz_kernels.h"   // LzBatch / LzUnBatch, residency knobs, the kernels

namespace {

typedef LzStage Stage;                                 // lizard_gpu_ctx.h (plain C: shared with lizard_pipeline_host.c)
typedef LzCtx Ctx;
Ctx g_ctx[LZ_MAX_DEVICES];
struct CtxDefaults {                                   // (the contexts are zero-initialised statics; these fields start elsewhere)
    CtxDefaults() { for (Ctx& c : g_ctx) { pthread_mutex_init(&c.mu, nullptr); pthread_mutex_init(&c.comb.mu, nullptr); pthread_cond_init(&c.comb.cv, nullptr);
                                           c.device = -1; c.laneOrderOk = 1; c.hostKernelMs = -1.0f; } }
} g_ctx_defaults;
int g_default_device = 0;                              // process default (the last LizardGPU_setDevice of any thread)
pthread_mutex_t g_sel_mu = PTHREAD_MUTEX_INITIALIZER;
thread_local int  t_device = -1;                       // calling thread's selection, -1 = process default
thread_local char t_err[LZK_ERR_BYTES] = {0};

void set_err(const char* fmt, const char* a, const char* b)
{
    snprintf(t_err, sizeof t_err, fmt, a, b);
}

#define LZ_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            set_err("%s failed: %s", #call, hipGetErrorString(e_));                                    \
            return e_ == hipErrorOutOfMemory ? -LIZARDGPU_ERR_NOMEM : -LIZARDGPU_ERR_HIP;              \
        }                                                                                              \
    } while (0)

// Memory budget (LizardGPU_setMemoryBudget): bytes of device memory the large buffers of ONE device's context may take (scratch
// arenas, per-wave tables, hashChain work areas, the staging of the host-buffer entries); 0 = no cap.
std::atomic<size_t> g_budget{0};                              // read by every device's launches under their own locks, and without a lock by the getters
size_t budget_room(const Ctx& c) { const size_t b = g_budget.load(std::memory_order_relaxed); return !b ? (size_t)-1 : (b > c.devBytes ? b - c.devBytes : 0); }
int dev_alloc(Ctx& c, void** p, size_t bytes, const char* what)
{
    if (bytes > budget_room(c)) {
        snprintf(t_err, sizeof t_err, "%s: %zu bytes do not fit the memory budget of %zu bytes (%zu in use; LizardGPU_setMemoryBudget)", what, bytes, g_budget.load(), c.devBytes);
        return -LIZARDGPU_ERR_NOMEM;
    }
    LZ_HIP(hipMalloc(p, bytes));
    c.devBytes += bytes;
    return 0;
}
void dev_free(Ctx& c, void* p, size_t bytes)
{
    if (!p) return;
    (void)hipFree(p);
    c.devBytes -= bytes < c.devBytes ? bytes : c.devBytes;
}
const size_t kScratchBytes = (size_t)LZ_MAX_WAVES * LZ_SCRATCH_BYTES;     // per CU

int selected_device()
{
    if (t_device >= 0) return t_device;
    pthread_mutex_lock(&g_sel_mu);
    const int d = g_default_device;
    pthread_mutex_unlock(&g_sel_mu);
    return d;
}

// Locks the selected device's context and makes that device current for the calling thread (HIP's current device
// is per thread); restores the caller's device on exit.  (lzk_guard_acquire / _release: the same for the C files.)
void guard_acquire(LzGuard& g)
{
    g.c = nullptr; g.saved = -1; g.rc = 0;
    t_err[0] = 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        snprintf(t_err, sizeof t_err, "no HIP device visible");
        g.rc = -LIZARDGPU_ERR_NO_DEVICE; return;
    }
    const int dev = selected_device();
    if (dev < 0 || dev >= count || dev >= LZ_MAX_DEVICES) {
        snprintf(t_err, sizeof t_err, "device %d out of range (%d visible)", dev, count);
        g.rc = -LIZARDGPU_ERR_ARG; return;
    }
    if (hipGetDevice(&g.saved) != hipSuccess) g.saved = -1;
    g.c = &g_ctx[dev];
    pthread_mutex_lock(&g.c->mu);
    if (hipSetDevice(dev) != hipSuccess) {
        snprintf(t_err, sizeof t_err, "hipSetDevice(%d) failed", dev); g.rc = -LIZARDGPU_ERR_HIP;
        pthread_mutex_unlock(&g.c->mu); g.c = nullptr;
        if (g.saved >= 0) (void)hipSetDevice(g.saved);
        g.saved = -1;
        return;
    }
    g.c->device = dev;
}
void guard_release(LzGuard& g)
{
    if (g.c) pthread_mutex_unlock(&g.c->mu);
    if (g.saved >= 0) (void)hipSetDevice(g.saved);
    g.c = nullptr; g.saved = -1;
}
struct Guard : LzGuard {
    Guard() { guard_acquire(*this); }
    ~Guard() { guard_release(*this); }
};

static int ctx_init_body(Ctx& c);
void ctx_release(Ctx& c);
int ctx_init(Ctx& c)
{
    if (c.ready) return 0;
    const int rc = ctx_init_body(c);
    if (rc) {                                                   // a failure half way: what was taken goes back (and is no longer counted against the budget)
        c.ready = 1; ctx_release(c);
        (void)hipGetLastError();
    }
    return rc;
}
static int ctx_init_body(Ctx& c)
{
    hipDeviceProp_t prop;
    LZ_HIP(hipGetDeviceProperties(&prop, c.device));
    c.cus = prop.multiProcessorCount;
    { const int rc_ = dev_alloc(c, (void**)&c.scratch, (size_t)c.cus * kScratchBytes, "scratch arena"); if (rc_) return rc_; }   // one slot per resident wave (one workgroup per CU)
#ifdef LZ_PROFILE
    LZ_HIP(hipMemset(c.scratch, 0, (size_t)c.cus * LZ_MAX_WAVES * LZ_SCRATCH_BYTES));
#endif
    LZ_HIP(hipMalloc((void**)&c.counter, 64));
    LZ_HIP(hipEventCreate(&c.ev0));
    LZ_HIP(hipEventCreate(&c.ev1));
    c.nExtra = 0; c.nextExtra = 0; c.maxArenas = LZ_ARENAS_MAX; c.lastStream = nullptr; c.lastEv0 = c.ev0; c.lastEv1 = c.ev1;
    if (const char* e = getenv("LIZARDGPU_ARENAS")) { const int n = atoi(e); if (n >= 1 && n <= LZ_ARENAS_MAX) c.maxArenas = n; }
    for (Stage& s : c.stage) {
        LZ_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        LZ_HIP(hipEventCreate(&s.k0)); LZ_HIP(hipEventCreate(&s.k1));
        LZ_HIP(hipEventCreateWithFlags(&s.meta, hipEventDisableTiming));
        LZ_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        LZ_HIP(hipEventCreateWithFlags(&s.up, hipEventDisableTiming));
    }
    // Self-check of the hardware property the exchange round (levels 10/30) and the hashChain build rely on.  A device that
    // fails it keeps working for the other levels; the dependent ones are refused, loudly (launch()).
    // LIZARDGPU_FORCE_LANE_ORDER_FAILURE=1 takes the failure branch on a healthy device (tests).
    {
        LZ_HIP(hipMemset(c.counter, 0, 4));
        hipLaunchKernelGGL(lz_selfcheck_lane_order_kernel, dim3(16), dim3(64), 0, 0, c.counter, 256u);
        LZ_HIP(hipGetLastError());
        u32 bad = 0;
        LZ_HIP(hipMemcpy(&bad, c.counter, 4, hipMemcpyDeviceToHost));
        const char* force = getenv("LIZARDGPU_FORCE_LANE_ORDER_FAILURE");
        c.laneOrderOk = bad == 0 && !(force && force[0] == '1');
        if (!c.laneOrderOk)
            fprintf(stderr, "liblizard_amd: device %d: lanes of one LDS atomic are NOT served in lane order (%u violations%s); "
                            "levels 10, 30, 12-17 and 32-38 are refused on this device\n", c.device, bad, bad ? "" : ", forced by LIZARDGPU_FORCE_LANE_ORDER_FAILURE");
    }
    c.ready = true;
    return 0;
}

void ctx_release(Ctx& c)
{
    if (!c.ready) return;
    (void)hipSetDevice(c.device);
    (void)hipDeviceSynchronize();
    for (Stage& s : c.stage) {
        if (s.h_in) (void)hipHostFree(s.h_in);
        if (s.h_out) (void)hipHostFree(s.h_out);
        if (s.h_sizes) (void)hipHostFree(s.h_sizes);
        if (s.h_offsets) (void)hipHostFree(s.h_offsets);
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_slots) (void)hipFree(s.d_slots);
        if (s.d_packed) (void)hipFree(s.d_packed);
        if (s.d_sizes) (void)hipFree(s.d_sizes);
        if (s.d_offsets) (void)hipFree(s.d_offsets);
        if (s.up) (void)hipEventDestroy(s.up);
        if (s.k0) (void)hipEventDestroy(s.k0);
        if (s.k1) (void)hipEventDestroy(s.k1);
        if (s.meta) (void)hipEventDestroy(s.meta);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        memset(&s, 0, sizeof s);
    }
    lzk_combiner_free(&c);
    for (int i = 0; i < c.nExtra; i++) {
        LzArena& x = c.extra[i];
        if (x.scratch) (void)hipFree(x.scratch);
        if (x.counter) (void)hipFree(x.counter);
        if (x.tables) (void)hipFree(x.tables);
        if (x.pfTables) (void)hipFree(x.pfTables);
        if (x.ev0) (void)hipEventDestroy(x.ev0);
        if (x.ev1) (void)hipEventDestroy(x.ev1);
        memset(&x, 0, sizeof x);
    }
    c.nExtra = 0;
    if (c.tables) (void)hipFree(c.tables);
    if (c.pfTables) (void)hipFree(c.pfTables);
    if (c.hcSlots) (void)hipFree(c.hcSlots);
    c.hcNSlots = 0; c.hcHasBest = 0; c.hcSlotBytes = 0;
    if (c.scratch) (void)hipFree(c.scratch);
    if (c.counter) (void)hipFree(c.counter);
    if (c.ev0) (void)hipEventDestroy(c.ev0);
    if (c.ev1) (void)hipEventDestroy(c.ev1);
    c.tables = c.pfTables = c.hcSlots = c.scratch = nullptr; c.counter = nullptr; c.hcMaxBlock = 0;
    c.ev0 = c.ev1 = c.lastEv0 = c.lastEv1 = nullptr; c.timed = 0; c.laneOrderOk = 1; c.ready = 0;
    c.tablesSlots = c.pfSlots = 0; c.devBytes = 0; c.idleLaunches = 0;
}

void free_extra_arenas(Ctx& c)                                   // (their launches have finished: caller's business)
{
    for (int i = 0; i < c.nExtra; i++) {
        LzArena& x = c.extra[i];
        if (c.lastEv0 == x.ev0 || c.lastEv1 == x.ev1) { c.lastEv0 = c.ev0; c.lastEv1 = c.ev1; }      // (LizardGPU_lastKernelMs must not look at destroyed events)
        dev_free(c, x.scratch, (size_t)c.cus * kScratchBytes);
        if (x.counter) (void)hipFree(x.counter);
        dev_free(c, x.tables, x.tablesSlots * LZ_TABWIDE_BYTES(18));
        dev_free(c, x.pfTables, x.pfSlots * LZ_PF_SLOT_BYTES);
        if (x.ev0) (void)hipEventDestroy(x.ev0);
        if (x.ev1) (void)hipEventDestroy(x.ev1);
        memset(&x, 0, sizeof x);
    }
    c.nExtra = 0; c.nextExtra = 0;
}
// Everything but the context's own arena: extra arenas, per-wave tables, hashChain work areas (`keep`: 0 none, 1 tables, 2 pfTables,
// 3 hcSlots).  The device is idle afterwards (synchronised first: other launches may still be using what is freed).
void free_tables_except(Ctx& c, int keep)
{
    (void)hipDeviceSynchronize();
    free_extra_arenas(c);
    if (keep != 1 && c.tables) { dev_free(c, c.tables, c.tablesSlots * LZ_TABWIDE_BYTES(18)); c.tables = nullptr; c.tablesSlots = 0; }
    if (keep != 2 && c.pfTables) { dev_free(c, c.pfTables, c.pfSlots * LZ_PF_SLOT_BYTES); c.pfTables = nullptr; c.pfSlots = 0; }
    if (keep != 3 && c.hcSlots) { dev_free(c, c.hcSlots, c.hcNSlots * c.hcSlotBytes); c.hcSlots = nullptr; c.hcNSlots = 0; c.hcMaxBlock = 0; c.hcHasBest = 0; c.hcSlotBytes = 0; }
}
// Per-wave table slots of `stride` bytes under the budget: as many as resident waves if they fit — after giving up what other
// levels left behind, if need be — else as many as fit (the waves without one leave, lz_wave_main); none: out of memory.
int alloc_table_slots(Ctx& c, uint8_t** at, size_t* slotsAt, size_t stride, int kind, bool ownArena, const char* what)
{
    const size_t want = (size_t)c.cus * LZ_MAX_WAVES;
    if (want * stride > budget_room(c) && ownArena) free_tables_except(c, kind);
    size_t slots = budget_room(c) / stride;
    if (slots > want) slots = want;
    if (!slots) { snprintf(t_err, sizeof t_err, "%s: not one table of %zu bytes fits the memory budget of %zu bytes (%zu in use)", what, stride, g_budget.load(), c.devBytes); return -LIZARDGPU_ERR_NOMEM; }
    const int rc = dev_alloc(c, (void**)at, slots * stride, what);
    if (rc) return rc;
    *slotsAt = slots;
    return 0;
}

int clamp_level(int level)                                       // reference lizard_compress.c:303-308
{
    if (level > LIZARD_MAX_CLEVEL) level = LIZARD_MAX_CLEVEL;
    if (level < LIZARD_MIN_CLEVEL) level = LIZARD_DEFAULT_CLEVEL;
    return level;
}

// Largest block the GPU path takes at a level (0 = level not on the GPU path).  The table forms of the fast and priceFast
// parsers keep positions modulo a power of two and sweep (lz_block.h, lz_pricefast.h): any size the reference takes
// (lib/lizard_compress.h:121).  hashChain keeps full positions; its per-wave work area grows with the block (launch()).
// hashChain (13-17 / 34-38) and noChain (12 / 32 / 33: the same kernels with one candidate per search, lz_hashchain.h)
bool hc_level(int lv) { return (lv >= 12 && lv <= 17) || (lv >= 32 && lv <= 38); }

size_t level_max_block(int lv)
{
    const bool hcLevel = hc_level(lv);
    if (lv == 10 || lv == 30 || lv == 11 || lv == 31 || lv == 20 || lv == 40 || lv == 21 || lv == 41 || lv == 22 || lv == 42 || hcLevel) return (size_t)LIZARD_MAX_INPUT_SIZE;
    return 0;
}

// d_srcSizes / d_srcOffsets (may be nullptr, both or neither): a ragged batch — block b is d_srcSizes[b] bytes (1..blockSize) at
// d_src + d_srcOffsets[b]; lastBlockSize is ignored then (pass blockSize).
int launch(Ctx& c, const void* d_src, size_t nBlocks, size_t blockSize, size_t lastBlockSize, void* d_dst,
           size_t dstStride, u32* d_sizes, int level, hipStream_t stream, hipEvent_t k0 = nullptr, hipEvent_t k1 = nullptr,
           const u32* d_srcSizes = nullptr, const u64* d_srcOffsets = nullptr)
{
    const int lv = clamp_level(level);
    if (!LizardGPU_levelSupported(lv)) { snprintf(t_err, sizeof t_err, "level %d has no GPU kernel", lv); return -LIZARDGPU_ERR_LEVEL; }
    if (!d_src || !d_dst || !d_sizes || nBlocks == 0 || nBlocks > 0xFFFFFFFFu || blockSize == 0 || blockSize > LIZARD_MAX_INPUT_SIZE
        || lastBlockSize == 0 || lastBlockSize > blockSize || dstStride < (size_t)LIZARD_COMPRESSBOUND((int)blockSize)) {
        snprintf(t_err, sizeof t_err, "bad argument (null pointer, zero size, lastBlockSize > blockSize or dstStride < Lizard_compressBound)");
        return -LIZARDGPU_ERR_ARG;
    }
    if (blockSize > level_max_block(lv)) {
        snprintf(t_err, sizeof t_err, "level %d: blocks above %zu bytes are not supported on the GPU path", lv, level_max_block(lv));
        return -LIZARDGPU_ERR_ARG;
    }
    int rc = ctx_init(c);
    if (rc) return rc;
    if (!c.laneOrderOk && (lv == 10 || lv == 30 || hc_level(lv))) {
        snprintf(t_err, sizeof t_err, "level %d refused: device %d failed the self-check \"lanes of one LDS atomic are served in lane order\" its kernel relies on", lv, c.device);
        return -LIZARDGPU_ERR_HARDWARE;
    }
    LzBatch a;
    a.src = (const u8*)d_src; a.blockSize = blockSize; a.nBlocks = (u32)nBlocks; a.lastBlockSize = (u32)lastBlockSize;
    a.dst = (u8*)d_dst; a.dstStride = dstStride; a.sizes = d_sizes; a.level = (u32)lv;
    a.scratch = c.scratch; a.counter = c.counter; a.tables = nullptr; a.tableStride = 0; a.tableSlots = 0xFFFFFFFFu;
    a.srcSizes = d_srcSizes; a.srcOffsets = d_srcOffsets; a.activeWaves = 0xFFFFFFFFu;
    // one workgroup of W waves per CU; small batches launch only as many workgroups as they have blocks for
    const bool hcLevel = hc_level(lv);
    const bool fastMixed = blockSize <= (4u << 20);                         // global-table waves hold 22-bit positions
    const bool pfSmall = blockSize <= (256u << 10);                         // 18-bit LDS tables
    const bool huf = lv >= 30;
    u32 W;
    u32 perGroup = 0;                                                        // blocks in flight per workgroup when that is not W (producer / consumer form)
    if ((lv == 10 || lv == 30) && LZ_FAST12_SPLIT) { W = huf ? LZ_SPLIT_PROD_HUF + LZ_SPLIT_CONS_HUF : LZ_SPLIT_PROD + LZ_SPLIT_CONS; perGroup = huf ? LZ_SPLIT_PROD_HUF : LZ_SPLIT_PROD; }
    else if (lv == 10 || lv == 30) W = fastMixed ? (huf ? LZ_WAVES_FAST_HUF : LZ_WAVES_FAST) : (huf ? LZ_WAVES_FASTLDS_HUF : LZ_WAVES_FASTLDS);
    else if (lv == 11 || lv == 31) W = LZ_WAVES_FAST18;
    else if (hcLevel)              W = LZ_WAVES_HC;
    else if (lv == 21 || lv == 41) W = pfSmall ? (huf ? LZ_PF18_W_HUF : LZ_PF18_W) : LZ_PF_W;
    else if (lv == 20 || lv == 40) W = LZ_WAVES_FASTBIG;
    else                           W = LZ_PF22_W;
    if (!perGroup) perGroup = W;
    // Which arena?  (lizard_gpu_ctx.h, LzArena.)  The context's own unless this is a small launch on another stream than the one
    // that used it last and that launch is still running; then: an extra arena last used by this stream, or an idle one, or a new
    // one, or — all busy, none left to make — the extra ones in turn.
    LzArena* ar = nullptr;                                                   // nullptr: the context's own
    const bool smallLaunch = nBlocks < (size_t)c.cus * perGroup;
    if (smallLaunch && !hcLevel && c.maxArenas > 1 && c.timed && c.lastStream != stream && hipEventQuery(c.ev1) != hipSuccess) {
        for (int i = 0; i < c.nExtra && !ar; i++) if (c.extra[i].lastStream == stream) ar = &c.extra[i];
        for (int i = 0; i < c.nExtra && !ar; i++) if (!c.extra[i].timed || hipEventQuery(c.extra[i].ev1) == hipSuccess) ar = &c.extra[i];   // (anything but success: busy)
        if (!ar && c.nExtra < c.maxArenas - 1 && budget_room(c) >= (size_t)c.cus * kScratchBytes) {
            LzArena& x = c.extra[c.nExtra];
            memset(&x, 0, sizeof x);
            hipError_t e = hipMalloc((void**)&x.scratch, (size_t)c.cus * kScratchBytes);
            if (e == hipSuccess) c.devBytes += (size_t)c.cus * kScratchBytes;
            if (e == hipSuccess) e = hipMalloc((void**)&x.counter, 64);
            if (e == hipSuccess) e = hipEventCreate(&x.ev0);
            if (e == hipSuccess) e = hipEventCreate(&x.ev1);
            if (e == hipSuccess) {
                ar = &x; c.nExtra++;
                if (getenv("LIZARDGPU_VERBOSE")) fprintf(stderr, "liblizard_amd: device %d: arena %d for small launches on concurrent streams (LIZARDGPU_ARENAS caps them)\n", c.device, c.nExtra + 1);
            } else {                                                         // no room: share the context's own after all
                (void)hipGetLastError();
                dev_free(c, x.scratch, (size_t)c.cus * kScratchBytes);
                if (x.counter) (void)hipFree(x.counter);
                if (x.ev0) (void)hipEventDestroy(x.ev0);
                if (x.ev1) (void)hipEventDestroy(x.ev1);
                memset(&x, 0, sizeof x);
            }
        }
        if (!ar && c.nExtra) { ar = &c.extra[c.nextExtra % c.nExtra]; c.nextExtra++; }
        (void)hipGetLastError();                                             // (hipEventQuery's "not ready" is not an error to report)
    }
    // extra arenas that no launch has asked for in a while go back (64 launches on the context's own arena, all of theirs finished)
    if (ar) c.idleLaunches = 0;
    else if (c.nExtra && smallLaunch && !hcLevel && ++c.idleLaunches >= 64) {   // (launches that could have asked for one: large and hashChain launches say nothing about the small streams)
        bool idle = true;
        for (int i = 0; i < c.nExtra; i++) if (c.extra[i].timed && hipEventQuery(c.extra[i].ev1) != hipSuccess) idle = false;
        (void)hipGetLastError();
        if (idle) free_extra_arenas(c);
        c.idleLaunches = 0;
    }
    uint8_t** tablesAt = ar ? &ar->tables : &c.tables;
    uint8_t** pfTablesAt = ar ? &ar->pfTables : &c.pfTables;
    size_t* tablesSlotsAt = ar ? &ar->tablesSlots : &c.tablesSlots;
    size_t* pfSlotsAt = ar ? &ar->pfSlots : &c.pfSlots;
    if (ar) { a.scratch = ar->scratch; a.counter = ar->counter; }
    // An extra arena whose tables do not fit the memory budget any more (the hashChain work areas or the context's own tables took
    // the room): the launch goes to the context's own arena after all — it waits for the launch that holds it, and may give up
    // what other levels left behind (ownArena) — instead of failing where it would have succeeded on a quiet device.
    auto own_arena_after_all = [&]() {
        (void)hipGetLastError();
        ar = nullptr; a.scratch = c.scratch; a.counter = c.counter;
        tablesAt = &c.tables; pfTablesAt = &c.pfTables; tablesSlotsAt = &c.tablesSlots; pfSlotsAt = &c.pfSlots;
    };
    if (hcLevel) {
        // per wave: bins, links, and 6 bytes + 1 bit per block position (chain, packed chain words, hit bits) — 10 bytes at levels
        // 16/17/37/38, whose first searches are decided ahead of the parse (best[], the LAST array of the slot: the other levels'
        // slots simply end before it).  One slot per resident wave while that fits in about half of the free memory (256 KiB
        // blocks: 14 GiB, 18 GiB with best[]; 4 MiB blocks: 110 GiB, or as many slots as fit under the 128 GiB cap with best[]);
        // larger blocks get as many slots as fit and the other waves leave (lz_wave_main).
        const size_t cap = (blockSize + 65535u) & ~(size_t)65535u;
        const bool needBest = lv == 16 || lv == 17 || lv == 37 || lv == 38;
        if (!c.hcSlots || c.hcMaxBlock < cap || (needBest && !c.hcHasBest)) {
            const bool withBest = needBest || c.hcHasBest;
            const size_t capA = cap > c.hcMaxBlock ? cap : c.hcMaxBlock;                 // (a re-allocation never shrinks what the slots take)
            const size_t slotBytes = LZ_HC_SLOT_BYTES(capA) - (withBest ? 0u : 4u * capA);
            if (c.hcSlots) { LZ_HIP(hipDeviceSynchronize()); dev_free(c, c.hcSlots, c.hcNSlots * c.hcSlotBytes); c.hcSlots = nullptr; c.hcNSlots = 0; }
            size_t freeB = 0, totalB = 0;
            LZ_HIP(hipMemGetInfo(&freeB, &totalB));
            size_t budget = freeB / 2u;
            if (budget > ((size_t)128 << 30)) budget = (size_t)128 << 30;
            if (g_budget.load(std::memory_order_relaxed)) {      // LizardGPU_setMemoryBudget: what other levels left behind goes first
                if (budget_room(c) < (size_t)c.cus * LZ_MAX_WAVES * slotBytes) free_tables_except(c, 3);
                if (budget > budget_room(c)) budget = budget_room(c);
            }
            // LIZARDGPU_HC_WORKAREA_MB caps the reservation (a caller that shares the device with other allocations); fewer work
            // areas than resident waves only means fewer hashChain blocks in flight (the waves without one leave)
            if (const char* e = getenv("LIZARDGPU_HC_WORKAREA_MB")) {
                const size_t mb = (size_t)strtoull(e, nullptr, 10);
                if (mb > 0 && (mb << 20) < budget) budget = mb << 20;
            }
            size_t nSlots = budget / slotBytes;
            if (nSlots > (size_t)c.cus * LZ_MAX_WAVES) nSlots = (size_t)c.cus * LZ_MAX_WAVES;
            if (nSlots == 0) { snprintf(t_err, sizeof t_err, "level %d: no room for a hashChain work area of %zu bytes", lv, slotBytes); return -LIZARDGPU_ERR_NOMEM; }
            { const int rc_ = dev_alloc(c, (void**)&c.hcSlots, nSlots * slotBytes, "hashChain work areas"); if (rc_) return rc_; }
            c.hcMaxBlock = capA; c.hcNSlots = nSlots; c.hcHasBest = withBest; c.hcSlotBytes = slotBytes;
            if (getenv("LIZARDGPU_VERBOSE"))
                fprintf(stderr, "liblizard_amd: device %d: hashChain levels reserve %zu work areas of %zu bytes (%.1f GiB of %.1f GiB free; LIZARDGPU_HC_WORKAREA_MB caps it)\n",
                        c.device, nSlots, slotBytes, (double)(nSlots * slotBytes) / (double)(1u << 30), (double)freeB / (double)(1u << 30));
        }
        a.tables = c.hcSlots; a.tableStride = c.hcSlotBytes; a.tableSlots = (u32)c.hcNSlots;
    } else if (lv == 11 || lv == 31 || lv == 22 || lv == 42) {
        if (!*tablesAt && ar && alloc_table_slots(c, tablesAt, tablesSlotsAt, LZ_TABWIDE_BYTES(18), 1, false, "tables of levels 11/31/22/42")) own_arena_after_all();
        if (!*tablesAt && (rc = alloc_table_slots(c, tablesAt, tablesSlotsAt, LZ_TABWIDE_BYTES(18), 1, !ar, "tables of levels 11/31/22/42"))) return rc;
        a.tables = *tablesAt; a.tableStride = LZ_TABWIDE_BYTES(18); a.tableSlots = (u32)*tablesSlotsAt;
    } else if (!((lv == 10 || lv == 30) && LZ_FAST12_SPLIT)) {             // (the producer / consumer form keeps every table in LDS)
        if (!*pfTablesAt && ar && alloc_table_slots(c, pfTablesAt, pfSlotsAt, LZ_PF_SLOT_BYTES, 2, false, "tables of levels 21/41")) own_arena_after_all();
        if (!*pfTablesAt && (rc = alloc_table_slots(c, pfTablesAt, pfSlotsAt, LZ_PF_SLOT_BYTES, 2, !ar, "tables of levels 21/41"))) return rc;
        a.tables = *pfTablesAt; a.tableStride = LZ_PF_SLOT_BYTES; a.tableSlots = (u32)*pfSlotsAt;
    }
    u32 grid = (u32)((nBlocks + perGroup - 1) / perGroup);
    if (grid > (u32)c.cus) grid = (u32)c.cus;
#if LZ_SPREAD_SMALL
    // A launch with fewer blocks than the chip has block-claiming waves is spread over ALL CUs, ceil(nBlocks / grid) claiming
    // waves per workgroup, instead of filling nBlocks / perGroup CUs to the brim: 256 blocks are 256 CUs x 1 wave, not 20 x 13.
    if (nBlocks < (size_t)c.cus * perGroup) {
        grid = nBlocks < (size_t)c.cus ? (u32)nBlocks : (u32)c.cus;
        a.activeWaves = (u32)((nBlocks + grid - 1) / grid);
    }
#endif
    // The scratch arena, the tables and the block counter are shared by all launches on this device: a launch
    // on another stream first waits (on the GPU) for the previous one to finish.
    hipEvent_t const e0 = ar ? ar->ev0 : c.ev0, e1 = ar ? ar->ev1 : c.ev1;
    // (always, also on the stream that used the arena last: a destroyed stream's handle can come back for a new stream, and a
    //  same-stream wait costs nothing)
    if (ar ? ar->timed != 0 : c.timed != 0) LZ_HIP(hipStreamWaitEvent(stream, e1, 0));
    LZ_HIP(hipMemsetAsync(a.counter, 0, 4, stream));
    LZ_HIP(hipEventRecord(e0, stream));
    if (k0) LZ_HIP(hipEventRecord(k0, stream));
    const dim3 g(grid), t(64 * W);
    switch (lv) {
#if LZ_FAST12_SPLIT
    case 10: hipLaunchKernelGGL(lz_fast12_split_kernel<false>, g, t, 0, stream, a); break;
    case 30: hipLaunchKernelGGL(lz_fast12_split_kernel<true>, g, t, 0, stream, a); break;
#else
    case 10: if (fastMixed) hipLaunchKernelGGL((lz_fast12_kernel<false, true>), g, t, 0, stream, a);
             else           hipLaunchKernelGGL((lz_fast12_kernel<false, false>), g, t, 0, stream, a);
             break;
    case 30: if (fastMixed) hipLaunchKernelGGL((lz_fast12_kernel<true, true>), g, t, 0, stream, a);
             else           hipLaunchKernelGGL((lz_fast12_kernel<true, false>), g, t, 0, stream, a);
             break;
#endif
    case 11: hipLaunchKernelGGL(lz_fast18_kernel<false>, g, t, 0, stream, a); break;
    case 31: hipLaunchKernelGGL(lz_fast18_kernel<true>, g, t, 0, stream, a); break;
    case 12:                   hipLaunchKernelGGL((lz_hashchain_kernel<false, 6>), g, t, 0, stream, a); break;
    case 13:                   hipLaunchKernelGGL((lz_hashchain_kernel<false, 7>), g, t, 0, stream, a); break;
    case 14:                   hipLaunchKernelGGL((lz_hashchain_kernel<false, 8>), g, t, 0, stream, a); break;
    case 15:                   hipLaunchKernelGGL((lz_hashchain_kernel<false, 9>), g, t, 0, stream, a); break;
    case 16: case 17:          hipLaunchKernelGGL((lz_hashchain_kernel<false, 4>), g, t, 0, stream, a); break;
    case 32:                   hipLaunchKernelGGL((lz_hashchain_kernel<true, 6, 14>), g, t, 0, stream, a); break;
    case 33:                   hipLaunchKernelGGL((lz_hashchain_kernel<true, 6>), g, t, 0, stream, a); break;
    case 34:                   hipLaunchKernelGGL((lz_hashchain_kernel<true, 7>), g, t, 0, stream, a); break;
    case 35:                   hipLaunchKernelGGL((lz_hashchain_kernel<true, 8>), g, t, 0, stream, a); break;
    case 36:                   hipLaunchKernelGGL((lz_hashchain_kernel<true, 9>), g, t, 0, stream, a); break;
    case 37: case 38:          hipLaunchKernelGGL((lz_hashchain_kernel<true, 4>), g, t, 0, stream, a); break;
    case 20: hipLaunchKernelGGL(lz_fastbig14_kernel<false>, g, t, 0, stream, a); break;
    case 40: hipLaunchKernelGGL(lz_fastbig14_kernel<true>, g, t, 0, stream, a); break;
    case 22: hipLaunchKernelGGL(lz_pricefast18_kernel<false>, g, t, 0, stream, a); break;
    case 42: hipLaunchKernelGGL(lz_pricefast18_kernel<true>, g, t, 0, stream, a); break;
    case 21: if (pfSmall) hipLaunchKernelGGL((lz_pricefast14_kernel<false, true>), g, t, 0, stream, a);
             else         hipLaunchKernelGGL((lz_pricefast14_kernel<false, false>), g, t, 0, stream, a);
             break;
    default: if (pfSmall) hipLaunchKernelGGL((lz_pricefast14_kernel<true, true>), g, t, 0, stream, a);
             else         hipLaunchKernelGGL((lz_pricefast14_kernel<true, false>), g, t, 0, stream, a);
             break;
    }
    LZ_HIP(hipGetLastError());
    LZ_HIP(hipEventRecord(e1, stream));
    if (k1) LZ_HIP(hipEventRecord(k1, stream));
    if (ar) { ar->timed = 1; ar->lastStream = stream; } else { c.timed = true; c.lastStream = stream; }
    c.lastEv0 = e0; c.lastEv1 = e1;
    c.lastSplit = LZ_FAST12_SPLIT && (lv == 10 || lv == 30);
    return 0;
}

int launch_decompress(Ctx& c, const void* d_src, const u64* d_offsets, size_t srcStride, const u32* d_srcSizes, size_t nBlocks,
                      void* d_dst, size_t dstStride, u32* d_outSizes, hipStream_t stream)
{
    if (!d_src || !d_dst || !d_outSizes || (!d_offsets && !d_srcSizes) || nBlocks == 0 || nBlocks > 0xFFFFFFFFu || dstStride == 0) {
        snprintf(t_err, sizeof t_err, "bad argument (null pointer or zero size)");
        return -LIZARDGPU_ERR_ARG;
    }
    int rc = ctx_init(c);
    if (rc) return rc;
    LzUnBatch a;
    a.src = (const u8*)d_src; a.offsets = d_offsets; a.srcStride = srcStride; a.srcSizes = d_srcSizes;
    a.dst = (u8*)d_dst; a.dstStride = dstStride; a.outSizes = d_outSizes; a.nBlocks = (u32)nBlocks;
    a.scratch = c.scratch; a.counter = c.counter;
    u32 grid = (u32)((nBlocks + LZ_WAVES_DEC - 1) / LZ_WAVES_DEC);
    if (grid > (u32)c.cus) grid = (u32)c.cus;
    if (c.timed) LZ_HIP(hipStreamWaitEvent(stream, c.ev1, 0));      // scratch arena and counter are shared with the compress launches
    LZ_HIP(hipMemsetAsync(c.counter, 0, 4, stream));
    LZ_HIP(hipEventRecord(c.ev0, stream));
    hipLaunchKernelGGL(lz_decompress_kernel, dim3(grid), dim3(64 * LZ_WAVES_DEC), 0, stream, a);
    LZ_HIP(hipGetLastError());
    LZ_HIP(hipEventRecord(c.ev1, stream));
    c.timed = true; c.lastStream = stream; c.lastEv0 = c.ev0; c.lastEv1 = c.ev1;
    c.hostKernelMs = -1.0f;
    return 0;
}

}  // namespace

#include "lizard_shard.h"   // single-process multi-device entry + RCCL size gather (uses Guard / launch above)

extern "C" {

int LizardGPU_levelSupported(int level)
{
    return level_max_block(clamp_level(level)) != 0;
}

size_t LizardGPU_maxBlockSize(int level) { return level_max_block(clamp_level(level)); }

int LizardGPU_deviceCount(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return count;
}

int LizardGPU_setDevice(int device)
{
    t_err[0] = 0;
    const int count = LizardGPU_deviceCount();
    if (device < 0 || device >= LZ_MAX_DEVICES || (count > 0 && device >= count)) {
        snprintf(t_err, sizeof t_err, "LizardGPU_setDevice(%d): %d device(s) visible", device, count);
        return -LIZARDGPU_ERR_ARG;
    }
    t_device = device;
    pthread_mutex_lock(&g_sel_mu);
    g_default_device = device;
    pthread_mutex_unlock(&g_sel_mu);
    return 0;
}

const char* LizardGPU_lastError(void) { return t_err; }

int LizardGPU_residentWaves(void)
{
    Guard g;
    if (g.rc) return g.rc;
    int rc = ctx_init(*g.c);
    return rc ? rc : g.c->cus * (LZ_FAST12_SPLIT ? LZ_SPLIT_PROD : LZ_WAVES_FAST);      // level-10 residency: blocks in flight = waves with a hash table (13 per CU, all in LDS)
}

static void release_all_contexts(size_t newBudget, bool setBudget)
{
    int saved = -1;
    if (hipGetDevice(&saved) != hipSuccess) saved = -1;
    // every context is quiesced and locked BEFORE the budget changes, and released under the new one: no launch on any device
    // runs with holdings counted against the old budget while it reads the new (round 5 stored it under context 0's lock alone)
    for (int d = 0; d < LZ_MAX_DEVICES; d++) { lzk_combiner_quiesce(&g_ctx[d]); pthread_mutex_lock(&g_ctx[d].mu); }
    if (setBudget) g_budget.store(newBudget, std::memory_order_relaxed);
    for (int d = 0; d < LZ_MAX_DEVICES; d++) ctx_release(g_ctx[d]);
    for (int d = LZ_MAX_DEVICES - 1; d >= 0; d--) { pthread_mutex_unlock(&g_ctx[d].mu); lzk_combiner_resume(&g_ctx[d]); }
    if (saved >= 0) (void)hipSetDevice(saved);
}

void LizardGPU_shutdown(void)
{
    release_all_contexts(0, false);
    lz_shard_shutdown();
}

int LizardGPU_setMemoryBudget(size_t bytes)
{
    t_err[0] = 0;
    if (bytes) {
        int count = 0, cus = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) count = 0;
        for (int d = 0; d < count; d++) {                   // the budget holds for every device: the floor is the largest one's
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > cus) cus = prop.multiProcessorCount;
        }
        if (!cus) {
            (void)hipGetLastError();
            snprintf(t_err, sizeof t_err, "no HIP device visible");
            return -LIZARDGPU_ERR_NO_DEVICE;
        }
        const size_t floor_ = (size_t)cus * kScratchBytes + ((size_t)256 << 20);
        if (bytes < floor_) {
            snprintf(t_err, sizeof t_err, "LizardGPU_setMemoryBudget(%zu): below the minimum of %zu bytes (one scratch arena of %zu bytes + 256 MiB)", bytes, floor_, floor_ - ((size_t)256 << 20));
            return -LIZARDGPU_ERR_ARG;
        }
    }
    release_all_contexts(bytes, true);                  // what the contexts hold goes back; the next call allocates under the new budget
    return 0;
}

size_t LizardGPU_memoryBudget(void) { return g_budget.load(std::memory_order_relaxed); }

size_t LizardGPU_memoryInUse(void)
{
    Guard g;
    return g.rc ? 0 : g.c->devBytes;
}

// Give back what is idle on the selected device: extra arenas, per-wave tables, hashChain work areas, the staging buffers of the
// host-buffer entries and of the one-block combiner.  The context's own scratch arena stays.  Waits for the device to be idle.
int LizardGPU_trim(void)
{
    LzCtx* const cpeek = lzk_ctx_peek();
    if (!cpeek) return -LIZARDGPU_ERR_NO_DEVICE;
    lzk_combiner_quiesce(cpeek);
    int rc = 0;
    {
        Guard g;
        rc = g.rc;
        if (!rc && g.c->ready) {
            Ctx& c = *g.c;
            free_tables_except(c, 0);
            for (Stage& s : c.stage) {
                dev_free(c, s.d_in, s.d_in_cap); dev_free(c, s.d_slots, s.d_slots_cap); dev_free(c, s.d_packed, s.d_packed_cap);
                s.d_in = s.d_slots = s.d_packed = nullptr; s.d_in_cap = s.d_slots_cap = s.d_packed_cap = 0;
                if (s.h_in) (void)hipHostFree(s.h_in);
                if (s.h_out) (void)hipHostFree(s.h_out);
                s.h_in = s.h_out = nullptr; s.h_in_cap = s.h_out_cap = 0;
            }
            if (g.c == cpeek) lzk_combiner_free(&c);
        }
    }
    lzk_combiner_resume(cpeek);
    return rc;
}

int LizardGPU_compressBlocks_device(const void* d_src, size_t nBlocks, size_t blockSize, size_t lastBlockSize,
                                    void* d_dst, size_t dstStride, uint32_t* d_sizes, int level, void* stream)
{
    Guard g;
    if (g.rc) return g.rc;
    g.c->hostKernelMs = -1.0f;
    return launch(*g.c, d_src, nBlocks, blockSize, lastBlockSize, d_dst, dstStride, d_sizes, level, (hipStream_t)stream);
}

int LizardGPU_decompressBlocks_device(const void* d_src, size_t srcStride, const uint32_t* d_srcSizes, size_t nBlocks,
                                      void* d_dst, size_t dstStride, uint32_t* d_outSizes, void* stream)
{
    Guard g;
    if (g.rc) return g.rc;
    return launch_decompress(*g.c, d_src, nullptr, srcStride, d_srcSizes, nBlocks, d_dst, dstStride, d_outSizes, (hipStream_t)stream);
}

// ---- the shim of lizard_gpu_ctx.h: what the host C files (lizard_pipeline_host.c) need from this side ----
void  lzk_guard_acquire(LzGuard* g) { guard_acquire(*g); }
void  lzk_guard_release(LzGuard* g) { guard_release(*g); }
char* lzk_err(void) { return t_err; }
int   lzk_ctx_init(LzCtx* c) { return ctx_init(*c); }
int   lzk_clamp_level(int level) { return clamp_level(level); }
int   lzk_launch(LzCtx* c, const void* d_src, size_t nBlocks, size_t blockSize, size_t lastBlockSize, void* d_dst, size_t dstStride,
                 uint32_t* d_sizes, int level, hipStream_t stream, hipEvent_t k0, hipEvent_t k1, const uint32_t* d_srcSizes,
                 const uint64_t* d_srcOffsets)
{
    return launch(*c, d_src, nBlocks, blockSize, lastBlockSize, d_dst, dstStride, d_sizes, level, stream, k0, k1, d_srcSizes, (const u64*)d_srcOffsets);
}
int    lzk_dev_alloc(LzCtx* c, void** p, size_t bytes)
{
    // staging of a host-buffer call (context locked by the caller): under a budget, the tables and work areas earlier launches left
    // behind make room for it (this call's own launch gets as many table slots as still fit)
    if (bytes > budget_room(*c)) free_tables_except(*c, 0);
    return dev_alloc(*c, p, bytes, "staging buffer");
}
void   lzk_dev_free(LzCtx* c, void* p, size_t bytes) { dev_free(*c, p, bytes); }
size_t lzk_budget(void) { return g_budget.load(std::memory_order_relaxed); }
size_t lzk_budget_room_for_staging(const LzCtx* c)
{
    const size_t b = g_budget.load(std::memory_order_relaxed), arena = (size_t)c->cus * kScratchBytes;
    return !b ? (size_t)-1 : (b > arena ? b - arena : 0);
}
LzCtx* lzk_ctx_peek(void)
{
    int count = 0;
    t_err[0] = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        snprintf(t_err, sizeof t_err, "no HIP device visible");
        return nullptr;
    }
    const int dev = selected_device();
    if (dev < 0 || dev >= count || dev >= LZ_MAX_DEVICES) { snprintf(t_err, sizeof t_err, "device %d out of range (%d visible)", dev, count); return nullptr; }
    return &g_ctx[dev];
}
int   lzk_launch_decompress(LzCtx* c, const void* d_src, const uint64_t* d_offsets, size_t srcStride, const uint32_t* d_srcSizes,
                            size_t nBlocks, void* d_dst, size_t dstStride, uint32_t* d_outSizes, hipStream_t stream)
{
    return launch_decompress(*c, d_src, (const u64*)d_offsets, srcStride, d_srcSizes, nBlocks, d_dst, dstStride, d_outSizes, stream);
}
void  lzk_pack_launch(const void* d_in, const void* d_slots, size_t slot, const uint32_t* d_sizes, uint64_t* d_offsets, void* d_packed,
                      uint32_t nb, uint32_t blockSize, uint32_t lastBlockSize, int mode, hipStream_t stream)
{
    static_assert(LZK_PACK_PAYLOAD == LZ_PACK_PAYLOAD && LZK_PACK_FRAME == LZ_PACK_FRAME, "pack modes");
    lz_pack_launch((const u8*)d_in, (const u8*)d_slots, slot, d_sizes, (u64*)d_offsets, (u8*)d_packed, nb, blockSize, lastBlockSize, mode, stream);
}

#ifdef LZ_PROFILE
// Profile builds only: sum of the per-wave phase clocks since the last call (scratch slot tails), then reset.
int LizardGPU_profileDump(unsigned long long out[16])
{
    Guard g;
    for (int k = 0; k < 16; k++) out[k] = 0;
    if (g.rc) return g.rc;
    Ctx& c = *g.c;
    int rc = ctx_init(c);
    if (rc) return rc;
    (void)hipDeviceSynchronize();
    for (int w = 0; w < c.cus * LZ_MAX_WAVES; w++) {
        unsigned long long v[16];
        // the one-wave-per-block kernels keep a wave's record at the tail of its scratch slot; the producer / consumer kernels of
        // levels 10 / 30 (whose sequence buffers cover those tails) at the end of the workgroup's arena
        const size_t group = (size_t)(w / LZ_MAX_WAVES) * LZ_MAX_WAVES * LZ_SCRATCH_BYTES;
        u8* at = c.lastSplit ? c.scratch + group + (size_t)LZ_MAX_WAVES * LZ_SCRATCH_BYTES - (size_t)(w % LZ_MAX_WAVES + 1) * 128
                             : c.scratch + (size_t)(w + 1) * LZ_SCRATCH_BYTES - 128;
        if (hipMemcpy(v, at, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -LIZARDGPU_ERR_HIP;
        for (int k = 0; k < 15; k++) out[k] += v[k];
        (void)hipMemset(at, 0, sizeof v);
    }
    return 0;
}
#endif

int LizardGPU_arenasInUse(void)
{
    Guard g;
    if (g.rc) return 0;
    return g.c->ready ? 1 + g.c->nExtra : 0;
}

float LizardGPU_lastKernelMs(void)
{
    Guard g;
    if (g.rc) return -1.0f;
    Ctx& c = *g.c;
    if (c.hostKernelMs >= 0.0f) return c.hostKernelMs;
    float ms = -1.0f;
    if (c.lastEv1 && c.timed && hipEventSynchronize(c.lastEv1) == hipSuccess) {
        if (hipEventElapsedTime(&ms, c.lastEv0, c.lastEv1) != hipSuccess) ms = -1.0f;
    }
    return ms;
}

}  // extern "C"
