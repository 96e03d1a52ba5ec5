// lizard_gpu.hip — gfx950 kernels + the thin extern "C" shim the host C layer (lizard_host.c) calls.
//
// The kernels live in lz_kernels.h (one wavefront per Lizard API block, a persistent grid of one workgroup per CU).
// This file: one context PER DEVICE (arenas, tables, streams, pinned staging), a per-thread device
// selection and error text, the pipelined host-buffer path (pinned double-buffered staging, device-side
// compaction of the compressed blocks, one D2H per chunk) and the single-process multi-device entry with an
// RCCL all-gather of the per-block sizes (lizard_shard.h).
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/lizard_amd.h"
#include "lizard_gpu_shim.h"
#include "lz_kernels.h"   // LzBatch / LzUnBatch, residency knobs, the kernels

namespace {

// ------------------------------------------------------------------------------------------------
// One stage of the host-buffer pipeline: pinned staging on the host side, input / slot / packed buffers on the
// device side, its own stream.  Two stages alternate so that the copies of one chunk overlap the kernels of the other.
#define LZ_STAGES 3                                         // chunks in flight in the host-buffer pipeline
struct Stage {
    hipStream_t stream = nullptr;
    hipEvent_t  k0 = nullptr, k1 = nullptr, meta = nullptr, done = nullptr, up = nullptr;   // up: the chunk's input is on the device
    u8*  h_in = nullptr;     size_t h_in_cap = 0;        // pinned
    u8*  h_out = nullptr;    size_t h_out_cap = 0;       // pinned
    u32* h_sizes = nullptr;  u64* h_offsets = nullptr;   size_t h_meta_cap = 0;   // pinned, nBlocks (+1)
    u8*  d_in = nullptr;     size_t d_in_cap = 0;
    u8*  d_slots = nullptr;  size_t d_slots_cap = 0;
    u8*  d_packed = nullptr; size_t d_packed_cap = 0;
    u32* d_sizes = nullptr;  u64* d_offsets = nullptr;   size_t d_meta_cap = 0;
};

struct Ctx {
    bool  ready = false;
    int   device = -1;
    int   cus = 0;
    u8*   tables = nullptr;     // levels 11/31/22/42, allocated on first use
    u8*   pfTables = nullptr;   // levels 10/30/21/41: 64 KiB per resident wave for the waves whose table is not in LDS
    u8*   hcSlots = nullptr;    // hashChain levels, allocated (and zeroed) on first use / when a larger block size arrives
    size_t hcMaxBlock = 0;
    size_t hcNSlots = 0;
    u8*   scratch = nullptr;
    u32*  counter = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool  timed = false;
    bool  laneOrderOk = true;   // self-check at context creation: lanes of one DS atomic are served in lane order (lz_selfcheck_lane_order_kernel)
    float hostKernelMs = -1.0f; // sum over the chunks of the last host-buffer call (< 0: last call was a device call)
    Stage stage[LZ_STAGES];
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
};

#define LZ_MAX_DEVICES 16
Ctx g_ctx[LZ_MAX_DEVICES];
int g_default_device = 0;                              // process default (the last LizardGPU_setDevice of any thread)
pthread_mutex_t g_sel_mu = PTHREAD_MUTEX_INITIALIZER;
thread_local int  t_device = -1;                       // calling thread's selection, -1 = process default
thread_local char t_err[256] = {0};
size_t g_chunk_bytes = 0;                              // host pipeline chunk (input bytes), 0 = not read yet

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

int selected_device()
{
    if (t_device >= 0) return t_device;
    pthread_mutex_lock(&g_sel_mu);
    const int d = g_default_device;
    pthread_mutex_unlock(&g_sel_mu);
    return d;
}

// Locks the selected device's context and makes that device current for the calling thread (HIP's current device
// is per thread); restores the caller's device on exit.
struct Guard {
    Ctx* c = nullptr;
    int  saved = -1, rc = 0;
    Guard()
    {
        t_err[0] = 0;
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
            (void)hipGetLastError();
            snprintf(t_err, sizeof t_err, "no HIP device visible");
            rc = -LIZARDGPU_ERR_NO_DEVICE; return;
        }
        const int dev = selected_device();
        if (dev < 0 || dev >= count || dev >= LZ_MAX_DEVICES) {
            snprintf(t_err, sizeof t_err, "device %d out of range (%d visible)", dev, count);
            rc = -LIZARDGPU_ERR_ARG; return;
        }
        if (hipGetDevice(&saved) != hipSuccess) saved = -1;
        c = &g_ctx[dev];
        pthread_mutex_lock(&c->mu);
        if (hipSetDevice(dev) != hipSuccess) { snprintf(t_err, sizeof t_err, "hipSetDevice(%d) failed", dev); rc = -LIZARDGPU_ERR_HIP; return; }
        c->device = dev;
    }
    ~Guard()
    {
        if (c) pthread_mutex_unlock(&c->mu);
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};

int ctx_init(Ctx& c)
{
    if (c.ready) return 0;
    hipDeviceProp_t prop;
    LZ_HIP(hipGetDeviceProperties(&prop, c.device));
    c.cus = prop.multiProcessorCount;
    LZ_HIP(hipMalloc((void**)&c.scratch, (size_t)c.cus * LZ_MAX_WAVES * LZ_SCRATCH_BYTES));   // one slot per resident wave (one workgroup per CU)
#ifdef LZ_PROFILE
    LZ_HIP(hipMemset(c.scratch, 0, (size_t)c.cus * LZ_MAX_WAVES * LZ_SCRATCH_BYTES));
#endif
    LZ_HIP(hipMalloc((void**)&c.counter, 64));
    LZ_HIP(hipEventCreate(&c.ev0));
    LZ_HIP(hipEventCreate(&c.ev1));
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
                            "levels 10, 30, 13-17 and 34-38 are refused on this device\n", c.device, bad, bad ? "" : ", forced by LIZARDGPU_FORCE_LANE_ORDER_FAILURE");
    }
    c.ready = true;
    return 0;
}

template <typename T>
int ensure_dev(T** p, size_t* cap, size_t need)
{
    if (*cap >= need) return 0;
    if (*p) { LZ_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
    LZ_HIP(hipMalloc((void**)p, need));
    *cap = need;
    return 0;
}
template <typename T>
int ensure_pinned(T** p, size_t* cap, size_t need)
{
    if (*cap >= need) return 0;
    if (*p) { LZ_HIP(hipHostFree(*p)); *p = nullptr; *cap = 0; }
    LZ_HIP(hipHostMalloc((void**)p, need, hipHostMallocDefault));
    *cap = need;
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
        s = Stage();
    }
    if (c.tables) (void)hipFree(c.tables);
    if (c.pfTables) (void)hipFree(c.pfTables);
    if (c.hcSlots) (void)hipFree(c.hcSlots);
    c.hcNSlots = 0;
    if (c.scratch) (void)hipFree(c.scratch);
    if (c.counter) (void)hipFree(c.counter);
    if (c.ev0) (void)hipEventDestroy(c.ev0);
    if (c.ev1) (void)hipEventDestroy(c.ev1);
    c.tables = c.pfTables = c.hcSlots = c.scratch = nullptr; c.counter = nullptr; c.hcMaxBlock = 0;
    c.ev0 = c.ev1 = nullptr; c.timed = false; c.laneOrderOk = true; c.ready = false;
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
size_t level_max_block(int lv)
{
    const bool hcLevel = (lv >= 13 && lv <= 17) || (lv >= 34 && lv <= 38);
    if (lv == 10 || lv == 30 || lv == 11 || lv == 31 || lv == 21 || lv == 41 || lv == 22 || lv == 42 || hcLevel) return (size_t)LIZARD_MAX_INPUT_SIZE;
    return 0;
}

int launch(Ctx& c, const void* d_src, size_t nBlocks, size_t blockSize, size_t lastBlockSize, void* d_dst,
           size_t dstStride, u32* d_sizes, int level, hipStream_t stream, hipEvent_t k0 = nullptr, hipEvent_t k1 = nullptr)
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
    if (!c.laneOrderOk && (lv == 10 || lv == 30 || (lv >= 13 && lv <= 17) || (lv >= 34 && lv <= 38))) {
        snprintf(t_err, sizeof t_err, "level %d refused: device %d failed the self-check \"lanes of one LDS atomic are served in lane order\" its kernel relies on", lv, c.device);
        return -LIZARDGPU_ERR_HARDWARE;
    }
    LzBatch a;
    a.src = (const u8*)d_src; a.blockSize = blockSize; a.nBlocks = (u32)nBlocks; a.lastBlockSize = (u32)lastBlockSize;
    a.dst = (u8*)d_dst; a.dstStride = dstStride; a.sizes = d_sizes; a.level = (u32)lv;
    a.scratch = c.scratch; a.counter = c.counter; a.tables = nullptr; a.tableStride = 0; a.tableSlots = 0xFFFFFFFFu;
    // one workgroup of W waves per CU; small batches launch only as many workgroups as they have blocks for
    const bool hcLevel = (lv >= 13 && lv <= 17) || (lv >= 34 && lv <= 38);
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
    else                           W = LZ_PF22_W;
    if (hcLevel) {
        // per wave: bins, links, and 6 bytes + 1 bit per block position (chain, packed chain words, hit bits).  One slot per resident
        // wave while that fits in about half of the free memory (19 GiB for 256 KiB blocks, 110 GiB for 4 MiB blocks); larger
        // blocks get as many slots as fit and the other waves leave (lz_wave_main).
        const size_t cap = (blockSize + 65535u) & ~(size_t)65535u;
        if (!c.hcSlots || c.hcMaxBlock < cap) {
            if (c.hcSlots) { LZ_HIP(hipDeviceSynchronize()); LZ_HIP(hipFree(c.hcSlots)); c.hcSlots = nullptr; c.hcMaxBlock = 0; c.hcNSlots = 0; }
            size_t freeB = 0, totalB = 0;
            LZ_HIP(hipMemGetInfo(&freeB, &totalB));
            size_t budget = freeB / 2u;
            if (budget > ((size_t)128 << 30)) budget = (size_t)128 << 30;
            size_t nSlots = budget / LZ_HC_SLOT_BYTES(cap);
            if (nSlots > (size_t)c.cus * LZ_MAX_WAVES) nSlots = (size_t)c.cus * LZ_MAX_WAVES;
            if (nSlots == 0) { snprintf(t_err, sizeof t_err, "level %d: no room for a hashChain work area of %zu bytes", lv, (size_t)LZ_HC_SLOT_BYTES(cap)); return -LIZARDGPU_ERR_NOMEM; }
            LZ_HIP(hipMalloc((void**)&c.hcSlots, nSlots * LZ_HC_SLOT_BYTES(cap)));
            c.hcMaxBlock = cap; c.hcNSlots = nSlots;
        }
        a.tables = c.hcSlots; a.tableStride = LZ_HC_SLOT_BYTES(c.hcMaxBlock); a.tableSlots = (u32)c.hcNSlots;
    } else if (lv == 11 || lv == 31 || lv == 22 || lv == 42) {
        if (!c.tables) LZ_HIP(hipMalloc((void**)&c.tables, (size_t)c.cus * LZ_MAX_WAVES * LZ_TABWIDE_BYTES(18)));
        a.tables = c.tables; a.tableStride = LZ_TABWIDE_BYTES(18);
    } else {
        if (!c.pfTables) LZ_HIP(hipMalloc((void**)&c.pfTables, (size_t)c.cus * LZ_MAX_WAVES * LZ_PF_SLOT_BYTES));
        a.tables = c.pfTables; a.tableStride = LZ_PF_SLOT_BYTES;
    }
    if (!perGroup) perGroup = W;
    u32 grid = (u32)((nBlocks + perGroup - 1) / perGroup);
    if (grid > (u32)c.cus) grid = (u32)c.cus;
    // The scratch arena, the tables and the block counter are shared by all launches on this device: a launch
    // on another stream first waits (on the GPU) for the previous one to finish.
    if (c.timed) LZ_HIP(hipStreamWaitEvent(stream, c.ev1, 0));
    LZ_HIP(hipMemsetAsync(c.counter, 0, 4, stream));
    LZ_HIP(hipEventRecord(c.ev0, stream));
    if (k0) LZ_HIP(hipEventRecord(k0, stream));
    const dim3 g(grid), t(64 * W);
    switch (lv) {
    case 10: if (LZ_FAST12_SPLIT) hipLaunchKernelGGL(lz_fast12_split_kernel<false>, g, t, 0, stream, a);
             else if (fastMixed) hipLaunchKernelGGL((lz_fast12_kernel<false, true>), g, t, 0, stream, a);
             else           hipLaunchKernelGGL((lz_fast12_kernel<false, false>), g, t, 0, stream, a);
             break;
    case 30: if (LZ_FAST12_SPLIT) hipLaunchKernelGGL(lz_fast12_split_kernel<true>, g, t, 0, stream, a);
             else if (fastMixed) hipLaunchKernelGGL((lz_fast12_kernel<true, true>), g, t, 0, stream, a);
             else           hipLaunchKernelGGL((lz_fast12_kernel<true, false>), g, t, 0, stream, a);
             break;
    case 11: hipLaunchKernelGGL(lz_fast18_kernel<false>, g, t, 0, stream, a); break;
    case 31: hipLaunchKernelGGL(lz_fast18_kernel<true>, g, t, 0, stream, a); break;
    case 13: case 14: case 15: hipLaunchKernelGGL((lz_hashchain_kernel<false, 5>), g, t, 0, stream, a); break;
    case 16: case 17:          hipLaunchKernelGGL((lz_hashchain_kernel<false, 4>), g, t, 0, stream, a); break;
    case 34: case 35: case 36: hipLaunchKernelGGL((lz_hashchain_kernel<true, 5>), g, t, 0, stream, a); break;
    case 37: case 38:          hipLaunchKernelGGL((lz_hashchain_kernel<true, 4>), g, t, 0, stream, a); break;
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
    LZ_HIP(hipEventRecord(c.ev1, stream));
    if (k1) LZ_HIP(hipEventRecord(k1, stream));
    c.timed = true;
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
    c.timed = true;
    c.hostKernelMs = -1.0f;
    return 0;
}

// Host copies between caller memory and the pinned staging buffers are what bounds the PCIe-inclusive rate (one core moves
// ~10 GB/s): large copies are cut into slices for a few short-lived threads.
#ifndef LZ_COPY_THREADS
#define LZ_COPY_THREADS 4
#endif
struct CopyJob { void* d; const void* s; size_t n; };
void* copy_thread(void* a) { CopyJob* j = (CopyJob*)a; memcpy(j->d, j->s, j->n); return nullptr; }
void par_memcpy(void* dst, const void* src, size_t n)
{
    const int kThreads = LZ_COPY_THREADS;
    if (n < ((size_t)16 << 20)) { memcpy(dst, src, n); return; }
    pthread_t th[kThreads]; CopyJob job[kThreads]; bool started[kThreads];
    const size_t slice = ((n / kThreads) + 4095) & ~(size_t)4095;
    for (int i = 0; i < kThreads; i++) {
        const size_t off = (size_t)i * slice;
        job[i].d = (u8*)dst + off; job[i].s = (const u8*)src + off; job[i].n = off >= n ? 0 : (n - off < slice ? n - off : slice);
        started[i] = i > 0 && job[i].n && pthread_create(&th[i], nullptr, copy_thread, &job[i]) == 0;
    }
    for (int i = 0; i < kThreads; i++) if (!started[i] && job[i].n) memcpy(job[i].d, job[i].s, job[i].n);   // slice 0, and any slice whose thread did not start
    for (int i = 0; i < kThreads; i++) if (started[i]) pthread_join(th[i], nullptr);
}

bool is_pinned_host(const void* p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

size_t chunk_bytes()
{
    if (!g_chunk_bytes) {
        const char* e = getenv("LIZARDGPU_CHUNK_MB");
        size_t mb = e ? (size_t)strtoul(e, nullptr, 10) : 0;
        if (mb < 1 || mb > 65536) mb = 256;                  // measured on a 4 GiB job: 256 MiB 37 GB/s, 512 MiB 26, 1 GiB 23 (fill and drain of the pipeline)
        g_chunk_bytes = mb << 20;
    }
    return g_chunk_bytes;
}

// ---- the host-buffer pipeline -----------------------------------------------------------------------
// Input is cut into chunks of whole blocks.  Per chunk, on its stage's stream: host -> pinned staging (skipped when
// the caller's buffer is itself pinned) -> H2D -> block kernels -> exclusive scan of the record sizes -> compaction
// of the valid bytes into one packed buffer -> D2H of sizes/offsets, then of exactly the packed bytes.  The host
// then hands each chunk's result to `sink` (run_host_job_inner: an issuing and a draining thread, LZ_STAGES chunks in flight).
struct HostJob {
    const u8* src; size_t nBlocks, blockSize, lastBlockSize; int level;
    int mode;                                  // LZ_PACK_PAYLOAD or LZ_PACK_FRAME (lz_pack.h)
    // sink: chunk [first, first+nb) finished; packed bytes at `data` (size `bytes`), per-block offsets inside it
    // (offsets[nb] = bytes) and compressed sizes.  Returns 0 or a negative error.
    int (*sink)(void* user, size_t first, size_t nb, const u8* data, size_t bytes, const u64* offsets, const u32* sizes);
    void* user;
};

struct ChunkState { size_t first = 0, nb = 0, inBytes = 0, packedBytes = 0; bool active = false; };

int stage_issue(Ctx& c, Stage& s, const HostJob& j, ChunkState& ch, bool srcPinned, hipEvent_t prevUp)
{
    const size_t slot = ((size_t)LIZARD_COMPRESSBOUND((int)j.blockSize) + 63) & ~(size_t)63;
    const size_t last = (ch.first + ch.nb == j.nBlocks) ? j.lastBlockSize : j.blockSize;
    ch.inBytes = (ch.nb - 1) * j.blockSize + last;
    const size_t packedCap = ch.nb * (slot + 8);
    int rc;
    if ((rc = ensure_dev(&s.d_in, &s.d_in_cap, ch.inBytes + 64))) return rc;
    if ((rc = ensure_dev(&s.d_slots, &s.d_slots_cap, ch.nb * slot))) return rc;
    if ((rc = ensure_dev(&s.d_packed, &s.d_packed_cap, packedCap))) return rc;
    if ((rc = ensure_pinned(&s.h_out, &s.h_out_cap, packedCap + 64))) return rc;    // worst case once: a buffer that follows the chunks' sizes is re-pinned again and again
    if (s.d_meta_cap < ch.nb + 1) {
        if (s.d_sizes) { LZ_HIP(hipFree(s.d_sizes)); s.d_sizes = nullptr; }
        if (s.d_offsets) { LZ_HIP(hipFree(s.d_offsets)); s.d_offsets = nullptr; }
        s.d_meta_cap = 0;
        LZ_HIP(hipMalloc((void**)&s.d_sizes, (ch.nb + 1) * sizeof(u32)));
        LZ_HIP(hipMalloc((void**)&s.d_offsets, (ch.nb + 1) * sizeof(u64)));
        s.d_meta_cap = ch.nb + 1;
    }
    if (s.h_meta_cap < ch.nb + 1) {
        if (s.h_sizes) { LZ_HIP(hipHostFree(s.h_sizes)); s.h_sizes = nullptr; }
        if (s.h_offsets) { LZ_HIP(hipHostFree(s.h_offsets)); s.h_offsets = nullptr; }
        s.h_meta_cap = 0;
        LZ_HIP(hipHostMalloc((void**)&s.h_sizes, (ch.nb + 1) * sizeof(u32), hipHostMallocDefault));
        LZ_HIP(hipHostMalloc((void**)&s.h_offsets, (ch.nb + 1) * sizeof(u64), hipHostMallocDefault));
        s.h_meta_cap = ch.nb + 1;
    }
    const u8* from = j.src + ch.first * j.blockSize;
    if (!srcPinned) {
        if ((rc = ensure_pinned(&s.h_in, &s.h_in_cap, ch.inBytes))) return rc;
        par_memcpy(s.h_in, from, ch.inBytes);
        from = s.h_in;
    }
    // uploads run one after the other (an upload that shares the link with the next chunk's finishes late, and its kernels with it)
    if (prevUp) LZ_HIP(hipStreamWaitEvent(s.stream, prevUp, 0));
    LZ_HIP(hipMemcpyAsync(s.d_in, from, ch.inBytes, hipMemcpyHostToDevice, s.stream));
    LZ_HIP(hipEventRecord(s.up, s.stream));
    if ((rc = launch(c, s.d_in, ch.nb, j.blockSize, last, s.d_slots, slot, s.d_sizes, j.level, s.stream, s.k0, s.k1))) return rc;
    lz_pack_launch(s.d_in, s.d_slots, slot, s.d_sizes, s.d_offsets, s.d_packed, (u32)ch.nb, (u32)j.blockSize, (u32)last, j.mode, s.stream);
    LZ_HIP(hipGetLastError());
    LZ_HIP(hipMemcpyAsync(s.h_sizes, s.d_sizes, ch.nb * sizeof(u32), hipMemcpyDeviceToHost, s.stream));
    LZ_HIP(hipMemcpyAsync(s.h_offsets, s.d_offsets, (ch.nb + 1) * sizeof(u64), hipMemcpyDeviceToHost, s.stream));
    LZ_HIP(hipEventRecord(s.meta, s.stream));
    ch.active = true;
    return 0;
}

int stage_fetch(Stage& s, ChunkState& ch)                       // sizes known -> request exactly the packed bytes
{
    LZ_HIP(hipEventSynchronize(s.meta));
    ch.packedBytes = (size_t)s.h_offsets[ch.nb];
    int rc;
    if ((rc = ensure_pinned(&s.h_out, &s.h_out_cap, ch.packedBytes + 64))) return rc;
    LZ_HIP(hipMemcpyAsync(s.h_out, s.d_packed, ch.packedBytes, hipMemcpyDeviceToHost, s.stream));
    LZ_HIP(hipEventRecord(s.done, s.stream));
    return 0;
}

int run_host_job_inner(Ctx& c, const HostJob& j);
int run_host_job(Ctx& c, const HostJob& j)
{
    const int rc = run_host_job_inner(c, j);
    if (rc) {                                                   // a failed chunk may leave copies of the other stage in flight: drain them
        char keep[sizeof t_err];
        memcpy(keep, t_err, sizeof keep);
        for (Stage& s : c.stage) if (s.stream) (void)hipStreamSynchronize(s.stream);
        (void)hipGetLastError();
        memcpy(t_err, keep, sizeof keep);
    }
    return rc;
}
// Two threads per job.  The calling thread stages and issues chunk after chunk (host -> pinned, H2D, kernels, compaction,
// sizes D2H); a drain thread follows it chunk by chunk: waits for the sizes, requests exactly the packed bytes, waits for them
// and hands them to the sink.  With LZ_STAGES chunks in flight neither side waits for the other's host copies, and the two
// PCIe directions run side by side.
struct Pipe {
    Ctx* c; const HostJob* j;
    size_t nChunks = 0, perChunk = 0;
    bool srcPinned = false;
    ChunkState ch[LZ_STAGES];
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
    size_t issued = 0, drained = 0;            // chunks issued by the caller / handed to the sink by the drain thread
    int err = 0;                               // first error of either side
    char errText[sizeof t_err] = {0};
    float kernelMs = 0.0f;
};
void pipe_fail(Pipe& p, int rc)
{
    pthread_mutex_lock(&p.mu);
    if (!p.err) { p.err = rc; memcpy(p.errText, t_err, sizeof p.errText); }
    pthread_cond_broadcast(&p.cv);
    pthread_mutex_unlock(&p.mu);
}
int drain_chunk(Pipe& p, size_t i)
{
    Stage& s = p.c->stage[i % LZ_STAGES];
    ChunkState& ch = p.ch[i % LZ_STAGES];
    int rc;
    if ((rc = stage_fetch(s, ch))) return rc;
    LZ_HIP(hipEventSynchronize(s.done));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s.k0, s.k1) == hipSuccess) p.kernelMs += ms;
    ch.active = false;
    return p.j->sink(p.j->user, ch.first, ch.nb, s.h_out, ch.packedBytes, s.h_offsets, s.h_sizes);
}
void* drain_thread(void* a)
{
    Pipe& p = *(Pipe*)a;
    t_err[0] = 0;
    if (hipSetDevice(p.c->device) != hipSuccess) { snprintf(t_err, sizeof t_err, "hipSetDevice(%d) failed", p.c->device); pipe_fail(p, -LIZARDGPU_ERR_HIP); return nullptr; }
    for (size_t i = 0; i < p.nChunks; i++) {
        pthread_mutex_lock(&p.mu);
        while (p.issued <= i && !p.err) pthread_cond_wait(&p.cv, &p.mu);
        const bool stop = p.err != 0;
        pthread_mutex_unlock(&p.mu);
        if (stop) return nullptr;
        const int rc = drain_chunk(p, i);
        if (rc) { pipe_fail(p, rc); return nullptr; }
        pthread_mutex_lock(&p.mu);
        p.drained = i + 1;
        pthread_cond_broadcast(&p.cv);
        pthread_mutex_unlock(&p.mu);
    }
    return nullptr;
}
int run_host_job_inner(Ctx& c, const HostJob& j)
{
    int rc = ctx_init(c);
    if (rc) return rc;
    if (!j.src || j.nBlocks == 0 || j.blockSize == 0 || j.lastBlockSize == 0 || j.lastBlockSize > j.blockSize) {
        snprintf(t_err, sizeof t_err, "bad argument (null pointer, zero size or lastBlockSize > blockSize)");
        return -LIZARDGPU_ERR_ARG;
    }
    Pipe p;
    p.c = &c; p.j = &j;
    p.perChunk = chunk_bytes() / j.blockSize;
    if (p.perChunk == 0) p.perChunk = 1;
    p.nChunks = (j.nBlocks + p.perChunk - 1) / p.perChunk;
    p.srcPinned = is_pinned_host(j.src);
    c.hostKernelMs = 0.0f;
    pthread_t th;
    const bool threaded = p.nChunks > 1 && pthread_create(&th, nullptr, drain_thread, &p) == 0;
    for (size_t i = 0; i < p.nChunks; i++) {
        if (threaded) {                                          // the stage of chunk i is free once chunk i - LZ_STAGES is drained
            pthread_mutex_lock(&p.mu);
            while (i >= p.drained + LZ_STAGES && !p.err) pthread_cond_wait(&p.cv, &p.mu);
            const bool stop = p.err != 0;
            pthread_mutex_unlock(&p.mu);
            if (stop) break;
        }
        ChunkState& cur = p.ch[i % LZ_STAGES];
        cur.first = i * p.perChunk;
        cur.nb = j.nBlocks - cur.first < p.perChunk ? j.nBlocks - cur.first : p.perChunk;
        if ((rc = stage_issue(c, c.stage[i % LZ_STAGES], j, cur, p.srcPinned, i ? c.stage[(i - 1) % LZ_STAGES].up : nullptr))) { pipe_fail(p, rc); break; }
        if (threaded) {
            pthread_mutex_lock(&p.mu);
            p.issued = i + 1;
            pthread_cond_broadcast(&p.cv);
            pthread_mutex_unlock(&p.mu);
        } else if ((rc = drain_chunk(p, i))) { pipe_fail(p, rc); break; }
    }
    if (threaded) pthread_join(th, nullptr);
    c.hostKernelMs = p.kernelMs;
    if (p.err) { memcpy(t_err, p.errText, sizeof p.errText); return p.err; }
    return 0;
}

struct SlotSink { u8* dst; size_t dstStride; u32* cSizes; };
int slot_sink(void* user, size_t first, size_t nb, const u8* data, size_t, const u64* offsets, const u32* sizes)
{
    SlotSink* k = (SlotSink*)user;
    for (size_t i = 0; i < nb; i++) {
        memcpy(k->dst + (first + i) * k->dstStride, data + offsets[i], sizes[i]);
        k->cSizes[first + i] = sizes[i];
    }
    return 0;
}

struct PackedSink { u8* dst; size_t cap; size_t used; u64* offsets; u32* cSizes; };
int packed_sink(void* user, size_t first, size_t nb, const u8* data, size_t bytes, const u64* offsets, const u32* sizes)
{
    PackedSink* k = (PackedSink*)user;
    if (k->used + bytes > k->cap) { snprintf(t_err, sizeof t_err, "packed output does not fit in dstCapacity"); return -LIZARDGPU_ERR_ARG; }
    par_memcpy(k->dst + k->used, data, bytes);
    for (size_t i = 0; i < nb; i++) {
        if (k->offsets) k->offsets[first + i] = k->used + offsets[i];
        if (k->cSizes) k->cSizes[first + i] = sizes[i];
    }
    k->used += bytes;
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

void LizardGPU_shutdown(void)
{
    int saved = -1;
    if (hipGetDevice(&saved) != hipSuccess) saved = -1;
    for (int d = 0; d < LZ_MAX_DEVICES; d++) {
        Ctx& c = g_ctx[d];
        pthread_mutex_lock(&c.mu);
        ctx_release(c);
        pthread_mutex_unlock(&c.mu);
    }
    lz_shard_shutdown();
    if (saved >= 0) (void)hipSetDevice(saved);
}

int LizardGPU_compressBlocks_device(const void* d_src, size_t nBlocks, size_t blockSize, size_t lastBlockSize,
                                    void* d_dst, size_t dstStride, uint32_t* d_sizes, int level, void* stream)
{
    Guard g;
    if (g.rc) return g.rc;
    g.c->hostKernelMs = -1.0f;
    return launch(*g.c, d_src, nBlocks, blockSize, lastBlockSize, d_dst, dstStride, d_sizes, level, (hipStream_t)stream);
}

int LizardGPU_compressBlocks_host(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize,
                                  void* dst, size_t dstStride, uint32_t* cSizes, int level)
{
    Guard g;
    if (g.rc) return g.rc;
    if (!dst || !cSizes || blockSize > LIZARD_MAX_INPUT_SIZE || dstStride < (size_t)LIZARD_COMPRESSBOUND((int)blockSize)) {
        snprintf(t_err, sizeof t_err, "bad argument (null pointer or dstStride < Lizard_compressBound(blockSize))");
        return -LIZARDGPU_ERR_ARG;
    }
    SlotSink k = { (u8*)dst, dstStride, cSizes };
    HostJob j = { (const u8*)src, nBlocks, blockSize, lastBlockSize, level, LZ_PACK_PAYLOAD, slot_sink, &k };
    return run_host_job(*g.c, j);
}

int LizardGPU_compressBlocks_host_packed(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize,
                                         void* dst, size_t dstCapacity, uint64_t* offsets, uint32_t* cSizes, int level)
{
    Guard g;
    if (g.rc) return g.rc;
    if (!dst) { snprintf(t_err, sizeof t_err, "bad argument (null dst)"); return -LIZARDGPU_ERR_ARG; }
    PackedSink k = { (u8*)dst, dstCapacity, 0, (u64*)offsets, cSizes };
    HostJob j = { (const u8*)src, nBlocks, blockSize, lastBlockSize, level, LZ_PACK_PAYLOAD, packed_sink, &k };
    int rc = run_host_job(*g.c, j);
    if (!rc && offsets) offsets[nBlocks] = k.used;
    return rc;
}

int LizardGPU_decompressBlocks_device(const void* d_src, size_t srcStride, const uint32_t* d_srcSizes, size_t nBlocks,
                                      void* d_dst, size_t dstStride, uint32_t* d_outSizes, void* stream)
{
    Guard g;
    if (g.rc) return g.rc;
    return launch_decompress(*g.c, d_src, nullptr, srcStride, d_srcSizes, nBlocks, d_dst, dstStride, d_outSizes, (hipStream_t)stream);
}

int LizardGPU_decompressBlocks_host(const void* src, const uint64_t* offsets, size_t nBlocks, void* dst, size_t dstStride, uint32_t* outSizes)
{
    Guard g;
    if (g.rc) return g.rc;
    if (!src || !offsets || !dst || !outSizes || nBlocks == 0 || dstStride == 0) { snprintf(t_err, sizeof t_err, "bad argument"); return -LIZARDGPU_ERR_ARG; }
    Ctx& c = *g.c;
    int rc = ctx_init(c);
    if (rc) return rc;
    Stage& s = c.stage[0];
    // the offsets are input like the blocks themselves: non-decreasing, every block below 4 GiB, the slots addressable
    for (size_t i = 0; i < nBlocks; i++) {
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0xFFFFFFFFull) {
            snprintf(t_err, sizeof t_err, "bad argument (offsets[%zu..%zu] are not a block)", i, i + 1); return -LIZARDGPU_ERR_ARG;
        }
    }
    if (dstStride > (size_t)-1 / nBlocks) { snprintf(t_err, sizeof t_err, "bad argument (nBlocks * dstStride overflows)"); return -LIZARDGPU_ERR_ARG; }
    const size_t inBytes = (size_t)(offsets[nBlocks] - offsets[0]);
    u64* d_off = nullptr;
    if ((rc = ensure_dev(&s.d_in, &s.d_in_cap, inBytes + 64))) return rc;
    if ((rc = ensure_dev(&s.d_slots, &s.d_slots_cap, nBlocks * dstStride))) return rc;
    if ((rc = ensure_dev(&s.d_packed, &s.d_packed_cap, (nBlocks + 1) * sizeof(u64) + nBlocks * sizeof(u32)))) return rc;   // offsets + sizes ride here
    d_off = (u64*)s.d_packed;
    u32* d_out = (u32*)(d_off + nBlocks + 1);
    std::vector<u64> rel(nBlocks + 1);
    for (size_t i = 0; i <= nBlocks; i++) rel[i] = offsets[i] - offsets[0];
    // (`rel` and the caller's buffers are read by copies in flight: every way out of here, also a failing one, drains the stream first)
    rc = [&]() -> int {
        LZ_HIP(hipMemcpyAsync(s.d_in, (const u8*)src + offsets[0], inBytes, hipMemcpyHostToDevice, s.stream));
        LZ_HIP(hipMemcpyAsync(d_off, rel.data(), (nBlocks + 1) * sizeof(u64), hipMemcpyHostToDevice, s.stream));
        int r = launch_decompress(c, s.d_in, d_off, 0, nullptr, nBlocks, s.d_slots, dstStride, d_out, s.stream);
        if (r) return r;
        LZ_HIP(hipMemcpyAsync(outSizes, d_out, nBlocks * sizeof(u32), hipMemcpyDeviceToHost, s.stream));
        LZ_HIP(hipMemcpyAsync(dst, s.d_slots, nBlocks * dstStride, hipMemcpyDeviceToHost, s.stream));
        return 0;
    }();
    if (hipStreamSynchronize(s.stream) != hipSuccess && !rc) { snprintf(t_err, sizeof t_err, "hipStreamSynchronize failed"); rc = -LIZARDGPU_ERR_HIP; }
    return rc;
}

// twin of Lizard_decompress_safe (reference lib/lizard_decompress.h:64 / lizard_decompress.c:267): one block, host buffers
int LizardGPU_decompress_safe(const char* source, char* dest, int compressedSize, int maxDecompressedSize)
{
    if (compressedSize < 0 || maxDecompressedSize < 0 || !source || !dest) return -1;
    if (compressedSize == 0) return 0;                          // reference: inputSize < 1 -> 0
    uint64_t offs[2] = { 0, (uint64_t)compressedSize };
    uint32_t out = 0;
    std::vector<char> tmp((size_t)maxDecompressedSize + 1);
    const int rc = LizardGPU_decompressBlocks_host(source, offs, 1, tmp.data(), (size_t)maxDecompressedSize + 1, &out);
    if (rc || out == 0xFFFFFFFFu || out > (uint32_t)maxDecompressedSize) return -1;
    memcpy(dest, tmp.data(), out);
    return (int)out;
}

// Internal (lizard_frame_host.c): frame block records — LE32 size word (bit 31 = stored raw) + payload — of nBlocks
// independent blocks, packed back to back into dst exactly as LizardF_compressUpdate writes them
// (lizard_frame.c:456-469).  *written receives the byte count.
int lzgpu_frame_records(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize, void* dst, size_t dstCapacity,
                        size_t* written, int level)
{
    Guard g;
    if (g.rc) return g.rc;
    PackedSink k = { (u8*)dst, dstCapacity, 0, nullptr, nullptr };
    HostJob j = { (const u8*)src, nBlocks, blockSize, lastBlockSize, level, LZ_PACK_FRAME, packed_sink, &k };
    int rc = run_host_job(*g.c, j);
    if (written) *written = k.used;
    return rc;
}

// Internal shim for the one-block reference entry points (lizard_host.c): compress one host block,
// honouring the reference's maxDstSize contract: returns the compressed size, 0 if it does not fit
// (reference lib/lizard_compress.c:543-546), < 0 on a GPU failure.
int lzgpu_compress_one(const void* src, int srcSize, void* dst, int maxDstSize, int level)
{
    if (srcSize < 0 || (unsigned)srcSize > (unsigned)LIZARD_MAX_INPUT_SIZE) return 0;
    Guard g;
    if (g.rc) return g.rc;
    Ctx& c = *g.c;
    int rc = ctx_init(c);
    if (rc) return rc;
    if (srcSize == 0) {                 // reference: level byte only (lizard_compress.c:488-494)
        if (maxDstSize < 1) return 0;
        ((u8*)dst)[0] = (u8)clamp_level(level);
        return 1;
    }
    Stage& s = c.stage[0];
    const size_t slot = ((size_t)LIZARD_COMPRESSBOUND(srcSize) + 63) & ~(size_t)63;
    if ((rc = ensure_dev(&s.d_in, &s.d_in_cap, (size_t)srcSize + 64))) return rc;
    if ((rc = ensure_dev(&s.d_slots, &s.d_slots_cap, slot))) return rc;
    if ((rc = ensure_pinned(&s.h_in, &s.h_in_cap, (size_t)srcSize))) return rc;
    if ((rc = ensure_pinned(&s.h_out, &s.h_out_cap, slot + 64))) return rc;
    if (s.d_meta_cap < 2) {
        LZ_HIP(hipMalloc((void**)&s.d_sizes, 2 * sizeof(u32)));
        LZ_HIP(hipMalloc((void**)&s.d_offsets, 2 * sizeof(u64)));
        s.d_meta_cap = 2;
    }
    if (s.h_meta_cap < 2) {
        LZ_HIP(hipHostMalloc((void**)&s.h_sizes, 2 * sizeof(u32), hipHostMallocDefault));
        LZ_HIP(hipHostMalloc((void**)&s.h_offsets, 2 * sizeof(u64), hipHostMallocDefault));
        s.h_meta_cap = 2;
    }
    memcpy(s.h_in, src, (size_t)srcSize);
    LZ_HIP(hipMemcpyAsync(s.d_in, s.h_in, (size_t)srcSize, hipMemcpyHostToDevice, s.stream));
    c.hostKernelMs = -1.0f;
    if ((rc = launch(c, s.d_in, 1, (size_t)srcSize, (size_t)srcSize, s.d_slots, slot, s.d_sizes, level, s.stream))) return rc;
    LZ_HIP(hipMemcpyAsync(s.h_sizes, s.d_sizes, sizeof(u32), hipMemcpyDeviceToHost, s.stream));
    LZ_HIP(hipStreamSynchronize(s.stream));
    const u32 csize = s.h_sizes[0];
    // The reference's room checks compare against oend = dst + maxDstSize (lizard_compress.c:238, :489): whatever fits is
    // written.  A ONE-byte block is the case the reference gets through by accident: Lizard_compress_generic decrements
    // maxOutputSize after the level byte, writeBlock's raw branch tests `*op + blockSize + 4 > oend` only for the sub-block,
    // and with maxDstSize = srcSize - 1 = 0 (the frame layer's call, lizard_frame.c:461) the unsigned room test wraps: the
    // 6-byte block (level, 0x80, LE24 1, the byte) is emitted and its size returned.  Same here.
    if ((int)csize > maxDstSize && !(srcSize == 1 && maxDstSize == 0)) return 0;
    LZ_HIP(hipMemcpyAsync(s.h_out, s.d_slots, csize, hipMemcpyDeviceToHost, s.stream));
    LZ_HIP(hipStreamSynchronize(s.stream));
    memcpy(dst, s.h_out, csize);
    return (int)csize;
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
        if (hipMemcpy(v, c.scratch + (size_t)(w + 1) * LZ_SCRATCH_BYTES - 128, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -LIZARDGPU_ERR_HIP;
        for (int k = 0; k < 15; k++) out[k] += v[k];
        (void)hipMemset(c.scratch + (size_t)(w + 1) * LZ_SCRATCH_BYTES - 128, 0, sizeof v);
    }
    return 0;
}
#endif

float LizardGPU_lastKernelMs(void)
{
    Guard g;
    if (g.rc) return -1.0f;
    Ctx& c = *g.c;
    if (c.hostKernelMs >= 0.0f) return c.hostKernelMs;
    float ms = -1.0f;
    if (c.timed && hipEventSynchronize(c.ev1) == hipSuccess) {
        if (hipEventElapsedTime(&ms, c.ev0, c.ev1) != hipSuccess) ms = -1.0f;
    }
    return ms;
}

}  // extern "C"
