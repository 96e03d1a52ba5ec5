// lizard_gpu.hip — gfx950 kernels + the thin extern "C" shim the host C layer (lizard_host.c) calls.
//
// Launch geometry: ONE wavefront per Lizard API block.  The grid is persistent — one workgroup of up to
// 16 independent waves per CU (they never synchronise with each other) — and every wave pulls block indices
// from a device counter, so tail blocks do not strand CUs.  Each wave owns a hash table (LDS slice or
// global-memory slot, see lz_wave_main) and a scratch slot in a global arena (sequence list / stream staging,
// written and re-read once per sub-block).  Blocks never communicate: no inter-workgroup synchronisation.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lizard_amd.h"
#include "lz_block.h"
#include "lz_datagen.h"

namespace {

struct LzBatch {
    const u8* src;  u64 blockSize;  u32 nBlocks;  u32 lastBlockSize;
    u8* dst;        u64 dstStride;  u32* sizes;   u32 level;
    u8* scratch;    u32* counter;
    u8* tables;     // levels 11/31: one LZ_TABWIDE_BYTES(18) hash table per resident wave (global memory);
                    // hashChain levels: one LZ_HC_SLOT_BYTES(maxBlock) slot per resident wave
    u64 tableStride;
};

// Residency by construction.  LDS is what limits the number of tables in flight, and the hardware hands it out
// in 512-byte granules per WORKGROUP: thirteen independent 64-thread workgroups of 12 560 B each get 12 800 B
// apiece, so only twelve fit in a CU's 160 KiB (the occupancy API, which divides raw sizes, says thirteen).
// Instead ONE workgroup per CU carries W waves and private slices of one allocation; NLDS of them keep their
// hash table in LDS, the others in a global-memory slot (DESIGN.md section 4 lists the split per level).
#define LZ_WAVES_FAST      16
#define LZ_NLDS_FAST       12
#define LZ_WAVES_FAST_HUF  16
#define LZ_NLDS_FAST_HUF   5
#define LZ_WAVES_FASTLDS      13             // all tables in LDS (blocks above 4 MiB)
#define LZ_WAVES_FASTLDS_HUF  9

// NLDS of the W waves keep their hash table in LDS, the others in the wave's global-memory slot (a.tables).
template <int PARSER, int HASHLOG, int AUX, bool HUF, int W, int WSWORDS, int NLDS = (HASHLOG > 14 ? 0 : W)>
__device__ __forceinline__ void lz_wave_main(const LzBatch& a)
{
    struct Slice { u64 ring[PARSER != LZ_PARSER_PRICEFAST ? LZ_SEQ_RING : 1]; u32 ws[WSWORDS]; };
    __shared__ u32 ldsTables[NLDS ? NLDS : 1][NLDS ? LZ_TAB_BYTES(HASHLOG) / 4u : 1];
    __shared__ Slice lds[W];
    // fast parser, mixed residency: the global-table waves need the round tag array of LzTabWide; with the Huffman
    // stage it aliases their workspace, without it they get their own 2 KiB here
    constexpr bool kMixedFast = PARSER == LZ_PARSER_FAST && NLDS != 0 && NLDS != W;
    constexpr bool kOwnTags = kMixedFast && !HUF;
    __shared__ u32 wideTags[kOwnTags ? W - NLDS : 1][kOwnTags ? (1u << LZ_WIDE_TAGLOG) / 4u : 1];
    const u32 wave = lz_uniform(threadIdx.x >> 6);               // readfirstlane: the wave index (and everything derived from it) lives in SGPRs
    Slice& my = lds[wave];
    const u64 slot = (u64)blockIdx.x * W + wave;
    u8* scratch = a.scratch + slot * LZ_SCRATCH_BYTES;
    void* tableMem;
    if constexpr (NLDS == W)      tableMem = (void*)ldsTables[wave];
    else if constexpr (NLDS == 0) tableMem = (void*)(a.tables + slot * a.tableStride);
    else tableMem = wave < (u32)NLDS ? (void*)ldsTables[wave] : (void*)(a.tables + slot * a.tableStride);   // (priceFast: u32 slots there)
    const bool tab32 = NLDS != W && wave >= (u32)NLDS;               // my table is in global memory, u32 slots
    u8* const ws = (kOwnTags && tab32) ? (u8*)wideTags[kOwnTags ? wave - NLDS : 0] : (u8*)my.ws;
    for (;;) {
        lz_converge();
        const u32 b = lz_claim_index(a.counter);
        if (b >= a.nBlocks) break;
        const u32 n = (b == a.nBlocks - 1u) ? a.lastBlockSize : (u32)a.blockSize;
        const u32 c = lz_compress_block<PARSER, HASHLOG, AUX, HUF>(a.src + (u64)b * a.blockSize, n, a.dst + (u64)b * a.dstStride,
                                                                  a.level, tableMem, ws, scratch, my.ring, tab32);
        if (lz_lane() == 0) a.sizes[b] = c;
        lz_converge();
    }
}

// levels 10 / 30: fastSmall parser, 2^12-slot table + sequence ring (+ Huffman workspace).  MIXED: 16 waves, of which
// 12 (5 with the Huffman workspaces) keep the 24-bit-slot table (12 KiB) in LDS and the others a u32-slot table
// in global memory — full 22-bit positions there, hence blocks up to 4 MiB; larger blocks run the all-LDS form
// (13 / 9 waves), whose 17-bit relative positions have no size limit.
template <bool HUF, bool MIXED>
__global__ __launch_bounds__(64 * (MIXED ? (HUF ? LZ_WAVES_FAST_HUF : LZ_WAVES_FAST) : (HUF ? LZ_WAVES_FASTLDS_HUF : LZ_WAVES_FASTLDS)))
void lz_fast12_kernel(LzBatch a)
{
    if constexpr (MIXED)
        lz_wave_main<LZ_PARSER_FAST, 12, 0, HUF, (HUF ? LZ_WAVES_FAST_HUF : LZ_WAVES_FAST), (HUF ? LZ_HUF_WS_WORDS : 1),
                     (HUF ? LZ_NLDS_FAST_HUF : LZ_NLDS_FAST)>(a);
    else
        lz_wave_main<LZ_PARSER_FAST, 12, 0, HUF, (HUF ? LZ_WAVES_FASTLDS_HUF : LZ_WAVES_FASTLDS), (HUF ? LZ_HUF_WS_WORDS : 1),
                     (HUF ? LZ_WAVES_FASTLDS_HUF : LZ_WAVES_FASTLDS)>(a);
}

// levels 11 / 31: fast parser, 2^18-slot table (u32 slots, 1 MiB per wave in global memory: L2 / Infinity Cache)
#define LZ_WAVES_FAST18 16
template <bool HUF>
__global__ __launch_bounds__(64 * LZ_WAVES_FAST18) void lz_fast18_kernel(LzBatch a)
{
    lz_wave_main<LZ_PARSER_FAST, 18, 0, HUF, LZ_WAVES_FAST18, (HUF ? LZ_HUF_WS_WORDS : (1u << LZ_WIDE_TAGLOG) / 4u)>(a);
}

// levels 13-17 / 34-38: hashChain parser (searchLength 5 for rows 13-15, 4 for 16-17; searchNum comes from the
// level at run time).  Per wave: head table + chain array in global memory, 2 KiB tag array / Huffman workspace in LDS.
#define LZ_WAVES_HC 16
template <bool HUF, int SEARCHLEN>
__global__ __launch_bounds__(64 * LZ_WAVES_HC) void lz_hashchain_kernel(LzBatch a)
{
    lz_wave_main<LZ_PARSER_HASHCHAIN, 18, SEARCHLEN, HUF, LZ_WAVES_HC, (HUF ? LZ_HUF_WS_WORDS : (1u << LZ_HC_TAGLOG) / 4u)>(a);
}

// levels 21 / 41: priceFast + LIZv1, 2^14-slot table of 24-bit positions (48 KiB).  Only three such tables fit a
// CU's LDS, and the parse is a latency chain, so the workgroup carries 16 waves anyway: NLDS of them keep the
// table in LDS, the others in their global-memory slot (L2 / Infinity Cache; ~5x slower per wave, but there are
// many more of them).  LDS: 2 x 48 KiB + 16 x 2 KiB tag arrays (level 21), 1 x 48 KiB + 16 x 5.3 KiB Huffman
// workspaces (level 41).
#define LZ_WAVES_PF 16
#define LZ_PF_TAGLOG 11
#define LZ_PF_SLOT_BYTES 65536u
template <bool HUF>
__global__ __launch_bounds__(64 * LZ_WAVES_PF) void lz_pricefast14_kernel(LzBatch a)
{
    lz_wave_main<LZ_PARSER_PRICEFAST, 14, LZ_PF_TAGLOG, HUF, LZ_WAVES_PF, (HUF ? LZ_HUF_WS_WORDS : (1u << LZ_PF_TAGLOG) / 4u), (HUF ? 1 : 2)>(a);
}

// levels 22 / 42: priceFast + LIZv1 with a 2^18-slot table: 1 MiB of u32 slots per wave, all in global memory
template <bool HUF>
__global__ __launch_bounds__(64 * LZ_WAVES_PF) void lz_pricefast18_kernel(LzBatch a)
{
    lz_wave_main<LZ_PARSER_PRICEFAST, 18, LZ_PF_TAGLOG, HUF, LZ_WAVES_PF, (HUF ? LZ_HUF_WS_WORDS : (1u << LZ_PF_TAGLOG) / 4u), 0>(a);
}

// synthetic input: one thread per block, block b = RDG_genBuffer(blockSize, P, seed0 + b)
__global__ __launch_bounds__(64) void lz_datagen_kernel(u8* dst, u64 nBlocks, u64 blockSize, u32 matchProba32,
                                                        int zeroRuns, const u8* lt, u32 seed0)
{
    const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nBlocks) lz_rdg_fill(dst + b * blockSize, (size_t)blockSize, matchProba32, zeroRuns, lt, seed0 + (u32)b);
}

// ------------------------------------------------------------------------------------------------
struct Ctx {
    int   device = -1;
    int   cus = 0;
    int   waves = 0;            // persistent grid size (level 10)
    int   wavesHuf = 0;         // persistent grid size (level 30: larger LDS workspace)
    int   wavesPf = 0, wavesPfHuf = 0;   // levels 21 / 41
    u8*   tables = nullptr;     // levels 11 / 31, allocated on first use
    u8*   pfTables = nullptr;   // levels 21 / 41: 64 KiB per resident wave for the waves whose table is not in LDS
    u8*   hcSlots = nullptr;    // hashChain levels, allocated (and zeroed) on first use / when a larger block size arrives
    size_t hcMaxBlock = 0;
    u8*   scratch = nullptr;
    u32*  counter = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool  timed = false;
    // host-path staging
    u8*   d_src = nullptr;  size_t d_src_cap = 0;
    u8*   d_dst = nullptr;  size_t d_dst_cap = 0;
    u32*  d_sizes = nullptr; size_t d_sizes_cap = 0;
    hipStream_t stream = nullptr;
    char  err[256] = {0};
};

Ctx g_ctx;
int g_want_device = 0;
pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

#define LZ_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            snprintf(g_ctx.err, sizeof g_ctx.err, "%s failed: %s", #call, hipGetErrorString(e_));      \
            return -LIZARDGPU_ERR_HIP;                                                                 \
        }                                                                                              \
    } while (0)

int ctx_init_locked()
{
    if (g_ctx.device == g_want_device && g_ctx.scratch) return 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        snprintf(g_ctx.err, sizeof g_ctx.err, "no HIP device visible");
        return -LIZARDGPU_ERR_NO_DEVICE;
    }
    LZ_HIP(hipSetDevice(g_want_device));
    hipDeviceProp_t prop;
    LZ_HIP(hipGetDeviceProperties(&prop, g_want_device));
    g_ctx.cus = prop.multiProcessorCount;
    g_ctx.waves = g_ctx.cus * LZ_WAVES_FAST18;            // scratch slots for the largest W (one workgroup per CU, see lz_wave_main)
    g_ctx.wavesHuf = g_ctx.cus * LZ_WAVES_FAST_HUF;
    g_ctx.wavesPf = g_ctx.wavesPfHuf = g_ctx.cus * LZ_WAVES_PF;
    LZ_HIP(hipMalloc((void**)&g_ctx.scratch, (size_t)g_ctx.waves * LZ_SCRATCH_BYTES));
#ifdef LZ_PROFILE
    LZ_HIP(hipMemset(g_ctx.scratch, 0, (size_t)g_ctx.waves * LZ_SCRATCH_BYTES));
#endif
    LZ_HIP(hipMalloc((void**)&g_ctx.counter, 64));
    LZ_HIP(hipEventCreate(&g_ctx.ev0));
    LZ_HIP(hipEventCreate(&g_ctx.ev1));
    LZ_HIP(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
    g_ctx.device = g_want_device;
    return 0;
}

int launch_locked(const void* d_src, size_t nBlocks, size_t blockSize, size_t lastBlockSize, void* d_dst,
                  size_t dstStride, u32* d_sizes, int level, hipStream_t stream)
{
    if (!LizardGPU_levelSupported(level)) return -LIZARDGPU_ERR_LEVEL;
    if (!d_src || !d_dst || !d_sizes || nBlocks == 0 || nBlocks > 0xFFFFFFFFu) return -LIZARDGPU_ERR_ARG;
    if (blockSize == 0 || blockSize > LIZARD_MAX_INPUT_SIZE || lastBlockSize == 0 || lastBlockSize > blockSize) return -LIZARDGPU_ERR_ARG;
    if (dstStride < (size_t)LIZARD_COMPRESSBOUND((int)blockSize)) return -LIZARDGPU_ERR_ARG;
    if ((level == 21 || level == 41 || level == 22 || level == 42) && blockSize >= (1u << 24) - 1u) {      // 24-bit table positions (lz_pricefast.h)
        snprintf(g_ctx.err, sizeof g_ctx.err, "levels 21/22/41/42: blocks of 16 MiB or more are not supported on the GPU path");
        return -LIZARDGPU_ERR_ARG;
    }
    int rc = ctx_init_locked();
    if (rc) return rc;
    LzBatch a;
    a.src = (const u8*)d_src; a.blockSize = blockSize; a.nBlocks = (u32)nBlocks; a.lastBlockSize = (u32)lastBlockSize;
    a.dst = (u8*)d_dst; a.dstStride = dstStride; a.sizes = d_sizes; a.level = (u32)level;
    a.scratch = g_ctx.scratch; a.counter = g_ctx.counter; a.tables = g_ctx.tables;
    int lv = level > LIZARD_MAX_CLEVEL ? LIZARD_MAX_CLEVEL : level;
    if (lv < LIZARD_MIN_CLEVEL) lv = LIZARD_DEFAULT_CLEVEL;
    a.level = (u32)lv;
    // one workgroup of W waves per CU; small batches launch only as many workgroups as they have blocks for
    const bool hcLevel = (lv >= 13 && lv <= 17) || (lv >= 34 && lv <= 38);
    const bool fastMixed = blockSize <= (4u << 20);                         // global-table waves hold 22-bit positions
    const u32 W = lv == 10 ? (fastMixed ? LZ_WAVES_FAST : LZ_WAVES_FASTLDS) : lv == 30 ? (fastMixed ? LZ_WAVES_FAST_HUF : LZ_WAVES_FASTLDS_HUF) : (lv == 11 || lv == 31) ? LZ_WAVES_FAST18 : hcLevel ? LZ_WAVES_HC : LZ_WAVES_PF;
    a.tableStride = LZ_TABWIDE_BYTES(18);
    if (hcLevel) {
        if (blockSize > (4u << 20)) {
            snprintf(g_ctx.err, sizeof g_ctx.err, "hashChain levels: blocks above 4 MiB are not supported on the GPU path");
            return -LIZARDGPU_ERR_ARG;
        }
        const size_t cap = (blockSize + 65535u) & ~(size_t)65535u;
        if (!g_ctx.hcSlots || g_ctx.hcMaxBlock < cap) {
            if (g_ctx.hcSlots) { LZ_HIP(hipDeviceSynchronize()); LZ_HIP(hipFree(g_ctx.hcSlots)); g_ctx.hcSlots = nullptr; g_ctx.hcMaxBlock = 0; }
            const size_t bytes = (size_t)g_ctx.cus * LZ_WAVES_HC * LZ_HC_SLOT_BYTES(cap);
            LZ_HIP(hipMalloc((void**)&g_ctx.hcSlots, bytes));
            LZ_HIP(hipMemset(g_ctx.hcSlots, 0, bytes));              // epoch 0 = never used (lz_hc_begin)
            g_ctx.hcMaxBlock = cap;
        }
        a.tables = g_ctx.hcSlots; a.tableStride = LZ_HC_SLOT_BYTES(g_ctx.hcMaxBlock);
    }
    if (lv == 22 || lv == 42) {                                  // same per-wave footprint as levels 11/31
        if (!g_ctx.tables) LZ_HIP(hipMalloc((void**)&g_ctx.tables, (size_t)g_ctx.cus * LZ_WAVES_FAST18 * LZ_TABWIDE_BYTES(18)));
        a.tables = g_ctx.tables;
    }
    if (lv == 11 || lv == 31) {
        if (blockSize > (4u << 20)) {
            snprintf(g_ctx.err, sizeof g_ctx.err, "levels 11/31: blocks above 4 MiB are not supported on the GPU path");
            return -LIZARDGPU_ERR_ARG;
        }
        if (!g_ctx.tables) {
            LZ_HIP(hipMalloc((void**)&g_ctx.tables, (size_t)g_ctx.cus * LZ_WAVES_FAST18 * LZ_TABWIDE_BYTES(18)));
            a.tables = g_ctx.tables;
        }
    }
    if (lv == 10 || lv == 30 || lv == 21 || lv == 41) {
        if (!g_ctx.pfTables) LZ_HIP(hipMalloc((void**)&g_ctx.pfTables, (size_t)g_ctx.cus * LZ_WAVES_PF * LZ_PF_SLOT_BYTES));
        a.tables = g_ctx.pfTables; a.tableStride = LZ_PF_SLOT_BYTES;
    }
    u32 grid = (u32)((nBlocks + W - 1) / W);
    if (grid > (u32)g_ctx.cus) grid = (u32)g_ctx.cus;
    // The scratch arena, the tables and the block counter are shared by all launches of this process: a launch
    // on another stream first waits (on the GPU) for the previous one to finish.
    if (g_ctx.timed) LZ_HIP(hipStreamWaitEvent(stream, g_ctx.ev1, 0));
    LZ_HIP(hipMemsetAsync(g_ctx.counter, 0, 4, stream));
    LZ_HIP(hipEventRecord(g_ctx.ev0, stream));
    switch (lv) {
    case 10: if (fastMixed) hipLaunchKernelGGL((lz_fast12_kernel<false, true>), dim3(grid), dim3(64 * LZ_WAVES_FAST), 0, stream, a);
             else           hipLaunchKernelGGL((lz_fast12_kernel<false, false>), dim3(grid), dim3(64 * LZ_WAVES_FASTLDS), 0, stream, a);
             break;
    case 30: if (fastMixed) hipLaunchKernelGGL((lz_fast12_kernel<true, true>), dim3(grid), dim3(64 * LZ_WAVES_FAST_HUF), 0, stream, a);
             else           hipLaunchKernelGGL((lz_fast12_kernel<true, false>), dim3(grid), dim3(64 * LZ_WAVES_FASTLDS_HUF), 0, stream, a);
             break;
    case 11: hipLaunchKernelGGL(lz_fast18_kernel<false>, dim3(grid), dim3(64 * LZ_WAVES_FAST18), 0, stream, a); break;
    case 31: hipLaunchKernelGGL(lz_fast18_kernel<true>, dim3(grid), dim3(64 * LZ_WAVES_FAST18), 0, stream, a); break;
    case 13: case 14: case 15: hipLaunchKernelGGL((lz_hashchain_kernel<false, 5>), dim3(grid), dim3(64 * LZ_WAVES_HC), 0, stream, a); break;
    case 16: case 17:          hipLaunchKernelGGL((lz_hashchain_kernel<false, 4>), dim3(grid), dim3(64 * LZ_WAVES_HC), 0, stream, a); break;
    case 34: case 35: case 36: hipLaunchKernelGGL((lz_hashchain_kernel<true, 5>), dim3(grid), dim3(64 * LZ_WAVES_HC), 0, stream, a); break;
    case 37: case 38:          hipLaunchKernelGGL((lz_hashchain_kernel<true, 4>), dim3(grid), dim3(64 * LZ_WAVES_HC), 0, stream, a); break;
    case 22: hipLaunchKernelGGL(lz_pricefast18_kernel<false>, dim3(grid), dim3(64 * LZ_WAVES_PF), 0, stream, a); break;
    case 42: hipLaunchKernelGGL(lz_pricefast18_kernel<true>, dim3(grid), dim3(64 * LZ_WAVES_PF), 0, stream, a); break;
    case 21: hipLaunchKernelGGL(lz_pricefast14_kernel<false>, dim3(grid), dim3(64 * LZ_WAVES_PF), 0, stream, a); break;
    default: hipLaunchKernelGGL(lz_pricefast14_kernel<true>, dim3(grid), dim3(64 * LZ_WAVES_PF), 0, stream, a); break;
    }
    LZ_HIP(hipGetLastError());
    LZ_HIP(hipEventRecord(g_ctx.ev1, stream));
    g_ctx.timed = true;
    return 0;
}

template <typename T>
int ensure(T** p, size_t* cap, size_t need)
{
    if (*cap >= need) return 0;
    if (*p) { LZ_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
    LZ_HIP(hipMalloc((void**)p, need));
    *cap = need;
    return 0;
}

}  // namespace

extern "C" {

int LizardGPU_levelSupported(int level)
{
    if (level > LIZARD_MAX_CLEVEL) level = LIZARD_MAX_CLEVEL;        // reference lizard_compress.c:303-308
    if (level < LIZARD_MIN_CLEVEL) level = LIZARD_DEFAULT_CLEVEL;
    return level == 10 || level == 30 || level == 11 || level == 31 || level == 21 || level == 41 || level == 22 || level == 42
        || (level >= 13 && level <= 17) || (level >= 34 && level <= 38);
}

int LizardGPU_setDevice(int device)
{
    pthread_mutex_lock(&g_mu);
    g_want_device = device;
    pthread_mutex_unlock(&g_mu);
    return 0;
}

const char* LizardGPU_lastError(void) { return g_ctx.err; }

int LizardGPU_residentWaves(void)
{
    pthread_mutex_lock(&g_mu);
    int rc = ctx_init_locked();
    int w = rc ? rc : g_ctx.cus * LZ_WAVES_FAST;      // level-10 residency (16 waves per CU: 12 LDS tables + 4 in global memory)
    pthread_mutex_unlock(&g_mu);
    return w;
}

int LizardGPU_compressBlocks_device(const void* d_src, size_t nBlocks, size_t blockSize, size_t lastBlockSize,
                                    void* d_dst, size_t dstStride, uint32_t* d_sizes, int level, void* stream)
{
    pthread_mutex_lock(&g_mu);
    int rc = launch_locked(d_src, nBlocks, blockSize, lastBlockSize, d_dst, dstStride, d_sizes, level, (hipStream_t)stream);
    pthread_mutex_unlock(&g_mu);
    return rc;
}

static int host_locked(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize, void* dst,
                       size_t dstStride, uint32_t* cSizes, int level)
{
    if (!src || !dst || !cSizes || nBlocks == 0 || blockSize == 0 || lastBlockSize == 0 || lastBlockSize > blockSize)
        return -LIZARDGPU_ERR_ARG;
    int rc = ctx_init_locked();
    if (rc) return rc;
    const size_t srcBytes = (nBlocks - 1) * blockSize + lastBlockSize;
    const size_t slot = ((size_t)LIZARD_COMPRESSBOUND((int)blockSize) + 63) & ~(size_t)63;
    if (dstStride < (size_t)LIZARD_COMPRESSBOUND((int)blockSize)) return -LIZARDGPU_ERR_ARG;
    if ((rc = ensure(&g_ctx.d_src, &g_ctx.d_src_cap, srcBytes + 64))) return rc;
    if ((rc = ensure(&g_ctx.d_dst, &g_ctx.d_dst_cap, nBlocks * slot))) return rc;
    if ((rc = ensure(&g_ctx.d_sizes, &g_ctx.d_sizes_cap, nBlocks * sizeof(u32)))) return rc;
    hipStream_t s = g_ctx.stream;
    LZ_HIP(hipMemcpyAsync(g_ctx.d_src, src, srcBytes, hipMemcpyHostToDevice, s));
    rc = launch_locked(g_ctx.d_src, nBlocks, blockSize, lastBlockSize, g_ctx.d_dst, slot, g_ctx.d_sizes, level, s);
    if (rc) return rc;
    LZ_HIP(hipMemcpyAsync(cSizes, g_ctx.d_sizes, nBlocks * sizeof(u32), hipMemcpyDeviceToHost, s));
    LZ_HIP(hipStreamSynchronize(s));
    // payload: only the valid bytes of each slot travel back
    for (size_t i = 0; i < nBlocks; i++)
        LZ_HIP(hipMemcpyAsync((u8*)dst + i * dstStride, g_ctx.d_dst + i * slot, cSizes[i], hipMemcpyDeviceToHost, s));
    LZ_HIP(hipStreamSynchronize(s));
    return 0;
}

int LizardGPU_compressBlocks_host(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize,
                                  void* dst, size_t dstStride, uint32_t* cSizes, int level)
{
    pthread_mutex_lock(&g_mu);
    int rc = host_locked(src, nBlocks, blockSize, lastBlockSize, dst, dstStride, cSizes, level);
    pthread_mutex_unlock(&g_mu);
    return rc;
}

// Internal shim for the one-block reference entry points (lizard_host.c): compress one host block,
// honouring the reference's maxDstSize contract: returns the compressed size, 0 if it does not fit
// (reference lib/lizard_compress.c:543-546), < 0 on a GPU failure.
int lzgpu_compress_one(const void* src, int srcSize, void* dst, int maxDstSize, int level)
{
    if (srcSize < 0 || (unsigned)srcSize > (unsigned)LIZARD_MAX_INPUT_SIZE || maxDstSize < 1) return 0;
    pthread_mutex_lock(&g_mu);
    int rc = ctx_init_locked();
    int result = 0;
    if (!rc && srcSize == 0) {          // reference: level byte only (lizard_compress.c:488-494)
        ((u8*)dst)[0] = (u8)level; result = 1;
    } else if (!rc) {
        const size_t slot = (size_t)LIZARD_COMPRESSBOUND(srcSize);
        u32 csize = 0;
        do {
            if ((rc = ensure(&g_ctx.d_src, &g_ctx.d_src_cap, (size_t)srcSize + 64))) break;
            if ((rc = ensure(&g_ctx.d_dst, &g_ctx.d_dst_cap, slot))) break;
            if ((rc = ensure(&g_ctx.d_sizes, &g_ctx.d_sizes_cap, sizeof(u32)))) break;
            hipStream_t s = g_ctx.stream;
            if (hipMemcpyAsync(g_ctx.d_src, src, (size_t)srcSize, hipMemcpyHostToDevice, s) != hipSuccess) { rc = -LIZARDGPU_ERR_HIP; break; }
            if ((rc = launch_locked(g_ctx.d_src, 1, (size_t)srcSize, (size_t)srcSize, g_ctx.d_dst, slot, g_ctx.d_sizes, level, s))) break;
            if (hipMemcpyAsync(&csize, g_ctx.d_sizes, sizeof(u32), hipMemcpyDeviceToHost, s) != hipSuccess) { rc = -LIZARDGPU_ERR_HIP; break; }
            if (hipStreamSynchronize(s) != hipSuccess) { rc = -LIZARDGPU_ERR_HIP; break; }
            if (csize > (u32)maxDstSize) { result = 0; break; }
            if (hipMemcpy(dst, g_ctx.d_dst, csize, hipMemcpyDeviceToHost) != hipSuccess) { rc = -LIZARDGPU_ERR_HIP; break; }
            result = (int)csize;
        } while (0);
    }
    pthread_mutex_unlock(&g_mu);
    return rc ? rc : result;
}

void LizardGPU_datagen_host(void* buffer, size_t size, double matchProba, double litProba, unsigned seed)
{
    uint8_t lt[LZ_RDG_LTSIZE];
    lz_rdg_table(lt, matchProba, litProba);
    lz_rdg_fill((uint8_t*)buffer, size, (uint32_t)(32768 * matchProba), matchProba >= 1.0, lt, seed);
}

int LizardGPU_datagen_device(void* d_dst, size_t nBlocks, size_t blockSize, double matchProba, double litProba,
                             unsigned seed0, void* stream)
{
    if (!d_dst || nBlocks == 0 || blockSize == 0) return -LIZARDGPU_ERR_ARG;
    pthread_mutex_lock(&g_mu);
    int rc = ctx_init_locked();
    if (!rc) do {
        uint8_t lt[LZ_RDG_LTSIZE];
        lz_rdg_table(lt, matchProba, litProba);
        u8* d_lt = nullptr;
        if (hipMalloc((void**)&d_lt, LZ_RDG_LTSIZE) != hipSuccess) { rc = -LIZARDGPU_ERR_NOMEM; break; }
        hipStream_t s = (hipStream_t)stream;
        if (hipMemcpyAsync(d_lt, lt, LZ_RDG_LTSIZE, hipMemcpyHostToDevice, s) != hipSuccess) { rc = -LIZARDGPU_ERR_HIP; (void)hipFree(d_lt); break; }
        hipLaunchKernelGGL(lz_datagen_kernel, dim3((unsigned)((nBlocks + 63) / 64)), dim3(64), 0, s, (u8*)d_dst, (u64)nBlocks,
                           (u64)blockSize, (u32)(32768 * matchProba), (int)(matchProba >= 1.0), (const u8*)d_lt, (u32)seed0);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = -LIZARDGPU_ERR_HIP;
        (void)hipFree(d_lt);
    } while (0);
    pthread_mutex_unlock(&g_mu);
    return rc;
}

#ifdef LZ_PROFILE
// Profile builds only: sum of the per-wave phase clocks since the last call (scratch slot heads), then reset.
int LizardGPU_profileDump(unsigned long long out[16])
{
    pthread_mutex_lock(&g_mu);
    int rc = ctx_init_locked();
    for (int k = 0; k < 16; k++) out[k] = 0;
    if (!rc) {
        (void)hipDeviceSynchronize();
        for (int w = 0; w < g_ctx.waves; w++) {
            unsigned long long v[16];
            if (hipMemcpy(v, g_ctx.scratch + (size_t)(w + 1) * LZ_SCRATCH_BYTES - 128, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) { rc = -LIZARDGPU_ERR_HIP; break; }
            for (int k = 0; k < 15; k++) out[k] += v[k];
            (void)hipMemset(g_ctx.scratch + (size_t)(w + 1) * LZ_SCRATCH_BYTES - 128, 0, sizeof v);
        }
    }
    pthread_mutex_unlock(&g_mu);
    return rc;
}
#endif

float LizardGPU_lastKernelMs(void)
{
    float ms = -1.0f;
    pthread_mutex_lock(&g_mu);
    if (g_ctx.timed && hipEventSynchronize(g_ctx.ev1) == hipSuccess) {
        if (hipEventElapsedTime(&ms, g_ctx.ev0, g_ctx.ev1) != hipSuccess) ms = -1.0f;
    }
    pthread_mutex_unlock(&g_mu);
    return ms;
}

}  // extern "C"
