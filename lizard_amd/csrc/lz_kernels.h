// lz_kernels.h — the gfx950 kernels of the library and their residency knobs (device side only; the host side that
// launches them is lizard_gpu.hip).
//
// Launch geometry: ONE wavefront per Lizard API block.  The grid is persistent — one workgroup of up to 16 independent
// waves per CU (they never synchronise with each other) — and every wave pulls block indices from a device counter, so
// tail blocks do not strand CUs.  Each wave owns a hash table (LDS slice or global-memory slot, see lz_wave_main) and a
// scratch slot in a global arena (sequence list / Huffman staging, written and re-read once per sub-block).  Blocks never
// communicate: no inter-workgroup synchronisation.
//
// Every lz_*.h of this directory is device code; bench.py hashes them (roofline.traffic_source): a counter pass recorded
// in profiles/pmc_traffic.json speaks for the kernels only as long as these files are the ones it was measured on.
#pragma once
#include "lz_block.h"
#include "lz_split.h"
#include "lz_pack.h"
#include "lz_unpack.h"

namespace {

struct LzBatch {
    const u8* src;  u64 blockSize;  u32 nBlocks;  u32 lastBlockSize;
    u8* dst;        u64 dstStride;  u32* sizes;   u32 level;
    u8* scratch;    u32* counter;
    u8* tables;     // per resident wave, for the waves whose hash table is not in LDS: levels 10/30/21/41 a 64 KiB slot,
                    // levels 11/31/22/42 LZ_TABWIDE_BYTES(18), hashChain levels LZ_HC_SLOT_BYTES(maxBlock)
    u64 tableStride;
    u32 tableSlots;  // number of table slots behind `tables` (hashChain with large blocks: fewer than resident waves; the waves without one leave)
    const u32* srcSizes;  // nullptr: every block is blockSize bytes except the last (lastBlockSize); else block b is srcSizes[b] bytes (<= blockSize,
                          // >= 1): the ragged batches of the one-block entry points' combiner (lizard_pipeline_host.c)
    const u64* srcOffsets; // nullptr: block b starts at src + b * blockSize; else at src + srcOffsets[b] (ragged batches, packed back to back)
    u32 activeWaves; // block-claiming waves per workgroup (<= the kernel's waves / producers).  A launch smaller than the machine spreads
                     // its blocks over all CUs — ceil(nBlocks / grid) claiming waves each — instead of filling a few CUs with 13-16 waves:
                     // a wave with the CU's LDS pipe, L1 and issue slots nearly to itself parses ~1.5x faster
};

// Residency by construction.  LDS is what limits the number of tables in flight, and the hardware hands it out
// in 512-byte granules per WORKGROUP: thirteen independent 64-thread workgroups of 12 560 B each get 12 800 B
// apiece, so only twelve fit in a CU's 160 KiB (the occupancy API, which divides raw sizes, says thirteen).
// Instead ONE workgroup per CU carries W waves and private slices of one allocation; NLDS of them keep their
// hash table in LDS, the others in a global-memory slot (DESIGN.md section 4 lists the split per level).
// The splits are compile-time knobs so that tuning variants can be built side by side (lizard_amd/variants).
// Level 10: all thirteen waves keep their table in LDS.  Three more waves with 16 KiB tables in global memory (12 + 4)
// were 2.5 % faster but doubled the fabric traffic of a launch (156 GB instead of 77 GB for 27 GB of algorithmic bytes):
// every put into a global table leaves L2 as a 32-byte write, and the tables' lines are re-fetched at 128 bytes.
#ifndef LZ_WAVES_FAST
#define LZ_WAVES_FAST      13
#endif
#ifndef LZ_NLDS_FAST
#define LZ_NLDS_FAST       13
#endif
#ifndef LZ_WAVES_FAST_HUF
#define LZ_WAVES_FAST_HUF  16
#endif
#ifndef LZ_NLDS_FAST_HUF
#define LZ_NLDS_FAST_HUF   11
#endif
#ifndef LZ_HC_POOL
#define LZ_HC_POOL         3              // chain-build regions (32.3 KiB each) shared by the waves of a hashChain workgroup (one more without Huffman)
#endif
#ifndef LZ_HUF_POOL
#define LZ_HUF_POOL        5              // Huffman workspaces shared by the 16 waves of a level-30 workgroup (0 = one each)
#endif
#define LZ_WAVES_FASTLDS      13             // all tables in LDS (blocks above 4 MiB)
#define LZ_WAVES_FASTLDS_HUF  11
#ifndef LZ_WIDE_OCC
#define LZ_WIDE_OCC 1                        // occupancy summaries of the 2^18-slot global tables (levels 11/31, 22/42)
#endif
#ifndef LZ_SPREAD_SMALL
#define LZ_SPREAD_SMALL 1                    // launches smaller than the machine: one claiming wave per CU before a second one anywhere (lizard_gpu.hip launch())
#endif
#ifndef LZ_MAX_WAVES
#define LZ_MAX_WAVES          16             // scratch / table slots per CU
#endif

// NLDS of the W waves keep their hash table in LDS (form LDSKIND), the others in the wave's global-memory slot.
template <int PARSER, int HASHLOG, int AUX, bool HUF, int W, int WSWORDS, int NLDS = (HASHLOG > 14 ? 0 : W), u32 LDSKIND = LZ_TABKIND_LDS, int POOL = 0, int OCCLOG = 0, int WIDETAGLOG = LZ_WIDE_TAGLOG>
__device__ __forceinline__ void lz_wave_main(const LzBatch& a)
{
    struct Slice { u64 ring[LZ_SEQ_RING]; u32 ws[WSWORDS]; };
    constexpr u32 kTabWords = (LDSKIND == LZ_TABKIND_LDS18 ? LZ_TAB24C_BYTES(HASHLOG) : PARSER == LZ_PARSER_PRICEFAST ? (4u << HASHLOG) : LZ_TAB_BYTES(HASHLOG)) / 4u + 1u;
    __shared__ u32 ldsTables[NLDS ? NLDS : 1][NLDS ? kTabWords : 1];
    __shared__ Slice lds[W];
    // mixed residency: only the global-table waves need a round tag array (LzTabWide / LzTab32; the LDS-table waves find
    // same-slot lanes through the table itself); with the Huffman stage it aliases their workspace, without it they get
    // their own here
    constexpr bool kMixed = PARSER != LZ_PARSER_HASHCHAIN && NLDS != 0 && NLDS != W;
    constexpr bool kOwnTags = kMixed && (!HUF || POOL != 0);
    // POOL != 0: the waves borrow their Huffman workspace from a pool of POOL slots (lz_pool_acquire) instead of owning one
    __shared__ u32 hufPool[POOL ? POOL : 1][POOL ? LZ_HUF_WS_WORDS : 1];
    __shared__ u32 hufPoolMask;
    // hashChain: the chain build of a block borrows one of HCPOOL 32 KiB regions (lz_hc_build)
    constexpr int HCPOOL = PARSER == LZ_PARSER_HASHCHAIN ? ((HUF && !POOL) ? LZ_HC_POOL : LZ_HC_POOL + 1) : 0;   // sixteen Huffman workspaces take a region's worth of LDS (a pool of them does not)
    // levels 11 / 31: occupancy summary of the wave's 2^18-slot table (LzTabWide::occ), 2^OCCLOG bits + a spare word
    constexpr u32 kOccWords = OCCLOG ? ((1u << OCCLOG) >> 5) + 1u : 1u;
    __shared__ u32 wideOcc[OCCLOG ? W : 1][kOccWords];
    __shared__ u32 hcPoolMask;                                   // (a mask of its own: the hashChain levels with the Huffman stage use both pools)
    __shared__ u32 hcPoolMem[HCPOOL ? HCPOOL : 1][HCPOOL ? LZ_HC_REGION_WORDS : 1];
    if constexpr (POOL != 0 || HCPOOL != 0) { if (threadIdx.x == 0) { hufPoolMask = 0; hcPoolMask = 0; } __syncthreads(); }
    LzHufPool hcPool; hcPool.base = &hcPoolMem[0][0]; hcPool.mask = &hcPoolMask; hcPool.count = (u32)HCPOOL; hcPool.stride = LZ_HC_REGION_WORDS;
    constexpr u32 kTagWords = (PARSER == LZ_PARSER_FAST ? (1u << LZ_WIDE_TAGLOG) : (1u << AUX)) / 4u;
    __shared__ u32 wideTags[kOwnTags ? W - NLDS : 1][kOwnTags ? kTagWords : 1];
    const u32 wave = lz_uniform(threadIdx.x >> 6);               // readfirstlane: the wave index (and everything derived from it) lives in SGPRs
    Slice& my = lds[wave];
    const u64 slot = (u64)blockIdx.x * LZ_MAX_WAVES + wave;
    u8* scratch = a.scratch + slot * LZ_SCRATCH_BYTES;
    // table slots are dealt out wave-index-major, so that a launch with fewer slots than waves keeps every CU busy
    const u64 tslot = (u64)wave * gridDim.x + blockIdx.x;
    if (tslot >= a.tableSlots) return;
    if (wave >= a.activeWaves) return;
    void* tableMem;
    if constexpr (NLDS == W)      tableMem = (void*)ldsTables[wave];
    else if constexpr (NLDS == 0) tableMem = (void*)(a.tables + tslot * a.tableStride);
    else tableMem = wave < (u32)NLDS ? (void*)ldsTables[wave] : (void*)(a.tables + tslot * a.tableStride);
    const u32 tabKind = (NLDS != W && wave >= (u32)NLDS) ? LZ_TABKIND_GLOBAL : LDSKIND;
    u8* const ws = (kOwnTags && tabKind == LZ_TABKIND_GLOBAL) ? (u8*)wideTags[kOwnTags ? wave - NLDS : 0] : (u8*)my.ws;
#ifdef LZ_LDS_PRIO
    if (tabKind != LZ_TABKIND_GLOBAL) __builtin_amdgcn_s_setprio(LZ_LDS_PRIO);   // the LDS-table waves are the fast ones: they issue first
#endif
    for (;;) {
        lz_converge();
        const u32 b = lz_claim_index(a.counter);
        if (b >= a.nBlocks) break;
        const u32 n = a.srcSizes ? lz_uniform(a.srcSizes[b]) : (b == a.nBlocks - 1u) ? a.lastBlockSize : (u32)a.blockSize;
        const u8* const srcb = a.src + (a.srcOffsets ? lz_uniform64(a.srcOffsets[b]) : (u64)b * a.blockSize);
        const u32 c = lz_compress_block<PARSER, HASHLOG, AUX, HUF>(srcb, n, a.dst + (u64)b * a.dstStride,
                                                                  a.level, tableMem, ws, scratch, my.ring, tabKind,
                                                                  POOL ? &hufPool[0][0] : nullptr, POOL ? &hufPoolMask : nullptr, (u32)POOL,
                                                                  &hcPool, (u32)a.blockSize, OCCLOG ? wideOcc[OCCLOG ? wave : 0] : nullptr, (u32)OCCLOG, (u32)WIDETAGLOG);
        if (lz_lane() == 0) a.sizes[b] = c;
        lz_converge();
    }
}

#ifndef LZ_FAST12_SPLIT
#define LZ_FAST12_SPLIT 1
#endif
#if !LZ_FAST12_SPLIT
#include "variants/lz_fast12_onewave.h"     // the one-wave-per-block form of rounds 1-2 (A/B builds only)
#endif

// levels 10 / 30, producer / consumer form (lz_split.h): NP waves with a 12 KiB table in LDS only parse, NC waves without a table
// run the container (encode pass, huff0) of the sub-blocks the producers hand over.  Every block size: the 17-bit relative
// positions of LzTab have no size limit.  LZ_FAST12_SPLIT=0 builds the one-wave-per-block form above instead (tuning variants).
#ifndef LZ_SPLIT_PROD
#define LZ_SPLIT_PROD 13
#endif
#ifndef LZ_SPLIT_CONS
#define LZ_SPLIT_CONS 3
#endif
#ifndef LZ_SPLIT_PROD_HUF
#define LZ_SPLIT_PROD_HUF 12                                 // round 6: the container got 35 % cheaper (lz_huf.h, lz_copy_literal_runs), four consumers keep up with twelve tables
#endif
#ifndef LZ_SPLIT_CONS_HUF
#define LZ_SPLIT_CONS_HUF 4
#endif
#ifndef LZ_SPLIT_BUFS
#define LZ_SPLIT_BUFS 2                                      // sequence buffers per producer (level 10: LDS has room for 32 mailbox words per consumer)
#endif
#ifndef LZ_SPLIT_BUFS_HUF
#define LZ_SPLIT_BUFS_HUF 2
#endif
template <bool HUF>
__global__ __launch_bounds__(64 * (HUF ? LZ_SPLIT_PROD_HUF + LZ_SPLIT_CONS_HUF : LZ_SPLIT_PROD + LZ_SPLIT_CONS)) void lz_fast12_split_kernel(LzBatch a)
{
    constexpr u32 NP = HUF ? LZ_SPLIT_PROD_HUF : LZ_SPLIT_PROD, NC = HUF ? LZ_SPLIT_CONS_HUF : LZ_SPLIT_CONS;
    constexpr u32 NB = HUF ? LZ_SPLIT_BUFS_HUF : LZ_SPLIT_BUFS, QN = NP * NB <= 32u ? 32u : 64u;
    static_assert(NP * NB <= QN, "a mailbox holds every buffer of every producer");
    static_assert(LZ_SPLIT_ARENA_BYTES(NP, NC, NB) <= (size_t)LZ_MAX_WAVES * LZ_SCRATCH_BYTES, "the workgroup's scratch arena");
    static_assert(NP * 4u <= LZ_SPLIT_OPS_BYTES, "one position word per producer");
    constexpr u32 kTabWords = LZ_TAB_BYTES(12) / 4u + 1u;
    __shared__ u32 tables[NP][kTabWords];
    __shared__ u64 rings[NP][LZ_SEQ_RING];
    __shared__ u32 hufWs[HUF ? NC : 1][HUF ? LZ_HUF_WS_WORDS : 1];
    __shared__ u32 shared[LZ_SPLIT_SHARED_WORDS(NP, NC, QN)];
    const u32 wave = lz_uniform(threadIdx.x >> 6);
    const LzSplitShared sh = lz_split_shared(shared, NP, NC);
    if (wave == 0) lz_split_shared_init(sh, NP, NC, NB, QN);
    __syncthreads();
    LzSplitArgs s;
    s.src = a.src; s.blockSize = a.blockSize; s.nBlocks = a.nBlocks; s.lastBlockSize = a.lastBlockSize;
    s.dst = a.dst; s.dstStride = a.dstStride; s.sizes = a.sizes; s.level = a.level; s.counter = a.counter;
    s.arena = a.scratch + (u64)blockIdx.x * LZ_MAX_WAVES * LZ_SCRATCH_BYTES; s.nProd = NP; s.nCons = NC; s.nBufs = NB; s.qn = QN;
    s.srcSizes = a.srcSizes; s.srcOffsets = a.srcOffsets; s.activeProd = a.activeWaves < NP ? a.activeWaves : NP;
    if (wave < NP) lz_split_producer<12>(s, sh, wave, (void*)tables[wave < NP ? wave : 0], rings[wave < NP ? wave : 0]);
    else           lz_split_consumer<HUF>(s, sh, wave - NP, hufWs[HUF ? wave - NP : 0]);
}

// levels 11 / 31: fast parser, 2^18-slot table (u32 slots, 1 MiB per wave in global memory: L2 / Infinity Cache)
#define LZ_WAVES_FAST18 16
#ifndef LZ_WIDE_HUF_POOL
#define LZ_WIDE_HUF_POOL 4                   // levels 31 / 42: Huffman workspaces from a pool of this many, which leaves room for the 8 KiB summary of levels 11 / 22 (1 KiB tags); 0 = one each, 4 KiB summary
#endif
template <bool HUF>
__global__ __launch_bounds__(64 * LZ_WAVES_FAST18) void lz_fast18_kernel(LzBatch a)
{
    // tag array: 1 KiB; occupancy summary: 8 KiB (4 slots per bit); level 31 borrows its Huffman workspace from a pool (round 6: with one
    // workspace per wave doubling as a 2 KiB tag array the summary had 4 KiB: 40.9 -> 42.2 GB/s, level 42 32.9 -> 34.1, profiles/r06zq_*)
    constexpr bool kOwnWs = HUF && !LZ_WIDE_HUF_POOL;
    lz_wave_main<LZ_PARSER_FAST, 18, 0, HUF, LZ_WAVES_FAST18, (kOwnWs ? LZ_HUF_WS_WORDS : (1u << 10) / 4u), 0, LZ_TABKIND_LDS, (HUF ? LZ_WIDE_HUF_POOL : 0),
                 (LZ_WIDE_OCC ? (kOwnWs ? 15 : 16) : 0), (kOwnWs ? LZ_WIDE_TAGLOG : 10)>(a);
}

// levels 13-17 / 34-38: hashChain parser (searchLength 5 for rows 13-15, 4 for 16-17; searchNum comes from the
// level at run time).  Per wave: bins + chain array in global memory, Huffman workspace in LDS; the chain build of a
// block borrows one of LZ_HC_POOL 32 KiB LDS regions of the workgroup.
// levels 12 / 33 (noChain, hashLog 18) are <*, 6, 18> — hash5, one candidate per search, searches of one memory trip — and level 32
// (hashLog 14) is <true, 6, 14>; levels 13 / 14 / 15 and twins (searchNum 2 / 4 / 8) are <*, 7 / 8 / 9, 18>: the candidates measured side
// by side in one trip.
#ifndef LZ_WAVES_HC
#define LZ_WAVES_HC 16
#endif
#ifndef LZ_HC_HUF_POOL
#define LZ_HC_HUF_POOL 0                     // levels 32-38: Huffman workspaces from a pool of this many — and a fourth chain-build region; 0 = one workspace each, three regions
#endif
template <bool HUF, int SEARCHLEN, int HLOG = 18>
__global__ __launch_bounds__(64 * LZ_WAVES_HC) void lz_hashchain_kernel(LzBatch a)
{
    lz_wave_main<LZ_PARSER_HASHCHAIN, HLOG, SEARCHLEN, HUF, LZ_WAVES_HC, ((HUF && !LZ_HC_HUF_POOL) ? LZ_HUF_WS_WORDS : 1), 0, LZ_TABKIND_LDS,
                 (HUF ? LZ_HC_HUF_POOL : 0)>(a);
}

// levels 20 / 40: fastBig + LIZv1 (lz_fastbig.h), 2^14 u32 slots per wave in its global-memory slot (the slots of levels 21 / 41);
// in LDS per wave a 1 KiB tag array for the rounds and 8 KiB of slot codes (4 bits per slot: which slots are worth reading); the
// Huffman workspaces of level 40 come from a pool of four
#ifndef LZ_WAVES_FASTBIG
#define LZ_WAVES_FASTBIG 16
#endif
#ifndef LZ_FASTBIG_CODES
#define LZ_FASTBIG_CODES 1                       // 0: every probe reads its slot (tuning variants; the table is cleared per block then)
#endif
#define LZ_FASTBIG_HUF_POOL 4
template <bool HUF>
__global__ __launch_bounds__(64 * LZ_WAVES_FASTBIG) void lz_fastbig14_kernel(LzBatch a)
{
    lz_wave_main<LZ_PARSER_FASTBIG, 14, 10, HUF, LZ_WAVES_FASTBIG, (1u << 10) / 4u, 0, LZ_TABKIND_LDS, (HUF ? LZ_FASTBIG_HUF_POOL : 0),
                 (LZ_FASTBIG_CODES ? 16 : 0)>(a);
}

// levels 21 / 41: priceFast + LIZv1, 2^14-slot table.  A wave whose table is in LDS is bound by the issue latency of its own
// instruction chain (one exposed memory trip per sequence, lz_pricefast.h); the waves beyond the LDS's three tables keep the
// table in a global-memory slot (every probe a 128-byte line) and fill the issue slots the LDS waves leave.  Two forms by block size:
//   SMALL (blocks <= 256 KiB, the benchmark configuration): 18-bit position + 6 check bits, 48 KiB (LzTab24c) — three tables per CU;
//   general (any size): u32 slots, position mod 2^24 + 8 check bits, 64 KiB (LzTab32L) — two tables per CU.
// The remaining waves of the workgroup keep the same u32 slots in their 64 KiB global-memory slot (LzTab32G).
#ifndef LZ_PF_W
#define LZ_PF_W 12
#endif
#define LZ_PF22_W 16
#ifndef LZ_PF_NLDS
#define LZ_PF_NLDS 2
#endif
#ifndef LZ_PF_NLDS_HUF
#define LZ_PF_NLDS_HUF 2
#endif
#ifndef LZ_PF_TAGLOG
#define LZ_PF_TAGLOG 11
#endif
#ifndef LZ_PF18_W
#define LZ_PF18_W 11
#endif
#ifndef LZ_PF18_NLDS
#define LZ_PF18_NLDS 3
#endif
#ifndef LZ_PF18_TAGLOG
#define LZ_PF18_TAGLOG 10
#endif
#ifndef LZ_PF18_HUF_POOL
#define LZ_PF18_HUF_POOL 2                 // Huffman workspaces shared by the waves of a level-41 workgroup (0 = one each)
#endif
#ifndef LZ_PF18_W_HUF
#define LZ_PF18_W_HUF 11
#endif
#ifndef LZ_PF18_NLDS_HUF
#define LZ_PF18_NLDS_HUF (LZ_PF18_HUF_POOL ? 3 : 2)
#endif
#define LZ_PF_SLOT_BYTES 65536u
template <bool HUF, bool SMALL>
__global__ __launch_bounds__(64 * (SMALL ? (HUF ? LZ_PF18_W_HUF : LZ_PF18_W) : LZ_PF_W)) void lz_pricefast14_kernel(LzBatch a)
{
    if constexpr (SMALL)
        // level 41: the Huffman workspaces come from a pool (the parse is most of a wave's time), which leaves room for as many
        // LDS tables as level 21 has
        lz_wave_main<LZ_PARSER_PRICEFAST, 14, (HUF ? (LZ_PF18_HUF_POOL ? 10 : 11) : LZ_PF18_TAGLOG), HUF, (HUF ? LZ_PF18_W_HUF : LZ_PF18_W),
                     (HUF && !LZ_PF18_HUF_POOL ? LZ_HUF_WS_WORDS : 1), (HUF ? LZ_PF18_NLDS_HUF : LZ_PF18_NLDS), LZ_TABKIND_LDS18,
                     (HUF ? LZ_PF18_HUF_POOL : 0)>(a);
    else
        lz_wave_main<LZ_PARSER_PRICEFAST, 14, LZ_PF_TAGLOG, HUF, LZ_PF_W, (HUF ? LZ_HUF_WS_WORDS : 1),
                     (HUF ? LZ_PF_NLDS_HUF : LZ_PF_NLDS)>(a);
}

// levels 22 / 42: priceFast + LIZv1 with a 2^18-slot table: 1 MiB of u32 slots per wave, all in global memory
template <bool HUF>
__global__ __launch_bounds__(64 * LZ_PF22_W) void lz_pricefast18_kernel(LzBatch a)
{
    // tag array 1 KiB (with Huffman: the 2 KiB workspace doubles as it); occupancy summary of the table 8 KiB (4 KiB with Huffman)
    constexpr bool kOwnWs = HUF && !LZ_WIDE_HUF_POOL;
    lz_wave_main<LZ_PARSER_PRICEFAST, 18, (kOwnWs ? LZ_PF_TAGLOG : 10), HUF, LZ_PF22_W, (kOwnWs ? LZ_HUF_WS_WORDS : (1u << 10) / 4u), 0, LZ_TABKIND_LDS,
                 (HUF ? LZ_WIDE_HUF_POOL : 0), (LZ_WIDE_OCC ? (kOwnWs ? 15 : 16) : 0)>(a);
}

// Decompression: one wave per block, same persistent grid; block b is read from
// src + offsets[b] (packed form) or src + b * srcStride (slot form), srcSizes[b] bytes.
struct LzUnBatch {
    const u8* src; const u64* offsets; u64 srcStride; const u32* srcSizes;
    u8* dst; u64 dstStride; u32* outSizes; u32 nBlocks;
    u8* scratch; u32* counter;
};
#define LZ_WAVES_DEC 16
__global__ __launch_bounds__(64 * LZ_WAVES_DEC) void lz_decompress_kernel(LzUnBatch a)
{
    __shared__ u32 ws[LZ_WAVES_DEC][LZD_WS_WORDS];
    const u32 wave = lz_uniform(threadIdx.x >> 6);
    u8* stage = a.scratch + ((u64)blockIdx.x * LZ_MAX_WAVES + wave) * LZ_SCRATCH_BYTES;     // 4 x LZD_STAGE_BYTES fit a scratch slot
    for (;;) {
        lz_converge();
        const u32 b = lz_claim_index(a.counter);
        if (b >= a.nBlocks) break;
        const u8* in = a.offsets ? a.src + a.offsets[b] : a.src + (u64)b * a.srcStride;
        const u32 n = a.offsets ? (u32)(a.offsets[b + 1] - a.offsets[b]) : a.srcSizes[b];
        const u32 cap = a.dstStride > 0x7E000000ull ? 0x7E000000u : (u32)a.dstStride;
        const u32 r = lz_decompress_block(in, n, a.dst + (u64)b * a.dstStride, cap, stage, ws[wave]);
        if (lz_lane() == 0) a.outSizes[b] = r;
        lz_converge();
    }
}

// Self-check of the one undocumented hardware property the kernels rely on (lz_wave.h: lz_lds_mskor_rtn2, lz_lds_xchg_rtn,
// lz_lds_add_rtn): the lanes of ONE DS atomic instruction that hit the same LDS dword are served in ascending lane order.
// Run once per device when its context is created (ctx_init); levels whose kernels depend on it are refused if it fails.
// Per trial: pseudo-random slot per lane (spans 1 / 4 / 64 / 256 slots: everything from "all lanes on one dword" to few
// collisions), a returning exchange, a returning masked-or on 16-bit fields and a returning add; a lane's result must be what
// the closest lower lane on its slot left behind (or the initial value).  *bad counts the lanes that saw something else.
__global__ __launch_bounds__(64) void lz_selfcheck_lane_order_kernel(u32* bad, u32 trials)
{
    __shared__ u32 s[256];
    const u32 lane = threadIdx.x;
    u32 nbad = 0, rng = 0x9E3779B9u * (blockIdx.x * 64u + lane + 1u);
    for (u32 t = 0; t < trials; t++) {
        rng = rng * 1664525u + 1013904223u;
        const u32 span = (t & 3u) == 0 ? 1u : (t & 3u) == 1 ? 4u : (t & 3u) == 2 ? 64u : 256u;
        const u32 a = (rng >> 16) % span;
        u64 same = 0;                                                       // lanes below me on my slot
        for (u32 l = 0; l < 64u; l++) if ((u32)__shfl((int)a, (int)l) == a && l < lane) same |= 1ull << l;
        const u32 prevLane = same ? 63u - (u32)__builtin_clzll(same) : 64u;
        // exchange
        for (u32 i = lane; i < 256u; i += 64u) s[i] = 0xABCD0000u + i;
        __syncthreads();
        const u32 x = lz_lds_xchg_rtn(&s[a], 0x1000u + lane + t * 64u);
        if (x != (prevLane < 64u ? 0x1000u + prevLane + t * 64u : 0xABCD0000u + a)) nbad++;
        __syncthreads();
        // masked-or on 16-bit fields, two per dword (LzTab::xchg)
        for (u32 i = lane; i < 256u; i += 64u) s[i] = 0;
        __syncthreads();
        const u32 sh = (a & 1u) * 16u;
        u32 o1, o2;
        lz_lds_mskor_rtn2((LZ_LDS u32*)&s[a >> 1], 0xFFFFu << sh, ((lane + 1u + t) & 0xFFFFu) << sh, (LZ_LDS u32*)&s[128u + (a >> 1)], 0xFFFFu << sh, ((lane + 7u + t) & 0xFFFFu) << sh, o1, o2);
        if (((o1 >> sh) & 0xFFFFu) != (prevLane < 64u ? ((prevLane + 1u + t) & 0xFFFFu) : 0u)) nbad++;
        if (((o2 >> sh) & 0xFFFFu) != (prevLane < 64u ? ((prevLane + 7u + t) & 0xFFFFu) : 0u)) nbad++;
        __syncthreads();
        // add: consecutive tickets in lane order (lz_hc_build's stable partition)
        for (u32 i = lane; i < 256u; i += 64u) s[i] = 1000u * i;
        __syncthreads();
        const u32 y = lz_lds_add_rtn(&s[a], 1u);
        if (y != 1000u * a + lz_popc64(same)) nbad++;
        __syncthreads();
    }
    if (nbad) atomicAdd(bad, nbad);
}

}  // namespace
