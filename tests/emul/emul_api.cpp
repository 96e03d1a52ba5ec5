// tests/emul/emul_api.cpp — TEST INFRASTRUCTURE ONLY: C entry points that run the product's kernel
// bodies (lizard_amd/csrc/lz_block.h, ...) on the CPU SIMT emulator. Loaded by tests via ctypes.
#include "lz_wave.h"            // tests/emul/lz_wave.h (emulator) — must come first, see the shared guard
#include "../../lizard_amd/csrc/lz_block.h"
#include "../../lizard_amd/csrc/lz_unpack.h"
#include "../../lizard_amd/csrc/lz_split.h"
#include <thread>
#include <vector>

namespace {
struct Args { const u8* src; u32 n; u8* dst; u32 level; u32* table; u8* tag; u8* scratch; u64* ring; u32 result; u32 tabKind; u32* hcRegion; u32 maxBlock; u32 poolMask; u32* wideOcc; };

template <int PARSER, int HASHLOG, int AUX, bool HUF>
void entry_block(void* a)
{
    Args* x = (Args*)a;
    LzHufPool hcPool; hcPool.base = x->hcRegion; hcPool.mask = &x->poolMask; hcPool.count = 1; hcPool.stride = LZ_HC_REGION_WORDS;   // a pool of one chain-build region
    u32 r = lz_compress_block<PARSER, HASHLOG, AUX, HUF>(x->src, x->n, x->dst, x->level, x->table, x->tag, x->scratch, x->ring, x->tabKind,
                                                         nullptr, nullptr, 0, &hcPool, x->maxBlock, x->wideOcc, x->wideOcc ? 16u : 0u);
    if (lz_lane() == 0) x->result = r;
}
}  // namespace



unsigned long long lzemu_stats[64];
// event counters of the kernels' LZ_STAT marks since the last reset
extern "C" void emul_stats(unsigned long long* out, int reset)
{
    for (int i = 0; i < 64; i++) { out[i] = __atomic_load_n(&lzemu_stats[i], __ATOMIC_RELAXED); if (reset) __atomic_store_n(&lzemu_stats[i], 0ull, __ATOMIC_RELAXED); }
}

// hashChain levels keep one persistent global slot that is never cleared (as the host library does): whatever an earlier
// block left in it (bins, links, saved head tables) must not matter.
static u8* g_hcSlot = nullptr;
static const size_t kHcMaxBlock = 18u << 20;
static u8* hc_slot()
{
    if (!g_hcSlot) { g_hcSlot = (u8*)malloc(LZ_HC_SLOT_BYTES(kHcMaxBlock)); memset(g_hcSlot, 0xB7, LZ_HC_SLOT_BYTES(kHcMaxBlock)); }
    return g_hcSlot;
}

// Compress one block with the emulated wave. dst must hold Lizard_compressBound(n) bytes.
// `seed` drives the lane scheduling order (any value must give identical output).
extern "C" int emul_compress_block(const void* src, int n, void* dst, int level, unsigned seed)
{
    Args a;
    int base = level >= 30 ? level - 20 : level;
    if (level >= 34 && level <= 38) base = level - 21;
    const bool ncLevel = level == 12 || level == 32 || level == 33;       // noChain: the hashChain kernels with one candidate per search
    const bool hcLevel = (base >= 13 && base <= 17) || ncLevel;
    int hashLog = ncLevel ? (level == 32 ? 14 : 18) : base == 10 ? 12 : base == 11 ? 18 : base == 20 ? 14 : base == 21 ? 14 : base == 22 ? 18 : hcLevel ? 18 : 0;
    if (!hashLog) return -1;
    if (hcLevel && (size_t)n > kHcMaxBlock) return -1;
    a.src = (const u8*)src; a.n = (u32)n; a.dst = (u8*)dst; a.level = (u32)level; a.result = 0;
    // odd seeds run the u32-slot (global-memory) table layout of the mixed-residency kernels; seeds = 2 mod 4 the packed
    // 18 + 6 bit LDS table of the priceFast kernel for blocks up to 256 KiB; the others the u32-slot LDS form
    a.tabKind = (base == 20 || ((base == 21 || base == 22 || base == 10) && (seed & 1u))) ? LZ_TABKIND_GLOBAL
              : (base == 21 && n <= (1 << 18) && (seed & 3u) == 2u) ? LZ_TABKIND_LDS18 : LZ_TABKIND_LDS;
    a.table = (u32*)aligned_alloc(64, (sizeof(u32) << hashLog) + 64);
    a.tag = (u8*)malloc(8192);
    a.scratch = (u8*)malloc(LZ_SCRATCH_BYTES);
    u64 ring[LZ_SEQ_RING]; memset(ring, 0xEE, sizeof ring); a.ring = ring;
    memset(a.table, 0xA5, (sizeof(u32) << hashLog) + 64);   // garbage: the kernel must initialise its state
    memset(a.tag, 0x5A, 8192);
    memset(a.scratch, 0xCC, LZ_SCRATCH_BYTES);
    static_assert(4 * LZ_HUF_WS_WORDS <= 8192, "emulated LDS workspace too small");
    const bool huf = level >= 30;
    void* garbageTable = a.table;
    if (hcLevel) a.table = (u32*)hc_slot();
    a.maxBlock = (u32)kHcMaxBlock; a.poolMask = 0;
    a.wideOcc = ((base == 11 || base == 22 || base == 20) && !(seed & 2u)) ? (u32*)malloc(8192 + 8) : nullptr;     // levels 11/31, 22/42: with and without the occupancy summary; 20/40: with and without the slot codes
    if (a.wideOcc) memset(a.wideOcc, 0x77, 8192 + 8);
    a.hcRegion = (u32*)aligned_alloc(64, 4 * LZ_HC_REGION_WORDS + 64);
    memset(a.hcRegion, 0x3C, 4 * LZ_HC_REGION_WORDS);
    if (level == 32) lzemu::run_wave(entry_block<LZ_PARSER_HASHCHAIN, 14, 6, true>, &a, seed);
    else if (ncLevel) lzemu::run_wave(huf ? entry_block<LZ_PARSER_HASHCHAIN, 18, 6, true> : entry_block<LZ_PARSER_HASHCHAIN, 18, 6, false>, &a, seed);
    else switch (base) {
    case 13:                   lzemu::run_wave(huf ? entry_block<LZ_PARSER_HASHCHAIN, 18, 7, true> : entry_block<LZ_PARSER_HASHCHAIN, 18, 7, false>, &a, seed); break;
    case 14:                   lzemu::run_wave(huf ? entry_block<LZ_PARSER_HASHCHAIN, 18, 8, true> : entry_block<LZ_PARSER_HASHCHAIN, 18, 8, false>, &a, seed); break;
    case 15:                   lzemu::run_wave(huf ? entry_block<LZ_PARSER_HASHCHAIN, 18, 9, true> : entry_block<LZ_PARSER_HASHCHAIN, 18, 9, false>, &a, seed); break;
    case 16: case 17:          lzemu::run_wave(huf ? entry_block<LZ_PARSER_HASHCHAIN, 18, 4, true> : entry_block<LZ_PARSER_HASHCHAIN, 18, 4, false>, &a, seed); break;
    case 10: lzemu::run_wave(huf ? entry_block<LZ_PARSER_FAST, 12, 0, true> : entry_block<LZ_PARSER_FAST, 12, 0, false>, &a, seed); break;
    case 11: lzemu::run_wave(huf ? entry_block<LZ_PARSER_FAST, 18, 0, true> : entry_block<LZ_PARSER_FAST, 18, 0, false>, &a, seed); break;
    case 20: lzemu::run_wave(huf ? entry_block<LZ_PARSER_FASTBIG, 14, 10, true> : entry_block<LZ_PARSER_FASTBIG, 14, 10, false>, &a, seed); break;
    case 21: lzemu::run_wave(huf ? entry_block<LZ_PARSER_PRICEFAST, 14, 12, true> : entry_block<LZ_PARSER_PRICEFAST, 14, 12, false>, &a, seed); break;
    default: lzemu::run_wave(huf ? entry_block<LZ_PARSER_PRICEFAST, 18, 12, true> : entry_block<LZ_PARSER_PRICEFAST, 18, 12, false>, &a, seed); break;
    }
    free(garbageTable); free(a.tag); free(a.scratch); free(a.hcRegion); free(a.wideOcc);
    return (int)a.result;
}

// One Huffman-candidate stream through lz_put_stream_huf (Lizard_writeStream semantics):
// out receives LE24 n ‖ LE24 c ‖ payload (accepted) or LE24 n ‖ raw bytes; returns bytes written,
// *huffed tells which.
namespace {
struct HufArgs { const u8* stream; u32 n; u8* out; u32* ws; u32 result; u32 huffed; };
void entry_huf(void* a)
{
    HufArgs* x = (HufArgs*)a;
    u32 h = 0;
    u32 r = lz_put_stream_huf(x->out, x->stream, x->n, x->ws, &h);
    if (lz_lane() == 0) { x->result = r; x->huffed = h; }
}
}  // namespace

extern "C" int emul_put_stream_huf(const void* stream, int n, void* out, int* huffed, unsigned seed)
{
    HufArgs a;
    a.stream = (const u8*)stream; a.n = (u32)n; a.out = (u8*)out; a.result = 0; a.huffed = 0;
    a.ws = (u32*)malloc(4 * LZ_HUF_WS_WORDS);
    memset(a.ws, 0x77, 4 * LZ_HUF_WS_WORDS);
    lzemu::run_wave(entry_huf, &a, seed);
    free(a.ws);
    *huffed = (int)a.huffed;
    return (int)a.result;
}


// Wave-wide helpers of lz_block.h against scalar definitions on one buffer: forward/backward common lengths
// (lz_count_fwd, lz_count_back, lz_count_both), lz_copy and the scan/reduce primitives.  Returns the number of
// disagreements (0 = all good).  `pairs` = n triples (P, M, limitOrAnchor) with M < P.
namespace {
struct HelperArgs { const u8* buf; u32 n; const u32* trip; u32 ntrip; u8* tmp; u32 bad; };
void entry_helpers(void* a)
{
    HelperArgs* x = (HelperArgs*)a;
    const u32 lane = lz_lane();
    u32 bad = 0;
    for (u32 t = 0; t < x->ntrip; t++) {
        const u32 P = x->trip[3 * t], M = x->trip[3 * t + 1], lim = x->trip[3 * t + 2];   // M < P <= lim <= n - 16
        // scalar definitions
        u32 f = 0; while (P + f < lim && x->buf[P + f] == x->buf[M + f]) f++;
        const u32 anchor = M > P / 2u ? P / 2u : M / 2u;                                       // some anchor <= P
        u32 b = 0; while (P - b > anchor && M - b > 0u && x->buf[P - b - 1u] == x->buf[M - b - 1u]) b++;
        const u32 gf = lz_count_fwd(x->buf, P, M, lim);
        const u32 gb = lz_count_back(x->buf, P, M, anchor);
        u32 hf = 0, hb = 0;
        lz_count_both(x->buf, P, M, lim, anchor, hf, hb);
        if (gf != f || gb != b || hf != f || hb != b) bad++;
        // copy of the matched span into tmp and back-check
        const u32 len = f < 3000u ? f + (t & 7u) : 3000u;
        lz_copy(x->tmp, x->buf + M, len);
        lz_wave_sync();
        u32 diff = 0;
        for (u32 i = lane; i < len; i += 64u) diff += x->tmp[i] != x->buf[M + i];
        if (lz_wave_reduce_add(diff)) bad++;
        lz_wave_sync();
        // scans: values derived from the data
        const u32 v = x->buf[(P + lane) % x->n];
        const u32 ex = lz_wave_scan_excl_add(v);
        u32 want = 0; for (u32 l = 0; l < lane; l++) want += x->buf[(P + l) % x->n];
        u32 mx = 0; for (u32 l = 0; l < 64u; l++) { const u32 w = x->buf[(P + l) % x->n]; mx = w > mx ? w : mx; }
        const u64 wrong = lz_ballot(ex != want);
        if (wrong || lz_readlane(lz_wave_reduce_max(v), 63u) != mx) bad++;
    }
    if (lane == 0) x->bad = bad;
}
}  // namespace

extern "C" int emul_check_helpers(const void* buf, int n, const unsigned* triples, int ntriples, unsigned seed)
{
    HelperArgs a;
    a.buf = (const u8*)buf; a.n = (u32)n; a.trip = triples; a.ntrip = (u32)ntriples; a.bad = 0;
    a.tmp = (u8*)malloc(4096);
    lzemu::run_wave(entry_helpers, &a, seed);
    free(a.tmp);
    return (int)a.bad;
}


// One block through the product's decoder body (lz_unpack.h).  Returns the decoded size, -1 = refused.
namespace {
struct DecArgs { const u8* in; u32 n; u8* out; u32 cap; u8* stage; u32* ws; u32 result; };
void entry_dec(void* a)
{
    DecArgs* x = (DecArgs*)a;
    const u32 r = lz_decompress_block(x->in, x->n, x->out, x->cap, x->stage, x->ws);
    if (lz_lane() == 0) x->result = r;
}
}  // namespace

extern "C" int emul_decompress_block(const void* src, int n, void* dst, int cap, unsigned seed)
{
    DecArgs a;
    a.in = (const u8*)src; a.n = (u32)n; a.out = (u8*)dst; a.cap = (u32)cap; a.result = 0;
    a.stage = (u8*)malloc(4 * LZD_STAGE_BYTES);
    a.ws = (u32*)malloc(4 * LZD_WS_WORDS);
    memset(a.stage, 0xDD, 4 * LZD_STAGE_BYTES);
    memset(a.ws, 0x3C, 4 * LZD_WS_WORDS);
    lzemu::run_wave(entry_dec, &a, seed);
    free(a.stage); free(a.ws);
    return a.result == LZD_ERR ? -1 : (int)a.result;
}


// Levels 10 / 30 in the producer / consumer form (lizard_amd/csrc/lz_split.h): nProd + nCons emulated waves, each on an OS thread
// of its own, share one "LDS" (mailboxes, free masks, counters) and one scratch arena, exactly as the waves of one workgroup of
// lz_fast12_split_kernel do.  Block i is src + i*blockSize (the last one lastBlockSize bytes) -> dst + i*dstStride, sizes[i].
namespace {
struct SplitWave { LzSplitArgs a; LzSplitShared sh; u32 wave; void* table; u64* ring; u32* hufWs; bool huf; };
void entry_split_init(void* p) { SplitWave* w = (SplitWave*)p; lz_split_shared_init(w->sh, w->a.nProd, w->a.nCons, w->a.nBufs, w->a.qn); }
void entry_split(void* p)
{
    SplitWave* w = (SplitWave*)p;
    if (w->wave < w->a.nProd) lz_split_producer<12>(w->a, w->sh, w->wave, w->table, w->ring);
    else if (w->huf)          lz_split_consumer<true>(w->a, w->sh, w->wave - w->a.nProd, w->hufWs);
    else                      lz_split_consumer<false>(w->a, w->sh, w->wave - w->a.nProd, w->hufWs);
}
}  // namespace

// srcSizes / activeProd: the ragged batches and the producer cap of small launches (LzBatch::srcSizes, ::activeWaves); nullptr / 0 = off
extern "C" int emul_compress_split_ragged(const void* src, int nBlocks, int blockSize, int lastBlockSize, const unsigned* srcSizes, void* dst,
                                          int dstStride, unsigned* sizes, int level, int nProd, int nCons, int activeProd, unsigned seed);
extern "C" int emul_compress_split(const void* src, int nBlocks, int blockSize, int lastBlockSize, void* dst, int dstStride,
                                   unsigned* sizes, int level, int nProd, int nCons, unsigned seed)
{
    return emul_compress_split_ragged(src, nBlocks, blockSize, lastBlockSize, nullptr, dst, dstStride, sizes, level, nProd, nCons, 0, seed);
}
extern "C" int emul_compress_split_ragged(const void* src, int nBlocks, int blockSize, int lastBlockSize, const unsigned* srcSizes, void* dst,
                                          int dstStride, unsigned* sizes, int level, int nProd, int nCons, int activeProd, unsigned seed)
{
    const u32 nBufs = 2u + (seed & 1u), qn = 64u;              // odd seeds: three buffers per producer
    if ((level != 10 && level != 30) || nProd < 1 || nCons < 1 || (u32)nProd * nBufs > qn) return -1;
    LzSplitArgs a;
    a.src = (const u8*)src; a.blockSize = (u64)blockSize; a.nBlocks = (u32)nBlocks; a.lastBlockSize = (u32)lastBlockSize;
    a.srcSizes = srcSizes; a.activeProd = activeProd > 0 ? (u32)activeProd : 0xFFFFFFFFu;
    a.dst = (u8*)dst; a.dstStride = (u64)dstStride; a.sizes = sizes; a.level = (u32)level;
    u32 counter = 0; a.counter = &counter;
    const size_t arenaBytes = LZ_SPLIT_ARENA_BYTES((size_t)nProd, (size_t)nCons, nBufs);
    a.arena = (u8*)malloc(arenaBytes); memset(a.arena, 0xC7, arenaBytes);
    a.nProd = (u32)nProd; a.nCons = (u32)nCons; a.nBufs = nBufs; a.qn = qn;
    std::vector<u32> shared(LZ_SPLIT_SHARED_WORDS((u32)nProd, (u32)nCons, qn), 0xA5A5A5A5u);
    const LzSplitShared sh = lz_split_shared(shared.data(), (u32)nProd, (u32)nCons);
    const size_t tabBytes = LZ_TAB_BYTES(12) + 64;
    std::vector<u8> tables((size_t)nProd * tabBytes, 0x5A);
    std::vector<u64> rings((size_t)nProd * LZ_SEQ_RING, 0xEEEEEEEEEEEEEEEEull);
    std::vector<u32> ws((size_t)nCons * LZ_HUF_WS_WORDS, 0x77777777u);
    std::vector<SplitWave> waves((size_t)(nProd + nCons));
    for (int w = 0; w < nProd + nCons; w++) {
        SplitWave& x = waves[(size_t)w];
        x.a = a; x.sh = sh; x.wave = (u32)w; x.huf = level >= 30;
        x.table = w < nProd ? (void*)&tables[(size_t)w * tabBytes] : nullptr;
        x.ring = w < nProd ? &rings[(size_t)w * LZ_SEQ_RING] : nullptr;
        x.hufWs = w >= nProd ? &ws[(size_t)(w - nProd) * LZ_HUF_WS_WORDS] : nullptr;
    }
    lzemu::run_wave(entry_split_init, &waves[0], seed);
    std::vector<std::thread> th;
    for (int w = 0; w < nProd + nCons; w++) th.emplace_back([&waves, w, seed] { lzemu::run_wave(entry_split, &waves[(size_t)w], seed * 31u + (unsigned)w + 1u); });
    for (auto& t : th) t.join();
    free(a.arena);
    return 0;
}
