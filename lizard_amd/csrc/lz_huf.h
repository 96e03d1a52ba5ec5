// lz_huf.h — huff0 stage of the Lizard container on one wavefront (levels >= 30).
//
// Bit-exact with the reference's HUF_compress() as Lizard calls it (lib/entropy/huf_compress.c:609 ->
// HUF_compress2(.., 255, 11) -> 4 streams, :517-573) and with the accept rule of Lizard_writeStream
// (lib/lizard_compress.c:143-168).  Mapping onto the wave:
//   histogram      LDS atomics, 4 bytes per lane per step                     (fse_compress.c:315-438)
//   sort           every lane ranks 4 symbols against all counts (a stable descending sort is unique,
//                  so the reference's bucketed insertion sort needs no emulation)   (huf_compress.c:305)
//   tree, depth limit, canonical codes, weight header (FSE)   <=256 symbols, integer, inherently
//                  serial: lane 0 alone, tables in LDS                       (huf_compress.c:223-402,81-165)
//   exact sizes    per-segment sum of code lengths -> the 4 bitstream sizes are known BEFORE encoding,
//                  so the accept/reject decision is taken first, rejected streams are never encoded
//                  and accepted ones are written straight to their final place in dst
//   bit packing    4 symbols per lane per step, wave prefix sum of bit lengths, ds_or into an LDS
//                  staging ring, whole dwords stored coalesced              (huf_compress.c:427-513)
// All cross-lane hand-offs in this file go through LDS (histogram, tables, staging ring, the header size),
// so lz_lds_sync() is the ordering point: it does not wait for the wave's pending global stores.
#pragma once
#include "lz_wave.h"

// Sub-phase clocks of the stage in -DLZ_PROFILE builds (slots 8..14 of the per-wave profile record).
#ifdef LZ_PROFILE
#define LZ_HPROF_PARAM , u64* hp_
#define LZ_HPROF_ARG(st) , (st).prof
#define LZ_HPROF(k) do { const u64 t_ = __builtin_readcyclecounter(); hp_[k] += t_ - hp_[15]; hp_[15] = t_; } while (0)
#else
#define LZ_HPROF_PARAM
#define LZ_HPROF_ARG(st)
#define LZ_HPROF(k) ((void)0)
#endif
// -DLZ_PROFILE -DLZ_HPROF_FINE: the tree / codes / header phase split up — slots 0 merge, 1 depths, 2 depth limit, 3 canonical codes,
// 10 weight header alone (the parsers' marks 0..4 are dropped in such a build, lz_block.h)
#if defined(LZ_PROFILE) && defined(LZ_HPROF_FINE)
#define LZ_HPROF_F(k) LZ_HPROF(k)
#else
#define LZ_HPROF_F(k) ((void)0)
#endif

#define LZ_HUF_MAXBITS     12u    // HUF_TABLELOG_MAX, huf.h:118
#define LZ_HUF_DEFAULTLOG  11u    // HUF_TABLELOG_DEFAULT, huf.h:119

// LDS workspace (u32 words): 2 KiB per wave.  Only the wave-parallel steps use it (histogram, sort scatter, the two
// index exchanges, the code table, the packer's ring); everything serial lives in registers (see LzV64 / LzV256).
#define LZ_HUF_WS_CTAB     0u                          // u16[256]: val | nbBits << 12
#define LZ_HUF_WS_COUNT    128u                        // u32[256]: histogram -> sort scatter -> depth / code-length exchange
#define LZ_HUF_WS_STAGE    128u                        // packer staging ring: OVER the histogram (dead once the code table exists) and the words behind it
#define LZ_HUF_STAGE_WORDS 384u                        // 1 024 codes of <= 11 bits (HUF_compress2(.., 255, 11)) + 31 pending bits + two words of spill = 356
#define LZ_HUF_WS_WORDS    512u                        // (the parser's 2 KiB round tag array aliases it)

LZ_DEV u32 lz_highbit(u32 v) { return 31u - (u32)__builtin_clz(v); }   // BIT_highbit32, v != 0

// FSE_optimalTableLog_internal, fse_compress.c:477-496
LZ_DEV u32 lz_fse_optimal_tablelog(u32 maxTableLog, u32 srcSize, u32 maxSym, u32 minus)
{
    const u32 hb = lz_highbit(srcSize - 1u);
    const u32 maxBitsSrc = hb - minus, minBitsSrc = hb + 1u, minBitsSym = lz_highbit(maxSym) + 2u;
    const u32 minBits = minBitsSrc < minBitsSym ? minBitsSrc : minBitsSym;
    u32 tl = maxTableLog;
    if (maxBitsSrc < tl) tl = maxBitsSrc;
    if (minBits > tl) tl = minBits;
    if (tl < 5u) tl = 5u;
    if (tl > 12u) tl = 12u;
    return tl;
}

// ---- tables in registers --------------------------------------------------------------------------------------
// The tree build, the depth limit, the FSE weight header: <= 256 symbols, integer, every step depends on the one before.
// The reference runs them on one core with its tables in L1; one lane with tables in LDS pays ~130 clocks per
// dependent access.  Here they run as WAVE-UNIFORM code — every lane executes the same scalar control flow, loop
// counters and accumulators live in SGPRs — on tables spread over the lanes of a VGPR: entry i of a 64-entry table is
// lane i of one register (v_readlane / v_writelane with a scalar index, a few clocks, no memory), a 256-entry table is
// four such registers.  Indices and stored values must be wave-uniform.
struct LzV64 {
    u32 r;
    LZ_DEVM u32  get(u32 i) const { return lz_readlane(r, i); }
    LZ_DEVM void set(u32 i, u32 x) { r = lz_writelane(r, x, i); }
};
struct LzV256 {
    u32 r0, r1, r2, r3;
    // branch-free: four independent v_readlane and a scalar select (a 4-way branch costs more than the three spare reads)
    LZ_DEVM u32 get(u32 i) const
    {
        const u32 l = i & 63u, c = i >> 6;
        const u32 a = lz_readlane(r0, l), b = lz_readlane(r1, l), d = lz_readlane(r2, l), e = lz_readlane(r3, l);
        return c == 0 ? a : c == 1 ? b : c == 2 ? d : e;
    }
    LZ_DEVM void set(u32 i, u32 x)
    {
        const u32 l = i & 63u, c = i >> 6;
        r0 = lz_writelane(r0, x, c == 0 ? l : 64u); r1 = lz_writelane(r1, x, c == 1 ? l : 64u);
        r2 = lz_writelane(r2, x, c == 2 ? l : 64u); r3 = lz_writelane(r3, x, c == 3 ? l : 64u);
    }
    // the register that holds entries 64c .. 64c+63 (c wave-uniform), and putting one back: what a cursor that moves through the
    // table in one direction works on (lz_put_stream_huf's two-queue merge)
    LZ_DEVM u32  reg(u32 c) const { return c == 0 ? r0 : c == 1 ? r1 : c == 2 ? r2 : r3; }
    LZ_DEVM void putReg(u32 c, u32 v) { r0 = c == 0 ? v : r0; r1 = c == 1 ? v : r1; r2 = c == 2 ? v : r2; r3 = c == 3 ? v : r3; }
};

// LSB-first bit writer whose output is a register table of 64 dwords (the weight header is < 256 bytes)
struct LzBitV { LzV64 out; u64 acc; u32 nb, words; };
LZ_DEV void lz_bv_init(LzBitV& b) { b.out.r = 0; b.acc = 0; b.nb = 0; b.words = 0; }
LZ_DEV void lz_bv_flush(LzBitV& b) { if (b.nb >= 32u) { b.out.set(b.words & 63u, (u32)b.acc); b.words++; b.acc >>= 32; b.nb -= 32u; } }
LZ_DEV void lz_bv_add(LzBitV& b, u32 v, u32 n)                  // n <= 16
{
    b.acc |= (u64)(v & ((1u << n) - 1u)) << b.nb;
    b.nb += n;
    lz_bv_flush(b);
}
LZ_DEV void lz_bv_align(LzBitV& b) { b.nb = (b.nb + 7u) & ~7u; lz_bv_flush(b); }     // next bits start on a byte boundary
LZ_DEV u32  lz_bv_bytes(const LzBitV& b) { return 4u * b.words + (b.nb >> 3); }    // after lz_bv_align

// HUF_setMaxHeight, huf_compress.c:223-297, on the sorted leaves 0..lastNonNull: bits = code length by rank, leaf = count << 8 | symbol.
// The reference's three walks over the ranks — cut the over-long codes (:233-238), skip the codes of exactly maxNbBits (:239), find
// the last rank of every shorter length (:251-257) — are lane-parallel here (round 6: they were ~250 iterations of wave-uniform code
// on register tables); the repayment loops (:259-292), a few dozen steps that depend on each other, stay wave-uniform.
LZ_DEV int lz_v256_highest(u64 m0, u64 m1, u64 m2, u64 m3)      // highest position whose bit is set in the four 64-entry masks, -1 if none
{
    return m3 ? 255 - (int)lz_clz64(m3) : m2 ? 191 - (int)lz_clz64(m2) : m1 ? 127 - (int)lz_clz64(m1) : m0 ? 63 - (int)lz_clz64(m0) : -1;
}
LZ_DEV u32 lz_huf_set_max_height(const LzV256& leaf, LzV256& bits, u32 lastNonNull, u32 maxNbBits)
{
    const u32 largestBits = bits.get(lastNonNull);
    if (largestBits <= maxNbBits) return largestBits;
    const u32 noSymbol = 0xF0F0F0F0u;
    const u32 lane = lz_lane();
    const u32 baseCost = 1u << (largestBits - maxNbBits);
    int totalCost, n;
    {
        // :233-238 — from the last rank downwards while the code is too long: the ranks above the highest one that fits
        const int stop = lz_v256_highest(lz_ballot(bits.r0 <= maxNbBits && lane <= lastNonNull), lz_ballot(bits.r1 <= maxNbBits && 64u + lane <= lastNonNull),
                                         lz_ballot(bits.r2 <= maxNbBits && 128u + lane <= lastNonNull), lz_ballot(bits.r3 <= maxNbBits && 192u + lane <= lastNonNull));
        u32 cost = 0;
#define LZ_CUT(R, BASE) do { const bool cut_ = (int)((BASE) + lane) > stop && (BASE) + lane <= lastNonNull; \
                             cost += cut_ ? baseCost - (1u << (largestBits - (cut_ ? (R) : largestBits))) : 0u; (R) = cut_ ? maxNbBits : (R); } while (0)
        LZ_CUT(bits.r0, 0u); LZ_CUT(bits.r1, 64u); LZ_CUT(bits.r2, 128u); LZ_CUT(bits.r3, 192u);
#undef LZ_CUT
        totalCost = (int)lz_wave_reduce_add(cost);
        // :239 — on to the last rank whose code is shorter than maxNbBits (-1: the reference stops on its barrier node)
        n = lz_v256_highest(lz_ballot(bits.r0 != maxNbBits && (int)lane <= stop), lz_ballot(bits.r1 != maxNbBits && (int)(64u + lane) <= stop),
                            lz_ballot(bits.r2 != maxNbBits && (int)(128u + lane) <= stop), lz_ballot(bits.r3 != maxNbBits && (int)(192u + lane) <= stop));
    }
    totalCost >>= (largestBits - maxNbBits);
    // :251-257 — rankLast[maxNbBits - v] = the rank at which the walk from n downwards first sees a length <= v, if that length IS v:
    // H(v) = highest rank <= n with a length <= v; the walk's running minimum drops to v there iff H(v) != H(v - 1).
    LzV64 rankLast; rankLast.r = noSymbol;                       // [LZ_HUF_MAXBITS + 2]
    {
        int below = -1;                                          // H(v - 1); no code has length 0
        for (u32 v = 1; v < maxNbBits; v++) {
            const int h = lz_v256_highest(lz_ballot(bits.r0 <= v && (int)lane <= n), lz_ballot(bits.r1 <= v && (int)(64u + lane) <= n),
                                          lz_ballot(bits.r2 <= v && (int)(128u + lane) <= n), lz_ballot(bits.r3 <= v && (int)(192u + lane) <= n));
            if (h != below) rankLast.set(maxNbBits - v, (u32)h);
            below = h;
        }
    }
    while (totalCost > 0) {
        u32 dec = lz_highbit((u32)totalCost) + 1u;
        for (; dec > 1u; dec--) {
            const u32 highPos = rankLast.get(dec), lowPos = rankLast.get(dec - 1u);
            if (highPos == noSymbol) continue;
            if (lowPos == noSymbol) break;
            if ((leaf.get(highPos) >> 8) <= 2u * (leaf.get(lowPos) >> 8)) break;
        }
        while (dec <= LZ_HUF_MAXBITS && rankLast.get(dec) == noSymbol) dec++;
        totalCost -= 1 << (dec - 1u);
        if (rankLast.get(dec - 1u) == noSymbol) rankLast.set(dec - 1u, rankLast.get(dec));
        {
            const u32 rl = rankLast.get(dec);
            bits.set(rl, bits.get(rl) + 1u);
            if (rl == 0) rankLast.set(dec, noSymbol);
            else {
                rankLast.set(dec, rl - 1u);
                if (bits.get(rl - 1u) != maxNbBits - dec) rankLast.set(dec, noSymbol);
            }
        }
    }
    while (totalCost < 0) {
        if (rankLast.get(1) == noSymbol) {
            while (bits.get((u32)n) == maxNbBits) n--;
            bits.set((u32)(n + 1), bits.get((u32)(n + 1)) - 1u);
            rankLast.set(1, (u32)(n + 1));
            totalCost++;
            continue;
        }
        {
            const u32 rl = rankLast.get(1) + 1u;
            bits.set(rl, bits.get(rl) - 1u);
            rankLast.set(1, rl);
        }
        totalCost++;
    }
    return maxNbBits;
}

// FSE_normalizeCount + FSE_normalizeM2, fse_compress.c:507-641.  count / norm: tables of <= 13 entries (norm holds
// signed values).  Returns false on the reference's error returns.
LZ_DEV bool lz_fse_normalize(LzV64& norm, u32 tableLog, const LzV64& count, u32 total, u32 maxSym)
{
    const u64 scale = 62u - tableLog, step = ((u64)1 << 62) / total, vStep = 1ULL << (scale - 20u);
    int still = 1 << tableLog;
    u32 largest = 0;
    int largestP = 0;
    const u32 lowThreshold = total >> tableLog;
    {
        const u32 minBitsSrc = lz_highbit(total - 1u) + 1u, minBitsSym = lz_highbit(maxSym) + 2u;
        if (tableLog < (minBitsSrc < minBitsSym ? minBitsSrc : minBitsSym)) return false;
    }
    for (u32 s = 0; s <= maxSym; s++) {
        const u32 c = count.get(s);
        if (c == 0) { norm.set(s, 0); continue; }
        if (c <= lowThreshold) { norm.set(s, (u32)-1); still--; }
        else {
            int proba = (int)(short)((c * step) >> scale);
            if (proba < 8) {                                  // rtbTable, fse_compress.c:592
                const u32 r = proba == 1 ? 473195u : proba == 2 ? 504333u : proba == 3 ? 520860u : proba == 4 ? 550000u
                            : proba == 5 ? 700000u : proba == 6 ? 750000u : proba == 7 ? 830000u : 0u;
                const u64 restToBeat = vStep * r;
                proba += (c * step) - ((u64)proba << scale) > restToBeat;
            }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm.set(s, (u32)proba);
            still -= proba;
        }
    }
    if (-still < ((int)norm.get(largest) >> 1)) { norm.set(largest, (u32)((int)norm.get(largest) + still)); return true; }
    // FSE_normalizeM2
    {
        u32 distributed = 0, toDistribute;
        u64 tot = total;
        u32 lowOne = (u32)((tot * 3u) >> (tableLog + 1u));
        for (u32 s = 0; s <= maxSym; s++) {
            const u32 c = count.get(s);
            if (c == 0) { norm.set(s, 0); continue; }
            if (c <= lowThreshold) { norm.set(s, (u32)-1); distributed++; tot -= c; continue; }
            if (c <= lowOne) { norm.set(s, 1); distributed++; tot -= c; continue; }
            norm.set(s, (u32)-2);
        }
        toDistribute = (1u << tableLog) - distributed;
        if ((tot / toDistribute) > lowOne) {
            lowOne = (u32)((tot * 3u) / (toDistribute * 2u));
            for (u32 s = 0; s <= maxSym; s++) {
                const u32 c = count.get(s);
                if ((int)norm.get(s) == -2 && c <= lowOne) { norm.set(s, 1); distributed++; tot -= c; }
            }
            toDistribute = (1u << tableLog) - distributed;
        }
        if (distributed == maxSym + 1u) {
            u32 maxV = 0, maxC = 0;
            for (u32 s = 0; s <= maxSym; s++) { const u32 c = count.get(s); if (c > maxC) { maxV = s; maxC = c; } }
            norm.set(maxV, (u32)((int)norm.get(maxV) + (int)toDistribute));
            return true;
        }
        const u64 vStepLog = 62u - tableLog, mid = (1ULL << (vStepLog - 1u)) - 1u;
        const u64 rStep = ((((u64)1 << vStepLog) * toDistribute) + mid) / tot;
        u64 tmpTotal = mid;
        for (u32 s = 0; s <= maxSym; s++) if ((int)norm.get(s) == -2) {
            const u64 end = tmpTotal + (count.get(s) * rStep);
            const u32 weight = (u32)(end >> vStepLog) - (u32)(tmpTotal >> vStepLog);
            if (weight < 1u) return false;
            norm.set(s, weight);
            tmpTotal = end;
        }
    }
    return true;
}

// HUF_compressWeights, huf_compress.c:81-121: FSE-compress the weights of symbols 0..wtSize-1 (byte stream, LSB first) into the
// LDS words lds[256 ...] (lds: 384 words; the first 256 are work space).  wt4: lane l holds the weights of symbols 4l..4l+3, one per
// byte.  count[w]: number of symbols with weight w.  Returns the compressed size, 0 = not compressible, 1 = all equal,
// 0xFFFFFFFF = reference error.
// The two interleaved FSE states are two chains of ~wtSize / 2 dependent steps: they run on LANES 0 and 1 side by side (round 6;
// they were ~255 steps of wave-uniform code, 25 instructions each, 10 % of a level-30 consumer's time).  What a step needs of its
// symbol (deltaNbBits, deltaFindState) is looked up for all steps at once beforehand, what it emits (value, bit count) goes to the
// same LDS word; the bit stream is put together from those words by all lanes, like the code streams.
LZ_DEV u32 lz_huf_compress_weights(LzBitV& b, u32 wt4, u32 wtSize, const LzV64& count, u32* lds)
{
#define LZ_WT(i) ((lz_readlane(wt4, (i) >> 2) >> (8u * ((i) & 3u))) & 255u)
    lz_bv_init(b);
    if (wtSize <= 1u) return 0;
    u32 maxSym = LZ_HUF_MAXBITS, maxCount = 0;
    while (!count.get(maxSym)) maxSym--;
    for (u32 s = 0; s <= maxSym; s++) { const u32 c = count.get(s); if (c > maxCount) maxCount = c; }
    if (maxCount == wtSize) return 1;
    if (maxCount == 1u) return 0;
    const u32 tableLog = lz_fse_optimal_tablelog(6u, wtSize, maxSym, 2u);
    LzV64 norm; norm.r = 0;
    if (!lz_fse_normalize(norm, tableLog, count, wtSize, maxSym)) return 0xFFFFFFFFu;
    {   // FSE_writeNCount_generic, fse_compress.c:204-289
        int nbBits = (int)tableLog + 1, remaining = (1 << tableLog) + 1, threshold = 1 << tableLog;
        u32 charnum = 0;
        bool previous0 = false;
        lz_bv_add(b, tableLog - 5u, 4u);
        while (remaining > 1) {
            if (previous0) {
                u32 start = charnum;
                while (!norm.get(charnum)) charnum++;
                while (charnum >= start + 24u) { start += 24u; lz_bv_add(b, 0xFFFFu, 16u); }
                while (charnum >= start + 3u) { start += 3u; lz_bv_add(b, 3u, 2u); }
                lz_bv_add(b, charnum - start, 2u);
            }
            int cnt = (int)norm.get(charnum++);
            const int mx = (2 * threshold - 1) - remaining;
            remaining -= cnt < 0 ? -cnt : cnt;
            cnt++;
            if (cnt >= threshold) cnt += mx;
            lz_bv_add(b, (u32)cnt, (u32)(nbBits - (cnt < mx)));
            previous0 = (cnt == 1);
            if (remaining < 1) return 0xFFFFFFFFu;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (charnum > maxSym + 1u) return 0xFFFFFFFFu;
        lz_bv_align(b);
    }
    // FSE_buildCTable_wksp, fse_compress.c:103-182: all tables have at most 64 entries
    LzV64 cumul, tableSym, stateTable, dBits, dFind;
    cumul.r = tableSym.r = stateTable.r = dBits.r = dFind.r = 0;
    {
        const u32 tableSize = 1u << tableLog, mask = tableSize - 1u, step = (tableSize >> 1) + (tableSize >> 3) + 3u;
        u32 high = tableSize - 1u, position = 0, total = 0;
        cumul.set(0, 0);
        for (u32 u = 1; u <= maxSym + 1u; u++) {
            const int nv = (int)norm.get(u - 1u);
            if (nv == -1) { cumul.set(u, cumul.get(u - 1u) + 1u); tableSym.set(high--, u - 1u); }
            else cumul.set(u, cumul.get(u - 1u) + (u32)nv);
        }
        for (u32 s = 0; s <= maxSym; s++) {
            const int nv = (int)norm.get(s);
            for (int k = 0; k < nv; k++) {
                tableSym.set(position, s);
                position = (position + step) & mask;
                while (position > high) position = (position + step) & mask;
            }
        }
        if (position != 0) return 0xFFFFFFFFu;
        for (u32 u = 0; u < tableSize; u++) { const u32 sy = tableSym.get(u); const u32 cu = cumul.get(sy); stateTable.set(cu, tableSize + u); cumul.set(sy, cu + 1u); }
        for (u32 s = 0; s <= maxSym; s++) {
            const int nv = (int)norm.get(s);
            if (nv == 0) { dBits.set(s, 0); dFind.set(s, 0); }
            else if (nv == -1 || nv == 1) { dBits.set(s, (tableLog << 16) - (1u << tableLog)); dFind.set(s, total - 1u); total++; }
            else {
                const u32 maxBitsOut = tableLog - lz_highbit((u32)nv - 1u);
                const u32 minStatePlus = (u32)nv << maxBitsOut;
                dBits.set(s, (maxBitsOut << 16) - minStatePlus); dFind.set(s, total - (u32)nv); total += (u32)nv;
            }
        }
    }
    // FSE_compress_usingCTable_generic, fse_compress.c:701-758 (+ fse.h:525-564): two interleaved states.  In emission order the
    // steps are e = 0 .. wtSize - 3, step e encodes the weight of symbol wtSize - 3 - e; CState1 takes the even steps when wtSize is
    // odd (its extra first step, :718-722) and the odd ones otherwise, CState2 the others; the states start from the last two weights.
    if (wtSize <= 2u) return 0;
    const u32 lane = lz_lane();
    const u32 pos = lz_bv_bytes(b);                             // bytes of the NCount header in front of the stream
    const u32 steps = wtSize - 2u;
    u32* const arr = lds;                                       // [256] per step: deltaNbBits << 8 | deltaFindState, then value | bits << 8
    u32* const hb = lds + 256u;                                 // [128] the header's bytes
#define LZ_WTV(i) ((lz_shfl(wt4, (i) >> 2) >> (8u * ((i) & 3u))) & 255u)   /* weight of symbol i, i per lane */
    for (u32 j = 0; j < 4u; j++) {
        const u32 e = 4u * lane + j;
        const u32 sy = LZ_WTV(e < steps ? wtSize - 3u - e : 0u);
        const u32 db = lz_shfl(dBits.r, sy), df = lz_shfl(dFind.r, sy);
        if (e < steps) arr[e] = (db << 8) | (df & 255u);
    }
    hb[lane] = 0; hb[64u + lane] = 0;
    lz_lds_sync();
    u32 S;                                                      // lane 0: CState1, lane 1: CState2 (the other lanes idle along)
    const u32 par = (lane & 1u) ^ ((wtSize & 1u) ^ 1u);         // my steps are e = par, par + 2, ...
    {   // FSE_initCState2, fse.h:538-549
        const u32 sy = LZ_WTV(wtSize - 1u - par);
        const u32 db = lz_shfl(dBits.r, sy);
        const int df = (int)lz_shfl(dFind.r, sy);
        const u32 nbo = (db + (1u << 15)) >> 16;
        const u32 v = (nbo << 16) - db;
        S = lz_shfl(stateTable.r, (u32)((int)(v >> nbo) + df) & 63u);
    }
    for (u32 e = par; e - par < steps; e += 2u) {              // uniform trip count; a lane whose last step does not exist idles through it
        const bool mine = e < steps;
        const u32 w = arr[mine ? e : 0u];
        const u32 nbo = (S + (w >> 8)) >> 16;                   // FSE_encodeSymbol, fse.h:551-558
        const u32 Snext = lz_shfl(stateTable.r, (u32)((int)(S >> nbo) + (int)(signed char)(w & 255u)) & 63u);
        if (mine && lane < 2u) arr[e] = (S & ((1u << nbo) - 1u)) | (nbo << 8);
        S = mine ? Snext : S;
    }
    lz_lds_sync();
    // FSE_flushCState (:754-755) CState2 then CState1, and the end mark of BIT_closeCStream
    if (lane < 2u) arr[steps + 1u - lane] = (S & ((1u << tableLog) - 1u)) | (tableLog << 8);
    if (lane == 2u) arr[steps + 2u] = 1u | (1u << 8);
    lz_lds_sync();
    u32 acc = 0, len = 0;                                      // my four steps' bits: <= 24
    for (u32 j = 0; j < 4u; j++) {
        const u32 e = 4u * lane + j;
        const u32 w = e < steps + 3u ? arr[e] : 0u;
        acc |= (w & 255u) << len; len += w >> 8;
    }
    const u32 off = lz_wave_scan_excl_add(len);
    const u32 total = lz_readlane(off + len, 63u);
    if (len) {
        const u32 at = 8u * pos + off, sh = at & 31u;
        lz_lds_atomic_or(&hb[at >> 5], acc << sh);
        if (sh && (acc >> (32u - sh))) lz_lds_atomic_or(&hb[(at >> 5) + 1u], acc >> (32u - sh));
    }
    // the NCount bytes: whole dwords in the register table, the rest still in the accumulator
    if (lane < b.words) lz_lds_atomic_or(&hb[lane], b.out.r);
    if (lane == b.words && b.nb) lz_lds_atomic_or(&hb[lane], (u32)b.acc);
    lz_lds_sync();
#undef LZ_WTV
#undef LZ_WT
    return pos + ((total + 7u) >> 3);
}

// One 1X bitstream (huf_compress.c:427-470): symbols src[a..b) appended LAST -> FIRST, LSB first, then a
// single '1'.  Returns the stream's bytes, ceil((bits + 1) / 8) (uniform).  All lanes call; `stage` is LDS.
// A step takes 1 024 symbols: lane l the 16 that follow 16 l in append order, i.e. the 16 bytes that END at src[hi] — one
// 16-byte load, requested a step ahead — as four groups of four codes (<= 44 bits each: a u64).  One wave scan of the lanes'
// bit counts per step, <= 12 ds_or per lane into the ring, whole dwords stored coalesced.  (Round 6: 256 symbols per step
// cost one scan, one ring hand-over and three LDS waits per 256 symbols — a third of a consumer's time at level 30.)
LZ_DEV void lz_huf_pack4(u32 w4, const u16* ctab, u64& acc, u32& len)
{
    const u32 e0 = ctab[w4 >> 24], e1 = ctab[(w4 >> 16) & 255u], e2 = ctab[(w4 >> 8) & 255u], e3 = ctab[w4 & 255u];
    const u32 l0 = e0 >> 12, l1 = e1 >> 12, l2 = e2 >> 12, l3 = e3 >> 12;
    const u32 p01 = (e0 & 0xFFFu) | ((e1 & 0xFFFu) << l0), p23 = (e2 & 0xFFFu) | ((e3 & 0xFFFu) << l2);     // <= 24 bits each
    acc = (u64)p01 | ((u64)p23 << (l0 + l1));
    len = l0 + l1 + l2 + l3;
}
LZ_DEV void lz_huf_stage_or(u32* stage, u32 pos, u64 acc, u32 len)
{
    if (len) {
        const u32 w = pos >> 5, sh = pos & 31u;
        const u32 lo = (u32)acc, hi32 = (u32)(acc >> 32);
        const u32 w0 = lo << sh;
        const u32 w1 = sh ? ((lo >> (32u - sh)) | (hi32 << sh)) : hi32;
        const u32 w2 = sh ? (hi32 >> (32u - sh)) : 0u;
        if (w0) lz_lds_atomic_or(&stage[w], w0);
        if (w1) lz_lds_atomic_or(&stage[w + 1u], w1);
        if (w2) lz_lds_atomic_or(&stage[w + 2u], w2);
    }
}
LZ_DEV u32 lz_huf_pack_segment(const u8* src, u32 a, u32 b, u8* out, const u16* ctab, u32* stage)
{
    const u32 lane = lz_lane();
    constexpr u32 PER = 16u, STEP = 64u * PER;
    for (u32 i = lane; i < LZ_HUF_STAGE_WORDS; i += 64u) stage[i] = 0;
    lz_lds_sync();
    u32 cur = 0;            // uniform: bits pending in stage[0] (< 32)
    u32 wordsOut = 0;       // uniform: dwords already stored to `out`
    u32 remaining = b - a;  // uniform: symbols not yet appended; next to append is src[a + remaining - 1]
    // my 16 bytes of a full step, requested one step ahead (the step itself is a chain of LDS trips: lookups, scan, or, store)
    u32 wn[4] = { 0, 0, 0, 0 };
    if (remaining >= STEP) {
        const u8* q = src + a + remaining - PER - lane * PER;
        #pragma unroll
        for (u32 g = 0; g < 4u; g++) wn[g] = lz_ld32(q + 4u * g);
    }
    while (remaining > 0) {
        const u32 take = remaining < STEP ? remaining : STEP;
        const u32 first = lane * PER;                     // index (in append order) of my first symbol in this step
        u64 acc[4] = { 0, 0, 0, 0 }; u32 len[4] = { 0, 0, 0, 0 };
        u32 wc[4];
        #pragma unroll
        for (u32 g = 0; g < 4u; g++) wc[g] = wn[g];
        if (remaining >= 2u * STEP) {                     // uniform condition
            const u8* q = src + a + remaining - STEP - PER - lane * PER;
            #pragma unroll
            for (u32 g = 0; g < 4u; g++) wn[g] = lz_ld32(q + 4u * g);
        }
        if (first < take) {
            const u32 cnt = take - first < PER ? take - first : PER;
            const u32 hi = a + remaining - 1u - first;    // my first symbol; the others sit below it
            if (cnt == PER) {
                if (take != STEP) {
                    #pragma unroll
                    for (u32 g = 0; g < 4u; g++) wc[g] = lz_ld32(src + hi - 15u + 4u * g);
                }
                #pragma unroll
                for (u32 g = 0; g < 4u; g++) lz_huf_pack4(wc[3u - g], ctab, acc[g], len[g]);      // group g: src[hi - 4g] .. src[hi - 4g - 3]
            } else {
                for (u32 j = 0; j < cnt; j++) {
                    const u32 e = ctab[src[hi - j]], g = j >> 2;
                    const u64 v = (u64)(e & 0xFFFu) << len[g];
                    acc[0] |= g == 0 ? v : 0ull; acc[1] |= g == 1 ? v : 0ull; acc[2] |= g == 2 ? v : 0ull; acc[3] |= g == 3 ? v : 0ull;
                    len[0] += g == 0 ? e >> 12 : 0u; len[1] += g == 1 ? e >> 12 : 0u; len[2] += g == 2 ? e >> 12 : 0u; len[3] += g == 3 ? e >> 12 : 0u;
                }
            }
        }
        const u32 mine = len[0] + len[1] + len[2] + len[3];
        const u32 off = lz_wave_scan_excl_add(mine);
        const u32 total = lz_readlane(off + mine, 63u);
        {
            u32 pos = cur + off;
            #pragma unroll
            for (u32 g = 0; g < 4u; g++) { lz_huf_stage_or(stage, pos, acc[g], len[g]); pos += len[g]; }
        }
        lz_lds_sync();
        const u32 T = cur + total, full = T >> 5;
        // store the completed dwords (<= 353 per step: 1 024 codes of at most 11 bits + the pending bits), coalesced
        for (u32 i = lane; i < full; i += 64u) lz_st32(out + 4u * (wordsOut + i), stage[i]);
        const u32 carry = stage[full];
        lz_lds_sync();
        for (u32 i = lane; i < full + 3u; i += 64u) stage[i] = (i == 0) ? carry : 0u;
        lz_lds_sync();
        wordsOut += full; cur = T & 31u; remaining -= take;
    }
    // end mark + the last partial dword, byte by byte
    const u32 lastWord = stage[0] | (1u << cur);
    const u32 done = 4u * wordsOut, nbytes = done + ((cur + 1u + 7u) >> 3);
    if (lane < 4u && done + lane < nbytes) out[done + lane] = (u8)(lastWord >> (8u * lane));
    lz_lds_sync();
    return nbytes;
}

// Lizard_writeStream for a Huffman candidate (lizard_compress.c:141-183) at `op`:
//   accepted  -> LE24 n, LE24 c, c bytes of huff0 payload;  returns 6 + c and sets *huffed = 1
//   otherwise -> LE24 n, raw bytes;                          returns 3 + n and sets *huffed = 0
// ws = LZ_HUF_WS_WORDS words of LDS (may alias the parser's tag array). All lanes call.
LZ_DEV u32 lz_put_stream_huf(u8* op, const u8* stream, u32 n, u32* ws, u32* huffed LZ_HPROF_PARAM)
{
    const u32 lane = lz_lane();
    *huffed = 0;
    if (n <= 1024u) {                                          // lizard_compress.c:143
        if (lane == 0) { op[0] = (u8)n; op[1] = (u8)(n >> 8); op[2] = (u8)(n >> 16); }
        lz_converge();
        for (u32 i = lane * 4u; i < (n & ~3u); i += 256u) lz_st32(op + 3 + i, lz_ld32(stream + i));
        for (u32 i = (n & ~3u) + lane; i < n; i += 64u) op[3 + i] = stream[i];
        return 3u + n;
    }
    u32* count = ws + LZ_HUF_WS_COUNT;
    u16* ctab = (u16*)(ws + LZ_HUF_WS_CTAB);
    u8* payload = op + 6;

    LZ_HPROF(14);
    // ---- histogram (FSE_count_wksp, fse_compress.c:431) ----
    for (u32 i = lane; i < 256u; i += 64u) count[i] = 0;
    lz_lds_sync();
    {
        const u32 n4 = n & ~3u, n16 = n & ~1023u;
        for (u32 i = lane * 4u; i < n16; i += 1024u) {             // four loads in flight: a step is one memory trip long
            u32 w[4];
            #pragma unroll
            for (u32 k = 0; k < 4u; k++) w[k] = lz_ld32(stream + i + k * 256u);
            #pragma unroll
            for (u32 k = 0; k < 4u; k++) {
                lz_lds_atomic_add(&count[w[k] & 255u], 1u);
                lz_lds_atomic_add(&count[(w[k] >> 8) & 255u], 1u);
                lz_lds_atomic_add(&count[(w[k] >> 16) & 255u], 1u);
                lz_lds_atomic_add(&count[w[k] >> 24], 1u);
            }
        }
        for (u32 i = n16 + lane * 4u; i < n4; i += 256u) {
            const u32 w = lz_ld32(stream + i);
            lz_lds_atomic_add(&count[w & 255u], 1u);
            lz_lds_atomic_add(&count[(w >> 8) & 255u], 1u);
            lz_lds_atomic_add(&count[(w >> 16) & 255u], 1u);
            lz_lds_atomic_add(&count[w >> 24], 1u);
        }
        for (u32 i = n4 + lane; i < n; i += 64u) lz_lds_atomic_add(&count[stream[i]], 1u);
    }
    lz_lds_sync();
    u32 c4[4];                                                 // my four symbols: 4*lane .. 4*lane+3
    for (u32 k = 0; k < 4u; k++) c4[k] = count[lane * 4u + k];
    u32 myMax = c4[0], myTop = 0; bool any = c4[0] != 0;
    for (u32 k = 1; k < 4u; k++) { if (c4[k] > myMax) myMax = c4[k]; if (c4[k]) { myTop = k; any = true; } }
    LZ_HPROF(8);                                               // histogram
    const u32 largest = lz_wave_reduce_max(myMax);
    const u64 anyMask = lz_ballot(any);
    const u32 topLane = 63u - lz_clz64(anyMask);               // n > 0 -> some symbol is present
    const u32 maxSym = topLane * 4u + lz_readlane(myTop, topLane);

    bool accept = false;                                       // uniform
    u32 csize = 0;                                             // uniform
    if (largest == n) {                                        // single symbol: RLE, 1 byte (huf_compress.c:544)
        if (lane == 0) payload[0] = stream[0];
        lz_converge();
        csize = 1; accept = true;
    } else if (largest > (n >> 7) + 1u) {                      // :545 otherwise "not compressible"
        // ---- sort: leaf[rank] = (count, symbol), descending count, ties ascending symbol (:305-325) ----
        // One compare per pair: key = count << 8 | 255 - symbol is distinct per symbol and orders them exactly so; rank = keys above mine.
        // (Symbols past maxSym have no count: their keys lie below every key of a symbol up to maxSym, so whole groups of four are read.)
        u32 key[4], rank[4] = { 0, 0, 0, 0 };
        for (u32 k = 0; k < 4u; k++) key[k] = (c4[k] << 8) | (255u - (lane * 4u + k));
        for (u32 k = 0; k < 4u; k++) count[lane * 4u + k] = key[k];
        lz_lds_sync();
        for (u32 t = 0; t <= maxSym; t += 4u) {
            u32 kt[4];
            #pragma unroll
            for (u32 j = 0; j < 4u; j++) kt[j] = count[t + j];
            #pragma unroll
            for (u32 j = 0; j < 4u; j++)
                for (u32 k = 0; k < 4u; k++) rank[k] += kt[j] > key[k] ? 1u : 0u;
        }
        const u32 nonNull = lz_popc64(lz_ballot(c4[0] != 0)) + lz_popc64(lz_ballot(c4[1] != 0))
                          + lz_popc64(lz_ballot(c4[2] != 0)) + lz_popc64(lz_ballot(c4[3] != 0)) - 1u;   // last rank with a count (>= 1 here)
        lz_lds_sync();                                         // every lane has read the counts
        for (u32 k = 0; k < 4u; k++) {
            const u32 s = lane * 4u + k;
            if (s <= maxSym) count[rank[k]] = (c4[k] << 8) | s;      // ranks of the symbols 0..maxSym are a permutation of 0..maxSym
        }
        lz_lds_sync();
        LzV256 leaf;                                           // by rank: count << 8 | symbol (0 past maxSym)
        leaf.r0 = lane <= maxSym ? count[lane] : 0u;
        leaf.r1 = 64u + lane <= maxSym ? count[64u + lane] : 0u;
        leaf.r2 = 128u + lane <= maxSym ? count[128u + lane] : 0u;
        leaf.r3 = 192u + lane <= maxSym ? count[192u + lane] : 0u;
        LZ_HPROF(9);                                           // rank sort
        // ---- tree (huf_compress.c:334-376): the two-queue merge.  Internal node k stands for the reference's huffNode[256 + k].
        // The serial loop only DECIDES: per node the sum of its two children and how many of them were leaves (0, 1, 2).  Both
        // queues are consumed in order — leaves from the smallest upwards, nodes in the order they were made — so that count is
        // all the tree there is: the parents of every leaf and node follow from prefix sums over it, lane-parallel, behind the loop
        // (round 5 tracked both parent tables inside the loop: twice the instructions and a dozen branches per node).  A step is
        // branch-free scalar code on the two heads of each queue (:357-360: a leaf is taken only when strictly smaller). ----
        const u32 kRoot = nonNull - 1u;
        LzV256 T;                                              // by node: leaves among its two children
        {
            const u32 BAR = 1u << 31, BIG = 1u << 30;          // :351 barrier below the leaves, :350 nodes not made yet
            LzV256 A, Bq;                                      // leaf counts in ASCENDING order (i = nonNull - rank; BAR behind them); node counts (BIG = not made yet)
            A.r0 = lane <= nonNull ? count[nonNull - lane] >> 8 : BAR;
            A.r1 = 64u + lane <= nonNull ? count[nonNull - 64u - lane] >> 8 : BAR;
            A.r2 = 128u + lane <= nonNull ? count[nonNull - 128u - lane] >> 8 : BAR;
            A.r3 = 192u + lane <= nonNull ? count[nonNull - 192u - lane] >> 8 : BAR;
            Bq.r0 = Bq.r1 = Bq.r2 = Bq.r3 = BIG;
            T.r0 = T.r1 = T.r2 = T.r3 = 0;
            u32 li = 0, ni = 0, nodeNb = 0;                    // next leaf, next node, the node being made
            // The 64 entries around each cursor ride in one register: leaves read (chunk cA; chunk 4 = the barrier behind 256 leaves),
            // nodes read (cR), nodes written (cW, live in Bw / Tw).  The loop runs in stretches that no cursor can leave its 64
            // entries in — a step moves a read cursor by two at most — so a step of a stretch has no cursor test in it: two compares,
            // their selects, one add, two v_writelane, four v_readlane.  Between stretches ONE step goes through the general accessors.
            u32 cA = 0, cR = 0, cW = 0;
            u32 Ac = A.r0, Br = BIG, Bw = BIG, Tw = 0;
            u32 s0 = lz_readlane(Ac, 0u), s1 = lz_readlane(Ac, 1u), n0 = BIG, n1 = BIG;     // (nonNull >= 1: at least two leaves)
#define LZ_MERGE_DECIDE()  const bool c1 = s0 < n0;                                                                    \
                           const u32 x = c1 ? s1 : s0, y = c1 ? n0 : n1;  /* what the second pick chooses between */ \
                           const bool c2 = x < y;                                                                      \
                           const u32 sum = (c1 ? s0 : n0) + (c2 ? x : y);                                              \
                           const u32 dl = (c1 ? 1u : 0u) + (c2 ? 1u : 0u)
            while (nodeNb <= kRoot) {
                u32 la = li & 63u, na = ni & 63u, wa = nodeNb & 63u;
                u32 safe = kRoot + 1u - nodeNb;
                { const u32 m = 64u - wa; safe = m < safe ? m : safe; }
                { const u32 m = la <= 62u ? (62u - la) >> 1 : 0u; safe = m < safe ? m : safe; }
                { const u32 m = na <= 62u ? (62u - na) >> 1 : 0u; safe = m < safe ? m : safe; }
                safe = lz_uniform(safe);                       // (a scalar trip count: the min chain above tends to end up in a VGPR)
                if (safe) {
                    const bool live = cR == cW;                // the node heads sit in the register that is being written
                    const u32 la0 = la, na0 = na;
                    for (u32 it = 0; it < safe; it++) {
                        LZ_MERGE_DECIDE();
                        lz_writelane2(Bw, sum, Tw, dl, wa);
                        wa++; la += dl; na += 2u - dl;
                        const u32 from = live ? Bw : Br;
                        s0 = lz_readlane(Ac, la); s1 = lz_readlane(Ac, la + 1u);
                        n0 = lz_readlane(from, na); n1 = lz_readlane(from, na + 1u);
                    }
                    li += la - la0; ni += na - na0; nodeNb += safe;
                } else {
                    LZ_MERGE_DECIDE();
                    lz_writelane2(Bw, sum, Tw, dl, wa);
                    li += dl; ni += 2u - dl; nodeNb++;
                }
                if ((nodeNb & 63u) == 0u) { Bq.putReg(cW, Bw); T.putReg(cW, Tw); cW++; Bw = BIG; Tw = 0; }
                if (!safe || (nodeNb & 63u) == 0u) {
                    // re-seat the registers around the cursors; the heads through the general accessors
                    cA = li >> 6; Ac = cA < 4u ? A.reg(cA) : BAR;
                    cR = ni >> 6; Br = Bq.reg(cR);
#define LZ_GETB(i) ((i) >= 256u ? BIG : ((i) >> 6) == cW ? lz_readlane(Bw, (i) & 63u) : Bq.get(i))
                    s0 = li < 256u ? A.get(li) : BAR; s1 = li + 1u < 256u ? A.get(li + 1u) : BAR;
                    n0 = LZ_GETB(ni); n1 = LZ_GETB(ni + 1u);
#undef LZ_GETB
                }
            }
#undef LZ_MERGE_DECIDE
            T.putReg(cW, Tw);
        }
        LZ_HPROF_F(0);                                         // merge
        // ---- parents (lane-parallel): node k = 64 j + lane took dl leaves and 2 - dl nodes; with Lb = leaves taken by the nodes before
        // it, those are the leaves i = Lb, Lb + 1 (rank nonNull - i) and the nodes 2 k - Lb, ...  Node parents go to count[] in the form
        // the pointer jumping wants (distance 1 << 8 | parent; the root points at itself), leaf parents to a u16 table behind it. ----
        u16* const parLeaf = (u16*)(ws + LZ_HUF_WS_COUNT + 256u);      // [256], by rank
        lz_lds_sync();                                         // the sorted leaves in count[] have been read
        for (u32 j = 0; j < 4u; j++) count[64u * j + lane] = kRoot;   // (entries past the root are never looked at)
        lz_lds_sync();
        {
            const u32 t[4] = { T.r0, T.r1, T.r2, T.r3 };
            u32 carry = 0;
            for (u32 j = 0; j < 4u; j++) {
                const u32 exc = lz_wave_scan_excl_add(t[j]);
                const u32 k = 64u * j + lane, lb = carry + exc, nb = 2u * k - lb;
                carry += lz_readlane(exc + t[j], 63u);
                if (k <= kRoot) {
                    if (t[j] >= 1u) parLeaf[nonNull - lb] = (u16)k;
                    if (t[j] == 2u) parLeaf[nonNull - lb - 1u] = (u16)k;
                    if (t[j] <= 1u) count[nb] = (1u << 8) | k;
                    if (t[j] == 0u) count[nb + 1u] = (1u << 8) | k;
                }
            }
        }
        // depths (:371-376).  Node depths by pointer jumping in LDS: word k = distance << 8 | ancestor, the root points at itself;
        // after r rounds every node has jumped 2^r levels, eight rounds cover any tree of 256 leaves.  Then the leaves.
        lz_lds_sync();
        {
            u32 w[4];
            for (u32 j = 0; j < 4u; j++) w[j] = count[64u * j + lane];
            lz_lds_sync();
            for (u32 round = 0; round < 8u; round++) {
                u32 nw[4];
                for (u32 j = 0; j < 4u; j++) { const u32 up = count[w[j] & 255u]; nw[j] = ((w[j] >> 8) + (up >> 8)) << 8 | (up & 255u); }
                lz_lds_sync();
                for (u32 j = 0; j < 4u; j++) { w[j] = nw[j]; count[64u * j + lane] = w[j]; }
                lz_lds_sync();
            }
        }
        LzV256 bits;                                           // by rank: code length (0 past nonNull)
        bits.r0 = lane <= nonNull ? (count[parLeaf[lane]] >> 8) + 1u : 0u;
        bits.r1 = 64u + lane <= nonNull ? (count[parLeaf[64u + lane]] >> 8) + 1u : 0u;
        bits.r2 = 128u + lane <= nonNull ? (count[parLeaf[128u + lane]] >> 8) + 1u : 0u;
        bits.r3 = 192u + lane <= nonNull ? (count[parLeaf[192u + lane]] >> 8) + 1u : 0u;
        LZ_HPROF_F(1);                                         // depths
        u32 huffLog = lz_fse_optimal_tablelog(LZ_HUF_DEFAULTLOG, n, maxSym, 1u);   // HUF_optimalTableLog :66
        huffLog = lz_huf_set_max_height(leaf, bits, nonNull, huffLog);            // :379
        LZ_HPROF_F(2);                                         // depth limit
        // ---- code length per symbol: through LDS by rank ----
        lz_lds_sync();
        count[lane] = bits.r0; count[64u + lane] = bits.r1; count[128u + lane] = bits.r2; count[192u + lane] = bits.r3;
        lz_lds_sync();
        u32 nb4[4];
        for (u32 k = 0; k < 4u; k++) nb4[k] = (lane * 4u + k <= maxSym && c4[k] != 0) ? count[rank[k]] : 0u;
        // ---- canonical values (:381-398): symbols of one length are numbered in symbol order, the longest codes get the
        // smallest values.  Counting "symbols before me with my length" = exclusive prefix sums of 12 counters, packed three
        // to a dword (10 bits each; there are at most 256 symbols): four wave scans instead of a pass per symbol. ----
        u32 pk[4] = { 0, 0, 0, 0 };
        for (u32 k = 0; k < 4u; k++) if (nb4[k]) { const u32 b_ = nb4[k] - 1u; pk[b_ / 3u] += 1u << (10u * (b_ % 3u)); }
        u32 pre[4], tot[4];
        for (u32 j = 0; j < 4u; j++) { const u32 inc = lz_wave_scan_excl_add(pk[j]); pre[j] = inc; tot[j] = lz_readlane(inc + pk[j], 63u); }
        LzV64 valPerRank; valPerRank.r = 0;
        LzV64 wcount; wcount.r = 0;                             // weight histogram of the symbols 0..maxSym-1 (HUF_compressWeights)
        {
            u32 mn = 0;
            for (u32 i = huffLog; i > 0; i--) {                // :388-393
                const u32 b_ = i - 1u;
                const u32 cnt = (tot[b_ / 3u] >> (10u * (b_ % 3u))) & 1023u;     // nbPerRank[i]
                valPerRank.set(i, mn);
                mn = (mn + cnt) >> 1;
                wcount.set(huffLog + 1u - i, cnt);
            }
        }
        u32 wt4 = 0;                                           // my four weights, one per byte (symbol maxSym itself has none)
        {
            u32 seen[4] = { 0, 0, 0, 0 };                      // my own earlier symbols, same packing
            for (u32 k = 0; k < 4u; k++) {
                const u32 s = lane * 4u + k, nb = nb4[k];
                const u32 first = lz_shfl(valPerRank.r, nb);   // valPerRank[nb]; outside any branch: a cross-lane read needs its source lane active
                u32 code = 0;
                if (nb) {
                    const u32 b_ = nb - 1u, sh = 10u * (b_ % 3u);
                    const u32 before = ((pre[b_ / 3u] + seen[b_ / 3u]) >> sh) & 1023u;
                    code = ((first + before) & 0xFFFu) | (nb << 12);
                    seen[b_ / 3u] += 1u << sh;
                    if (s < maxSym) wt4 |= (huffLog + 1u - nb) << (8u * k);
                }
                ctab[s] = (u16)code;
            }
        }
        LZ_HPROF_F(3);                                         // canonical codes
        // ---- HUF_writeCTable (:132-165): weights of symbols 0..maxSym-1, FSE-compressed or as nibbles ----
        u32 hdr = 0;                                           // header size, 0 = reference error -> raw
        {
            // the symbol maxSym is in wcount but has no weight in the header: take it out; weight 0 = absent symbols below maxSym
            const u32 nbLast = (lz_readlane(nb4[0] | (nb4[1] << 8) | (nb4[2] << 16) | (nb4[3] << 24), maxSym >> 2) >> (8u * (maxSym & 3u))) & 255u;
            const u32 wLast = huffLog + 1u - nbLast;
            wcount.set(wLast, wcount.get(wLast) - 1u);
            wcount.set(0, maxSym - nonNull);
            LzBitV b;
            u32* const hlds = ws + LZ_HUF_WS_COUNT;                // the histogram's words and the leaf-parent table are free by now
            const u32 h = lz_huf_compress_weights(b, wt4, maxSym, wcount, hlds);
            if (h == 0xFFFFFFFFu) hdr = 0;
            else if (h > 1u && h < maxSym / 2u) {
                if (lane == 0) payload[0] = (u8)h;
                lz_converge();
                const u32 hw = hlds[256u + lane];
                for (u32 k = 0; k < 4u; k++) if (4u * lane + k < h) payload[1u + 4u * lane + k] = (u8)(hw >> (8u * k));
                hdr = h + 1u;
            }
            else if (maxSym > 128u) hdr = 0;                   // :158 ERROR(GENERIC)
            else {
                if (lane == 0) payload[0] = (u8)(128u + (maxSym - 1u));
                lz_converge();
                // two weights per byte: symbols 2i, 2i+1 -> my four symbols give bytes 2*lane and 2*lane+1 (weights past maxSym-1 are 0)
                const u32 nBytes = (maxSym + 1u) / 2u;
                if (2u * lane < nBytes) payload[1u + 2u * lane] = (u8)(((wt4 & 15u) << 4) | ((wt4 >> 8) & 15u));
                if (2u * lane + 1u < nBytes) payload[2u + 2u * lane] = (u8)((((wt4 >> 16) & 15u) << 4) | ((wt4 >> 24) & 15u));
                hdr = nBytes + 1u;
            }
        }
        lz_lds_sync();                                         // ctab visible to every lane
        LZ_HPROF(10);                                          // tree / codes / header
        if (hdr != 0 && hdr + 12u < n) {                       // :556
            // ---- accept or not (huf_compress.c:570, lizard_compress.c:157) BEFORE anything is encoded.  The four bitstreams hold
            // B = sum over the symbols of count x code length bits — known from the histogram — plus an end mark each, rounded up to
            // bytes each: their bytes lie in [ceil((B + 4) / 8), that + 3].  Only a stream whose fate the three bytes decide (rare)
            // has its segments' code lengths summed up (huf_compress.c:473-513); round 5 did that for every stream (6 % of a
            // level-30 consumer's time).  Accepted streams are packed one after the other, each where the one before ended. ----
            const u32 seg = (n + 3u) / 4u;
            const u32 B = lz_wave_reduce_add(c4[0] * nb4[0] + c4[1] * nb4[1] + c4[2] * nb4[2] + c4[3] * nb4[3]);
            const u32 lo = hdr + 6u + ((B + 4u + 7u) >> 3), hi = lo + 3u;
            bool pack = hi < n - 1u && hi + hi / 8u + 512u < n;
            if (!pack && lo < n - 1u && lo + lo / 8u + 512u < n) {
                LZ_STAT(60);
                u32 tot = hdr + 6u;
                for (u32 k = 0; k < 4u; k++) {
                    const u32 a = k * seg, b = (k == 3u) ? n : (k + 1u) * seg;
                    u32 bits = 0;                                  // 4 symbols per lane per step (vector-memory cost is per instruction)
                    const u32 n4 = (b - a) & ~3u, n16 = (b - a) & ~1023u;
                    for (u32 i = lane * 4u; i < n16; i += 1024u) {     // four loads in flight
                        u32 w[4];
                        #pragma unroll
                        for (u32 j = 0; j < 4u; j++) w[j] = lz_ld32(stream + a + i + j * 256u);
                        #pragma unroll
                        for (u32 j = 0; j < 4u; j++)
                            bits += ((u32)ctab[w[j] & 255u] >> 12) + ((u32)ctab[(w[j] >> 8) & 255u] >> 12)
                                  + ((u32)ctab[(w[j] >> 16) & 255u] >> 12) + ((u32)ctab[w[j] >> 24] >> 12);
                    }
                    for (u32 i = n16 + lane * 4u; i < n4; i += 256u) {
                        const u32 w = lz_ld32(stream + a + i);
                        bits += ((u32)ctab[w & 255u] >> 12) + ((u32)ctab[(w >> 8) & 255u] >> 12)
                              + ((u32)ctab[(w >> 16) & 255u] >> 12) + ((u32)ctab[w >> 24] >> 12);
                    }
                    for (u32 i = a + n4 + lane; i < b; i += 64u) bits += (u32)ctab[stream[i]] >> 12;
                    bits = lz_wave_reduce_add(bits);
                    tot += (bits + 1u + 7u) >> 3;
                }
                pack = tot < n - 1u && tot + tot / 8u + 512u < n;
            }
            LZ_HPROF(11);                                      // accept or not
            if (pack) {
                u8* q = payload + hdr + 6u;
                u32 segBytes[4];
                for (u32 k = 0; k < 4u; k++) {
                    const u32 a = k * seg, b = (k == 3u) ? n : (k + 1u) * seg;
                    segBytes[k] = lz_huf_pack_segment(stream, a, b, q, ctab, ws + LZ_HUF_WS_STAGE);
                    q += segBytes[k];
                }
                if (lane == 0) {
                    lz_st16(payload + hdr, segBytes[0]); lz_st16(payload + hdr + 2, segBytes[1]); lz_st16(payload + hdr + 4, segBytes[2]);
                }
                lz_converge();
                csize = hdr + 6u + segBytes[0] + segBytes[1] + segBytes[2] + segBytes[3]; accept = true;
                LZ_HPROF(12);                                  // bit packing
            }
        }
    }
    if (accept && csize + csize / 8u + 512u < n) {             // lizard_compress.c:157 (also for the RLE byte)
        if (lane == 0) { op[0] = (u8)n; op[1] = (u8)(n >> 8); op[2] = (u8)(n >> 16); op[3] = (u8)csize; op[4] = (u8)(csize >> 8); op[5] = (u8)(csize >> 16); }
        lz_converge();
        *huffed = 1;
        return 6u + csize;
    }
    if (lane == 0) { op[0] = (u8)n; op[1] = (u8)(n >> 8); op[2] = (u8)(n >> 16); }
    lz_converge();
    for (u32 i = lane * 4u; i < (n & ~3u); i += 256u) lz_st32(op + 3 + i, lz_ld32(stream + i));
    for (u32 i = (n & ~3u) + lane; i < n; i += 64u) op[3 + i] = stream[i];
    return 3u + n;
}
