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

#define LZ_HUF_MAXBITS     12u    // HUF_TABLELOG_MAX, huf.h:118
#define LZ_HUF_DEFAULTLOG  11u    // HUF_TABLELOG_DEFAULT, huf.h:119

// LDS workspace (u32 words). The node area and the packer's staging ring alias (disjoint in time).
#define LZ_HUF_WS_COUNT    0u                          // u32 count[256]; after the sort: weights (bytes 0..255) and,
#define LZ_HUF_WS_FSE      96u                         //   from word 96 on, 160 words of FSE tables for the weight header
#define LZ_HUF_WS_CTAB     256u                        // u16 ctab[256]: val | nbBits << 12
#define LZ_HUF_WS_NODECNT  384u                        // u32 nodeCount[1 + 512] (slot 0 = barrier node -1)
#define LZ_HUF_WS_PARENT   (LZ_HUF_WS_NODECNT + 514u)  // u16 parent[512]
#define LZ_HUF_WS_BYTE     (LZ_HUF_WS_PARENT + 256u)   // u8 byte[256] (leaves only)
#define LZ_HUF_WS_NBITS    (LZ_HUF_WS_BYTE + 64u)      // u8 nbBits[512]
#define LZ_HUF_WS_WORDS    (LZ_HUF_WS_NBITS + 128u)    // 1346 words = 5384 B
#define LZ_HUF_WS_STAGE    LZ_HUF_WS_NODECNT           // packer staging ring (>= 100 words), aliases nodes
#define LZ_HUF_STAGE_WORDS 104u

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

// ---- single-lane LSB-first bit writer into global memory (weight header only; tiny) ----
struct LzBitW { u8* p; u32 pos; u64 acc; u32 nb; };
LZ_DEV void lz_bw_add(LzBitW& b, u32 v, u32 n)
{
    if (n == 0) return;
    b.acc |= (u64)(v & (u32)((1ull << n) - 1ull)) << b.nb;
    b.nb += n;
    while (b.nb >= 8u) { b.p[b.pos++] = (u8)b.acc; b.acc >>= 8; b.nb -= 8u; }
}
LZ_DEV u32 lz_bw_finish(LzBitW& b) { if (b.nb) { b.p[b.pos++] = (u8)b.acc; b.acc = 0; b.nb = 0; } return b.pos; }

// HUF_setMaxHeight, huf_compress.c:223-297 (lane 0 only). nodeCnt/nbits index the sorted leaves 0..last.
LZ_DEV u32 lz_huf_set_max_height(const u32* nodeCnt, u8* nbits, u32 lastNonNull, u32 maxNbBits)
{
    const u32 largestBits = nbits[lastNonNull];
    if (largestBits <= maxNbBits) return largestBits;
    const u32 noSymbol = 0xF0F0F0F0u;
    int totalCost = 0;
    const u32 baseCost = 1u << (largestBits - maxNbBits);
    int n = (int)lastNonNull;
    u32 rankLast[LZ_HUF_MAXBITS + 2];
    while (nbits[n] > maxNbBits) {
        totalCost += (int)(baseCost - (1u << (largestBits - nbits[n])));
        nbits[n] = (u8)maxNbBits;
        n--;
    }
    while (n >= 0 && nbits[n] == maxNbBits) n--;          // the reference stops on the barrier node (nbBits 0)
    totalCost >>= (largestBits - maxNbBits);
    for (u32 i = 0; i < LZ_HUF_MAXBITS + 2; i++) rankLast[i] = noSymbol;
    {
        u32 cur = maxNbBits;
        for (int pos = n; pos >= 0; pos--) {
            if (nbits[pos] >= cur) continue;
            cur = nbits[pos];
            rankLast[maxNbBits - cur] = (u32)pos;
        }
    }
    while (totalCost > 0) {
        u32 dec = lz_highbit((u32)totalCost) + 1u;
        for (; dec > 1u; dec--) {
            const u32 highPos = rankLast[dec], lowPos = rankLast[dec - 1u];
            if (highPos == noSymbol) continue;
            if (lowPos == noSymbol) break;
            if (nodeCnt[highPos] <= 2u * nodeCnt[lowPos]) break;
        }
        while (dec <= LZ_HUF_MAXBITS && rankLast[dec] == noSymbol) dec++;
        totalCost -= 1 << (dec - 1u);
        if (rankLast[dec - 1u] == noSymbol) rankLast[dec - 1u] = rankLast[dec];
        nbits[rankLast[dec]]++;
        if (rankLast[dec] == 0) rankLast[dec] = noSymbol;
        else {
            rankLast[dec]--;
            if (nbits[rankLast[dec]] != maxNbBits - dec) rankLast[dec] = noSymbol;
        }
    }
    while (totalCost < 0) {
        if (rankLast[1] == noSymbol) {
            while (nbits[n] == maxNbBits) n--;
            nbits[n + 1]--;
            rankLast[1] = (u32)(n + 1);
            totalCost++;
            continue;
        }
        nbits[rankLast[1] + 1u]--;
        rankLast[1]++;
        totalCost++;
    }
    return maxNbBits;
}

// FSE_normalizeCount + FSE_normalizeM2, fse_compress.c:507-641 (lane 0). Returns false on the
// reference's error returns.
LZ_DEV bool lz_fse_normalize(short* norm, u32 tableLog, const u32* count, u32 total, u32 maxSym)
{
    const u32 rtb[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    const u64 scale = 62u - tableLog, step = ((u64)1 << 62) / total, vStep = 1ULL << (scale - 20u);
    int still = 1 << tableLog;
    u32 largest = 0;
    short largestP = 0;
    const u32 lowThreshold = total >> tableLog;
    {
        const u32 minBitsSrc = lz_highbit(total - 1u) + 1u, minBitsSym = lz_highbit(maxSym) + 2u;
        if (tableLog < (minBitsSrc < minBitsSym ? minBitsSrc : minBitsSym)) return false;
    }
    for (u32 s = 0; s <= maxSym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = -1; still--; }
        else {
            short proba = (short)((count[s] * step) >> scale);
            if (proba < 8) {
                u32 r = 0;                                    // rtb[proba] without a dynamically indexed private array
                for (int k = 0; k < 8; k++) if (k == proba) r = rtb[k];
                const u64 restToBeat = vStep * r;
                proba += (count[s] * step) - ((u64)proba << scale) > restToBeat;
            }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm[s] = proba;
            still -= proba;
        }
    }
    if (-still < (norm[largest] >> 1)) { norm[largest] += (short)still; return true; }
    // FSE_normalizeM2
    {
        u32 distributed = 0, toDistribute;
        u64 tot = total;
        u32 lowOne = (u32)((tot * 3u) >> (tableLog + 1u));
        for (u32 s = 0; s <= maxSym; s++) {
            if (count[s] == 0) { norm[s] = 0; continue; }
            if (count[s] <= lowThreshold) { norm[s] = -1; distributed++; tot -= count[s]; continue; }
            if (count[s] <= lowOne) { norm[s] = 1; distributed++; tot -= count[s]; continue; }
            norm[s] = -2;
        }
        toDistribute = (1u << tableLog) - distributed;
        if ((tot / toDistribute) > lowOne) {
            lowOne = (u32)((tot * 3u) / (toDistribute * 2u));
            for (u32 s = 0; s <= maxSym; s++)
                if (norm[s] == -2 && count[s] <= lowOne) { norm[s] = 1; distributed++; tot -= count[s]; }
            toDistribute = (1u << tableLog) - distributed;
        }
        if (distributed == maxSym + 1u) {
            u32 maxV = 0, maxC = 0;
            for (u32 s = 0; s <= maxSym; s++) if (count[s] > maxC) { maxV = s; maxC = count[s]; }
            norm[maxV] += (short)toDistribute;
            return true;
        }
        const u64 vStepLog = 62u - tableLog, mid = (1ULL << (vStepLog - 1u)) - 1u;
        const u64 rStep = ((((u64)1 << vStepLog) * toDistribute) + mid) / tot;
        u64 tmpTotal = mid;
        for (u32 s = 0; s <= maxSym; s++) if (norm[s] == -2) {
            const u64 end = tmpTotal + (count[s] * rStep);
            const u32 weight = (u32)(end >> vStepLog) - (u32)(tmpTotal >> vStepLog);
            if (weight < 1u) return false;
            norm[s] = (short)weight;
            tmpTotal = end;
        }
    }
    return true;
}

// HUF_compressWeights, huf_compress.c:81-121 (lane 0): FSE-compress wt[0..wtSize) to dst.
// Returns compressed size, 0 = not compressible, 1 = all equal, 0xFFFFFFFF = reference error.
// `fse` = 160 words of LDS scratch, wt lives in LDS too.
LZ_DEV u32 lz_huf_compress_weights(u8* dst, const u8* wt, u32 wtSize, u32* fse)
{
    u32*  count      = fse;                  // [13]
    short* norm      = (short*)(fse + 16);   // [13]
    u32*  cumul      = fse + 24;             // [15]
    u32*  dBits      = fse + 40;             // [13]
    int*  dFind      = (int*)(fse + 56);     // [13]
    u16*  stateTable = (u16*)(fse + 72);     // [64]
    u8*   tableSym   = (u8*)(fse + 104);     // [64]
    if (wtSize <= 1u) return 0;
    for (u32 s = 0; s <= LZ_HUF_MAXBITS; s++) count[s] = 0;
    for (u32 s = 0; s < wtSize; s++) count[wt[s]]++;
    u32 maxSym = LZ_HUF_MAXBITS, maxCount = 0;
    while (!count[maxSym]) maxSym--;
    for (u32 s = 0; s <= maxSym; s++) if (count[s] > maxCount) maxCount = count[s];
    if (maxCount == wtSize) return 1;
    if (maxCount == 1u) return 0;
    const u32 tableLog = lz_fse_optimal_tablelog(6u, wtSize, maxSym, 2u);
    if (!lz_fse_normalize(norm, tableLog, count, wtSize, maxSym)) return 0xFFFFFFFFu;
    u32 pos;
    {   // FSE_writeNCount_generic, fse_compress.c:204-289
        LzBitW b; b.p = dst; b.pos = 0; b.acc = 0; b.nb = 0;
        int nbBits = (int)tableLog + 1, remaining = (1 << tableLog) + 1, threshold = 1 << tableLog;
        u32 charnum = 0;
        bool previous0 = false;
        lz_bw_add(b, tableLog - 5u, 4u);
        while (remaining > 1) {
            if (previous0) {
                u32 start = charnum;
                while (!norm[charnum]) charnum++;
                while (charnum >= start + 24u) { start += 24u; lz_bw_add(b, 0xFFFFu, 16u); }
                while (charnum >= start + 3u) { start += 3u; lz_bw_add(b, 3u, 2u); }
                lz_bw_add(b, charnum - start, 2u);
            }
            int cnt = norm[charnum++];
            const int mx = (2 * threshold - 1) - remaining;
            remaining -= cnt < 0 ? -cnt : cnt;
            cnt++;
            if (cnt >= threshold) cnt += mx;
            lz_bw_add(b, (u32)cnt, (u32)(nbBits - (cnt < mx)));
            previous0 = (cnt == 1);
            if (remaining < 1) return 0xFFFFFFFFu;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (charnum > maxSym + 1u) return 0xFFFFFFFFu;
        pos = lz_bw_finish(b);
    }
    {   // FSE_buildCTable_wksp, fse_compress.c:103-182
        const u32 tableSize = 1u << tableLog, mask = tableSize - 1u, step = (tableSize >> 1) + (tableSize >> 3) + 3u;
        u32 high = tableSize - 1u, position = 0, total = 0;
        cumul[0] = 0;
        for (u32 u = 1; u <= maxSym + 1u; u++) {
            if (norm[u - 1u] == -1) { cumul[u] = cumul[u - 1u] + 1u; tableSym[high--] = (u8)(u - 1u); }
            else cumul[u] = cumul[u - 1u] + (u32)norm[u - 1u];
        }
        for (u32 s = 0; s <= maxSym; s++)
            for (int k = 0; k < norm[s]; k++) {
                tableSym[position] = (u8)s;
                position = (position + step) & mask;
                while (position > high) position = (position + step) & mask;
            }
        if (position != 0) return 0xFFFFFFFFu;
        for (u32 u = 0; u < tableSize; u++) { const u32 sy = tableSym[u]; stateTable[cumul[sy]++] = (u16)(tableSize + u); }
        for (u32 s = 0; s <= maxSym; s++) {
            if (norm[s] == 0) { dBits[s] = 0; dFind[s] = 0; }
            else if (norm[s] == -1 || norm[s] == 1) { dBits[s] = (tableLog << 16) - (1u << tableLog); dFind[s] = (int)total - 1; total++; }
            else {
                const u32 maxBitsOut = tableLog - lz_highbit((u32)norm[s] - 1u);
                const u32 minStatePlus = (u32)norm[s] << maxBitsOut;
                dBits[s] = (maxBitsOut << 16) - minStatePlus; dFind[s] = (int)total - norm[s]; total += (u32)norm[s];
            }
        }
    }
    // FSE_compress_usingCTable_generic, fse_compress.c:701-758 (+ fse.h:525-564): two interleaved states
    if (wtSize <= 2u) return 0;
    LzBitW b; b.p = dst + pos; b.pos = 0; b.acc = 0; b.nb = 0;
    long long st1, st2;
    u32 i = wtSize;
#define LZ_FSE_INIT2(S, sym) do { const u32 nbo_ = (dBits[sym] + (1u << 15)) >> 16; const long long v_ = ((long long)nbo_ << 16) - dBits[sym]; \
                                  (S) = stateTable[(v_ >> nbo_) + dFind[sym]]; } while (0)
#define LZ_FSE_ENC(S, sym) do { const u32 nbo_ = (u32)(((S) + dBits[sym]) >> 16); lz_bw_add(b, (u32)(S), nbo_); \
                                (S) = stateTable[((S) >> nbo_) + dFind[sym]]; } while (0)
    if (wtSize & 1u) { LZ_FSE_INIT2(st1, wt[i - 1u]); LZ_FSE_INIT2(st2, wt[i - 2u]); i -= 3u; LZ_FSE_ENC(st1, wt[i]); }
    else             { LZ_FSE_INIT2(st2, wt[i - 1u]); LZ_FSE_INIT2(st1, wt[i - 2u]); i -= 2u; }
    bool useSt2 = true;
    while (i > 0) { i--; if (useSt2) LZ_FSE_ENC(st2, wt[i]); else LZ_FSE_ENC(st1, wt[i]); useSt2 = !useSt2; }
    lz_bw_add(b, (u32)st2, tableLog);
    lz_bw_add(b, (u32)st1, tableLog);
    lz_bw_add(b, 1u, 1u);                                     // BIT_closeCStream end mark
#undef LZ_FSE_INIT2
#undef LZ_FSE_ENC
    return pos + lz_bw_finish(b);
}

// One 1X bitstream (huf_compress.c:427-470): symbols src[a..b) appended LAST -> FIRST, LSB first, then a
// single '1'.  nbytes = ceil((bits+1)/8) is known from the size pass.  All lanes call; `stage` is LDS.
LZ_DEV void lz_huf_pack_segment(const u8* src, u32 a, u32 b, u8* out, u32 nbytes, const u16* ctab, u32* stage)
{
    const u32 lane = lz_lane();
    for (u32 i = lane; i < LZ_HUF_STAGE_WORDS; i += 64u) stage[i] = 0;
    lz_lds_sync();
    u32 cur = 0;            // uniform: bits pending in stage[0] (< 32)
    u32 wordsOut = 0;       // uniform: dwords already stored to `out`
    u32 remaining = b - a;  // uniform: symbols not yet appended; next to append is src[a + remaining - 1]
    while (remaining > 0) {
        // lane takes up to 4 symbols: src[hi-3 .. hi], appended in the order hi, hi-1, hi-2, hi-3
        const u32 take = remaining < 256u ? remaining : 256u;
        const u32 first = lane * 4u;                      // index (in append order) of my first symbol in this step
        u64 acc = 0; u32 len = 0;
        if (first < take) {
            const u32 cnt = take - first < 4u ? take - first : 4u;
            const u32 hi = a + remaining - 1u - first;
            if (cnt == 4u) {                                  // one dword load, four independent table lookups
                const u32 w4 = lz_ld32(src + hi - 3u);
                const u32 e0 = ctab[w4 >> 24], e1 = ctab[(w4 >> 16) & 255u], e2 = ctab[(w4 >> 8) & 255u], e3 = ctab[w4 & 255u];
                acc = (u64)(e0 & 0xFFFu);               len = e0 >> 12;
                acc |= (u64)(e1 & 0xFFFu) << len;       len += e1 >> 12;
                acc |= (u64)(e2 & 0xFFFu) << len;       len += e2 >> 12;
                acc |= (u64)(e3 & 0xFFFu) << len;       len += e3 >> 12;
            } else {
                for (u32 j = 0; j < cnt; j++) {
                    const u32 e = ctab[src[hi - j]];
                    acc |= (u64)(e & 0xFFFu) << len;
                    len += e >> 12;
                }
            }
        }
        const u32 off = lz_wave_scan_excl_add(len);
        const u32 total = lz_readlane(off + len, 63u);
        if (len) {
            const u32 pos = cur + off, w = pos >> 5, sh = pos & 31u;
            const u32 lo = (u32)acc, hi32 = (u32)(acc >> 32);
            const u32 w0 = lo << sh;
            const u32 w1 = sh ? ((lo >> (32u - sh)) | (hi32 << sh)) : hi32;
            const u32 w2 = sh ? (hi32 >> (32u - sh)) : 0u;
            if (w0) lz_lds_atomic_or(&stage[w], w0);
            if (w1) lz_lds_atomic_or(&stage[w + 1u], w1);
            if (w2) lz_lds_atomic_or(&stage[w + 2u], w2);
        }
        lz_lds_sync();
        const u32 T = cur + total, full = T >> 5;
        // store the completed dwords (<= 97 per step), coalesced
        for (u32 i = lane; i < full; i += 64u) lz_st32(out + 4u * (wordsOut + i), stage[i]);
        const u32 carry = stage[full];
        lz_lds_sync();
        for (u32 i = lane; i < LZ_HUF_STAGE_WORDS; i += 64u) stage[i] = (i == 0) ? carry : 0u;
        lz_lds_sync();
        wordsOut += full; cur = T & 31u; remaining -= take;
    }
    // end mark + the last partial dword, byte by byte
    const u32 lastWord = stage[0] | (1u << cur);
    const u32 done = 4u * wordsOut;
    if (lane < 4u && done + lane < nbytes) out[done + lane] = (u8)(lastWord >> (8u * lane));
    lz_lds_sync();
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
    u32* nodeCnt = ws + LZ_HUF_WS_NODECNT + 1;                 // nodeCnt[-1] = barrier
    u16* parent = (u16*)(ws + LZ_HUF_WS_PARENT);
    u8* nbyte = (u8*)(ws + LZ_HUF_WS_BYTE);
    u8* nbits = (u8*)(ws + LZ_HUF_WS_NBITS);
    u32* fse = ws + LZ_HUF_WS_FSE;
    u8* payload = op + 6;

    LZ_HPROF(14);
    // ---- histogram (FSE_count_wksp, fse_compress.c:431) ----
    for (u32 i = lane; i < 256u; i += 64u) count[i] = 0;
    lz_lds_sync();
    {
        const u32 n4 = n & ~3u;
        for (u32 i = lane * 4u; i < n4; i += 256u) {
            const u32 w = lz_ld32(stream + i);
            lz_lds_atomic_add(&count[w & 255u], 1u);
            lz_lds_atomic_add(&count[(w >> 8) & 255u], 1u);
            lz_lds_atomic_add(&count[(w >> 16) & 255u], 1u);
            lz_lds_atomic_add(&count[w >> 24], 1u);
        }
        for (u32 i = n4 + lane; i < n; i += 64u) lz_lds_atomic_add(&count[stream[i]], 1u);
    }
    lz_lds_sync();
    u32 c4[4];
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
        // ---- sort: node[rank] = (count, symbol), descending count, ties ascending symbol (:305-325) ----
        u32 rank[4] = { 0, 0, 0, 0 };
        for (u32 t = 0; t <= maxSym; t++) {
            const u32 ct = count[t];
            for (u32 k = 0; k < 4u; k++) {
                const u32 s = lane * 4u + k;
                rank[k] += (ct > c4[k] || (ct == c4[k] && t < s)) ? 1u : 0u;
            }
        }
        lz_lds_sync();
        for (u32 i = lane; i < 514u; i += 64u) ws[LZ_HUF_WS_NODECNT + i] = 0;
        for (u32 i = lane; i < 256u; i += 64u) ((u32*)parent)[i] = 0;
        for (u32 i = lane; i < 128u; i += 64u) { if (i < 64u) ((u32*)nbyte)[i] = 0; ((u32*)nbits)[i] = 0; }
        lz_lds_sync();
        for (u32 k = 0; k < 4u; k++) {
            const u32 s = lane * 4u + k;
            if (s <= maxSym) { nodeCnt[rank[k]] = c4[k]; nbyte[rank[k]] = (u8)s; }
        }
        lz_lds_sync();
        LZ_HPROF(9);                                           // rank sort
        // ---- lane 0: tree, depth limit, canonical codes, weight header ----
        u32 hdr = 0;                                           // header size, 0 = reference error -> raw (lane 0's value,
                                                               // broadcast through LDS: fse[159])
        u32 huffLog = lz_fse_optimal_tablelog(LZ_HUF_DEFAULTLOG, n, maxSym, 1u);   // HUF_optimalTableLog :66
        if (lane == 0) {
            const u32 START = 256u;                            // STARTNODE, :333
            u32 nonNull = maxSym;
            while (nodeCnt[nonNull] == 0) nonNull--;
            int lowS = (int)nonNull, lowN = (int)START;
            u32 nodeNb = START;
            const u32 nodeRoot = nodeNb + (u32)lowS - 1u;
            nodeCnt[nodeNb] = nodeCnt[lowS] + nodeCnt[lowS - 1];
            parent[lowS] = parent[lowS - 1] = (u16)nodeNb;
            nodeNb++; lowS -= 2;
            for (u32 i = nodeNb; i <= nodeRoot; i++) nodeCnt[i] = 1u << 30;
            nodeCnt[-1] = 1u << 31;                            // barrier, :351
            while (nodeNb <= nodeRoot) {                       // :353-369 (ties -> internal node)
                const int n1 = (nodeCnt[lowS] < nodeCnt[lowN]) ? lowS-- : lowN++;
                const int n2 = (nodeCnt[lowS] < nodeCnt[lowN]) ? lowS-- : lowN++;
                nodeCnt[nodeNb] = nodeCnt[n1] + nodeCnt[n2];
                parent[n1] = parent[n2] = (u16)nodeNb;
                nodeNb++;
            }
            nbits[nodeRoot] = 0;                               // :371-376
            for (u32 i = nodeRoot - 1u; i >= START; i--) nbits[i] = (u8)(nbits[parent[i]] + 1u);
            for (u32 i = 0; i <= nonNull; i++) nbits[i] = (u8)(nbits[parent[i]] + 1u);
            huffLog = lz_huf_set_max_height(nodeCnt, nbits, nonNull, huffLog);   // :379
            // canonical values (:381-398) -> ctab[symbol] = val | nbBits << 12  (val < 2^nbBits <= 2^12)
            u32 nbPerRank[LZ_HUF_MAXBITS + 1], valPerRank[LZ_HUF_MAXBITS + 1];
            for (u32 i = 0; i <= LZ_HUF_MAXBITS; i++) { nbPerRank[i] = 0; valPerRank[i] = 0; }
            for (u32 i = 0; i <= nonNull; i++) nbPerRank[nbits[i]]++;
            { u32 mn = 0; for (u32 i = huffLog; i > 0; i--) { valPerRank[i] = mn; mn = (mn + nbPerRank[i]) & 0xFFFFu; mn >>= 1; } }
            for (u32 i = 0; i < 256u; i++) ctab[i] = 0;
            for (u32 i = 0; i <= maxSym; i++) ctab[nbyte[i]] = (u16)((u32)nbits[i] << 12);
            for (u32 s = 0; s <= maxSym; s++) { const u32 nb = ctab[s] >> 12; ctab[s] = (u16)(ctab[s] | ((valPerRank[nb]++) & 0xFFFu)); }
            // HUF_writeCTable (:132-165): weights of symbols 0..maxSym-1 into count[] (free now), as bytes
            u8* wt = (u8*)count;
            for (u32 s = 0; s < maxSym; s++) { const u32 nb = ctab[s] >> 12; wt[s] = nb ? (u8)(huffLog + 1u - nb) : 0; }
            const u32 h = lz_huf_compress_weights(payload + 1, wt, maxSym, fse);
            if (h == 0xFFFFFFFFu) hdr = 0;
            else if (h > 1u && h < maxSym / 2u) { payload[0] = (u8)h; hdr = h + 1u; }
            else if (maxSym > 128u) hdr = 0;                   // :158 ERROR(GENERIC)
            else {
                payload[0] = (u8)(128u + (maxSym - 1u));
                wt[maxSym] = 0;
                for (u32 s = 0; s < maxSym; s += 2u) payload[s / 2u + 1u] = (u8)((wt[s] << 4) + wt[s + 1u]);
                hdr = (maxSym + 1u) / 2u + 1u;
            }
            fse[159] = hdr;
        }
        lz_lds_sync();
        hdr = lz_uniform(fse[159]);
        LZ_HPROF(10);                                          // lane-0 tree / codes / header
        if (hdr != 0 && hdr + 12u < n) {                       // :556
            // ---- exact stream sizes: sum of code lengths per segment (huf_compress.c:473-513) ----
            const u32 seg = (n + 3u) / 4u;
            u32 segBytes[4], tot = hdr + 6u;
            for (u32 k = 0; k < 4u; k++) {
                const u32 a = k * seg, b = (k == 3u) ? n : (k + 1u) * seg;
                u32 bits = 0;                                  // 4 symbols per lane per step (vector-memory cost is per instruction)
                const u32 n4 = (b - a) & ~3u;
                for (u32 i = lane * 4u; i < n4; i += 256u) {
                    const u32 w = lz_ld32(stream + a + i);
                    bits += ((u32)ctab[w & 255u] >> 12) + ((u32)ctab[(w >> 8) & 255u] >> 12)
                          + ((u32)ctab[(w >> 16) & 255u] >> 12) + ((u32)ctab[w >> 24] >> 12);
                }
                for (u32 i = a + n4 + lane; i < b; i += 64u) bits += (u32)ctab[stream[i]] >> 12;
                bits = lz_wave_reduce_add(bits);
                segBytes[k] = (bits + 1u + 7u) >> 3;
                tot += segBytes[k];
            }
            LZ_HPROF(11);                                      // exact sizes
            if (tot < n - 1u && tot + tot / 8u + 512u < n) {   // :570 and lizard_compress.c:157
                if (lane == 0) {
                    lz_st16(payload + hdr, segBytes[0]); lz_st16(payload + hdr + 2, segBytes[1]); lz_st16(payload + hdr + 4, segBytes[2]);
                }
                lz_converge();
                u8* q = payload + hdr + 6u;
                for (u32 k = 0; k < 4u; k++) {
                    const u32 a = k * seg, b = (k == 3u) ? n : (k + 1u) * seg;
                    lz_huf_pack_segment(stream, a, b, q, segBytes[k], ctab, ws + LZ_HUF_WS_STAGE);
                    q += segBytes[k];
                }
                csize = tot; accept = true;
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
