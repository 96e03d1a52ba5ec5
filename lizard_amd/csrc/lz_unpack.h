// lz_unpack.h — device-side Lizard block DEcompression for ONE wavefront per block (gfx950), SURVEY.md §8f rank 4.
//
// Decodes what reference lib/lizard_decompress.c:115-265 decodes (independent blocks: no dictionary, no prefix):
//   * the sub-block container: raw sub-blocks, five streams each stored raw or huff0-compressed (Lizard_readStream :72-112)
//   * huff0 streams: weight header as nibbles or FSE-compressed (HUF_readStats, entropy_common.c:170-231; FSE_readNCount
//     :71-160; FSE_buildDTable / FSE_decompress_usingDTable, fse_decompress.c:88-155,214-268), single-symbol decoding table
//     (huf_decompress.c HUF_readDTableX2), 1 or 4 bitstreams read from their last byte backwards (bitstream.h)
//   * fastLZ4 codewords (lizard_decompress_lz4.h:7-160) and LIZv1 codewords (lizard_decompress_liz.h:14-222)
// The result is the original block, so parity here means: equal to the input the block was compressed from and equal to
// what the reference decoder returns (tests/test_decompress.py), and every malformed input the fuzz tests produce is
// refused (LZD_ERR) without reading outside the compressed block or writing outside the output slot.
//
// Mapping onto the wave.  Blocks are independent: one wave per block, thousands in flight.  Inside a block the sequence
// chain is serial (every copy starts where the previous one ended), so the wave walks the tokens with wave-uniform scalar
// code and does the byte moving with all 64 lanes: 64 tokens and a 256-byte window of the literals stream (where the
// length escapes and fastLZ4 offsets live) are fetched per load, literal runs are copied 8 bytes per lane, matches too when
// they do not overlap their own output, byte-wise modulo the offset when they do.  The four huff0 bitstreams of a stream
// are independent: four lanes decode them side by side from a decoding table in LDS that all 64 lanes fill.
#pragma once
#include "lz_wave.h"

#define LZD_ERR 0xFFFFFFFFu

// LDS workspace of one decoding wave (u32 words)
#define LZD_WS_DT     0u          // u16[4096]  huff0 decoding table: symbol | nbBits << 8         (8 KiB)
#define LZD_WS_WT     2048u       // u8[256]    weights
#define LZD_WS_FSE    2112u       // u32[64]    FSE decoding table of the weight header: symbol | nbBits << 8 | newState << 16
#define LZD_WS_MISC   2176u       // u32[32]    lane-0 results handed to the wave
#define LZD_WS_WORDS  2208u

LZ_DEV u32 lzd_le24(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16); }

// ---- backward bit reader over bytes [p, p+n): bit k of the stream is bit (k & 7) of byte k >> 3; `pos` counts the
// unread bits (they are bits pos-1 .. 0); reading below bit 0 yields zeros (bitstream.h:331-347 look/skip semantics) ----
struct LzdBits { const u8* p; u32 n; int pos; };
LZ_DEV bool lzd_bits_init(LzdBits& b, const u8* p, u32 n)        // bitstream.h:270-319: the last byte holds the end mark
{
    b.p = p; b.n = n; b.pos = 0;
    if (n == 0) return false;
    const u32 last = p[n - 1];
    if (last == 0) return false;
    b.pos = (int)(8u * (n - 1u) + (31u - (u32)__builtin_clz(last)));
    return true;
}
LZ_DEV u32 lzd_bits_peek(const LzdBits& b, u32 nb)              // the next nb (<= 16) bits, most significant first in the stream
{
    if (nb == 0) return 0;
    const int lo = b.pos - (int)nb;                              // lowest bit wanted (may be negative)
    const int byte0 = lo >= 0 ? lo >> 3 : 0;
    u64 w = 0;
    for (u32 k = 0; k < 4u; k++) { const u32 i = (u32)byte0 + k; if (i < b.n) w |= (u64)b.p[i] << (8u * k); }
    const u32 v = lo >= 0 ? (u32)(w >> (lo & 7)) : (u32)(w << (u32)(-lo));
    return v & ((1u << nb) - 1u);
}

// ---- weight header of a huff0 stream (HUF_readStats).  Lane 0 only.  Returns the header size or LZD_ERR. ----
LZ_DEV u32 lzd_read_weights_lane0(const u8* src, u32 srcSize, u8* wt, u32* fse, u32* misc, u32& nbSym, u32& tableLog)
{
    if (srcSize == 0) return LZD_ERR;
    u32 iSize = src[0], oSize;
    if (iSize >= 128u) {                                         // weights as nibbles (entropy_common.c:183-193)
        oSize = iSize - 127u;
        iSize = (oSize + 1u) / 2u;
        if (iSize + 1u > srcSize || oSize >= 256u) return LZD_ERR;
        for (u32 n = 0; n < oSize; n += 2u) { wt[n] = src[1u + n / 2u] >> 4; wt[n + 1u] = src[1u + n / 2u] & 15u; }
    } else {                                                     // FSE-compressed weights (:194-199, FSE_decompress_wksp with maxLog 6)
        if (iSize + 1u > srcSize) return LZD_ERR;
        const u8* ip = src + 1;
        short* norm = (short*)(misc + 8);                        // [16] in LDS: weights are < 12, a header naming more symbols is corrupt
        u16* symbolNext = (u16*)(misc + 16);                     // [16]
        u32 maxSV = 15, tl = 0, hdr = 0;
        {   // FSE_readNCount, entropy_common.c:71-160, as a plain LSB-first bit reader over ip[0..iSize)
            if (iSize < 4u) return LZD_ERR;
            u32 bp = 0;                                          // bit position
            auto take = [&](u32 nbits) -> u32 {                  // bits [bp, bp+nbits), zeros past the end
                u64 w = 0;
                for (u32 k = 0; k < 5u; k++) { const u32 i = (bp >> 3) + k; if (i < iSize) w |= (u64)ip[i] << (8u * k); }
                return (u32)(w >> (bp & 7u)) & ((1u << nbits) - 1u);
            };
            int nbBits = (int)take(4) + 5; bp += 4;
            if (nbBits > 15) return LZD_ERR;
            tl = (u32)nbBits;
            int remaining = (1 << nbBits) + 1, threshold = 1 << nbBits;
            nbBits++;
            u32 charnum = 0;
            bool previous0 = false;
            while (remaining > 1 && charnum <= maxSV) {
                if (previous0) {
                    u32 n0 = charnum;
                    while (take(16) == 0xFFFFu) { n0 += 24u; bp += 16u; if (bp > 8u * iSize + 32u) return LZD_ERR; }
                    while (take(2) == 3u) { n0 += 3u; bp += 2u; }
                    n0 += take(2); bp += 2u;
                    if (n0 > maxSV) return LZD_ERR;
                    while (charnum < n0) norm[charnum++] = 0;
                }
                {
                    const int mx = (2 * threshold - 1) - remaining;
                    int count;
                    if ((int)take((u32)nbBits - 1u) < mx) { count = (int)take((u32)nbBits - 1u); bp += (u32)nbBits - 1u; }
                    else { count = (int)take((u32)nbBits); if (count >= threshold) count -= mx; bp += (u32)nbBits; }
                    count--;
                    remaining -= count < 0 ? -count : count;
                    norm[charnum++] = (short)count;
                    previous0 = !count;
                    while (remaining < threshold) { nbBits--; threshold >>= 1; }
                }
                if (bp > 8u * iSize) return LZD_ERR;
            }
            if (remaining != 1) return LZD_ERR;
            maxSV = charnum - 1u;
            hdr = (bp + 7u) >> 3;
        }
        if (tl > 6u || hdr > iSize) return LZD_ERR;
        const u32 tableSize = 1u << tl;
        {   // FSE_buildDTable, fse_decompress.c:88-155
            u32 high = tableSize - 1u;
            for (u32 s = 0; s <= maxSV; s++) {
                if (norm[s] == -1) { fse[high--] = s; symbolNext[s] = 1; } else symbolNext[s] = (u16)norm[s];
            }
            const u32 mask = tableSize - 1u, step = (tableSize >> 1) + (tableSize >> 3) + 3u;
            u32 position = 0;
            for (u32 s = 0; s <= maxSV; s++)
                for (int i = 0; i < norm[s]; i++) {
                    fse[position] = s;
                    position = (position + step) & mask;
                    while (position > high) position = (position + step) & mask;
                }
            if (position != 0) return LZD_ERR;
            for (u32 u = 0; u < tableSize; u++) {
                const u32 sym = fse[u] & 255u;
                const u32 next = symbolNext[sym]++;
                const u32 nb = tl - (31u - (u32)__builtin_clz(next));
                fse[u] = sym | (nb << 8) | (((next << nb) - tableSize) << 16);
            }
        }
        {   // FSE_decompress_usingDTable, fse_decompress.c:214-268: two states, until the bitstream is used up
            LzdBits b;
            if (!lzd_bits_init(b, ip + hdr, iSize - hdr)) return LZD_ERR;
            u32 st1 = lzd_bits_peek(b, tl); b.pos -= (int)tl;
            u32 st2 = lzd_bits_peek(b, tl); b.pos -= (int)tl;
            if (b.pos < 0) return LZD_ERR;
            oSize = 0;
            for (;;) {
                if (oSize > 253u) return LZD_ERR;                // at most 255 weights, the last one is implied
                { const u32 e = fse[st1]; wt[oSize++] = (u8)e; const u32 nb = (e >> 8) & 255u; st1 = (e >> 16) + lzd_bits_peek(b, nb); b.pos -= (int)nb; }
                if (b.pos < 0) { wt[oSize++] = (u8)fse[st2]; break; }
                if (oSize > 253u) return LZD_ERR;
                { const u32 e = fse[st2]; wt[oSize++] = (u8)e; const u32 nb = (e >> 8) & 255u; st2 = (e >> 16) + lzd_bits_peek(b, nb); b.pos -= (int)nb; }
                if (b.pos < 0) { wt[oSize++] = (u8)fse[st1]; break; }
            }
        }
    }
    // weight statistics, the implied last weight (:201-226)
    u32 rank[13];
    for (u32 i = 0; i < 13u; i++) rank[i] = 0;
    u32 total = 0;
    for (u32 n = 0; n < oSize; n++) {
        if (wt[n] >= 12u) return LZD_ERR;
        u32 w = wt[n];
        for (u32 i = 0; i < 13u; i++) if (i == w) rank[i]++;   // (no dynamically indexed private array)
        total += (1u << w) >> 1;
    }
    if (total == 0) return LZD_ERR;
    tableLog = (31u - (u32)__builtin_clz(total)) + 1u;
    if (tableLog > 12u) return LZD_ERR;
    {
        const u32 rest = (1u << tableLog) - total;
        const u32 hb = 31u - (u32)__builtin_clz(rest);
        if ((1u << hb) != rest) return LZD_ERR;
        wt[oSize] = (u8)(hb + 1u);
        for (u32 i = 0; i < 13u; i++) if (i == hb + 1u) rank[i]++;
    }
    if (rank[1] < 2u || (rank[1] & 1u)) return LZD_ERR;
    nbSym = oSize + 1u;
    return iSize + 1u;
}

// ---- one huff0 stream (HUF_decompress, huf_decompress.c): cSrc[0..cSize) -> dst[0..n).  All lanes call.  false = corrupt ----
LZ_DEV bool lzd_huf_decompress(const u8* cSrc, u32 cSize, u8* dst, u32 n, u32* ws)
{
    const u32 lane = lz_lane();
    u16* dt = (u16*)(ws + LZD_WS_DT);
    u8* wt = (u8*)(ws + LZD_WS_WT);
    u32* fse = ws + LZD_WS_FSE;
    u32* misc = ws + LZD_WS_MISC;
    if (n == 0 || cSize > n) return false;
    if (cSize == n) { for (u32 i = lane; i < n; i += 64u) dst[i] = cSrc[i]; return true; }     // stored
    if (cSize == 1u) { const u8 v = cSrc[0]; for (u32 i = lane; i < n; i += 64u) dst[i] = v; return true; }   // one symbol
    // weights -> (nbSym, tableLog, header size), by lane 0, handed over through LDS
    lz_lds_sync();
    if (lane == 0) {
        u32 nbSym = 0, tl = 0;
        const u32 h = lzd_read_weights_lane0(cSrc, cSize, wt, fse, misc, nbSym, tl);
        misc[0] = h; misc[1] = nbSym; misc[2] = tl;
    }
    lz_converge();
    lz_lds_sync();
    const u32 hSize = lz_uniform(misc[0]), nbSym = lz_uniform(misc[1]), tableLog = lz_uniform(misc[2]);
    if (hSize == LZD_ERR) return false;
    // decoding table (HUF_readDTableX2): symbols of weight w take (1 << w) >> 1 consecutive cells, weights ascending, symbols in
    // order inside a weight.  "cells before me" = rank start of my weight + (same-weight symbols before me) * cell count:
    // the packed prefix sums of lz_huf.h in the other direction.
    {
        u32 w4[4], pk[4] = { 0, 0, 0, 0 };
        for (u32 k = 0; k < 4u; k++) { const u32 s = 4u * lane + k; w4[k] = s < nbSym ? wt[s] : 0u; if (w4[k]) { const u32 b_ = w4[k] - 1u; pk[b_ / 3u] += 1u << (10u * (b_ % 3u)); } }
        u32 pre[4], tot[4];
        for (u32 j = 0; j < 4u; j++) { const u32 ex = lz_wave_scan_excl_add(pk[j]); pre[j] = ex; tot[j] = lz_readlane(ex + pk[j], 63u); }
        u32 rankStart[13];
        { u32 next = 0; for (u32 w = 1; w <= 12u; w++) { const u32 b_ = w - 1u; rankStart[w] = next; next += ((tot[b_ / 3u] >> (10u * (b_ % 3u))) & 1023u) << (w - 1u); } rankStart[0] = 0;
          if (next != (1u << tableLog)) return false; }
        u32 seen[4] = { 0, 0, 0, 0 }, start[4], cells[4], ent[4];
        for (u32 k = 0; k < 4u; k++) {
            const u32 w = w4[k];
            start[k] = 0; cells[k] = 0; ent[k] = 0;
            if (w) {
                const u32 b_ = w - 1u, sh = 10u * (b_ % 3u);
                const u32 before = ((pre[b_ / 3u] + seen[b_ / 3u]) >> sh) & 1023u;
                u32 rs = 0;
                for (u32 i = 1; i <= 12u; i++) if (i == w) rs = rankStart[i];
                cells[k] = 1u << (w - 1u);
                start[k] = rs + before * cells[k];
                ent[k] = (4u * lane + k) | ((tableLog + 1u - w) << 8);
                seen[b_ / 3u] += 1u << sh;
            }
        }
        lz_lds_sync();
        for (u32 k = 0; k < 4u; k++)
            for (u32 l = 0; l < 64u; l++) {                      // symbol 4l + k: the wave fills its cells together
                const u32 c = lz_readlane(cells[k], l);
                if (!c) continue;
                const u32 st = lz_readlane(start[k], l), e = lz_readlane(ent[k], l);
                for (u32 i = lane; i < c; i += 64u) dt[st + i] = (u16)e;
            }
        lz_lds_sync();
    }
    // bitstreams: one (n < 256 is never produced by the encoder, but HUF_decompress accepts 1X only through its own entry) —
    // HUF_decompress always uses the 4-stream layout (huf_decompress.c HUF_decompress4X_*): 6-byte jump table, 4 segments
    const u8* ip = cSrc + hSize;
    const u32 rem = cSize - hSize;
    if (rem < 10u) return false;
    const u32 l1 = (u32)ip[0] | ((u32)ip[1] << 8), l2 = (u32)ip[2] | ((u32)ip[3] << 8), l3 = (u32)ip[4] | ((u32)ip[5] << 8);
    if (6u + l1 + l2 + l3 > rem) return false;
    const u32 l4 = rem - 6u - l1 - l2 - l3;
    const u32 seg = (n + 3u) / 4u;
    if (3u * seg > n) return false;
    bool bad = false;
    if (lane < 4u) {                                             // four independent backward bitstreams, one lane each
        const u32 off = lane == 0 ? 0u : lane == 1 ? l1 : lane == 2 ? l1 + l2 : l1 + l2 + l3;
        const u32 len = lane == 0 ? l1 : lane == 1 ? l2 : lane == 2 ? l3 : l4;
        const u32 cnt = lane == 3u ? n - 3u * seg : seg;
        u8* o = dst + lane * seg;
        // backward bit reader with a 64-bit register window: acc holds the next `avail` unread bits, left-aligned; refilled
        // four bytes at a time from the end of the segment towards its start; bits below the start read as zeros
        const u8* sp = ip + 6u + off;
        const u32 last = len ? sp[len - 1u] : 0u;
        if (!last) bad = true;
        else {
            const u32 hb = 31u - (u32)__builtin_clz(last);       // position of the end mark in the last byte
            int remaining = (int)(8u * (len - 1u) + hb);         // unread bits of the whole segment
            u64 acc = 0; u32 avail = 0, bytePos = len;
            u32 nextW = bytePos >= 4u ? lz_ld32(sp + bytePos - 4u) : 0u;     // the next refill word is always requested one refill ahead
            auto refill = [&]() {
                while (avail <= 32u && bytePos > 0u) {
                    if (bytePos >= 4u) {
                        acc |= (u64)nextW << (32u - avail); avail += 32u; bytePos -= 4u;
                        nextW = bytePos >= 4u ? lz_ld32(sp + bytePos - 4u) : 0u;
                    } else { bytePos--; acc |= (u64)sp[bytePos] << (56u - avail); avail += 8u; }
                }
            };
            refill();
            acc <<= (8u - hb); avail -= (8u - hb);               // the end mark and the padding above it
            const u32 shift = 64u - tableLog;
            u32 four = 0;                                        // four symbols per store
            for (u32 i = 0; i < cnt; i++) {
                refill();
                const u32 e = dt[(u32)(acc >> shift)];
                const u32 nb = e >> 8;
                four |= (e & 255u) << (8u * (i & 3u));
                if ((i & 3u) == 3u) { lz_st32(o + i - 3u, four); four = 0; }
                acc <<= nb; avail = avail > nb ? avail - nb : 0u;
                remaining -= (int)nb;
            }
            for (u32 i = cnt & ~3u; i < cnt; i++) o[i] = (u8)(four >> (8u * (i & 3u)));
            if (remaining != 0) bad = true;                      // BIT_endOfDStream: every bit used, none borrowed
        }
    }
    return lz_ballot(bad) == 0;
}

// ---- one stream of the container (Lizard_readStream, lizard_decompress.c:72-112) ----
// in[pos..end): returns the stream as (ptr, len), decoded into `stage` when it is huff0-compressed; advances pos.
LZ_DEV bool lzd_read_stream(bool huf, const u8* in, u32& pos, u32 end, u8* stage, u32 stageCap, const u8*& ptr, u32& len, u32* ws)
{
    if (!huf) {
        if (pos + 3u > end) return false;
        len = lzd_le24(in + pos);
        if (len > end - pos - 3u) return false;
        ptr = in + pos + 3u; pos += 3u + len;
        return true;
    }
    if (pos + 6u > end) return false;
    const u32 n = lzd_le24(in + pos), c = lzd_le24(in + pos + 3u);
    if (n > stageCap || c > end - pos - 6u) return false;
    lz_wave_sync();                                              // earlier readers of the staging area are done
    if (!lzd_huf_decompress(in + pos + 6u, c, stage, n, ws)) return false;
    lz_wave_sync();                                              // decoded bytes visible to every lane
    ptr = stage; len = n; pos += 6u + c;
    return true;
}

// Length escape (lizard_decompress_lz4.h:47-58): value and size of the escape at p (1, 3 or 4 bytes), read from a 32-bit window
LZ_DEV void lzd_len_ext(u32 w, u32& value, u32& size)
{
    const u32 b0 = w & 255u;
    if (b0 < 254u) { value = b0; size = 1u; }
    else if (b0 == 254u) { value = (w >> 8) & 0xFFFFu; size = 3u; }
    else { value = w >> 8; size = 4u; }
}

// eight bytes at byte offset sh (uniform, 0..248) of the 256-byte window a wave holds in `wv` (lane l = bytes 4l..4l+3)
LZ_DEV u64 lzd_win_u64(u32 wv, u32 sh)
{
    const u32 l0 = sh >> 2, r = 8u * (sh & 3u);
    const u64 lo = (u64)lz_readlane(wv, l0) | ((u64)lz_readlane(wv, l0 + 1u > 63u ? 63u : l0 + 1u) << 32);
    const u64 hi = lz_readlane(wv, l0 + 2u > 63u ? 63u : l0 + 2u);
    return r ? (lo >> r) | (hi << (64u - r)) : lo;
}

// The wave's view of the literals stream, where the length escapes (and, fastLZ4, the offsets) live: a 256-byte window, lane l
// = bytes 4l..4l+3 from `base`, reloaded only when a read leaves it.  Bytes from nl on read as zero (the stream's last dword is
// put together byte by byte: an escape may sit in the last 1..3 bytes of the stream, lizard_decompress_liz.h:142 only asks for
// literalsPtr <= iend - 1).
struct LzdWin { u32 wv; u32 base; bool valid; };
LZ_DEV u64 lzd_win_fetch(LzdWin& w, const u8* pl, u32 nl, u32 at)               // the 8 bytes at stream offset `at` (uniform)
{
    if (!w.valid || at < w.base || at + 8u > w.base + 256u) {
        const u32 wa = at + 4u * lz_lane();
        u32 v = 0;
        if (wa + 4u <= nl) v = lz_ld32(pl + wa);
        else for (u32 k = 0; k < 4u; k++) if (wa + k < nl) v |= (u32)pl[wa + k] << (8u * k);
        w.wv = v; w.base = at; w.valid = true;
    }
    return lzd_win_u64(w.wv, at - w.base);
}

// wave-wide copy inside global memory, n uniform; src and dst do not overlap in a way that matters (src + n <= dst or src >= dst + n
// or the caller uses lzd_copy_match)
LZ_DEV void lzd_copy(u8* dst, const u8* src, u32 n)
{
    const u32 lane = lz_lane();
    const u32 n8 = n & ~7u;
    for (u32 i = lane * 8u; i < n8; i += 512u) lz_st64(dst + i, lz_ld64(src + i));
    for (u32 i = n8 + lane; i < n; i += 64u) dst[i] = src[i];
}
// match copy: out[op + i] = out[op - off + i] with LZ semantics (the source may run into the bytes being written)
LZ_DEV void lzd_copy_match(u8* out, u32 op, u32 off, u32 n)
{
    if (off >= n) { lzd_copy(out + op, out + op - off, n); return; }
    const u32 lane = lz_lane();
    const u8* base = out + op - off;
    for (u32 i = lane; i < n; i += 64u) out[op + i] = base[i % off];     // the pattern of the last `off` bytes repeats
}

// ---- one block.  in[0..inSize) -> out[0..outCap).  stage: 4 x (128 KiB + 32) bytes of global scratch; ws: LZD_WS_WORDS of LDS.
// Returns the decoded size (uniform) or LZD_ERR. ----
#define LZD_STAGE_BYTES (131072u + 32u)
LZ_DEV u32 lz_decompress_block(const u8* in, u32 inSize, u8* out, u32 outCap, u8* stage, u32* ws)
{
    const u32 lane = lz_lane();
    if (inSize < 1u) return 0;                                   // lizard_decompress.c:139
    const u32 level = in[0];
    if (level < 10u || level > 49u) return LZD_ERR;              // :143
    const bool lz4 = (level >= 10u && level <= 19u) || (level >= 30u && level <= 39u);   // decompressType, lizard_common.h:234-284
    u32 pos = 1u, op = 0;
    while (pos < inSize) {                                       // :161
        const u32 res = in[pos++];
        if (res == 128u) {                                       // LIZARD_FLAG_UNCOMPRESSED, :164-180
            if (pos + 3u > inSize) return LZD_ERR;
            const u32 length = lzd_le24(in + pos);
            pos += 3u;
            if (length > inSize - pos || length > outCap - op) return LZD_ERR;
            lzd_copy(out + op, in + pos, length);
            op += length; pos += length;
            continue;
        }
        if (res & 16u) return LZD_ERR;                           // LIZARD_FLAG_LEN is never produced, :182-184
        if (pos + 15u > inSize) return LZD_ERR;                  // :186
        {   // the `len` stream: stored raw, unused by the codewords
            const u32 l = lzd_le24(in + pos);
            if (l > inSize - pos - 3u) return LZD_ERR;
            pos += 3u + l;
        }
        const u8 *p16, *p24, *pf, *pl;
        u32 n16, n24, nf, nl;
        if (!lzd_read_stream(res & 4u, in, pos, inSize, stage + 2u * LZD_STAGE_BYTES, LZD_STAGE_BYTES, p16, n16, ws)) return LZD_ERR;
        if (!lzd_read_stream(res & 8u, in, pos, inSize, stage + 3u * LZD_STAGE_BYTES, LZD_STAGE_BYTES, p24, n24, ws)) return LZD_ERR;
        if (!lzd_read_stream(res & 2u, in, pos, inSize, stage + 1u * LZD_STAGE_BYTES, LZD_STAGE_BYTES, pf, nf, ws)) return LZD_ERR;
        if (!lzd_read_stream(res & 1u, in, pos, inSize, stage, LZD_STAGE_BYTES, pl, nl, ws)) return LZD_ERR;
        // ---- sequences ----
        u32 lp = 0, o16 = 0, o24 = 0;                            // uniform cursors into literals / off16 / off24
        u32 last_off = 0;                                        // LIZv1 repeat offset; the encoder never opens a sub-block with a repeat
        LzdWin win; win.wv = 0; win.base = 0; win.valid = false; // (a token without an escape — most LIZv1 tokens — never touches it)
        for (u32 tbase = 0; tbase < nf; tbase += 64u) {
            const u32 cnt = nf - tbase < 64u ? nf - tbase : 64u;
            const u32 tokv = lane < cnt ? pf[tbase + lane] : 0u;  // 64 tokens per load
            for (u32 t = 0; t < cnt; t++) {
                const u32 token = lz_readlane(tokv, t);
                u32 L, ml, off, used = 0;                        // used: bytes of the window consumed before the literals
                if (lz4) {                                       // lizard_decompress_lz4.h:41-110
                    L = token & 15u;
                    if (L == 15u) {
                        if (lp + 5u > nl) return LZD_ERR;        // :47
                        u32 v, s; lzd_len_ext((u32)lzd_win_fetch(win, pl, nl, lp), v, s);
                        L = v + 15u; used = s;
                    }
                    if (L > nl - lp - used || nl - lp - used - L < 2u) return LZD_ERR;
                    const u32 at = lp + used + L;                // offset, then the match-length escape
                    const u64 six = lzd_win_fetch(win, pl, nl, at);  // the offset and up to four escape bytes (zeros from nl on)
                    off = (u32)six & 0xFFFFu;
                    const u32 w2 = (u32)(six >> 16);
                    ml = token >> 4;
                    u32 mext = 0;
                    if (ml == 15u) {
                        if (at + 2u + 5u > nl) return LZD_ERR;   // :101
                        u32 v, s; lzd_len_ext(w2, v, s);
                        ml = v + 15u; mext = s;
                    }
                    ml += 4u;
                    if (L > outCap - op) return LZD_ERR;
                    lzd_copy(out + op, pl + lp + used, L);
                    op += L;
                    lp = at + 2u + mext;
                } else {                                         // lizard_decompress_liz.h:57-162
                    if (token >= 32u) {
                        L = token & 7u;
                        if (L == 7u) {
                            if (lp + 1u > nl) return LZD_ERR;
                            u32 v, s; lzd_len_ext((u32)lzd_win_fetch(win, pl, nl, lp), v, s);
                            L = v + 7u; used = s;
                        }
                        if (used > nl - lp || L > nl - lp - used) return LZD_ERR;
                        if (L > outCap - op) return LZD_ERR;
                        lzd_copy(out + op, pl + lp + used, L);
                        op += L; lp += used + L;
                        if ((token >> 7) == 0u) {                // new 16-bit offset (:96-110)
                            if (o16 + 2u > n16) return LZD_ERR;
                            last_off = lz_uniform((u32)p16[o16] | ((u32)p16[o16 + 1u] << 8));
                            o16 += 2u;
                        }
                        ml = (token >> 3) & 15u;
                        if (ml == 15u) {
                            if (lp + 1u > nl) return LZD_ERR;
                            u32 v, s; lzd_len_ext((u32)lzd_win_fetch(win, pl, nl, lp), v, s);
                            ml = v + 15u;
                            if (s > nl - lp) return LZD_ERR;
                            lp += s;
                        }
                    } else {
                        L = 0;
                        if (token < 31u) ml = token + 16u;       // :134-141
                        else {                                   // :142-161
                            if (lp + 1u > nl) return LZD_ERR;
                            u32 v, s; lzd_len_ext((u32)lzd_win_fetch(win, pl, nl, lp), v, s);
                            if (s > nl - lp) return LZD_ERR;
                            lp += s;
                            ml = v + 31u + 16u;
                        }
                        if (o24 + 3u > n24) return LZD_ERR;
                        last_off = lz_uniform(lzd_le24(p24 + o24));
                        o24 += 3u;
                    }
                    off = last_off;
                }
                if (ml) {                                        // (LIZv1: a literal-only token has match length 0)
                    if (off == 0u || off > op || ml > outCap - op) return LZD_ERR;   // lz4.h:93 / liz.h:165
                    lz_wave_sync();                              // the bytes the match copies from have been written by other lanes
                    lzd_copy_match(out, op, off, ml);
                    op += ml;
                }
            }
        }
        // last literals (lz4.h:141-148 / liz.h:204-211): the rest of the literals stream; the offset streams must be used up
        if (lp > nl || nl - lp > outCap - op) return LZD_ERR;
        lzd_copy(out + op, pl + lp, nl - lp);
        op += nl - lp;
        lz_wave_sync();                                          // staging areas are reused by the next sub-block
    }
    return op;
}
