/*
 * lizard_decode_host.c — the reference's one-block DEcompression ABI (lib/lizard_decompress.h) for host buffers, written
 * from the format: Lizard_decompress_safe, _safe_partial, _safe_continue, _safe_usingDict, _safe_forceExtDict and the
 * Lizard_streamDecode_t helpers (reference lib/lizard_decompress.c:267-371).
 *
 * Why a host decoder in a GPU library.  Decompression is not the path this library accelerates (north_star: the block
 * COMPRESSOR; SURVEY.md section 2 lists the decoder as out of scope, section 8f rank 4 as a later row, built as the batch
 * entries LizardGPU_decompressBlocks_*).  But section 8b lists these symbols among those a link-time replacement of
 * liblizard must export — the reference's CLI, fuzzer, frametest and fullbench call them for every block they verify — and
 * one block per call is a latency problem, not a throughput problem: one wave takes ~1 ms for what a host core does in
 * ~0.1 ms.  So the one-block entry points decode on the calling thread; callers with many independent blocks use
 * LizardGPU_decompressBlocks_host / _device (one wave per block, thousands in flight), and LizardF_decompress batches whole
 * independent-block frames through them.  The COMPRESSOR has no host path: Lizard_compress* return 0 loudly without a GPU.
 *
 * Format rules followed (every test below cites the line of the reference decoder that makes the same decision, so that
 * valid blocks decode identically and malformed blocks are refused; this file never reads outside src[0..srcSize) or the
 * dictionary, and never writes outside dst[0..maxDecompressedSize)):
 *   container         lib/lizard_decompress.c:115-265 (Lizard_decompress_generic), :72-112 (Lizard_readStream)
 *   fastLZ4 codewords lib/lizard_decompress_lz4.h:7-160
 *   LIZv1 codewords   lib/lizard_decompress_liz.h:14-220
 *   huff0 streams     lib/entropy/huf_decompress.c:832-845 (HUF_decompress), entropy_common.c:170-231 (HUF_readStats),
 *                     :71-160 (FSE_readNCount), fse_decompress.c:88-155,214-268, bitstream.h:270-347
 */
#include "../../include/lizard_amd.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define LZH_DICT_SIZE   ((size_t)1 << 24)          /* LIZARD_DICT_SIZE, lizard_common.h:76 */
#define LZH_WILD        16                         /* WILDCOPYLENGTH, lizard_common.h:77: the margins the reference's copies need */
#define LZH_STAGE       ((size_t)1 << 17)          /* LIZARD_HUF_BLOCK_SIZE: a decoded stream is at most one sub-block */

static u32 le16(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8); }
static u32 le24(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16); }

/* =============================== huff0 ===============================
 * One stream: [weight header][LE16 x 3 segment sizes][4 bitstreams].  Each bitstream is read from its last byte backwards;
 * the highest set bit of the last byte is the end mark (bitstream.h:270-319). */

typedef struct { const u8* p; size_t n; long pos; } BitsRev;   /* pos = number of unread bits; bit k = bit (k & 7) of byte k >> 3 */

static int bits_init(BitsRev* b, const u8* p, size_t n)
{
    u32 last;
    b->p = p; b->n = n; b->pos = 0;
    if (n == 0) return 0;
    last = p[n - 1];
    if (last == 0) return 0;
    b->pos = (long)(8 * (n - 1)) + (31 - __builtin_clz(last));
    return 1;
}
/* the next nb (<= 16) bits; bits below the start of the stream read as zeros (bitstream.h:331-347) */
static u32 bits_peek(const BitsRev* b, u32 nb)
{
    long lo;
    size_t byte0, k;
    u64 w = 0;
    if (nb == 0) return 0;
    lo = b->pos - (long)nb;
    byte0 = lo >= 0 ? (size_t)(lo >> 3) : 0;
    for (k = 0; k < 4; k++) if (byte0 + k < b->n) w |= (u64)b->p[byte0 + k] << (8 * k);
    return (u32)(lo >= 0 ? (w >> (lo & 7)) : (w << (u32)(-lo))) & ((1u << nb) - 1u);
}

/* Weight header (HUF_readStats, entropy_common.c:170-231).  wt[0..nbSym) and tableLog out; returns the header size, 0 = corrupt. */
static size_t read_weights(const u8* src, size_t srcSize, u8* wt, u32* nbSym, u32* tableLog)
{
    size_t iSize, oSize, n;
    u32 total = 0, rank1 = 0;
    if (srcSize == 0) return 0;
    iSize = src[0];
    if (iSize >= 128) {                                                       /* :183-193 weights as nibbles */
        oSize = iSize - 127;
        iSize = (oSize + 1) / 2;
        if (iSize + 1 > srcSize || oSize >= 256) return 0;
        for (n = 0; n < oSize; n += 2) { wt[n] = src[1 + n / 2] >> 4; wt[n + 1] = src[1 + n / 2] & 15; }
    } else {                                                                  /* :194-199 FSE-compressed weights, table log <= 6 */
        const u8* ip = src + 1;
        short norm[16];
        u16 symbolNext[16];
        u32 fse[64];
        u32 maxSV = 15, tl, hdr;
        if (iSize + 1 > srcSize) return 0;
        {   /* FSE_readNCount, entropy_common.c:71-160, as a plain LSB-first bit reader over ip[0..iSize) */
            size_t bp = 0;
            int nbBits, remaining, threshold, previous0 = 0;
            u32 charnum = 0;
#define TAKE(nbits, out) do { u64 w_ = 0; size_t k_; for (k_ = 0; k_ < 5; k_++) if ((bp >> 3) + k_ < iSize) w_ |= (u64)ip[(bp >> 3) + k_] << (8 * k_); \
                              (out) = (u32)(w_ >> (bp & 7)) & ((1u << (nbits)) - 1u); } while (0)
            u32 v;
            if (iSize < 4) return 0;
            TAKE(4, v); nbBits = (int)v + 5; bp += 4;
            if (nbBits > 15) return 0;
            tl = (u32)nbBits;
            remaining = (1 << nbBits) + 1; threshold = 1 << nbBits;
            nbBits++;
            while (remaining > 1 && charnum <= maxSV) {
                if (previous0) {
                    u32 n0 = charnum;
                    for (;;) { TAKE(16, v); if (v != 0xFFFFu) break; n0 += 24; bp += 16; if (bp > 8 * iSize + 32) return 0; }
                    for (;;) { TAKE(2, v); if (v != 3u) break; n0 += 3; bp += 2; }
                    n0 += v; bp += 2;
                    if (n0 > maxSV) return 0;
                    while (charnum < n0) norm[charnum++] = 0;
                }
                {
                    const int mx = (2 * threshold - 1) - remaining;
                    int count;
                    TAKE((u32)nbBits - 1u, v);
                    if ((int)v < mx) { count = (int)v; bp += (size_t)nbBits - 1; }
                    else { TAKE((u32)nbBits, v); count = (int)v; if (count >= threshold) count -= mx; bp += (size_t)nbBits; }
                    count--;
                    remaining -= count < 0 ? -count : count;
                    norm[charnum++] = (short)count;
                    previous0 = !count;
                    while (remaining < threshold) { nbBits--; threshold >>= 1; }
                }
                if (bp > 8 * iSize) return 0;
            }
#undef TAKE
            if (remaining != 1) return 0;
            maxSV = charnum - 1;
            hdr = (u32)((bp + 7) >> 3);
        }
        if (tl > 6 || hdr > iSize) return 0;
        {   /* FSE_buildDTable, fse_decompress.c:88-155 */
            const u32 tableSize = 1u << tl, mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
            u32 high = tableSize - 1, position = 0, s, u;
            int i;
            for (s = 0; s <= maxSV; s++) {
                if (norm[s] == -1) { fse[high--] = s; symbolNext[s] = 1; } else symbolNext[s] = (u16)norm[s];
            }
            for (s = 0; s <= maxSV; s++)
                for (i = 0; i < norm[s]; i++) {
                    fse[position] = s;
                    position = (position + step) & mask;
                    while (position > high) position = (position + step) & mask;
                }
            if (position != 0) return 0;
            for (u = 0; u < tableSize; u++) {
                const u32 sym = fse[u] & 255u;
                const u32 next = symbolNext[sym]++;
                const u32 nb = tl - (31u - (u32)__builtin_clz(next));
                fse[u] = sym | (nb << 8) | (((next << nb) - tableSize) << 16);
            }
        }
        {   /* FSE_decompress_usingDTable, fse_decompress.c:214-268: two states, until the bitstream is used up */
            BitsRev b;
            u32 st1, st2;
            if (!bits_init(&b, ip + hdr, iSize - hdr)) return 0;
            st1 = bits_peek(&b, tl); b.pos -= (long)tl;
            st2 = bits_peek(&b, tl); b.pos -= (long)tl;
            if (b.pos < 0) return 0;
            oSize = 0;
            for (;;) {
                u32 e, nb;
                if (oSize > 253) return 0;                                    /* at most 255 weights, the last one is implied */
                e = fse[st1]; wt[oSize++] = (u8)e; nb = (e >> 8) & 255u; st1 = (e >> 16) + bits_peek(&b, nb); b.pos -= (long)nb;
                if (b.pos < 0) { wt[oSize++] = (u8)fse[st2]; break; }
                if (oSize > 253) return 0;
                e = fse[st2]; wt[oSize++] = (u8)e; nb = (e >> 8) & 255u; st2 = (e >> 16) + bits_peek(&b, nb); b.pos -= (long)nb;
                if (b.pos < 0) { wt[oSize++] = (u8)fse[st1]; break; }
            }
        }
    }
    for (n = 0; n < oSize; n++) {                                             /* :201-226 the implied last weight */
        if (wt[n] >= 12) return 0;
        if (wt[n] == 1) rank1++;
        total += (1u << wt[n]) >> 1;
    }
    if (total == 0) return 0;
    *tableLog = (31u - (u32)__builtin_clz(total)) + 1u;
    if (*tableLog > 12) return 0;
    {
        const u32 rest = (1u << *tableLog) - total;
        const u32 hb = 31u - (u32)__builtin_clz(rest);
        if ((1u << hb) != rest) return 0;
        wt[oSize] = (u8)(hb + 1);
        if (hb + 1 == 1) rank1++;
    }
    if (rank1 < 2 || (rank1 & 1)) return 0;
    *nbSym = (u32)oSize + 1;
    return iSize + 1;
}

/* Backward bit reader of one huff0 bitstream (the construction of bitstream.h:270-347): `c` = the 8 bytes at ptr (little endian),
 * of which the top `used` bits are consumed; the next bits are (c << used) >> (64 - nb).  Streams shorter than 8 bytes sit in the
 * low bytes of c with ptr == start.  Used up exactly <=> ptr == start && used == 64. */
typedef struct { const u8* start; const u8* ptr; u64 c; u32 used; } HufRd;
static u64 hufrd_ld64(const u8* p) { u64 w; memcpy(&w, p, 8); return w; }      /* (little-endian hosts: x86-64) */
static int hufrd_init(HufRd* r, const u8* p, size_t n)
{
    u32 last, hb;
    size_t i;
    if (n == 0) return 0;
    last = p[n - 1];
    if (!last) return 0;                                                       /* no end mark */
    hb = 31u - (u32)__builtin_clz(last);
    r->start = p;
    if (n >= 8) { r->ptr = p + n - 8; r->c = hufrd_ld64(r->ptr); r->used = 8 - hb; }
    else {
        r->ptr = p; r->c = 0;
        for (i = 0; i < n; i++) r->c |= (u64)p[i] << (8 * i);
        r->used = (8 - hb) + (u32)(8 - n) * 8;
    }
    return 1;
}
static int hufrd_fast(const HufRd* r) { return r->ptr >= r->start + 8; }     /* a whole word can be refilled (then used <= 63 + ...: see callers) */
static void hufrd_reload_fast(HufRd* r) { r->ptr -= r->used >> 3; r->used &= 7; r->c = hufrd_ld64(r->ptr); }
static void hufrd_reload(HufRd* r)
{
    if (r->used > 64) r->used = 65;                                            /* read past the start: stays wrong, reported at the end */
    if (hufrd_fast(r)) { hufrd_reload_fast(r); return; }
    if (r->ptr == r->start) return;
    {
        u32 nb = r->used >> 3;
        if ((size_t)(r->ptr - r->start) < nb) nb = (u32)(r->ptr - r->start);
        r->ptr -= nb; r->used -= nb * 8; r->c = hufrd_ld64(r->ptr);
    }
}

/* dst[0..n) from cSrc[0..cSize) (HUF_decompress, huf_decompress.c:832-845).  1 = ok, 0 = corrupt. */
static int huf_decompress(u8* dst, size_t n, const u8* cSrc, size_t cSize)
{
    u16 dt[4096];                                                             /* symbol | nbBits << 8, single-symbol table (HUF_readDTableX2) */
    u8 wt[256];
    u32 nbSym = 0, tableLog = 0;
    size_t hSize;
    if (n == 0 || cSize > n) return 0;                                        /* :837-838 */
    if (cSize == n) { memcpy(dst, cSrc, n); return 1; }                       /* :839 stored */
    if (cSize == 1) { memset(dst, cSrc[0], n); return 1; }                    /* :840 one symbol */
    hSize = read_weights(cSrc, cSize, wt, &nbSym, &tableLog);
    if (!hSize) return 0;
    {   /* cells: weights ascending, symbols in order inside a weight; a symbol of weight w takes (1 << w) >> 1 cells */
        u32 rankStart[14], w, s, next = 0;
        u32 count[13];
        memset(count, 0, sizeof count);
        for (s = 0; s < nbSym; s++) count[wt[s]]++;
        for (w = 1; w <= 12; w++) { rankStart[w] = next; next += count[w] << (w - 1); }
        if (next != (1u << tableLog)) return 0;
        for (s = 0; s < nbSym; s++) {
            const u32 ww = wt[s];
            if (ww) {
                const u32 cells = 1u << (ww - 1), e = s | ((tableLog + 1 - ww) << 8);
                u32 i, st = rankStart[ww];
                for (i = 0; i < cells; i++) dt[st + i] = (u16)e;
                rankStart[ww] = st + cells;
            }
        }
    }
    {   /* HUF_decompress4X2_usingDTable: 6-byte jump table, 4 segments of ceil(n/4) symbols (the last one takes the rest).
         * The four bitstreams are independent: they are decoded side by side, four symbols per stream between two refills (a symbol
         * takes at most 12 bits), which is where the speed of the reference's decoder comes from as well (huf_decompress.c:222-262). */
        const u8* ip = cSrc + hSize;
        const size_t rem = cSize - hSize;
        size_t len[4], off[4], cnt[4], done[4], seg, quads, q;
        HufRd r[4];
        u8* o[4];
        const u32 shift = 64 - tableLog;
        int k;
        if (rem < 10) return 0;
        len[0] = le16(ip); len[1] = le16(ip + 2); len[2] = le16(ip + 4);
        if (6 + len[0] + len[1] + len[2] > rem) return 0;
        len[3] = rem - 6 - len[0] - len[1] - len[2];
        off[0] = 0; off[1] = len[0]; off[2] = len[0] + len[1]; off[3] = len[0] + len[1] + len[2];
        seg = (n + 3) / 4;
        if (3 * seg > n) return 0;
        cnt[0] = cnt[1] = cnt[2] = seg; cnt[3] = n - 3 * seg;
        for (k = 0; k < 4; k++) {
            if (!hufrd_init(&r[k], ip + 6 + off[k], len[k])) return 0;
            o[k] = dst + (size_t)k * seg; done[k] = 0;
        }
        /* side by side while every stream can refill a whole word and has four symbols to go */
        quads = cnt[3] / 4;
        for (q = 0; q < quads; q++) {
            int j;
            if (!(hufrd_fast(&r[0]) && hufrd_fast(&r[1]) && hufrd_fast(&r[2]) && hufrd_fast(&r[3]))) break;
            hufrd_reload_fast(&r[0]); hufrd_reload_fast(&r[1]); hufrd_reload_fast(&r[2]); hufrd_reload_fast(&r[3]);
            for (j = 0; j < 4; j++) {                                         /* used <= 7 + 4 * 12 < 64 throughout */
                const u32 e0 = dt[(u32)((r[0].c << r[0].used) >> shift)], e1 = dt[(u32)((r[1].c << r[1].used) >> shift)];
                const u32 e2 = dt[(u32)((r[2].c << r[2].used) >> shift)], e3 = dt[(u32)((r[3].c << r[3].used) >> shift)];
                o[0][4 * q + (size_t)j] = (u8)e0; r[0].used += e0 >> 8;
                o[1][4 * q + (size_t)j] = (u8)e1; r[1].used += e1 >> 8;
                o[2][4 * q + (size_t)j] = (u8)e2; r[2].used += e2 >> 8;
                o[3][4 * q + (size_t)j] = (u8)e3; r[3].used += e3 >> 8;
            }
        }
        done[0] = done[1] = done[2] = done[3] = 4 * q;
        /* the rest of every stream, one symbol at a time; below the start a stream reads as zeros, and a stream that is not used up
         * exactly is corrupt (BIT_endOfDStream, huf_decompress.c:197-199) */
        for (k = 0; k < 4; k++) {
            size_t i;
            for (i = done[k]; i < cnt[k]; i++) {
                u32 e;
                hufrd_reload(&r[k]);
                if (r[k].used >= 64) return 0;                                /* nothing left to read a symbol from */
                e = dt[(u32)((r[k].c << r[k].used) >> shift)];
                o[k][i] = (u8)e;
                r[k].used += e >> 8;
            }
            hufrd_reload(&r[k]);
            if (!(r[k].ptr == r[k].start && r[k].used == 64)) return 0;
        }
    }
    return 1;
}

/* =============================== container + codewords =============================== */

typedef struct {
    const u8 *lit, *litEnd;          /* literals stream (length escapes and, fastLZ4, the offsets live here too) */
    const u8 *fl, *flEnd;            /* flags (tokens) */
    const u8 *o16, *o16End;
    const u8 *o24, *o24End;
} Streams;

typedef struct {
    u8* dst0;                        /* start of this call's output */
    const u8* prefixStart;           /* lowest address of the contiguous history in front of the output (== dst0 without a prefix) */
    const u8* ext; size_t extSize;   /* external dictionary: the bytes logically in front of prefixStart */
    int partial;
    size_t target;
} Env;

/* Length escape (lizard_decompress_lz4.h:47-58): 1 byte < 254, 254 + LE16, 255 + LE24.  Advances *pp; 0 if it runs past end. */
static int len_ext(const u8** pp, const u8* end, size_t* value)
{
    const u8* p = *pp;
    u32 b0;
    if (p >= end) return 0;
    b0 = *p;
    if (b0 < 254) { *value = b0; *pp = p + 1; return 1; }
    if (b0 == 254) { if (end - p < 3) return 0; *value = le16(p + 1); *pp = p + 3; return 1; }
    if (end - p < 4) return 0;
    *value = le24(p + 1); *pp = p + 4;
    return 1;
}

/* out[op..op+ml) = the bytes `off` back, LZ semantics; the source may lie in the prefix, in the external dictionary, or straddle
 * both (lizard_decompress_lz4.h:112-133).  0 = offset outside the history. */
/* n bytes in steps of 16 / 8, writing (and reading) up to 15 / 7 bytes past the end: every caller has checked the reference's
 * WILDCOPYLENGTH margins (16 bytes of room behind the copy in the output, and behind the literals in their stream) */
static void wild16(u8* d, const u8* s, size_t n) { u8* const e = d + n; do { memcpy(d, s, 16); d += 16; s += 16; } while (d < e); }
static void wild8(u8* d, const u8* s, size_t n)  { u8* const e = d + n; do { memcpy(d, s, 8); d += 8; s += 8; } while (d < e); }

static int copy_match(const Env* e, u8* op, size_t off, size_t ml)
{
    const size_t back = (size_t)(op - e->prefixStart);                        /* contiguous bytes available behind op */
    if (off <= back) {
        const u8* m = op - off;
        if (off >= 16) wild16(op, m, ml);                                     /* (ml >= 3; room: ml <= oend - op - WILDCOPYLENGTH) */
        else if (off >= 8) wild8(op, m, ml);
        else { size_t i; for (i = 0; i < ml; i++) op[i] = m[i]; }
        return 1;
    }
    {
        const size_t d = off - back;                                          /* distance into the external dictionary, from its end */
        if (d > e->extSize) return 0;
        if (ml <= d) { memmove(op, e->ext + e->extSize - d, ml); return 1; }
        memcpy(op, e->ext + e->extSize - d, d);
        {   /* the rest continues at the start of the prefix and may run into the bytes just written */
            const u8* from = e->prefixStart;
            size_t i;
            op += d;
            for (i = 0; i < ml - d; i++) op[i] = from[i];
        }
        return 1;
    }
}

/* One sub-block of fastLZ4 codewords (lizard_decompress_lz4.h:7-160).  Returns bytes written (> 0), 0 / negative as the reference. */
static long decode_lz4(const Env* e, Streams* s, u8* const dest, size_t cap)
{
    u8* op = dest;
    u8* const oend = dest + cap;
    const u8* lp = s->lit;
    const u8* const iend = s->litEnd;
    const u8* const oexit = dest + e->target;                                 /* :27 relative to the sub-block, as the reference has it */
    if (cap == 0) return (s->flEnd - s->fl == 1 && *s->fl == 0) ? 0 : -1;     /* :39 */
    while (s->fl < s->flEnd) {
        const u32 token = *s->fl++;
        size_t L = token & 15u, ml, off;
        if (L == 15) {
            if (iend - lp < 5) return -1;                                     /* :47 */
            if (!len_ext(&lp, iend, &L)) return -1;
            L += 15;
        }
        if ((long)L > (long)(oend - op) - LZH_WILD || (long)L > (long)(iend - lp) - (2 + LZH_WILD)) return -1;   /* :63 */
        if (L) wild16(op, lp, L);                                             /* (both margins just checked) */
        op += L; lp += L;
        if (e->partial && op >= oexit) return (long)(op - dest);              /* :77 */
        off = le16(lp); lp += 2;                                              /* :80-81 */
        ml = token >> 4;
        if (ml == 15) {
            if (iend - lp < 5) return -1;                                     /* :88 */
            if (!len_ext(&lp, iend, &ml)) return -1;
            ml += 15;
        }
        ml += 4;
        if ((long)ml > (long)(oend - op) - LZH_WILD) return -1;               /* :113 / :136 */
        if (off == 0 || !copy_match(e, op, off, ml)) return -1;               /* :84 (offset 0 copies unwritten bytes in the reference: refused here) */
        op += ml;
        if (e->partial && op >= oexit) return (long)(op - dest);              /* :142 */
    }
    {   /* last literals: the rest of the stream (:145-151) */
        const long rest = (long)(iend - lp);
        if (rest < 0 || rest > (long)(oend - op)) return -1;
        if (rest) memcpy(op, lp, (size_t)rest);                               /* (dest may be NULL with a capacity of 0) */
        op += rest;
    }
    return (long)(op - dest);
}

/* One sub-block of LIZv1 codewords (lizard_decompress_liz.h:14-220). */
static long decode_lizv1(const Env* e, Streams* s, u8* const dest, size_t cap)
{
    u8* op = dest;
    u8* const oend = dest + cap;
    const u8* lp = s->lit;
    const u8* const iend = s->litEnd;
    const u8* const oexit = dest + e->target;
    size_t last_off = 0;                                                      /* LIZARD_INIT_LAST_OFFSET, lizard_decompress.c:240 */
    if (cap == 0) return (s->flEnd - s->fl == 1 && *s->fl == 0) ? 0 : -1;     /* :46 */
    while (s->fl < s->flEnd) {
        u32 token;
        size_t ml;
        if (e->partial && op >= oexit) return (long)(op - dest);              /* :54 */
        token = *s->fl++;
        if (token >= 32) {
            size_t L = token & 7u;
            if (L == 7) {                                                     /* :61-77 */
                if (!len_ext(&lp, iend, &L)) return -1;
                L += 7;
            }
            if ((long)L > (long)(oend - op) - LZH_WILD || (long)(iend - lp) < LZH_WILD || L > (size_t)(iend - lp)) return -1;   /* :81 */
            if (L + LZH_WILD <= (size_t)(iend - lp)) { if (L) wild16(op, lp, L); }
            else memcpy(op, lp, L);
            op += L; lp += L;
            if ((token >> 7) == 0) {                                          /* :96-110 a new 16-bit offset */
                if (s->o16End - s->o16 < 2) return -1;
                last_off = le16(s->o16); s->o16 += 2;
            }
            ml = (token >> 3) & 15u;
            if (ml == 15) {                                                   /* :113-128 */
                if (!len_ext(&lp, iend, &ml)) return -1;
                ml += 15;
            }
        } else if (token < 31) {                                              /* :134-141 24-bit offset, match length 16..46 */
            if (s->o24End - s->o24 < 3) return -1;
            ml = token + 16;
            last_off = le24(s->o24); s->o24 += 3;
        } else {                                                              /* :142-161 24-bit offset, match length 47+ */
            if (!len_ext(&lp, iend, &ml)) return -1;
            ml += 31 + 16;
            if (s->o24End - s->o24 < 3) return -1;
            last_off = le24(s->o24); s->o24 += 3;
        }
        if ((long)ml > (long)(oend - op) - LZH_WILD) return -1;               /* :171 / :194 (also for a literal-only token) */
        if (ml) {
            if (last_off == 0 || !copy_match(e, op, last_off, ml)) return -1; /* :165 */
            op += ml;
        } else if (last_off > (size_t)(op - e->prefixStart) + e->extSize) return -1;
    }
    {   /* last literals (:204-211) */
        const long rest = (long)(iend - lp);
        if (rest < 0 || rest > (long)(oend - op)) return -1;
        if (rest) memcpy(op, lp, (size_t)rest);                               /* (dest may be NULL with a capacity of 0) */
        op += rest;
    }
    return (long)(op - dest);
}

/* one stream of a sub-block (Lizard_readStream, lizard_decompress.c:72-112): raw = LE24 length + bytes; huff0 = LE24 decoded
 * length, LE24 compressed length, huff0 stream, decoded into `stage` */
static int read_stream(int huf, const u8** ipp, const u8* iend, u8* stage, const u8** ptr, const u8** end)
{
    const u8* ip = *ipp;
    if (!huf) {
        size_t n;
        if (iend - ip < 3) return 0;
        n = le24(ip);
        if (n > (size_t)(iend - ip - 3)) return 0;
        *ptr = ip + 3; *end = ip + 3 + n; *ipp = ip + 3 + n;
        return 1;
    }
    {
        size_t n, c;
        if (iend - ip < 6) return 0;
        n = le24(ip); c = le24(ip + 3);
        if (n > LZH_STAGE || c > (size_t)(iend - ip - 6)) return 0;           /* :96 */
        if (!huf_decompress(stage, n, ip + 6, c)) return 0;
        *ptr = stage; *end = stage + n; *ipp = ip + 6 + c;
        return 1;
    }
}

/* Lizard_decompress_generic (lizard_decompress.c:115-265) */
static int decode_block(const u8* src, int srcSize, u8* dst, int dstCap, int partial, int target,
                        const u8* prefixStart, const u8* ext, size_t extSize)
{
    const u8* ip = src;
    const u8* const iend = src + srcSize;
    u8* op = dst;
    u8* const oend = dst + dstCap;
    u8* stage = NULL;                                                         /* 4 x 128 KiB, only when a stream is huff0-compressed */
    Env e;
    int lz4, level, result = -1;
    if (srcSize < 1) return 0;                                                /* :139 */
    level = *ip++;
    if (level < LIZARD_MIN_CLEVEL || level > LIZARD_MAX_CLEVEL) return -1;    /* :143 */
    lz4 = (level >= 10 && level <= 19) || (level >= 30 && level <= 39);       /* decompressType, lizard_common.h:234-284 */
    e.dst0 = dst; e.prefixStart = prefixStart; e.ext = ext; e.extSize = extSize; e.partial = partial; e.target = (size_t)(target > 0 ? target : 0);
    while (ip < iend) {                                                       /* :161 */
        const u32 res = *ip++;
        Streams s;
        long r;
        if (res == 128) {                                                     /* :164-180 stored sub-block */
            size_t n;
            if (iend - ip < 3) goto done;
            n = le24(ip); ip += 3;
            if (n > (size_t)(iend - ip) || n > (size_t)(oend - op)) goto done;
            if (n) memcpy(op, ip, n);
            op += n; ip += n;
            if (partial && op >= dst + e.target) break;
            continue;
        }
        if (res & 16) goto done;                                              /* :182 */
        if (iend - ip < 15) goto done;                                        /* :186 */
        {   /* the `len` stream: always raw, unused by these codewords (:187-189) */
            const size_t n = le24(ip);
            if (n > (size_t)(iend - ip - 3) || (size_t)(iend - ip - 3) - n < 3) goto done;
            ip += 3 + n;
        }
        if ((res & 15) && !stage) { stage = (u8*)malloc(4 * (LZH_STAGE + 32)); if (!stage) goto done; }
        if (!read_stream(res & 4, &ip, iend, stage ? stage + 2 * (LZH_STAGE + 32) : NULL, &s.o16, &s.o16End)) goto done;
        if (!read_stream(res & 8, &ip, iend, stage ? stage + 3 * (LZH_STAGE + 32) : NULL, &s.o24, &s.o24End)) goto done;
        if (!read_stream(res & 2, &ip, iend, stage ? stage + 1 * (LZH_STAGE + 32) : NULL, &s.fl, &s.flEnd)) goto done;
        if (!read_stream(res & 1, &ip, iend, stage, &s.lit, &s.litEnd)) goto done;
        r = lz4 ? decode_lz4(&e, &s, op, (size_t)(oend - op)) : decode_lizv1(&e, &s, op, (size_t)(oend - op));
        if (r <= 0) { result = r < 0 ? -1 : 0; goto out; }                    /* :248 */
        op += r;
        if (partial && op >= dst + e.target) break;                           /* :252 */
    }
    result = (int)(op - dst);
    goto out;
done:
    result = -1;
out:
    free(stage);
    return result;
}

/* =============================== the reference's entry points =============================== */

/* reference lib/lizard_decompress.h:64 / lizard_decompress.c:267 */
int Lizard_decompress_safe(const char* source, char* dest, int compressedSize, int maxDecompressedSize)
{
    if (maxDecompressedSize < 0) return -1;
    return decode_block((const u8*)source, compressedSize, (u8*)dest, maxDecompressedSize, 0, 0, (const u8*)dest, NULL, 0);
}

/* reference lib/lizard_decompress.h:80 / lizard_decompress.c:272 */
int Lizard_decompress_safe_partial(const char* source, char* dest, int compressedSize, int targetOutputSize, int maxDecompressedSize)
{
    if (maxDecompressedSize < 0) return -1;
    return decode_block((const u8*)source, compressedSize, (u8*)dest, maxDecompressedSize, 1, targetOutputSize, (const u8*)dest, NULL, 0);
}

/* Lizard_streamDecode_t: callers that include the reference's lizard_common.h keep one on their stack (tests/fuzzer.c:658)
 * and initialise it with Lizard_setStreamDecode or memset (lib/lizard_decompress.h:97): four words, valid when zero
 * (reference lib/lizard_common.h:195-200). */
struct Lizard_streamDecode_s { const u8* extDict; size_t extSize; const u8* prefixEnd; size_t prefixSize; };

Lizard_streamDecode_t* Lizard_createStreamDecode(void) { return (Lizard_streamDecode_t*)calloc(1, sizeof(Lizard_streamDecode_t)); }   /* :288 */
int Lizard_freeStreamDecode(Lizard_streamDecode_t* s) { free(s); return 0; }                                                          /* :294 */

int Lizard_setStreamDecode(Lizard_streamDecode_t* s, const char* dictionary, int dictSize)      /* :307 */
{
    s->prefixSize = (size_t)dictSize;
    s->prefixEnd = (const u8*)dictionary + dictSize;
    s->extDict = NULL; s->extSize = 0;
    return 1;
}

/* :325-350: the previous output is the history — contiguous with dest (prefix grows) or left where it was (becomes the external
 * dictionary) */
int Lizard_decompress_safe_continue(Lizard_streamDecode_t* s, const char* source, char* dest, int compressedSize, int maxOutputSize)
{
    int r;
    if (maxOutputSize < 0) return -1;
    if (s->prefixEnd == (const u8*)dest) {
        r = decode_block((const u8*)source, compressedSize, (u8*)dest, maxOutputSize, 0, 0, s->prefixEnd - s->prefixSize, s->extDict, s->extSize);
        if (r <= 0) return r;
        s->prefixSize += (size_t)r; s->prefixEnd += r;
    } else {
        s->extSize = s->prefixSize;
        s->extDict = s->prefixEnd - s->extSize;
        r = decode_block((const u8*)source, compressedSize, (u8*)dest, maxOutputSize, 0, 0, (const u8*)dest, s->extDict, s->extSize);
        if (r <= 0) return r;
        s->prefixSize = (size_t)r; s->prefixEnd = (const u8*)dest + r;
    }
    return r;
}

/* :360-370 */
int Lizard_decompress_safe_usingDict(const char* source, char* dest, int compressedSize, int maxOutputSize, const char* dictStart, int dictSize)
{
    if (maxOutputSize < 0 || dictSize < 0) return -1;
    if (dictSize == 0) return decode_block((const u8*)source, compressedSize, (u8*)dest, maxOutputSize, 0, 0, (const u8*)dest, NULL, 0);
    if (dictStart + dictSize == dest)                                         /* the dictionary is a prefix (no offset reaches beyond LIZARD_DICT_SIZE) */
        return decode_block((const u8*)source, compressedSize, (u8*)dest, maxOutputSize, 0, 0,
                            (const u8*)dest - ((size_t)dictSize > LZH_DICT_SIZE ? LZH_DICT_SIZE : (size_t)dictSize), NULL, 0);
    return decode_block((const u8*)source, compressedSize, (u8*)dest, maxOutputSize, 0, 0, (const u8*)dest, (const u8*)dictStart, (size_t)dictSize);
}

/* :374-377 (debug function of the reference; tests/fullbench.c calls it) */
int Lizard_decompress_safe_forceExtDict(const char* source, char* dest, int compressedSize, int maxOutputSize, const char* dictStart, int dictSize)
{
    if (maxOutputSize < 0 || dictSize < 0) return -1;
    return decode_block((const u8*)source, compressedSize, (u8*)dest, maxOutputSize, 0, 0, (const u8*)dest, (const u8*)dictStart, (size_t)dictSize);
}
