/*
 * oracle/lizard_oracle.c — TEST INFRASTRUCTURE ONLY (see lizard_oracle.h for the contract).
 *
 * CPU restatement of the Lizard 1.0 block-compress path for the levels the HIP path implements.
 * Index-based (no pointer arithmetic into caller memory), single translation unit, C99.
 * Every function cites the reference file:line whose behaviour it restates (paths relative to the
 * reference checkout, e.g. /root/reference).
 */
#include "lizard_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- constants (reference lib/lizard_common.h:72-123, lib/lizard_compress.h:121-124) ------------ */
#define LZO_SUBBLOCK          (1 << 17)   /* LIZARD_BLOCK_SIZE */
#define LZO_SUBBLOCK_PAD      (LZO_SUBBLOCK + 32)
#define LZO_MAX_INPUT         0x7E000000
#define LZO_DICT_SIZE         (1u << 24)  /* LIZARD_DICT_SIZE: index of the first byte of a fresh block */
#define LZO_MINMATCH          4
#define LZO_LASTLITERALS      16          /* WILDCOPYLENGTH */
#define LZO_MFLIMIT           (LZO_LASTLITERALS + LZO_MINMATCH)
#define LZO_MIN_OFFSET        8           /* LIZARD_FAST_MIN_OFFSET / LIZARD_PRICEFAST_MIN_OFFSET */
#define LZO_16BIT_OFFSET      (1u << 16)
#define LZO_MM_LONGOFF        16
#define LZO_FLAG_LITERALS     1
#define LZO_FLAG_FLAGS        2
#define LZO_FLAG_OFFSET16     4
#define LZO_FLAG_OFFSET24     8
#define LZO_FLAG_LEN          16
#define LZO_FLAG_UNCOMPRESSED 128

typedef enum { P_FAST, P_PRICEFAST, P_HASHCHAIN, P_NOCHAIN } lzo_parser;
typedef enum { C_LZ4, C_LIZV1 } lzo_codewords;

typedef struct {
    unsigned windowLog, hashLog, minMatchLongOff;
    unsigned contentLog, searchNum, searchLength;   /* hashChain only (lizard_common.h:240-244) */
    lzo_parser parser;
    lzo_codewords codewords;
    int huffman;
} lzo_params;

/* reference lib/lizard_common.h:234-284 (rows for the levels in scope) + lizard_compress.c:374-377 */
static int lzo_get_params(int level, lzo_params* p)
{
    int base = level >= 30 ? level - 20 : level;
    p->huffman = level >= 30;
    p->contentLog = 0; p->searchNum = 0; p->searchLength = 5;
    if (level >= 34 && level <= 38) base = level - 21;          /* 34..38 repeat the rows of 13..17 (lizard_common.h:264-268) */
    else if (level == 39 || level == 18 || level == 19) return 0;
    /* noChain, LZ4 codewords: lizard_common.h:239 (12), :262-263 (32: hashLog 14, 33: hashLog 18) */
    if (level == 12 || level == 32 || level == 33) {
        p->windowLog = 16; p->hashLog = level == 32 ? 14 : 18; p->minMatchLongOff = 0; p->searchNum = 1;
        p->parser = P_NOCHAIN; p->codewords = C_LZ4; return 1;
    }
    switch (base) {
    case 10: p->windowLog = 16; p->hashLog = 12; p->minMatchLongOff = 0;  p->parser = P_FAST;      p->codewords = C_LZ4;   return 1;
    case 11: p->windowLog = 16; p->hashLog = 18; p->minMatchLongOff = 0;  p->parser = P_FAST;      p->codewords = C_LZ4;   return 1;
    /* hashChain, LZ4 codewords: lizard_common.h:240-244 (13-17) and :264-268 (34-38) */
    case 13: case 14: case 15: case 16: case 17: {
        static const unsigned snum[5] = { 2, 4, 8, 16, 256 }, slen[5] = { 5, 5, 5, 4, 4 };
        p->windowLog = 16; p->hashLog = 18; p->minMatchLongOff = 0; p->contentLog = 16;
        p->searchNum = snum[base - 13]; p->searchLength = slen[base - 13];
        p->parser = P_HASHCHAIN; p->codewords = C_LZ4; return 1; }
    /* fastBig, LIZv1 codewords: lizard_common.h:248 (20) and :270 (40) */
    case 20: p->windowLog = 22; p->hashLog = 14; p->minMatchLongOff = 16; p->parser = P_FAST;      p->codewords = C_LIZV1; return 1;
    case 21: p->windowLog = 22; p->hashLog = 14; p->minMatchLongOff = 16; p->parser = P_PRICEFAST; p->codewords = C_LIZV1; return 1;
    case 22: p->windowLog = 22; p->hashLog = 18; p->minMatchLongOff = 16; p->parser = P_PRICEFAST; p->codewords = C_LIZV1; return 1;
    default: return 0;
    }
}

int lzo_level_supported(int level)
{
    lzo_params p;
    if (level < 10 || level > 49) return 0;
    if (level >= 30 && lzo_huf_compress(NULL, 0, NULL, 0) == (size_t)-2) return 0;   /* huff0 stage not restated yet */
    return lzo_get_params(level, &p);
}

/* reference lib/lizard_compress.h:124 */
int lzo_compress_bound(int isize)
{
    if ((unsigned)isize > (unsigned)LZO_MAX_INPUT) return 0;
    return isize + 1 + 1 + ((isize / LZO_SUBBLOCK) + 1) * 4;
}

/* ---- little-endian helpers ---------------------------------------------------------------------- */
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
static void wr16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void wr24(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); }

/* 64-bit build hash: reference lib/lizard_compress.c:77,90-91 (hash5 of an 8-byte read), selected by
 * lizard_parser_fastsmall.h:4-9, lizard_parser_fast.h:7-12 and searchLength 5 (lizard_compress.c:105). */
static uint32_t hash5(const uint8_t* p, unsigned hashLog)
{
    return (uint32_t)(((rd64(p) * 889523592379ULL) << 24) >> (64 - hashLog));
}

/* reference lib/lizard_compress.c:87-88 (searchLength 4) */
static uint32_t hash4(const uint8_t* p, unsigned hashLog) { return (rd32(p) * 2654435761U) >> (32 - hashLog); }

/* reference lib/lizard_common.h:475-490: length of the common prefix of src[a..] and src[b..],
 * never letting a reach `limit`. (The 8/4/2/1-byte stepping of the reference is unobservable.) */
static uint32_t count_eq(const uint8_t* src, uint32_t a, uint32_t b, uint32_t limit)
{
    uint32_t n = 0;
    while (a + n < limit && src[a + n] == src[b + n]) n++;
    return n;
}

/* ---- per-call state ----------------------------------------------------------------------------- */
typedef struct {
    lzo_params prm;
    const uint8_t* src;    /* block start; position p has reference index p + 2^24 */
    uint32_t* table;       /* 2^hashLog entries, zero = empty (LIZARD_RESET_MEM state) */
    uint8_t *lit, *flags, *off16, *off24;     /* sub-block stream staging (lizard_compress.c:379-384) */
    uint32_t nlit, nflags, noff16, noff24;    /* the `len` stream is never written by these codewords */
    uint32_t last_off;     /* LIZv1 repeat offset, reset per sub-block (lizard_compress.c:137) */
    uint32_t* chain;       /* hashChain: 2^contentLog deltas (DELTANEXT, lizard_compress.c:58) */
    uint32_t nextToUpdate; /* hashChain: first position not yet inserted (block-relative), lizard_compress.c:334 */
} lzo_ctx;

/* length escape shared by all codewords: reference lib/lizard_compress_lz4.h:21-23 */
static void put_len_ext(lzo_ctx* c, uint32_t v)
{
    if (v >= (1u << 16)) { c->lit[c->nlit] = 255; wr24(c->lit + c->nlit + 1, v); c->nlit += 4; }
    else if (v >= 254)   { c->lit[c->nlit] = 254; wr16(c->lit + c->nlit + 1, v); c->nlit += 3; }
    else c->lit[c->nlit++] = (uint8_t)v;
}

/* reference lib/lizard_compress_lz4.h:3-71 — fastLZ4 sequence: token -> flags stream, everything else
 * (literal-length ext, literals, LE16 offset, match-length ext) -> literals stream. */
static void emit_lz4(lzo_ctx* c, uint32_t anchor, uint32_t ip, uint32_t matchLength, uint32_t match)
{
    uint32_t litLen = ip - anchor, token, ml = matchLength - LZO_MINMATCH;
    if (litLen >= 15) { token = 15; put_len_ext(c, litLen - 15); } else token = litLen;
    memcpy(c->lit + c->nlit, c->src + anchor, litLen); c->nlit += litLen;
    wr16(c->lit + c->nlit, ip - match); c->nlit += 2;
    if (ml >= 15) { token += 15u << 4; put_len_ext(c, ml - 15); } else token += ml << 4;
    c->flags[c->nflags++] = (uint8_t)token;
}

/* reference lib/lizard_compress_liz.h:43-165 — LIZv1 sequence. `match == ip` encodes a repeat offset. */
static void emit_lizv1(lzo_ctx* c, uint32_t anchor, uint32_t ip, uint32_t matchLength, uint32_t match)
{
    uint32_t offset = ip - match, litLen = ip - anchor, token = 0;
    int haveToken = 0;
    if (litLen > 0 || offset < LZO_16BIT_OFFSET) {                       /* liz.h:54 */
        if (litLen >= 7) { token = 7; put_len_ext(c, litLen - 7); } else token = litLen;
        memcpy(c->lit + c->nlit, c->src + anchor, litLen); c->nlit += litLen;
        haveToken = 1;
        if (offset >= LZO_16BIT_OFFSET) {                                /* liz.h:83-93: literal-only token */
            c->flags[c->nflags++] = (uint8_t)(token + (1u << 7));
            haveToken = 0; token = 0;
        }
    }
    if (offset >= LZO_16BIT_OFFSET) {                                    /* liz.h:96-121: 24-bit offset */
        if (matchLength - LZO_MM_LONGOFF >= 31) { token = 31; put_len_ext(c, matchLength - LZO_MM_LONGOFF - 31); }
        else token = matchLength - LZO_MM_LONGOFF;
        wr24(c->off24 + c->noff24, offset); c->noff24 += 3;
        c->last_off = offset;
    } else {                                                             /* liz.h:122-148 */
        (void)haveToken;
        if (offset == 0) token += 1u << 7;
        else { c->last_off = offset; wr16(c->off16 + c->noff16, offset); c->noff16 += 2; }
        if (matchLength >= 15) { token += 15u << 3; put_len_ext(c, matchLength - 15); }
        else token += matchLength << 3;
    }
    c->flags[c->nflags++] = (uint8_t)token;
}

static void emit_seq(lzo_ctx* c, uint32_t anchor, uint32_t ip, uint32_t ml, uint32_t match)
{
    if (c->prm.codewords == C_LZ4) emit_lz4(c, anchor, ip, ml, match); else emit_lizv1(c, anchor, ip, ml, match);
}

/* reference lib/lizard_compress_lz4.h:74-86, lizard_compress_liz.h:168-179 */
static void emit_last_literals(lzo_ctx* c, uint32_t anchor, uint32_t end)
{
    memcpy(c->lit + c->nlit, c->src + anchor, end - anchor); c->nlit += end - anchor;
}

/* ---- fastSmall / fast parser: reference lib/lizard_parser_fastsmall.h:34-189 == lizard_parser_fast.h:41-196,
 * and fastBig (levels 20 / 40): lib/lizard_parser_fastbig.h:36-175 — the same loop with the level's hashLog and window, LIZv1
 * codewords (:127), and the long-offset rule: a candidate 65 536 or more back counts only when the forward count behind the first
 * four bytes plus the backward extension is at least minMatchLongOff (:92-98 in the search, :143-146 in the post-match probe). */
static void parse_fast(lzo_ctx* c, uint32_t S, uint32_t E)
{
    const uint8_t* src = c->src;
    const unsigned hl = c->prm.hashLog;
    const uint32_t mmlo = c->prm.minMatchLongOff;   /* 0: fast / fastSmall (LIZARD_FAST_LONGOFF_MM 0, fast.h:2) */
    const uint32_t maxDist = (1u << c->prm.windowLog) - 1;
    /* fast.h:57-58: lowLimit fixed at sub-block entry; ctx->lowLimit = dictLimit = 2^24 for a fresh block */
    const uint32_t idxS = S + LZO_DICT_SIZE;
    const uint32_t lowLimit = (LZO_DICT_SIZE + maxDist >= idxS) ? LZO_DICT_SIZE : idxS - maxDist;
    uint32_t anchor = S, ip = S;

    if (E - S < LZO_MFLIMIT + 1) { emit_last_literals(c, anchor, E); return; }   /* fast.h:63 */
    {
        const uint32_t mflimit = E - LZO_MFLIMIT, matchlimit = E - LZO_LASTLITERALS;
        uint32_t match = 0, ml = 0;
        c->table[hash5(src + ip, hl)] = ip + LZO_DICT_SIZE;                       /* fast.h:66 */
        ip++;
        for (;;) {
            /* search run: fast.h:75-134 */
            uint32_t fwd = ip, step = 1, nb = 1u << 6;
            for (;;) {
                uint32_t h, e, cur;
                ip = fwd; fwd += step; step = nb++ >> 6;
                if (fwd > mflimit) goto tail;                                     /* fast.h:84 */
                h = hash5(src + ip, hl);
                e = c->table[h];
                cur = ip + LZO_DICT_SIZE;
                c->table[h] = cur;                                                /* fast.h:86-88 */
                if (e < lowLimit || e >= cur || e + maxDist < cur) continue;      /* fast.h:90 */
                match = e - LZO_DICT_SIZE;
                if (ip - match < LZO_MIN_OFFSET) continue;                        /* fast.h:95 */
                if (rd32(src + match) != rd32(src + ip)) continue;                /* fast.h:97 */
                ml = LZO_MINMATCH + count_eq(src, ip + 4, match + 4, matchlimit); /* fast.h:100 */
                {
                    uint32_t bk = 0;                                              /* fast.h:102-103 / fastbig.h:94-100 */
                    while (ip - bk > anchor && match - bk > 0 && src[ip - bk - 1] == src[match - bk - 1]) bk++;
                    if (mmlo && !(ml - LZO_MINMATCH + bk >= mmlo || ip - match < LZO_16BIT_OFFSET)) continue;   /* fastbig.h:96 */
                    ip -= bk; match -= bk; ml += bk;
                }
                break;
            }
            for (;;) {
                uint32_t h, e, cur;
                emit_seq(c, anchor, ip, ml, match);                               /* fast.h:138 */
                ip += ml; anchor = ip;
                if (ip > mflimit) goto tail;                                      /* fast.h:143 */
                c->table[hash5(src + ip - 2, hl)] = ip - 2 + LZO_DICT_SIZE;       /* fast.h:146 */
                h = hash5(src + ip, hl);
                e = c->table[h];
                cur = ip + LZO_DICT_SIZE;
                c->table[h] = cur;                                                /* fast.h:149-150 */
                if (e >= lowLimit && e < cur && e + maxDist >= cur) {             /* fast.h:151 */
                    match = e - LZO_DICT_SIZE;
                    if (ip - match >= LZO_MIN_OFFSET && rd32(src + match) == rd32(src + ip)) {
                        ml = LZO_MINMATCH + count_eq(src, ip + 4, match + 4, matchlimit);   /* :159, no back-extension */
                        if (mmlo && !(ml - LZO_MINMATCH >= mmlo || ip - match < LZO_16BIT_OFFSET)) break;     /* fastbig.h:145 */
                        continue;
                    }
                }
                break;
            }
            ip++;                                                                 /* fast.h:184 */
        }
    }
tail:
    emit_last_literals(c, anchor, E);                                             /* fast.h:187-190 */
}

/* ---- priceFast parser: reference lib/lizard_parser_pricefast.h ---------------------------------- */

/* table update rule, pricefast.h:170-171 / :190-191 */
static void pf_update(lzo_ctx* c, uint32_t h, uint32_t cur)
{
    uint32_t t = c->table[h];
    if (t >= cur || cur >= t + LZO_MIN_OFFSET) c->table[h] = cur;
}

/* hash-candidate test shared by Lizard_FindMatchFast (:63-72) and Lizard_FindMatchFaster (:106-114).
 * Returns match length (0 = none) and sets *mpos. */
static uint32_t pf_hash_candidate(const lzo_ctx* c, uint32_t e, uint32_t ip, uint32_t matchlimit, uint32_t* mpos)
{
    const uint8_t* src = c->src;
    const uint32_t maxDist = (1u << c->prm.windowLog) - 1;
    const uint32_t cur = ip + LZO_DICT_SIZE;
    const uint32_t lowLimit = (LZO_DICT_SIZE + maxDist >= cur) ? LZO_DICT_SIZE : cur - maxDist;   /* :11-13, per probe */
    uint32_t m, ml;
    if (!(e < cur && e >= lowLimit)) return 0;
    m = e - LZO_DICT_SIZE;
    if (ip - m < LZO_MIN_OFFSET) return 0;
    if (rd32(src + m) != rd32(src + ip)) return 0;
    ml = LZO_MINMATCH + count_eq(src, ip + 4, m + 4, matchlimit);
    if (ml >= c->prm.minMatchLongOff || ip - m < LZO_16BIT_OFFSET) { *mpos = m; return ml; }
    return 0;
}

/* Lizard_FindMatchFast, pricefast.h:3-87: repeat-offset probe first, then the hash candidate. */
static uint32_t pf_find_fast(const lzo_ctx* c, uint32_t e, uint32_t ip, uint32_t matchlimit, uint32_t* mpos)
{
    const uint8_t* src = c->src;
    if (c->last_off >= LZO_MIN_OFFSET) {                                          /* :19 */
        const uint32_t maxDist = (1u << c->prm.windowLog) - 1;
        const int64_t cur = (int64_t)ip + LZO_DICT_SIZE;
        const int64_t lowLimit = ((int64_t)LZO_DICT_SIZE + maxDist >= cur) ? (int64_t)LZO_DICT_SIZE : cur - maxDist;
        const int64_t idxLO = cur - (int64_t)c->last_off;
        if (idxLO >= lowLimit) {                                                  /* :21; >= dictLimit always holds then */
            uint32_t m = (uint32_t)(idxLO - LZO_DICT_SIZE);
            if (rd32(src + m) == rd32(src + ip)) {                                /* :24-30: returns immediately */
                *mpos = m;
                return LZO_MINMATCH + count_eq(src, ip + 4, m + 4, matchlimit);
            }
        }
    }
    return pf_hash_candidate(c, e, ip, matchlimit, mpos);
}

static void parse_pricefast(lzo_ctx* c, uint32_t S, uint32_t E)
{
    const uint8_t* src = c->src;
    const unsigned hl = c->prm.hashLog;
    uint32_t anchor = S, ip = S;
    uint32_t mflimit, matchlimit;
    /* pricefast.h:137-138: mflimit = iend - MFLIMIT computed with pointers; a sub-block shorter than
     * MFLIMIT simply never enters the loop (ip+1 < mflimit is false even if mflimit is before S). */
    if (E - S < LZO_MFLIMIT + 1) { emit_last_literals(c, anchor, E); return; }
    mflimit = E - LZO_MFLIMIT; matchlimit = E - LZO_LASTLITERALS;
    ip++;                                                                         /* :155 */
    while (ip < mflimit) {                                                        /* :158 */
        uint32_t ml, ml2 = 0, ref = 0, ref2 = 0, start2 = 0;
        uint32_t h = hash5(src + ip, hl);
        ml = pf_find_fast(c, c->table[h], ip, matchlimit, &ref);                  /* :168 */
        pf_update(c, h, ip + LZO_DICT_SIZE);                                      /* :170-171 */
        if (!ml) { ip++; continue; }                                              /* :173 */
        if (ip - ref == c->last_off) { ml2 = 0; ref = ip; goto encode; }          /* :174 (last_off==0 never equals: ip-ref>=4... see note) */
        while (ip > anchor && ref > 0 && src[ip - 1] == src[ref - 1]) { ip--; ref--; ml++; }   /* :176-182 */
    search:
        if (ip + ml >= mflimit) goto encode;                                      /* :185 */
        start2 = ip + ml - 2;
        {
            uint32_t h2 = hash5(src + start2, hl);
            ml2 = pf_hash_candidate(c, c->table[h2], start2, matchlimit, &ref2);  /* :189 */
            pf_update(c, h2, start2 + LZO_DICT_SIZE);                             /* :190-191 */
        }
        if (!ml2) goto encode;
        while (start2 > ip && ref2 > 0 && src[start2 - 1] == src[ref2 - 1]) { start2--; ref2--; ml2++; }   /* :195-201 */
        if (ml2 <= ml) { ml2 = 0; goto encode; }                                  /* :203 */
        if (start2 <= ip) { ip = start2; ref = ref2; ml = ml2; ml2 = 0; goto encode; }          /* :205-210 */
        if (start2 - ip < 3) { ip = start2; ref = ref2; ml = ml2; ml2 = 0; goto search; }       /* :212-217 */
        if (start2 < ip + ml) {                                                   /* :219-228 */
            uint32_t correction = ml - (start2 - ip);
            start2 += correction; ref2 += correction; ml2 -= correction;
            if (ml2 < 3) ml2 = 0;
            if (ml2 < c->prm.minMatchLongOff && start2 - ref2 >= LZO_16BIT_OFFSET) ml2 = 0;
        }
    encode:
        emit_seq(c, anchor, ip, ml, ref);                                         /* :231 */
        ip += ml; anchor = ip;
        if (ml2) { ip = start2; ref = ref2; ml = ml2; ml2 = 0; goto search; }     /* :233-238 */
    }
    emit_last_literals(c, anchor, E);
}

/* ---- hashChain parser: reference lib/lizard_parser_hashchain.h ------------------------------------ */
#define HC_OPTIMAL_ML 18        /* (ML_MASK_LZ4-1)+MINMATCH, hashchain.h:3 */

static uint32_t hc_hash(const lzo_ctx* c, uint32_t pos)
{
    return c->prm.searchLength == 4 ? hash4(c->src + pos, c->prm.hashLog) : hash5(c->src + pos, c->prm.hashLog);
}

/* Lizard_Insert, hashchain.h:13-43: chain + conditional head update for every position below `target`. */
static void hc_insert(lzo_ctx* c, uint32_t target)
{
    const uint32_t mask = (1u << c->prm.contentLog) - 1, maxDist = (1u << c->prm.windowLog) - 1;
    uint32_t pos = c->nextToUpdate;
    while (pos < target) {
        const uint32_t idx = pos + LZO_DICT_SIZE, h = hc_hash(c, pos), head = c->table[h];
        uint32_t delta = idx - head;
        if (delta > maxDist) delta = maxDist;
        c->chain[idx & mask] = delta;
        if (head >= idx || idx >= head + LZO_MIN_OFFSET) c->table[h] = idx;
        pos++;
    }
    c->nextToUpdate = target;
}

/* Lizard_InsertAndFindBestMatch, hashchain.h:45-107. Returns length (0 = none), *ref = match position. */
static int hc_find_best(lzo_ctx* c, uint32_t ip, uint32_t iLimit, uint32_t* ref)
{
    const uint8_t* src = c->src;
    const uint32_t mask = (1u << c->prm.contentLog) - 1, maxDist = (1u << c->prm.windowLog) - 1;
    const uint32_t cur = ip + LZO_DICT_SIZE;
    const uint32_t lowLimit = (LZO_DICT_SIZE + maxDist >= cur) ? LZO_DICT_SIZE : cur - maxDist;
    int attempts = (int)c->prm.searchNum;
    uint32_t ml = 0, mi;
    hc_insert(c, ip);
    mi = c->table[hc_hash(c, ip)];
    while (mi < cur && mi >= lowLimit && attempts) {
        const uint32_t m = mi - LZO_DICT_SIZE;
        uint32_t delta;
        attempts--;
        if (ip - m >= LZO_MIN_OFFSET && src[m + ml] == src[ip + ml] && rd32(src + m) == rd32(src + ip)) {
            const uint32_t mlt = LZO_MINMATCH + count_eq(src, ip + 4, m + 4, iLimit);
            if (mlt > ml) { ml = mlt; *ref = m; }
        }
        delta = c->chain[mi & mask];
        if (delta > mi) break;
        mi -= delta;
    }
    return (int)ml;
}

/* Lizard_InsertAndGetWiderMatch, hashchain.h:109-185. */
static int hc_wider(lzo_ctx* c, uint32_t ip, uint32_t iLow, uint32_t iHigh, int longest, uint32_t* ref, uint32_t* start)
{
    const uint8_t* src = c->src;
    const uint32_t mask = (1u << c->prm.contentLog) - 1, maxDist = (1u << c->prm.windowLog) - 1;
    const uint32_t cur = ip + LZO_DICT_SIZE;
    const uint32_t lowLimit = (LZO_DICT_SIZE + maxDist >= cur) ? LZO_DICT_SIZE : cur - maxDist;
    const int LLdelta = (int)(ip - iLow);
    int attempts = (int)c->prm.searchNum;
    uint32_t mi;
    hc_insert(c, ip);
    mi = c->table[hc_hash(c, ip)];
    while (mi < cur && mi >= lowLimit && attempts) {
        const uint32_t m = mi - LZO_DICT_SIZE;
        uint32_t delta;
        attempts--;
        if (ip - m >= LZO_MIN_OFFSET && src[iLow + longest] == src[m - LLdelta + longest] && rd32(src + m) == rd32(src + ip)) {
            int mlt = LZO_MINMATCH + (int)count_eq(src, ip + 4, m + 4, iHigh), back = 0;
            while (ip + back > iLow && m + back > 0 && src[ip + back - 1] == src[m + back - 1]) back--;
            mlt -= back;
            if (mlt > longest) { longest = mlt; *ref = m + back; *start = ip + back; }
        }
        delta = c->chain[mi & mask];
        if (delta > mi) break;
        mi -= delta;
    }
    return longest;
}

/* ---- noChain parser (levels 12 / 32 / 33): reference lib/lizard_parser_nochain.h -------------------
 * The hashChain parser without a chain: one candidate per search (the bucket's head), hash5 whatever the level's
 * searchLength (nochain.h:4), the same conditional head update. */
/* Lizard_InsertNoChain, nochain.h:8-25 */
static void nc_insert(lzo_ctx* c, uint32_t target)
{
    uint32_t pos = c->nextToUpdate;
    while (pos < target) {
        const uint32_t idx = pos + LZO_DICT_SIZE, h = hash5(c->src + pos, c->prm.hashLog), head = c->table[h];
        if (head >= idx || idx >= head + LZO_MIN_OFFSET) c->table[h] = idx;
        pos++;
    }
    c->nextToUpdate = target;
}

/* Lizard_InsertAndFindBestMatchNoChain, nochain.h:28-80 */
static int nc_find_best(lzo_ctx* c, uint32_t ip, uint32_t iLimit, uint32_t* ref)
{
    const uint8_t* src = c->src;
    const uint32_t maxDist = (1u << c->prm.windowLog) - 1, cur = ip + LZO_DICT_SIZE;
    const uint32_t lowLimit = (LZO_DICT_SIZE + maxDist >= cur) ? LZO_DICT_SIZE : cur - maxDist;
    uint32_t ml = 0, mi;
    nc_insert(c, ip);
    mi = c->table[hash5(src + ip, c->prm.hashLog)];
    if (mi < cur && mi >= lowLimit) {
        const uint32_t m = mi - LZO_DICT_SIZE;
        if (ip - m >= LZO_MIN_OFFSET && src[m] == src[ip] && rd32(src + m) == rd32(src + ip)) {
            ml = LZO_MINMATCH + count_eq(src, ip + 4, m + 4, iLimit);
            *ref = m;
        }
    }
    return (int)ml;
}

/* Lizard_InsertAndGetWiderMatchNoChain, nochain.h:83-143 */
static int nc_wider(lzo_ctx* c, uint32_t ip, uint32_t iLow, uint32_t iHigh, int longest, uint32_t* ref, uint32_t* start)
{
    const uint8_t* src = c->src;
    const uint32_t maxDist = (1u << c->prm.windowLog) - 1, cur = ip + LZO_DICT_SIZE;
    const uint32_t lowLimit = (LZO_DICT_SIZE + maxDist >= cur) ? LZO_DICT_SIZE : cur - maxDist;
    const int LLdelta = (int)(ip - iLow);
    uint32_t mi;
    nc_insert(c, ip);
    mi = c->table[hash5(src + ip, c->prm.hashLog)];
    if (mi < cur && mi >= lowLimit) {
        const uint32_t m = mi - LZO_DICT_SIZE;
        if (ip - m >= LZO_MIN_OFFSET && src[iLow + longest] == src[m - LLdelta + longest] && rd32(src + m) == rd32(src + ip)) {
            int mlt = LZO_MINMATCH + (int)count_eq(src, ip + 4, m + 4, iHigh), back = 0;
            while (ip + back > iLow && m + back > 0 && src[ip + back - 1] == src[m + back - 1]) back--;
            mlt -= back;
            if (mlt > longest) { longest = mlt; *ref = m + back; *start = ip + back; }
        }
    }
    return longest;
}

static int find_best(lzo_ctx* c, uint32_t ip, uint32_t iLimit, uint32_t* ref)
{
    return c->prm.parser == P_NOCHAIN ? nc_find_best(c, ip, iLimit, ref) : hc_find_best(c, ip, iLimit, ref);
}
static int wider(lzo_ctx* c, uint32_t ip, uint32_t iLow, uint32_t iHigh, int longest, uint32_t* ref, uint32_t* start)
{
    return c->prm.parser == P_NOCHAIN ? nc_wider(c, ip, iLow, iHigh, longest, ref, start) : hc_wider(c, ip, iLow, iHigh, longest, ref, start);
}

/* Lizard_compress_hashChain, hashchain.h:188-369 (LZ4HC-style three-match lazy arbitration), and Lizard_compress_noChain,
 * nochain.h:146-318: the same loop over the searches above, except that noChain has no "match2 does not fit" exit in the first
 * shortening step (nochain.h:210 against hashchain.h:255-261). */
static void parse_hashchain(lzo_ctx* c, uint32_t S, uint32_t E)
{
    const int nochain = c->prm.parser == P_NOCHAIN;
    int anchor = (int)S, ip = (int)S, mflimit, matchlimit;
    int ml, ml2, ml3, ml0;
    uint32_t ref = 0, ref2 = 0, ref3 = 0, ref0, start2 = 0, start3 = 0;
    int start0;
    /* mflimit is a pointer difference in the reference; a sub-block shorter than MFLIMIT never enters the loop */
    mflimit = (int)E - (int)LZO_MFLIMIT; matchlimit = (int)E - (int)LZO_LASTLITERALS;
    ip++;
    while (ip < mflimit) {
        ml = find_best(c, (uint32_t)ip, (uint32_t)matchlimit, &ref);
        if (!ml) { ip++; continue; }
        start0 = ip; ref0 = ref; ml0 = ml;
    search2:
        if (ip + ml < mflimit) ml2 = wider(c, (uint32_t)(ip + ml - 2), (uint32_t)(ip + 1), (uint32_t)matchlimit, ml, &ref2, &start2);
        else ml2 = ml;
        if (ml2 == ml) {                                                          /* :216-219 */
            emit_seq(c, (uint32_t)anchor, (uint32_t)ip, (uint32_t)ml, ref); ip += ml; anchor = ip;
            continue;
        }
        if (start0 < ip) {                                                        /* :221-227 */
            if ((int)start2 < ip + ml0) { ip = start0; ref = ref0; ml = ml0; }
        }
        if ((int)start2 - ip < 3) {                                               /* :230-235 */
            ml = ml2; ip = (int)start2; ref = ref2;
            goto search2;
        }
    search3:
        if ((int)start2 - ip < HC_OPTIMAL_ML) {                                   /* :243-260 */
            int correction, new_ml = ml;
            if (new_ml > HC_OPTIMAL_ML) new_ml = HC_OPTIMAL_ML;
            if (ip + new_ml > (int)start2 + ml2 - LZO_MINMATCH) {
                new_ml = (int)start2 - ip + ml2 - LZO_MINMATCH;
                if (!nochain && new_ml < LZO_MINMATCH) {                           /* hashchain.h:257-260 only */
                    emit_seq(c, (uint32_t)anchor, (uint32_t)ip, (uint32_t)ml, ref); ip += ml; anchor = ip;
                    continue;
                }
            }
            correction = new_ml - ((int)start2 - ip);
            if (correction > 0) { start2 += (uint32_t)correction; ref2 += (uint32_t)correction; ml2 -= correction; }
        }
        if ((int)start2 + ml2 < mflimit)                                          /* :263-265 */
            ml3 = wider(c, start2 + (uint32_t)ml2 - 3, start2, (uint32_t)matchlimit, ml2, &ref3, &start3);
        else ml3 = ml2;
        if (ml3 == ml2) {                                                         /* :267-275 */
            if ((int)start2 < ip + ml) ml = (int)start2 - ip;
            emit_seq(c, (uint32_t)anchor, (uint32_t)ip, (uint32_t)ml, ref); ip += ml; anchor = ip;
            ip = (int)start2;
            emit_seq(c, (uint32_t)anchor, (uint32_t)ip, (uint32_t)ml2, ref2); ip += ml2; anchor = ip;
            continue;
        }
        if ((int)start3 < ip + ml + 3) {                                          /* :277-305 */
            if ((int)start3 >= ip + ml) {
                if ((int)start2 < ip + ml) {
                    int correction = ip + ml - (int)start2;
                    start2 += (uint32_t)correction; ref2 += (uint32_t)correction; ml2 -= correction;
                    if (ml2 < LZO_MINMATCH) { start2 = start3; ref2 = ref3; ml2 = ml3; }
                }
                emit_seq(c, (uint32_t)anchor, (uint32_t)ip, (uint32_t)ml, ref); ip += ml; anchor = ip;
                ip = (int)start3; ref = ref3; ml = ml3;
                start0 = (int)start2; ref0 = ref2; ml0 = ml2;
                goto search2;
            }
            start2 = start3; ref2 = ref3; ml2 = ml3;
            goto search3;
        }
        if ((int)start2 < ip + ml) {                                              /* :311-338 */
            if ((int)start2 - ip < 15) {
                int correction;
                if (ml > HC_OPTIMAL_ML) ml = HC_OPTIMAL_ML;
                if (ip + ml > (int)start2 + ml2 - LZO_MINMATCH) {
                    ml = (int)start2 - ip + ml2 - LZO_MINMATCH;
                    if (ml < LZO_MINMATCH) {
                        emit_seq(c, (uint32_t)anchor, (uint32_t)ip, (uint32_t)ml, ref); ip += ml; anchor = ip;
                        ip = (int)start3; ref = ref3; ml = ml3;
                        start0 = (int)start2; ref0 = ref2; ml0 = ml2;
                        goto search2;
                    }
                }
                correction = ml - ((int)start2 - ip);
                if (correction > 0) { start2 += (uint32_t)correction; ref2 += (uint32_t)correction; ml2 -= correction; }
            } else {
                ml = (int)start2 - ip;
            }
        }
        emit_seq(c, (uint32_t)anchor, (uint32_t)ip, (uint32_t)ml, ref); ip += ml; anchor = ip;   /* :339 */
        ip = (int)start2; ref = ref2; ml = ml2;
        start2 = start3; ref2 = ref3; ml2 = ml3;
        goto search3;
    }
    emit_last_literals(c, (uint32_t)anchor, E);
}

/* ---- sub-block container: reference lib/lizard_compress.c:141-250 ------------------------------- */

#define LZO_HUF_GAIN(c)   ((c) + ((c) / 8) + 512)     /* lizard_compress.c:59 */
#define LZO_BLOCK_GAIN(c) ((c) + ((c) / 32) + 512)    /* lizard_compress.c:60 */

/* Lizard_writeStream, lizard_compress.c:141-183. Returns 1 (Huffman-coded), 0 (raw), -1 (dst overflow). */
static int write_stream(int useHuff, const uint8_t* stream, uint32_t n, uint8_t** op, uint8_t* oend, uint8_t* hufTmp, size_t hufTmpCap)
{
    if (useHuff && n > 1024) {
        size_t csize;
        if (*op + 6 > oend) return -1;
        /* :148-154: compress into dst when HUF_compressBound fits, else into the side buffer; the bytes
         * are the same either way, so always stage in hufTmp. */
        csize = lzo_huf_compress(hufTmp, hufTmpCap, stream, n);
        if (csize != (size_t)-1 && csize > 0 && LZO_HUF_GAIN(csize) < n) {
            if ((size_t)(oend - (*op + 6)) < csize) return -1;
            wr24(*op, n); wr24(*op + 3, (uint32_t)csize);
            memcpy(*op + 6, hufTmp, csize);
            *op += csize + 6;
            return 1;
        }
    }
    if (*op + 3 + n > oend) return -1;
    wr24(*op, n); *op += 3;
    memcpy(*op, stream, n); *op += n;
    return 0;
}

/* Lizard_writeBlock, lizard_compress.c:186-250. Returns 0 ok, 1 error (dst overflow). */
static int write_block(lzo_ctx* c, const uint8_t* ip, uint32_t n, uint8_t** op, uint8_t* oend, uint8_t* hufTmp, size_t hufTmpCap)
{
    const uint32_t sum = c->nflags + c->nlit + c->noff16 + c->noff24;   /* lenLen is always 0 */
    const int huffType = c->prm.huffman ? (LZO_FLAG_LITERALS + LZO_FLAG_FLAGS) : 0;  /* lizard_compress.c:374-377 */
    uint8_t* start = *op;
    int res;
    if (c->nlit < LZO_LASTLITERALS || sum + 5 * 3 + 1 > n) goto raw;             /* :201 */
    *start = 0; *op += 1;
    res = write_stream(0, NULL, 0, op, oend, hufTmp, hufTmpCap);                  /* len stream, :206 */
    if (res < 0) return 1;
    res = write_stream(huffType & LZO_FLAG_OFFSET16, c->off16, c->noff16, op, oend, hufTmp, hufTmpCap);
    if (res < 0) return 1;
    *start += (uint8_t)(res * LZO_FLAG_OFFSET16);
    res = write_stream(huffType & LZO_FLAG_OFFSET24, c->off24, c->noff24, op, oend, hufTmp, hufTmpCap);
    if (res < 0) return 1;
    *start += (uint8_t)(res * LZO_FLAG_OFFSET24);
    res = write_stream(huffType & LZO_FLAG_FLAGS, c->flags, c->nflags, op, oend, hufTmp, hufTmpCap);
    if (res < 0) return 1;
    *start += (uint8_t)(res * LZO_FLAG_FLAGS);
    res = write_stream(huffType & LZO_FLAG_LITERALS, c->lit, c->nlit, op, oend, hufTmp, hufTmpCap);
    if (res < 0) return 1;
    *start += (uint8_t)(res * LZO_FLAG_LITERALS);
    if (LZO_BLOCK_GAIN((uint32_t)(*op - start)) > n) goto raw;                   /* :228 */
    return 0;
raw:
    if ((uint32_t)(oend - start) < n + 4) return 1;                               /* :238 */
    *start = LZO_FLAG_UNCOMPRESSED;
    wr24(start + 1, n);
    memcpy(start + 4, ip, n);
    *op = start + 4 + n;
    return 0;
}

/* ---- block entry: reference lib/lizard_compress.c:472-547 (generic) + :583-593 (extState) -------- */
int lzo_compress(const void* srcv, void* dstv, int srcSize, int dstCapacity, int level)
{
    lzo_ctx c;
    const uint8_t* src = (const uint8_t*)srcv;
    uint8_t* dst = (uint8_t*)dstv;
    uint8_t *op, *oend, *hufTmp;
    size_t hufTmpCap;
    uint32_t pos = 0;
    int result = 0;

    /* level clamp: lizard_compress.c:303-308 */
    if (level > 49) level = 49;
    if (level < 10) level = 17;
    if (!lzo_get_params(level, &c.prm)) return 0;
    if (srcSize < 0 || (unsigned)srcSize > (unsigned)LZO_MAX_INPUT) return 0;   /* fast.h:62 */
    if (dstCapacity < 1) return 0;   /* the reference would still store the level byte; never exercised */

    c.src = src;
    c.table = (uint32_t*)calloc((size_t)1 << c.prm.hashLog, sizeof(uint32_t));
    c.chain = (uint32_t*)calloc((size_t)1 << 16, sizeof(uint32_t));
    c.nextToUpdate = 0;
    c.lit   = (uint8_t*)malloc(4 * (size_t)LZO_SUBBLOCK_PAD);
    hufTmpCap = LZO_SUBBLOCK_PAD + (LZO_SUBBLOCK_PAD >> 8) + 8 + 129 + 64;      /* >= HUF_compressBound */
    hufTmp  = (uint8_t*)malloc(hufTmpCap);
    if (!c.table || !c.chain || !c.lit || !hufTmp) goto done;
    c.flags = c.lit + LZO_SUBBLOCK_PAD;
    c.off16 = c.flags + LZO_SUBBLOCK_PAD;
    c.off24 = c.off16 + LZO_SUBBLOCK_PAD;

    op = dst; oend = dst + dstCapacity;
    *op++ = (uint8_t)level;                                                       /* :488 */
    while (pos < (uint32_t)srcSize) {                                             /* :494 */
        uint32_t part = (uint32_t)srcSize - pos;
        if (part > LZO_SUBBLOCK) part = LZO_SUBBLOCK;
        c.nlit = c.nflags = c.noff16 = c.noff24 = 0; c.last_off = 0;              /* Lizard_initBlock :130-138 */
        if (c.prm.parser == P_FAST) parse_fast(&c, pos, pos + part);
        else if (c.prm.parser == P_HASHCHAIN || c.prm.parser == P_NOCHAIN) parse_hashchain(&c, pos, pos + part);
        else parse_pricefast(&c, pos, pos + part);
        if (write_block(&c, src + pos, part, &op, oend, hufTmp, hufTmpCap)) goto done;   /* :535 */
        pos += part;
    }
    result = (int)(op - dst);
done:
    free(c.table); free(c.chain); free(c.lit); free(hufTmp);
    return result;
}

/* ---- synthetic data: reference programs/datagen.c:59-160 ---------------------------------------- */
static uint32_t rdg_rand(uint32_t* s)
{
    uint32_t r = *s;
    r *= 2654435761U; r ^= 2246822519U; r = (r << 13) | (r >> 19);
    *s = r;
    return r;
}

static uint32_t rdg_randlength(uint32_t* s)   /* datagen.c:96 */
{
    if ((rdg_rand(s) >> 7) & 7) return rdg_rand(s) & 15;
    return (rdg_rand(s) & 511) + 15;
}

void lzo_datagen(void* buffer, size_t size, double matchProba, double litProba, unsigned seedIn)
{
    enum { LTSIZE = 1 << 13 };
    uint8_t lt[LTSIZE];
    uint8_t* b = (uint8_t*)buffer;
    uint32_t seed = seedIn;
    size_t pos = 0;
    uint32_t matchProba32 = (uint32_t)(32768 * matchProba);
    if (litProba == 0.0) litProba = matchProba / 4.5;                             /* datagen.c:156 */
    {   /* RDG_fillLiteralDistrib, datagen.c:70-84 */
        uint8_t firstChar = litProba <= 0.0 ? 0 : '(', lastChar = litProba <= 0.0 ? 255 : '}';
        uint8_t ch = litProba <= 0.0 ? 0 : '0';
        uint32_t u = 0;
        while (u < LTSIZE) {
            uint32_t weight = (uint32_t)((double)(LTSIZE - u) * litProba) + 1;
            uint32_t end = u + weight < LTSIZE ? u + weight : LTSIZE;
            while (u < end) lt[u++] = ch;
            ch++;
            if (ch > lastChar) ch = firstChar;
        }
    }
    if (size == 0) return;
    /* RDG_genBlock, datagen.c:97-150 with prefixSize 0 */
    while (matchProba >= 1.0) {
        size_t size0 = rdg_rand(&seed) & 3;
        size0 = (size_t)1 << (16 + size0 * 2);
        size0 += rdg_rand(&seed) & (size0 - 1);
        if (size < pos + size0) { memset(b + pos, 0, size - pos); return; }
        memset(b + pos, 0, size0);
        pos += size0;
        b[pos - 1] = lt[rdg_rand(&seed) & (LTSIZE - 1)];
    }
    if (pos == 0) { b[0] = lt[rdg_rand(&seed) & (LTSIZE - 1)]; pos = 1; }
    while (pos < size) {
        if (((rdg_rand(&seed) >> 3) & 32767) < matchProba32) {
            uint32_t length = rdg_randlength(&seed) + 4;
            uint32_t offset = ((rdg_rand(&seed) >> 3) & 32767) + 1;
            size_t match, d;
            if (offset > pos) offset = (uint32_t)pos;
            match = pos - offset;
            d = pos + length; if (d > size) d = size;
            while (pos < d) b[pos++] = b[match++];
        } else {
            uint32_t length = rdg_randlength(&seed);
            size_t d = pos + length; if (d > size) d = size;
            while (pos < d) b[pos++] = lt[rdg_rand(&seed) & (LTSIZE - 1)];
        }
    }
}
