/*
 * oracle/huf_oracle.c — TEST INFRASTRUCTURE ONLY (see lizard_oracle.h).
 *
 * CPU restatement of the huff0 compressor exactly as Lizard uses it:
 *   HUF_compress(dst, cap, src, n) == HUF_compress2(.., 255, 11) -> 4-stream, reference
 *   lib/entropy/huf_compress.c:609,601,593,517.
 * The Lizard caller always provides cap >= HUF_compressBound(n) (lizard_compress.c:148-154), so the
 * "destination too small" paths of the reference are unreachable and are not restated; everything
 * that decides bytes is: histogram, accept/reject heuristics, length-limited tree, canonical codes,
 * weight header (FSE-compressed or raw nibbles), 4 backward bitstreams.
 * Returns: compressed size; 0 = not compressible; 1 = single-symbol RLE; (size_t)-1 = reference error
 * return (the Lizard caller treats it like "not compressible").
 */
#include "lizard_oracle.h"

#include <string.h>

#define HUF_ERR ((size_t)-1)
#define HUF_MAXBITS 12          /* HUF_TABLELOG_MAX, huf.h:118 */
#define HUF_DEFAULT_LOG 11      /* HUF_TABLELOG_DEFAULT, huf.h:119 */

static unsigned highbit(uint32_t v) { unsigned r = 0; while (v >>= 1) r++; return r; }   /* BIT_highbit32 */

/* ---- LSB-first bit writer. The reference's BIT_CStream (bitstream.h:185-248) flushes whole bytes of a
 * 64-bit container; its cadence is unobservable, the bytes are the little-endian serialisation of the
 * appended bit string, closed by a single '1' bit. */
typedef struct { uint8_t* p; size_t pos; uint64_t acc; unsigned nb; } bitw;
static void bw_init(bitw* b, uint8_t* p) { b->p = p; b->pos = 0; b->acc = 0; b->nb = 0; }
static void bw_add(bitw* b, uint32_t v, unsigned n)
{
    if (n == 0) return;
    b->acc |= (uint64_t)(v & (uint32_t)((1ull << n) - 1)) << b->nb;
    b->nb += n;
    while (b->nb >= 8) { b->p[b->pos++] = (uint8_t)b->acc; b->acc >>= 8; b->nb -= 8; }
}
static size_t bw_finish(bitw* b) { if (b->nb) { b->p[b->pos++] = (uint8_t)b->acc; b->acc = 0; b->nb = 0; } return b->pos; }
static size_t bw_close(bitw* b) { bw_add(b, 1, 1); return bw_finish(b); }     /* BIT_closeCStream, bitstream.h:240 */

/* ---- FSE pieces, used only for the <=255 Huffman weights (alphabet 0..12) --------------------------- */

/* FSE_optimalTableLog_internal, fse_compress.c:477-496 */
static unsigned fse_optimal_tablelog(unsigned maxTableLog, size_t srcSize, unsigned maxSym, unsigned minus)
{
    unsigned maxBitsSrc = highbit((uint32_t)(srcSize - 1)) - minus;
    unsigned minBitsSrc = highbit((uint32_t)(srcSize - 1)) + 1, minBitsSym = highbit(maxSym) + 2;
    unsigned minBits = minBitsSrc < minBitsSym ? minBitsSrc : minBitsSym;
    unsigned tl = maxTableLog ? maxTableLog : 11;
    if (maxBitsSrc < tl) tl = maxBitsSrc;
    if (minBits > tl) tl = minBits;
    if (tl < 5) tl = 5;
    if (tl > 12) tl = 12;
    return tl;
}

/* FSE_normalizeM2, fse_compress.c:507-579. Returns 0 ok, -1 error. */
static int fse_normalize_m2(short* norm, unsigned tableLog, const unsigned* count, size_t total, unsigned maxSym)
{
    unsigned s, distributed = 0, toDistribute;
    uint32_t lowThreshold = (uint32_t)(total >> tableLog);
    uint32_t lowOne = (uint32_t)((total * 3) >> (tableLog + 1));
    for (s = 0; s <= maxSym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = -1; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = -2;
    }
    toDistribute = (1u << tableLog) - distributed;
    if ((total / toDistribute) > lowOne) {
        lowOne = (uint32_t)((total * 3) / (toDistribute * 2));
        for (s = 0; s <= maxSym; s++)
            if (norm[s] == -2 && count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; }
        toDistribute = (1u << tableLog) - distributed;
    }
    if (distributed == maxSym + 1) {
        unsigned maxV = 0, maxC = 0;
        for (s = 0; s <= maxSym; s++) if (count[s] > maxC) { maxV = s; maxC = count[s]; }
        norm[maxV] += (short)toDistribute;
        return 0;
    }
    {
        uint64_t vStepLog = 62 - tableLog, mid = (1ULL << (vStepLog - 1)) - 1;
        uint64_t rStep = ((((uint64_t)1 << vStepLog) * toDistribute) + mid) / total;
        uint64_t tmpTotal = mid;
        for (s = 0; s <= maxSym; s++) if (norm[s] == -2) {
            uint64_t end = tmpTotal + (count[s] * rStep);
            uint32_t weight = (uint32_t)(end >> vStepLog) - (uint32_t)(tmpTotal >> vStepLog);
            if (weight < 1) return -1;
            norm[s] = (short)weight;
            tmpTotal = end;
        }
    }
    return 0;
}

/* FSE_normalizeCount, fse_compress.c:582-641. Returns 0 ok, -1 error. (The "count == total" RLE early
 * return cannot trigger here: the caller has already handled maxCount == total.) */
static int fse_normalize(short* norm, unsigned tableLog, const unsigned* count, size_t total, unsigned maxSym)
{
    static const uint32_t rtb[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    uint64_t scale = 62 - tableLog, step = ((uint64_t)1 << 62) / total, vStep = 1ULL << (scale - 20);
    int still = 1 << tableLog;
    unsigned s, largest = 0;
    short largestP = 0;
    uint32_t lowThreshold = (uint32_t)(total >> tableLog);
    {   /* :589: tableLog must be able to represent the distribution */
        unsigned minBitsSrc = highbit((uint32_t)(total - 1)) + 1, minBitsSym = highbit(maxSym) + 2;
        if (tableLog < (minBitsSrc < minBitsSym ? minBitsSrc : minBitsSym)) return -1;
    }
    for (s = 0; s <= maxSym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = -1; still--; }
        else {
            short proba = (short)((count[s] * step) >> scale);
            if (proba < 8) {
                uint64_t restToBeat = vStep * rtb[proba];
                proba += (count[s] * step) - ((uint64_t)proba << scale) > restToBeat;
            }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm[s] = proba;
            still -= proba;
        }
    }
    if (-still >= (norm[largest] >> 1)) return fse_normalize_m2(norm, tableLog, count, total, maxSym);
    norm[largest] += (short)still;
    return 0;
}

/* FSE_writeNCount_generic, fse_compress.c:204-279 as an append-only bit string. Returns bytes, or 0 on
 * the reference's error returns. */
static size_t fse_write_ncount(uint8_t* out, const short* norm, unsigned maxSym, unsigned tableLog)
{
    bitw b;
    int nbBits = (int)tableLog + 1, remaining = (1 << tableLog) + 1, threshold = 1 << tableLog;
    unsigned charnum = 0;
    int previous0 = 0;
    bw_init(&b, out);
    bw_add(&b, tableLog - 5, 4);
    while (remaining > 1) {
        if (previous0) {
            unsigned start = charnum;
            while (!norm[charnum]) charnum++;
            while (charnum >= start + 24) { start += 24; bw_add(&b, 0xFFFF, 16); }
            while (charnum >= start + 3) { start += 3; bw_add(&b, 3, 2); }
            bw_add(&b, charnum - start, 2);
        }
        {
            int count = norm[charnum++];
            int max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bw_add(&b, (uint32_t)count, (unsigned)(nbBits - (count < max)));
            previous0 = (count == 1);
            if (remaining < 1) return 0;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
    }
    if (charnum > maxSym + 1) return 0;
    /* :281-284: the final flush always stores two bytes but advances by ceil(bitCount/8) */
    return bw_finish(&b);
}

/* HUF_compressWeights, huf_compress.c:81-121: FSE-compress the weight array. Returns compressed size,
 * 0 (not compressible), 1 (all weights equal), HUF_ERR. */
static size_t huf_compress_weights(uint8_t* dst, const uint8_t* wt, size_t wtSize)
{
    unsigned count[13], maxSym = HUF_MAXBITS, maxCount = 0, s, tableLog;
    short norm[13];
    uint16_t stateTable[64];
    int dFind[13]; uint32_t dBits[13];
    size_t pos;
    if (wtSize <= 1) return 0;
    memset(count, 0, sizeof count);
    for (s = 0; s < wtSize; s++) count[wt[s]]++;                    /* FSE_count_simple, fse_compress.c:315 */
    while (!count[maxSym]) maxSym--;
    for (s = 0; s <= maxSym; s++) if (count[s] > maxCount) maxCount = count[s];
    if (maxCount == wtSize) return 1;
    if (maxCount == 1) return 0;
    tableLog = fse_optimal_tablelog(6, wtSize, maxSym, 2);          /* MAX_FSE_TABLELOG_FOR_HUFF_HEADER */
    if (fse_normalize(norm, tableLog, count, wtSize, maxSym)) return HUF_ERR;
    pos = fse_write_ncount(dst, norm, maxSym, tableLog);
    if (pos == 0) return HUF_ERR;
    {   /* FSE_buildCTable_wksp, fse_compress.c:103-182 */
        const unsigned tableSize = 1u << tableLog, mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
        uint8_t tableSymbol[64];
        unsigned cumul[15], high = tableSize - 1, position = 0, u, total = 0;
        cumul[0] = 0;
        for (u = 1; u <= maxSym + 1; u++) {
            if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; tableSymbol[high--] = (uint8_t)(u - 1); }
            else cumul[u] = cumul[u - 1] + (unsigned)norm[u - 1];
        }
        for (s = 0; s <= maxSym; s++) {
            int k;
            for (k = 0; k < norm[s]; k++) {
                tableSymbol[position] = (uint8_t)s;
                position = (position + step) & mask;
                while (position > high) position = (position + step) & mask;
            }
        }
        if (position != 0) return HUF_ERR;
        for (u = 0; u < tableSize; u++) { unsigned sy = tableSymbol[u]; stateTable[cumul[sy]++] = (uint16_t)(tableSize + u); }
        for (s = 0; s <= maxSym; s++) {
            if (norm[s] == 0) { dBits[s] = 0; dFind[s] = 0; }
            else if (norm[s] == -1 || norm[s] == 1) {
                dBits[s] = (tableLog << 16) - (1u << tableLog); dFind[s] = (int)total - 1; total++;
            } else {
                unsigned maxBitsOut = tableLog - highbit((uint32_t)norm[s] - 1);
                unsigned minStatePlus = (unsigned)norm[s] << maxBitsOut;
                dBits[s] = (maxBitsOut << 16) - minStatePlus; dFind[s] = (int)total - norm[s]; total += (unsigned)norm[s];
            }
        }
    }
    {   /* FSE_compress_usingCTable_generic, fse_compress.c:701-758 + state ops fse.h:525-564:
         * two interleaved states walking the weights from the end */
        bitw b;
        int64_t st[2];        /* st[0] = CState1, st[1] = CState2 */
        size_t i = wtSize, csize;
        int which;
        if (wtSize <= 2) return 0;
#define FSE_INIT2(S, sym) do { uint32_t nbo = (dBits[sym] + (1u << 15)) >> 16; int64_t v = ((int64_t)nbo << 16) - dBits[sym]; \
                               (S) = stateTable[(v >> nbo) + dFind[sym]]; } while (0)
#define FSE_ENC(S, sym) do { uint32_t nbo = (uint32_t)(((S) + dBits[sym]) >> 16); bw_add(&b, (uint32_t)(S), nbo); \
                             (S) = stateTable[((S) >> nbo) + dFind[sym]]; } while (0)
        bw_init(&b, dst + pos);
        if (wtSize & 1) { FSE_INIT2(st[0], wt[i - 1]); FSE_INIT2(st[1], wt[i - 2]); i -= 2; i--; FSE_ENC(st[0], wt[i]); }
        else            { FSE_INIT2(st[1], wt[i - 1]); FSE_INIT2(st[0], wt[i - 2]); i -= 2; }
        which = 1;                                           /* the rest alternates CState2, CState1, ... */
        while (i > 0) { i--; FSE_ENC(st[which], wt[i]); which ^= 1; }
        bw_add(&b, (uint32_t)st[1], tableLog);               /* FSE_flushCState(CState2), then CState1 */
        bw_add(&b, (uint32_t)st[0], tableLog);
        csize = bw_close(&b);
#undef FSE_INIT2
#undef FSE_ENC
        return pos + csize;
    }
}

/* ---- Huffman tree: HUF_buildCTable_wksp, huf_compress.c:334-402 ---------------------------------- */
typedef struct { uint32_t count; uint16_t parent; uint8_t byte, nbBits; } hnode;

/* HUF_setMaxHeight, huf_compress.c:223-297 */
static unsigned huf_set_max_height(hnode* node, unsigned lastNonNull, unsigned maxNbBits)
{
    const unsigned largestBits = node[lastNonNull].nbBits;
    int totalCost = 0;
    unsigned baseCost, rankLast[HUF_MAXBITS + 2];
    const unsigned noSymbol = 0xF0F0F0F0u;
    int n = (int)lastNonNull, pos;
    if (largestBits <= maxNbBits) return largestBits;
    baseCost = 1u << (largestBits - maxNbBits);
    while (node[n].nbBits > maxNbBits) {
        totalCost += (int)(baseCost - (1u << (largestBits - node[n].nbBits)));
        node[n].nbBits = (uint8_t)maxNbBits;
        n--;
    }
    while (node[n].nbBits == maxNbBits) n--;
    totalCost >>= (largestBits - maxNbBits);
    for (pos = 0; pos < HUF_MAXBITS + 2; pos++) rankLast[pos] = noSymbol;
    {
        unsigned cur = maxNbBits;
        for (pos = n; pos >= 0; pos--) {
            if (node[pos].nbBits >= cur) continue;
            cur = node[pos].nbBits;
            rankLast[maxNbBits - cur] = (unsigned)pos;
        }
    }
    while (totalCost > 0) {
        unsigned dec = highbit((uint32_t)totalCost) + 1;
        for (; dec > 1; dec--) {
            unsigned highPos = rankLast[dec], lowPos = rankLast[dec - 1];
            if (highPos == noSymbol) continue;
            if (lowPos == noSymbol) break;
            if (node[highPos].count <= 2 * node[lowPos].count) break;
        }
        while (dec <= HUF_MAXBITS && rankLast[dec] == noSymbol) dec++;
        totalCost -= 1 << (dec - 1);
        if (rankLast[dec - 1] == noSymbol) rankLast[dec - 1] = rankLast[dec];
        node[rankLast[dec]].nbBits++;
        if (rankLast[dec] == 0) rankLast[dec] = noSymbol;
        else {
            rankLast[dec]--;
            if (node[rankLast[dec]].nbBits != maxNbBits - dec) rankLast[dec] = noSymbol;
        }
    }
    while (totalCost < 0) {
        if (rankLast[1] == noSymbol) {
            while (node[n].nbBits == maxNbBits) n--;
            node[n + 1].nbBits--;
            rankLast[1] = (unsigned)(n + 1);
            totalCost++;
            continue;
        }
        node[rankLast[1] + 1].nbBits--;
        rankLast[1]++;
        totalCost++;
    }
    return maxNbBits;
}

/* Fills nbBits[256]/val[256]; returns the maximum code length used. */
static unsigned huf_build(uint8_t* nbBits, uint16_t* val, const unsigned* count, unsigned maxSym, unsigned maxNbBits)
{
    enum { START = 256 };
    hnode store[2 * 256 + 2];
    hnode* node = store + 1;                       /* node[-1] is the barrier, :351 */
    unsigned n, nonNull, nodeNb = START, nodeRoot;
    int lowS, lowN;
    memset(store, 0, sizeof store);
    {   /* HUF_sort, :305-325: descending by count, ties keep ascending symbol order */
        struct { unsigned base, cur; } rank[32];
        memset(rank, 0, sizeof rank);
        for (n = 0; n <= maxSym; n++) rank[highbit(count[n] + 1)].base++;
        for (n = 30; n > 0; n--) rank[n - 1].base += rank[n].base;
        for (n = 0; n < 32; n++) rank[n].cur = rank[n].base;
        for (n = 0; n <= maxSym; n++) {
            unsigned c = count[n], r = highbit(c + 1) + 1, pos = rank[r].cur++;
            while (pos > rank[r].base && c > node[pos - 1].count) { node[pos] = node[pos - 1]; pos--; }
            node[pos].count = c; node[pos].byte = (uint8_t)n;
        }
    }
    nonNull = maxSym;
    while (node[nonNull].count == 0) nonNull--;
    lowS = (int)nonNull; nodeRoot = nodeNb + (unsigned)lowS - 1; lowN = (int)nodeNb;
    node[nodeNb].count = node[lowS].count + node[lowS - 1].count;
    node[lowS].parent = node[lowS - 1].parent = (uint16_t)nodeNb;
    nodeNb++; lowS -= 2;
    for (n = nodeNb; n <= nodeRoot; n++) node[n].count = 1u << 30;
    store[0].count = 1u << 31;
    while (nodeNb <= nodeRoot) {                   /* :363-369: ties go to the internal node */
        int n1 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        int n2 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        node[nodeNb].count = node[n1].count + node[n2].count;
        node[n1].parent = node[n2].parent = (uint16_t)nodeNb;
        nodeNb++;
    }
    node[nodeRoot].nbBits = 0;
    for (n = nodeRoot - 1; n >= START; n--) node[n].nbBits = (uint8_t)(node[node[n].parent].nbBits + 1);
    for (n = 0; n <= nonNull; n++) node[n].nbBits = (uint8_t)(node[node[n].parent].nbBits + 1);
    maxNbBits = huf_set_max_height(node, nonNull, maxNbBits);
    {   /* canonical values, :381-398 */
        uint16_t nbPerRank[HUF_MAXBITS + 1] = { 0 }, valPerRank[HUF_MAXBITS + 1] = { 0 }, min = 0;
        for (n = 0; n <= nonNull; n++) nbPerRank[node[n].nbBits]++;
        for (n = maxNbBits; n > 0; n--) { valPerRank[n] = min; min = (uint16_t)(min + nbPerRank[n]); min >>= 1; }
        for (n = 0; n <= maxSym; n++) nbBits[node[n].byte] = node[n].nbBits;
        for (n = 0; n <= maxSym; n++) val[n] = valPerRank[nbBits[n]]++;
    }
    return maxNbBits;
}

/* HUF_compress1X_usingCTable, huf_compress.c:427-470: symbols appended last -> first. */
static size_t huf_encode_1x(uint8_t* dst, const uint8_t* src, size_t n, const uint8_t* nbBits, const uint16_t* val)
{
    bitw b;
    bw_init(&b, dst);
    while (n > 0) { n--; bw_add(&b, val[src[n]], nbBits[src[n]]); }
    return bw_close(&b);
}

size_t lzo_huf_compress(void* dstv, size_t dstCapacity, const void* srcv, size_t n)
{
    const uint8_t* src = (const uint8_t*)srcv;
    uint8_t* dst = (uint8_t*)dstv;
    unsigned count[256], maxSym = 255, largest = 0, s, huffLog;
    uint8_t nbBits[256];
    uint16_t val[256];
    size_t hSize, pos;
    if (!dst) return 0;                                               /* probe used by lzo_level_supported */
    if (n == 0 || dstCapacity == 0) return 0;                         /* huf_compress.c:534-535 */
    if (n > 128 * 1024) return HUF_ERR;                               /* HUF_BLOCKSIZE_MAX, :536 */
    if (dstCapacity < n + (n >> 8) + 8 + 129) return HUF_ERR;         /* oracle contract: cap >= HUF_compressBound */
    memset(count, 0, sizeof count);
    for (pos = 0; pos < n; pos++) count[src[pos]]++;                   /* FSE_count_wksp, fse_compress.c:431 */
    while (!count[maxSym]) maxSym--;
    for (s = 0; s <= maxSym; s++) if (count[s] > largest) largest = count[s];
    if (largest == n) { dst[0] = src[0]; return 1; }                   /* :544 */
    if (largest <= (n >> 7) + 1) return 0;                            /* :545 */
    huffLog = fse_optimal_tablelog(HUF_DEFAULT_LOG, n, maxSym, 1);    /* HUF_optimalTableLog, :66 */
    memset(nbBits, 0, sizeof nbBits);
    huffLog = huf_build(nbBits, val, count, maxSym, huffLog);          /* :550-551 */
    {   /* HUF_writeCTable, huf_compress.c:132-165 */
        uint8_t wt[256];
        size_t h;
        for (s = 0; s < maxSym; s++) wt[s] = nbBits[s] ? (uint8_t)(huffLog + 1 - nbBits[s]) : 0;
        h = huf_compress_weights(dst + 1, wt, maxSym);
        if (h == HUF_ERR) return HUF_ERR;
        if (h > 1 && h < maxSym / 2) { dst[0] = (uint8_t)h; hSize = h + 1; }
        else {
            if (maxSym > 128) return HUF_ERR;                          /* :158 */
            dst[0] = (uint8_t)(128 + (maxSym - 1));
            wt[maxSym] = 0;
            for (s = 0; s < maxSym; s += 2) dst[s / 2 + 1] = (uint8_t)((wt[s] << 4) + wt[s + 1]);
            hSize = (maxSym + 1) / 2 + 1;
        }
    }
    if (hSize + 12 >= n) return 0;                                    /* :556 */
    pos = hSize;
    {   /* HUF_compress4X_usingCTable, :473-513 */
        const size_t seg = (n + 3) / 4;
        uint8_t* jump = dst + pos;
        size_t c, off = 0;
        int k;
        if (n < 12) return 0;
        pos += 6;
        for (k = 0; k < 4; k++) {
            size_t len = k < 3 ? seg : n - 3 * seg;
            c = huf_encode_1x(dst + pos, src + off, len, nbBits, val);
            if (k < 3) { jump[2 * k] = (uint8_t)c; jump[2 * k + 1] = (uint8_t)(c >> 8); }
            pos += c; off += len;
        }
    }
    if (pos >= n - 1) return 0;                                       /* :570 */
    return pos;
}
