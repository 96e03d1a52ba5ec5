// tools/lz_datagen.h — deterministic synthetic-input generator used by the benchmark and the tests (tooling, not in the product library).
//
// Restates the *behaviour* of the reference's data generator (programs/datagen.c:59-160,
// RDG_genBuffer / RDG_genBlock with prefixSize 0) so that measurements run on exactly the byte
// distribution BASELINE.json names ("datagen P50"), and so that the known answers of SURVEY.md §8c can
// be reproduced.  One generator call is one serial LCG walk, so the device version runs ONE THREAD PER
// BLOCK (block b of a batch is RDG_genBuffer(blockSize, P, seed0 + b)); the integer core below is
// shared verbatim by the host entry point and the device kernel, the only floating-point step (the
// literal distribution table, datagen.c:70-84) is done once on the host.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LZ_HD __host__ __device__ inline
#else
#define LZ_HD static inline
#endif

#define LZ_RDG_LTSIZE 8192u   /* LTLOG 13, datagen.c:46-48 */

LZ_HD uint32_t lz_rdg_rand(uint32_t* s)                       /* datagen.c:59-67 */
{
    uint32_t r = *s;
    r *= 2654435761U; r ^= 2246822519U; r = (r << 13) | (r >> 19);
    *s = r;
    return r;
}

LZ_HD uint32_t lz_rdg_randlength(uint32_t* s)                 /* RDG_RANDLENGTH, datagen.c:96 */
{
    if ((lz_rdg_rand(s) >> 7) & 7) return lz_rdg_rand(s) & 15;
    return (lz_rdg_rand(s) & 511) + 15;
}

/* RDG_genBlock with prefixSize 0 (datagen.c:97-150); matchProba32 = (U32)(32768*matchProba),
 * zeroRuns = (matchProba >= 1.0). */
LZ_HD void lz_rdg_fill(uint8_t* b, size_t size, uint32_t matchProba32, int zeroRuns, const uint8_t* lt, uint32_t seed)
{
    size_t pos = 0;
    if (size == 0) return;
    while (zeroRuns) {                                         /* datagen.c:105-118 */
        size_t size0 = lz_rdg_rand(&seed) & 3;
        size0 = (size_t)1 << (16 + size0 * 2);
        size0 += lz_rdg_rand(&seed) & (size0 - 1);
        if (size < pos + size0) { for (; pos < size; pos++) b[pos] = 0; return; }
        for (size_t e = pos + size0; pos < e; pos++) b[pos] = 0;
        b[pos - 1] = lt[lz_rdg_rand(&seed) & (LZ_RDG_LTSIZE - 1)];
    }
    b[0] = lt[lz_rdg_rand(&seed) & (LZ_RDG_LTSIZE - 1)]; pos = 1;   /* datagen.c:121 */
    while (pos < size) {
        if (((lz_rdg_rand(&seed) >> 3) & 32767) < matchProba32) {     /* copy within 32 KiB, :127-138 */
            uint32_t length = lz_rdg_randlength(&seed) + 4;
            uint32_t offset = ((lz_rdg_rand(&seed) >> 3) & 32767) + 1;
            size_t match, d;
            if (offset > pos) offset = (uint32_t)pos;
            match = pos - offset;
            d = pos + length; if (d > size) d = size;
            while (pos < d) b[pos++] = b[match++];
        } else {                                                      /* literal noise, :139-147 */
            uint32_t length = lz_rdg_randlength(&seed);
            size_t d = pos + length; if (d > size) d = size;
            while (pos < d) b[pos++] = lt[lz_rdg_rand(&seed) & (LZ_RDG_LTSIZE - 1)];
        }
    }
}

/* RDG_fillLiteralDistrib (datagen.c:70-84) + the litProba default of RDG_genBuffer (:156). Host only. */
static inline void lz_rdg_table(uint8_t* lt, double matchProba, double litProba)
{
    double ld = litProba == 0.0 ? matchProba / 4.5 : litProba;
    uint8_t firstChar = ld <= 0.0 ? 0 : '(', lastChar = ld <= 0.0 ? 255 : '}';
    uint8_t ch = ld <= 0.0 ? 0 : '0';
    uint32_t u = 0;
    while (u < LZ_RDG_LTSIZE) {
        uint32_t weight = (uint32_t)((double)(LZ_RDG_LTSIZE - u) * ld) + 1;
        uint32_t end = u + weight < LZ_RDG_LTSIZE ? u + weight : LZ_RDG_LTSIZE;
        while (u < end) lt[u++] = ch;
        ch++;
        if (ch > lastChar) ch = firstChar;
    }
}
