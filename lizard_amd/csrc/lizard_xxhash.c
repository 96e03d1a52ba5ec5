/*
 * lizard_xxhash.c — XXH32 / XXH64 under the names the reference library exports (lib/xxhash/xxhash.h built with
 * XXH_NAMESPACE=Lizard_, lib/Makefile:52): Lizard_XXH32, Lizard_XXH64 and their streaming forms.  The reference's programs
 * call them directly (programs/bench.c:313 block checksums, tests/fuzzer.c, tests/frametest.c), and the frame layer needs
 * XXH32 for the header and content checksums (lib/lizard_frame.c:219-223, :585-594) — a link-time replacement of liblizard
 * has to bring them.  Written from the xxHash specification (public algorithm: four lanes of
 * multiply-rotate accumulation over 16 / 32-byte stripes, a merge, the tail, an avalanche).
 *
 * The state structs are the CALLER's: programs keep XXH32_state_t / XXH64_state_t of the reference's size on their stack
 * (48 / 88 bytes, lib/xxhash/xxhash.h:257-279), so the layouts here stay inside those sizes.
 */
#include "lizard_xxhash.h"

#include <stdlib.h>
#include <string.h>

#define P32_1 2654435761u
#define P32_2 2246822519u
#define P32_3 3266489917u
#define P32_4 668265263u
#define P32_5 374761393u

#define P64_1 11400714785074694791ull
#define P64_2 14029467366897019727ull
#define P64_3 1609587929392839161ull
#define P64_4 9650029242287828579ull
#define P64_5 2870177450012600261ull

static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

static uint32_t round32(uint32_t acc, uint32_t in) { return rotl32(acc + in * P32_2, 13) * P32_1; }
static uint64_t round64(uint64_t acc, uint64_t in) { return rotl64(acc + in * P64_2, 31) * P64_1; }
static uint64_t merge64(uint64_t h, uint64_t v) { return (h ^ round64(0, v)) * P64_1 + P64_4; }

/* ---- XXH32 ---- */
int Lizard_XXH32_reset(Lizard_XXH32_state_t* s, unsigned seed)
{
    memset(s, 0, sizeof *s);
    s->v[0] = seed + P32_1 + P32_2; s->v[1] = seed + P32_2; s->v[2] = seed; s->v[3] = seed - P32_1;
    return 0;
}

int Lizard_XXH32_update(Lizard_XXH32_state_t* s, const void* input, size_t len)
{
    const uint8_t* p = (const uint8_t*)input;
    if (!input) return len ? 1 : 0;
    s->total32 += (uint32_t)len;
    s->large |= (uint32_t)(len >= 16 || s->total32 >= 16);
    if (s->fill) {
        const size_t take = 16 - s->fill < len ? 16 - s->fill : len;
        memcpy(s->buf + s->fill, p, take);
        s->fill += (uint32_t)take; p += take; len -= take;
        if (s->fill < 16) return 0;
        s->v[0] = round32(s->v[0], rd32(s->buf)); s->v[1] = round32(s->v[1], rd32(s->buf + 4));
        s->v[2] = round32(s->v[2], rd32(s->buf + 8)); s->v[3] = round32(s->v[3], rd32(s->buf + 12));
        s->fill = 0;
    }
    {
        uint32_t v1 = s->v[0], v2 = s->v[1], v3 = s->v[2], v4 = s->v[3];
        while (len >= 16) {
            v1 = round32(v1, rd32(p)); v2 = round32(v2, rd32(p + 4));
            v3 = round32(v3, rd32(p + 8)); v4 = round32(v4, rd32(p + 12));
            p += 16; len -= 16;
        }
        s->v[0] = v1; s->v[1] = v2; s->v[2] = v3; s->v[3] = v4;
    }
    if (len) { memcpy(s->buf, p, len); s->fill = (uint32_t)len; }
    return 0;
}

unsigned Lizard_XXH32_digest(const Lizard_XXH32_state_t* s)
{
    const uint8_t* p = s->buf;
    const uint8_t* const end = p + s->fill;
    uint32_t h = s->large ? rotl32(s->v[0], 1) + rotl32(s->v[1], 7) + rotl32(s->v[2], 12) + rotl32(s->v[3], 18)
                          : s->v[2] /* == seed: no stripe was consumed */ + P32_5;
    h += s->total32;
    while (p + 4 <= end) { h = rotl32(h + rd32(p) * P32_3, 17) * P32_4; p += 4; }
    while (p < end) { h = rotl32(h + (uint32_t)*p * P32_5, 11) * P32_1; p++; }
    h ^= h >> 15; h *= P32_2; h ^= h >> 13; h *= P32_3; h ^= h >> 16;
    return h;
}

unsigned Lizard_XXH32(const void* input, size_t len, unsigned seed)
{
    Lizard_XXH32_state_t s;
    Lizard_XXH32_reset(&s, seed);
    Lizard_XXH32_update(&s, input, len);
    return Lizard_XXH32_digest(&s);
}

Lizard_XXH32_state_t* Lizard_XXH32_createState(void) { return (Lizard_XXH32_state_t*)calloc(1, sizeof(Lizard_XXH32_state_t)); }
int Lizard_XXH32_freeState(Lizard_XXH32_state_t* s) { free(s); return 0; }
void Lizard_XXH32_copyState(Lizard_XXH32_state_t* dst, const Lizard_XXH32_state_t* src) { memcpy(dst, src, sizeof *dst); }
void Lizard_XXH32_canonicalFromHash(unsigned char* dst, unsigned hash)
{ dst[0] = (unsigned char)(hash >> 24); dst[1] = (unsigned char)(hash >> 16); dst[2] = (unsigned char)(hash >> 8); dst[3] = (unsigned char)hash; }
unsigned Lizard_XXH32_hashFromCanonical(const unsigned char* src)
{ return ((unsigned)src[0] << 24) | ((unsigned)src[1] << 16) | ((unsigned)src[2] << 8) | (unsigned)src[3]; }

/* ---- XXH64 ---- */
int Lizard_XXH64_reset(Lizard_XXH64_state_t* s, unsigned long long seed)
{
    memset(s, 0, sizeof *s);
    s->v[0] = seed + P64_1 + P64_2; s->v[1] = seed + P64_2; s->v[2] = seed; s->v[3] = seed - P64_1;
    return 0;
}

int Lizard_XXH64_update(Lizard_XXH64_state_t* s, const void* input, size_t len)
{
    const uint8_t* p = (const uint8_t*)input;
    if (!input) return len ? 1 : 0;
    s->total += len;
    if (s->fill) {
        const size_t take = 32 - s->fill < len ? 32 - s->fill : len;
        memcpy(s->buf + s->fill, p, take);
        s->fill += (uint32_t)take; p += take; len -= take;
        if (s->fill < 32) return 0;
        s->v[0] = round64(s->v[0], rd64(s->buf)); s->v[1] = round64(s->v[1], rd64(s->buf + 8));
        s->v[2] = round64(s->v[2], rd64(s->buf + 16)); s->v[3] = round64(s->v[3], rd64(s->buf + 24));
        s->fill = 0;
    }
    {
        uint64_t v1 = s->v[0], v2 = s->v[1], v3 = s->v[2], v4 = s->v[3];
        while (len >= 32) {
            v1 = round64(v1, rd64(p)); v2 = round64(v2, rd64(p + 8));
            v3 = round64(v3, rd64(p + 16)); v4 = round64(v4, rd64(p + 24));
            p += 32; len -= 32;
        }
        s->v[0] = v1; s->v[1] = v2; s->v[2] = v3; s->v[3] = v4;
    }
    if (len) { memcpy(s->buf, p, len); s->fill = (uint32_t)len; }
    return 0;
}

unsigned long long Lizard_XXH64_digest(const Lizard_XXH64_state_t* s)
{
    const uint8_t* p = s->buf;
    const uint8_t* const end = p + s->fill;
    uint64_t h;
    if (s->total >= 32) {
        h = rotl64(s->v[0], 1) + rotl64(s->v[1], 7) + rotl64(s->v[2], 12) + rotl64(s->v[3], 18);
        h = merge64(h, s->v[0]); h = merge64(h, s->v[1]); h = merge64(h, s->v[2]); h = merge64(h, s->v[3]);
    } else h = s->v[2] /* == seed */ + P64_5;
    h += s->total;
    while (p + 8 <= end) { h ^= round64(0, rd64(p)); h = rotl64(h, 27) * P64_1 + P64_4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P64_1; h = rotl64(h, 23) * P64_2 + P64_3; p += 4; }
    while (p < end) { h ^= (uint64_t)*p * P64_5; h = rotl64(h, 11) * P64_1; p++; }
    h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32;
    return h;
}

unsigned long long Lizard_XXH64(const void* input, size_t len, unsigned long long seed)
{
    Lizard_XXH64_state_t s;
    Lizard_XXH64_reset(&s, seed);
    Lizard_XXH64_update(&s, input, len);
    return Lizard_XXH64_digest(&s);
}

Lizard_XXH64_state_t* Lizard_XXH64_createState(void) { return (Lizard_XXH64_state_t*)calloc(1, sizeof(Lizard_XXH64_state_t)); }
int Lizard_XXH64_freeState(Lizard_XXH64_state_t* s) { free(s); return 0; }
void Lizard_XXH64_copyState(Lizard_XXH64_state_t* dst, const Lizard_XXH64_state_t* src) { memcpy(dst, src, sizeof *dst); }
void Lizard_XXH64_canonicalFromHash(unsigned char* dst, unsigned long long hash)
{ int i; for (i = 0; i < 8; i++) dst[i] = (unsigned char)(hash >> (56 - 8 * i)); }
unsigned long long Lizard_XXH64_hashFromCanonical(const unsigned char* src)
{ unsigned long long h = 0; int i; for (i = 0; i < 8; i++) h = (h << 8) | src[i]; return h; }

unsigned Lizard_XXH_versionNumber(void) { return 0 * 100 * 100 + 6 * 100 + 2; }     /* lib/xxhash/xxhash.h:150-153: 0.6.2 */
