/* tests/decode_fuzz.c — TEST INFRASTRUCTURE: the host half of the drop-in ABI (lizard_amd/csrc/lizard_decode_host.c,
 * lizard_frame_host.c, lizard_xxhash.c) compiled with AddressSanitizer + UndefinedBehaviorSanitizer and fed valid and damaged
 * input in exact-size heap buffers, so that a read or write ONE byte outside a buffer traps (the canaries of
 * tests/test_decode_host.py only see writes).  No GPU and no HIP: the three sources are host C; the two GPU hooks the frame
 * layer calls are stubbed below (every block is then stored raw — the frame DEcoder and the raw path are what is fuzzed;
 * compressed blocks come from the oracle).
 *
 *   usage: decode_fuzz <seed> <cases>
 *   build: gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all tests/decode_fuzz.c lizard_amd/csrc/lizard_decode_host.c
 *              lizard_amd/csrc/lizard_frame_host.c lizard_amd/csrc/lizard_xxhash.c -Iinclude -Ioracle -Loracle -llizard_oracle -lpthread
 * Exit 0: every case behaved (valid input decoded exactly; damaged input refused or decoded to SOMETHING without touching a byte
 * outside its buffers).  A sanitizer report aborts the process. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lizard_amd.h"
#include "lizard_oracle.h"

/* ---- the GPU side of the library, absent here ---- */
int lzgpu_frame_records(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize, void* dst, size_t dstCapacity, size_t* written, int level)
{ (void)src; (void)nBlocks; (void)blockSize; (void)lastBlockSize; (void)dst; (void)dstCapacity; (void)written; (void)level; return -1; }
int LizardGPU_levelSupported(int level) { (void)level; return 0; }
const char* LizardGPU_lastError(void) { return "decode_fuzz: no device"; }
void lzgpu_note_degraded(const char* what, int level) { (void)what; (void)level; }

static unsigned g_rng;
static unsigned rnd(void) { g_rng = g_rng * 1664525u + 1013904223u; return g_rng >> 8; }

static unsigned char* exact(const void* p, size_t n) { unsigned char* q = malloc(n ? n : 1); if (n) memcpy(q, p, n); return q; }

static void damage(unsigned char* p, size_t n)
{
    const unsigned kind = rnd() % 3;
    if (!n) return;
    if (kind == 0) { unsigned k = 1 + rnd() % 3; while (k--) p[rnd() % n] ^= (unsigned char)(1u << (rnd() % 8)); }
    else if (kind == 1) { size_t at = rnd() % n, len = 1 + rnd() % 8; for (size_t i = at; i < at + len && i < n; i++) p[i] = (unsigned char)rnd(); }
    else p[rnd() % n] = (unsigned char)(rnd() % 3 == 0 ? 0xFF : 0x00);
}

static int fuzz_block(const unsigned char* data, int n, int level)
{
    const int bound = lzo_compress_bound(n);
    unsigned char* comp = malloc((size_t)bound);
    const int c = lzo_compress(data, comp, n, bound, level);
    int bad = 0;
    if (c <= 0) { free(comp); return 1; }
    {   /* valid: every entry point */
        unsigned char* src = exact(comp, (size_t)c);
        unsigned char* out = malloc((size_t)n ? (size_t)n : 1);
        if (Lizard_decompress_safe((char*)src, (char*)out, c, n) != n || memcmp(out, data, (size_t)n)) bad = 1;
        if (n > 40 && Lizard_decompress_safe((char*)src, (char*)out, c, n - 1) >= 0) bad = 1;
        { const int t = n ? (int)(rnd() % (unsigned)n) : 0; const int r = Lizard_decompress_safe_partial((char*)src, (char*)out, c, t, n); if (r < t || r > n || memcmp(out, data, (size_t)t)) bad = 1; }
        {   /* a dictionary that the block does not need, as prefix-less external memory */
            unsigned char* dict = malloc(1000);
            memset(dict, 7, 1000);
            if (Lizard_decompress_safe_usingDict((char*)src, (char*)out, c, n, (char*)dict, 1000) != n || memcmp(out, data, (size_t)n)) bad = 1;
            if (Lizard_decompress_safe_forceExtDict((char*)src, (char*)out, c, n, (char*)dict, 1000) != n) bad = 1;
            free(dict);
        }
        free(src); free(out);
    }
    for (int k = 0; k < 24; k++) {   /* damaged / truncated / extended, exact-size buffers: the sanitizer is the judge */
        size_t len = (size_t)c;
        unsigned char* src;
        const unsigned how = rnd() % 4;
        if (how == 0 && c > 1) len = 1 + rnd() % ((unsigned)c - 1);
        else if (how == 1) len = (size_t)c + 1 + rnd() % 16;
        src = malloc(len);
        memcpy(src, comp, len < (size_t)c ? len : (size_t)c);
        for (size_t i = (size_t)c; i < len; i++) src[i] = (unsigned char)rnd();
        if (how >= 2) damage(src, len);
        {
            int cap = n + ((int)(rnd() % 3) - 1) * 50;
            unsigned char* out;
            if (cap < 1) cap = 1;
            out = malloc((size_t)cap);
            const int r = Lizard_decompress_safe((char*)src, (char*)out, (int)len, cap);
            if (r > cap) bad = 1;
            (void)Lizard_decompress_safe_partial((char*)src, (char*)out, (int)len, cap / 2, cap);
            free(out);
        }
        free(src);
    }
    free(comp);
    return bad;
}

/* a frame made of oracle-compressed blocks (independent mode) or raw blocks, decoded in random pieces; then damaged */
static size_t make_frame(const unsigned char* data, size_t n, int level, unsigned bsid, int crc, unsigned char* dst)
{
    static const size_t sizes[8] = { 131072, 131072, 262144, 1u << 20, 4u << 20, 0, 0, 0 };
    const size_t bs = sizes[bsid];
    unsigned char* p = dst;
    unsigned hdr[2];
    p[0] = 0x06; p[1] = 0x22; p[2] = 0x4D; p[3] = 0x18;
    p[4] = (unsigned char)((1u << 6) | (1u << 5) | ((unsigned)crc << 2)); p[5] = (unsigned char)(bsid << 4);
    p[6] = (unsigned char)(Lizard_XXH32(p + 4, 2, 0) >> 8);
    p += 7;
    (void)hdr;
    for (size_t off = 0; off < n; off += bs) {
        const size_t len = n - off < bs ? n - off : bs;
        const int c = (rnd() % 4) ? lzo_compress(data + off, p + 4, (int)len, (int)len - 1 > 0 ? (int)len - 1 : 0, level) : 0;
        unsigned word;
        if (c > 0) word = (unsigned)c; else { word = (unsigned)len | 0x80000000u; memcpy(p + 4, data + off, len); }
        p[0] = (unsigned char)word; p[1] = (unsigned char)(word >> 8); p[2] = (unsigned char)(word >> 16); p[3] = (unsigned char)(word >> 24);
        p += 4 + (word & 0x7FFFFFFFu);
    }
    memset(p, 0, 4); p += 4;
    if (crc) { const unsigned x = Lizard_XXH32(data, n, 0); p[0] = (unsigned char)x; p[1] = (unsigned char)(x >> 8); p[2] = (unsigned char)(x >> 16); p[3] = (unsigned char)(x >> 24); p += 4; }
    return (size_t)(p - dst);
}

static int decode_frame(const unsigned char* frame, size_t flen, const unsigned char* want, size_t n, int mustWork)
{
    LizardF_decompressionContext_t d;
    size_t so = 0, doo = 0, r = 1;
    unsigned char* got = malloc(n + 1);
    int bad = 0, guard = 0;
    if (LizardF_createDecompressionContext(&d, LIZARDF_VERSION)) return 1;
    while ((so < flen || r != 0) && !bad) {
        size_t in = flen - so, used = 0;
        unsigned char* src;
        int stop = 0;
        if (in) { in = 1 + rnd() % in; if (rnd() % 3 == 0 && in > 9) in = 1 + rnd() % 9; }
        src = exact(frame + so, in);                     /* this piece of input, exact size; the decoder is called on it until it is
                                                            used up (it must be handed the SAME buffer again where it stopped, :1004-1006) */
        do {
            size_t cap = 1 + rnd() % (rnd() % 2 ? 70000u : 300000u), ds, ss = in - used;
            unsigned char* dst = malloc(cap);
            ds = cap;
            r = LizardF_decompress(d, dst, &ds, src + used, &ss, NULL);
            if (LizardF_isError(r)) { bad = mustWork; stop = 1; }
            else if (ss > in - used || ds > cap) { bad = 1; stop = 1; }
            else {
                if (doo + ds <= n) memcpy(got + doo, dst, ds); else if (mustWork) bad = 1;
                doo += ds; used += ss;
                if (ss == 0 && ds == 0 && ++guard > 1000) { bad = mustWork; stop = 1; }      /* no progress: a truncated frame keeps asking */
            }
            free(dst);
        } while (!stop && (used < in || (r != 0 && in == 0 && so == flen)));
        so += used;
        free(src);
        if (stop) break;
        if (r == 0 && so == flen) break;
        if (so == flen && r != 0 && in == 0) { bad = mustWork; break; }
    }
    if (mustWork && !bad && (doo != n || memcmp(got, want, n))) bad = 1;
    LizardF_freeDecompressionContext(d);
    free(got);
    return bad;
}

int main(int argc, char** argv)
{
    const unsigned seed = argc > 1 ? (unsigned)strtoul(argv[1], NULL, 10) : 1u;
    const int cases = argc > 2 ? atoi(argv[2]) : 200;
    static const int levels[] = { 10, 11, 13, 17, 21, 22, 30, 31, 35, 41, 42 };
    int bad = 0;
    g_rng = seed * 2654435761u + 1u;
    for (int k = 0; k < cases; k++) {
        const int n = (rnd() % 4 == 0) ? (int)(rnd() % 64) : (int)(rnd() % 200000);
        const int level = levels[rnd() % (sizeof levels / sizeof *levels)];
        unsigned char* data = malloc((size_t)n + 1);
        lzo_datagen(data, (size_t)n, (rnd() % 11) / 10.0, 0.0, rnd());
        if (rnd() % 5 == 0 && n > 100) memset(data + n / 3, data[0], (size_t)n / 4);         /* a run */
        if (n && fuzz_block(data, n, level)) { fprintf(stderr, "decode_fuzz: block case %d (n %d, level %d) misbehaved\n", k, n, level); bad++; }
        if (k % 4 == 0) {
            const size_t fn = (size_t)n * 3 + 5;
            unsigned char* big = malloc(fn);
            unsigned char* frame = malloc(fn + fn / 2 + 4096);
            size_t flen;
            for (size_t i = 0; i < fn; i++) big[i] = data[n ? i % (size_t)n : 0] ^ (unsigned char)(i >> 12);
            flen = make_frame(big, fn, level, 1 + rnd() % 2, (int)(rnd() % 2), frame);
            if (decode_frame(frame, flen, big, fn, 1)) { fprintf(stderr, "decode_fuzz: frame case %d misbehaved\n", k); bad++; }
            for (int j = 0; j < 6; j++) {
                unsigned char* f2 = exact(frame, flen);
                size_t l2 = flen;
                if (rnd() % 3 == 0) l2 = 1 + rnd() % flen; else damage(f2, flen);
                (void)decode_frame(f2, l2, big, fn, 0);
                free(f2);
            }
            free(big); free(frame);
        }
        {   /* xxhash, streaming in pieces vs one shot */
            struct { unsigned long long w[11]; } s64; struct { unsigned w[12]; } s32;
            size_t pos = 0;
            Lizard_XXH32_reset((void*)&s32, seed); Lizard_XXH64_reset((void*)&s64, seed);
            while (pos < (size_t)n) { size_t take = 1 + rnd() % 97; if (take > (size_t)n - pos) take = (size_t)n - pos;
                { unsigned char* piece = exact(data + pos, take); Lizard_XXH32_update((void*)&s32, piece, take); Lizard_XXH64_update((void*)&s64, piece, take); free(piece); } pos += take; }
            if (Lizard_XXH32_digest((void*)&s32) != Lizard_XXH32(data, (size_t)n, seed) || Lizard_XXH64_digest((void*)&s64) != Lizard_XXH64(data, (size_t)n, seed)) bad++;
        }
        free(data);
    }
    printf("decode_fuzz: seed %u, %d cases, %d misbehaved\n", seed, cases, bad);
    return bad ? 1 : 0;
}
