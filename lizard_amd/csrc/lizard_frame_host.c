/*
 * lizard_frame_host.c — `.liz` frame production on top of the batched GPU block path.
 *
 * A streaming twin of the reference's frame compressor — LizardGPU_compressBegin / _compressUpdate / _flush /
 * _compressEnd follow lib/lizard_frame.c:362-424, :501-599, :610-637, :651-677 call for call — and the one-shot
 * LizardGPU_compressFrame (lizard_frame.c:260-316) built on it exactly like the reference builds its own.  Output is
 * byte for byte what the reference writes for the same preferences and the same sequence of calls in
 * independent-block mode when it is built with -DLIZARD_RESET_MEM (the zero-state oracle of DESIGN.md §2).
 *
 * What is different is WHERE the work happens: the reference compresses one block per Lizard_compress_extState
 * call (:544-556); here every run of blocks an Update call covers — the full blocks it finds in the caller's buffer
 * plus, with autoFlush, the ragged tail — is ONE batch: the blocks are compressed by the block kernels, and the
 * frame's block records (LE32 size word, bit 31 = stored raw, payload; :456-469) are assembled ON THE DEVICE by a
 * prefix sum + compaction (lz_pack.h), so that exactly the bytes of the frame body cross PCIe, once, through pinned
 * staging (lzgpu_frame_records, lizard_gpu.hip).  The XXH32 content checksum is inherently sequential; it runs on a
 * helper thread of the host while the GPU works.
 *
 * The symbols carry the LizardGPU_ prefix on purpose: lib/lizard_frame.c holds frame compression and decompression
 * in one object, so a program keeps linking the reference's LizardF_* (decoder included) and calls these where it
 * wants GPU-rate frames.  Linked-block frames are a serial dependency chain between blocks:
 * they are refused here (blockMode_invalid), never emulated.  XXH32 (lib/xxhash/xxhash.c, public algorithm) is
 * restated below.
 */
#include "../../include/lizard_amd.h"
#include "lizard_gpu_shim.h"

#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LZF_MAGIC            0x184D2206u          /* lizard_frame.c:118 */
#define LZF_MAX_HEADER       15u                  /* maxFHSize, :123 */
#define LZF_ERR(code)        ((size_t)-(long)(LIZARDGPU_FRAME_ERR_##code))

/* ---- XXH32 (xxhash specification), streaming form ---- */
#define XP1 2654435761u
#define XP2 2246822519u
#define XP3 3266489917u
#define XP4 668265263u
#define XP5 374761393u
typedef struct { uint32_t v[4]; uint8_t buf[16]; uint32_t fill; uint64_t total; uint32_t seed; } xxh32_t;
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t rd32le(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t xround(uint32_t acc, uint32_t in) { return rotl32(acc + in * XP2, 13) * XP1; }
static void xxh32_reset(xxh32_t* s, uint32_t seed)
{
    s->v[0] = seed + XP1 + XP2; s->v[1] = seed + XP2; s->v[2] = seed; s->v[3] = seed - XP1;
    s->fill = 0; s->total = 0; s->seed = seed;
}
static void xxh32_update(xxh32_t* s, const void* data, size_t len)
{
    const uint8_t* p = (const uint8_t*)data;
    s->total += len;
    if (s->fill) {
        const size_t take = 16 - s->fill < len ? 16 - s->fill : len;
        memcpy(s->buf + s->fill, p, take);
        s->fill += (uint32_t)take; p += take; len -= take;
        if (s->fill < 16) return;
        s->v[0] = xround(s->v[0], rd32le(s->buf)); s->v[1] = xround(s->v[1], rd32le(s->buf + 4));
        s->v[2] = xround(s->v[2], rd32le(s->buf + 8)); s->v[3] = xround(s->v[3], rd32le(s->buf + 12));
        s->fill = 0;
    }
    {
        uint32_t v1 = s->v[0], v2 = s->v[1], v3 = s->v[2], v4 = s->v[3];
        while (len >= 16) {
            v1 = xround(v1, rd32le(p)); v2 = xround(v2, rd32le(p + 4));
            v3 = xround(v3, rd32le(p + 8)); v4 = xround(v4, rd32le(p + 12));
            p += 16; len -= 16;
        }
        s->v[0] = v1; s->v[1] = v2; s->v[2] = v3; s->v[3] = v4;
    }
    if (len) { memcpy(s->buf, p, len); s->fill = (uint32_t)len; }
}
static uint32_t xxh32_digest(const xxh32_t* s)
{
    const uint8_t* p = s->buf;
    const uint8_t* const end = p + s->fill;
    uint32_t h = s->total >= 16 ? rotl32(s->v[0], 1) + rotl32(s->v[1], 7) + rotl32(s->v[2], 12) + rotl32(s->v[3], 18)
                                : s->seed + XP5;
    h += (uint32_t)s->total;
    while (p + 4 <= end) { h = rotl32(h + rd32le(p) * XP3, 17) * XP4; p += 4; }
    while (p < end) { h = rotl32(h + (uint32_t)*p * XP5, 11) * XP1; p++; }
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
    return h;
}
static uint32_t xxh32(const void* data, size_t len, uint32_t seed)
{
    xxh32_t s;
    xxh32_reset(&s, seed);
    xxh32_update(&s, data, len);
    return xxh32_digest(&s);
}

static void wr32le(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

/* LizardF_getBlockSize, lizard_frame.c:192-201 (0 = default = 128 KiB) */
static size_t block_size_of(unsigned id)
{
    static const size_t sizes[7] = { (size_t)128 << 10, (size_t)256 << 10, (size_t)1 << 20, (size_t)4 << 20,
                                     (size_t)16 << 20, (size_t)64 << 20, (size_t)256 << 20 };
    if (id == 0) id = 1;
    return id - 1 < 7 ? sizes[id - 1] : 0;
}

/* LizardF_optimalBSID, lizard_frame.c:203-217 */
static unsigned optimal_bsid(unsigned requested, size_t srcSize)
{
    unsigned proposed = 1;
    while (requested > proposed) {
        if (srcSize <= block_size_of(proposed)) return proposed;
        proposed++;
    }
    return requested;
}

static int clamp_level(int level)                                /* Lizard_createStream clamps, lizard_compress.c:303-308 */
{
    if (level > LIZARD_MAX_CLEVEL) level = LIZARD_MAX_CLEVEL;
    if (level < LIZARD_MIN_CLEVEL) level = LIZARD_DEFAULT_CLEVEL;
    return level;
}

unsigned LizardGPU_frameIsError(size_t code) { return code > (size_t)-(long)LIZARDGPU_FRAME_ERR_maxCode; }   /* :179-182 */

/* ---- streaming context (twin of LizardF_cctx_t, lizard_frame.c:96-113) ---- */
struct LizardGPU_cctx_s {
    LizardGPU_framePrefs_t prefs;
    unsigned stage;                  /* 0 = expects Begin, 1 = header written */
    size_t   blockSize;
    int      level;
    uint8_t* tmpIn;  size_t tmpInSize, tmpCap;
    unsigned long long totalIn;
    xxh32_t  xxh;
};

int LizardGPU_createCompressionContext(LizardGPU_cctx_t** cctxPtr)
{
    if (!cctxPtr) return -(int)LIZARDGPU_FRAME_ERR_GENERIC;
    *cctxPtr = (LizardGPU_cctx_t*)calloc(1, sizeof(LizardGPU_cctx_t));
    return *cctxPtr ? 0 : -(int)LIZARDGPU_FRAME_ERR_allocation_failed;
}

int LizardGPU_freeCompressionContext(LizardGPU_cctx_t* cctx)
{
    if (cctx) { free(cctx->tmpIn); free(cctx); }
    return 0;
}

/* LizardF_compressBound, lizard_frame.c:432-451 (same value) */
size_t LizardGPU_compressBound(size_t srcSize, const LizardGPU_framePrefs_t* prefsPtr)
{
    LizardGPU_framePrefs_t worst;
    memset(&worst, 0, sizeof worst);
    worst.frameInfo.contentChecksumFlag = 1;
    {
        const LizardGPU_framePrefs_t* p = prefsPtr ? prefsPtr : &worst;
        const size_t blockSize = block_size_of(p->frameInfo.blockSizeID);
        if (!blockSize) return LZF_ERR(maxBlockSize_invalid);
        {
            const size_t nbBlocks = srcSize / blockSize + 1;
            const size_t last = p->autoFlush ? srcSize % blockSize : blockSize;
            return 4 * nbBlocks + blockSize * (nbBlocks - 1) + last + 4 + (size_t)p->frameInfo.contentChecksumFlag * 4;
        }
    }
}

/* LizardF_compressBegin, lizard_frame.c:362-424 */
size_t LizardGPU_compressBegin(LizardGPU_cctx_t* c, void* dstBuffer, size_t dstMaxSize, const LizardGPU_framePrefs_t* prefsPtr)
{
    uint8_t* dst = (uint8_t*)dstBuffer;
    if (!c || !dst) return LZF_ERR(GENERIC);
    if (dstMaxSize < LZF_MAX_HEADER) return LZF_ERR(dstMaxSize_tooSmall);
    if (c->stage != 0) return LZF_ERR(GENERIC);
    if (prefsPtr) c->prefs = *prefsPtr; else memset(&c->prefs, 0, sizeof c->prefs);
    if (c->prefs.frameInfo.blockSizeID == 0) c->prefs.frameInfo.blockSizeID = 1;                      /* :385 */
    c->blockSize = block_size_of(c->prefs.frameInfo.blockSizeID);
    if (!c->blockSize) return LZF_ERR(maxBlockSize_invalid);
    if (c->prefs.frameInfo.frameType != 0) return LZF_ERR(frameType_unknown);
    if (c->prefs.frameInfo.blockMode != 1) return LZF_ERR(blockMode_invalid);                        /* linked: not on the GPU path */
    c->level = clamp_level(c->prefs.compressionLevel);
    if (!LizardGPU_levelSupported(c->level)) return LZF_ERR(compressionLevel_invalid);
    if (c->blockSize > LizardGPU_maxBlockSize(c->level)) return LZF_ERR(maxBlockSize_invalid);       /* per-level limit of the kernels */
    if (!c->prefs.autoFlush && c->tmpCap < c->blockSize) {                                           /* :389-399 */
        free(c->tmpIn);
        c->tmpIn = (uint8_t*)malloc(c->blockSize);
        c->tmpCap = c->tmpIn ? c->blockSize : 0;
        if (!c->tmpIn) return LZF_ERR(allocation_failed);
    }
    c->tmpInSize = 0; c->totalIn = 0;
    xxh32_reset(&c->xxh, 0);
    wr32le(dst, LZF_MAGIC); dst += 4;                                                                /* :403-424 */
    {
        uint8_t* const headerStart = dst;
        *dst++ = (uint8_t)((1u << 6) + ((c->prefs.frameInfo.blockMode & 1u) << 5) + ((c->prefs.frameInfo.contentChecksumFlag & 1u) << 2)
                           + ((c->prefs.frameInfo.contentSize > 0) << 3));
        *dst++ = (uint8_t)((c->prefs.frameInfo.blockSizeID & 7u) << 4);
        if (c->prefs.frameInfo.contentSize) {
            wr32le(dst, (uint32_t)c->prefs.frameInfo.contentSize); wr32le(dst + 4, (uint32_t)(c->prefs.frameInfo.contentSize >> 32));
            dst += 8;
        }
        *dst = (uint8_t)(xxh32(headerStart, (size_t)(dst - headerStart), 0) >> 8);                   /* :219-223 */
        dst++;
    }
    c->stage = 1;
    return (size_t)(dst - (uint8_t*)dstBuffer);
}

struct crc_job { xxh32_t* st; const void* data; size_t len; };
static void* crc_thread(void* arg) { struct crc_job* j = (struct crc_job*)arg; xxh32_update(j->st, j->data, j->len); return NULL; }

/* LizardF_compressUpdate, lizard_frame.c:501-599, independent blocks */
size_t LizardGPU_compressUpdate(LizardGPU_cctx_t* c, void* dstBuffer, size_t dstMaxSize, const void* srcBuffer, size_t srcSize)
{
    const uint8_t* src = (const uint8_t*)srcBuffer;
    const uint8_t* const srcEnd = src + srcSize;
    uint8_t* dst = (uint8_t*)dstBuffer;
    uint8_t* const dstEnd = dst + dstMaxSize;
    pthread_t th;
    struct crc_job job;
    int threaded = 0;
    size_t result = 0;
    if (!c || c->stage != 1) return LZF_ERR(GENERIC);
    if (dstMaxSize < LizardGPU_compressBound(srcSize, &c->prefs)) return LZF_ERR(dstMaxSize_tooSmall);
    if (c->prefs.frameInfo.contentChecksumFlag == 1 && srcSize) {                                    /* :593-594, beside the GPU work */
        job.st = &c->xxh; job.data = srcBuffer; job.len = srcSize;
        if (srcSize >= ((size_t)1 << 20) && pthread_create(&th, NULL, crc_thread, &job) == 0) threaded = 1;
        else xxh32_update(&c->xxh, srcBuffer, srcSize);
    }
    do {
        size_t w = 0;
        if (c->tmpInSize > 0) {                                                                      /* :526-546: complete the buffered block */
            const size_t need = c->blockSize - c->tmpInSize;
            if (need > srcSize) {
                memcpy(c->tmpIn + c->tmpInSize, src, srcSize);
                c->tmpInSize += srcSize; src = srcEnd;
            } else {
                memcpy(c->tmpIn + c->tmpInSize, src, need);
                src += need;
                if (lzgpu_frame_records(c->tmpIn, 1, c->blockSize, c->blockSize, dst, (size_t)(dstEnd - dst), &w, c->level)) { result = LZF_ERR(GENERIC); break; }
                dst += w;
                c->tmpInSize = 0;
            }
        }
        {   /* :548-560: every full block of the caller's buffer and, with autoFlush, the ragged tail — one batch */
            const size_t avail = (size_t)(srcEnd - src);
            const size_t full = avail / c->blockSize, tail = avail % c->blockSize;
            const size_t nb = full + ((c->prefs.autoFlush && tail) ? 1 : 0);
            if (nb) {
                const size_t last = (c->prefs.autoFlush && tail) ? tail : c->blockSize;
                if (lzgpu_frame_records(src, nb, c->blockSize, last, dst, (size_t)(dstEnd - dst), &w, c->level)) { result = LZF_ERR(GENERIC); break; }
                dst += w;
                src += (nb - 1) * c->blockSize + last;
            }
        }
        if (src < srcEnd) {                                                                          /* :585-590: keep the rest (< blockSize) */
            memcpy(c->tmpIn, src, (size_t)(srcEnd - src));
            c->tmpInSize = (size_t)(srcEnd - src);
        }
        c->totalIn += srcSize;
        result = (size_t)(dst - (uint8_t*)dstBuffer);
    } while (0);
    if (threaded) pthread_join(th, NULL);
    return result;
}

/* LizardF_flush, lizard_frame.c:610-637 */
size_t LizardGPU_flush(LizardGPU_cctx_t* c, void* dstBuffer, size_t dstMaxSize)
{
    size_t w = 0;
    if (!c) return LZF_ERR(GENERIC);
    if (c->tmpInSize == 0) return 0;
    if (c->stage != 1) return LZF_ERR(GENERIC);
    if (dstMaxSize < c->tmpInSize + 8) return LZF_ERR(dstMaxSize_tooSmall);
    if (lzgpu_frame_records(c->tmpIn, 1, c->tmpInSize, c->tmpInSize, dstBuffer, dstMaxSize, &w, c->level)) return LZF_ERR(GENERIC);
    c->tmpInSize = 0;
    return w;
}

/* LizardF_compressEnd, lizard_frame.c:651-677 */
size_t LizardGPU_compressEnd(LizardGPU_cctx_t* c, void* dstBuffer, size_t dstMaxSize)
{
    uint8_t* dst = (uint8_t*)dstBuffer;
    const size_t f = LizardGPU_flush(c, dstBuffer, dstMaxSize);
    if (LizardGPU_frameIsError(f)) return f;
    dst += f;
    if (dstMaxSize - f < 4 + (size_t)c->prefs.frameInfo.contentChecksumFlag * 4) return LZF_ERR(dstMaxSize_tooSmall);
    wr32le(dst, 0); dst += 4;
    if (c->prefs.frameInfo.contentChecksumFlag == 1) { wr32le(dst, xxh32_digest(&c->xxh)); dst += 4; }
    c->stage = 0;
    if (c->prefs.frameInfo.contentSize && c->prefs.frameInfo.contentSize != c->totalIn) return LZF_ERR(frameSize_wrong);
    return (size_t)(dst - (uint8_t*)dstBuffer);
}

/* LizardF_compressFrameBound, lizard_frame.c:229-248 */
size_t LizardGPU_compressFrameBound(size_t srcSize, const LizardGPU_framePrefs_t* prefsPtr)
{
    LizardGPU_framePrefs_t prefs;
    if (prefsPtr) prefs = *prefsPtr; else memset(&prefs, 0, sizeof prefs);
    prefs.frameInfo.blockSizeID = optimal_bsid(prefs.frameInfo.blockSizeID, srcSize);
    prefs.autoFlush = 1;
    {
        const size_t b = LizardGPU_compressBound(srcSize, &prefs);
        return LizardGPU_frameIsError(b) ? b : LZF_MAX_HEADER + b;
    }
}

/* LizardF_compressFrame, lizard_frame.c:260-316 */
size_t LizardGPU_compressFrame(void* dstBuffer, size_t dstMaxSize, const void* srcBuffer, size_t srcSize,
                               const LizardGPU_framePrefs_t* prefsPtr)
{
    LizardGPU_cctx_t c;
    LizardGPU_framePrefs_t prefs;
    uint8_t* const dstStart = (uint8_t*)dstBuffer;
    uint8_t* dst = dstStart;
    size_t r;
    memset(&c, 0, sizeof c);
    if (prefsPtr) prefs = *prefsPtr; else memset(&prefs, 0, sizeof prefs);
    if (prefs.frameInfo.contentSize != 0) prefs.frameInfo.contentSize = (unsigned long long)srcSize;      /* :279-280 */
    prefs.frameInfo.blockSizeID = optimal_bsid(prefs.frameInfo.blockSizeID, srcSize);                       /* :282 */
    prefs.autoFlush = 1;
    if (!block_size_of(prefs.frameInfo.blockSizeID)) return LZF_ERR(maxBlockSize_invalid);
    if (srcSize <= block_size_of(prefs.frameInfo.blockSizeID)) prefs.frameInfo.blockMode = 1;               /* :284-285 */
    if (dstMaxSize < LizardGPU_compressFrameBound(srcSize, &prefs)) return LZF_ERR(dstMaxSize_tooSmall);   /* :289 */
    r = LizardGPU_compressBegin(&c, dst, dstMaxSize, &prefs);
    if (LizardGPU_frameIsError(r)) return r;
    dst += r;
    r = LizardGPU_compressUpdate(&c, dst, dstMaxSize - (size_t)(dst - dstStart), srcBuffer, srcSize);
    if (LizardGPU_frameIsError(r)) return r;
    dst += r;
    r = LizardGPU_compressEnd(&c, dst, dstMaxSize - (size_t)(dst - dstStart));
    if (LizardGPU_frameIsError(r)) return r;
    dst += r;
    return (size_t)(dst - dstStart);
}
