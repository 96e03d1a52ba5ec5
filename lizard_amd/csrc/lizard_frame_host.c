/*
 * lizard_frame_host.c — the reference's FRAME layer (lib/lizard_frame.h: LizardF_*) on top of the batched GPU block path
 * (SURVEY.md section 8b "symbols a replacement must export", section 8f rank 2).
 *
 * Compression.  LizardF_compressBegin / _compressUpdate / _flush / _compressEnd follow lib/lizard_frame.c:362-424, :501-599,
 * :610-637, :651-677 call for call and LizardF_compressFrame (:260-316) is built on them like the reference builds its own:
 * same header, same LE32 block records (bit 31 = stored raw), same end mark and XXH32 content checksum, for the same
 * preferences and the same sequence of calls.  What is different is WHERE the work happens.  The reference compresses one block
 * per Lizard_compress_extState call (:544-556); here every run of blocks an Update call covers — the full blocks it finds in the
 * caller's buffer plus, with autoFlush, the ragged tail — is ONE batch: the blocks are compressed by the block kernels and the
 * frame's block records are assembled ON THE DEVICE by a prefix sum + compaction (lz_pack.h), so that exactly the bytes of the
 * frame body cross PCIe, once, through pinned staging (lzgpu_frame_records, lizard_pipeline_host.c).  The XXH32 content checksum
 * is inherently sequential; it runs on a helper thread of the host while the GPU works.  A program written against
 * lizard_frame.h — the reference's CLI (programs/lizardio.c), tests/frametest.c, tests/fullbench.c — reaches the batch path
 * without a source change.
 *   independent blocks : byte for byte what the reference writes when it is built with -DLIZARD_RESET_MEM.
 *   linked blocks      : the reference's Lizard_compress_continue chain is serial by definition; this library's
 *                        Lizard_compress_continue is history-free (include/lizard_amd.h), so a linked frame carries the linked
 *                        flag and independently compressed blocks — valid, decodable by any Lizard frame decoder, same batch
 *                        path; not the reference's linked-mode bytes.
 *   a level without a GPU kernel, or no usable GPU: Lizard_compress_extState returns 0 there, and the reference's frame layer
 *                        stores such blocks raw (:456-469).  Same here, after one loud line on stderr.
 * The LizardGPU_* twins of round 2/3 remain as the STRICT form: they promise byte identity with the reference and therefore
 * refuse linked frames above one block, levels without a kernel and GPU failures instead of storing raw.
 *
 * Decompression.  LizardF_decompress / _getFrameInfo (lib/lizard_frame.c:743-1362) as a state machine of its own: header,
 * block header, raw block, compressed block, flush, suffix, skippable frames; any segmentation of input and output.  Blocks
 * are decoded by this library's host decoder (lizard_decode_host.c) — the decoder is not the path this library accelerates —
 * straight into the caller's buffer when a whole block fits there, otherwise through an internal buffer that, for linked
 * frames, also holds the 16 MiB of history the next block may refer to (the caller's buffer is never relied on as history).
 */
#include "../../include/lizard_amd.h"
#include "lizard_gpu_shim.h"
#include "lizard_xxhash.h"

#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LZF_MAGIC            0x184D2206u          /* lizard_frame.c:118 */
#define LZF_MAGIC_SKIPPABLE  0x184D2A50u          /* :117 */
#define LZF_MIN_HEADER       7u                   /* minFHSize, :122 */
#define LZF_MAX_HEADER       15u                  /* maxFHSize, :123 */
#define LZF_DICT             ((size_t)1 << 24)    /* LIZARD_DICT_SIZE */
#define LZF_ERR(code)        ((size_t)-(long)(LizardF_ERROR_##code))

/* LizardF_errorCodes, lib/lizard_frame_static.h:57-67 */
enum {
    LizardF_OK_NoError = 0, LizardF_ERROR_GENERIC, LizardF_ERROR_maxBlockSize_invalid, LizardF_ERROR_blockMode_invalid,
    LizardF_ERROR_contentChecksumFlag_invalid, LizardF_ERROR_compressionLevel_invalid, LizardF_ERROR_headerVersion_wrong,
    LizardF_ERROR_blockChecksum_unsupported, LizardF_ERROR_reservedFlag_set, LizardF_ERROR_allocation_failed,
    LizardF_ERROR_srcSize_tooLarge, LizardF_ERROR_dstMaxSize_tooSmall, LizardF_ERROR_frameHeader_incomplete,
    LizardF_ERROR_frameType_unknown, LizardF_ERROR_frameSize_wrong, LizardF_ERROR_srcPtr_wrong, LizardF_ERROR_decompressionFailed,
    LizardF_ERROR_headerChecksum_invalid, LizardF_ERROR_contentChecksum_invalid, LizardF_ERROR_maxCode
};
static const char* const kErrorNames[] = {
    "OK_NoError", "ERROR_GENERIC", "ERROR_maxBlockSize_invalid", "ERROR_blockMode_invalid", "ERROR_contentChecksumFlag_invalid",
    "ERROR_compressionLevel_invalid", "ERROR_headerVersion_wrong", "ERROR_blockChecksum_unsupported", "ERROR_reservedFlag_set",
    "ERROR_allocation_failed", "ERROR_srcSize_tooLarge", "ERROR_dstMaxSize_tooSmall", "ERROR_frameHeader_incomplete",
    "ERROR_frameType_unknown", "ERROR_frameSize_wrong", "ERROR_srcPtr_wrong", "ERROR_decompressionFailed",
    "ERROR_headerChecksum_invalid", "ERROR_contentChecksum_invalid", "ERROR_maxCode"
};

unsigned LizardF_isError(size_t code) { return code > (size_t)-(long)LizardF_ERROR_maxCode; }          /* lizard_frame.c:179-182 */
const char* LizardF_getErrorName(size_t code)                                                          /* :184-189 */
{
    return LizardF_isError(code) ? kErrorNames[(size_t)0 - code] : "Unspecified error code";
}

static uint32_t rd32le(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static void wr32le(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

/* LizardF_getBlockSize, lizard_frame.c:192-201 (0 = default = 128 KiB); 0 = invalid id */
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

/* ================================================= compression ================================================= */

/* twin of LizardF_cctx_t, lizard_frame.c:96-113 (the block compressor's state lives on the device) */
struct LizardF_cctx_s {
    LizardF_preferences_t prefs;
    unsigned version;
    unsigned stage;                  /* 0 = expects Begin, 1 = header written */
    unsigned strict;                 /* the LizardGPU_* twins: refuse what cannot be byte-identical instead of degrading */
    size_t   blockSize;
    int      level;
    uint8_t* tmpIn;  size_t tmpInSize, tmpCap;
    unsigned long long totalIn;
    Lizard_XXH32_state_t xxh;
};

size_t LizardF_createCompressionContext(LizardF_compressionContext_t* cctxPtr, unsigned version)      /* :329-343 */
{
    LizardF_compressionContext_t c;
    if (!cctxPtr) return LZF_ERR(GENERIC);
    c = (LizardF_compressionContext_t)calloc(1, sizeof(struct LizardF_cctx_s));
    if (!c) return LZF_ERR(allocation_failed);
    c->version = version;
    *cctxPtr = c;
    return LizardF_OK_NoError;
}

size_t LizardF_freeCompressionContext(LizardF_compressionContext_t c)                                   /* :346-357 */
{
    if (c) { free(c->tmpIn); free(c); }
    return LizardF_OK_NoError;
}

/* LizardF_compressBound, lizard_frame.c:432-451 (same value) */
size_t LizardF_compressBound(size_t srcSize, const LizardF_preferences_t* prefsPtr)
{
    LizardF_preferences_t worst;
    memset(&worst, 0, sizeof worst);
    worst.frameInfo.contentChecksumFlag = 1;
    {
        const LizardF_preferences_t* p = prefsPtr ? prefsPtr : &worst;
        const size_t blockSize = block_size_of(p->frameInfo.blockSizeID);
        if (!blockSize) return LZF_ERR(maxBlockSize_invalid);
        {
            const size_t nbBlocks = (unsigned)(srcSize / blockSize) + 1;
            const size_t last = p->autoFlush ? srcSize % blockSize : blockSize;
            return 4 * nbBlocks + blockSize * (nbBlocks - 1) + last + 4 + (size_t)p->frameInfo.contentChecksumFlag * 4;
        }
    }
}

/* LizardF_compressBegin, lizard_frame.c:362-424 */
size_t LizardF_compressBegin(LizardF_compressionContext_t c, void* dstBuffer, size_t dstMaxSize, const LizardF_preferences_t* prefsPtr)
{
    uint8_t* dst = (uint8_t*)dstBuffer;
    if (!c || !dst) return LZF_ERR(GENERIC);
    if (dstMaxSize < LZF_MAX_HEADER) return LZF_ERR(dstMaxSize_tooSmall);
    if (c->stage != 0) return LZF_ERR(GENERIC);
    if (prefsPtr) c->prefs = *prefsPtr; else memset(&c->prefs, 0, sizeof c->prefs);
    if (c->prefs.frameInfo.blockSizeID == 0) c->prefs.frameInfo.blockSizeID = 1;                      /* :385 */
    c->blockSize = block_size_of(c->prefs.frameInfo.blockSizeID);
    if (!c->blockSize) return LZF_ERR(maxBlockSize_invalid);
    c->level = clamp_level(c->prefs.compressionLevel);
    if (c->strict) {
        if (c->prefs.frameInfo.frameType != 0) return LZF_ERR(frameType_unknown);
        if (c->prefs.frameInfo.blockMode != 1) return LZF_ERR(blockMode_invalid);                    /* linked: not byte-identical */
        if (!LizardGPU_levelSupported(c->level)) return LZF_ERR(compressionLevel_invalid);
    }
    if (!c->prefs.autoFlush && c->tmpCap < c->blockSize) {                                           /* :389-399 */
        free(c->tmpIn);
        c->tmpIn = (uint8_t*)malloc(c->blockSize);
        c->tmpCap = c->tmpIn ? c->blockSize : 0;
        if (!c->tmpIn) return LZF_ERR(allocation_failed);
    }
    c->tmpInSize = 0; c->totalIn = 0;
    Lizard_XXH32_reset(&c->xxh, 0);
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
        *dst = (uint8_t)(Lizard_XXH32(headerStart, (size_t)(dst - headerStart), 0) >> 8);            /* :219-223 */
        dst++;
    }
    c->stage = 1;
    return (size_t)(dst - (uint8_t*)dstBuffer);
}

/* The block records of nb blocks (blockSize each, the last one `last` bytes) at src, packed into dst: one batch on the GPU.
 * Non-strict contexts degrade like the reference's frame layer does when Lizard_compress_extState returns 0 (:456-469): every
 * block of the batch is stored raw, after one line on stderr.
 *
 * `reserve` = bytes of dst the caller still needs after the records (the end mark and content checksum that
 * LizardF_compressBound counts, :443).  It matters for ONE input: a 1-byte block.  The reference compresses it into a 6-byte
 * block (its room test wraps at maxDstSize 0, lizard_compress.c:238) and so writes a 10-byte record where LizardF_compressBound
 * counted 5 — with a buffer of exactly the bound and nothing saved elsewhere it runs up to 5 bytes past dstMaxSize
 * (LizardF_compressEnd has no room test, :651-677).  This library never writes past dstMaxSize: when the faithful record leaves
 * less than `reserve`, a non-strict context stores that one byte raw (a 5-byte record, what the bound counted; same content
 * for every decoder), a strict one reports dstMaxSize_tooSmall. */
static void raw_record(uint8_t* dst, const uint8_t* src, size_t n)
{
    wr32le(dst, (uint32_t)n | 0x80000000u);
    memcpy(dst + 4, src, n);
}

static int frame_records(LizardF_compressionContext_t c, const uint8_t* src, size_t nb, size_t blockSize, size_t last,
                         uint8_t* dst, size_t cap, size_t reserve, size_t* written)
{
    const int onGpu = LizardGPU_levelSupported(c->level);
    size_t i, need;
    if (onGpu && lzgpu_frame_records(src, nb, blockSize, last, dst, cap, written, c->level) == 0) {
        const size_t w = *written;
        if (last == 1 && !c->strict && cap - w < reserve && w >= 10 && rd32le(dst + w - 10) == 6u) {      /* see above */
            raw_record(dst + w - 10, src + (nb - 1) * blockSize, 1);
            *written = w - 5;
        }
        return 0;
    }
    if (c->strict) return (onGpu && last == 1 && cap < (nb - 1) * (blockSize + 4) + 10) ? -2 : -1;     /* -2: that record cannot fit */
    if (onGpu && last == 1 && cap < (nb - 1) * (blockSize + 4) + 10 + reserve) {
        /* the batch may have failed only for the 1-byte block's 10-byte record: the others on the GPU, that byte raw */
        size_t w = 0;
        if (nb == 1 || lzgpu_frame_records(src, nb - 1, blockSize, blockSize, dst, cap, &w, c->level) == 0) {
            if (cap - w < 5) return -1;
            raw_record(dst + w, src + (nb - 1) * blockSize, 1);
            *written = w + 5;
            return 0;
        }
    }
    lzgpu_note_degraded(onGpu ? "frame blocks are stored uncompressed: the GPU batch failed" : "frame blocks are stored uncompressed: level not implemented on the GPU path", c->level);
    need = (nb - 1) * (blockSize + 4) + last + 4;
    if (need > cap) return -1;
    for (i = 0; i < nb; i++) {
        const size_t n = i + 1 == nb ? last : blockSize;
        raw_record(dst, src + i * blockSize, n);
        dst += 4 + n;
    }
    *written = need;
    return 0;
}

struct crc_job { Lizard_XXH32_state_t* st; const void* data; size_t len; };
static void* crc_thread(void* arg) { struct crc_job* j = (struct crc_job*)arg; Lizard_XXH32_update(j->st, j->data, j->len); return NULL; }

/* LizardF_compressUpdate, lizard_frame.c:501-599.  compressOptionsPtr only carries stableSrc, which tells the reference whether
 * it must save the dictionary of a linked stream (:563-571): there is no dictionary here. */
size_t LizardF_compressUpdate(LizardF_compressionContext_t c, void* dstBuffer, size_t dstMaxSize, const void* srcBuffer, size_t srcSize,
                              const LizardF_compressOptions_t* compressOptionsPtr)
{
    const uint8_t* src = (const uint8_t*)srcBuffer;
    const uint8_t* const srcEnd = src + srcSize;
    uint8_t* dst = (uint8_t*)dstBuffer;
    uint8_t* const dstEnd = dst + dstMaxSize;
    pthread_t th;
    struct crc_job job;
    int threaded = 0;
    size_t result = 0, frameEnd;
    (void)compressOptionsPtr;
    if (!c || c->stage != 1) return LZF_ERR(GENERIC);
    frameEnd = 4 + (size_t)c->prefs.frameInfo.contentChecksumFlag * 4;                                /* part of LizardF_compressBound, :443 */
    if (dstMaxSize < LizardF_compressBound(srcSize, &c->prefs)) return LZF_ERR(dstMaxSize_tooSmall);
    if (c->prefs.frameInfo.contentChecksumFlag == 1 && srcSize) {                                    /* :593-594, beside the GPU work */
        job.st = &c->xxh; job.data = srcBuffer; job.len = srcSize;
        if (srcSize >= ((size_t)1 << 20) && pthread_create(&th, NULL, crc_thread, &job) == 0) threaded = 1;
        else Lizard_XXH32_update(&c->xxh, srcBuffer, srcSize);
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
                if (frame_records(c, c->tmpIn, 1, c->blockSize, c->blockSize, dst, (size_t)(dstEnd - dst), 0, &w)) { result = LZF_ERR(GENERIC); break; }
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
                const int e = frame_records(c, src, nb, c->blockSize, last, dst, (size_t)(dstEnd - dst), frameEnd, &w);
                if (e) { result = e == -2 ? LZF_ERR(dstMaxSize_tooSmall) : LZF_ERR(GENERIC); break; }
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
static size_t flush_buffered(LizardF_compressionContext_t c, void* dstBuffer, size_t dstMaxSize, size_t reserve)
{
    size_t w = 0;
    if (!c) return LZF_ERR(GENERIC);
    if (c->tmpInSize == 0) return 0;
    if (c->stage != 1) return LZF_ERR(GENERIC);
    if (dstMaxSize < c->tmpInSize + 8) return LZF_ERR(dstMaxSize_tooSmall);
    {
        const int e = frame_records(c, c->tmpIn, 1, c->tmpInSize, c->tmpInSize, (uint8_t*)dstBuffer, dstMaxSize, reserve, &w);
        if (e) return e == -2 ? LZF_ERR(dstMaxSize_tooSmall) : LZF_ERR(GENERIC);
    }
    c->tmpInSize = 0;
    return w;
}

size_t LizardF_flush(LizardF_compressionContext_t c, void* dstBuffer, size_t dstMaxSize, const LizardF_compressOptions_t* compressOptionsPtr)
{
    (void)compressOptionsPtr;
    return flush_buffered(c, dstBuffer, dstMaxSize, 0);
}

/* LizardF_compressEnd, lizard_frame.c:651-677 */
size_t LizardF_compressEnd(LizardF_compressionContext_t c, void* dstBuffer, size_t dstMaxSize, const LizardF_compressOptions_t* compressOptionsPtr)
{
    uint8_t* dst = (uint8_t*)dstBuffer;
    size_t f, frameEnd;
    (void)compressOptionsPtr;
    if (!c) return LZF_ERR(GENERIC);
    frameEnd = 4 + (size_t)c->prefs.frameInfo.contentChecksumFlag * 4;
    f = flush_buffered(c, dstBuffer, dstMaxSize, frameEnd);
    if (LizardF_isError(f)) return f;
    dst += f;
    if (dstMaxSize - f < frameEnd) return LZF_ERR(dstMaxSize_tooSmall);                                /* the reference writes regardless */
    wr32le(dst, 0); dst += 4;
    if (c->prefs.frameInfo.contentChecksumFlag == 1) { wr32le(dst, Lizard_XXH32_digest(&c->xxh)); dst += 4; }
    c->stage = 0;
    if (c->prefs.frameInfo.contentSize && c->prefs.frameInfo.contentSize != c->totalIn) return LZF_ERR(frameSize_wrong);
    return (size_t)(dst - (uint8_t*)dstBuffer);
}

/* LizardF_compressFrameBound, lizard_frame.c:229-248 */
size_t LizardF_compressFrameBound(size_t srcSize, const LizardF_preferences_t* prefsPtr)
{
    LizardF_preferences_t prefs;
    if (prefsPtr) prefs = *prefsPtr; else memset(&prefs, 0, sizeof prefs);
    prefs.frameInfo.blockSizeID = optimal_bsid(prefs.frameInfo.blockSizeID, srcSize);
    prefs.autoFlush = 1;
    {
        const size_t b = LizardF_compressBound(srcSize, &prefs);
        return LizardF_isError(b) ? b : LZF_MAX_HEADER + b;
    }
}

static size_t compress_frame(void* dstBuffer, size_t dstMaxSize, const void* srcBuffer, size_t srcSize,
                             const LizardF_preferences_t* prefsPtr, unsigned strict)
{
    struct LizardF_cctx_s c;
    LizardF_preferences_t prefs;
    uint8_t* const dstStart = (uint8_t*)dstBuffer;
    uint8_t* dst = dstStart;
    size_t r;
    memset(&c, 0, sizeof c);
    c.strict = strict;
    if (prefsPtr) prefs = *prefsPtr; else memset(&prefs, 0, sizeof prefs);
    if (prefs.frameInfo.contentSize != 0) prefs.frameInfo.contentSize = (unsigned long long)srcSize;      /* :279-280 */
    prefs.frameInfo.blockSizeID = optimal_bsid(prefs.frameInfo.blockSizeID, srcSize);                       /* :282 */
    prefs.autoFlush = 1;
    if (!block_size_of(prefs.frameInfo.blockSizeID)) return LZF_ERR(maxBlockSize_invalid);
    if (srcSize <= block_size_of(prefs.frameInfo.blockSizeID)) prefs.frameInfo.blockMode = 1;               /* :284-285 */
    if (dstMaxSize < LizardF_compressFrameBound(srcSize, &prefs)) return LZF_ERR(dstMaxSize_tooSmall);     /* :289 */
    r = LizardF_compressBegin(&c, dst, dstMaxSize, &prefs);
    if (LizardF_isError(r)) return r;
    dst += r;
    r = LizardF_compressUpdate(&c, dst, dstMaxSize - (size_t)(dst - dstStart), srcBuffer, srcSize, NULL);
    if (LizardF_isError(r)) return r;
    dst += r;
    r = LizardF_compressEnd(&c, dst, dstMaxSize - (size_t)(dst - dstStart), NULL);
    if (LizardF_isError(r)) return r;
    dst += r;
    return (size_t)(dst - dstStart);
}

/* LizardF_compressFrame, lizard_frame.c:260-316 */
size_t LizardF_compressFrame(void* dstBuffer, size_t dstMaxSize, const void* srcBuffer, size_t srcSize, const LizardF_preferences_t* prefsPtr)
{
    return compress_frame(dstBuffer, dstMaxSize, srcBuffer, srcSize, prefsPtr, 0);
}

/* ---- the strict twins (include/lizard_amd.h part 3): LizardGPU_framePrefs_t == LizardF_preferences_t ---- */
unsigned LizardGPU_frameIsError(size_t code) { return LizardF_isError(code); }
size_t LizardGPU_compressFrameBound(size_t srcSize, const LizardGPU_framePrefs_t* p) { return LizardF_compressFrameBound(srcSize, p); }
size_t LizardGPU_compressBound(size_t srcSize, const LizardGPU_framePrefs_t* p) { return LizardF_compressBound(srcSize, p); }
size_t LizardGPU_compressFrame(void* dst, size_t cap, const void* src, size_t n, const LizardGPU_framePrefs_t* p) { return compress_frame(dst, cap, src, n, p, 1); }
int LizardGPU_createCompressionContext(LizardGPU_cctx_t** cctxPtr)
{
    const size_t r = LizardF_createCompressionContext(cctxPtr, LIZARDF_VERSION);
    if (LizardF_isError(r)) return -(int)((size_t)0 - r);
    (*cctxPtr)->strict = 1;
    return 0;
}
int    LizardGPU_freeCompressionContext(LizardGPU_cctx_t* c) { LizardF_freeCompressionContext(c); return 0; }
size_t LizardGPU_compressBegin(LizardGPU_cctx_t* c, void* dst, size_t cap, const LizardGPU_framePrefs_t* p) { return LizardF_compressBegin(c, dst, cap, p); }
size_t LizardGPU_compressUpdate(LizardGPU_cctx_t* c, void* dst, size_t cap, const void* src, size_t n) { return LizardF_compressUpdate(c, dst, cap, src, n, NULL); }
size_t LizardGPU_flush(LizardGPU_cctx_t* c, void* dst, size_t cap) { return LizardF_flush(c, dst, cap, NULL); }
size_t LizardGPU_compressEnd(LizardGPU_cctx_t* c, void* dst, size_t cap) { return LizardF_compressEnd(c, dst, cap, NULL); }

/* ================================================= decompression ================================================= */

enum { DS_HEADER = 0, DS_BLOCK_HEADER, DS_RAW, DS_BLOCK, DS_FLUSH, DS_SUFFIX, DS_SKIP };

/* twin of LizardF_dctx_t, lizard_frame.c:115-135 */
struct LizardF_dctx_s {
    LizardF_frameInfo_t info;
    unsigned version;
    unsigned stage;                              /* DS_*; 0 = between frames (the value LizardF_freeDecompressionContext reports) */
    int      infoValid;                          /* a frame header has been decoded since the last frame ended */
    unsigned long long remaining;                /* content size still expected (headers that carry one) */
    size_t   maxBlockSize;
    const uint8_t* srcExpect;
    uint8_t  small[16];  size_t smallFill, smallTarget;      /* frame header / block header / suffix being collected */
    uint8_t* in;   size_t inCap, inFill, inTarget;           /* a compressed block that arrives in pieces */
    uint8_t* hist; size_t histCap, histFill;                 /* decoded bytes: history of a linked frame, or one block waiting for room in dst */
    size_t   outStart, outEnd;                               /* hist[outStart..outEnd) still has to reach the caller */
    size_t   left;                                           /* bytes left of a raw block / a skippable frame */
    Lizard_XXH32_state_t xxh;
};

size_t LizardF_createDecompressionContext(LizardF_decompressionContext_t* dctxPtr, unsigned version)  /* :689-697 */
{
    LizardF_decompressionContext_t d;
    if (!dctxPtr) return LZF_ERR(GENERIC);
    d = (LizardF_decompressionContext_t)calloc(1, sizeof(struct LizardF_dctx_s));
    if (!d) return LZF_ERR(GENERIC);
    d->version = version;
    *dctxPtr = d;
    return LizardF_OK_NoError;
}

size_t LizardF_freeDecompressionContext(LizardF_decompressionContext_t d)                              /* :699-710: 0 iff no frame is under way */
{
    size_t result = LizardF_OK_NoError;
    if (d) { result = d->stage; free(d->in); free(d->hist); free(d); }
    return result;
}

/* Header of a frame at p[0..n), n = 7 or 15 as its FLG byte says (LizardF_decodeHeader, :756-857).  0 or an error code. */
static size_t decode_header(LizardF_decompressionContext_t d, const uint8_t* p, size_t n)
{
    const unsigned flg = p[4], bd = p[5];
    const unsigned version = (flg >> 6) & 3u, blockMode = (flg >> 5) & 1u, blockChecksum = (flg >> 4) & 1u, contentSizeFlag = (flg >> 3) & 1u,
                   contentChecksum = (flg >> 2) & 1u, bsid = (bd >> 4) & 7u;
    size_t need;
    if (version != 1) return LZF_ERR(headerVersion_wrong);                                           /* :811-816 */
    if (blockChecksum) return LZF_ERR(blockChecksum_unsupported);
    if (flg & 3u) return LZF_ERR(reservedFlag_set);
    if (bd & 0x80u) return LZF_ERR(reservedFlag_set);
    if (bsid < 1) return LZF_ERR(maxBlockSize_invalid);
    if (bd & 0x0Fu) return LZF_ERR(reservedFlag_set);
    if ((uint8_t)(Lizard_XXH32(p + 4, n - 5, 0) >> 8) != p[n - 1]) return LZF_ERR(headerChecksum_invalid);   /* :819-820 */
    memset(&d->info, 0, sizeof d->info);
    d->info.blockMode = (LizardF_blockMode_t)blockMode;
    d->info.contentChecksumFlag = (LizardF_contentChecksum_t)contentChecksum;
    d->info.blockSizeID = (LizardF_blockSizeID_t)bsid;
    d->maxBlockSize = block_size_of(bsid);
    d->remaining = 0;
    if (contentSizeFlag) {
        d->info.contentSize = (unsigned long long)rd32le(p + 6) | ((unsigned long long)rd32le(p + 10) << 32);
        d->remaining = d->info.contentSize;
    }
    if (contentChecksum) Lizard_XXH32_reset(&d->xxh, 0);
    /* decoded bytes: one block, plus — linked — the history behind it with room to slide only every 16 MiB */
    need = d->maxBlockSize + (blockMode == 0 ? 2 * LZF_DICT : 0);
    if (d->histCap < need) {
        free(d->hist);
        d->hist = (uint8_t*)malloc(need);
        d->histCap = d->hist ? need : 0;
        if (!d->hist) return LZF_ERR(GENERIC);
    }
    d->histFill = 0; d->outStart = d->outEnd = 0; d->inFill = 0; d->inTarget = 0;
    d->infoValid = 1;
    return 0;
}

/* room for `n` more decoded bytes behind the history of a linked frame: when the buffer is full the last 16 MiB slide to its
 * front (nothing is waiting to be flushed when this is called) */
static void hist_make_room(LizardF_decompressionContext_t d, size_t n)
{
    if (d->histFill + n > d->histCap) {
        const size_t keep = d->histFill < LZF_DICT ? d->histFill : LZF_DICT;
        memmove(d->hist, d->hist + d->histFill - keep, keep);
        d->histFill = keep;
    }
}

/* LizardF_decompress, lizard_frame.c:980-1362 */
size_t LizardF_decompress(LizardF_decompressionContext_t d, void* dstBuffer, size_t* dstSizePtr, const void* srcBuffer, size_t* srcSizePtr,
                          const LizardF_decompressOptions_t* decompressOptionsPtr)
{
    const uint8_t* const srcStart = (const uint8_t*)srcBuffer;
    const uint8_t* src = srcStart;
    const uint8_t* const srcEnd = srcStart + *srcSizePtr;
    uint8_t* const dstStart = (uint8_t*)dstBuffer;
    uint8_t* dst = dstStart;
    uint8_t* const dstEnd = dstStart + *dstSizePtr;
    size_t hint = 1;
    int more = 1;
    (void)decompressOptionsPtr;                              /* stableDst: the caller's buffer is never used as history here */
    *srcSizePtr = 0; *dstSizePtr = 0;
    if (d->srcExpect && srcStart != d->srcExpect) return LZF_ERR(srcPtr_wrong);                      /* :1004-1006 */
    while (more) {
        switch (d->stage) {
        case DS_HEADER: {
            /* 5 bytes name the kind of frame and the size of its header: 8 (skippable), 7 or 15 */
            if (d->smallTarget == 0) { d->smallFill = 0; d->smallTarget = 5; d->infoValid = 0; }
            {
                size_t n = d->smallTarget - d->smallFill;
                if (n > (size_t)(srcEnd - src)) n = (size_t)(srcEnd - src);
                memcpy(d->small + d->smallFill, src, n);
                d->smallFill += n; src += n;
            }
            if (d->smallFill < d->smallTarget) {
                hint = ((d->smallTarget < LZF_MIN_HEADER ? LZF_MIN_HEADER : d->smallTarget) - d->smallFill) + 4;   /* rest of the header + a block header */
                more = 0; break;
            }
            if (d->smallTarget == 5) {
                const uint32_t magic = rd32le(d->small);
                if ((magic & 0xFFFFFFF0u) == LZF_MAGIC_SKIPPABLE) { d->smallTarget = 8; break; }
                if (magic != LZF_MAGIC) { d->smallTarget = 0; return LZF_ERR(frameType_unknown); }
                d->smallTarget = ((d->small[4] >> 3) & 1u) ? LZF_MAX_HEADER : LZF_MIN_HEADER;
                break;
            }
            if (d->smallTarget == 8) {                                                                /* skippable frame: LE32 size, then that many bytes */
                memset(&d->info, 0, sizeof d->info);
                d->info.frameType = LizardF_skippableFrame;
                d->info.contentSize = rd32le(d->small + 4);                                          /* :1303-1304 */
                d->left = (size_t)d->info.contentSize;
                d->infoValid = 1;
                d->smallTarget = 0;
                d->stage = DS_SKIP;
                break;
            }
            {
                const size_t e = decode_header(d, d->small, d->smallTarget);
                d->smallTarget = 0;
                if (LizardF_isError(e)) return e;
            }
            d->stage = DS_BLOCK_HEADER;
            break;
        }
        case DS_BLOCK_HEADER: {
            const uint8_t* h;
            uint32_t word;
            size_t size;
            if (d->smallTarget == 0 && (size_t)(srcEnd - src) >= 4) { h = src; src += 4; }
            else {
                size_t n;
                if (d->smallTarget == 0) { d->smallFill = 0; d->smallTarget = 4; }
                n = 4 - d->smallFill;
                if (n > (size_t)(srcEnd - src)) n = (size_t)(srcEnd - src);
                memcpy(d->small + d->smallFill, src, n);
                d->smallFill += n; src += n;
                if (d->smallFill < 4) { hint = 4 - d->smallFill; more = 0; break; }                   /* :1062-1066 */
                h = d->small;
                d->smallTarget = 0;
            }
            word = rd32le(h);
            size = word & 0x7FFFFFFFu;
            if (size == 0) { d->stage = DS_SUFFIX; break; }                                           /* end mark */
            if (size > d->maxBlockSize) return LZF_ERR(GENERIC);                                      /* :1076 */
            if (word & 0x80000000u) { d->left = size; d->stage = DS_RAW; break; }
            d->inTarget = size; d->inFill = 0;
            d->stage = DS_BLOCK;
            if (dst == dstEnd) { hint = size + 4; more = 0; }                                         /* :1083-1086 */
            break;
        }
        case DS_RAW: {                                                                                /* :1090-1112: streams through, piece by piece */
            size_t n = d->left;
            if (n > (size_t)(srcEnd - src)) n = (size_t)(srcEnd - src);
            if (n > (size_t)(dstEnd - dst)) n = (size_t)(dstEnd - dst);
            if (n) {
                memcpy(dst, src, n);
                if (d->info.contentChecksumFlag) Lizard_XXH32_update(&d->xxh, src, n);
                if (d->info.contentSize) d->remaining -= n;
                if (d->info.blockMode == LizardF_blockLinked) {                                      /* the bytes are history for later blocks */
                    hist_make_room(d, n);
                    memcpy(d->hist + d->histFill, src, n);
                    d->histFill += n;
                }
                src += n; dst += n; d->left -= n;
            }
            if (d->left == 0) { d->stage = DS_BLOCK_HEADER; break; }
            hint = d->left + 4;
            more = 0;
            break;
        }
        case DS_BLOCK: {
            const uint8_t* from;
            int n;
            if (d->inFill == 0 && (size_t)(srcEnd - src) >= d->inTarget) { from = src; src += d->inTarget; }   /* whole block in the caller's buffer */
            else {
                size_t k;
                if (d->inCap < d->maxBlockSize) {
                    free(d->in);
                    d->in = (uint8_t*)malloc(d->maxBlockSize);
                    d->inCap = d->in ? d->maxBlockSize : 0;
                    if (!d->in) return LZF_ERR(GENERIC);
                }
                k = d->inTarget - d->inFill;
                if (k > (size_t)(srcEnd - src)) k = (size_t)(srcEnd - src);
                memcpy(d->in + d->inFill, src, k);
                d->inFill += k; src += k;
                if (d->inFill < d->inTarget) { hint = (d->inTarget - d->inFill) + 4; more = 0; break; }   /* :1132-1136 */
                from = d->in;
            }
            if (d->info.blockMode == LizardF_blockLinked) {
                /* into the internal buffer, right behind the history (a prefix: no offset reaches further than LIZARD_DICT_SIZE back) */
                size_t dict;
                hist_make_room(d, d->maxBlockSize);
                dict = d->histFill < LZF_DICT ? d->histFill : LZF_DICT;
                n = Lizard_decompress_safe_usingDict((const char*)from, (char*)d->hist + d->histFill, (int)d->inTarget, (int)d->maxBlockSize,
                                                     (const char*)d->hist + d->histFill - dict, (int)dict);
                if (n < 0) return LZF_ERR(decompressionFailed);
                d->outStart = d->histFill; d->outEnd = d->histFill + (size_t)n; d->histFill += (size_t)n;
                if (d->info.contentChecksumFlag) Lizard_XXH32_update(&d->xxh, d->hist + d->outStart, (size_t)n);
                if (d->info.contentSize) d->remaining -= (unsigned long long)n;
                d->stage = DS_FLUSH;
            } else if ((size_t)(dstEnd - dst) >= d->maxBlockSize) {                                   /* :1148-1170: straight into the caller's buffer */
                n = Lizard_decompress_safe((const char*)from, (char*)dst, (int)d->inTarget, (int)d->maxBlockSize);
                if (n < 0) return LZF_ERR(GENERIC);
                if (d->info.contentChecksumFlag) Lizard_XXH32_update(&d->xxh, dst, (size_t)n);
                if (d->info.contentSize) d->remaining -= (unsigned long long)n;
                dst += n;
                d->stage = DS_BLOCK_HEADER;
            } else {                                                                                  /* :1172-1206: wait in the internal buffer for room */
                n = Lizard_decompress_safe((const char*)from, (char*)d->hist, (int)d->inTarget, (int)d->maxBlockSize);
                if (n < 0) return LZF_ERR(decompressionFailed);
                if (d->info.contentChecksumFlag) Lizard_XXH32_update(&d->xxh, d->hist, (size_t)n);
                if (d->info.contentSize) d->remaining -= (unsigned long long)n;
                d->outStart = 0; d->outEnd = (size_t)n;
                d->stage = DS_FLUSH;
            }
            d->inFill = 0;
            break;
        }
        case DS_FLUSH: {                                                                              /* :1208-1228 */
            size_t n = d->outEnd - d->outStart;
            if (n > (size_t)(dstEnd - dst)) n = (size_t)(dstEnd - dst);
            memcpy(dst, d->hist + d->outStart, n);
            d->outStart += n; dst += n;
            if (d->outStart == d->outEnd) { d->stage = DS_BLOCK_HEADER; break; }
            hint = 4;
            more = 0;
            break;
        }
        case DS_SUFFIX: {                                                                             /* :1230-1272 */
            if (d->smallTarget == 0 && d->remaining) return LZF_ERR(frameSize_wrong);
            if (!d->info.contentChecksumFlag) { hint = 0; d->stage = DS_HEADER; more = 0; break; }
            {
                size_t n;
                if (d->smallTarget == 0) { d->smallFill = 0; d->smallTarget = 4; }
                n = 4 - d->smallFill;
                if (n > (size_t)(srcEnd - src)) n = (size_t)(srcEnd - src);
                memcpy(d->small + d->smallFill, src, n);
                d->smallFill += n; src += n;
                if (d->smallFill < 4) { hint = 4 - d->smallFill; more = 0; break; }
                d->smallTarget = 0;
                if (rd32le(d->small) != Lizard_XXH32_digest(&d->xxh)) return LZF_ERR(contentChecksum_invalid);
            }
            hint = 0; d->stage = DS_HEADER; more = 0;
            break;
        }
        case DS_SKIP: {                                                                               /* :1309-1319 */
            size_t n = d->left;
            if (n > (size_t)(srcEnd - src)) n = (size_t)(srcEnd - src);
            src += n; d->left -= n;
            hint = d->left;
            more = 0;
            if (!hint) d->stage = DS_HEADER;
            break;
        }
        default: return LZF_ERR(GENERIC);
        }
    }
    d->srcExpect = src < srcEnd ? src : NULL;                                                        /* :1354-1357 */
    *srcSizePtr = (size_t)(src - srcStart);
    *dstSizePtr = (size_t)(dst - dstStart);
    return hint;
}

/* LizardF_getFrameInfo, lizard_frame.c:871-894 */
size_t LizardF_getFrameInfo(LizardF_decompressionContext_t d, LizardF_frameInfo_t* frameInfoPtr, const void* srcBuffer, size_t* srcSizePtr)
{
    if (d->stage != DS_HEADER) {                                                                      /* already decoded: report it, consume nothing */
        size_t o = 0, i = 0;
        const uint8_t* const keep = d->srcExpect;
        size_t hint;
        *srcSizePtr = 0;
        *frameInfoPtr = d->info;
        d->srcExpect = NULL;
        hint = LizardF_decompress(d, NULL, &o, NULL, &i, NULL);
        d->srcExpect = keep;
        return hint;
    }
    {
        const uint8_t* p = (const uint8_t*)srcBuffer;
        size_t hSize, o = 0, hint;
        if (*srcSizePtr < 5) { *srcSizePtr = 0; return LZF_ERR(frameHeader_incomplete); }            /* LizardF_headerSize, :725-745 */
        if ((rd32le(p) & 0xFFFFFFF0u) == LZF_MAGIC_SKIPPABLE) hSize = 8;
        else if (rd32le(p) != LZF_MAGIC) { *srcSizePtr = 0; return LZF_ERR(frameType_unknown); }
        else hSize = ((p[4] >> 3) & 1u) ? LZF_MAX_HEADER : LZF_MIN_HEADER;
        if (*srcSizePtr < hSize) { *srcSizePtr = 0; return LZF_ERR(frameHeader_incomplete); }
        *srcSizePtr = hSize;
        d->smallTarget = 0;                                                                           /* the header is read from srcBuffer, whole */
        hint = LizardF_decompress(d, NULL, &o, srcBuffer, srcSizePtr, NULL);
        if (LizardF_isError(hint)) return hint;
        if (!d->infoValid) return LZF_ERR(frameHeader_incomplete);
        *frameInfoPtr = d->info;
        return hint;
    }
}
