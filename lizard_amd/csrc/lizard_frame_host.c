/*
 * lizard_frame_host.c — one-shot `.liz` frame production on top of the batched GPU block path
 * (SURVEY.md §8f rank 2).  LizardGPU_compressFrame() writes byte for byte what the reference's
 * LizardF_compressFrame() (lib/lizard_frame.c:260-316) writes for the same preferences, in
 * independent-block mode, when the reference is built with -DLIZARD_RESET_MEM (the zero-state
 * oracle of DESIGN.md §2): header (:403-424), one record per block (:456-469: LE32 size, bit 31 = stored
 * raw when the compressed block does not fit in srcSize-1), end mark and XXH32 content checksum
 * (:651-658).  All full blocks and the ragged last one go through ONE LizardGPU_compressBlocks_host
 * call per chunk instead of one Lizard_compress_extState call per block (:544-556).
 *
 * The symbols carry the LizardGPU_ prefix on purpose: lib/lizard_frame.c holds frame compression and
 * decompression in one object, so a program keeps linking the reference's LizardF_* (decoder included)
 * and calls these where it wants GPU-rate frames.  Linked-block frames are a serial
 * dependency chain between blocks and stay on the reference: they are refused here, never emulated.
 * XXH32 (lib/xxhash/xxhash.c, public algorithm) is restated below; the content checksum is inherently
 * sequential and runs on the host.
 */
#include "../../include/lizard_amd.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LZF_MAGIC            0x184D2206u          /* lizard_frame.c:118 */
#define LZF_RAW_FLAG         0x80000000u          /* :119 */
#define LZF_MAX_HEADER       15u                  /* maxFHSize, :123 */
#define LZF_CHUNK_BYTES      ((size_t)1 << 30)    /* input bytes per batch call */

/* ---- XXH32, seed-parameterised one-shot (xxhash specification) ---- */
#define XP1 2654435761u
#define XP2 2246822519u
#define XP3 3266489917u
#define XP4 668265263u
#define XP5 374761393u
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t rd32le(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t xround(uint32_t acc, uint32_t in) { return rotl32(acc + in * XP2, 13) * XP1; }
static uint32_t xxh32(const void* data, size_t len, uint32_t seed)
{
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* const end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
        const uint8_t* const limit = end - 16;
        do {
            v1 = xround(v1, rd32le(p)); v2 = xround(v2, rd32le(p + 4));
            v3 = xround(v3, rd32le(p + 8)); v4 = xround(v4, rd32le(p + 12));
            p += 16;
        } while (p <= limit);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = seed + XP5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl32(h + rd32le(p) * XP3, 17) * XP4; p += 4; }
    while (p < end) { h = rotl32(h + (uint32_t)*p * XP5, 11) * XP1; p++; }
    h ^= h >> 15; h *= XP2; h ^= h >> 13; h *= XP3; h ^= h >> 16;
    return h;
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

unsigned LizardGPU_frameIsError(size_t code) { return code > (size_t)-(long)LIZARDGPU_FRAME_ERR_maxCode; }   /* :179-182 */

/* LizardF_compressFrameBound, lizard_frame.c:229-248 (+ LizardF_compressBound :432-451 with autoFlush = 1) */
size_t LizardGPU_compressFrameBound(size_t srcSize, const LizardGPU_framePrefs_t* prefsPtr)
{
    LizardGPU_framePrefs_t prefs;
    if (prefsPtr) prefs = *prefsPtr; else memset(&prefs, 0, sizeof prefs);
    {
        const unsigned bsid = optimal_bsid(prefs.frameInfo.blockSizeID, srcSize);
        const size_t blockSize = block_size_of(bsid);
        if (!blockSize) return (size_t)-(long)LIZARDGPU_FRAME_ERR_maxBlockSize_invalid;
        {
            const size_t nbBlocks = srcSize / blockSize + 1, last = srcSize % blockSize;
            const size_t frameEnd = 4 + (size_t)prefs.frameInfo.contentChecksumFlag * 4;
            return LZF_MAX_HEADER + 4 * nbBlocks + blockSize * (nbBlocks - 1) + last + frameEnd;
        }
    }
}

size_t LizardGPU_compressFrame(void* dstBuffer, size_t dstMaxSize, const void* srcBuffer, size_t srcSize,
                               const LizardGPU_framePrefs_t* prefsPtr)
{
    LizardGPU_framePrefs_t prefs;
    uint8_t* const dstStart = (uint8_t*)dstBuffer;
    uint8_t* dst = dstStart;
    const uint8_t* src = (const uint8_t*)srcBuffer;
    size_t blockSize;
    int level;

    if (prefsPtr) prefs = *prefsPtr; else memset(&prefs, 0, sizeof prefs);
    if (prefs.frameInfo.contentSize != 0) prefs.frameInfo.contentSize = (unsigned long long)srcSize;      /* :279-280 */
    prefs.frameInfo.blockSizeID = optimal_bsid(prefs.frameInfo.blockSizeID, srcSize);                       /* :282 */
    blockSize = block_size_of(prefs.frameInfo.blockSizeID);
    if (!blockSize) return (size_t)-(long)LIZARDGPU_FRAME_ERR_maxBlockSize_invalid;
    if (srcSize <= blockSize) prefs.frameInfo.blockMode = 1;                                                /* :284-285 */
    if (prefs.frameInfo.blockMode != 1) return (size_t)-(long)LIZARDGPU_FRAME_ERR_blockMode_invalid;       /* linked: not on the GPU path */
    if (dstMaxSize < LizardGPU_compressFrameBound(srcSize, &prefs)) return (size_t)-(long)LIZARDGPU_FRAME_ERR_dstMaxSize_tooSmall;   /* :289 */
    if (prefs.frameInfo.blockSizeID == 0) prefs.frameInfo.blockSizeID = 1;                                  /* :385 */

    level = prefs.compressionLevel;                                                                         /* Lizard_createStream clamps, lizard_compress.c:303-308 */
    if (level > LIZARD_MAX_CLEVEL) level = LIZARD_MAX_CLEVEL;
    if (level < LIZARD_MIN_CLEVEL) level = LIZARD_DEFAULT_CLEVEL;
    if (!LizardGPU_levelSupported(level)) return (size_t)-(long)LIZARDGPU_FRAME_ERR_compressionLevel_invalid;

    /* header, lizard_frame.c:403-424 */
    wr32le(dst, LZF_MAGIC); dst += 4;
    {
        uint8_t* const headerStart = dst;
        *dst++ = (uint8_t)((1u << 6) + ((prefs.frameInfo.blockMode & 1u) << 5) + ((prefs.frameInfo.contentChecksumFlag & 1u) << 2)
                           + ((prefs.frameInfo.contentSize > 0) << 3));
        *dst++ = (uint8_t)((prefs.frameInfo.blockSizeID & 7u) << 4);
        if (prefs.frameInfo.contentSize) {
            wr32le(dst, (uint32_t)prefs.frameInfo.contentSize); wr32le(dst + 4, (uint32_t)(prefs.frameInfo.contentSize >> 32));
            dst += 8;
        }
        *dst = (uint8_t)(xxh32(headerStart, (size_t)(dst - headerStart), 0) >> 8);                          /* :219-223 */
        dst++;
    }

    /* blocks, lizard_frame.c:544-556 + :456-469, in chunks of whole blocks */
    if (srcSize) {
        const size_t stride = ((size_t)LIZARD_COMPRESSBOUND((int)blockSize) + 63) & ~(size_t)63;
        size_t perChunk = LZF_CHUNK_BYTES / blockSize, done = 0;
        const size_t nBlocks = (srcSize + blockSize - 1) / blockSize;
        uint8_t* slots;
        uint32_t* cs;
        if (perChunk == 0) perChunk = 1;
        if (perChunk > nBlocks) perChunk = nBlocks;
        slots = (uint8_t*)malloc(perChunk * stride);
        cs = (uint32_t*)malloc(perChunk * sizeof(uint32_t));
        if (!slots || !cs) { free(slots); free(cs); return (size_t)-(long)LIZARDGPU_FRAME_ERR_allocation_failed; }
        while (done < nBlocks) {
            const size_t nb = nBlocks - done < perChunk ? nBlocks - done : perChunk;
            const uint8_t* const chunk = src + done * blockSize;
            const size_t chunkBytes = (done + nb == nBlocks) ? srcSize - done * blockSize : nb * blockSize;
            const size_t last = chunkBytes - (nb - 1) * blockSize;
            size_t i;
            if (LizardGPU_compressBlocks_host(chunk, nb, blockSize, last, slots, stride, cs, level) != 0) {
                free(slots); free(cs);
                return (size_t)-(long)LIZARDGPU_FRAME_ERR_GENERIC;
            }
            for (i = 0; i < nb; i++) {
                const size_t n = (i + 1 == nb) ? last : blockSize;
                /* :461-467: the block call gets maxDstSize = srcSize-1 and a 0 return means "store raw".  A
                 * 1-byte block is the exception the reference makes by accident: maxDstSize 0 puts oend before
                 * the sub-block start, the unsigned room test of lizard_compress.c:238 wraps, and the 6-byte
                 * "compressed" block is emitted (level, raw marker, LE24 1, the byte). */
                if (n != 1 && (cs[i] == 0 || cs[i] > n - 1)) {
                    wr32le(dst, (uint32_t)n | LZF_RAW_FLAG);
                    memcpy(dst + 4, chunk + i * blockSize, n); dst += 4 + n;
                } else {
                    wr32le(dst, cs[i]);
                    memcpy(dst + 4, slots + i * stride, cs[i]); dst += 4 + cs[i];
                }
            }
            done += nb;
        }
        free(slots); free(cs);
    }

    /* end mark + content checksum, lizard_frame.c:651-658 */
    wr32le(dst, 0); dst += 4;
    if (prefs.frameInfo.contentChecksumFlag == 1) { wr32le(dst, xxh32(src, srcSize, 0)); dst += 4; }
    return (size_t)(dst - dstStart);
}
