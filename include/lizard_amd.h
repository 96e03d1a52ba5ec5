/*
 * lizard_amd.h — C ABI of liblizard_amd.so, the MI355X-native Lizard block-compress path.
 *
 * Part 1 is the reference's own block-compression ABI (same names, argument meaning and error
 * behaviour), so that a program written against inikep/lizard's lib/lizard_compress.h links against
 * this library unchanged.  Each entry cites the reference declaration it replaces.
 * Part 2 is the batch extension the GPU needs (the reference API is one-block-synchronous; see
 * SURVEY.md §8b): many independent blocks per call, host- or device-resident.
 *
 * Plain C, plain pointers and sizes; no HIP or torch types in any signature (a HIP stream is passed
 * as an opaque void*).
 */
#ifndef LIZARD_AMD_H
#define LIZARD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ======================= Part 1: reference block API (lib/lizard_compress.h) ======================= */

#define LIZARD_MIN_CLEVEL      10            /* reference lib/lizard_compress.h:86 */
#define LIZARD_MAX_CLEVEL      49            /* reference lib/lizard_compress.h:88 */
#define LIZARD_DEFAULT_CLEVEL  17            /* reference lib/lizard_compress.h:92 */
#define LIZARD_MAX_INPUT_SIZE  0x7E000000    /* reference lib/lizard_compress.h:121 */
#define LIZARD_BLOCK_SIZE      (1<<17)       /* reference lib/lizard_compress.h:122 */
#define LIZARD_COMPRESSBOUND(isize)  ((unsigned)(isize) > (unsigned)LIZARD_MAX_INPUT_SIZE ? 0 : (isize) + 1 + 1 + ((isize/LIZARD_BLOCK_SIZE)+1)*4)   /* :124 */

typedef struct Lizard_stream_s Lizard_stream_t;   /* opaque, reference lib/lizard_compress.h:94 */

/* reference lib/lizard_compress.h:66 / lib/lizard_compress.c:66 */
int Lizard_versionNumber(void);

/* reference lib/lizard_compress.h:99 / lib/lizard_compress.c:596.
 * Returns bytes written into dst, 0 on failure (dst too small, level not accelerated, GPU error).
 * Never writes past dst+maxDstSize, never reads outside src[0..srcSize).  (maxDstSize <= 0 returns 0;
 * the reference's room test wraps there, lizard_compress.c:238/:489, and it writes past the buffer.) */
int Lizard_compress(const char* src, char* dst, int srcSize, int maxDstSize, int compressionLevel);

/* reference lib/lizard_compress.h:135 / lib/lizard_compress.c:67 */
int Lizard_compressBound(int inputSize);

/* reference lib/lizard_compress.h:145-147 / lib/lizard_compress.c:311,583.
 * `state` must be pointer-aligned (misaligned => 0, as the reference) and Lizard_sizeofState(level)
 * bytes; its content on entry is irrelevant (the reference re-initialises it on every call, and the
 * match-finder tables of this implementation live on the device). */
int Lizard_sizeofState(int compressionLevel);
int Lizard_compress_extState(void* state, const char* src, char* dst, int srcSize, int maxDstSize, int compressionLevel);

/* reference lib/lizard_compress.h:159-160,166 / lib/lizard_compress.c:392,417,401 */
Lizard_stream_t* Lizard_createStream(int compressionLevel);
int              Lizard_freeStream(Lizard_stream_t* streamPtr);
Lizard_stream_t* Lizard_resetStream(Lizard_stream_t* streamPtr, int compressionLevel);

/* reference lib/lizard_common.h:493-497 / lib/lizard_compress.c:68,612-630 (level-10 aliases) */
int Lizard_sizeofState_MinLevel(void);
int Lizard_compress_MinLevel(const char* source, char* dest, int sourceSize, int maxDestSize);
int Lizard_compress_extState_MinLevel(void* state, const char* source, char* dest, int inputSize, int maxDestSize);
Lizard_stream_t* Lizard_resetStream_MinLevel(Lizard_stream_t* streamPtr);
Lizard_stream_t* Lizard_createStream_MinLevel(void);

/* reference lib/lizard_compress.h:178,198,188 / lib/lizard_compress.c:426,454,550 — linked blocks and
 * dictionaries.  Present so that reference programs link unchanged; NOT implemented on this path (a serial
 * chain between blocks): each returns 0, the reference's failure value, after one message on stderr. */
int Lizard_loadDict(Lizard_stream_t* streamPtr, const char* dictionary, int dictSize);
int Lizard_saveDict(Lizard_stream_t* streamPtr, char* safeBuffer, int dictSize);
int Lizard_compress_continue(Lizard_stream_t* streamPtr, const char* src, char* dst, int srcSize, int maxDstSize);

/* ======================= Part 2: batch extension (ours) ======================= */

/* Error codes returned (negated) by the LizardGPU_* entry points. */
enum {
    LIZARDGPU_OK = 0,
    LIZARDGPU_ERR_NO_DEVICE = 1,     /* no HIP device / HIP runtime failure at init */
    LIZARDGPU_ERR_LEVEL = 2,         /* level not implemented on the GPU path */
    LIZARDGPU_ERR_ARG = 3,           /* bad size / stride / null pointer */
    LIZARDGPU_ERR_HIP = 4,           /* a HIP call failed (see LizardGPU_lastError) */
    LIZARDGPU_ERR_NOMEM = 5
};

/* 1 if `compressionLevel` (after the reference's clamp, lib/lizard_compress.c:303-308) runs on the GPU:
 * 10 (fastSmall), 11 (fast, blocks <= 4 MiB), 13..17 (hashChain, blocks <= 4 MiB), 21, 22 (priceFast, blocks
 * < 16 MiB) and their huff0 twins 30, 31, 34..38, 41, 42 — every row of Lizard_defaultParameters
 * (lib/lizard_common.h:234-284) whose parser is fastSmall, fast, hashChain or priceFast. */
int LizardGPU_levelSupported(int compressionLevel);

/* Select the HIP device used by this process for subsequent calls (default: current device 0).
 * One process per GPU is the intended deployment (torch.distributed / RCCL rank = device). */
int LizardGPU_setDevice(int device);

/* Human-readable text of the last HIP failure on the calling thread's context ("" if none). */
const char* LizardGPU_lastError(void);

/* Compress nBlocks independent blocks that are already RESIDENT IN DEVICE MEMORY.
 *   d_src      : device pointer, block i starts at d_src + i*blockSize; every block is blockSize bytes
 *                except the last, which is lastBlockSize (1..blockSize)
 *   d_dst      : device pointer, block i's output starts at d_dst + i*dstStride;
 *                dstStride >= Lizard_compressBound(blockSize)
 *   d_sizes    : device pointer to nBlocks uint32: compressed size of each block (never 0: with a
 *                bound-sized slot compression cannot fail, reference lib/lizard_compress.h:104)
 *   level      : 10..49, must satisfy LizardGPU_levelSupported
 *   stream     : hipStream_t as void* (NULL = default stream). The call only enqueues work; outputs are
 *                valid after the stream is synchronised.  Launches of one process share the per-wave scratch
 *                arena and the tables: a launch on a different stream waits on the device for the previous
 *                one (stream-ordered, no host synchronisation).
 * Each block is what Lizard_compress_extState() produces on a zero-initialised state
 * (reference lib/lizard_compress.c:583 built with -DLIZARD_RESET_MEM). Returns 0 or -LIZARDGPU_ERR_*. */
int LizardGPU_compressBlocks_device(const void* d_src, size_t nBlocks, size_t blockSize, size_t lastBlockSize,
                                    void* d_dst, size_t dstStride, uint32_t* d_sizes,
                                    int level, void* stream);

/* Same for HOST-resident buffers: stages src to the device, runs the kernels, copies sizes and payload
 * back (dst slot i at dst + i*dstStride, cSizes[i] bytes valid). Synchronous. */
int LizardGPU_compressBlocks_host(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize,
                                  void* dst, size_t dstStride, uint32_t* cSizes, int level);

/* Synthetic input, the reference's benchmark generator (programs/datagen.c:153 RDG_genBuffer).
 * Host: fills buffer[0..size) exactly like RDG_genBuffer(buffer, size, matchProba, litProba, seed).
 * Device: block b (b < nBlocks, blockSize bytes each, back to back at d_dst) is
 * RDG_genBuffer(blockSize, matchProba, litProba, seed0 + b); synchronous. */
void LizardGPU_datagen_host(void* buffer, size_t size, double matchProba, double litProba, unsigned seed);
int  LizardGPU_datagen_device(void* d_dst, size_t nBlocks, size_t blockSize, double matchProba, double litProba,
                              unsigned seed0, void* stream);

/* Kernel-only duration of the most recent LizardGPU_compressBlocks_* call measured with HIP events on
 * the launch stream, in milliseconds (blocks until that launch finished). < 0 if unavailable. */
float LizardGPU_lastKernelMs(void);

/* Number of resident waves (= blocks compressed concurrently) the launcher uses on this device. */
int LizardGPU_residentWaves(void);

/* =====================================================================================================
 * Part 3 — one-shot frame production on the batched GPU path.
 *
 * LizardGPU_compressFrame() writes byte for byte what the reference's LizardF_compressFrame()
 * (lib/lizard_frame.h:134, lib/lizard_frame.c:260-316) writes for the same preferences in
 * independent-block mode (reference built with -DLIZARD_RESET_MEM, the zero-state oracle): frame header,
 * LE32-sized block records (bit 31 = stored raw), end mark, XXH32 content checksum.  Every block of the
 * frame goes through the batched kernels instead of one Lizard_compress_extState call per block.
 * The reference keeps frame compression and decompression in ONE object (lizard_frame.c), so these
 * symbols are prefixed LizardGPU_ and coexist with a linked reference LizardF_* (INTEGRATION.md §3).
 *
 * LizardGPU_framePrefs_t is layout-identical to LizardF_preferences_t (lib/lizard_frame.h:111-125): a
 * caller holding the reference type passes (const LizardGPU_framePrefs_t*)&prefs.
 * Return value: bytes written, or an error code that LizardGPU_frameIsError() recognises; codes are the
 * reference's (size_t)-LizardF_ERROR_* values (lib/lizard_frame_static.h:57-67).  Refused, never emulated:
 * linked-block frames larger than one block (blockMode_invalid) and levels without a GPU kernel
 * (compressionLevel_invalid).
 * ===================================================================================================== */
typedef struct {
    unsigned           blockSizeID;          /* LizardF_blockSizeID_t: 0 = default (128 KiB), 1..7 = 128K,256K,1M,4M,16M,64M,256M */
    unsigned           blockMode;            /* LizardF_blockMode_t: 0 = linked, 1 = independent */
    unsigned           contentChecksumFlag;  /* 0 / 1 */
    unsigned           frameType;            /* 0 = frame */
    unsigned long long contentSize;          /* != 0: the header carries srcSize */
    unsigned           reserved[2];
} LizardGPU_frameInfo_t;                     /* == LizardF_frameInfo_t */

typedef struct {
    LizardGPU_frameInfo_t frameInfo;
    int      compressionLevel;               /* clamped like Lizard_createStream (lib/lizard_compress.c:303-308) */
    unsigned autoFlush;                      /* ignored: one-shot frames always flush (lizard_frame.c:283) */
    unsigned reserved[4];
} LizardGPU_framePrefs_t;                    /* == LizardF_preferences_t */

enum {                                       /* LizardF_errorCodes, lib/lizard_frame_static.h:57-67 */
    LIZARDGPU_FRAME_ERR_GENERIC = 1, LIZARDGPU_FRAME_ERR_maxBlockSize_invalid = 2, LIZARDGPU_FRAME_ERR_blockMode_invalid = 3,
    LIZARDGPU_FRAME_ERR_compressionLevel_invalid = 5, LIZARDGPU_FRAME_ERR_allocation_failed = 9,
    LIZARDGPU_FRAME_ERR_dstMaxSize_tooSmall = 11, LIZARDGPU_FRAME_ERR_maxCode = 19
};

/* replaces LizardF_compressFrameBound, lib/lizard_frame.h:122 / lizard_frame.c:229 (same value) */
size_t LizardGPU_compressFrameBound(size_t srcSize, const LizardGPU_framePrefs_t* preferencesPtr);
/* replaces LizardF_compressFrame, lib/lizard_frame.h:134 / lizard_frame.c:260 (host buffers, synchronous) */
size_t LizardGPU_compressFrame(void* dstBuffer, size_t dstMaxSize, const void* srcBuffer, size_t srcSize,
                               const LizardGPU_framePrefs_t* preferencesPtr);
/* replaces LizardF_isError, lib/lizard_frame.h:59 / lizard_frame.c:179 */
unsigned LizardGPU_frameIsError(size_t code);

#ifdef __cplusplus
}
#endif
#endif /* LIZARD_AMD_H */
