/*
 * lizard_amd.h — C ABI of liblizard_amd.so, the MI355X-native Lizard block-compress path.
 *
 * The library is a link-time replacement of the reference's liblizard: it exports every symbol of
 * lib/dll/liblizard.def, of lib/lizard_frame.h and the Lizard_XXH* hash functions the reference's programs use, with the
 * reference's names, argument meaning and error behaviour — the sources under programs/, tests/fuzzer.c, tests/frametest.c and
 * tests/fullbench.c link against it with no reference object at all (oracle/Makefile) — plus the LizardGPU_*
 * extension.  Each entry cites the reference declaration it replaces.
 *   Part 1   block compression      lib/lizard_compress.h       on the GPU, no host path
 *   Part 1b  block decompression    lib/lizard_decompress.h     one block per call: on the calling thread (not the accelerated path)
 *   Part 1c  frames                 lib/lizard_frame.h          compression batched on the GPU; decompression on the host
 *   Part 1d  XXH32 / XXH64          lib/xxhash/xxhash.h         (XXH_NAMESPACE=Lizard_, lib/Makefile:52)
 *   Part 2   batch extension (ours): many independent blocks per call, host- or device-resident, several GPUs, decompression
 *   Part 3   the strict frame twins of rounds 2-3 (LizardGPU_compressFrame ...)
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
 * Never writes past dst+maxDstSize, never reads outside src[0..srcSize).  (maxDstSize <= 0 returns 0, where the
 * reference's room test wraps, lizard_compress.c:238/:489, and it writes past the buffer — with the one exception its own
 * frame layer produces: a 1-byte block with maxDstSize 0 (lizard_frame.c:461) returns the reference's 6 bytes.) */
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

/* reference lib/lizard_compress.h:178,198,188 / lib/lizard_compress.c:426,454,550 — streaming ("linked blocks") and
 * dictionaries.  In the reference a stream is one serial chain (hash table, window and repeat offset carried from call to
 * call) — nothing a GPU can parallelise.  Implemented HISTORY-FREE: Lizard_compress_continue compresses its block on the GPU
 * without referring to earlier data (same return contract as Lizard_compress_extState at the stream's level); loadDict and
 * saveDict do the reference's book-keeping (returned sizes; saveDict copies the last min(dictSize, prefix) bytes of the
 * previous block into safeBuffer).  Every block is valid for Lizard_decompress_safe_continue / _usingDict — a decoder never
 * requires a block to use its history — so frames in the frame layer's default linked mode are compressed and decodable;
 * their bytes are those of independent blocks, not the reference's linked-mode bytes. */
int Lizard_loadDict(Lizard_stream_t* streamPtr, const char* dictionary, int dictSize);
int Lizard_saveDict(Lizard_stream_t* streamPtr, char* safeBuffer, int dictSize);
int Lizard_compress_continue(Lizard_stream_t* streamPtr, const char* src, char* dst, int srcSize, int maxDstSize);

/* ======================= Part 1b: reference block decompression (lib/lizard_decompress.h) =======================
 * One block per call, on the calling thread (lizard_amd/csrc/lizard_decode_host.c): the decoder is not the path this library
 * accelerates, and one block is ~0.1 ms of host work against ~1 ms as a single wavefront.  Many independent blocks:
 * LizardGPU_decompressBlocks_host / _device below.  Same results as the reference for every VALID block (every block a Lizard
 * compressor produces: offsets >= 8).  Malformed input is refused (negative result) without reading outside
 * source[0..compressedSize) or writing outside dest[0..maxDecompressedSize) — and that is where the identity ends: an offset of
 * 0 is refused here, offsets 1..7 copy with true LZ overlap semantics (the reference's 8- and 16-byte wild copies produce a
 * different byte pattern for them), and a literal run that overruns the literals stream is refused (the reference has no such
 * test).  tests/decode_fuzz.c runs the decoder under ASan / UBSan on damaged blocks. */
typedef struct Lizard_streamDecode_s Lizard_streamDecode_t;   /* 4 words, valid when zeroed (reference lib/lizard_common.h:195-200) */
/* reference lib/lizard_decompress.h:64 / lib/lizard_decompress.c:267 */
int Lizard_decompress_safe(const char* source, char* dest, int compressedSize, int maxDecompressedSize);
/* reference lib/lizard_decompress.h:80 / lib/lizard_decompress.c:272: may stop once targetOutputSize bytes exist (returns >= that) */
int Lizard_decompress_safe_partial(const char* source, char* dest, int compressedSize, int targetOutputSize, int maxDecompressedSize);
/* reference lib/lizard_decompress.h:102-103,110 / lib/lizard_decompress.c:288-314 */
Lizard_streamDecode_t* Lizard_createStreamDecode(void);
int Lizard_freeStreamDecode(Lizard_streamDecode_t* Lizard_stream);
int Lizard_setStreamDecode(Lizard_streamDecode_t* Lizard_streamDecode, const char* dictionary, int dictSize);
/* reference lib/lizard_decompress.h:131 / lib/lizard_decompress.c:325: previous output is the history (prefix if contiguous) */
int Lizard_decompress_safe_continue(Lizard_streamDecode_t* Lizard_streamDecode, const char* source, char* dest, int compressedSize, int maxDecompressedSize);
/* reference lib/lizard_decompress.h:142 / lib/lizard_decompress.c:360, :374 */
int Lizard_decompress_safe_usingDict(const char* source, char* dest, int compressedSize, int maxDecompressedSize, const char* dictStart, int dictSize);
int Lizard_decompress_safe_forceExtDict(const char* source, char* dest, int compressedSize, int maxOutputSize, const char* dictStart, int dictSize);

/* ======================= Part 1c: reference frame API (lib/lizard_frame.h) =======================
 * Types as lib/lizard_frame.h:59-234 declares them (define LIZARD_AMD_NO_FRAME_TYPES before this header when the reference's
 * lizard_frame.h is included too).  Compression: every run of blocks an Update / compressFrame call covers is ONE batch on the
 * GPU (lizard_amd/csrc/lizard_frame_host.c).  Independent blocks: byte for byte the reference's frames (zero-state build).
 * Linked blocks: valid frames whose blocks do not use their history (Lizard_compress_continue above).  A level without a GPU
 * kernel or a GPU failure stores the blocks raw, as the reference's frame layer does when Lizard_compress_extState returns 0
 * (lib/lizard_frame.c:456-469), after a line on stderr. */
#ifndef LIZARD_AMD_NO_FRAME_TYPES
typedef size_t LizardF_errorCode_t;
typedef enum { LizardF_default = 0, LizardF_max128KB = 1, LizardF_max256KB = 2, LizardF_max1MB = 3, LizardF_max4MB = 4,
               LizardF_max16MB = 5, LizardF_max64MB = 6, LizardF_max256MB = 7 } LizardF_blockSizeID_t;
typedef enum { LizardF_blockLinked = 0, LizardF_blockIndependent } LizardF_blockMode_t;
typedef enum { LizardF_noContentChecksum = 0, LizardF_contentChecksumEnabled } LizardF_contentChecksum_t;
typedef enum { LizardF_frame = 0, LizardF_skippableFrame } LizardF_frameType_t;
typedef struct {
    LizardF_blockSizeID_t     blockSizeID;          /* 0 = default (128 KiB) */
    LizardF_blockMode_t       blockMode;            /* 0 = linked (default), 1 = independent */
    LizardF_contentChecksum_t contentChecksumFlag;
    LizardF_frameType_t       frameType;
    unsigned long long        contentSize;          /* != 0: the header carries the content size */
    unsigned                  reserved[2];
} LizardF_frameInfo_t;                              /* lib/lizard_frame.h:111-118 */
typedef struct {
    LizardF_frameInfo_t frameInfo;
    int      compressionLevel;                      /* clamped like Lizard_createStream (lib/lizard_compress.c:303-308) */
    unsigned autoFlush;                             /* 1 = every Update call ends on a block boundary (no buffering) */
    unsigned reserved[4];
} LizardF_preferences_t;                            /* lib/lizard_frame.h:120-125 */
typedef struct LizardF_cctx_s* LizardF_compressionContext_t;      /* lib/lizard_frame.h:148 */
typedef struct { unsigned stableSrc; unsigned reserved[3]; } LizardF_compressOptions_t;      /* :150-154 */
typedef struct LizardF_dctx_s* LizardF_decompressionContext_t;    /* :229 */
typedef struct { unsigned stableDst; unsigned reserved[3]; } LizardF_decompressOptions_t;    /* :231-234 */
#define LIZARDF_VERSION 100
#endif
/* reference lib/lizard_frame.h:59-60 / lib/lizard_frame.c:179-189; codes are (size_t)-LizardF_ERROR_* (lib/lizard_frame_static.h:57-67) */
unsigned    LizardF_isError(size_t code);
const char* LizardF_getErrorName(size_t code);
/* reference lib/lizard_frame.h:131,143 / lib/lizard_frame.c:229,260 */
size_t LizardF_compressFrameBound(size_t srcSize, const LizardF_preferences_t* preferencesPtr);
size_t LizardF_compressFrame(void* dstBuffer, size_t dstMaxSize, const void* srcBuffer, size_t srcSize, const LizardF_preferences_t* preferencesPtr);
/* reference lib/lizard_frame.h:159-160,172,181,193,205,216 / lib/lizard_frame.c:329,346,362,432,501,610,651 */
size_t LizardF_createCompressionContext(LizardF_compressionContext_t* cctxPtr, unsigned version);
size_t LizardF_freeCompressionContext(LizardF_compressionContext_t cctx);
size_t LizardF_compressBegin(LizardF_compressionContext_t cctx, void* dstBuffer, size_t dstMaxSize, const LizardF_preferences_t* prefsPtr);
size_t LizardF_compressBound(size_t srcSize, const LizardF_preferences_t* prefsPtr);
size_t LizardF_compressUpdate(LizardF_compressionContext_t cctx, void* dstBuffer, size_t dstMaxSize, const void* srcBuffer, size_t srcSize,
                              const LizardF_compressOptions_t* cOptPtr);
size_t LizardF_flush(LizardF_compressionContext_t cctx, void* dstBuffer, size_t dstMaxSize, const LizardF_compressOptions_t* cOptPtr);
size_t LizardF_compressEnd(LizardF_compressionContext_t cctx, void* dstBuffer, size_t dstMaxSize, const LizardF_compressOptions_t* cOptPtr);
/* reference lib/lizard_frame.h:247-248,265,297 / lib/lizard_frame.c:689,699,871,980.  Blocks are decoded by Part 1b's decoder;
 * the context keeps the 16 MiB history of a linked frame itself (stableDst is accepted and not needed). */
size_t LizardF_createDecompressionContext(LizardF_decompressionContext_t* dctxPtr, unsigned version);
size_t LizardF_freeDecompressionContext(LizardF_decompressionContext_t dctx);
size_t LizardF_getFrameInfo(LizardF_decompressionContext_t dctx, LizardF_frameInfo_t* frameInfoPtr, const void* srcBuffer, size_t* srcSizePtr);
size_t LizardF_decompress(LizardF_decompressionContext_t dctx, void* dstBuffer, size_t* dstSizePtr, const void* srcBuffer, size_t* srcSizePtr,
                          const LizardF_decompressOptions_t* dOptPtr);

/* ======================= Part 1d: XXH32 / XXH64 (lib/xxhash/xxhash.h with XXH_NAMESPACE=Lizard_) =======================
 * programs/bench.c, tests/fuzzer.c and tests/frametest.c call these; the state structs are the caller's (the reference's
 * XXH32_state_t / XXH64_state_t: 48 / 88 bytes), this library's layouts fit inside them.  XXH_errorcode: 0 = ok. */
struct Lizard_XXH32_state_s; struct Lizard_XXH64_state_s;
unsigned Lizard_XXH_versionNumber(void);
unsigned Lizard_XXH32(const void* input, size_t length, unsigned seed);                                   /* xxhash.h:167 */
struct Lizard_XXH32_state_s* Lizard_XXH32_createState(void);                                             /* :171-177 */
int      Lizard_XXH32_freeState(struct Lizard_XXH32_state_s* statePtr);
void     Lizard_XXH32_copyState(struct Lizard_XXH32_state_s* dst, const struct Lizard_XXH32_state_s* src);
int      Lizard_XXH32_reset(struct Lizard_XXH32_state_s* statePtr, unsigned seed);
int      Lizard_XXH32_update(struct Lizard_XXH32_state_s* statePtr, const void* input, size_t length);
unsigned Lizard_XXH32_digest(const struct Lizard_XXH32_state_s* statePtr);
void     Lizard_XXH32_canonicalFromHash(unsigned char* dst, unsigned hash);                              /* :204-205 (big-endian bytes) */
unsigned Lizard_XXH32_hashFromCanonical(const unsigned char* src);
unsigned long long Lizard_XXH64(const void* input, size_t length, unsigned long long seed);              /* :225 */
struct Lizard_XXH64_state_s* Lizard_XXH64_createState(void);                                             /* :229-235 */
int      Lizard_XXH64_freeState(struct Lizard_XXH64_state_s* statePtr);
void     Lizard_XXH64_copyState(struct Lizard_XXH64_state_s* dst, const struct Lizard_XXH64_state_s* src);
int      Lizard_XXH64_reset(struct Lizard_XXH64_state_s* statePtr, unsigned long long seed);
int      Lizard_XXH64_update(struct Lizard_XXH64_state_s* statePtr, const void* input, size_t length);
unsigned long long Lizard_XXH64_digest(const struct Lizard_XXH64_state_s* statePtr);
void     Lizard_XXH64_canonicalFromHash(unsigned char* dst, unsigned long long hash);                    /* :239-240 */
unsigned long long Lizard_XXH64_hashFromCanonical(const unsigned char* src);

/* ======================= Part 2: batch extension (ours) ======================= */

/* Error codes returned (negated) by the LizardGPU_* entry points. */
enum {
    LIZARDGPU_OK = 0,
    LIZARDGPU_ERR_NO_DEVICE = 1,     /* no HIP device / HIP runtime failure at init */
    LIZARDGPU_ERR_LEVEL = 2,         /* level not implemented on the GPU path */
    LIZARDGPU_ERR_ARG = 3,           /* bad size / stride / null pointer */
    LIZARDGPU_ERR_HIP = 4,           /* a HIP call failed (see LizardGPU_lastError) */
    LIZARDGPU_ERR_NOMEM = 5,         /* device or pinned-host allocation failed */
    LIZARDGPU_ERR_RCCL = 6,          /* RCCL could not be loaded or a collective failed */
    LIZARDGPU_ERR_HARDWARE = 7       /* the device failed the library's self-check of a property the level's kernel relies on:
                                      * lanes of one LDS atomic that hit the same dword are served in ascending lane order
                                      * (checked once per device at context creation; levels 10/30 and the hashChain levels) */
};

/* 1 if `compressionLevel` (after the reference's clamp, lib/lizard_compress.c:303-308) runs on the GPU:
 * 10 (fastSmall), 11 (fast), 12 (noChain), 13..17 (hashChain), 20 (fastBig), 21, 22 (priceFast) and their huff0 twins 30, 31, 32, 33,
 * 34..38, 40, 41, 42 — every row of Lizard_defaultParameters (lib/lizard_common.h:234-284) whose parser is fastSmall, fast, noChain, hashChain,
 * fastBig or priceFast (23 of the 40 levels; the others are the price-based parsers),
 * at every block size the reference takes (LIZARD_MAX_INPUT_SIZE, lib/lizard_compress.h:121). */
int LizardGPU_levelSupported(int compressionLevel);
/* Largest block (bytes) the GPU path takes at this level: LIZARD_MAX_INPUT_SIZE, or 0 if the level has no GPU kernel. */
size_t LizardGPU_maxBlockSize(int compressionLevel);

/* Devices.  The library keeps one context per device (arenas, tables, streams, pinned staging), created on first
 * use and serialised by its own lock; host threads working on different devices run concurrently.
 * LizardGPU_setDevice selects the device for the CALLING THREAD's later calls and becomes the default of threads
 * that never selected one (initially device 0); an index outside 0..deviceCount-1 is refused (-LIZARDGPU_ERR_ARG).
 * Every entry point makes its device current for the duration of the call and restores the caller's.
 * LizardGPU_shutdown releases every device allocation, stream, pinned buffer and RCCL communicator of the
 * process (synchronises the devices first); the next call re-creates what it needs. */
int  LizardGPU_deviceCount(void);
int  LizardGPU_setDevice(int device);
void LizardGPU_shutdown(void);

/* Text of the calling thread's last failure ("" after a successful call; thread-local storage). */
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

/* Same for HOST-resident buffers (dst slot i at dst + i*dstStride, cSizes[i] bytes valid).  Synchronous.
 * Pipelined in chunks of whole blocks (256 MiB of input, LIZARDGPU_CHUNK_MB overrides), three chunks in flight: pinned
 * staging, H2D, kernels, device-side compaction of the valid bytes, ONE D2H per chunk.  The calling thread stages and issues,
 * a second host thread (alive for the duration of the call) drains: uploads, kernels, downloads and the host copies of
 * different chunks run side by side.  A src that is already pinned (hipHostMalloc / hipHostRegister) is read by DMA directly.
 * _host_packed writes the compressed blocks back to back instead: block i at dst + offsets[i], cSizes[i] bytes,
 * offsets[nBlocks] = total (offsets / cSizes may be NULL); -LIZARDGPU_ERR_ARG if dstCapacity is too small. */
int LizardGPU_compressBlocks_host(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize,
                                  void* dst, size_t dstStride, uint32_t* cSizes, int level);
int LizardGPU_compressBlocks_host_packed(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize,
                                         void* dst, size_t dstCapacity, uint64_t* offsets, uint32_t* cSizes, int level);

/* ---- several GPUs ----
 * Blocks are independent: rank r of R owns the contiguous range LizardGPU_shardRange gives it, no halo; the one
 * exchange is the RCCL all-gather of the per-block compressed sizes (uint32 per block, over xGMI inside a node),
 * followed by an exclusive prefix sum = byte offset of every block in the concatenated output.
 *
 * One process driving nDevices GPUs: device devices[r] (r if devices == NULL) holds its blocks back to back at d_src[r]
 * and gets its slots at d_dst[r]; on return (synchronous) every device's d_allSizes[r][0..nBlocks) holds ALL sizes
 * and d_offsets[r][0..nBlocks] the offsets (last entry = total).  The ragged last block belongs to the last rank.
 *
 * One process per GPU (torchrun): rank 0 calls LizardGPU_commUniqueId (128 bytes), the launcher's transport carries
 * them to the other ranks, every rank calls LizardGPU_commInitRank once, then LizardGPU_gatherSizes_device after each
 * batch: d_localSizes = this rank's shard (may already sit at d_allSizes + first), enqueued on `stream`.
 * LizardGPU_offsetsFromSizes is the host form of the prefix sum (offsets[nBlocks] = total). */
void LizardGPU_shardRange(size_t nBlocks, int rank, int nRanks, size_t* first, size_t* count);
void LizardGPU_offsetsFromSizes(const uint32_t* sizes, size_t nBlocks, uint64_t* offsets);
int  LizardGPU_compressBlocks_sharded(int nDevices, const int* devices, const void* const* d_src, size_t nBlocks,
                                      size_t blockSize, size_t lastBlockSize, void* const* d_dst, size_t dstStride,
                                      uint32_t* const* d_allSizes, uint64_t* const* d_offsets, int level);
int  LizardGPU_commUniqueId(void* id128);
int  LizardGPU_commInitRank(const void* id128, int nRanks, int rank);
int  LizardGPU_gatherSizes_device(const uint32_t* d_localSizes, size_t nBlocks, uint32_t* d_allSizes,
                                  uint64_t* d_offsets, void* stream);
int  LizardGPU_commDestroy(void);
/* The transport of the size exchange is a table of four calls (u32 elements; `comm` and `stream` are passed through as the
 * library received them; allGather is in place when send == recv + rank*count, as ncclAllGather; every call returns 0 or a
 * negative LIZARDGPU_ERR_*).  RCCL fills it by default — resolved from a copy the process has already mapped (torch's in a
 * torchrun job) before librccl.so.1 is loaded, LizardGPU_rcclShared() = 1 / 0 / -1 (not resolved yet).  LizardGPU_setCollectives
 * installs another transport (NULL: back to RCCL; a table with a NULL member is refused); with one installed NO RCCL is involved:
 * LizardGPU_compressBlocks_sharded and LizardGPU_gatherSizes_device pass the rank index as `comm`, and LizardGPU_commInitRank only
 * records rank and rank count (its id argument is ignored).  The exchange logic itself (lizard_amd/csrc/lizard_shard_core.h) is transport-agnostic:
 * tests/shard_fake.cpp runs it with 2 and 3 ranks over shared memory on a CPU. */
typedef struct {
    int (*allGather)(const void* send, void* recv, size_t count, void* comm, void* stream);
    int (*broadcast)(const void* send, void* recv, size_t count, int root, void* comm, void* stream);
    int (*groupStart)(void);
    int (*groupEnd)(void);
} LizardGPU_Collectives;
int  LizardGPU_setCollectives(const LizardGPU_Collectives* table);
int  LizardGPU_rcclShared(void);
/* What carries the size exchange of this process, as the transport itself reports it — a figure measured over N ranks can say
 * whether RCCL saw N ranks.  LizardGPU_commInitRank and the communicators of LizardGPU_compressBlocks_sharded are checked against
 * ncclCommCount / ncclCommUserRank when they are made (a mismatch fails the call with -LIZARDGPU_ERR_RCCL).
 *   info[0]  0 = RCCL, 1 = a table installed with LizardGPU_setCollectives
 *   info[1]  ranks the caller asked for in LizardGPU_commInitRank (0 = no rank communicator)
 *   info[2]  ranks the RCCL rank communicator reports (ncclCommCount; 0 = none, -1 = this RCCL does not export the call)
 *   info[3]  this process's rank as RCCL reports it (ncclCommUserRank; -1 = n/a)
 *   info[4]  ncclGetVersion (0 = RCCL not loaded)
 *   info[5]  ranks of the single-process communicators of LizardGPU_compressBlocks_sharded (0 = none)
 * Always returns 0. */
int  LizardGPU_commInfo(int info[6]);

/* ---- decompression ----
 * Independent blocks as Lizard_decompress_safe() decodes them (reference lib/lizard_decompress.h:64, lizard_decompress.c:267;
 * no dictionary, no prefix): every level's container, huff0 streams, fastLZ4 and LIZv1 codewords.  One wave per block.
 * _device: block i is d_srcSizes[i] bytes at d_src + i*srcStride (the layout LizardGPU_compressBlocks_device leaves behind)
 * and is decoded to d_dst + i*dstStride (capacity dstStride); d_outSizes[i] = decoded size, 0xFFFFFFFF if the block is
 * corrupt or does not fit.  Enqueued on `stream`.  _host: block i is src[offsets[i] .. offsets[i+1]) (the layout of
 * LizardGPU_compressBlocks_host_packed); synchronous.  LizardGPU_decompress_safe is the one-block twin of
 * Lizard_decompress_safe: decoded size, or a negative value for a corrupt block / maxDecompressedSize too small.
 * Memory-safe on any input: never reads outside a compressed block, never writes outside a block's output slot. */
int LizardGPU_decompressBlocks_device(const void* d_src, size_t srcStride, const uint32_t* d_srcSizes, size_t nBlocks,
                                      void* d_dst, size_t dstStride, uint32_t* d_outSizes, void* stream);
int LizardGPU_decompressBlocks_host(const void* src, const uint64_t* offsets, size_t nBlocks, void* dst, size_t dstStride,
                                    uint32_t* outSizes);
int LizardGPU_decompress_safe(const char* source, char* dest, int compressedSize, int maxDecompressedSize);


/* Kernel-only duration of the most recent LizardGPU_compressBlocks_* call on the selected device, measured with HIP
 * events on the launch stream, in milliseconds (blocks until that launch finished; host-buffer calls: the sum over
 * their chunks). < 0 if unavailable. */
float LizardGPU_lastKernelMs(void);

/* Number of resident waves (= blocks compressed concurrently) the launcher uses on this device. */
int LizardGPU_residentWaves(void);

/* Launches on different streams.  A launch needs a scratch arena (2.6 GiB), a block counter and — levels 11/31, 21/41, 22/42 —
 * per-wave tables to itself.  Launches that fill the machine, the hashChain levels and decompression share the context's own and
 * run one after the other (the second waits on the device for the first); a compress launch SMALLER than the machine that arrives
 * on another stream while that arena is busy gets one of up to three more, allocated on first need, and runs beside it on the CUs
 * it leaves free.  LIZARDGPU_ARENAS=1..4 caps the total (default 4; 1 = every launch waits for the previous one).
 * Returns how many exist right now on the selected device (>= 1 once it was used; 0 before / without a device). */
int LizardGPU_arenasInUse(void);

/* Device memory.  A context (one per device) keeps a scratch arena (one slot per resident wave: 2.7 GB on a 256-CU device), and
 * allocates on first use: up to three more arenas for small launches on concurrent streams (released again after 64 launches that
 * did not need them), per-wave tables (levels 21/41: 256 MiB; 11/31/22/42: 4 GiB), hashChain work areas (levels 12-17 / 32-38: up
 * to half of the free memory, at most 128 GiB), and the staging of the host-buffer entries (three chunks of 256 MiB in flight).
 * LizardGPU_setMemoryBudget(bytes) bounds the sum per device (0 = no bound, the default): tables and work areas then get fewer
 * slots than there are resident waves (fewer blocks in flight: slower, same bytes), what another level left behind is given up
 * first, extra arenas are only made when they fit, host-buffer chunks shrink to bytes / 64; an allocation that still does not fit
 * fails its call with -LIZARDGPU_ERR_NOMEM (Lizard_compress*: 0).  The call releases what every context holds (like
 * LizardGPU_shutdown, communicators excepted) — set it before first use or between uses, not under load; below one scratch arena
 * + 256 MiB it is refused (-LIZARDGPU_ERR_ARG, LizardGPU_lastError names the minimum).  The environment knobs stay
 * (LIZARDGPU_ARENAS, LIZARDGPU_HC_WORKAREA_MB, LIZARDGPU_CHUNK_MB).
 * LizardGPU_memoryInUse: bytes the selected device's context holds in those buffers.  LizardGPU_trim: gives back everything but the
 * context's own scratch arena on the selected device (waits for the device to be idle). */
int    LizardGPU_setMemoryBudget(size_t bytes);
size_t LizardGPU_memoryBudget(void);
size_t LizardGPU_memoryInUse(void);
int    LizardGPU_trim(void);

/* Calls of this process that could not compress on the GPU (level without a kernel, no device, HIP failure) and therefore returned
 * 0 (Lizard_compress*) or stored their blocks raw (LizardF_compress*, non-strict): every one is counted here, and the 1st, 2nd,
 * 4th, 8th ... prints one line on stderr with the reason.  The strict twins (LizardGPU_compressFrame ...) return an error instead. */
unsigned long long LizardGPU_degradedCalls(void);

/* The one-block entry points of part 1 (Lizard_compress, _extState, _continue) are COMBINED: callers that arrive while a launch is
 * in flight leave together in the next one — one ragged batch, one block per CU — instead of queueing behind a lock, so N host
 * threads compress N blocks per launch (lizard_amd/csrc/lizard_pipeline_host.c).  Launches made / blocks carried so far on the
 * selected device (either pointer may be NULL); 0 or -LIZARDGPU_ERR_*. */
int LizardGPU_combinerStats(unsigned long long* batches, unsigned long long* blocks);
/* Tuning aid: where the batches' time went, seconds since the process started — [0] waiting for the device context, [1] the
 * members' copy-in, [2] the GPU part (H2D, kernel, compaction, D2H), [3] until the last member had copied out. */
int LizardGPU_combinerProfile(double out[4]);

/* =====================================================================================================
 * Part 3 — frame production on the batched GPU path.
 *
 * A twin of the reference's frame COMPRESSOR: LizardGPU_compressBegin / _compressUpdate / _flush / _compressEnd
 * follow lib/lizard_frame.c:362-424, :501-599, :610-637, :651-677 call for call, and LizardGPU_compressFrame is built
 * on them like LizardF_compressFrame (lizard_frame.c:260-316).  They write byte for byte what the reference writes for
 * the same preferences and the same sequence of calls in independent-block mode (reference built with
 * -DLIZARD_RESET_MEM, the zero-state oracle): frame header, LE32-sized block records (bit 31 = stored raw), end
 * mark, XXH32 content checksum.  Every run of blocks an Update call covers is ONE batch through the block kernels,
 * and the block records are assembled on the device (prefix sum of sizes + compaction), so only the frame body
 * crosses PCIe, once.  The content checksum runs on a host thread beside the GPU work.
 * The reference keeps frame compression and decompression in ONE object (lizard_frame.c), so these symbols are
 * prefixed LizardGPU_ and coexist with a linked reference LizardF_* (INTEGRATION.md §2).
 *
 * LizardGPU_framePrefs_t is layout-identical to LizardF_preferences_t (lib/lizard_frame.h:111-125): a caller
 * holding the reference type passes (const LizardGPU_framePrefs_t*)&prefs.
 * Size_t results: bytes written, or an error code that LizardGPU_frameIsError() recognises; codes are the
 * reference's (size_t)-LizardF_ERROR_* values (lib/lizard_frame_static.h:57-67).  Refused, never emulated:
 * linked-block frames larger than one block (blockMode_invalid), levels without a GPU kernel
 * (compressionLevel_invalid) and skippable frames (frameType_unknown).
 * ===================================================================================================== */
typedef LizardF_frameInfo_t   LizardGPU_frameInfo_t;
typedef LizardF_preferences_t LizardGPU_framePrefs_t;

enum {                                       /* LizardF_errorCodes, lib/lizard_frame_static.h:57-67 */
    LIZARDGPU_FRAME_ERR_GENERIC = 1, LIZARDGPU_FRAME_ERR_maxBlockSize_invalid = 2, LIZARDGPU_FRAME_ERR_blockMode_invalid = 3,
    LIZARDGPU_FRAME_ERR_compressionLevel_invalid = 5, LIZARDGPU_FRAME_ERR_allocation_failed = 9,
    LIZARDGPU_FRAME_ERR_dstMaxSize_tooSmall = 11, LIZARDGPU_FRAME_ERR_frameType_unknown = 13,
    LIZARDGPU_FRAME_ERR_frameSize_wrong = 14, LIZARDGPU_FRAME_ERR_maxCode = 19
};

/* replaces LizardF_isError, lib/lizard_frame.h:59 / lizard_frame.c:179 */
unsigned LizardGPU_frameIsError(size_t code);
/* replaces LizardF_compressFrameBound, lib/lizard_frame.h:122 / lizard_frame.c:229 (same value) */
size_t LizardGPU_compressFrameBound(size_t srcSize, const LizardGPU_framePrefs_t* preferencesPtr);
/* replaces LizardF_compressFrame, lib/lizard_frame.h:134 / lizard_frame.c:260 (host buffers, synchronous) */
size_t LizardGPU_compressFrame(void* dstBuffer, size_t dstMaxSize, const void* srcBuffer, size_t srcSize,
                               const LizardGPU_framePrefs_t* preferencesPtr);

/* Streaming form.  LizardGPU_cctx_t replaces LizardF_compressionContext_t (lib/lizard_frame.h:145); the functions
 * replace LizardF_createCompressionContext / _freeCompressionContext (:157-158, int result: 0 or -LIZARDGPU_FRAME_ERR_*),
 * LizardF_compressBegin (:169), LizardF_compressBound (:178), LizardF_compressUpdate (:190), LizardF_flush (:202) and
 * LizardF_compressEnd (:213).  The reference's LizardF_compressOptions_t only carries stableSrc, which matters to
 * linked blocks alone, so the argument is dropped. */
typedef struct LizardF_cctx_s LizardGPU_cctx_t;      /* the object LizardF_createCompressionContext makes, in its strict mode */
int    LizardGPU_createCompressionContext(LizardGPU_cctx_t** cctxPtr);
int    LizardGPU_freeCompressionContext(LizardGPU_cctx_t* cctx);
size_t LizardGPU_compressBegin(LizardGPU_cctx_t* cctx, void* dstBuffer, size_t dstMaxSize, const LizardGPU_framePrefs_t* preferencesPtr);
size_t LizardGPU_compressBound(size_t srcSize, const LizardGPU_framePrefs_t* preferencesPtr);
size_t LizardGPU_compressUpdate(LizardGPU_cctx_t* cctx, void* dstBuffer, size_t dstMaxSize, const void* srcBuffer, size_t srcSize);
size_t LizardGPU_flush(LizardGPU_cctx_t* cctx, void* dstBuffer, size_t dstMaxSize);
size_t LizardGPU_compressEnd(LizardGPU_cctx_t* cctx, void* dstBuffer, size_t dstMaxSize);

#ifdef __cplusplus
}
#endif
#endif /* LIZARD_AMD_H */
