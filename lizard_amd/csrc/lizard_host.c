/*
 * lizard_host.c — host C layer: the reference's one-block compression ABI (lib/lizard_compress.h) on
 * top of the GPU shim.  There is NO CPU compressor in this library: a level the GPU path does not
 * implement, a missing device or a HIP failure makes these functions return 0 ("compression failed",
 * reference lib/lizard_compress.h:113-114) after printing the reason once on stderr — loud, never a
 * silent fallback.  Callers that need throughput use the batch entry points of include/lizard_amd.h;
 * one block per call pays a full H2D + launch + D2H round trip.
 */
#include "../../include/lizard_amd.h"
#include "lizard_gpu_shim.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LIZARD_VERSION_NUMBER (1 * 100 * 100 + 0 * 100 + 0)   /* reference lib/lizard_compress.h:71-75 */

/* Opaque to callers (pointer-aligned).  The match-finder tables live on the device; what a stream remembers on the host
 * is the book-keeping Lizard_saveDict needs: where the previous block ended and how much contiguous history precedes it. */
struct Lizard_stream_s { const char* end; size_t prefix; int compressionLevel; };

static int verify_level(int level)                               /* reference lib/lizard_compress.c:303-308 */
{
    if (level > LIZARD_MAX_CLEVEL) level = LIZARD_MAX_CLEVEL;
    if (level < LIZARD_MIN_CLEVEL) level = LIZARD_DEFAULT_CLEVEL;
    return level;
}

/* A call that could not compress (level without a kernel, no device, HIP failure) returns 0 — "compression failed", as the
 * reference's contract says (lib/lizard_compress.h:113) — and the frame layer then stores its blocks raw.  That is silent towards
 * the caller, so it is loud on stderr: the 1st, 2nd, 4th, 8th ... such call of the process prints a line with the reason (a
 * long-running service keeps hearing about it without being flooded), and LizardGPU_degradedCalls() counts every one of them. */
static unsigned long long g_degraded;
void lzgpu_note_degraded(const char* what, int level)
{
    const unsigned long long n = __atomic_add_fetch(&g_degraded, 1ull, __ATOMIC_RELAXED);
    if ((n & (n - 1ull)) == 0)
        fprintf(stderr, "liblizard_amd: %s (level %d): %s — no CPU fallback in this library [occurrence %llu in this process, see LizardGPU_degradedCalls()]\n",
                what, level, LizardGPU_lastError()[0] ? LizardGPU_lastError() : "level not implemented on the GPU path", n);
}
unsigned long long LizardGPU_degradedCalls(void) { return __atomic_load_n(&g_degraded, __ATOMIC_RELAXED); }
static void complain(const char* what, int level) { lzgpu_note_degraded(what, level); }

int Lizard_versionNumber(void) { return LIZARD_VERSION_NUMBER; }
int Lizard_compressBound(int isize) { return LIZARD_COMPRESSBOUND(isize); }

int Lizard_sizeofState(int compressionLevel)
{
    (void)compressionLevel;
    return (int)sizeof(struct Lizard_stream_s) + 64;
}

int Lizard_compress_extState(void* state, const char* src, char* dst, int srcSize, int maxDstSize, int compressionLevel)
{
    int level = verify_level(compressionLevel), r;
    if (((size_t)state & (sizeof(void*) - 1)) != 0) return 0;   /* reference lib/lizard_compress.c:586 */
    if (!LizardGPU_levelSupported(level)) { complain("returning 0: level not implemented on the GPU path", level); return 0; }
    r = lzgpu_compress_one(src, srcSize, dst, maxDstSize, level);
    if (r < 0) { complain("returning 0: GPU compression failed", level); return 0; }
    return r;
}

int Lizard_compress(const char* src, char* dst, int srcSize, int maxDstSize, int compressionLevel)
{
    struct Lizard_stream_s st;                                   /* reference mallocs a state here (:599) */
    return Lizard_compress_extState(&st, src, dst, srcSize, maxDstSize, compressionLevel);
}

Lizard_stream_t* Lizard_createStream(int compressionLevel)
{
    Lizard_stream_t* s = (Lizard_stream_t*)malloc((size_t)Lizard_sizeofState(compressionLevel));
    if (s) { s->compressionLevel = verify_level(compressionLevel); s->end = NULL; s->prefix = 0; }
    return s;
}

int Lizard_freeStream(Lizard_stream_t* s) { free(s); return 0; }

Lizard_stream_t* Lizard_resetStream(Lizard_stream_t* s, int compressionLevel)
{
    if (s) { s->compressionLevel = verify_level(compressionLevel); s->end = NULL; s->prefix = 0; }
    return s;
}

/* reference lib/lizard_compress.c:68,612-630 */
int Lizard_sizeofState_MinLevel(void) { return Lizard_sizeofState(LIZARD_MIN_CLEVEL); }
int Lizard_compress_extState_MinLevel(void* state, const char* source, char* dest, int inputSize, int maxOutputSize)
{ return Lizard_compress_extState(state, source, dest, inputSize, maxOutputSize, LIZARD_MIN_CLEVEL); }
int Lizard_compress_MinLevel(const char* source, char* dest, int inputSize, int maxOutputSize)
{ return Lizard_compress(source, dest, inputSize, maxOutputSize, LIZARD_MIN_CLEVEL); }
Lizard_stream_t* Lizard_createStream_MinLevel(void) { return Lizard_createStream(LIZARD_MIN_CLEVEL); }
Lizard_stream_t* Lizard_resetStream_MinLevel(Lizard_stream_t* s) { return Lizard_resetStream(s, LIZARD_MIN_CLEVEL); }

/* Streaming ("linked blocks") and dictionaries — reference lib/lizard_compress.h:178,188,198 / lib/lizard_compress.c:426,454,550.
 *
 * In the reference every block of a stream may reach back into the previous blocks (or the dictionary), which makes a
 * stream ONE serial chain: the hash table, the window and the repeat offset are carried from call to call.  That chain
 * has no parallelism to offer a GPU, so this library keeps the contract and drops the history: each call compresses
 * its block on the GPU WITHOUT referring to earlier data.  What callers rely on still holds —
 *   * the output of every call is a valid Lizard block that Lizard_decompress_safe_continue / _usingDict decode to
 *     the input (a decoder never requires a block to use its history), so frames written in the frame layer's default
 *     linked mode (lizard_frame.c:473-483) are compressed, valid and decodable;
 *   * Lizard_loadDict / Lizard_saveDict return the sizes the reference returns, and saveDict copies the last bytes
 *     of the previous block into the caller's buffer (lizard_compress.c:454-470), so lizard_frame.c's buffer
 *     management behaves as with the reference —
 * and what does not: the bytes differ from the reference's linked-mode output (matches into earlier blocks are not
 * found; the ratio is that of independent blocks).  Bit-exact parity is claimed for independent blocks only
 * (INTEGRATION.md section 1). */
#define LZ_DICT_SIZE (1 << 24)                                   /* LIZARD_DICT_SIZE, reference lib/lizard_common.h:71 */

int Lizard_loadDict(Lizard_stream_t* streamPtr, const char* dictionary, int dictSize)
{
    if (!streamPtr || dictSize < 0) return 0;
    if (dictSize > LZ_DICT_SIZE) { dictionary += dictSize - LZ_DICT_SIZE; dictSize = LZ_DICT_SIZE; }   /* :429-432 */
    streamPtr->end = dictionary + dictSize;                      /* :435 */
    streamPtr->prefix = (size_t)dictSize;
    return dictSize;
}

int Lizard_saveDict(Lizard_stream_t* streamPtr, char* safeBuffer, int dictSize)
{
    if (!streamPtr || !streamPtr->end) return 0;
    if (dictSize > LZ_DICT_SIZE) dictSize = LZ_DICT_SIZE;        /* :458-460 */
    if (dictSize < 4) dictSize = 0;
    if ((size_t)dictSize > streamPtr->prefix) dictSize = (int)streamPtr->prefix;
    memmove(safeBuffer, streamPtr->end - dictSize, (size_t)dictSize);   /* :461 */
    streamPtr->end = safeBuffer + dictSize;
    streamPtr->prefix = (size_t)dictSize;
    return dictSize;
}

int Lizard_compress_continue(Lizard_stream_t* streamPtr, const char* src, char* dst, int srcSize, int maxDstSize)
{
    static int noted = 0;
    int r;
    if (!streamPtr) return 0;
    if (!__atomic_exchange_n(&noted, 1, __ATOMIC_RELAXED) && getenv("LIZARDGPU_VERBOSE"))      /* said once, for those who ask */
        fprintf(stderr, "liblizard_amd: Lizard_compress_continue compresses every block WITHOUT its history (valid, decodable linked-mode "
                        "blocks; not the reference's linked-mode bytes — include/lizard_amd.h)\n");
    if (srcSize > 0) {   /* the reference moves `end` before it compresses (lizard_compress.c:491), so a failed call counts too;
                            :560-561: contiguous input extends the prefix, anything else starts a new one */
        streamPtr->prefix = (src == streamPtr->end ? streamPtr->prefix : 0) + (size_t)srcSize;
        streamPtr->end = src + srcSize;
    }
    if (!LizardGPU_levelSupported(streamPtr->compressionLevel)) { complain("level not implemented on the GPU path", streamPtr->compressionLevel); return 0; }
    r = lzgpu_compress_one(src, srcSize, dst, maxDstSize, streamPtr->compressionLevel);
    if (r < 0) { complain("GPU compression failed", streamPtr->compressionLevel); return 0; }
    return r;
}
