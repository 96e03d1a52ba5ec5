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

#define LIZARD_VERSION_NUMBER (1 * 100 * 100 + 0 * 100 + 0)   /* reference lib/lizard_compress.h:71-75 */

struct Lizard_stream_s { void* reserved; int compressionLevel; };   /* opaque to callers (pointer-aligned); tables live in LDS */

static int verify_level(int level)                               /* reference lib/lizard_compress.c:303-308 */
{
    if (level > LIZARD_MAX_CLEVEL) level = LIZARD_MAX_CLEVEL;
    if (level < LIZARD_MIN_CLEVEL) level = LIZARD_DEFAULT_CLEVEL;
    return level;
}

static void complain(const char* what, int level)
{
    static int warned = 0;
    if (!warned) {
        warned = 1;
        fprintf(stderr, "liblizard_amd: %s (level %d): %s — returning 0 (no CPU fallback in this library)\n",
                what, level, LizardGPU_lastError());
    }
}

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
    if (!LizardGPU_levelSupported(level)) { complain("level not implemented on the GPU path", level); return 0; }
    r = lzgpu_compress_one(src, srcSize, dst, maxDstSize, level);
    if (r < 0) { complain("GPU compression failed", level); return 0; }
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
    if (s) s->compressionLevel = verify_level(compressionLevel);
    return s;
}

int Lizard_freeStream(Lizard_stream_t* s) { free(s); return 0; }

Lizard_stream_t* Lizard_resetStream(Lizard_stream_t* s, int compressionLevel)
{
    if (s) s->compressionLevel = verify_level(compressionLevel);
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

/* Linked blocks and dictionaries (reference lib/lizard_compress.h:178,188,198 / lib/lizard_compress.c:426,454,550): every
 * block's window reaches into the previous one, a serial chain that has no place on this path.  The symbols
 * exist so that programs written against the reference (lizard_frame.c's linked mode, lizardio.c) link
 * unchanged; they fail the way the reference reports failure — 0 — after saying why, once, on stderr.  A frame
 * written in linked mode through them stores its blocks raw (lizard_frame.c:463-467). */
static void refuse_linked(const char* what)
{
    static int warned = 0;
    if (!warned) {
        warned = 1;
        fprintf(stderr, "liblizard_amd: %s: linked blocks / dictionaries are serial and not part of the GPU path "
                        "(use independent blocks, e.g. lizard -BI) — returning 0\n", what);
    }
}
int Lizard_loadDict(Lizard_stream_t* streamPtr, const char* dictionary, int dictSize)
{ (void)streamPtr; (void)dictionary; (void)dictSize; refuse_linked("Lizard_loadDict"); return 0; }
int Lizard_saveDict(Lizard_stream_t* streamPtr, char* safeBuffer, int dictSize)
{ (void)streamPtr; (void)safeBuffer; (void)dictSize; refuse_linked("Lizard_saveDict"); return 0; }
int Lizard_compress_continue(Lizard_stream_t* streamPtr, const char* src, char* dst, int srcSize, int maxDstSize)
{ (void)streamPtr; (void)src; (void)dst; (void)srcSize; (void)maxDstSize; refuse_linked("Lizard_compress_continue"); return 0; }
