/* lizard_pipeline_host.c — the host-buffer side of the library, in C: the pipelined batch entries
 * (LizardGPU_compressBlocks_host / _host_packed, the frame layer's record producer), the one-block shim behind the reference's
 * Lizard_compress (lizard_host.c) and the host forms of decompression.  It talks to the GPU through the HIP runtime's C API
 * (copies, events, streams) and through the thin shim of lizard_gpu_ctx.h (context lock, kernel launches): no kernels here.
 *
 * The host-buffer pipeline.  Input is cut into chunks of whole blocks.  Per chunk, on its stage's stream: host -> pinned staging
 * (skipped when the caller's buffer is itself pinned) -> H2D -> block kernels -> exclusive scan of the record sizes -> compaction
 * of the valid bytes into one packed buffer -> D2H of sizes / offsets, then of exactly the packed bytes.  Two threads per job: the
 * calling thread stages and issues chunk after chunk; a drain thread follows it chunk by chunk — waits for the sizes, requests
 * exactly the packed bytes, waits for them and hands them to the sink.  With LZ_STAGES chunks in flight neither side waits for
 * the other's host copies, and the two PCIe directions run side by side. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/lizard_amd.h"
#include "lizard_gpu_ctx.h"
#include "lizard_gpu_shim.h"

#define LZ_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            snprintf(lzk_err(), LZK_ERR_BYTES, "%s failed: %s", #call, hipGetErrorString(e_));         \
            return e_ == hipErrorOutOfMemory ? -LIZARDGPU_ERR_NOMEM : -LIZARDGPU_ERR_HIP;              \
        }                                                                                              \
    } while (0)

static int ensure_dev(LzCtx* c, void** p, size_t* cap, size_t need)
{
    int rc;
    if (*cap >= need) return 0;
    if (*p) { lzk_dev_free(c, *p, *cap); *p = NULL; *cap = 0; }
    if ((rc = lzk_dev_alloc(c, p, need))) return rc;        /* counted against the memory budget */
    *cap = need;
    return 0;
}
static int ensure_pinned(void** p, size_t* cap, size_t need)
{
    if (*cap >= need) return 0;
    if (*p) { LZ_HIP(hipHostFree(*p)); *p = NULL; *cap = 0; }
    LZ_HIP(hipHostMalloc(p, need, hipHostMallocDefault));
    *cap = need;
    return 0;
}

/* Host copies between caller memory and the pinned staging buffers are what bounds the PCIe-inclusive rate (one core moves
 * ~10 GB/s): large copies are cut into slices for a few short-lived threads. */
#ifndef LZ_COPY_THREADS
#define LZ_COPY_THREADS 4
#endif
typedef struct { void* d; const void* s; size_t n; } CopyJob;
static void* copy_thread(void* a) { CopyJob* j = (CopyJob*)a; memcpy(j->d, j->s, j->n); return NULL; }
static void par_memcpy(void* dst, const void* src, size_t n)
{
    pthread_t th[LZ_COPY_THREADS]; CopyJob job[LZ_COPY_THREADS]; int started[LZ_COPY_THREADS];
    size_t slice;
    int i;
    if (n < ((size_t)16 << 20)) { memcpy(dst, src, n); return; }
    slice = ((n / LZ_COPY_THREADS) + 4095) & ~(size_t)4095;
    for (i = 0; i < LZ_COPY_THREADS; i++) {
        const size_t off = (size_t)i * slice;
        job[i].d = (uint8_t*)dst + off; job[i].s = (const uint8_t*)src + off; job[i].n = off >= n ? 0 : (n - off < slice ? n - off : slice);
        started[i] = i > 0 && job[i].n && pthread_create(&th[i], NULL, copy_thread, &job[i]) == 0;
    }
    for (i = 0; i < LZ_COPY_THREADS; i++) if (!started[i] && job[i].n) memcpy(job[i].d, job[i].s, job[i].n);   /* slice 0, and any slice whose thread did not start */
    for (i = 0; i < LZ_COPY_THREADS; i++) if (started[i]) pthread_join(th[i], NULL);
}

/* sizes / offsets of a stage, device and pinned host, for `n` entries.  A failure leaves nothing half-allocated behind. */
static void free_meta(LzStage* s)
{
    if (s->d_sizes) (void)hipFree(s->d_sizes);
    if (s->d_offsets) (void)hipFree(s->d_offsets);
    if (s->h_sizes) (void)hipHostFree(s->h_sizes);
    if (s->h_offsets) (void)hipHostFree(s->h_offsets);
    s->d_sizes = NULL; s->d_offsets = NULL; s->h_sizes = NULL; s->h_offsets = NULL;
    s->d_meta_cap = 0; s->h_meta_cap = 0;
}
static int ensure_meta(LzStage* s, size_t n)
{
    hipError_t e;
    if (s->d_meta_cap >= n && s->h_meta_cap >= n) return 0;
    free_meta(s);
    if ((e = hipMalloc((void**)&s->d_sizes, n * sizeof(uint32_t))) == hipSuccess
        && (e = hipMalloc((void**)&s->d_offsets, n * sizeof(uint64_t))) == hipSuccess
        && (e = hipHostMalloc((void**)&s->h_sizes, n * sizeof(uint32_t), hipHostMallocDefault)) == hipSuccess
        && (e = hipHostMalloc((void**)&s->h_offsets, n * sizeof(uint64_t), hipHostMallocDefault)) == hipSuccess) {
        s->d_meta_cap = n; s->h_meta_cap = n;
        return 0;
    }
    snprintf(lzk_err(), LZK_ERR_BYTES, "allocation of %zu size / offset entries failed: %s", n, hipGetErrorString(e));
    free_meta(s);
    return e == hipErrorOutOfMemory ? -LIZARDGPU_ERR_NOMEM : -LIZARDGPU_ERR_HIP;
}

static int is_pinned_host(const void* p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return at.type == hipMemoryTypeHost;
}

static size_t g_chunk_bytes = 0;                           /* host pipeline chunk (input bytes), 0 = not read yet */
static size_t chunk_bytes(const LzCtx* ctx)
{
    if (!g_chunk_bytes) {
        const char* e = getenv("LIZARDGPU_CHUNK_MB");
        size_t mb = e ? (size_t)strtoul(e, NULL, 10) : 0;
        if (mb < 1 || mb > 65536) mb = 256;                  /* measured on a 4 GiB job: 256 MiB 37 GB/s, 512 MiB 26, 1 GiB 23 (fill and drain of the pipeline) */
        g_chunk_bytes = mb << 20;
    }
    {   /* under a memory budget (LizardGPU_setMemoryBudget) the three stages' device buffers — input, slots and packed output each,
         * about 3 x 3 chunks — must fit what the budget leaves beside the scratch arena: a sixteenth of that room per chunk */
        const size_t room = lzk_budget_room_for_staging(ctx);
        if (room != (size_t)-1) {
            size_t c = (room / 16) & ~(((size_t)1 << 20) - 1);
            if (c < ((size_t)4 << 20)) c = (size_t)4 << 20;
            if (c < g_chunk_bytes) return c;
        }
    }
    return g_chunk_bytes;
}

typedef struct {
    const uint8_t* src; size_t nBlocks, blockSize, lastBlockSize; int level;
    int mode;                                  /* LZK_PACK_PAYLOAD or LZK_PACK_FRAME */
    /* sink: chunk [first, first+nb) finished; packed bytes at `data` (size `bytes`), per-block offsets inside it
     * (offsets[nb] = bytes) and compressed sizes.  Returns 0 or a negative error. */
    int (*sink)(void* user, size_t first, size_t nb, const uint8_t* data, size_t bytes, const uint64_t* offsets, const uint32_t* sizes);
    void* user;
} HostJob;

typedef struct { size_t first, nb, inBytes, packedBytes; int active; } ChunkState;

static int stage_issue(LzCtx* c, LzStage* s, const HostJob* j, ChunkState* ch, int srcPinned, hipEvent_t prevUp)
{
    const size_t slot = ((size_t)LIZARD_COMPRESSBOUND((int)j->blockSize) + 63) & ~(size_t)63;
    const size_t last = (ch->first + ch->nb == j->nBlocks) ? j->lastBlockSize : j->blockSize;
    const size_t packedCap = ch->nb * (slot + 8);
    const uint8_t* from;
    int rc;
    ch->inBytes = (ch->nb - 1) * j->blockSize + last;
    if ((rc = ensure_dev(c, (void**)&s->d_in, &s->d_in_cap, ch->inBytes + 64))) return rc;
    if ((rc = ensure_dev(c, (void**)&s->d_slots, &s->d_slots_cap, ch->nb * slot))) return rc;
    if ((rc = ensure_dev(c, (void**)&s->d_packed, &s->d_packed_cap, packedCap))) return rc;
    if ((rc = ensure_pinned((void**)&s->h_out, &s->h_out_cap, packedCap + 64))) return rc;    /* worst case once: a buffer that follows the chunks' sizes is re-pinned again and again */
    if ((rc = ensure_meta(s, ch->nb + 1))) return rc;
    from = j->src + ch->first * j->blockSize;
    if (!srcPinned) {
        if ((rc = ensure_pinned((void**)&s->h_in, &s->h_in_cap, ch->inBytes))) return rc;
        par_memcpy(s->h_in, from, ch->inBytes);
        from = s->h_in;
    }
    /* uploads run one after the other (an upload that shares the link with the next chunk's finishes late, and its kernels with it) */
    if (prevUp) LZ_HIP(hipStreamWaitEvent(s->stream, prevUp, 0));
    LZ_HIP(hipMemcpyAsync(s->d_in, from, ch->inBytes, hipMemcpyHostToDevice, s->stream));
    LZ_HIP(hipEventRecord(s->up, s->stream));
    if ((rc = lzk_launch(c, s->d_in, ch->nb, j->blockSize, last, s->d_slots, slot, s->d_sizes, j->level, s->stream, s->k0, s->k1, NULL, NULL))) return rc;
    lzk_pack_launch(s->d_in, s->d_slots, slot, s->d_sizes, s->d_offsets, s->d_packed, (uint32_t)ch->nb, (uint32_t)j->blockSize, (uint32_t)last, j->mode, s->stream);
    LZ_HIP(hipGetLastError());
    LZ_HIP(hipMemcpyAsync(s->h_sizes, s->d_sizes, ch->nb * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    LZ_HIP(hipMemcpyAsync(s->h_offsets, s->d_offsets, (ch->nb + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream));
    LZ_HIP(hipEventRecord(s->meta, s->stream));
    ch->active = 1;
    return 0;
}

static int stage_fetch(LzStage* s, ChunkState* ch)              /* sizes known -> request exactly the packed bytes */
{
    int rc;
    LZ_HIP(hipEventSynchronize(s->meta));
    ch->packedBytes = (size_t)s->h_offsets[ch->nb];
    if ((rc = ensure_pinned((void**)&s->h_out, &s->h_out_cap, ch->packedBytes + 64))) return rc;
    LZ_HIP(hipMemcpyAsync(s->h_out, s->d_packed, ch->packedBytes, hipMemcpyDeviceToHost, s->stream));
    LZ_HIP(hipEventRecord(s->done, s->stream));
    return 0;
}

typedef struct {
    LzCtx* c; const HostJob* j;
    size_t nChunks, perChunk;
    int srcPinned;
    ChunkState ch[LZ_STAGES];
    pthread_mutex_t mu;
    pthread_cond_t cv;
    size_t issued, drained;                    /* chunks issued by the caller / handed to the sink by the drain thread */
    int err;                                   /* first error of either side */
    char errText[LZK_ERR_BYTES];
    float kernelMs;
} Pipe;

static void pipe_fail(Pipe* p, int rc)
{
    pthread_mutex_lock(&p->mu);
    if (!p->err) { p->err = rc; memcpy(p->errText, lzk_err(), LZK_ERR_BYTES); }
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->mu);
}
static int drain_chunk(Pipe* p, size_t i)
{
    LzStage* s = &p->c->stage[i % LZ_STAGES];
    ChunkState* ch = &p->ch[i % LZ_STAGES];
    float ms = 0.0f;
    int rc;
    if ((rc = stage_fetch(s, ch))) return rc;
    LZ_HIP(hipEventSynchronize(s->done));
    if (hipEventElapsedTime(&ms, s->k0, s->k1) == hipSuccess) p->kernelMs += ms;
    ch->active = 0;
    return p->j->sink(p->j->user, ch->first, ch->nb, s->h_out, ch->packedBytes, s->h_offsets, s->h_sizes);
}
static void* drain_thread(void* a)
{
    Pipe* p = (Pipe*)a;
    size_t i;
    lzk_err()[0] = 0;
    if (hipSetDevice(p->c->device) != hipSuccess) { snprintf(lzk_err(), LZK_ERR_BYTES, "hipSetDevice(%d) failed", p->c->device); pipe_fail(p, -LIZARDGPU_ERR_HIP); return NULL; }
    for (i = 0; i < p->nChunks; i++) {
        int stop, rc;
        pthread_mutex_lock(&p->mu);
        while (p->issued <= i && !p->err) pthread_cond_wait(&p->cv, &p->mu);
        stop = p->err != 0;
        pthread_mutex_unlock(&p->mu);
        if (stop) return NULL;
        rc = drain_chunk(p, i);
        if (rc) { pipe_fail(p, rc); return NULL; }
        pthread_mutex_lock(&p->mu);
        p->drained = i + 1;
        pthread_cond_broadcast(&p->cv);
        pthread_mutex_unlock(&p->mu);
    }
    return NULL;
}
static int run_host_job_inner(LzCtx* c, const HostJob* j)
{
    Pipe p;
    pthread_t th;
    int threaded, rc = lzk_ctx_init(c);
    size_t i;
    if (rc) return rc;
    if (!j->src || j->nBlocks == 0 || j->blockSize == 0 || j->lastBlockSize == 0 || j->lastBlockSize > j->blockSize) {
        snprintf(lzk_err(), LZK_ERR_BYTES, "bad argument (null pointer, zero size or lastBlockSize > blockSize)");
        return -LIZARDGPU_ERR_ARG;
    }
    /* what the launcher would refuse is refused HERE, before anything is allocated, pinned or copied */
    if (!LizardGPU_levelSupported(j->level)) { snprintf(lzk_err(), LZK_ERR_BYTES, "level %d has no GPU kernel", lzk_clamp_level(j->level)); return -LIZARDGPU_ERR_LEVEL; }
    if (j->blockSize > (size_t)LIZARD_MAX_INPUT_SIZE || j->nBlocks > 0xFFFFFFFFu) {
        snprintf(lzk_err(), LZK_ERR_BYTES, "bad argument (blockSize above LIZARD_MAX_INPUT_SIZE or more than 2^32 - 1 blocks)");
        return -LIZARDGPU_ERR_ARG;
    }
    memset(&p, 0, sizeof p);
    pthread_mutex_init(&p.mu, NULL); pthread_cond_init(&p.cv, NULL);
    p.c = c; p.j = j;
    p.perChunk = chunk_bytes(c) / j->blockSize;
    if (p.perChunk == 0) p.perChunk = 1;
    p.nChunks = (j->nBlocks + p.perChunk - 1) / p.perChunk;
    p.srcPinned = is_pinned_host(j->src);
    c->hostKernelMs = 0.0f;
    threaded = p.nChunks > 1 && pthread_create(&th, NULL, drain_thread, &p) == 0;
    for (i = 0; i < p.nChunks; i++) {
        ChunkState* cur;
        if (threaded) {                                          /* the stage of chunk i is free once chunk i - LZ_STAGES is drained */
            int stop;
            pthread_mutex_lock(&p.mu);
            while (i >= p.drained + LZ_STAGES && !p.err) pthread_cond_wait(&p.cv, &p.mu);
            stop = p.err != 0;
            pthread_mutex_unlock(&p.mu);
            if (stop) break;
        }
        cur = &p.ch[i % LZ_STAGES];
        cur->first = i * p.perChunk;
        cur->nb = j->nBlocks - cur->first < p.perChunk ? j->nBlocks - cur->first : p.perChunk;
        if ((rc = stage_issue(c, &c->stage[i % LZ_STAGES], j, cur, p.srcPinned, i ? c->stage[(i - 1) % LZ_STAGES].up : NULL))) { pipe_fail(&p, rc); break; }
        if (threaded) {
            pthread_mutex_lock(&p.mu);
            p.issued = i + 1;
            pthread_cond_broadcast(&p.cv);
            pthread_mutex_unlock(&p.mu);
        } else if ((rc = drain_chunk(&p, i))) { pipe_fail(&p, rc); break; }
    }
    if (threaded) pthread_join(th, NULL);
    c->hostKernelMs = p.kernelMs;
    rc = p.err;
    if (rc) memcpy(lzk_err(), p.errText, LZK_ERR_BYTES);
    pthread_mutex_destroy(&p.mu); pthread_cond_destroy(&p.cv);
    return rc;
}
static int run_host_job(LzCtx* c, const HostJob* j)
{
    const int rc = run_host_job_inner(c, j);
    if (rc) {                                                   /* a failed chunk may leave copies of the other stages in flight: drain them */
        char keep[LZK_ERR_BYTES];
        int i;
        memcpy(keep, lzk_err(), sizeof keep);
        for (i = 0; i < LZ_STAGES; i++) if (c->stage[i].stream) (void)hipStreamSynchronize(c->stage[i].stream);
        (void)hipGetLastError();
        memcpy(lzk_err(), keep, sizeof keep);
    }
    return rc;
}

typedef struct { uint8_t* dst; size_t dstStride; uint32_t* cSizes; } SlotSink;
static int slot_sink(void* user, size_t first, size_t nb, const uint8_t* data, size_t bytes, const uint64_t* offsets, const uint32_t* sizes)
{
    SlotSink* k = (SlotSink*)user;
    size_t i;
    (void)bytes;
    for (i = 0; i < nb; i++) {
        memcpy(k->dst + (first + i) * k->dstStride, data + offsets[i], sizes[i]);
        k->cSizes[first + i] = sizes[i];
    }
    return 0;
}

typedef struct { uint8_t* dst; size_t cap; size_t used; uint64_t* offsets; uint32_t* cSizes; } PackedSink;
static int packed_sink(void* user, size_t first, size_t nb, const uint8_t* data, size_t bytes, const uint64_t* offsets, const uint32_t* sizes)
{
    PackedSink* k = (PackedSink*)user;
    size_t i;
    if (k->used + bytes > k->cap) { snprintf(lzk_err(), LZK_ERR_BYTES, "packed output does not fit in dstCapacity"); return -LIZARDGPU_ERR_ARG; }
    par_memcpy(k->dst + k->used, data, bytes);
    for (i = 0; i < nb; i++) {
        if (k->offsets) k->offsets[first + i] = k->used + offsets[i];
        if (k->cSizes) k->cSizes[first + i] = sizes[i];
    }
    k->used += bytes;
    return 0;
}

/* ---- entry points (include/lizard_amd.h, lizard_gpu_shim.h) ---- */

int LizardGPU_compressBlocks_host(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize,
                                  void* dst, size_t dstStride, uint32_t* cSizes, int level)
{
    LzGuard g;
    int rc;
    lzk_guard_acquire(&g);
    if (g.rc) return g.rc;
    if (!dst || !cSizes || blockSize > LIZARD_MAX_INPUT_SIZE || dstStride < (size_t)LIZARD_COMPRESSBOUND((int)blockSize)) {
        snprintf(lzk_err(), LZK_ERR_BYTES, "bad argument (null pointer or dstStride < Lizard_compressBound(blockSize))");
        rc = -LIZARDGPU_ERR_ARG;
    } else {
        SlotSink k; HostJob j;
        k.dst = (uint8_t*)dst; k.dstStride = dstStride; k.cSizes = cSizes;
        j.src = (const uint8_t*)src; j.nBlocks = nBlocks; j.blockSize = blockSize; j.lastBlockSize = lastBlockSize; j.level = level;
        j.mode = LZK_PACK_PAYLOAD; j.sink = slot_sink; j.user = &k;
        rc = run_host_job(g.c, &j);
    }
    lzk_guard_release(&g);
    return rc;
}

int LizardGPU_compressBlocks_host_packed(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize,
                                         void* dst, size_t dstCapacity, uint64_t* offsets, uint32_t* cSizes, int level)
{
    LzGuard g;
    int rc;
    lzk_guard_acquire(&g);
    if (g.rc) return g.rc;
    if (!dst) { snprintf(lzk_err(), LZK_ERR_BYTES, "bad argument (null dst)"); rc = -LIZARDGPU_ERR_ARG; }
    else {
        PackedSink k; HostJob j;
        k.dst = (uint8_t*)dst; k.cap = dstCapacity; k.used = 0; k.offsets = offsets; k.cSizes = cSizes;
        j.src = (const uint8_t*)src; j.nBlocks = nBlocks; j.blockSize = blockSize; j.lastBlockSize = lastBlockSize; j.level = level;
        j.mode = LZK_PACK_PAYLOAD; j.sink = packed_sink; j.user = &k;
        rc = run_host_job(g.c, &j);
        if (!rc && offsets) offsets[nBlocks] = k.used;
    }
    lzk_guard_release(&g);
    return rc;
}

/* Internal (lizard_frame_host.c): frame block records — LE32 size word (bit 31 = stored raw) + payload — of nBlocks
 * independent blocks, packed back to back into dst exactly as LizardF_compressUpdate writes them
 * (reference lib/lizard_frame.c:456-469).  *written receives the byte count. */
int lzgpu_frame_records(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize, void* dst, size_t dstCapacity,
                        size_t* written, int level)
{
    LzGuard g;
    PackedSink k; HostJob j;
    int rc;
    lzk_guard_acquire(&g);
    if (g.rc) return g.rc;
    k.dst = (uint8_t*)dst; k.cap = dstCapacity; k.used = 0; k.offsets = NULL; k.cSizes = NULL;
    j.src = (const uint8_t*)src; j.nBlocks = nBlocks; j.blockSize = blockSize; j.lastBlockSize = lastBlockSize; j.level = level;
    j.mode = LZK_PACK_FRAME; j.sink = packed_sink; j.user = &k;
    rc = run_host_job(g.c, &j);
    if (written) *written = k.used;
    lzk_guard_release(&g);
    return rc;
}

/* ---- the one-block entry points (lizard_host.c: Lizard_compress, _extState, _continue) ----
 *
 * One block is one wavefront: ~3 ms for 256 KiB whatever else the chip is doing, so a caller that compresses block after block
 * gets ~80 MB/s, and N threads doing so behind a lock would share those 80 MB/s.  The COMBINER makes their calls leave together:
 * a caller that finds no batch under way becomes the LEADER of one — it takes every caller of its level that is waiting (itself
 * included), lays their inputs out back to back in pinned staging, lets every member copy its own input in (the host copies of a
 * batch run on the members' own threads, side by side), launches ONE ragged batch (per-block sizes and offsets, LzBatch::srcSizes /
 * ::srcOffsets; blocks spread one per CU, LzBatch::activeWaves), packs the valid bytes on the device, fetches them with one D2H,
 * and every member copies its own output out and returns.  Callers that arrive meanwhile queue up for the next batch; the next
 * leader is whichever of them wakes first.  N concurrent callers get N blocks per launch instead of one: the aggregate rate is
 * N x the one-block rate until the PCIe copies matter.
 *
 * Each block is compressed exactly as by LizardGPU_compressBlocks_* (same kernels, zero-state semantics), and the reference's
 * maxDstSize contract is applied per member: compressed size, 0 if it does not fit (reference lib/lizard_compress.c:543-546),
 * < 0 on a GPU failure. */
typedef struct LzOneJob {
    const void* src; int srcSize; void* dst; int maxDst; int level;
    int state;                                  /* JOB_* */
    int result;
    size_t inOff;                               /* my input's place in the batch's staging */
    const uint8_t* out; uint32_t csize;         /* my compressed block in the batch's pinned output */
    char errText[LZK_ERR_BYTES];
    struct LzOneJob* next;
} LzOneJob;
enum { JOB_QUEUED = 0, JOB_COPY_IN, JOB_COPIED_IN, JOB_RESULT, JOB_FAILED };
#ifndef LZ_ONE_MAX_JOBS
#define LZ_ONE_MAX_JOBS   1024                  /* members per batch (tests build a library with 3: lizard_amd/csrc/Makefile combiner-test) */
#endif
#define LZ_ONE_MAX_BYTES  ((size_t)1 << 30)     /* input bytes per batch */
#ifndef LZ_ONE_WINDOW_US
#define LZ_ONE_WINDOW_US  150L                  /* longest a leader waits for the stragglers of the previous batch */
#endif

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + (double)t.tv_nsec * 1e-9; }

/* the GPU part of a batch: inputs are in comb.st.h_in at job->inOff; on success every job has out / csize */
static int batch_on_gpu(LzCtx* c, LzOneJob** jobs, int n, size_t inBytes, size_t maxSize, int level)
{
    LzCombine* k = &c->comb;
    LzStage* s = &k->st;
    const size_t slot = ((size_t)LIZARD_COMPRESSBOUND((int)maxSize) + 63) & ~(size_t)63;
    size_t total;
    int i, rc;
    for (i = 0; i < n; i++) { k->h_srcSizes[i] = (uint32_t)jobs[i]->srcSize; k->h_srcOffsets[i] = jobs[i]->inOff; }
    LZ_HIP(hipMemcpyAsync(k->d_srcSizes, k->h_srcSizes, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    LZ_HIP(hipMemcpyAsync(k->d_srcOffsets, k->h_srcOffsets, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, s->stream));
    LZ_HIP(hipMemcpyAsync(s->d_in, s->h_in, inBytes, hipMemcpyHostToDevice, s->stream));
    c->hostKernelMs = -1.0f;
    if ((rc = lzk_launch(c, s->d_in, (size_t)n, maxSize, maxSize, s->d_slots, slot, s->d_sizes, level, s->stream, NULL, NULL, k->d_srcSizes, k->d_srcOffsets))) return rc;
    /* the compaction writes the valid bytes STRAIGHT INTO the pinned host buffer (posted PCIe writes, coalesced 16-byte stores):
     * no separate D2H of the payload and no host round trip between "sizes known" and "payload requested" */
    {
        void* d_out = NULL;
        LZ_HIP(hipHostGetDevicePointer(&d_out, s->h_out, 0));
        lzk_pack_launch(s->d_in, s->d_slots, slot, s->d_sizes, s->d_offsets, d_out, (uint32_t)n, (uint32_t)maxSize, (uint32_t)maxSize, LZK_PACK_PAYLOAD, s->stream);
    }
    LZ_HIP(hipGetLastError());
    LZ_HIP(hipMemcpyAsync(s->h_sizes, s->d_sizes, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    LZ_HIP(hipMemcpyAsync(s->h_offsets, s->d_offsets, ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream));
    LZ_HIP(hipStreamSynchronize(s->stream));
    total = (size_t)s->h_offsets[n];
    (void)total;
    for (i = 0; i < n; i++) { jobs[i]->out = s->h_out + s->h_offsets[i]; jobs[i]->csize = s->h_sizes[i]; }
    return 0;
}

static int batch_buffers(LzCtx* c, int n, size_t inBytes, size_t maxSize)
{
    LzCombine* k = &c->comb;
    LzStage* s = &k->st;
    const size_t slot = ((size_t)LIZARD_COMPRESSBOUND((int)maxSize) + 63) & ~(size_t)63;
    int rc = lzk_ctx_init(c);
    if (rc) return rc;
    if (!s->stream) LZ_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    if ((rc = ensure_dev(c, (void**)&s->d_in, &s->d_in_cap, inBytes + 64))) return rc;
    if ((rc = ensure_dev(c, (void**)&s->d_slots, &s->d_slots_cap, (size_t)n * slot))) return rc;
    if ((rc = ensure_pinned((void**)&s->h_in, &s->h_in_cap, inBytes + 64))) return rc;
    if ((rc = ensure_pinned((void**)&s->h_out, &s->h_out_cap, (size_t)n * slot))) return rc;
    if (k->raggedCap < (size_t)n + 1) {
        const size_t cap = (size_t)n + 65;
        if (s->d_sizes) { (void)hipFree(s->d_sizes); s->d_sizes = NULL; }
        if (s->d_offsets) { (void)hipFree(s->d_offsets); s->d_offsets = NULL; }
        if (k->d_srcSizes) { (void)hipFree(k->d_srcSizes); k->d_srcSizes = NULL; }
        if (k->d_srcOffsets) { (void)hipFree(k->d_srcOffsets); k->d_srcOffsets = NULL; }
        if (s->h_sizes) { (void)hipHostFree(s->h_sizes); s->h_sizes = NULL; }
        if (s->h_offsets) { (void)hipHostFree(s->h_offsets); s->h_offsets = NULL; }
        if (k->h_srcSizes) { (void)hipHostFree(k->h_srcSizes); k->h_srcSizes = NULL; }
        if (k->h_srcOffsets) { (void)hipHostFree(k->h_srcOffsets); k->h_srcOffsets = NULL; }
        k->raggedCap = 0;
        LZ_HIP(hipMalloc((void**)&s->d_sizes, cap * sizeof(uint32_t)));
        LZ_HIP(hipMalloc((void**)&s->d_offsets, cap * sizeof(uint64_t)));
        LZ_HIP(hipMalloc((void**)&k->d_srcSizes, cap * sizeof(uint32_t)));
        LZ_HIP(hipMalloc((void**)&k->d_srcOffsets, cap * sizeof(uint64_t)));
        LZ_HIP(hipHostMalloc((void**)&s->h_sizes, cap * sizeof(uint32_t), hipHostMallocDefault));
        LZ_HIP(hipHostMalloc((void**)&s->h_offsets, cap * sizeof(uint64_t), hipHostMallocDefault));
        LZ_HIP(hipHostMalloc((void**)&k->h_srcSizes, cap * sizeof(uint32_t), hipHostMallocDefault));
        LZ_HIP(hipHostMalloc((void**)&k->h_srcOffsets, cap * sizeof(uint64_t), hipHostMallocDefault));
        k->raggedCap = cap;
    }
    return 0;
}

void lzk_combiner_free(LzCtx* c)                /* context locked, no batch under way (lzk_combiner_quiesce) */
{
    LzCombine* k = &c->comb;
    LzStage* s = &k->st;
    if (s->h_in) (void)hipHostFree(s->h_in);
    if (s->h_out) (void)hipHostFree(s->h_out);
    if (s->h_sizes) (void)hipHostFree(s->h_sizes);
    if (s->h_offsets) (void)hipHostFree(s->h_offsets);
    if (k->h_srcSizes) (void)hipHostFree(k->h_srcSizes);
    if (k->h_srcOffsets) (void)hipHostFree(k->h_srcOffsets);
    if (s->d_in) lzk_dev_free(c, s->d_in, s->d_in_cap);
    if (s->d_slots) lzk_dev_free(c, s->d_slots, s->d_slots_cap);
    if (s->d_packed) lzk_dev_free(c, s->d_packed, s->d_packed_cap);
    if (s->d_sizes) (void)hipFree(s->d_sizes);
    if (s->d_offsets) (void)hipFree(s->d_offsets);
    if (k->d_srcSizes) (void)hipFree(k->d_srcSizes);
    if (k->d_srcOffsets) (void)hipFree(k->d_srcOffsets);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    memset(s, 0, sizeof *s);
    k->d_srcSizes = NULL; k->d_srcOffsets = NULL; k->h_srcSizes = NULL; k->h_srcOffsets = NULL; k->raggedCap = 0;
}
void lzk_combiner_quiesce(LzCtx* c)
{
    pthread_mutex_lock(&c->comb.mu);
    while (c->comb.busy) pthread_cond_wait(&c->comb.cv, &c->comb.mu);
    c->comb.busy = 1;                           /* callers queue up until lzk_combiner_resume */
    pthread_mutex_unlock(&c->comb.mu);
}
void lzk_combiner_resume(LzCtx* c)
{
    pthread_mutex_lock(&c->comb.mu);
    c->comb.busy = 0;
    pthread_cond_broadcast(&c->comb.cv);
    pthread_mutex_unlock(&c->comb.mu);
}

/* Called with comb.mu held by a caller whose job is queued and that found no batch under way: runs one batch (its own job is
 * in it) up to the point where every member knows its result.  Returns with comb.mu held. */
static void lead_batch(LzCtx* c, LzOneJob* mine)
{
    LzCombine* k = &c->comb;
    LzOneJob* jobs[LZ_ONE_MAX_JOBS];
    LzOneJob *j, *keepHead = NULL, *keepTail = NULL;
    size_t inBytes = 0, maxSize = 0;
    const int level = mine->level;
    int n = 0, i, rc;
    LzGuard g;
    double t0, t1, t2, t3;
    k->busy = 1;
    t0 = now_s();
    /* A launch takes the same ~3 ms whatever it carries: blocks per launch is what counts.  The callers of the previous batch come
     * back one by one; a leader that finds fewer of them queued than that batch had gives the stragglers a moment (at most
     * LZ_ONE_WINDOW_US) before it takes the queue.  A lone caller never waits (lastN == 1). */
    if (k->queued < k->lastN) {
        struct timespec until;
        clock_gettime(CLOCK_REALTIME, &until);
        until.tv_nsec += LZ_ONE_WINDOW_US * 1000L;
        if (until.tv_nsec >= 1000000000L) { until.tv_nsec -= 1000000000L; until.tv_sec += 1; }
        k->collecting = 1;
        while (k->queued < k->lastN && pthread_cond_timedwait(&k->cv, &k->mu, &until) == 0) {}
        k->collecting = 0;
    }
    /* members: my job FIRST (the leader is always a member of the batch it leads, however many callers are queued in front of
     * it), then every queued job of my level, in arrival order, while they fit; the others stay queued */
    mine->inOff = 0; inBytes = maxSize = (size_t)mine->srcSize;
    jobs[n++] = mine;
    for (j = k->head; j; ) {
        LzOneJob* const next = j->next;
        const size_t at = (inBytes + 255) & ~(size_t)255;
        if (j == mine) { j->next = NULL; j = next; continue; }              /* (already a member: leaves the queue) */
        if (j->level == level && n < LZ_ONE_MAX_JOBS && at + (size_t)j->srcSize <= LZ_ONE_MAX_BYTES) {
            j->inOff = at; inBytes = at + (size_t)j->srcSize;
            if ((size_t)j->srcSize > maxSize) maxSize = (size_t)j->srcSize;
            jobs[n++] = j;
        } else {
            j->next = NULL;
            if (keepTail) keepTail->next = j; else keepHead = j;
            keepTail = j;
        }
        j = next;
    }
    k->head = keepHead; k->tail = keepTail;
    k->queued -= n; k->lastN = n;
    k->batches++; k->jobs += (unsigned long long)n;
    pthread_mutex_unlock(&k->mu);

    lzk_guard_acquire(&g);                      /* context lock + this device current */
    rc = g.rc;
    if (!rc && g.c != c) {                      /* another thread moved the process default device since this caller queued: the batch's
                                                 * buffers and stream belong to `c`, whose lock this is not — fail the batch, touch nothing */
        snprintf(lzk_err(), LZK_ERR_BYTES, "the selected device changed while a batch of one-block calls was forming");
        rc = -LIZARDGPU_ERR_ARG;
    }
    if (!rc) rc = batch_buffers(c, n, inBytes, maxSize);
    t1 = t2 = now_s();
    if (!rc) {
        /* every member copies its own input in (mine: here) */
        pthread_mutex_lock(&k->mu);
        k->pendingIn = n - 1;
        for (i = 0; i < n; i++) if (jobs[i] != mine) jobs[i]->state = JOB_COPY_IN;
        pthread_cond_broadcast(&k->cv);
        pthread_mutex_unlock(&k->mu);
        memcpy(k->st.h_in + mine->inOff, mine->src, (size_t)mine->srcSize);
        pthread_mutex_lock(&k->mu);
        while (k->pendingIn) pthread_cond_wait(&k->cv, &k->mu);
        pthread_mutex_unlock(&k->mu);
        t2 = now_s();
        rc = batch_on_gpu(c, jobs, n, inBytes, maxSize, level);
        if (rc) { (void)hipStreamSynchronize(k->st.stream); (void)hipGetLastError(); }      /* nothing of this batch is left in flight */
    }
    if (g.c) lzk_guard_release(&g);
    t3 = now_s();

    pthread_mutex_lock(&k->mu);
    k->tLock += t1 - t0; k->tCopyIn += t2 - t1; k->tGpu += t3 - t2; k->tBusySince = t3;
    k->pendingOut = n;
    for (i = 0; i < n; i++) {
        if (rc) { jobs[i]->state = JOB_FAILED; jobs[i]->result = rc; memcpy(jobs[i]->errText, lzk_err(), LZK_ERR_BYTES); }
        else jobs[i]->state = JOB_RESULT;
    }
    pthread_cond_broadcast(&k->cv);
}

int lzgpu_compress_one(const void* src, int srcSize, void* dst, int maxDstSize, int level)
{
    LzOneJob job;
    LzCtx* c;
    LzCombine* k;
    if (srcSize < 0 || (unsigned)srcSize > (unsigned)LIZARD_MAX_INPUT_SIZE) return 0;
    c = lzk_ctx_peek();
    if (!c) return -LIZARDGPU_ERR_NO_DEVICE;
    if (srcSize == 0) {                         /* reference: level byte only (lizard_compress.c:488-494) */
        if (maxDstSize < 1) return 0;
        ((uint8_t*)dst)[0] = (uint8_t)lzk_clamp_level(level);
        return 1;
    }
    k = &c->comb;
    memset(&job, 0, sizeof job);
    job.src = src; job.srcSize = srcSize; job.dst = dst; job.maxDst = maxDstSize; job.level = lzk_clamp_level(level);
    pthread_mutex_lock(&k->mu);
    if (k->tail) k->tail->next = &job; else k->head = &job;
    k->tail = &job;
    k->queued++;
    if (k->collecting) pthread_cond_broadcast(&k->cv);      /* a leader is waiting for stragglers */
    for (;;) {
        if (job.state == JOB_QUEUED && !k->busy) { lead_batch(c, &job); continue; }
        if (job.state == JOB_COPY_IN) {
            pthread_mutex_unlock(&k->mu);
            memcpy(k->st.h_in + job.inOff, src, (size_t)srcSize);
            pthread_mutex_lock(&k->mu);
            job.state = JOB_COPIED_IN;
            if (--k->pendingIn == 0) pthread_cond_broadcast(&k->cv);
            continue;
        }
        if (job.state == JOB_RESULT || job.state == JOB_FAILED) break;
        pthread_cond_wait(&k->cv, &k->mu);
    }
    pthread_mutex_unlock(&k->mu);
    if (job.state == JOB_RESULT) {
        /* The reference's room checks compare against oend = dst + maxDstSize (lizard_compress.c:238, :489): whatever fits is
         * written.  A ONE-byte block is the case the reference gets through by accident: Lizard_compress_generic decrements
         * maxOutputSize after the level byte, writeBlock's raw branch tests `*op + blockSize + 4 > oend` only for the sub-block,
         * and with maxDstSize = srcSize - 1 = 0 (the frame layer's call, lizard_frame.c:461) the unsigned room test wraps: the
         * 6-byte block (level, 0x80, LE24 1, the byte) is emitted and its size returned.  Same here. */
        if ((int)job.csize > maxDstSize && !(srcSize == 1 && maxDstSize == 0)) job.result = 0;
        else { memcpy(dst, job.out, job.csize); job.result = (int)job.csize; }
    } else memcpy(lzk_err(), job.errText, LZK_ERR_BYTES);
    pthread_mutex_lock(&k->mu);                 /* the staging is free for the next batch once the last member has left */
    if (--k->pendingOut == 0) { k->tOut += now_s() - k->tBusySince; k->busy = 0; pthread_cond_broadcast(&k->cv); }
    pthread_mutex_unlock(&k->mu);
    return job.result;
}

/* where the leaders' time went, seconds since the process started: [0] waiting for the context + buffers, [1] members' copy-in,
 * [2] GPU part (H2D, kernel, pack, D2H), [3] until the last member had copied out (tuning aid, not in the public header) */
int LizardGPU_combinerProfile(double out[4])
{
    LzCtx* c = lzk_ctx_peek();
    if (!c) return -LIZARDGPU_ERR_NO_DEVICE;
    pthread_mutex_lock(&c->comb.mu);
    out[0] = c->comb.tLock; out[1] = c->comb.tCopyIn; out[2] = c->comb.tGpu; out[3] = c->comb.tOut;
    pthread_mutex_unlock(&c->comb.mu);
    return 0;
}

/* batches launched / blocks carried by the combiner of the selected device since the process started (tests, bench) */
int LizardGPU_combinerStats(unsigned long long* batches, unsigned long long* blocks)
{
    LzCtx* c = lzk_ctx_peek();
    if (!c) return -LIZARDGPU_ERR_NO_DEVICE;
    pthread_mutex_lock(&c->comb.mu);
    if (batches) *batches = c->comb.batches;
    if (blocks) *blocks = c->comb.jobs;
    pthread_mutex_unlock(&c->comb.mu);
    return 0;
}

/* block i is src[offsets[i] .. offsets[i+1]) (the layout of LizardGPU_compressBlocks_host_packed); synchronous */
static int decompress_host_locked(LzCtx* c, const void* src, const uint64_t* offsets, size_t nBlocks, void* dst, size_t dstStride, uint32_t* outSizes)
{
    LzStage* s = &c->stage[0];
    uint64_t* rel;
    uint64_t* d_off;
    uint32_t* d_out;
    size_t inBytes, i;
    int rc = lzk_ctx_init(c);
    if (rc) return rc;
    /* the offsets are input like the blocks themselves: non-decreasing, every block below 4 GiB, the slots addressable */
    for (i = 0; i < nBlocks; i++) {
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0xFFFFFFFFull) {
            snprintf(lzk_err(), LZK_ERR_BYTES, "bad argument (offsets[%zu..%zu] are not a block)", i, i + 1); return -LIZARDGPU_ERR_ARG;
        }
    }
    if (dstStride > (size_t)-1 / nBlocks) { snprintf(lzk_err(), LZK_ERR_BYTES, "bad argument (nBlocks * dstStride overflows)"); return -LIZARDGPU_ERR_ARG; }
    inBytes = (size_t)(offsets[nBlocks] - offsets[0]);
    if ((rc = ensure_dev(c, (void**)&s->d_in, &s->d_in_cap, inBytes + 64))) return rc;
    if ((rc = ensure_dev(c, (void**)&s->d_slots, &s->d_slots_cap, nBlocks * dstStride))) return rc;
    if ((rc = ensure_dev(c, (void**)&s->d_packed, &s->d_packed_cap, (nBlocks + 1) * sizeof(uint64_t) + nBlocks * sizeof(uint32_t)))) return rc;   /* offsets + sizes ride here */
    d_off = (uint64_t*)s->d_packed;
    d_out = (uint32_t*)(d_off + nBlocks + 1);
    rel = (uint64_t*)malloc((nBlocks + 1) * sizeof(uint64_t));
    if (!rel) { snprintf(lzk_err(), LZK_ERR_BYTES, "out of host memory"); return -LIZARDGPU_ERR_NOMEM; }
    for (i = 0; i <= nBlocks; i++) rel[i] = offsets[i] - offsets[0];
    /* (`rel` and the caller's buffers are read by copies in flight: every way out of here, also a failing one, drains the stream first) */
    do {
        hipError_t e;
        rc = -LIZARDGPU_ERR_HIP;
        if ((e = hipMemcpyAsync(s->d_in, (const uint8_t*)src + offsets[0], inBytes, hipMemcpyHostToDevice, s->stream)) != hipSuccess
            || (e = hipMemcpyAsync(d_off, rel, (nBlocks + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s->stream)) != hipSuccess) {
            snprintf(lzk_err(), LZK_ERR_BYTES, "hipMemcpyAsync failed: %s", hipGetErrorString(e)); break;
        }
        if ((rc = lzk_launch_decompress(c, s->d_in, d_off, 0, NULL, nBlocks, s->d_slots, dstStride, d_out, s->stream))) break;
        rc = -LIZARDGPU_ERR_HIP;
        if ((e = hipMemcpyAsync(outSizes, d_out, nBlocks * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream)) != hipSuccess
            || (e = hipMemcpyAsync(dst, s->d_slots, nBlocks * dstStride, hipMemcpyDeviceToHost, s->stream)) != hipSuccess) {
            snprintf(lzk_err(), LZK_ERR_BYTES, "hipMemcpyAsync failed: %s", hipGetErrorString(e)); break;
        }
        rc = 0;
    } while (0);
    if (hipStreamSynchronize(s->stream) != hipSuccess && !rc) { snprintf(lzk_err(), LZK_ERR_BYTES, "hipStreamSynchronize failed"); rc = -LIZARDGPU_ERR_HIP; }
    free(rel);
    return rc;
}
int LizardGPU_decompressBlocks_host(const void* src, const uint64_t* offsets, size_t nBlocks, void* dst, size_t dstStride, uint32_t* outSizes)
{
    LzGuard g;
    int rc;
    lzk_guard_acquire(&g);
    if (g.rc) return g.rc;
    if (!src || !offsets || !dst || !outSizes || nBlocks == 0 || dstStride == 0) { snprintf(lzk_err(), LZK_ERR_BYTES, "bad argument"); rc = -LIZARDGPU_ERR_ARG; }
    else rc = decompress_host_locked(g.c, src, offsets, nBlocks, dst, dstStride, outSizes);
    lzk_guard_release(&g);
    return rc;
}

/* twin of Lizard_decompress_safe (reference lib/lizard_decompress.h:64 / lizard_decompress.c:267): one block, host buffers */
int LizardGPU_decompress_safe(const char* source, char* dest, int compressedSize, int maxDecompressedSize)
{
    uint64_t offs[2];
    uint32_t out = 0;
    char* tmp;
    int rc;
    if (compressedSize < 0 || maxDecompressedSize < 0 || !source || !dest) return -1;
    if (compressedSize == 0) return 0;                          /* reference: inputSize < 1 -> 0 */
    offs[0] = 0; offs[1] = (uint64_t)compressedSize;
    tmp = (char*)malloc((size_t)maxDecompressedSize + 1);
    if (!tmp) return -1;
    rc = LizardGPU_decompressBlocks_host(source, offs, 1, tmp, (size_t)maxDecompressedSize + 1, &out);
    if (rc || out == 0xFFFFFFFFu || out > (uint32_t)maxDecompressedSize) { free(tmp); return -1; }
    memcpy(dest, tmp, out);
    free(tmp);
    return (int)out;
}
