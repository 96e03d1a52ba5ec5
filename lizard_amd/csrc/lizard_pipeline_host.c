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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

static int ensure_dev(void** p, size_t* cap, size_t need)
{
    if (*cap >= need) return 0;
    if (*p) { LZ_HIP(hipFree(*p)); *p = NULL; *cap = 0; }
    LZ_HIP(hipMalloc(p, need));
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

static int is_pinned_host(const void* p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return at.type == hipMemoryTypeHost;
}

static size_t g_chunk_bytes = 0;                           /* host pipeline chunk (input bytes), 0 = not read yet */
static size_t chunk_bytes(void)
{
    if (!g_chunk_bytes) {
        const char* e = getenv("LIZARDGPU_CHUNK_MB");
        size_t mb = e ? (size_t)strtoul(e, NULL, 10) : 0;
        if (mb < 1 || mb > 65536) mb = 256;                  /* measured on a 4 GiB job: 256 MiB 37 GB/s, 512 MiB 26, 1 GiB 23 (fill and drain of the pipeline) */
        g_chunk_bytes = mb << 20;
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
    if ((rc = ensure_dev((void**)&s->d_in, &s->d_in_cap, ch->inBytes + 64))) return rc;
    if ((rc = ensure_dev((void**)&s->d_slots, &s->d_slots_cap, ch->nb * slot))) return rc;
    if ((rc = ensure_dev((void**)&s->d_packed, &s->d_packed_cap, packedCap))) return rc;
    if ((rc = ensure_pinned((void**)&s->h_out, &s->h_out_cap, packedCap + 64))) return rc;    /* worst case once: a buffer that follows the chunks' sizes is re-pinned again and again */
    if (s->d_meta_cap < ch->nb + 1) {
        if (s->d_sizes) { LZ_HIP(hipFree(s->d_sizes)); s->d_sizes = NULL; }
        if (s->d_offsets) { LZ_HIP(hipFree(s->d_offsets)); s->d_offsets = NULL; }
        s->d_meta_cap = 0;
        LZ_HIP(hipMalloc((void**)&s->d_sizes, (ch->nb + 1) * sizeof(uint32_t)));
        LZ_HIP(hipMalloc((void**)&s->d_offsets, (ch->nb + 1) * sizeof(uint64_t)));
        s->d_meta_cap = ch->nb + 1;
    }
    if (s->h_meta_cap < ch->nb + 1) {
        if (s->h_sizes) { LZ_HIP(hipHostFree(s->h_sizes)); s->h_sizes = NULL; }
        if (s->h_offsets) { LZ_HIP(hipHostFree(s->h_offsets)); s->h_offsets = NULL; }
        s->h_meta_cap = 0;
        LZ_HIP(hipHostMalloc((void**)&s->h_sizes, (ch->nb + 1) * sizeof(uint32_t), hipHostMallocDefault));
        LZ_HIP(hipHostMalloc((void**)&s->h_offsets, (ch->nb + 1) * sizeof(uint64_t), hipHostMallocDefault));
        s->h_meta_cap = ch->nb + 1;
    }
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
    if ((rc = lzk_launch(c, s->d_in, ch->nb, j->blockSize, last, s->d_slots, slot, s->d_sizes, j->level, s->stream, s->k0, s->k1))) return rc;
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
    memset(&p, 0, sizeof p);
    pthread_mutex_init(&p.mu, NULL); pthread_cond_init(&p.cv, NULL);
    p.c = c; p.j = j;
    p.perChunk = chunk_bytes() / j->blockSize;
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

/* Internal shim for the one-block reference entry points (lizard_host.c): compress one host block, honouring the reference's
 * maxDstSize contract: returns the compressed size, 0 if it does not fit (reference lib/lizard_compress.c:543-546), < 0 on a GPU
 * failure. */
static int compress_one_locked(LzCtx* c, const void* src, int srcSize, void* dst, int maxDstSize, int level)
{
    LzStage* s = &c->stage[0];
    const size_t slot = ((size_t)LIZARD_COMPRESSBOUND(srcSize) + 63) & ~(size_t)63;
    uint32_t csize;
    int rc = lzk_ctx_init(c);
    if (rc) return rc;
    if (srcSize == 0) {                 /* reference: level byte only (lizard_compress.c:488-494) */
        if (maxDstSize < 1) return 0;
        ((uint8_t*)dst)[0] = (uint8_t)lzk_clamp_level(level);
        return 1;
    }
    if ((rc = ensure_dev((void**)&s->d_in, &s->d_in_cap, (size_t)srcSize + 64))) return rc;
    if ((rc = ensure_dev((void**)&s->d_slots, &s->d_slots_cap, slot))) return rc;
    if ((rc = ensure_pinned((void**)&s->h_in, &s->h_in_cap, (size_t)srcSize))) return rc;
    if ((rc = ensure_pinned((void**)&s->h_out, &s->h_out_cap, slot + 64))) return rc;
    if (s->d_meta_cap < 2) {
        LZ_HIP(hipMalloc((void**)&s->d_sizes, 2 * sizeof(uint32_t)));
        LZ_HIP(hipMalloc((void**)&s->d_offsets, 2 * sizeof(uint64_t)));
        s->d_meta_cap = 2;
    }
    if (s->h_meta_cap < 2) {
        LZ_HIP(hipHostMalloc((void**)&s->h_sizes, 2 * sizeof(uint32_t), hipHostMallocDefault));
        LZ_HIP(hipHostMalloc((void**)&s->h_offsets, 2 * sizeof(uint64_t), hipHostMallocDefault));
        s->h_meta_cap = 2;
    }
    memcpy(s->h_in, src, (size_t)srcSize);
    LZ_HIP(hipMemcpyAsync(s->d_in, s->h_in, (size_t)srcSize, hipMemcpyHostToDevice, s->stream));
    c->hostKernelMs = -1.0f;
    if ((rc = lzk_launch(c, s->d_in, 1, (size_t)srcSize, (size_t)srcSize, s->d_slots, slot, s->d_sizes, level, s->stream, NULL, NULL))) return rc;
    LZ_HIP(hipMemcpyAsync(s->h_sizes, s->d_sizes, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    LZ_HIP(hipStreamSynchronize(s->stream));
    csize = s->h_sizes[0];
    /* The reference's room checks compare against oend = dst + maxDstSize (lizard_compress.c:238, :489): whatever fits is
     * written.  A ONE-byte block is the case the reference gets through by accident: Lizard_compress_generic decrements
     * maxOutputSize after the level byte, writeBlock's raw branch tests `*op + blockSize + 4 > oend` only for the sub-block,
     * and with maxDstSize = srcSize - 1 = 0 (the frame layer's call, lizard_frame.c:461) the unsigned room test wraps: the
     * 6-byte block (level, 0x80, LE24 1, the byte) is emitted and its size returned.  Same here. */
    if ((int)csize > maxDstSize && !(srcSize == 1 && maxDstSize == 0)) return 0;
    LZ_HIP(hipMemcpyAsync(s->h_out, s->d_slots, csize, hipMemcpyDeviceToHost, s->stream));
    LZ_HIP(hipStreamSynchronize(s->stream));
    memcpy(dst, s->h_out, csize);
    return (int)csize;
}
int lzgpu_compress_one(const void* src, int srcSize, void* dst, int maxDstSize, int level)
{
    LzGuard g;
    int rc;
    if (srcSize < 0 || (unsigned)srcSize > (unsigned)LIZARD_MAX_INPUT_SIZE) return 0;
    lzk_guard_acquire(&g);
    if (g.rc) return g.rc;
    rc = compress_one_locked(g.c, src, srcSize, dst, maxDstSize, level);
    lzk_guard_release(&g);
    return rc;
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
    if ((rc = ensure_dev((void**)&s->d_in, &s->d_in_cap, inBytes + 64))) return rc;
    if ((rc = ensure_dev((void**)&s->d_slots, &s->d_slots_cap, nBlocks * dstStride))) return rc;
    if ((rc = ensure_dev((void**)&s->d_packed, &s->d_packed_cap, (nBlocks + 1) * sizeof(uint64_t) + nBlocks * sizeof(uint32_t)))) return rc;   /* offsets + sizes ride here */
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
