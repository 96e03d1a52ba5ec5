/* tests/combiner_fake.c — TEST INFRASTRUCTURE: the one-block COMBINER of lizard_amd/csrc/lizard_pipeline_host.c (queue, leaders,
 * two batches in flight, members' copy-in / copy-out, quiesce / resume) on a CPU, with no GPU and no HIP runtime: this file supplies
 * the few HIP calls and lzk_* shims the pipeline file needs — "device memory" is host memory, a "launch" compresses every block of
 * the ragged batch with the ORACLE (oracle/liblizard_oracle.so: test infrastructure, this harness is not a product path) on the
 * calling thread and then sleeps like a 3 ms kernel, streams are synchronous.  N threads hammer lzgpu_compress_one with mixed
 * sizes and levels and compare every result with the oracle, while another thread quiesces / resumes the combiner the way
 * LizardGPU_shutdown / _trim do.  Built with -DLZ_ONE_MAX_JOBS=3 as well (batches smaller than the crowd: the round-4 advisor
 * finding) and under -fsanitize=thread / address by tests/test_combiner_cpu.py.
 *   usage: combiner_fake [threads=24] [seconds=2] [closer=1]                                      exit 0 = no mismatch, no hang */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "../lizard_amd/csrc/lizard_pipeline_host.c"      /* the unit under test, compiled into this harness */
#include "lizard_oracle.h"

/* ---- the HIP runtime the pipeline file calls, on host memory ---- */
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags) { (void)flags; *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st) { (void)k; (void)st; memcpy(d, s, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t st) { (void)st; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned f) { (void)f; *st = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t st) { free(st); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned f) { (void)st; (void)e; (void)f; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) { (void)e; (void)st; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { (void)e; return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { (void)a; (void)b; *ms = 0.0f; return hipSuccess; }
hipError_t hipSetDevice(int d) { (void)d; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { (void)e; return "fake"; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned f) { (void)f; *d = h; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) { (void)p; memset(a, 0, sizeof *a); return hipErrorInvalidValue; }

/* ---- the shims of lizard_gpu_ctx.h ---- */
static LzCtx g_c;
static __thread char t_err[LZK_ERR_BYTES];
static int g_launches;
/* (the sleeps widen the windows between the steps of a batch — a real context lock is contended and real allocations take
 *  milliseconds the first time: a member that could slip into another role between two steps gets its chance here) */
void  lzk_guard_acquire(LzGuard* g) { usleep(200); pthread_mutex_lock(&g_c.mu); usleep(100); g->c = &g_c; g->saved = -1; g->rc = 0; }
void  lzk_guard_release(LzGuard* g) { if (g->c) pthread_mutex_unlock(&g_c.mu); g->c = NULL; }
char* lzk_err(void) { return t_err; }
int   lzk_ctx_init(LzCtx* c) { c->ready = 1; return 0; }
int   lzk_clamp_level(int level) { return level > 49 ? 49 : level < 10 ? 17 : level; }
LzCtx* lzk_ctx_peek(void) { return &g_c; }
int   lzk_dev_alloc(LzCtx* c, void** p, size_t n) { (void)c; *p = malloc(n ? n : 1); return *p ? 0 : -LIZARDGPU_ERR_NOMEM; }
void  lzk_dev_free(LzCtx* c, void* p, size_t n) { (void)c; (void)n; free(p); }
size_t lzk_budget(void) { return 0; }
size_t lzk_budget_room_for_staging(const LzCtx* c) { (void)c; return (size_t)-1; }
int LizardGPU_levelSupported(int level) { return lzo_level_supported(level); }
/* a "launch": every block of the ragged batch through the oracle, then the latency of a one-wave-per-block kernel */
int lzk_launch(LzCtx* c, const void* d_src, size_t nBlocks, size_t blockSize, size_t lastBlockSize, void* d_dst, size_t dstStride,
               uint32_t* d_sizes, int level, hipStream_t stream, hipEvent_t k0, hipEvent_t k1, const uint32_t* d_srcSizes, const uint64_t* d_srcOffsets)
{
    size_t b;
    (void)c; (void)stream; (void)k0; (void)k1; (void)lastBlockSize;
    __atomic_add_fetch(&g_launches, 1, __ATOMIC_RELAXED);
    for (b = 0; b < nBlocks; b++) {
        const size_t n = d_srcSizes ? d_srcSizes[b] : blockSize;
        const uint8_t* in = (const uint8_t*)d_src + (d_srcOffsets ? d_srcOffsets[b] : b * blockSize);
        d_sizes[b] = (uint32_t)lzo_compress(in, (uint8_t*)d_dst + b * dstStride, (int)n, (int)dstStride, level);
    }
    usleep(1500);
    return 0;
}
int lzk_launch_decompress(LzCtx* c, const void* a, const uint64_t* b, size_t s, const uint32_t* d, size_t n, void* e, size_t f, uint32_t* g, hipStream_t h)
{ (void)c; (void)a; (void)b; (void)s; (void)d; (void)n; (void)e; (void)f; (void)g; (void)h; return -LIZARDGPU_ERR_HIP; }
void lzk_pack_launch(const void* d_in, const void* d_slots, size_t slot, const uint32_t* d_sizes, uint64_t* d_offsets, void* d_packed,
                     uint32_t nb, uint32_t blockSize, uint32_t lastBlockSize, int mode, hipStream_t stream)
{
    uint64_t off = 0; uint32_t b;
    (void)d_in; (void)blockSize; (void)lastBlockSize; (void)mode; (void)stream;
    for (b = 0; b < nb; b++) { d_offsets[b] = off; memcpy((uint8_t*)d_packed + off, (const uint8_t*)d_slots + (size_t)b * slot, d_sizes[b]); off += d_sizes[b]; }
    d_offsets[nb] = off;
}
const char* LizardGPU_lastError(void) { return t_err; }

/* ---- the workload ---- */
static const int kSizes[] = { 1, 21, 100, 4096, 20000, 65537 };
static const int kLevels[] = { 10, 10, 10, 21, 30 };
typedef struct { int id; long calls, bad; } Worker;
static int g_stop;
#define STOPPED() __atomic_load_n(&g_stop, __ATOMIC_RELAXED)
static unsigned char* g_in[6]; static unsigned char* g_want[6][5]; static int g_wantSize[6][5];
static void* worker(void* a)
{
    Worker* w = (Worker*)a;
    unsigned char* dst = malloc(70000 + 1024);
    unsigned r = 12345u * (unsigned)(w->id + 1);
    while (!STOPPED()) {
        r = r * 1664525u + 1013904223u;
        const int si = (int)((r >> 8) % 6u), li = (int)((r >> 16) % 5u), n = kSizes[si];
        const int cap = ((r >> 24) & 3u) == 0 ? n - 1 : 70000;                 /* a quarter of the calls: the frame layer's capacity */
        const int got = lzgpu_compress_one(g_in[si], n, dst, cap, kLevels[li]);
        const int want = g_wantSize[si][li] <= cap || (n == 1 && cap == 0) ? g_wantSize[si][li] : 0;
        if (got != want || (got > 0 && memcmp(dst, g_want[si][li], (size_t)got) != 0)) w->bad++;
        w->calls++;
    }
    free(dst);
    return NULL;
}
static void* closer(void* a)                    /* what LizardGPU_shutdown / _trim / _setMemoryBudget do to the combiner, every few ms */
{
    (void)a;
    while (!STOPPED()) {
        usleep(7000);
        lzk_combiner_quiesce(&g_c);
        pthread_mutex_lock(&g_c.mu); lzk_combiner_free(&g_c); pthread_mutex_unlock(&g_c.mu);
        lzk_combiner_resume(&g_c);
    }
    return NULL;
}
static void* watchdog(void* a) { const double secs = *(double*)a; usleep((useconds_t)((secs + 20.0) * 1e6)); { fprintf(stderr, "combiner_fake: HANG (watchdog)\n"); _exit(3); } return NULL; }

int main(int argc, char** argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 24;
    double secs = argc > 2 ? atof(argv[2]) : 2.0;
    const int withCloser = argc > 3 ? atoi(argv[3]) : 1;
    pthread_t th[256], cl, wd;
    Worker w[256];
    long calls = 0, bad = 0;
    unsigned long long batches = 0, blocks = 0;
    int i, j;
    if (threads > 256) return 2;
    pthread_mutex_init(&g_c.mu, NULL); pthread_mutex_init(&g_c.comb.mu, NULL); pthread_cond_init(&g_c.comb.cv, NULL);
    for (i = 0; i < 6; i++) {
        g_in[i] = malloc((size_t)kSizes[i] + 64);
        lzo_datagen(g_in[i], (size_t)kSizes[i], 0.5, 0.0, (unsigned)i + 3u);
        for (j = 0; j < 5; j++) { g_want[i][j] = malloc(70000 + 1024); g_wantSize[i][j] = lzo_compress(g_in[i], g_want[i][j], kSizes[i], 70000, kLevels[j]); }
    }
    pthread_create(&wd, NULL, watchdog, &secs); pthread_detach(wd);
    for (i = 0; i < threads; i++) { w[i].id = i; w[i].calls = w[i].bad = 0; pthread_create(&th[i], NULL, worker, &w[i]); }
    if (withCloser) pthread_create(&cl, NULL, closer, NULL);
    usleep((useconds_t)(secs * 1e6));
    __atomic_store_n(&g_stop, 1, __ATOMIC_RELAXED);
    for (i = 0; i < threads; i++) { pthread_join(th[i], NULL); calls += w[i].calls; bad += w[i].bad; }
    if (withCloser) pthread_join(cl, NULL);
    LizardGPU_combinerStats(&batches, &blocks);
    printf("combiner_fake: %d threads, %ld calls, %ld mismatches, %llu batches for %llu blocks (%d launches), max %d members per batch\n",
           threads, calls, bad, batches, blocks, g_launches, (int)LZ_ONE_MAX_JOBS);
    return bad ? 1 : 0;
}
