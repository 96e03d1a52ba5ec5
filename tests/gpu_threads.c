/* tests/gpu_threads.c — TEST / BENCH INFRASTRUCTURE: aggregate rate of N host threads calling the reference's ONE-BLOCK entry
 * point (Lizard_compress of liblizard_amd.so) at once, every result compared with the oracle.
 *   usage: gpu_threads [threads=64] [level=10] [bytes=262144] [seconds=2] [json]
 * Prints one line per thread count (1, 8, 16, 32, ... up to `threads`): MB/s aggregate, calls, mismatches, and how many launches
 * carried how many blocks (LizardGPU_combinerStats).  No Python in the loop: 64 Python threads spend more time on the GIL than
 * the GPU spends on the batch.
 *   build: gcc -O2 tests/gpu_threads.c -o tests/gpu_threads -Iinclude -Ioracle -Llizard_amd -llizard_amd -Loracle -llizard_oracle
 *              -lpthread -Wl,-rpath,'$ORIGIN/../lizard_amd:$ORIGIN/../oracle' */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lizard_amd.h"
#include "lizard_oracle.h"

int LizardGPU_combinerProfile(double out[4]);
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

typedef struct { int id, level, n, bound; unsigned char* src; unsigned char* want; int wantSize; long calls, bad; } Worker;
static volatile int g_go, g_stop;
static pthread_barrier_t g_bar;

static void* worker(void* arg)
{
    Worker* w = (Worker*)arg;
    unsigned char* dst = malloc((size_t)w->bound + 64);
    pthread_barrier_wait(&g_bar);
    while (!g_stop) {
        const int r = Lizard_compress((const char*)w->src, (char*)dst, w->n, w->bound, w->level);
        if (r != w->wantSize || memcmp(dst, w->want, (size_t)r) != 0) w->bad++;
        w->calls++;
    }
    free(dst);
    return NULL;
}

int main(int argc, char** argv)
{
    const int maxThreads = argc > 1 ? atoi(argv[1]) : 64, level = argc > 2 ? atoi(argv[2]) : 10, n = argc > 3 ? atoi(argv[3]) : 262144;
    const double secs = argc > 4 ? atof(argv[4]) : 2.0;
    const int json = argc > 5;
    const int bound = Lizard_compressBound(n);
    Worker* w = calloc((size_t)maxThreads, sizeof *w);
    int t, threads, fails = 0, first = 1;
    for (t = 0; t < maxThreads; t++) {
        w[t].id = t; w[t].level = level; w[t].n = n; w[t].bound = bound;
        w[t].src = malloc((size_t)n + 64); w[t].want = malloc((size_t)bound + 64);
        lzo_datagen(w[t].src, (size_t)n, 0.5, 0.0, (unsigned)t);
        w[t].wantSize = lzo_compress(w[t].src, w[t].want, n, bound, level);
    }
    {   /* context creation outside the timed region */
        unsigned char* dst = malloc((size_t)bound + 64);
        if (Lizard_compress((const char*)w[0].src, (char*)dst, n, bound, level) != w[0].wantSize) { fprintf(stderr, "gpu_threads: first call failed: %s\n", LizardGPU_lastError()); return 1; }
        free(dst);
    }
    if (json) printf("[");
    for (threads = 1; threads <= maxThreads; threads = threads < 8 ? 8 : threads * 2) {
        pthread_t th[1024];
        unsigned long long b0 = 0, j0 = 0, b1 = 0, j1 = 0;
        long calls = 0, bad = 0;
        double t0, dt;
        if (threads > 1024) break;
        LizardGPU_combinerStats(&b0, &j0);
        g_stop = 0;
        pthread_barrier_init(&g_bar, NULL, (unsigned)threads + 1);
        for (t = 0; t < threads; t++) { w[t].calls = 0; w[t].bad = 0; pthread_create(&th[t], NULL, worker, &w[t]); }
        pthread_barrier_wait(&g_bar);
        t0 = now();
        { struct timespec ts; ts.tv_sec = (time_t)secs; ts.tv_nsec = (long)((secs - (double)(time_t)secs) * 1e9); nanosleep(&ts, NULL); }
        g_stop = 1;
        for (t = 0; t < threads; t++) pthread_join(th[t], NULL);
        dt = now() - t0;
        pthread_barrier_destroy(&g_bar);
        for (t = 0; t < threads; t++) { calls += w[t].calls; bad += w[t].bad; }
        LizardGPU_combinerStats(&b1, &j1);
        fails += bad != 0;
        if (!json && getenv("GPU_THREADS_PROFILE")) {
            static double p0[4]; double p1[4];
            LizardGPU_combinerProfile(p1);
            printf("   per launch: lock %.0f us, copy-in %.0f us, gpu %.0f us, copy-out %.0f us\n", 1e6 * (p1[0] - p0[0]) / (double)(b1 - b0),
                   1e6 * (p1[1] - p0[1]) / (double)(b1 - b0), 1e6 * (p1[2] - p0[2]) / (double)(b1 - b0), 1e6 * (p1[3] - p0[3]) / (double)(b1 - b0));
            memcpy(p0, p1, sizeof p0);
        }
        if (json) printf("%s{\"threads\": %d, \"MB_s\": %.1f, \"calls\": %ld, \"mismatches\": %ld, \"launches\": %llu, \"blocks\": %llu}", first ? "" : ", ",
                         threads, (double)calls * n / dt / 1e6, calls, bad, b1 - b0, j1 - j0);
        else printf("threads %4d level %d %d B: %9.1f MB/s aggregate, %ld calls, mismatches %ld, %llu launches for %llu blocks\n",
                    threads, level, n, (double)calls * n / dt / 1e6, calls, bad, b1 - b0, j1 - j0);
        fflush(stdout);
        first = 0;
        if (threads == maxThreads) break;
        if (threads * 2 > maxThreads && threads < maxThreads) threads = maxThreads / 2;     /* (the loop doubles it: ends on maxThreads) */
    }
    if (json) printf("]\n");
    return fails ? 1 : 0;
}
