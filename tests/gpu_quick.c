/* tests/gpu_quick.c — TEST INFRASTRUCTURE: a seconds-long GPU sanity check with a hang watchdog.
 * usage: gpu_quick [nblocks] [level] [reps] [matchProbaPercent]
 * Runs the product library (liblizard_amd.so, C ABI) on a handful of blocks per level, compares every
 * byte with the oracle (oracle/liblizard_oracle.so) and prints kernel-only throughput of a small
 * batch.  No Python/torch start-up cost: meant to be the FIRST command of every gpurun call so a
 * broken kernel costs seconds, not the GPU budget.  Exit: 0 ok, 1 mismatch, 2 watchdog (hang).
 *   build: gcc -O2 tests/gpu_quick.c -o tests/gpu_quick -Iinclude -Ioracle -Llizard_amd -llizard_amd \
 *              -Loracle -llizard_oracle -lpthread -Wl,-rpath,'$ORIGIN/../lizard_amd:$ORIGIN/../oracle'
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "lizard_amd.h"
#include "lizard_oracle.h"

static volatile int g_deadline = 0;      /* seconds left for the current case; 0 = idle */
static const char* volatile g_case = "";

static void* watchdog(void* arg)
{
    (void)arg;
    for (;;) {
        sleep(1);
        if (g_deadline > 0 && --g_deadline == 0) {
            fprintf(stderr, "gpu_quick: WATCHDOG — case '%s' did not finish (kernel hang?)\n", g_case);
            _exit(2);
        }
    }
    return NULL;
}

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

static double g_call_ms;                    /* the Lizard_compress call alone (one block through the reference ABI), without the checker */
static double now(void);
static int check_one(const unsigned char* src, int n, int level)
{
    int bound = Lizard_compressBound(n);
    unsigned char* a = malloc(bound + 16), *b = malloc(bound + 16);
    double t0 = now();
    int ra = Lizard_compress((const char*)src, (char*)a, n, bound, level);
    g_call_ms = (now() - t0) * 1e3;
    int rb = lzo_compress(src, b, n, bound, level);
    int ok = ra == rb && memcmp(a, b, ra) == 0;
    if (!ok) fprintf(stderr, "  MISMATCH level %d n %d: gpu %d oracle %d (%s)\n", level, n, ra, rb, LizardGPU_lastError());
    free(a); free(b);
    return ok;
}

int main(int argc, char** argv)
{
    static const int sizes[] = { 1, 20, 21, 100, 4096, 65537, 131072, 131073, 262144, 300000 };
    static const int levels[] = { 10, 30, 11, 31, 21, 41, 22, 42, 13, 14, 15, 16, 17, 34, 35, 36, 37, 38, 12, 32, 33, 20, 40 };
    int nb = argc > 1 ? atoi(argv[1]) : 512, bs = 262144, fails = 0;
    int only = argc > 2 ? atoi(argv[2]) : 0;          /* restrict to one level */
    int reps = argc > 3 ? atoi(argv[3]) : 1;          /* repeat the batch (profiling) */
    pthread_t th;
    pthread_create(&th, NULL, watchdog, NULL);

    g_case = "init"; g_deadline = 60;
    printf("resident waves: %d\n", LizardGPU_residentWaves());
    g_deadline = 0;

    unsigned char* buf = malloc((size_t)nb * bs);
    double proba = argc > 4 ? atof(argv[4]) / 100.0 : 0.5;   /* datagen match probability in percent (default P50) */
    for (int b = 0; b < nb; b++) lzo_datagen(buf + (size_t)b * bs, (size_t)bs, proba, 0.0, (unsigned)b);

    for (unsigned li = 0; li < sizeof levels / sizeof *levels; li++) {
        int level = levels[li];
        char name[64];
        if (!LizardGPU_levelSupported(level) || (only && level != only)) continue;
        for (unsigned si = 0; si < sizeof sizes / sizeof *sizes; si++) {
            snprintf(name, sizeof name, "L%d one block n=%d", level, sizes[si]);
            g_case = name; g_deadline = 30;
            double t0 = now();
            int ok = check_one(buf, sizes[si], level);
            g_deadline = 0;
            if (!ok) fails++;
            printf("%-28s %s  %.1f ms  (Lizard_compress call %.2f ms)\n", name, ok ? "ok" : "FAIL", (now() - t0) * 1e3, g_call_ms);
            fflush(stdout);
        }
        {   /* zeros + noise */
            unsigned char* z = calloc(1, bs);
            snprintf(name, sizeof name, "L%d zeros", level); g_case = name; g_deadline = 30;
            int ok = check_one(z, bs, level); g_deadline = 0; if (!ok) fails++;
            printf("%-28s %s\n", name, ok ? "ok" : "FAIL");
            srand(7); for (int i = 0; i < bs; i++) z[i] = (unsigned char)rand();
            snprintf(name, sizeof name, "L%d noise", level); g_case = name; g_deadline = 30;
            ok = check_one(z, bs, level); g_deadline = 0; if (!ok) fails++;
            printf("%-28s %s\n", name, ok ? "ok" : "FAIL");
            free(z);
        }
        {   /* batch through the host entry; kernel-only time from HIP events inside the library */
            size_t stride = (size_t)Lizard_compressBound(bs);
            unsigned char* out = malloc((size_t)nb * stride);
            uint32_t* cs = malloc(sizeof(uint32_t) * nb);
            snprintf(name, sizeof name, "L%d batch %d x %d", level, nb, bs); g_case = name; g_deadline = 120;
            int rc = 0; float ms = 0;
            for (int r = 0; r < reps && !rc; r++) {
                rc = LizardGPU_compressBlocks_host(buf, nb, bs, bs, out, stride, cs, level);
                ms = LizardGPU_lastKernelMs();
                if (reps > 2) printf("    rep %d: kernel %.3f ms\n", r, ms);
            }
            g_deadline = 0;
            size_t tot = 0; int bad = rc != 0;
            for (int b = 0; b < nb && !bad; b++) {
                tot += cs[b];
                if (b < 8 || b % 61 == 0) {
                    unsigned char* o = malloc(stride);
                    int r = lzo_compress(buf + (size_t)b * bs, o, bs, (int)stride, level);
                    if (r != (int)cs[b] || memcmp(o, out + (size_t)b * stride, r)) { bad = 1; fprintf(stderr, "  batch block %d differs\n", b); }
                    free(o);
                }
            }
            if (bad) fails++;
            printf("%-28s %s  kernel %.2f ms  %.2f GB/s input  ratio %.3f\n", name, bad ? "FAIL" : "ok", ms,
                   (double)nb * bs / (ms * 1e-3) / 1e9, tot ? (double)nb * bs / tot : 0.0);
            {   /* instrumented library variant only (LD_LIBRARY_PATH=lizard_amd/variants/prof) */
                int (*dump)(unsigned long long*) = (int (*)(unsigned long long*))dlsym(RTLD_DEFAULT, "LizardGPU_profileDump");
                unsigned long long pr[16];
                if (dump && dump(pr) == 0) {
                    static const char* nm[8] = { "roundA(bytes,hash,LDS,filter) | hc: rounds", "roundB(cand wait,settle) | hc: find_best", "extension | hc: wider #1", "glue+encode | hc: wider #2 + glue", "store drain", "container (split form: consumers)", "table init | hc: chain build | split: producer waits for a buffer", "split: consumer waits for a job" };
                    double sum = 0; for (int k = 0; k < 8; k++) sum += (double)pr[k];   /* slots 8.. are sub-phases of the container slot */
                    for (int k = 0; k < 8; k++) printf("    prof %-48s %6.2f %%  %.3g clk\n", nm[k], 100.0 * pr[k] / sum, (double)pr[k]);
                    printf("    prof raw:"); for (int k = 0; k < 15; k++) printf(" [%d]=%.4g", k, (double)pr[k]); printf("\n");
                    {   static const char* hn[7] = { "huf histogram | hc: chain walk", "huf rank sort | hc: measure", "huf lane-0 tree/codes/header | hc: select", "huf exact sizes", "huf bit packing", "-", "huf entry" };
                        for (int k = 8; k < 15; k++) if (pr[k]) printf("      (sub-phase) %-44s %6.2f %%\n", hn[k - 8], 100.0 * pr[k] / sum); }
                }
            }
            fflush(stdout);
            free(out); free(cs);
        }
    }
    printf("gpu_quick: %s\n", fails ? "FAILED" : "all ok");
    return fails ? 1 : 0;
}
