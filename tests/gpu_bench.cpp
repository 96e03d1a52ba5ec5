// tests/gpu_bench.cpp — TEST / MEASUREMENT INFRASTRUCTURE: device-resident timing of the product library through its C ABI,
// without Python start-up.  One process = one (level, blockSize, nBlocks) configuration:
//   gpu_bench <level> <blockSize> <nBlocks> [steps=3] [matchProbaPercent=50] [verifyBlocks=all]
// Input: block b = RDG_genBuffer(blockSize, P, seed b), generated on the device (tools/liblizard_datagen.so).
// Prints the mean kernel time (HIP events inside the library), input GB/s, ratio, and checks the output against the oracle
// (oracle/liblizard_oracle.so) on the host threads: EVERY block, sizes and bytes, by default ("verify ok (all N blocks)"); a number
// checks that many blocks spread over the batch and says so ("verify ok (sample of K)"), 0 says "verify SKIPPED".  A line never
// says "ok" for blocks it did not compare (round 3 found "verify ok" printed for a build with a different compressed total).  Used for tuning-variant sweeps (LD_LIBRARY_PATH picks
// the library build) and for the rocprofv3 counter passes (scripts/gpu_traffic.sh).  Exit: 0 ok, 1 mismatch, 2 error.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <dlfcn.h>
#include <thread>
#include <vector>

#include "lizard_amd.h"
#include "lizard_oracle.h"
extern "C" int LizardTools_datagen_device(void* d_dst, size_t nBlocks, size_t blockSize, double matchProba, double litProba, unsigned seed0, void* stream);   // tools/liblizard_datagen.so

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: gpu_bench level blockSize nBlocks [steps] [P%%] [verify]\n"); return 2; }
    const int level = atoi(argv[1]);
    const size_t bs = strtoull(argv[2], nullptr, 10), nb = strtoull(argv[3], nullptr, 10);
    const int steps = argc > 4 ? atoi(argv[4]) : 3;
    const double P = argc > 5 ? atof(argv[5]) / 100.0 : 0.5;
    const bool verAll = argc <= 6 || !strcmp(argv[6], "all");
    const size_t nver = verAll ? nb : strtoull(argv[6], nullptr, 10);
    const size_t stride = ((size_t)Lizard_compressBound((int)bs) + 63) & ~(size_t)63;
    unsigned char *src = nullptr, *dst = nullptr; uint32_t* sizes = nullptr;
    CK(hipMalloc((void**)&src, nb * bs)); CK(hipMalloc((void**)&dst, nb * stride)); CK(hipMalloc((void**)&sizes, nb * 4));
    if (int e = LizardTools_datagen_device(src, nb, bs, P, 0.0, 0, nullptr)) { fprintf(stderr, "datagen: hipError %d\n", e); return 2; }
    double ms = 0;
    for (int s = -1; s < steps; s++) {                       // one untimed warm-up launch
        int rc = LizardGPU_compressBlocks_device(src, nb, bs, bs, dst, stride, sizes, level, nullptr);
        if (rc) { fprintf(stderr, "launch: %d %s\n", rc, LizardGPU_lastError()); return 2; }
        const float k = LizardGPU_lastKernelMs();
        if (s >= 0) ms += k;
    }
    ms /= steps;
    std::vector<uint32_t> h(nb);
    CK(hipMemcpy(h.data(), sizes, nb * 4, hipMemcpyDeviceToHost));
    unsigned long long tot = 0;
    for (size_t i = 0; i < nb; i++) tot += h[i];
    int bad = 0;
    {
        // the blocks to compare: all of them, or nver spread over the batch; chunks of <= 1 GiB of input come over, the host threads share them
        std::vector<size_t> pick;
        for (size_t k = 0; k < nver && k < nb; k++) pick.push_back(verAll ? k : (nver > 1 ? k * (nb - 1) / (nver - 1) : 0));
        const size_t perChunk = std::max<size_t>(1, ((size_t)1 << 30) / bs);
        std::vector<unsigned char> hin, hout;
        unsigned nt = std::thread::hardware_concurrency();
        if (const char* e = getenv("GPU_BENCH_THREADS")) nt = (unsigned)atoi(e);
        if (nt < 1) nt = 1;
        if (nt > 64) nt = 64;
        std::atomic<int> nbad{0};
        for (size_t p0 = 0; p0 < pick.size(); ) {
            size_t p1 = p0;
            const size_t b0 = pick[p0];
            while (p1 < pick.size() && pick[p1] < b0 + perChunk) p1++;
            const size_t b1 = pick[p1 - 1] + 1;
            if (verAll) {                                        // contiguous: two copies
                hin.resize((b1 - b0) * bs); hout.resize((b1 - b0) * stride);
                CK(hipMemcpy(hin.data(), src + b0 * bs, (b1 - b0) * bs, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hout.data(), dst + b0 * stride, (b1 - b0) * stride, hipMemcpyDeviceToHost));
            } else {
                hin.resize((p1 - p0) * bs); hout.resize((p1 - p0) * stride);
                for (size_t k = p0; k < p1; k++) {
                    CK(hipMemcpy(hin.data() + (k - p0) * bs, src + pick[k] * bs, bs, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hout.data() + (k - p0) * stride, dst + pick[k] * stride, h[pick[k]] <= stride ? h[pick[k]] : stride, hipMemcpyDeviceToHost));
                }
            }
            std::atomic<size_t> next{p0};
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++) th.emplace_back([&] {
                std::vector<unsigned char> want(stride);
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= p1) break;
                    const size_t b = pick[k], at = verAll ? b - b0 : k - p0;
                    const int r = lzo_compress(hin.data() + at * bs, want.data(), (int)bs, (int)stride, level);
                    if (r != (int)h[b] || memcmp(hout.data() + at * stride, want.data(), (size_t)r)) {
                        if (nbad.fetch_add(1) < 5) fprintf(stderr, "block %zu differs from the oracle (gpu %u, oracle %d)\n", b, h[b], r);
                    }
                }
            });
            for (auto& x : th) x.join();
            p0 = p1;
        }
        bad = nbad.load();
    }
    char verdict[96];
    if (nver == 0) snprintf(verdict, sizeof verdict, "SKIPPED");
    else if (bad) snprintf(verdict, sizeof verdict, "FAILED (%d blocks differ)", bad);
    else if (verAll) snprintf(verdict, sizeof verdict, "ok (all %zu blocks, sizes and bytes)", nb);
    else snprintf(verdict, sizeof verdict, "ok (sample of %zu)", nver < nb ? nver : nb);
    printf("L%d %zu x %zu: kernel %.3f ms  %.2f GB/s input  ratio %.4f  compressed %llu  verify %s\n", level, nb, bs, ms,
           (double)nb * bs / (ms * 1e-3) / 1e9, (double)nb * bs / (double)tot, tot, verdict);
    {   // instrumented library variants only (LD_LIBRARY_PATH=lizard_amd/variants/<a -DLZ_PROFILE build>): phase clocks summed over the waves
        int (*dump)(unsigned long long*) = (int (*)(unsigned long long*))dlsym(RTLD_DEFAULT, "LizardGPU_profileDump");
        unsigned long long pr[16];
        if (dump && dump(pr) == 0) { printf("    prof raw (all launches):"); for (int k = 0; k < 15; k++) printf(" [%d]=%.4g", k, (double)pr[k]); printf("\n"); }
    }
    if (argc > 7 && argv[7][0] == 'd') {                     // decompress the batch back (slot layout) and compare a sample with the input
        unsigned char* back = nullptr; uint32_t* osz = nullptr;
        CK(hipMalloc((void**)&back, nb * bs)); CK(hipMalloc((void**)&osz, nb * 4));
        double dms = 0;
        for (int s = -1; s < steps; s++) {
            int rc = LizardGPU_decompressBlocks_device(dst, stride, sizes, nb, back, bs, osz, nullptr);
            if (rc) { fprintf(stderr, "decompress launch: %d %s\n", rc, LizardGPU_lastError()); return 2; }
            const float k = LizardGPU_lastKernelMs();
            if (s >= 0) dms += k;
        }
        dms /= steps;
        std::vector<uint32_t> ho(nb);
        CK(hipMemcpy(ho.data(), osz, nb * 4, hipMemcpyDeviceToHost));
        int dbad = 0;
        for (size_t i = 0; i < nb; i++) if (ho[i] != bs) dbad++;
        std::vector<unsigned char> a(bs), b2(bs);
        for (size_t k = 0; k < 64 && k < nb; k++) {
            const size_t b = k * (nb - 1) / 63;
            CK(hipMemcpy(a.data(), src + b * bs, bs, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b2.data(), back + b * bs, bs, hipMemcpyDeviceToHost));
            if (memcmp(a.data(), b2.data(), bs)) dbad++;
        }
        printf("L%d %zu x %zu: DEcompress kernel %.3f ms  %.2f GB/s output  round trip %s\n", level, nb, bs, dms,
               (double)nb * bs / (dms * 1e-3) / 1e9, dbad ? "FAILED" : "ok");
        bad += dbad;
    }
    return bad ? 1 : 0;
}
