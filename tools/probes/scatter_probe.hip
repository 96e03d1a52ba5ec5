// scatter_probe.hip — bench tooling (not part of the library): how many scattered small stores / loads per second does an MI355X
// take?  Every lane of every wave writes (or reads) W bytes at a pseudo-random W-aligned address inside a region of R bytes, P
// accesses per lane, the whole chip busy (grid = CUs x 8, 256 threads).  Prints accesses per second for region sizes below and
// above the Infinity Cache (256 MiB).   hipcc -O3 --offload-arch=gfx950 scatter_probe.hip -o scatter_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W, bool READ>
__global__ __launch_bounds__(256) void k(uint8_t* base, uint64_t regionMask, uint32_t iters, uint32_t* sink)
{
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    // each WAVE works inside its own window of `regionMask + 1` bytes?  No: one shared region, like thousands of per-wave tables side by side
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        uint64_t off = (((uint64_t)x << 7) ^ (x >> 9)) & regionMask & ~(uint64_t)(W - 1);
        if (READ) {
            if (W == 4) acc += *(const uint32_t*)(base + off);
            else if (W == 16) { uint4 v = *(const uint4*)(base + off); acc += v.x + v.w; }
        } else {
            if (W == 4) *(uint32_t*)(base + off) = x;
            else if (W == 16) *(uint4*)(base + off) = make_uint4(x, x, x, x);
            else if (W == 32) { *(uint4*)(base + off) = make_uint4(x, x, x, x); *(uint4*)(base + off + 16) = make_uint4(x, x, x, x); }
        }
    }
    if (READ && acc == 0x12345678u) *sink = acc;
}

template <int W, bool READ>
static void run(uint8_t* base, size_t region, int cus, uint32_t iters, uint32_t* sink)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const dim3 g(cus * 8), t(256);
    hipLaunchKernelGGL((k<W, READ>), g, t, 0, 0, base, (uint64_t)region - 1, iters / 8, sink);      // warm-up
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<W, READ>), g, t, 0, 0, base, (uint64_t)region - 1, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double n = (double)cus * 8 * 256 * iters;
    printf("%-5s %2d B  region %6zu MiB: %7.2f G accesses/s  (%.1f ms)\n", READ ? "load" : "store", W, region >> 20, n / (ms * 1e-3) / 1e9, ms);
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint8_t* base; uint32_t* sink;
    const size_t maxRegion = (size_t)8 << 30;
    CK(hipMalloc((void**)&base, maxRegion)); CK(hipMalloc((void**)&sink, 4));
    CK(hipMemset(base, 0, maxRegion));
    const uint32_t iters = 4096;
    for (size_t region : { (size_t)16 << 20, (size_t)128 << 20, (size_t)512 << 20, (size_t)2 << 30, (size_t)8 << 30 }) {
        run<4, false>(base, region, cus, iters, sink);
        run<16, false>(base, region, cus, iters, sink);
        run<32, false>(base, region, cus, iters, sink);
        run<4, true>(base, region, cus, iters, sink);
        run<16, true>(base, region, cus, iters, sink);
    }
    return 0;
}
