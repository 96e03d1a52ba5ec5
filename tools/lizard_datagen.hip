// tools/lizard_datagen.hip — BENCH / TEST TOOLING, not part of the product library (liblizard_amd.so does not contain it).
//
// The workload generator of bench.py, tests/gpu_bench and the GPU tests: block b of a batch is
// RDG_genBuffer(blockSize, P, litP, seed0 + b) of the reference's benchmark generator (programs/datagen.c:153; that file is
// GPL-2 in the reference, which is one reason this restatement of its behaviour lives with the tooling and not in the library).
// One generator call is one serial LCG walk, so the device version runs one thread per block.  Built in-tree as
// tools/liblizard_datagen.so by tools/Makefile (and __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lz_datagen.h"

__global__ __launch_bounds__(64) void lz_datagen_kernel(uint8_t* dst, uint64_t nBlocks, uint64_t blockSize, uint32_t matchProba32,
                                                        int zeroRuns, const uint8_t* lt, uint32_t seed0)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nBlocks) lz_rdg_fill(dst + b * blockSize, (size_t)blockSize, matchProba32, zeroRuns, lt, seed0 + (uint32_t)b);
}

extern "C" {

// fills buffer[0..size) exactly like RDG_genBuffer(buffer, size, matchProba, litProba, seed)
void LizardTools_datagen_host(void* buffer, size_t size, double matchProba, double litProba, unsigned seed)
{
    uint8_t lt[LZ_RDG_LTSIZE];
    lz_rdg_table(lt, matchProba, litProba);
    lz_rdg_fill((uint8_t*)buffer, size, (uint32_t)(32768 * matchProba), matchProba >= 1.0, lt, seed);
}

// block b (b < nBlocks, blockSize bytes each, back to back at d_dst on the CURRENT device) = RDG_genBuffer(blockSize, ..., seed0 + b);
// synchronous.  0, or the hipError_t that stopped it.
int LizardTools_datagen_device(void* d_dst, size_t nBlocks, size_t blockSize, double matchProba, double litProba, unsigned seed0, void* stream)
{
    if (!d_dst || nBlocks == 0 || blockSize == 0) return (int)hipErrorInvalidValue;
    uint8_t lt[LZ_RDG_LTSIZE];
    lz_rdg_table(lt, matchProba, litProba);
    uint8_t* d_lt = nullptr;
    hipError_t e = hipMalloc((void**)&d_lt, LZ_RDG_LTSIZE);
    if (e != hipSuccess) return (int)e;
    hipStream_t s = (hipStream_t)stream;
    e = hipMemcpyAsync(d_lt, lt, LZ_RDG_LTSIZE, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);                       // `lt` is a stack buffer
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lz_datagen_kernel, dim3((unsigned)((nBlocks + 63) / 64)), dim3(64), 0, s, (uint8_t*)d_dst, (uint64_t)nBlocks,
                           (uint64_t)blockSize, (uint32_t)(32768 * matchProba), (int)(matchProba >= 1.0), (const uint8_t*)d_lt, (uint32_t)seed0);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_lt);
    return (int)e;
}

}  // extern "C"
