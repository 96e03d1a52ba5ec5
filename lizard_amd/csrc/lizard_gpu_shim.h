/* lizard_gpu_shim.h — private C-ABI seam between the host C layer (lizard_host.c, lizard_frame_host.c) and the HIP
 * side (lizard_gpu.hip). Not installed; the public surface is include/lizard_amd.h. */
#ifndef LIZARD_GPU_SHIM_H
#define LIZARD_GPU_SHIM_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* one host block -> compressed size; 0 = does not fit in maxDstSize; < 0 = -LIZARDGPU_ERR_* */
int lzgpu_compress_one(const void* src, int srcSize, void* dst, int maxDstSize, int level);
/* frame block records (LE32 size word | raw flag, payload) of nBlocks independent blocks, packed into dst;
 * built on the device, one D2H per chunk.  0 or -LIZARDGPU_ERR_*. */
int lzgpu_frame_records(const void* src, size_t nBlocks, size_t blockSize, size_t lastBlockSize, void* dst, size_t dstCapacity,
                        size_t* written, int level);
/* a call degraded (returned 0 / stored its blocks raw) because the GPU path could not do it: counted, and reported on stderr for
 * the 1st, 2nd, 4th, 8th ... occurrence (lizard_host.c) */
void lzgpu_note_degraded(const char* what, int level);
#ifdef __cplusplus
}
#endif
#endif
