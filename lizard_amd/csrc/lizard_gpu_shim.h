/* lizard_gpu_shim.h — private C-ABI seam between the host C layer (lizard_host.c) and the HIP side
 * (lizard_gpu.hip). Not installed; the public surface is include/lizard_amd.h. */
#ifndef LIZARD_GPU_SHIM_H
#define LIZARD_GPU_SHIM_H
#ifdef __cplusplus
extern "C" {
#endif
/* one host block -> compressed size; 0 = does not fit in maxDstSize; < 0 = -LIZARDGPU_ERR_* */
int lzgpu_compress_one(const void* src, int srcSize, void* dst, int maxDstSize, int level);
#ifdef __cplusplus
}
#endif
#endif
