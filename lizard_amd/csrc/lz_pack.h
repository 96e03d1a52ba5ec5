// lz_pack.h — device-side compaction of a batch's compressed blocks (gfx950).
//
// The block kernels write every block into a bound-sized slot (Lizard_compressBound(blockSize), SURVEY §7 "variable-
// length output").  Before anything crosses PCIe the valid bytes are packed back to back:
//   lz_scan_kernel    exclusive prefix sum of the per-block record sizes -> byte offset of every record, total
//   lz_gather_kernel  one workgroup per block copies its record to packed + offset, 16 bytes per lane
// Record = the compressed block itself (LZ_PACK_PAYLOAD), or the frame layer's block record (LZ_PACK_FRAME): LE32 size
// word followed by the compressed block, or — when the block did not shrink below its input (the frame layer passes
// maxDstSize = srcSize-1 and stores the input itself on a 0 return, reference lib/lizard_frame.c:461-467) — by the raw
// input with bit 31 of the size word set.  A 1-byte block is always "compressed" (6 bytes): the reference's room test
// wraps at maxDstSize 0 (lizard_compress.c:238), see lzgpu_compress_one.
#pragma once
#include "lz_wave.h"

#define LZ_PACK_PAYLOAD 0
#define LZ_PACK_FRAME   1

LZ_DEV bool lz_frame_stored_raw(u32 n, u32 cs) { return n != 1u && (cs == 0u || cs > n - 1u); }
LZ_DEV u32 lz_record_bytes(u32 n, u32 cs, int mode)
{
    if (mode == LZ_PACK_PAYLOAD) return cs;
    return 4u + (lz_frame_stored_raw(n, cs) ? n : cs);
}

// offsets[i] = sum of record sizes of blocks < i, offsets[nBlocks] = total.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void lz_scan_kernel(const u32* sizes, u64* offsets, u32 nBlocks, u32 blockSize, u32 lastBlockSize, int mode)
{
    __shared__ u64 part[1024];
    const u32 t = threadIdx.x;
    const u32 per = (nBlocks + 1023u) / 1024u;
    const u32 lo = t * per < nBlocks ? t * per : nBlocks, hi = lo + per < nBlocks ? lo + per : nBlocks;
    u64 sum = 0;
    for (u32 i = lo; i < hi; i++) sum += lz_record_bytes(i == nBlocks - 1u ? lastBlockSize : blockSize, sizes[i], mode);
    part[t] = sum;
    __syncthreads();
    for (u32 d = 1; d < 1024u; d <<= 1) {                       // Hillis-Steele inclusive scan over the per-thread sums
        const u64 v = t >= d ? part[t - d] : 0ull;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    u64 run = part[t] - sum;
    for (u32 i = lo; i < hi; i++) {
        offsets[i] = run;
        run += lz_record_bytes(i == nBlocks - 1u ? lastBlockSize : blockSize, sizes[i], mode);
    }
    if (t == 1023u) offsets[nBlocks] = part[1023];
}

__global__ __launch_bounds__(256) void lz_gather_kernel(const u8* src, const u8* slots, u64 slotStride, const u32* sizes, const u64* offsets,
                                                        u8* packed, u32 nBlocks, u32 blockSize, u32 lastBlockSize, int mode)
{
    const u32 b = blockIdx.x;
    const u32 n = b == nBlocks - 1u ? lastBlockSize : blockSize;
    const u32 cs = sizes[b];
    u8* out = packed + offsets[b];
    const u8* from = slots + (u64)b * slotStride;
    u32 len = cs;
    if (mode == LZ_PACK_FRAME) {
        const bool raw = lz_frame_stored_raw(n, cs);
        if (threadIdx.x == 0) {
            const u32 word = raw ? (n | 0x80000000u) : cs;
            out[0] = (u8)word; out[1] = (u8)(word >> 8); out[2] = (u8)(word >> 16); out[3] = (u8)(word >> 24);
        }
        out += 4;
        if (raw) { from = src + (u64)b * blockSize; len = n; }
    }
    const u32 bulk = len & ~15u;
    for (u32 i = threadIdx.x * 16u; i < bulk; i += 256u * 16u)
        lz_st128(out + i, lz_ld128(from + i));
    for (u32 i = bulk + threadIdx.x; i < len; i += 256u) out[i] = from[i];
}

static inline void lz_pack_launch(const u8* d_src, const u8* d_slots, size_t slotStride, const u32* d_sizes, u64* d_offsets, u8* d_packed,
                                  u32 nBlocks, u32 blockSize, u32 lastBlockSize, int mode, hipStream_t stream)
{
    hipLaunchKernelGGL(lz_scan_kernel, dim3(1), dim3(1024), 0, stream, d_sizes, d_offsets, nBlocks, blockSize, lastBlockSize, mode);
    hipLaunchKernelGGL(lz_gather_kernel, dim3(nBlocks), dim3(256), 0, stream, d_src, d_slots, (u64)slotStride, d_sizes, (const u64*)d_offsets,
                       d_packed, nBlocks, blockSize, lastBlockSize, mode);
}
