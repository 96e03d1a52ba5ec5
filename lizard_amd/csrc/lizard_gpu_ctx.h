/* lizard_gpu_ctx.h — private seam between the host C layer and the HIP side (not installed; the public surface is
 * include/lizard_amd.h).  Plain C: the per-device context the two sides share, and the thin shim the host-buffer pipeline
 * (lizard_pipeline_host.c, C) calls into lizard_gpu.hip (kernel launches, device context).  Everything else the pipeline
 * needs from HIP is the runtime's own C API (hip_runtime_api.h). */
#ifndef LIZARD_GPU_CTX_H
#define LIZARD_GPU_CTX_H
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>

#define LZ_STAGES 3                                         /* chunks in flight in the host-buffer pipeline */
#define LZ_MAX_DEVICES 16

/* One stage of the host-buffer pipeline: pinned staging on the host side, input / slot / packed buffers on the device side,
 * its own stream.  The stages alternate so that the copies of one chunk overlap the kernels of another. */
typedef struct LzStage {
    hipStream_t stream;
    hipEvent_t  k0, k1, meta, done, up;                     /* up: the chunk's input is on the device */
    uint8_t*  h_in;     size_t h_in_cap;                    /* pinned */
    uint8_t*  h_out;    size_t h_out_cap;                   /* pinned */
    uint32_t* h_sizes;  uint64_t* h_offsets;  size_t h_meta_cap;   /* pinned, nBlocks (+1) */
    uint8_t*  d_in;     size_t d_in_cap;
    uint8_t*  d_slots;  size_t d_slots_cap;
    uint8_t*  d_packed; size_t d_packed_cap;
    uint32_t* d_sizes;  uint64_t* d_offsets;  size_t d_meta_cap;
} LzStage;

/* The combiner of the one-block entry points (Lizard_compress & co, lizard_pipeline_host.c): callers that arrive while a batch
 * is in flight queue up and leave together in the next launch.  Own staging and stream: the members of a batch copy their input
 * into h_in and their output out of h_out themselves, outside the context lock. */
struct LzOneJob;
typedef struct LzCombine {
    pthread_mutex_t mu;
    pthread_cond_t  cv;
    struct LzOneJob *head, *tail;       /* callers waiting for a batch */
    int   busy;                         /* a batch is under way: from the moment its leader takes it until its last member has copied out */
    int   pendingIn, pendingOut;        /* members of the current batch that still have to copy in / out */
    int   queued, lastN, collecting;    /* callers in the queue; members of the previous batch; a leader is waiting for stragglers */
    LzStage st;                         /* staging of the current batch */
    uint32_t* d_srcSizes; uint64_t* d_srcOffsets; uint32_t* h_srcSizes; uint64_t* h_srcOffsets; size_t raggedCap;
    unsigned long long batches, jobs;   /* statistics (LizardGPU_combinerStats) */
    double tLock, tCopyIn, tGpu, tOut;  /* seconds spent by leaders: waiting for the context, members' copy-in, GPU part, until the last member left */
    double tBusySince;
} LzCombine;

/* A scratch arena with its block counter and the per-wave tables of the levels that keep them in global memory.  Every launch
 * needs one to itself.  The context's own (LzCtx::scratch, counter, tables, pfTables, ev0/ev1) serves every launch that fills
 * the machine, the hashChain levels and decompression, one after the other; a compress launch SMALLER than the machine that
 * arrives on another stream while that one is busy gets one of up to LZ_ARENAS_MAX - 1 more (allocated on first need, 2.6 GiB +
 * tables each; LIZARDGPU_ARENAS=1..4 caps the total, default 4), so that small launches of different streams run side by side
 * on the CUs they leave each other. */
#define LZ_ARENAS_MAX 4
typedef struct LzArena {
    uint8_t* scratch; uint32_t* counter; uint8_t* tables; uint8_t* pfTables;
    size_t tablesSlots, pfSlots;  /* table slots behind tables / pfTables (fewer than resident waves under a memory budget) */
    hipEvent_t ev0, ev1;        /* around its last launch */
    hipStream_t lastStream;
    int timed;                  /* ev1 was recorded */
} LzArena;

typedef struct LzCtx {
    int   ready;
    int   device;
    int   cus;
    uint8_t* tables;            /* levels 11/31/22/42, allocated on first use */
    uint8_t* pfTables;          /* levels 21/41: 64 KiB per resident wave for the waves whose table is not in LDS */
    uint8_t* hcSlots;           /* hashChain levels, allocated on first use / when a larger block size arrives */
    size_t   hcMaxBlock, hcNSlots, hcSlotBytes;
    int      hcHasBest;         /* the slots end with the first-search table of levels 16/17/37/38 */
    size_t   tablesSlots, pfSlots;   /* table slots behind tables / pfTables (fewer than resident waves under a memory budget) */
    size_t   devBytes;          /* device memory this context holds in its large buffers (arenas, tables, work areas, staging): what
                                 * LizardGPU_setMemoryBudget bounds and LizardGPU_memoryInUse reports */
    int      idleLaunches;      /* launches on the context's own arena since an extra arena was last used (they are released after 64) */
    uint8_t* scratch;
    uint32_t* counter;
    hipEvent_t ev0, ev1;
    int   timed;
    hipStream_t lastStream;     /* of the last launch on the context's own arena */
    LzArena extra[LZ_ARENAS_MAX - 1];
    int   nExtra, maxArenas, nextExtra;
    hipEvent_t lastEv0, lastEv1;   /* around the most recent launch, whichever arena it used (LizardGPU_lastKernelMs) */
    int   lastSplit;            /* the last compress launch was the producer / consumer form (profile builds: where the records are) */
    int   laneOrderOk;          /* self-check at context creation: lanes of one DS atomic are served in lane order */
    float hostKernelMs;         /* sum over the chunks of the last host-buffer call (< 0: last call was a device call) */
    LzStage stage[LZ_STAGES];
    LzCombine comb;
    pthread_mutex_t mu;
} LzCtx;

/* Locks the selected device's context and makes that device current for the calling thread (HIP's current device is per
 * thread); release restores the caller's device.  rc != 0: nothing is held (the error text is set). */
typedef struct LzGuard { LzCtx* c; int saved; int rc; } LzGuard;

#ifdef __cplusplus
extern "C" {
#endif
void  lzk_guard_acquire(LzGuard* g);
void  lzk_guard_release(LzGuard* g);
char* lzk_err(void);                                        /* the calling thread's error text, LZK_ERR_BYTES bytes */
#define LZK_ERR_BYTES 256
int   lzk_ctx_init(LzCtx* c);
/* device memory of the context's large buffers: counted against the memory budget (-LIZARDGPU_ERR_NOMEM when it does not fit) */
int   lzk_dev_alloc(LzCtx* c, void** p, size_t bytes);
void  lzk_dev_free(LzCtx* c, void* p, size_t bytes);
size_t lzk_budget(void);                                    /* 0 = none */
size_t lzk_budget_room_for_staging(const LzCtx* c);         /* (size_t)-1 = no budget; else what the budget leaves beside the context's scratch arena */
int   lzk_clamp_level(int level);
/* the block kernels over nBlocks blocks resident at d_src (launcher of LizardGPU_compressBlocks_device); k0 / k1 (may be NULL)
 * are recorded around the kernel; d_srcSizes / d_srcOffsets (may be NULL): a ragged batch, block b = d_srcSizes[b] bytes at d_src + d_srcOffsets[b] */
int   lzk_launch(LzCtx* c, const void* d_src, size_t nBlocks, size_t blockSize, size_t lastBlockSize, void* d_dst, size_t dstStride,
                 uint32_t* d_sizes, int level, hipStream_t stream, hipEvent_t k0, hipEvent_t k1, const uint32_t* d_srcSizes,
                 const uint64_t* d_srcOffsets);
/* the calling thread's selected device's context WITHOUT locking it (NULL: no device, the error text is set) */
LzCtx* lzk_ctx_peek(void);
/* lizard_pipeline_host.c: release the combiner's buffers (context locked, no batch under way); keep batches out during a shutdown */
void  lzk_combiner_free(LzCtx* c);
void  lzk_combiner_quiesce(LzCtx* c);
void  lzk_combiner_resume(LzCtx* c);
int   lzk_launch_decompress(LzCtx* c, const void* d_src, const uint64_t* d_offsets, size_t srcStride, const uint32_t* d_srcSizes,
                            size_t nBlocks, void* d_dst, size_t dstStride, uint32_t* d_outSizes, hipStream_t stream);
/* exclusive scan of the record sizes + compaction of the valid bytes into d_packed (lz_pack.h); mode: LZK_PACK_* */
void  lzk_pack_launch(const void* d_in, const void* d_slots, size_t slot, const uint32_t* d_sizes, uint64_t* d_offsets, void* d_packed,
                      uint32_t nb, uint32_t blockSize, uint32_t lastBlockSize, int mode, hipStream_t stream);
#define LZK_PACK_PAYLOAD 0
#define LZK_PACK_FRAME   1
#ifdef __cplusplus
}
#endif
#endif
