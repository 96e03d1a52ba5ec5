// lizard_shard.h — multi-GPU side of the library (included by lizard_gpu.hip).
//
// Lizard blocks are independent, so a batch shards over GPUs by contiguous block ranges with no halo and no
// data-path collective; the one real exchange is the ALL-GATHER OF THE PER-BLOCK COMPRESSED SIZES (4 bytes per
// block) after which every rank computes the exclusive prefix sum = byte offset of every block in the concatenated
// output.  It runs on RCCL (rccl.h; over xGMI between the GPUs of a node) in both deployments:
//   * one process, several devices : LizardGPU_compressBlocks_sharded()  (ncclCommInitAll)
//   * one process per device       : LizardGPU_commUniqueId / _commInitRank / _gatherSizes_device — the torchrun form
//     bench.py uses: rank 0 creates the id, the launcher's own transport carries its 128 bytes to the other ranks.
// RCCL is loaded on first use (dlopen): single-GPU users of the library do not need it installed.
// Equal shard sizes use one ncclAllGather in place; ragged partitions (nBlocks not a multiple of the rank count)
// one ncclBroadcast per rank inside a group.  lz_scan_kernel (lz_pack.h) turns sizes into offsets on every device.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
pthread_mutex_t g_rccl_mu = PTHREAD_MUTEX_INITIALIZER;

// single-process communicators (one per device of the last device list) and the per-process rank communicator
ncclComm_t g_allComms[LZ_MAX_DEVICES];
int        g_allDevs[LZ_MAX_DEVICES];
int        g_allCount = 0;
ncclComm_t g_rankComm = nullptr;
int        g_rankCount = 0, g_rankIndex = -1;

int rccl_load()
{
    if (g_rccl.so) return 0;
    void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!so) { snprintf(t_err, sizeof t_err, "RCCL not found (dlopen librccl.so.1): %s", dlerror()); return -LIZARDGPU_ERR_RCCL; }
#define LZ_SYM(field, name) do { *(void**)(&g_rccl.field) = dlsym(so, name); \
        if (!g_rccl.field) { snprintf(t_err, sizeof t_err, "RCCL symbol %s missing", name); dlclose(so); return -LIZARDGPU_ERR_RCCL; } } while (0)
    LZ_SYM(GetUniqueId, "ncclGetUniqueId"); LZ_SYM(CommInitRank, "ncclCommInitRank"); LZ_SYM(CommInitAll, "ncclCommInitAll");
    LZ_SYM(CommDestroy, "ncclCommDestroy"); LZ_SYM(AllGather, "ncclAllGather"); LZ_SYM(Broadcast, "ncclBroadcast");
    LZ_SYM(GroupStart, "ncclGroupStart"); LZ_SYM(GroupEnd, "ncclGroupEnd"); LZ_SYM(GetErrorString, "ncclGetErrorString");
#undef LZ_SYM
    g_rccl.so = so;
    return 0;
}

#define LZ_NCCL(call)                                                                                  \
    do {                                                                                               \
        ncclResult_t r_ = (call);                                                                      \
        if (r_ != ncclSuccess) {                                                                       \
            snprintf(t_err, sizeof t_err, "%s failed: %s", #call, g_rccl.GetErrorString(r_));          \
            return -LIZARDGPU_ERR_RCCL;                                                                \
        }                                                                                              \
    } while (0)

void lz_shard_shutdown()
{
    pthread_mutex_lock(&g_rccl_mu);
    if (g_rccl.so) {
        for (int i = 0; i < g_allCount; i++) if (g_allComms[i]) (void)g_rccl.CommDestroy(g_allComms[i]);
        if (g_rankComm) (void)g_rccl.CommDestroy(g_rankComm);
    }
    g_allCount = 0; g_rankComm = nullptr; g_rankCount = 0; g_rankIndex = -1;
    pthread_mutex_unlock(&g_rccl_mu);
}

// Rank `rank`'s slice of the all-sizes array is already in place at all + first(rank); afterwards every rank holds
// all of it.  Called once per rank with that rank's communicator and stream (inside a group when one process
// drives several ranks).
int gather_in_place(ncclComm_t comm, int nRanks, size_t nBlocks, u32* all, hipStream_t stream)
{
    if (nBlocks % (size_t)nRanks == 0) {
        const size_t per = nBlocks / (size_t)nRanks;
        // in-place form: sendbuff = recvbuff + rank * sendcount (rccl.h, ncclAllGather)
        int rank = -1;
        for (int r = 0; r < g_allCount; r++) if (g_allComms[r] == comm) rank = r;
        if (comm == g_rankComm) rank = g_rankIndex;
        LZ_NCCL(g_rccl.AllGather(all + (size_t)rank * per, all, per, ncclUint32, comm, stream));
    } else {
        for (int root = 0; root < nRanks; root++) {
            size_t first, count;
            LizardGPU_shardRange(nBlocks, root, nRanks, &first, &count);
            if (count) LZ_NCCL(g_rccl.Broadcast(all + first, all + first, count, ncclUint32, root, comm, stream));
        }
    }
    return 0;
}

}  // namespace

extern "C" {

void LizardGPU_shardRange(size_t nBlocks, int rank, int nRanks, size_t* first, size_t* count)
{
    const size_t base = nBlocks / (size_t)nRanks, rem = nBlocks % (size_t)nRanks, r = (size_t)rank;
    if (first) *first = r * base + (r < rem ? r : rem);
    if (count) *count = base + (r < rem ? 1 : 0);
}

void LizardGPU_offsetsFromSizes(const uint32_t* sizes, size_t nBlocks, uint64_t* offsets)
{
    uint64_t run = 0;
    for (size_t i = 0; i < nBlocks; i++) { offsets[i] = run; run += sizes[i]; }
    offsets[nBlocks] = run;
}

int LizardGPU_compressBlocks_sharded(int nDevices, const int* devices, const void* const* d_src, size_t nBlocks,
                                     size_t blockSize, size_t lastBlockSize, void* const* d_dst, size_t dstStride,
                                     uint32_t* const* d_allSizes, uint64_t* const* d_offsets, int level)
{
    t_err[0] = 0;
    if (nDevices < 1 || nDevices > LZ_MAX_DEVICES || !d_src || !d_dst || !d_allSizes || !d_offsets || nBlocks < (size_t)nDevices
        || nBlocks > 0xFFFFFFFFu) {
        snprintf(t_err, sizeof t_err, "bad argument (device count, null array or fewer blocks than devices)");
        return -LIZARDGPU_ERR_ARG;
    }
    int devs[LZ_MAX_DEVICES];
    for (int r = 0; r < nDevices; r++) devs[r] = devices ? devices[r] : r;
    pthread_mutex_lock(&g_rccl_mu);
    int rc = rccl_load();
    if (!rc) {
        bool same = g_allCount == nDevices;
        for (int r = 0; same && r < nDevices; r++) same = g_allDevs[r] == devs[r];
        if (!same) {
            for (int i = 0; i < g_allCount; i++) if (g_allComms[i]) (void)g_rccl.CommDestroy(g_allComms[i]);
            g_allCount = 0;
            ncclResult_t r_ = g_rccl.CommInitAll(g_allComms, nDevices, devs);
            if (r_ != ncclSuccess) { snprintf(t_err, sizeof t_err, "ncclCommInitAll failed: %s", g_rccl.GetErrorString(r_)); rc = -LIZARDGPU_ERR_RCCL; }
            else { g_allCount = nDevices; memcpy(g_allDevs, devs, sizeof(int) * (size_t)nDevices); }
        }
    }
    pthread_mutex_unlock(&g_rccl_mu);
    if (rc) return rc;
    const int savedSel = t_device;
    hipStream_t streams[LZ_MAX_DEVICES];
    // 1. every device compresses its contiguous range; its sizes land in place inside its copy of the all-sizes array
    for (int r = 0; r < nDevices && !rc; r++) {
        size_t first, count;
        LizardGPU_shardRange(nBlocks, r, nDevices, &first, &count);
        t_device = devs[r];
        Guard g;
        if (g.rc) { rc = g.rc; break; }
        if ((rc = ctx_init(*g.c))) break;
        streams[r] = g.c->stage[0].stream;
        g.c->hostKernelMs = -1.0f;
        rc = launch(*g.c, d_src[r], count, blockSize, r == nDevices - 1 ? lastBlockSize : blockSize, d_dst[r], dstStride,
                    d_allSizes[r] + first, level, streams[r]);
    }
    // 2. the exchange: RCCL all-gather of the sizes, one rank per device, grouped
    if (!rc) do {
        if (g_rccl.GroupStart() != ncclSuccess) { snprintf(t_err, sizeof t_err, "ncclGroupStart failed"); rc = -LIZARDGPU_ERR_RCCL; break; }
        for (int r = 0; r < nDevices && !rc; r++) {
            if (hipSetDevice(devs[r]) != hipSuccess) { rc = -LIZARDGPU_ERR_HIP; break; }
            rc = gather_in_place(g_allComms[r], nDevices, nBlocks, d_allSizes[r], streams[r]);
        }
        if (g_rccl.GroupEnd() != ncclSuccess && !rc) { snprintf(t_err, sizeof t_err, "ncclGroupEnd failed"); rc = -LIZARDGPU_ERR_RCCL; }
    } while (0);
    // 3. every device: sizes -> global output offsets
    for (int r = 0; r < nDevices && !rc; r++) {
        if (hipSetDevice(devs[r]) != hipSuccess) { rc = -LIZARDGPU_ERR_HIP; break; }
        hipLaunchKernelGGL(lz_scan_kernel, dim3(1), dim3(1024), 0, streams[r], (const u32*)d_allSizes[r], (u64*)d_offsets[r], (u32)nBlocks, 0u, 0u, LZ_PACK_PAYLOAD);
        if (hipGetLastError() != hipSuccess) { snprintf(t_err, sizeof t_err, "offset scan launch failed on device %d", devs[r]); rc = -LIZARDGPU_ERR_HIP; }
    }
    for (int r = 0; r < nDevices; r++) {
        if (hipSetDevice(devs[r]) == hipSuccess && hipStreamSynchronize(streams[r]) != hipSuccess && !rc) {
            snprintf(t_err, sizeof t_err, "device %d: stream synchronise failed", devs[r]); rc = -LIZARDGPU_ERR_HIP;
        }
    }
    t_device = savedSel;
    return rc;
}

int LizardGPU_commUniqueId(void* id128)
{
    t_err[0] = 0;
    pthread_mutex_lock(&g_rccl_mu);
    int rc = rccl_load();
    if (!rc && g_rccl.GetUniqueId((ncclUniqueId*)id128) != ncclSuccess) { snprintf(t_err, sizeof t_err, "ncclGetUniqueId failed"); rc = -LIZARDGPU_ERR_RCCL; }
    pthread_mutex_unlock(&g_rccl_mu);
    return rc;
}

int LizardGPU_commInitRank(const void* id128, int nRanks, int rank)
{
    Guard g;                                                    // the communicator lives on the selected device
    if (g.rc) return g.rc;
    if (!id128 || nRanks < 1 || rank < 0 || rank >= nRanks) { snprintf(t_err, sizeof t_err, "bad argument"); return -LIZARDGPU_ERR_ARG; }
    pthread_mutex_lock(&g_rccl_mu);
    int rc = rccl_load();
    if (!rc) {
        if (g_rankComm) { (void)g_rccl.CommDestroy(g_rankComm); g_rankComm = nullptr; }
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        ncclResult_t r_ = g_rccl.CommInitRank(&g_rankComm, nRanks, id, rank);
        if (r_ != ncclSuccess) { snprintf(t_err, sizeof t_err, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r_)); g_rankComm = nullptr; rc = -LIZARDGPU_ERR_RCCL; }
        else { g_rankCount = nRanks; g_rankIndex = rank; }
    }
    pthread_mutex_unlock(&g_rccl_mu);
    return rc;
}

int LizardGPU_gatherSizes_device(const uint32_t* d_localSizes, size_t nBlocks, uint32_t* d_allSizes, uint64_t* d_offsets, void* stream)
{
    Guard g;
    if (g.rc) return g.rc;
    if (!g_rankComm) { snprintf(t_err, sizeof t_err, "LizardGPU_commInitRank has not been called"); return -LIZARDGPU_ERR_ARG; }
    if (!d_localSizes || !d_allSizes || !d_offsets || nBlocks < (size_t)g_rankCount || nBlocks > 0xFFFFFFFFu) {
        snprintf(t_err, sizeof t_err, "bad argument"); return -LIZARDGPU_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    size_t first, count;
    LizardGPU_shardRange(nBlocks, g_rankIndex, g_rankCount, &first, &count);
    if (d_localSizes != d_allSizes + first)
        LZ_HIP(hipMemcpyAsync(d_allSizes + first, d_localSizes, count * sizeof(u32), hipMemcpyDeviceToDevice, s));
    int rc = gather_in_place(g_rankComm, g_rankCount, nBlocks, d_allSizes, s);
    if (rc) return rc;
    hipLaunchKernelGGL(lz_scan_kernel, dim3(1), dim3(1024), 0, s, (const u32*)d_allSizes, (u64*)d_offsets, (u32)nBlocks, 0u, 0u, LZ_PACK_PAYLOAD);
    LZ_HIP(hipGetLastError());
    return 0;
}

int LizardGPU_commDestroy(void)
{
    pthread_mutex_lock(&g_rccl_mu);
    if (g_rccl.so && g_rankComm) (void)g_rccl.CommDestroy(g_rankComm);
    g_rankComm = nullptr; g_rankCount = 0; g_rankIndex = -1;
    pthread_mutex_unlock(&g_rccl_mu);
    return 0;
}

}  // extern "C"
