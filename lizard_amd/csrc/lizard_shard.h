// lizard_shard.h — multi-GPU side of the library (included by lizard_gpu.hip).
//
// Lizard blocks are independent, so a batch shards over GPUs by contiguous block ranges with no halo and no
// data-path collective; the one real exchange is the ALL-GATHER OF THE PER-BLOCK COMPRESSED SIZES (4 bytes per
// block) after which every rank computes the exclusive prefix sum = byte offset of every block in the concatenated
// output.  It runs on RCCL (rccl.h; over xGMI between the GPUs of a node) in both deployments:
//   * one process, several devices : LizardGPU_compressBlocks_sharded()  (ncclCommInitAll)
//   * one process per device       : LizardGPU_commUniqueId / _commInitRank / _gatherSizes_device — the torchrun form
//     bench.py uses: rank 0 creates the id, the launcher's own transport carries its 128 bytes to the other ranks.
// RCCL is resolved on first use: a copy the process has already mapped (torch's, in a torchrun job) is taken first
// (dlopen RTLD_NOLOAD) so that a job never carries two RCCL instances on the same GPUs; only then is librccl.so.1 loaded.
// Single-GPU users of the library do not need it installed.  The exchange itself (partition, in-place all-gather or ragged
// broadcasts, offsets) lives in lizard_shard_core.h, written against a small collective table: RCCL fills it here,
// LizardGPU_setCollectives swaps in another transport, and tests/shard_fake.cpp runs the same code with 2 and 3 ranks on a CPU.
// Equal shard sizes use one ncclAllGather in place; ragged partitions (nBlocks not a multiple of the rank count)
// one ncclBroadcast per rank inside a group.  lz_scan_kernel (lz_pack.h) turns sizes into offsets on every device.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "lizard_shard_core.h"

namespace {

struct Rccl {
    void* so = nullptr;
    bool  shared = false;                                  // resolved from a copy another library had already mapped
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;       // optional
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // optional: what the communicator itself says it spans
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;   // optional
    ncclResult_t (*GetVersion)(int*) = nullptr;            // optional
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
pthread_mutex_t g_rccl_mu = PTHREAD_MUTEX_INITIALIZER;     // held for the whole of every call that touches a communicator

// single-process communicators (one per device of the last device list) and the per-process rank communicator
ncclComm_t g_allComms[LZ_MAX_DEVICES];
int        g_allDevs[LZ_MAX_DEVICES];
int        g_allCount = 0;
ncclComm_t g_rankComm = nullptr;
int        g_rankCount = 0, g_rankIndex = -1;
int        g_rankSeen = 0, g_rankSeenIndex = -1;           // what the communicator reports after its creation (ncclCommCount / ncclCommUserRank)
int        g_allSeen = 0;                                  // ncclCommCount of the single-process communicators (0 = none / not reported)
// the transport of the size exchange: RCCL unless LizardGPU_setCollectives installed another one
LzCollectives g_userCol;
bool          g_haveUserCol = false;

int rccl_load()
{
    if (g_rccl.so) return 0;
    // a copy that is already in the process first (torchrun: torch/lib/librccl.so, soname librccl.so.1)
    void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    const bool shared = so != nullptr;
    if (!so) so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!so) { snprintf(t_err, sizeof t_err, "RCCL not found (dlopen librccl.so.1): %s", dlerror()); return -LIZARDGPU_ERR_RCCL; }
#define LZ_SYM(field, name) do { *(void**)(&g_rccl.field) = dlsym(so, name); \
        if (!g_rccl.field) { snprintf(t_err, sizeof t_err, "RCCL symbol %s missing", name); dlclose(so); return -LIZARDGPU_ERR_RCCL; } } while (0)
    LZ_SYM(GetUniqueId, "ncclGetUniqueId"); LZ_SYM(CommInitRank, "ncclCommInitRank"); LZ_SYM(CommInitAll, "ncclCommInitAll");
    LZ_SYM(CommDestroy, "ncclCommDestroy"); LZ_SYM(AllGather, "ncclAllGather"); LZ_SYM(Broadcast, "ncclBroadcast");
    LZ_SYM(GroupStart, "ncclGroupStart"); LZ_SYM(GroupEnd, "ncclGroupEnd"); LZ_SYM(GetErrorString, "ncclGetErrorString");
#undef LZ_SYM
    *(void**)(&g_rccl.CommAbort) = dlsym(so, "ncclCommAbort");
    *(void**)(&g_rccl.CommCount) = dlsym(so, "ncclCommCount");
    *(void**)(&g_rccl.CommUserRank) = dlsym(so, "ncclCommUserRank");
    *(void**)(&g_rccl.GetVersion) = dlsym(so, "ncclGetVersion");
    g_rccl.so = so; g_rccl.shared = shared;
    return 0;
}

int nccl_fail(const char* what, ncclResult_t r)
{
    snprintf(t_err, sizeof t_err, "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return -LIZARDGPU_ERR_RCCL;
}
// Ask a fresh communicator what it spans: a job that believes it runs N ranks over RCCL while the communicator holds another
// number must not report a scaling figure.  0 and *count / *rank filled (-1 where the library does not say), or an error.
int rccl_verify(ncclComm_t comm, int wantCount, int wantRank, int* count, int* rank)
{
    *count = -1; *rank = -1;
    if (g_rccl.CommCount) { const ncclResult_t r = g_rccl.CommCount(comm, count); if (r != ncclSuccess) return nccl_fail("ncclCommCount", r); }
    if (g_rccl.CommUserRank) { const ncclResult_t r = g_rccl.CommUserRank(comm, rank); if (r != ncclSuccess) return nccl_fail("ncclCommUserRank", r); }
    if ((*count >= 0 && *count != wantCount) || (*rank >= 0 && *rank != wantRank)) {
        snprintf(t_err, sizeof t_err, "RCCL communicator reports rank %d of %d, expected rank %d of %d", *rank, *count, wantRank, wantCount);
        return -LIZARDGPU_ERR_RCCL;
    }
    return 0;
}
// RCCL behind the collective table of lizard_shard_core.h
int rccl_all_gather(const void* send, void* recv, size_t count, void* comm, void* stream)
{
    const ncclResult_t r = g_rccl.AllGather(send, recv, count, ncclUint32, (ncclComm_t)comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : nccl_fail("ncclAllGather", r);
}
int rccl_broadcast(const void* send, void* recv, size_t count, int root, void* comm, void* stream)
{
    const ncclResult_t r = g_rccl.Broadcast(send, recv, count, ncclUint32, root, (ncclComm_t)comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : nccl_fail("ncclBroadcast", r);
}
int rccl_group_start() { const ncclResult_t r = g_rccl.GroupStart(); return r == ncclSuccess ? 0 : nccl_fail("ncclGroupStart", r); }
int rccl_group_end()   { const ncclResult_t r = g_rccl.GroupEnd();   return r == ncclSuccess ? 0 : nccl_fail("ncclGroupEnd", r); }
LzCollectives collectives()
{
    if (g_haveUserCol) return g_userCol;
    LzCollectives c; c.allGather = rccl_all_gather; c.broadcast = rccl_broadcast; c.groupStart = rccl_group_start; c.groupEnd = rccl_group_end;
    return c;
}
// the device-side steps around the exchange
int hip_copy_u32(uint32_t* dst, const uint32_t* src, size_t count, void* stream)
{
    LZ_HIP(hipMemcpyAsync(dst, src, count * sizeof(u32), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
int hip_scan_sizes(const uint32_t* sizes, uint64_t* offsets, size_t nBlocks, void* stream)
{
    hipLaunchKernelGGL(lz_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const u32*)sizes, (u64*)offsets, (u32)nBlocks, 0u, 0u, LZ_PACK_PAYLOAD);
    LZ_HIP(hipGetLastError());
    return 0;
}
const LzDeviceOps kHipOps = { hip_copy_u32, hip_scan_sizes };

thread_local const int* t_shardDevs = nullptr;             // device list of the sharded call in progress on this thread
int select_shard_device(int rank)
{
    if (hipSetDevice(t_shardDevs[rank]) != hipSuccess) { snprintf(t_err, sizeof t_err, "hipSetDevice(%d) failed", t_shardDevs[rank]); return -LIZARDGPU_ERR_HIP; }
    return 0;
}

// A collective failed inside a group: part of the ranks have posted, the others never will.  Synchronising their streams could
// wait for ever; the communicators are destroyed instead (the next call makes new ones).  RCCL only; called with g_rccl_mu held.
void abort_all_comms()
{
    if (!g_rccl.so) return;
    for (int i = 0; i < g_allCount; i++) if (g_allComms[i]) { if (g_rccl.CommAbort) (void)g_rccl.CommAbort(g_allComms[i]); else (void)g_rccl.CommDestroy(g_allComms[i]); g_allComms[i] = nullptr; }
    g_allCount = 0;
}

void lz_shard_shutdown()
{
    pthread_mutex_lock(&g_rccl_mu);
    if (g_rccl.so) {
        for (int i = 0; i < g_allCount; i++) if (g_allComms[i]) (void)g_rccl.CommDestroy(g_allComms[i]);
        if (g_rankComm) (void)g_rccl.CommDestroy(g_rankComm);
    }
    g_allCount = 0; g_rankComm = nullptr; g_rankCount = 0; g_rankIndex = -1; g_rankSeen = 0; g_rankSeenIndex = -1; g_allSeen = 0;
    pthread_mutex_unlock(&g_rccl_mu);
}

}  // namespace

extern "C" {

void LizardGPU_shardRange(size_t nBlocks, int rank, int nRanks, size_t* first, size_t* count)
{
    lz_shard_range(nBlocks, rank, nRanks, first, count);
}

void LizardGPU_offsetsFromSizes(const uint32_t* sizes, size_t nBlocks, uint64_t* offsets)
{
    lz_offsets_from_sizes(sizes, nBlocks, offsets);
}

int LizardGPU_setCollectives(const LizardGPU_Collectives* table)
{
    t_err[0] = 0;
    if (table && (!table->allGather || !table->broadcast || !table->groupStart || !table->groupEnd)) {
        snprintf(t_err, sizeof t_err, "LizardGPU_setCollectives: every member of the table must be set");
        return -LIZARDGPU_ERR_ARG;
    }
    pthread_mutex_lock(&g_rccl_mu);
    g_haveUserCol = table != nullptr;
    if (table) { g_userCol.allGather = table->allGather; g_userCol.broadcast = table->broadcast; g_userCol.groupStart = table->groupStart; g_userCol.groupEnd = table->groupEnd; }
    pthread_mutex_unlock(&g_rccl_mu);
    return 0;
}

// 1 = RCCL came from a copy the process had already mapped, 0 = loaded by this library, -1 = not resolved yet
int LizardGPU_rcclShared(void) { return g_rccl.so ? (g_rccl.shared ? 1 : 0) : -1; }

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
    int callerDev = -1;
    if (hipGetDevice(&callerDev) != hipSuccess) { (void)hipGetLastError(); callerDev = -1; }
    // the communicators stay locked until the exchange is enqueued: a concurrent call with another device list would destroy them
    pthread_mutex_lock(&g_rccl_mu);
    int rc = g_haveUserCol ? 0 : rccl_load();
    if (!rc && !g_haveUserCol) {
        bool same = g_allCount == nDevices;
        for (int r = 0; same && r < nDevices; r++) same = g_allDevs[r] == devs[r];
        if (!same) {
            for (int i = 0; i < g_allCount; i++) if (g_allComms[i]) (void)g_rccl.CommDestroy(g_allComms[i]);
            g_allCount = 0;
            const ncclResult_t r_ = g_rccl.CommInitAll(g_allComms, nDevices, devs);
            if (r_ != ncclSuccess) rc = nccl_fail("ncclCommInitAll", r_);
            else {
                g_allCount = nDevices; memcpy(g_allDevs, devs, sizeof(int) * (size_t)nDevices);
                g_allSeen = 0;
                for (int r = 0; r < nDevices && !rc; r++) { int cnt, rk; rc = rccl_verify(g_allComms[r], nDevices, r, &cnt, &rk); if (!rc && cnt > 0) g_allSeen = cnt; }
                if (rc) abort_all_comms();
            }
        }
    }
    const int savedSel = t_device;
    hipStream_t streams[LZ_MAX_DEVICES] = {};
    int nLaunched = 0;                                          // ranks whose stream carries work of this call
    bool exchangeFailed = false;                                // a collective failed inside the group: the streams are not synchronised
    // 1. every device compresses its contiguous range; its sizes land in place inside its copy of the all-sizes array
    for (int r = 0; r < nDevices && !rc; r++) {
        size_t first, count;
        lz_shard_range(nBlocks, r, nDevices, &first, &count);
        t_device = devs[r];
        Guard g;
        if (g.rc) { rc = g.rc; break; }
        if ((rc = ctx_init(*g.c))) break;
        streams[r] = g.c->stage[0].stream;
        nLaunched = r + 1;
        g.c->hostKernelMs = -1.0f;
        rc = launch(*g.c, d_src[r], count, blockSize, r == nDevices - 1 ? lastBlockSize : blockSize, d_dst[r], dstStride,
                    d_allSizes[r] + first, level, streams[r]);
    }
    // 2. the exchange (one rank per device, grouped) and 3. sizes -> global output offsets on every device
    if (!rc) {
        void* comms[LZ_MAX_DEVICES]; void* strs[LZ_MAX_DEVICES];
        for (int r = 0; r < nDevices; r++) { comms[r] = g_haveUserCol ? (void*)(intptr_t)r : (void*)g_allComms[r]; strs[r] = (void*)streams[r]; }
        t_shardDevs = devs;
        // everything that can fail without the transport is checked BEFORE the group opens: a group that is closed with only
        // some of its ranks posted can hang in ncclGroupEnd or in the synchronise below
        for (int r = 0; r < nDevices && !rc; r++) rc = select_shard_device(r);
        if (!rc) {
            rc = lz_exchange_all(collectives(), kHipOps, nDevices, comms, nBlocks, d_allSizes, d_offsets, strs, select_shard_device);
            if (rc && !g_haveUserCol) { exchangeFailed = true; abort_all_comms(); }
        }
        t_shardDevs = nullptr;
    }
    for (int r = 0; r < nLaunched && !exchangeFailed; r++) {
        if (hipSetDevice(devs[r]) == hipSuccess && hipStreamSynchronize(streams[r]) != hipSuccess && !rc) {
            snprintf(t_err, sizeof t_err, "device %d: stream synchronise failed", devs[r]); rc = -LIZARDGPU_ERR_HIP;
        }
    }
    pthread_mutex_unlock(&g_rccl_mu);
    t_device = savedSel;
    if (callerDev >= 0) (void)hipSetDevice(callerDev);          // every entry point leaves the caller's device as it found it
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

// Lock order of this file: g_rccl_mu first, then a device context (Guard) — as LizardGPU_compressBlocks_sharded does.
int LizardGPU_commInitRank(const void* id128, int nRanks, int rank)
{
    t_err[0] = 0;
    if (!id128 || nRanks < 1 || rank < 0 || rank >= nRanks) { snprintf(t_err, sizeof t_err, "bad argument"); return -LIZARDGPU_ERR_ARG; }
    pthread_mutex_lock(&g_rccl_mu);
    int rc;
    {
        Guard g;                                                // the communicator lives on the selected device
        rc = g.rc;
        if (!rc && g_haveUserCol) {
            // a transport installed with LizardGPU_setCollectives needs no RCCL communicator: the rank index is what it is handed as `comm`
            if (g_rankComm && g_rccl.so) (void)g_rccl.CommDestroy(g_rankComm);
            g_rankComm = nullptr; g_rankCount = nRanks; g_rankIndex = rank; g_rankSeen = 0; g_rankSeenIndex = -1;
        } else if (!rc && !(rc = rccl_load())) {
            if (g_rankComm) { (void)g_rccl.CommDestroy(g_rankComm); g_rankComm = nullptr; }
            ncclUniqueId id;
            memcpy(&id, id128, sizeof id);
            const ncclResult_t r_ = g_rccl.CommInitRank(&g_rankComm, nRanks, id, rank);
            if (r_ != ncclSuccess) { g_rankComm = nullptr; g_rankCount = 0; g_rankIndex = -1; rc = nccl_fail("ncclCommInitRank", r_); }
            else if ((rc = rccl_verify(g_rankComm, nRanks, rank, &g_rankSeen, &g_rankSeenIndex))) {
                (void)g_rccl.CommDestroy(g_rankComm); g_rankComm = nullptr; g_rankCount = 0; g_rankIndex = -1;
            }
            else { g_rankCount = nRanks; g_rankIndex = rank; }
        }
    }
    pthread_mutex_unlock(&g_rccl_mu);
    return rc;
}

int LizardGPU_gatherSizes_device(const uint32_t* d_localSizes, size_t nBlocks, uint32_t* d_allSizes, uint64_t* d_offsets, void* stream)
{
    pthread_mutex_lock(&g_rccl_mu);
    int rc;
    {
        Guard g;
        rc = g.rc;
        if (rc) {}
        else if (g_rankIndex < 0 || (!g_rankComm && !g_haveUserCol)) { snprintf(t_err, sizeof t_err, "LizardGPU_commInitRank has not been called"); rc = -LIZARDGPU_ERR_ARG; }
        else if (!d_localSizes || !d_allSizes || !d_offsets || nBlocks < (size_t)g_rankCount || nBlocks > 0xFFFFFFFFu) {
            snprintf(t_err, sizeof t_err, "bad argument"); rc = -LIZARDGPU_ERR_ARG;
        }
        else rc = lz_gather_sizes(collectives(), kHipOps, g_haveUserCol ? (void*)(intptr_t)g_rankIndex : (void*)g_rankComm, g_rankIndex, g_rankCount,
                                  d_localSizes, nBlocks, d_allSizes, d_offsets, stream);
    }
    pthread_mutex_unlock(&g_rccl_mu);
    return rc;
}

// What carries the size exchange of this process, as the transport itself reports it (bench.py puts it into its line so that an
// N-GPU figure says whether RCCL saw N ranks).  info[0] = 0 RCCL / 1 a table installed with LizardGPU_setCollectives;
// info[1] = ranks the caller asked for (LizardGPU_commInitRank, 0 = none); info[2] = ranks the RCCL communicator reports
// (ncclCommCount; 0 = no RCCL communicator, -1 = this RCCL does not say); info[3] = this rank as RCCL reports it (ncclCommUserRank,
// -1 = n/a); info[4] = ncclGetVersion (0 = RCCL not loaded / no such call); info[5] = ranks of the single-process communicators of
// LizardGPU_compressBlocks_sharded (0 = none).  Always 0.
int LizardGPU_commInfo(int info[6])
{
    pthread_mutex_lock(&g_rccl_mu);
    int ver = 0;
    if (g_rccl.so && g_rccl.GetVersion && g_rccl.GetVersion(&ver) != ncclSuccess) ver = 0;
    info[0] = g_haveUserCol ? 1 : 0; info[1] = g_rankCount; info[2] = g_rankComm ? g_rankSeen : 0; info[3] = g_rankComm ? g_rankSeenIndex : -1;
    info[4] = ver; info[5] = g_allCount ? g_allSeen : 0;
    pthread_mutex_unlock(&g_rccl_mu);
    return 0;
}

int LizardGPU_commDestroy(void)
{
    pthread_mutex_lock(&g_rccl_mu);
    if (g_rccl.so && g_rankComm) (void)g_rccl.CommDestroy(g_rankComm);
    g_rankComm = nullptr; g_rankCount = 0; g_rankIndex = -1; g_rankSeen = 0; g_rankSeenIndex = -1;
    pthread_mutex_unlock(&g_rccl_mu);
    return 0;
}

}  // extern "C"
