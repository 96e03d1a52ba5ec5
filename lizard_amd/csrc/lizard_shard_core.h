// lizard_shard_core.h — the multi-GPU exchange of the library with NOTHING device-specific in it.
//
// Blocks are independent, so a batch shards over ranks by contiguous block ranges; the one exchange is the all-gather
// of the per-block compressed sizes (uint32 per block), after which every rank turns sizes into byte offsets of the
// concatenated output.  This file holds that logic — partition, "my shard into place", in-place all-gather or the
// ragged per-root broadcasts, offsets — written against two small function tables:
//   LzCollectives  all-gather / broadcast / group of a transport (RCCL in the product: lizard_shard.h; a shared-memory
//                  fake in tests/shard_fake.cpp, which runs the very same code with 2 and 3 ranks on a CPU);
//   LzDeviceOps    the two device-side steps around the exchange (copy of a shard into place, sizes -> offsets scan).
// Communicators, streams and buffers are opaque pointers here.  Every function returns 0 or a negative LIZARDGPU_ERR_*.
// There is no reference code to match: the reference is single-threaded.
#pragma once
#include <stddef.h>
#include <stdint.h>

struct LzCollectives {
    // u32 elements.  In place when send == recv + rank * count (the rccl.h contract of ncclAllGather).
    int (*allGather)(const void* send, void* recv, size_t count, void* comm, void* stream);
    int (*broadcast)(const void* send, void* recv, size_t count, int root, void* comm, void* stream);
    int (*groupStart)(void);                      // one thread driving several ranks brackets their calls
    int (*groupEnd)(void);
};
struct LzDeviceOps {
    int (*copyU32)(uint32_t* dst, const uint32_t* src, size_t count, void* stream);            // device to device, on `stream`
    int (*scanSizes)(const uint32_t* sizes, uint64_t* offsets, size_t nBlocks, void* stream);  // offsets[i] = sum sizes[<i], offsets[n] = total
};

// rank r of nRanks owns blocks [first, first + count): the first (nBlocks % nRanks) ranks take one block more
static inline void lz_shard_range(size_t nBlocks, int rank, int nRanks, size_t* first, size_t* count)
{
    const size_t base = nBlocks / (size_t)nRanks, rem = nBlocks % (size_t)nRanks, r = (size_t)rank;
    if (first) *first = r * base + (r < rem ? r : rem);
    if (count) *count = base + (r < rem ? 1 : 0);
}

static inline void lz_offsets_from_sizes(const uint32_t* sizes, size_t nBlocks, uint64_t* offsets)
{
    uint64_t run = 0;
    for (size_t i = 0; i < nBlocks; i++) { offsets[i] = run; run += sizes[i]; }
    offsets[nBlocks] = run;
}

// Rank `rank`'s slice of the all-sizes array is already in place at all + first(rank); afterwards every rank holds
// all of it.  Called once per rank with that rank's communicator and stream (inside a group when one thread drives
// several ranks).  Equal shards: ONE all-gather in place; ragged partition: one broadcast per root.
static inline int lz_gather_in_place(const LzCollectives& col, void* comm, int rank, int nRanks, size_t nBlocks, uint32_t* all, void* stream)
{
    if (nBlocks % (size_t)nRanks == 0) {
        const size_t per = nBlocks / (size_t)nRanks;
        return col.allGather(all + (size_t)rank * per, all, per, comm, stream);
    }
    for (int root = 0; root < nRanks; root++) {
        size_t first, count;
        lz_shard_range(nBlocks, root, nRanks, &first, &count);
        if (!count) continue;
        const int rc = col.broadcast(all + first, all + first, count, root, comm, stream);
        if (rc) return rc;
    }
    return 0;
}

// One rank of a one-process-per-device job: local sizes (this rank's shard; may already sit at all + first) -> all
// sizes on every rank -> offsets.  Everything is enqueued on `stream`.
static inline int lz_gather_sizes(const LzCollectives& col, const LzDeviceOps& dev, void* comm, int rank, int nRanks,
                                  const uint32_t* local, size_t nBlocks, uint32_t* all, uint64_t* offsets, void* stream)
{
    size_t first, count;
    lz_shard_range(nBlocks, rank, nRanks, &first, &count);
    int rc = 0;
    if (local != all + first && count) rc = dev.copyU32(all + first, local, count, stream);
    if (!rc) rc = lz_gather_in_place(col, comm, rank, nRanks, nBlocks, all, stream);
    if (!rc) rc = dev.scanSizes(all, offsets, nBlocks, stream);
    return rc;
}

// One thread driving nRanks devices: rank r's shard already sits in place inside all[r].  select(r) makes rank r's device
// current (may be null).  The group is always closed, also on an error inside it.
static inline int lz_exchange_all(const LzCollectives& col, const LzDeviceOps& dev, int nRanks, void* const* comms, size_t nBlocks,
                                  uint32_t* const* all, uint64_t* const* offsets, void* const* streams, int (*select)(int rank))
{
    int rc = col.groupStart();
    if (rc) return rc;
    for (int r = 0; r < nRanks && !rc; r++) {
        if (select && (rc = select(r))) break;
        rc = lz_gather_in_place(col, comms[r], r, nRanks, nBlocks, all[r], streams[r]);
    }
    const int rcEnd = col.groupEnd();
    if (!rc) rc = rcEnd;
    for (int r = 0; r < nRanks && !rc; r++) {
        if (select && (rc = select(r))) break;
        rc = dev.scanSizes(all[r], offsets[r], nBlocks, streams[r]);
    }
    return rc;
}
