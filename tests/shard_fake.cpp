// tests/shard_fake.cpp — TEST INFRASTRUCTURE ONLY: the library's multi-GPU exchange (lizard_amd/csrc/lizard_shard_core.h: partition,
// "my shard into place", in-place all-gather or ragged per-root broadcasts, sizes -> offsets) run with 1..4 ranks on a CPU over
// a fake shared-memory transport.  The product fills the same collective table with RCCL (lizard_shard.h); on the pool's
// one-GPU boxes that code had only ever seen one rank.  Two deployments, as in the library:
//   * one thread per rank (the torchrun form, LizardGPU_gatherSizes_device -> lz_gather_sizes): every collective is a rendezvous;
//   * one thread driving all ranks inside a group (LizardGPU_compressBlocks_sharded -> lz_exchange_all): calls are queued and
//     run at groupEnd, where the ranks' call sequences must agree (kind, count, root) like NCCL demands.
// "Device memory" is host memory, "streams" are null.  Prints one line per scenario; exit code 0 = all good.
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../lizard_amd/csrc/lizard_shard_core.h"

namespace {

enum Kind { ALLGATHER, BROADCAST };
struct Op { Kind kind; const uint32_t* send; uint32_t* recv; size_t count; int root; };
struct World;
struct RankCtx { int rank; World* w; };
struct World {
    int n = 0;
    pthread_barrier_t bar;
    std::vector<Op> slot;                     // threaded mode: the op each rank is in
    std::vector<std::vector<Op>> queue;       // grouped mode: per-rank call sequence
    int inPlaceGathers = 0, gathers = 0, broadcasts = 0;
    int failBroadcastAt = -1;                 // grouped mode: the k-th broadcast call reports an error
    bool mismatch = false;
};
World* g_w = nullptr;
bool g_grouped = false;
int g_groupDepth = 0, g_groupEnds = 0, g_bcastCalls = 0;

void run_matched(World& w, const std::vector<Op>& ops)    // one collective, ops[r] = rank r's call
{
    const Op& o0 = ops[0];
    for (int r = 1; r < w.n; r++)
        if (ops[r].kind != o0.kind || ops[r].count != o0.count || ops[r].root != o0.root) { w.mismatch = true; return; }
    if (o0.kind == ALLGATHER) {
        // gather into temporaries first: in-place sends live inside the receive buffers
        std::vector<uint32_t> tmp((size_t)w.n * o0.count);
        for (int r = 0; r < w.n; r++) memcpy(&tmp[(size_t)r * o0.count], ops[r].send, o0.count * 4);
        for (int r = 0; r < w.n; r++) memcpy(ops[r].recv, tmp.data(), tmp.size() * 4);
    } else {
        std::vector<uint32_t> tmp(ops[o0.root].send, ops[o0.root].send + o0.count);
        for (int r = 0; r < w.n; r++) memcpy(ops[r].recv, tmp.data(), o0.count * 4);
    }
}

int submit(const Op& op, void* comm)
{
    RankCtx* rc = (RankCtx*)comm;
    World& w = *rc->w;
    if (g_grouped) {
        if (g_groupDepth != 1) { w.mismatch = true; return -9; }        // the single-thread form must bracket its calls
        w.queue[(size_t)rc->rank].push_back(op);
        return 0;
    }
    w.slot[(size_t)rc->rank] = op;
    pthread_barrier_wait(&w.bar);
    if (rc->rank == 0) run_matched(w, w.slot);
    pthread_barrier_wait(&w.bar);
    return 0;
}

int fake_all_gather(const void* send, void* recv, size_t count, void* comm, void*)
{
    RankCtx* rc = (RankCtx*)comm;
    __atomic_add_fetch(&rc->w->gathers, 1, __ATOMIC_RELAXED);
    if (send == (const uint32_t*)recv + (size_t)rc->rank * count) __atomic_add_fetch(&rc->w->inPlaceGathers, 1, __ATOMIC_RELAXED);
    return submit(Op{ALLGATHER, (const uint32_t*)send, (uint32_t*)recv, count, -1}, comm);
}
int fake_broadcast(const void* send, void* recv, size_t count, int root, void* comm, void*)
{
    RankCtx* rc = (RankCtx*)comm;
    __atomic_add_fetch(&rc->w->broadcasts, 1, __ATOMIC_RELAXED);
    if (g_grouped && rc->w->failBroadcastAt >= 0 && g_bcastCalls++ == rc->w->failBroadcastAt) return -7;
    return submit(Op{BROADCAST, (const uint32_t*)send, (uint32_t*)recv, count, root}, comm);
}
int fake_group_start() { g_groupDepth++; return 0; }
int fake_group_end()
{
    g_groupDepth--; g_groupEnds++;
    World& w = *g_w;
    size_t len = w.queue[0].size();
    for (int r = 1; r < w.n; r++) if (w.queue[(size_t)r].size() != len) { w.mismatch = true; len = 0; }
    for (size_t i = 0; i < len; i++) {
        std::vector<Op> ops;
        for (int r = 0; r < w.n; r++) ops.push_back(w.queue[(size_t)r][i]);
        run_matched(w, ops);
    }
    for (auto& q : w.queue) q.clear();
    return 0;
}
const LzCollectives kFake = { fake_all_gather, fake_broadcast, fake_group_start, fake_group_end };

int host_copy(uint32_t* dst, const uint32_t* src, size_t count, void*) { memcpy(dst, src, count * 4); return 0; }
int host_scan(const uint32_t* sizes, uint64_t* offsets, size_t n, void*) { lz_offsets_from_sizes(sizes, n, offsets); return 0; }
const LzDeviceOps kHost = { host_copy, host_scan };

uint32_t rnd(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }

struct RankJob { RankCtx ctx; size_t nBlocks; const uint32_t* truth; bool separateLocal; std::vector<uint32_t> all; std::vector<uint64_t> offs; int rc; };
void* rank_thread(void* a)
{
    RankJob& j = *(RankJob*)a;
    size_t first, count;
    lz_shard_range(j.nBlocks, j.ctx.rank, j.ctx.w->n, &first, &count);
    j.all.assign(j.nBlocks, 0xDEADBEEFu);                               // other ranks' shards: garbage until the exchange
    j.offs.assign(j.nBlocks + 1, ~0ull);
    std::vector<uint32_t> local(j.truth + first, j.truth + first + count);
    const uint32_t* lp = local.data();
    if (!j.separateLocal) { memcpy(&j.all[first], local.data(), count * 4); lp = &j.all[first]; }   // already in place
    j.rc = lz_gather_sizes(kFake, kHost, &j.ctx, j.ctx.rank, j.ctx.w->n, lp, j.nBlocks, j.all.data(), j.offs.data(), nullptr);
    return nullptr;
}

bool check(const std::vector<uint32_t>& all, const std::vector<uint64_t>& offs, const std::vector<uint32_t>& truth)
{
    if (all != truth) return false;
    uint64_t run = 0;
    for (size_t i = 0; i < truth.size(); i++) { if (offs[i] != run) return false; run += truth[i]; }
    return offs[truth.size()] == run;
}

int scenario(int nRanks, size_t nBlocks, uint64_t seed)
{
    std::vector<uint32_t> truth(nBlocks);
    for (auto& v : truth) v = 1u + rnd(seed) % 262158u;
    int bad = 0;
    // ---- one thread per rank ----
    for (int separate = 0; separate < 2; separate++) {
        World w; w.n = nRanks; w.slot.resize((size_t)nRanks); w.queue.resize((size_t)nRanks);
        pthread_barrier_init(&w.bar, nullptr, (unsigned)nRanks);
        g_w = &w; g_grouped = false;
        std::vector<RankJob> jobs((size_t)nRanks);
        std::vector<pthread_t> th((size_t)nRanks);
        for (int r = 0; r < nRanks; r++) { jobs[(size_t)r].ctx = RankCtx{r, &w}; jobs[(size_t)r].nBlocks = nBlocks; jobs[(size_t)r].truth = truth.data(); jobs[(size_t)r].separateLocal = separate != 0; }
        for (int r = 0; r < nRanks; r++) pthread_create(&th[(size_t)r], nullptr, rank_thread, &jobs[(size_t)r]);
        for (int r = 0; r < nRanks; r++) pthread_join(th[(size_t)r], nullptr);
        pthread_barrier_destroy(&w.bar);
        for (auto& j : jobs) if (j.rc || !check(j.all, j.offs, truth)) bad++;
        const bool equal = nBlocks % (size_t)nRanks == 0;
        if (w.mismatch) bad++;
        if (equal && (w.gathers != nRanks || w.inPlaceGathers != nRanks || w.broadcasts != 0)) bad++;   // ONE in-place all-gather per rank
        if (!equal && (w.gathers != 0 || w.broadcasts != nRanks * nRanks)) bad++;                        // one broadcast per root per rank
    }
    // ---- one thread, all ranks, grouped ----
    {
        World w; w.n = nRanks; w.slot.resize((size_t)nRanks); w.queue.resize((size_t)nRanks);
        g_w = &w; g_grouped = true; g_groupDepth = 0; g_groupEnds = 0; g_bcastCalls = 0;
        std::vector<RankCtx> ctx((size_t)nRanks);
        std::vector<std::vector<uint32_t>> all((size_t)nRanks);
        std::vector<std::vector<uint64_t>> offs((size_t)nRanks);
        std::vector<void*> comms, streams; std::vector<uint32_t*> ap; std::vector<uint64_t*> op;
        for (int r = 0; r < nRanks; r++) {
            ctx[(size_t)r] = RankCtx{r, &w};
            size_t first, count;
            lz_shard_range(nBlocks, r, nRanks, &first, &count);
            all[(size_t)r].assign(nBlocks, 0xDEADBEEFu); offs[(size_t)r].assign(nBlocks + 1, ~0ull);
            memcpy(&all[(size_t)r][first], &truth[first], count * 4);
            comms.push_back(&ctx[(size_t)r]); streams.push_back(nullptr); ap.push_back(all[(size_t)r].data()); op.push_back(offs[(size_t)r].data());
        }
        const int rc = lz_exchange_all(kFake, kHost, nRanks, comms.data(), nBlocks, ap.data(), op.data(), streams.data(), nullptr);
        if (rc || w.mismatch || g_groupDepth != 0 || g_groupEnds != 1) bad++;
        for (int r = 0; r < nRanks; r++) if (!check(all[(size_t)r], offs[(size_t)r], truth)) bad++;
        // a transport error inside the group: reported, and the group is still closed
        if (nBlocks % (size_t)nRanks != 0) {
            World w2; w2.n = nRanks; w2.slot.resize((size_t)nRanks); w2.queue.resize((size_t)nRanks); w2.failBroadcastAt = 1;
            g_w = &w2; g_groupDepth = 0; g_groupEnds = 0; g_bcastCalls = 0;
            for (auto& c : ctx) c.w = &w2;
            const int rc2 = lz_exchange_all(kFake, kHost, nRanks, comms.data(), nBlocks, ap.data(), op.data(), streams.data(), nullptr);
            if (rc2 != -7 || g_groupDepth != 0 || g_groupEnds != 1) bad++;
        }
    }
    printf("ranks %d blocks %zu (%s): %s\n", nRanks, nBlocks, nBlocks % (size_t)nRanks ? "ragged" : "equal", bad ? "FAILED" : "ok");
    return bad;
}

}  // namespace

int main()
{
    int bad = 0;
    const size_t sizes[] = { 4, 6, 7, 12, 13, 1000, 1001, 4099, 65536, 65537 };
    for (int nRanks = 1; nRanks <= 4; nRanks++)
        for (size_t nb : sizes) if (nb >= (size_t)nRanks) bad += scenario(nRanks, nb, 77u * (uint64_t)nRanks + nb);
    bad += scenario(3, 3, 5); bad += scenario(2, 2, 6); bad += scenario(8, 65536 * 8, 7); bad += scenario(8, 65536 * 8 + 3, 8);
    printf("%s\n", bad ? "FAILED" : "ALL OK");
    return bad ? 1 : 0;
}
