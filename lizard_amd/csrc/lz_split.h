// lz_split.h — levels 10 / 30 with the parse and the container decoupled: PRODUCER waves own a hash table in LDS and only
// parse; CONSUMER waves own no table and only run the container of finished sub-blocks (reference Lizard_writeBlock,
// lib/lizard_compress.c:186-250: fastLZ4 encode pass, lib/lizard_compress_lz4.h:3-86, and for levels >= 30 the huff0 stage,
// lib/entropy/huf_compress.c:517-609).  A hash table is what limits the number of blocks in flight on a CU (LDS), and in the
// one-wave-per-block form it sits idle while its wave encodes and entropy-codes (28 % of a level-30 wave's time, 8 % at level
// 10); here the tables never idle, and the Huffman workspaces exist once per consumer instead of being pooled.
//
// Hand-over.  A producer parses sub-block after sub-block of its block into one of its nBufs sequence buffers in global
// memory (buffer = a 64-byte job header + the sequence list), then publishes the buffer's index in the mailbox of the consumer
// the BLOCK was bound to when it was claimed (round robin): all sub-blocks of a block go through one mailbox, in order, so the
// consumer knows where in dst the next sub-block starts (the container's sizes are only known after the entropy stage).  The
// mailbox is a ring of qn words in LDS: producers take a ticket (one LDS atomic), write their word, the consumer polls
// the word at its head.  A consumer hands a buffer back through a bit in the producer's free mask.  Everything a job needs
// beyond the sequence list travels in the header; the waves of a workgroup share nothing else, and a producer only ever waits
// for a free buffer, a consumer only for a job: no cycle, no deadlock.  Memory order: the header and the list are global-memory
// stores; release / acquire fences at agent scope around the LDS word that publishes them.
//
// Test infrastructure runs this file on the CPU with several emulated waves on OS threads (tests/emul, emul_compress_split).
#pragma once
#include "lz_block.h"
#ifndef LZ_SPLIT_LEAN
#define LZ_SPLIT_LEAN 1                                      // the consumers sum up the stream sizes (lz_seq_sizes), not the producers
#endif

// Per kernel (LzSplitArgs): nBufs = sequence buffers per producer, qn = mailbox words per consumer (a power of two >= nProd x nBufs)
#define LZ_SPLIT_HDR      64u                                // job header bytes in front of a buffer's sequence list
#define LZ_SPLIT_BUF_BYTES (LZ_SPLIT_HDR + LZ_SEQ_BYTES)
#define LZ_SPLIT_OPS_BYTES 256u                              // per consumer, behind its staging areas: where each producer's current block stands in dst

// LDS shared by the waves of one workgroup (zeroed by the kernel before the waves part ways, except bufFree = all free).
// Laid out by lz_split_shared(): nextCons, prodDone, qTail[nCons], bufFree[nProd], q[nCons][qn].
struct LzSplitShared {
    u32* nextCons;                                           // round-robin binding of blocks to consumers
    u32* prodDone;                                           // producers that have left
    u32* qTail;                                              // [nCons] tickets handed out per mailbox
    u32* bufFree;                                            // [nProd] bit j: buffer j of the producer is free
    u32* q;                                                  // [nCons][qn] 0 = empty, else buffer index + 1
};
#define LZ_SPLIT_SHARED_WORDS(nProd, nCons, qn) (2u + (nCons) + (nProd) + (nCons) * (qn))
LZ_DEV LzSplitShared lz_split_shared(u32* mem, u32 nProd, u32 nCons)
{
    LzSplitShared sh;
    sh.nextCons = mem; sh.prodDone = mem + 1; sh.qTail = mem + 2; sh.bufFree = sh.qTail + nCons; sh.q = sh.bufFree + nProd;
    return sh;
}
// (one wave, before the others read it: the kernel puts a workgroup barrier behind this)
LZ_DEV void lz_split_shared_init(const LzSplitShared& sh, u32 nProd, u32 nCons, u32 nBufs, u32 qn)
{
    for (u32 i = lz_lane(); i < LZ_SPLIT_SHARED_WORDS(nProd, nCons, qn); i += 64u) sh.nextCons[i] = 0u;
    lz_lds_sync();
    for (u32 i = lz_lane(); i < nProd; i += 64u) sh.bufFree[i] = (1u << nBufs) - 1u;
    lz_lds_sync();
}

// job header, 16 words at the start of a sequence buffer (written by the producer, read by the consumer)
enum { LZJ_BLOCK = 0, LZJ_S, LZJ_E, LZJ_FLAGS, LZJ_NSEQ, LZJ_NLIT, LZJ_NFLAGS, LZJ_LASTLITS, LZJ_WORDS };
#define LZJ_FIRST 1u                                         // first sub-block of its block
#define LZJ_LAST  2u                                         // last sub-block of its block

struct LzSplitArgs {
    const u8* src; u64 blockSize; u32 nBlocks; u32 lastBlockSize;
    u8* dst; u64 dstStride; u32* sizes; u32 level;
    u32* counter;                                            // device-wide block counter
    u8* arena;                                               // this workgroup's scratch: producers' buffers, then consumers' staging
    u32 nProd, nCons;
    u32 nBufs, qn;                                           // sequence buffers per producer; mailbox words per consumer
    const u32* srcSizes;                                     // per-block input sizes of a ragged batch, or nullptr (LzBatch::srcSizes)
    const u64* srcOffsets;                                   // per-block input offsets of a ragged batch, or nullptr (LzBatch::srcOffsets)
    u32 activeProd;                                          // producers that claim blocks (LzBatch::activeWaves); the others leave at once
};
LZ_DEV u8* lz_split_buf(const LzSplitArgs& a, u32 bufIndex) { return a.arena + (u64)bufIndex * LZ_SPLIT_BUF_BYTES; }
#define LZ_SPLIT_CONS_BYTES (2u * LZ_SUBBLOCK_PAD + LZ_SPLIT_OPS_BYTES)
LZ_DEV u8* lz_split_staging(const LzSplitArgs& a, u32 cons) { return a.arena + (u64)a.nProd * a.nBufs * LZ_SPLIT_BUF_BYTES + (u64)cons * LZ_SPLIT_CONS_BYTES; }
#define LZ_SPLIT_ARENA_BYTES(nProd, nCons, nBufs) ((size_t)(nProd) * (nBufs) * LZ_SPLIT_BUF_BYTES + (size_t)(nCons) * LZ_SPLIT_CONS_BYTES)
#define LZ_SPLIT_PROF_END ((u64)16u * LZ_SCRATCH_BYTES)       // profile builds: 16 x 128-byte records at the end of the workgroup's arena (LZ_MAX_WAVES slots)

// ---- producer: claim blocks, parse their sub-blocks, publish one job per sub-block ----
template <int HASHLOG>
LZ_DEV void lz_split_producer(const LzSplitArgs& a, const LzSplitShared& sh, u32 prod, void* tableMem, u64* seqRing)
{
    const u32 lane = lz_lane();
    LzStreams st;
    st.ring = seqRing; st.lit = st.flags = st.off16 = st.off24 = nullptr;
#ifdef LZ_PROFILE
    st.prof_last = __builtin_readcyclecounter();
    for (int k = 0; k < 16; k++) st.prof[k] = 0;
#endif
    const LzTab tab = lz_tab_bind<HASHLOG>(tableMem);
    for (;;) {
        lz_converge();
        if (prod >= a.activeProd) break;                                         // a small launch: this producer's blocks went to other CUs
        const u32 b = lz_claim_index(a.counter);
        if (b >= a.nBlocks) break;
        const u32 n = a.srcSizes ? lz_uniform(a.srcSizes[b]) : (b == a.nBlocks - 1u) ? a.lastBlockSize : (u32)a.blockSize;
        const u8* src = a.src + (a.srcOffsets ? lz_uniform64(a.srcOffsets[b]) : (u64)b * a.blockSize);
        const u32 cons = lz_lds_claim(sh.nextCons) % a.nCons;                  // the block's consumer
        lz_tab_fresh<HASHLOG>(tab); st.sweepAt = LzTab::kSweepEvery;
        lz_lds_sync();
        for (u32 pos = 0; pos < n; ) {                                           // (n >= 1: the launcher refuses empty blocks)
            const u32 part = (n - pos) < LZ_SUBBLOCK ? (n - pos) : LZ_SUBBLOCK;
            // a free buffer of mine (the consumer sets the bit when it is done with the buffer)
            u32 j;
            for (;;) {
                lz_converge();
                const u32 freeBits = lz_lds_poll_u(&sh.bufFree[prod]);
                if (freeBits) { j = 31u - (u32)__builtin_clz(freeBits); break; }
                lz_sleep();
            }
            lz_lds_atomic_and(&sh.bufFree[prod], lane == 0 ? ~(1u << j) : 0xFFFFFFFFu);
            LZ_PROF(st, 6);                                                      // (profile builds) waiting for a free sequence buffer
            const u32 bufIndex = prod * a.nBufs + j;
            u8* const buf = lz_split_buf(a, bufIndex);
            st.seq = (u64*)(buf + LZ_SPLIT_HDR);
            st.nlit = st.nflags = st.noff16 = st.noff24 = 0;                      // Lizard_initBlock, lizard_compress.c:130-138
            st.nseq = 0; st.lastLits = 0;
#if LZ_FAST_128
            lz_parse_fast128<HASHLOG>(src, pos, pos + part, tab, st);
#else
            lz_parse_fast<HASHLOG, LzTab, LZ_SPLIT_LEAN != 0>(src, pos, pos + part, tab, st);
#endif
            // the job header
            {
                const u32 flags = (pos == 0 ? LZJ_FIRST : 0u) | (pos + part >= n ? LZJ_LAST : 0u);
                u32 v = 0;
                v = lane == LZJ_BLOCK ? b : v;        v = lane == LZJ_S ? pos : v;          v = lane == LZJ_E ? pos + part : v;
                v = lane == LZJ_FLAGS ? flags : v;    v = lane == LZJ_NSEQ ? st.nseq : v;   v = lane == LZJ_NLIT ? st.nlit : v;
                v = lane == LZJ_NFLAGS ? st.nflags : v; v = lane == LZJ_LASTLITS ? st.lastLits : v;
                if (lane < LZJ_WORDS) ((u32*)buf)[lane] = v;
            }
            lz_publish_release();                                                // header and list are written before the word that announces them
            const u32 t = lz_lds_claim(&sh.qTail[cons]);
            if (lane == 0) lz_lds_store(&sh.q[cons * a.qn + (t & (a.qn - 1u))], bufIndex + 1u);
            lz_converge();
            LZ_PROF(st, 3);                                                      // job header, publish
            pos += part;
        }
    }
    lz_lds_atomic_add(sh.prodDone, lane == 0 ? 1u : 0u);
    lz_converge();
#ifdef LZ_PROFILE
    // per-wave totals: profile record `prod` at the END of the workgroup's arena (the scratch-slot tails of the one-wave form lie
    // inside the sequence buffers here)
    if (lane == 0) { u64* pr = (u64*)(a.arena + LZ_SPLIT_PROF_END - (u64)(prod + 1u) * 128u); for (int k = 0; k < 15; k++) pr[k] += st.prof[k]; }
    lz_converge();
#endif
}

// ---- consumer: take jobs from my mailbox, run the container of each, hand the buffer back ----
template <bool HUF>
LZ_DEV void lz_split_consumer(const LzSplitArgs& a, const LzSplitShared& sh, u32 cons, u32* hufWs)
{
    const u32 lane = lz_lane();
    LzStreams st;
    st.ring = nullptr;
    st.lit = lz_split_staging(a, cons); st.flags = st.lit + LZ_SUBBLOCK_PAD; st.off16 = st.off24 = st.flags;
#ifdef LZ_PROFILE
    st.prof_last = __builtin_readcyclecounter();
    for (int k = 0; k < 16; k++) st.prof[k] = 0;
#endif
    LzHufPool pool; pool.base = hufWs; pool.mask = nullptr; pool.count = 0; pool.stride = LZ_HUF_WS_WORDS;   // the consumer owns its workspace
    u32 head = 0;                                                                // uniform
    for (;;) {
        lz_converge();
        u32 word = lz_lds_poll_u(&sh.q[cons * a.qn + (head & (a.qn - 1u))]);
        if (!word) {
            // nothing yet: leave once every producer has left AND the mailbox is still empty after that was seen
            if (lz_lds_poll_u(sh.prodDone) == a.nProd) {
                word = lz_lds_poll_u(&sh.q[cons * a.qn + (head & (a.qn - 1u))]);
                if (!word) break;
            } else { lz_sleep(); continue; }
        }
        LZ_PROF(st, 7);                                                          // (profile builds) consumer: waiting for a job
        lz_publish_acquire();                                                    // the header and the list behind the word
        if (lane == 0) lz_lds_store(&sh.q[cons * a.qn + (head & (a.qn - 1u))], 0u);
        lz_converge();
        head++;
        const u32 bufIndex = word - 1u;
        u8* const buf = lz_split_buf(a, bufIndex);
        const u32 hv = lane < LZJ_WORDS ? lz_ld_shared_u32((const u32*)buf + lane) : 0u;
        const u32 b = lz_readlane(hv, LZJ_BLOCK), S = lz_readlane(hv, LZJ_S), E = lz_readlane(hv, LZJ_E), flags = lz_readlane(hv, LZJ_FLAGS);
        st.seq = (u64*)(buf + LZ_SPLIT_HDR);
        st.nseq = lz_readlane(hv, LZJ_NSEQ); st.nlit = lz_readlane(hv, LZJ_NLIT); st.nflags = lz_readlane(hv, LZJ_NFLAGS);
        st.lastLits = lz_readlane(hv, LZJ_LASTLITS); st.noff16 = st.noff24 = 0;
        const u8* src = a.src + (a.srcOffsets ? lz_uniform64(a.srcOffsets[b]) : (u64)b * a.blockSize);
        u8* dst = a.dst + (u64)b * a.dstStride;
        // where this sub-block starts in dst: 1 (behind the level byte, lizard_compress.c:488) or where the block's previous
        // sub-block — handled by this consumer, from this producer — ended: kept in a word of this consumer's own scratch
        u32* const opSlot = (u32*)(st.lit + 2u * LZ_SUBBLOCK_PAD) + bufIndex / a.nBufs;
        u32 op;
        if (flags & LZJ_FIRST) { if (lane == 0) dst[0] = (u8)a.level; lz_converge(); op = 1u; }
        else op = lz_uniform(lz_ld_shared_u32(opSlot));
#if LZ_SPLIT_LEAN
        lz_seq_sizes(st);                                                        // the stream sizes the producer did not keep (lz_seq_push<LEAN>)
#endif
        LZ_PROF(st, 13);                                                         // (profile builds) job header + stream sizes
        if (E > S) op += lz_write_subblock_seq<HUF, false, 4>(src, S, E, dst + op, st, pool);
        lz_wave_sync();                                                          // my reads of the list and my staging traffic are done
        if (flags & LZJ_LAST) { if (lane == 0) a.sizes[b] = op; }
        else if (lane == 0) lz_st_shared_u32(opSlot, op);
        lz_converge();
        lz_wave_sync();
        lz_lds_atomic_or(&sh.bufFree[bufIndex / a.nBufs], lane == 0 ? 1u << (bufIndex % a.nBufs) : 0u);   // back to its producer
        LZ_PROF(st, 5);                                                          // consumer: the container of one sub-block
    }
#ifdef LZ_PROFILE
    if (lane == 0) { u64* pr = (u64*)(a.arena + LZ_SPLIT_PROF_END - (u64)(a.nProd + cons + 1u) * 128u); for (int k = 0; k < 15; k++) pr[k] += st.prof[k]; }
    lz_converge();
#endif
}
