// lz_hashchain.h — hashChain parser (levels 13-17 / 34-38, fastLZ4 codewords) on one wavefront; since round 6 also the noChain
// parser (levels 12 / 32 / 33, reference lib/lizard_parser_nochain.h), which is the same parse over a chain of length one: see
// "noChain" below.
//
// Bit-exact with reference lib/lizard_parser_hashchain.h (Lizard_Insert :13-43, Lizard_InsertAndFindBestMatch
// :45-107, Lizard_InsertAndGetWiderMatch :109-185, Lizard_compress_hashChain :188-369) on a zeroed state;
// parameters lizard_common.h:240-244 / :264-268 (windowLog 16, hashLog 18, contentLog 16,
// searchNum 2/4/8/16/256, searchLength 5/5/5/4/4).
//
// How the reference's state maps onto a wave.
//   * Lizard_Insert is a pure function of the data.  It visits every position exactly once and in
//     order (nextToUpdate only moves forward), and what it writes for position p — the distance from p to
//     the head of p's hash bucket at that moment, and the conditional head update ":38" — depends only on
//     positions < p.  A search at X starts from the head "after inserting everything below X", which is
//     exactly what Insert recorded for X itself.  So the chain is built AHEAD of the parse (phase A,
//     lz_hc_build), into prev[p] = distance to the previous head (0 = none inside the 64 KiB window), one
//     u16 per block position; the searches of phase B never touch a head table.
//     (Positions the reference never inserts — the tail after the last search — are invisible: a search at
//     X only ever follows links that start below X.  The reference never searches below nextToUpdate —
//     every search position exceeds the previous one, see the walk through :207-339 in DESIGN.md — and
//     tests/test_oracle pins that on the corpus.)
//   * chainTable is a 2^16 ring in the reference; entries are only read inside the 65535 window where
//     the ring has no aliasing, so a full per-position array is observably the same.  A link the reference
//     clamps to 65535 because it is longer (":30") ends its walk one step later (":92/:172" leave the
//     window); here it is stored as 0 = end of chain.  A true distance of exactly 65535 stays usable.
//   * Head table: never exists as a 2^18-slot array.  lz_hc_build brings the positions into hash-bin order and replays
//     each bin against a 2^12-slot table in LDS (blocks up to 4 MiB: 22-bit positions).
//   * Phase B: the outer "ip++ until a position has a match" loop (:204-206) stops at the first position one
//     of whose first searchNum chain candidates passes the reference's tests (any such candidate makes
//     ml >= 4 > 0) — a function of the position alone, evaluated for all positions ahead of the parse
//     (lz_hc_hits) into one bit each; the loop is a scan of those bits.  From there on the LZ4HC-style
//     arbitration is a serial chain per sequence and runs as
//     wave-uniform code; each search collects up to 64 chain candidates (one per lane), filters them in
//     parallel on the 4-byte test and measures the survivors with the wave-wide compare helpers.  The
//     reference's "first strictly longer match wins" over the chain order is "maximum length, earliest
//     on ties"; its one-byte pre-checks (:73, :146) are necessary conditions of "strictly longer" and
//     therefore unobservable.
//
//
// noChain (Lizard_InsertNoChain nochain.h:8-25, Lizard_InsertAndFindBestMatchNoChain :28-80, Lizard_InsertAndGetWiderMatchNoChain
// :83-143, Lizard_compress_noChain :146-318; parameters lizard_common.h:239, :262-263: windowLog 16, hashLog 18 / 14 / 18, always
// hash5, nochain.h:4).  Its insert is Lizard_Insert without the chain store (same conditional head update, same MIN_OFFSET), its two
// searches test ONE candidate — the bucket's head after inserting everything below the position, which is what prev[] records — and
// its main loop is hashchain.h:188-369 line by line except for one exit the hashChain parser has and noChain has not
// (hashchain.h:257-260 against nochain.h:210).  So: the chain build of the level's hashLog, searchNum 1, LzHc::noChain.
//
// Included from lz_block.h after the shared helpers.
#pragma once

#define LZ_HC_HASHLOG   18                      // hashChain levels and noChain levels 12 / 33; level 32: 14 (template parameter HLOG)
#define LZ_HC_NONE      0x80000000u             // "no head": p - NONE is >= 8 and > 65535 for every block position
#define LZ_HC_OPTIMAL_ML 18                     // (ML_MASK_LZ4-1)+MINMATCH, hashchain.h:3
// Chain build geometry (lz_hc_build): a block is handled in segments of 2^18 positions; a segment's positions are split
// into 2^6 bins by the top hash bits, each bin is replayed against a 2^12-slot head table in LDS, and the links return to
// position order through an LDS window of 2^14 positions.
#define LZ_HC_SEGLOG    18
#define LZ_HC_BINLOG    6
#define LZ_HC_SUBLOG    (LZ_HC_HASHLOG - LZ_HC_BINLOG)   // slots of a bin's head table (largest form; HLOG 14: 2^8)
#define LZ_HC_WINLOG    14
#define LZ_HC_ARENA_WORDS 8192u                 // 32 KiB: head table, then window
#define LZ_HC_REGION_WORDS (LZ_HC_ARENA_WORDS + 80u)     // + cursors[64] + a spare word
// Per-wave slot in global memory (nothing persists from block to block, nothing needs clearing):
//   bins   u32[2^18]  (hash low bits << 18 | position in segment), bin after bin
//   links  u32[2^18]  (link << 16 | position in its 2^14 window), in bin order
//   wins   u32[17*64] cursor of every bin at every window boundary
//   heads  u32[2^18]  head tables between segments (blocks above 2^18 positions only)
//   hits   u64[maxBlock/64 + 64]  one bit per position: "a search here finds a match" (lz_hc_hits)
//   prev   u16[maxBlock]   distance to the previous head of the position's bucket (0 = none): the chain
//   chain2 u32[maxBlock]   prev | two links at once << 16 (0 = the second is missing or out of every window): what the searches walk
//   best   u32[maxBlock]   result of the FIRST search at the position (lz_hc_hits): length | offset << 16, 0 = nothing,
//                          LZ_HC_BEST_OPEN = not decided ahead of the parse (the parse searches)
#define LZ_HC_BINS_BYTES  (4u << LZ_HC_SEGLOG)
#define LZ_HC_LINKS_BYTES (4u << LZ_HC_SEGLOG)
#define LZ_HC_WINS_BYTES  8192u
#define LZ_HC_HEADS_BYTES(maxBlock) ((size_t)(maxBlock) > (1u << LZ_HC_SEGLOG) ? (size_t)(4u << LZ_HC_HASHLOG) : 0u)
#define LZ_HC_HITS_BYTES(maxBlock) ((((size_t)(maxBlock) + 63u) / 64u + 64u) * 8u)
#define LZ_HC_SLOT_BYTES(maxBlock) ((size_t)LZ_HC_BINS_BYTES + LZ_HC_LINKS_BYTES + LZ_HC_WINS_BYTES + LZ_HC_HEADS_BYTES(maxBlock) + LZ_HC_HITS_BYTES(maxBlock) + 10u * (size_t)(maxBlock) + 512u)
#define LZ_HC_BEST_OPEN 0xFFFFFFFFu

struct LzHc {
    u32* bins;          // global
    u32* links;         // global
    u32* wins;          // global
    u32* heads;         // global (multi-segment blocks)
    u64* hits;          // global: bit p = a search at p finds a match
    u16* prev;          // global: per block position, distance to the previous head of its bucket (0 = none)
    u32* chain2;        // global: prev[p] | (prev[p] + prev[p - prev[p]]) << 16 (upper half 0 = no second link inside any window)
    u32* best;          // global: the first search's result per position (see above)
    u32  searchNum;     // uniform
    bool pre;           // uniform: best[] was filled for this block
    bool noChain;       // uniform: the noChain parser's arbitration (levels 12 / 32 / 33; searchNum is 1 then)
};

template <int SEARCHLEN, int HLOG>
LZ_DEV u32 lz_hc_hash(u64 bytes)
{
    if constexpr (SEARCHLEN == 4) return ((u32)bytes * 2654435761u) >> (32 - HLOG);              // lizard_compress.c:87-88
    else return lz_hash5<HLOG>(bytes);                                                            // :90-91
}

// Binds a wave's slot (all lanes call).
LZ_DEV void lz_hc_begin(LzHc& hc, void* slotMem, u32 maxBlock, u32 searchNum)
{
    u8* m = (u8*)slotMem;
    hc.bins = (u32*)m;   m += LZ_HC_BINS_BYTES;
    hc.links = (u32*)m;  m += LZ_HC_LINKS_BYTES;
    hc.wins = (u32*)m;   m += LZ_HC_WINS_BYTES;
    hc.heads = (u32*)m;  m += LZ_HC_HEADS_BYTES(maxBlock);
    hc.hits = (u64*)m;   m += LZ_HC_HITS_BYTES(maxBlock);
    hc.prev = (u16*)m;   m += 2u * (size_t)maxBlock + 128u;
    hc.chain2 = (u32*)m; m += 4u * (size_t)maxBlock + 128u;
    hc.best = (u32*)m;
    hc.searchNum = searchNum;
    hc.pre = false;
    hc.noChain = false;
}

// Eight source bytes at each of the positions base + 64 k + lane, k = 0..7 (clamped inside the segment; unconditional).
#ifndef LZ_HC_GRP
#define LZ_HC_GRP 8u
#endif
#ifndef LZ_HC_QUEUE
#define LZ_HC_QUEUE 16u                          // pass 2: steps between the request of an entry and its use
#endif
LZ_DEV void lz_hc_load_src(const u8* sp, u32 base, u32 segLen, u64 (&x)[LZ_HC_GRP])
{
    #pragma unroll
    for (u32 k = 0; k < LZ_HC_GRP; k++) {
        const u32 p = base + k * 64u + lz_lane();
        x[k] = lz_ld64(sp + (p < segLen ? p : 0u));
    }
}

// 256 words of `links` from lo, four per lane (lane l: lo + 4 l ..; unconditional: words past the range belong to the next
// bin or, at the very end, to the slot's next area, and are ignored).
struct __attribute__((packed, aligned(4))) lz_u128a4 { u32 x, y, z, w; };
LZ_DEV lz_u128a4 lz_hc_load_links(const LzHc& hc, u32 lo)
{
    return *(const lz_u128a4*)(hc.links + lo + 4u * lz_lane());
}

// One step of pass 2: the valid lanes (ascending positions, one bin) read the head of their bucket and become the head —
// Lizard_Insert's ":38" — through one returning exchange; returns the lane's link (":27-31").  `arena` = the bin's table,
// slot 2^SUBLOG is a spare for the lanes that sit out.
template <int SUBLOG>
LZ_DEV u32 lz_hc_visit(u32* arena, bool valid, u32 hl, u32 p)
{
    const u32 lane = lz_lane();
    const u32 old = lz_lds_xchg_rtn(arena + (valid ? hl : (1u << SUBLOG)), p);
    u32 seen = old;                                              // head as my own insertion sees it
    u64 pend = lz_ballot(valid && p - old < LZ_MIN_OFFSET);
    if (pend) {                                                  // ":38" did not happen for some lane: replay its bucket in order
        while (pend) {
            const u32 hv = lz_readlane(hl, lz_ctz64(pend));
            const bool mine = valid && hl == hv;
            const u64 g = lz_ballot(mine);
            u32 t = lz_readlane(old, lz_ctz64(g));               // the bucket's head before this step (uniform)
            for (u64 m = g; m; m &= m - 1ull) {
                const u32 k = lz_ctz64(m), pk = lz_readlane(p, k);
                if (lane == k) seen = t;
                t = (pk - t >= LZ_MIN_OFFSET) ? pk : t;
            }
            if (mine && lane == 63u - lz_clz64(g)) arena[hl] = t;
            pend &= ~g;
        }
        lz_lds_sync();
    }
    const u32 d = p - seen;
    return d <= LZ_MAX_DIST_LZ4 ? d : 0u;
}

// Phase A: Lizard_Insert (hashchain.h:13-43) for every position of the block that has 8 readable bytes:
//     prev[p] = p - head[h(p)]  (0 when farther than 65535, :27-31);   if (p - head[h(p)] >= MIN_OFFSET) head[h(p)] = p  (:38)
// in position order.  The head table (2^18 slots) does not fit LDS and a table in global memory costs a 128-byte fabric read
// and a 32-byte fabric write per position, so the positions are first brought into an order in which a small table is enough:
//   pass 0  histogram of the top 6 hash bits (LDS counters);
//   pass 1  stable partition into 64 bins: ONE returning LDS add per step hands every lane its slot in its bin — lanes of a
//           DS atomic are served in lane order (lz_wave.h), so a bin keeps position order; the entries go straight to their
//           slots (a bin's 128-byte line is complete ~32 steps later and is merged in L2); the cursors at every
//           2^14-position boundary are noted;
//   pass 2  bin after bin against a 2^12-slot head table in LDS: ONE returning LDS exchange per step is "read the head, become
//           the head" in position order.  That is :38 whenever the distance is >= MIN_OFFSET; a step in which some lane finds
//           its head closer than that (runs) replays the affected buckets with scalar code, exactly, before the next step.
//           The link goes to `links`, in bin order;
//   pass 3  windows of 2^14 positions: every bin's entries of the window (a contiguous, known range) drop their link into an
//           LDS window at their position; the window leaves as prev[], coalesced.
// Blocks above 2^18 positions run segment after segment; the head tables travel through `heads` in between.
// The LDS region (32.3 KiB) is borrowed from the workgroup's pool for the duration of the build.
template <int SEARCHLEN, int HLOG>
LZ_DEV void lz_hc_build(const u8* src, u32 n, const LzHc& hc, const LzHufPool& pool, LzStreams& st)
{
    static_assert(HLOG > LZ_HC_BINLOG && HLOG <= LZ_HC_HASHLOG, "64 bins of 2^(HLOG - 6) slots; the areas are sized for 2^18");
    constexpr int kSubLog = HLOG - LZ_HC_BINLOG;
    const u32 lane = lz_lane();
    const u32 nIns = n >= 8u ? n - 7u : 0u;
    if (!nIns) return;
    u32 poolSlot;
    u32* const arena = lz_pool_acquire(pool, poolSlot);
    LZ_PROF(st, 7);                                              // (instrumented build) waiting for a region
    u32* const cur = arena + LZ_HC_ARENA_WORDS;                  // [0..63] cursors, [64] spare
    u16* const win16 = (u16*)arena;
    constexpr u32 kSeg = 1u << LZ_HC_SEGLOG, kSub = 1u << kSubLog, kWin = 1u << LZ_HC_WINLOG;
    for (u32 seg = 0; seg < nIns; seg += kSeg) {
        const u32 segLen = (nIns - seg) < kSeg ? (nIns - seg) : kSeg;
        const bool firstSeg = seg == 0u, lastSeg = seg + segLen >= nIns;
        const u8* const sp = src + seg;
        // Source bytes are requested one group of 8 steps (512 positions) ahead of their use: the passes are chains of LDS
        // trips, and a memory round trip per step would be all they wait for.
        // ---- pass 0: bin sizes ----
        cur[lane] = 0u; if (lane == 0) cur[64] = 0u;
        lz_lds_sync();
        {
            u64 x[LZ_HC_GRP], y[LZ_HC_GRP];
            lz_hc_load_src(sp, 0u, segLen, x);
            for (u32 base = 0; base < segLen; base += 64u * LZ_HC_GRP) {
                lz_hc_load_src(sp, base + 64u * LZ_HC_GRP, segLen, y);
                #pragma unroll
                for (u32 k = 0; k < LZ_HC_GRP; k++) {
                    const u32 p = base + k * 64u + lane;
                    const u32 h = lz_hc_hash<SEARCHLEN, HLOG>(x[k]);
                    lz_lds_atomic_add(cur + (p < segLen ? h >> kSubLog : 64u), 1u);
                    x[k] = y[k];
                }
            }
        }
        lz_lds_sync();
        LZ_PROF(st, 11);                                         // pass 0
        const u32 cnt = cur[lane];
        const u32 myStart = lz_wave_scan_excl_add(cnt), myEnd = myStart + cnt;     // lane b: range of bin b in `bins`
        lz_lds_sync();
        cur[lane] = myStart;
        hc.wins[lane] = myStart;
        lz_lds_sync();
        // ---- pass 1: stable partition ----
        {
            u64 x[LZ_HC_GRP], y[LZ_HC_GRP];
            lz_hc_load_src(sp, 0u, segLen, x);
            for (u32 base4 = 0; base4 < segLen; base4 += 64u * LZ_HC_GRP) {
                lz_hc_load_src(sp, base4 + 64u * LZ_HC_GRP, segLen, y);
                #pragma unroll
                for (u32 k = 0; k < LZ_HC_GRP; k++) {
                    const u32 base = base4 + k * 64u;
                    if (base < segLen) {                             // uniform
                        if (base && !(base & (kWin - 1u))) hc.wins[(base >> LZ_HC_WINLOG) * 64u + lane] = cur[lane];
                        const u32 p = base + lane;
                        const bool valid = p < segLen;
                        const u32 h = lz_hc_hash<SEARCHLEN, HLOG>(x[k]);
                        const u32 bin = valid ? h >> kSubLog : 64u;
                        const u32 s = lz_lds_add_rtn(cur + bin, 1u);
                        if (valid) hc.bins[s] = ((h & (kSub - 1u)) << LZ_HC_SEGLOG) | p;    // a bin's line fills within ~32 steps: merged in L2
                    }
                    x[k] = y[k];
                }
            }
        }
        hc.wins[((segLen + kWin - 1u) >> LZ_HC_WINLOG) * 64u + lane] = myEnd;
        lz_wave_sync();                                          // bins are in memory; the arena changes hands
        LZ_PROF(st, 12);                                         // pass 1
        // ---- pass 2: heads and links, bin after bin ----
        // The entries of all bins lie back to back, so they are read on a flat grid of 64 (sixteen steps ahead); a step that
        // straddles a bin border is served part by part, with the table changing hands in between.
        {
            const u32 nSteps = (segLen + 63u) >> 6;
            u32 b = 0, bEnd = lz_readlane(myEnd, 0);                 // current bin and its end (uniform)
            bool fresh = true;                                       // the current bin's table is not set up yet
            u32 q[LZ_HC_QUEUE];                                      // entries of the next 16 steps
            #pragma unroll
            for (u32 k = 0; k < LZ_HC_QUEUE; k++) { const u32 nx = k * 64u + lane; q[k] = hc.bins[nx < segLen ? nx : 0u]; }
            for (u32 j8 = 0; j8 < nSteps; j8 += LZ_HC_QUEUE)
            #pragma unroll
            for (u32 k = 0; k < LZ_HC_QUEUE; k++) {
                const u32 j = j8 + k;
                if (j < nSteps) {                                    // uniform
                const u32 e0 = q[k];
                { const u32 nx = (j + LZ_HC_QUEUE) * 64u + lane; q[k] = hc.bins[nx < segLen ? nx : 0u]; }
                const u32 idx = j * 64u + lane, stepEnd = (j + 1u) * 64u < segLen ? (j + 1u) * 64u : segLen;
                const u32 hl = e0 >> LZ_HC_SEGLOG, p = seg + (e0 & (kSeg - 1u));
                u32 link;
                if (!fresh && bEnd >= stepEnd)                       // the whole step lies in the current bin: straight-line code
                    link = lz_hc_visit<kSubLog>(arena, idx < stepEnd, hl, p);
                else {
                    u32 from = j * 64u;                              // first entry of this step not served yet (uniform)
                    link = 0;
                    while (from < stepEnd) {
                        while (bEnd <= from) {                       // leave bins that end here (also empty ones)
                            if (!lastSeg) {
                                if (fresh) { uint4 none; none.x = none.y = none.z = none.w = LZ_HC_NONE;
                                             if (firstSeg) for (u32 i = lane * 4u; i < kSub; i += 256u) *(uint4*)(hc.heads + b * kSub + i) = none; }
                                else { lz_lds_sync(); for (u32 i = lane * 4u; i < kSub; i += 256u) *(uint4*)(hc.heads + b * kSub + i) = *(const uint4*)(arena + i); }
                            }
                            b++; bEnd = lz_readlane(myEnd, b & 63u); fresh = true;
                        }
                        if (fresh) {
                            lz_lds_sync();                           // the previous table's reads are done
                            if (firstSeg) { uint4 none; none.x = none.y = none.z = none.w = LZ_HC_NONE;
                                            for (u32 i = lane * 4u; i < kSub; i += 256u) *(uint4*)(arena + i) = none; }
                            else for (u32 i = lane * 4u; i < kSub; i += 256u) *(uint4*)(arena + i) = *(const uint4*)(hc.heads + b * kSub + i);
                            lz_lds_sync();
                            fresh = false;
                        }
                        const u32 to = bEnd < stepEnd ? bEnd : stepEnd;
                        const bool valid = idx >= from && idx < to;
                        const u32 l = lz_hc_visit<kSubLog>(arena, valid, hl, p);
                        if (valid) link = l;
                        from = to;
                    }
                }
                if (idx < segLen) hc.links[idx] = (link << 16) | (e0 & (kWin - 1u));     // link | position in its window
                }
            }
            if (!lastSeg) {                                          // the bins from the current one on
                for (;;) {
                    if (fresh) { uint4 none; none.x = none.y = none.z = none.w = LZ_HC_NONE;
                                 if (firstSeg) for (u32 i = lane * 4u; i < kSub; i += 256u) *(uint4*)(hc.heads + b * kSub + i) = none; }
                    else { lz_lds_sync(); for (u32 i = lane * 4u; i < kSub; i += 256u) *(uint4*)(hc.heads + b * kSub + i) = *(const uint4*)(arena + i); }
                    if (++b >= 64u) break;
                    fresh = true;
                }
            }
            lz_lds_sync();
        }
        lz_wave_sync();                                          // links (and wins) are in memory
        LZ_PROF(st, 13);                                         // pass 2
        // ---- pass 3: back to position order ----
        const u32 nWin = (segLen + kWin - 1u) >> LZ_HC_WINLOG;
        for (u32 w = 0; w < nWin; w++) {
            const u32 wLo = hc.wins[w * 64u + lane], wHi = hc.wins[(w + 1u) * 64u + lane];
            const u32 wBase = w << LZ_HC_WINLOG;
            lz_u128a4 q[LZ_HC_GRP];                                  // the next 8 bins' words, in flight
            u64 more = 0;                                            // bins with more than 256 entries in this window (uniform)
            #pragma unroll
            for (u32 k = 0; k < LZ_HC_GRP; k++) q[k] = lz_hc_load_links(hc, lz_readlane(wLo, k));
            for (u32 b8 = 0; b8 < 64u; b8 += LZ_HC_GRP)
            #pragma unroll
            for (u32 k = 0; k < LZ_HC_GRP; k++) {
                const u32 b = b8 + k;
                const u32 lo = lz_readlane(wLo, b), hi = lz_readlane(wHi, b);
                lz_u128a4 v = q[k];
                q[k] = lz_hc_load_links(hc, lz_readlane(wLo, (b + LZ_HC_GRP) & 63u));
                {   // straight-line: a loop here makes the compiler wait for every load in flight
                    const u32 idx = lo + 4u * lane;
                    if (idx < hi)      win16[v.x & (kWin - 1u)] = (u16)(v.x >> 16);
                    if (idx + 1u < hi) win16[v.y & (kWin - 1u)] = (u16)(v.y >> 16);
                    if (idx + 2u < hi) win16[v.z & (kWin - 1u)] = (u16)(v.z >> 16);
                    if (idx + 3u < hi) win16[v.w & (kWin - 1u)] = (u16)(v.w >> 16);
                }
                if (hi - lo > 256u) more |= 1ull << b;               // a bin with more than 256 entries in the window: below
            }
            while (more) {                                           // skewed data only
                const u32 b = lz_ctz64(more); more &= more - 1ull;
                const u32 lo = lz_readlane(wLo, b), hi = lz_readlane(wHi, b);
                for (u32 i = lo + 256u; i < hi; i += 256u) {
                    const u32 idx = i + 4u * lane;
                    const lz_u128a4 v = lz_hc_load_links(hc, i);
                    if (idx < hi)      win16[v.x & (kWin - 1u)] = (u16)(v.x >> 16);
                    if (idx + 1u < hi) win16[v.y & (kWin - 1u)] = (u16)(v.y >> 16);
                    if (idx + 2u < hi) win16[v.z & (kWin - 1u)] = (u16)(v.z >> 16);
                    if (idx + 3u < hi) win16[v.w & (kWin - 1u)] = (u16)(v.w >> 16);
                }
            }
            lz_lds_sync();
            const u32 wLen = (segLen - wBase) < kWin ? (segLen - wBase) : kWin;
            for (u32 o = lane * 8u; o < wLen; o += 512u)        // 16 bytes per lane; the tail may carry stale cells past nIns (never read)
                *(uint4*)(hc.prev + seg + wBase + o) = *(const uint4*)(win16 + o);
            lz_lds_sync();
        }
        lz_wave_sync();
        LZ_PROF(st, 14);                                         // pass 3
    }
    lz_pool_release(pool, poolSlot);
}

// The parse's outer loop (hashchain.h:204-206) advances position by position until Lizard_InsertAndFindBestMatch finds
// anything at all: "one of the first searchNum chain candidates of p lies inside the window, at least MIN_OFFSET back, and
// agrees with p in 4 bytes" (:66-73 make ml >= 4 > 0 exactly then).  That is a function of p alone once the chain exists, so it
// is evaluated for all positions of the block ahead of the parse, LZ_HC_BULK x 64 positions at a time with their memory trips
// side by side, into one bit per position; the outer loop becomes a bit scan.  (The parse used to walk the chains of 64
// positions per round on its own serial path: three dependent memory trips per sequence.)
#ifndef LZ_HC_BULK_PLAIN
#define LZ_HC_BULK_PLAIN 8u
#endif
// hit bits and packed chain words only (levels 13-15 / 34-36: the pass below does not pay there)
LZ_DEV void lz_hc_hits_plain(const u8* src, u32 n, const LzHc& hc)
{
    const u32 lane = lz_lane();
    const u32 nIns = n >= 8u ? n - 7u : 0u;                      // positions that have a chain link (lz_hc_build)
    for (u32 base = 0; base < nIns; base += 64u * LZ_HC_BULK_PLAIN) {
        u32 p[LZ_HC_BULK_PLAIN], m[LZ_HC_BULK_PLAIN], d[LZ_HC_BULK_PLAIN], f4[LZ_HC_BULK_PLAIN], two[LZ_HC_BULK_PLAIN];
        bool walking[LZ_HC_BULK_PLAIN], hit[LZ_HC_BULK_PLAIN];
        #pragma unroll
        for (u32 k = 0; k < LZ_HC_BULK_PLAIN; k++) {
            p[k] = base + k * 64u + lane;
            walking[k] = p[k] < nIns; hit[k] = false;
            m[k] = walking[k] ? p[k] : 0u;
            f4[k] = lz_ld32(src + m[k]); d[k] = hc.prev[m[k]];
            two[k] = d[k];                                       // becomes prev | two links << 16
        }
        for (u32 a = 0; a < hc.searchNum; a++) {
            bool any = false;
            #pragma unroll
            for (u32 k = 0; k < LZ_HC_BULK_PLAIN; k++) {
                walking[k] = walking[k] && d[k] != 0u && p[k] - (m[k] - d[k]) <= LZ_MAX_DIST_LZ4;
                any = any || walking[k];
                m[k] = walking[k] ? m[k] - d[k] : m[k];
            }
            if (!lz_ballot(any)) break;
            u32 c4[LZ_HC_BULK_PLAIN];
            #pragma unroll
            for (u32 k = 0; k < LZ_HC_BULK_PLAIN; k++) { c4[k] = lz_ld32(src + m[k]); d[k] = hc.prev[m[k]]; }
            if (a == 0u) {                                       // the second link, for lz_hc_search's two-at-a-time walk
                #pragma unroll
                for (u32 k = 0; k < LZ_HC_BULK_PLAIN; k++) {
                    const u32 sum = (p[k] - m[k]) + d[k];
                    if (walking[k] && d[k] != 0u && sum <= LZ_MAX_DIST_LZ4) two[k] |= sum << 16;
                }
            }
            #pragma unroll
            for (u32 k = 0; k < LZ_HC_BULK_PLAIN; k++)
                if (walking[k] && p[k] - m[k] >= LZ_MIN_OFFSET && c4[k] == f4[k]) { hit[k] = true; walking[k] = false; }
        }
        u64 mine = 0;
        #pragma unroll
        for (u32 k = 0; k < LZ_HC_BULK_PLAIN; k++) {
            const u64 w = lz_ballot(hit[k]); mine = lane == k ? w : mine;
            if (p[k] < nIns) hc.chain2[p[k]] = two[k];
        }
        if (lane < LZ_HC_BULK_PLAIN && base + lane * 64u < nIns) hc.hits[(base >> 6) + lane] = mine;
    }
    lz_wave_sync();
}

//
// Round 4: the FIRST search of a sequence (Lizard_InsertAndFindBestMatch, hashchain.h:45-107: longest forward match among the first
// searchNum chain candidates, the earlier one on ties) is a function of the position alone as well — only the two "wider" searches of
// the arbitration (:236-300) depend on the running match.  After the wider searches at positions without a hit were dropped (below),
// first searches are 70-85 % of the searches the parse still runs, each three or four dependent memory trips on the wave's serial
// path.  This pass therefore goes on past the first hit: every candidate brings its chain link AND 16 bytes in one trip, the
// position's own 16 bytes are in registers, so the 4-byte test (:73) and the forward length (:76) of a candidate come out of that trip,
// LZ_HC_BULK x 64 positions side by side.  best[p] = length | offset << 16.  Left to the parse (LZ_HC_BEST_OPEN): a candidate that
// agrees in all 16 bytes with more room behind them (5-18 % of the first searches on the bench data), and positions with more than
// LZ_HC_PRE_STEPS candidates (their hit bit is still exact: the walk goes on with the 4-byte test alone until the first hit).
#ifndef LZ_HC_BULK
#define LZ_HC_BULK 4u
#endif
#ifndef LZ_HC_PRE_STEPS
#define LZ_HC_PRE_STEPS 4u
#endif
#ifndef LZ_HC_PREPASS
#define LZ_HC_PREPASS 1
#endif

LZ_DEV void lz_hc_hits(const u8* src, u32 n, const LzHc& hc)
{
    const u32 lane = lz_lane();
    const u32 nIns = n >= 8u ? n - 7u : 0u;                      // positions that have a chain link (lz_hc_build)
    for (u32 base = 0; base < nIns; base += 64u * LZ_HC_BULK) {
        u32 m[LZ_HC_BULK], d[LZ_HC_BULK], two[LZ_HC_BULK], best[LZ_HC_BULK];
        u64 pA[LZ_HC_BULK], pB[LZ_HC_BULK];
        u32 state[LZ_HC_BULK];                                   // bit 0 walking, bit 1 hit, bit 2 open
        #pragma unroll
        for (u32 k = 0; k < LZ_HC_BULK; k++) {
            const u32 p = base + k * 64u + lane;
            state[k] = p < nIns ? 1u : 0u; best[k] = 0u;
            m[k] = p < nIns ? p : 0u;
            d[k] = hc.prev[m[k]];
            two[k] = d[k];                                       // becomes prev | two links << 16
            // (16 bytes at a position below nIns = n - 7 may reach past the block by up to 8: the second load is clamped)
            pA[k] = lz_ld64(src + m[k]);
            pB[k] = lz_ld64(src + (m[k] + 16u <= n ? m[k] + 8u : m[k]));
        }
        for (u32 a = 0; a < hc.searchNum; a++) {
            bool any = false;
            #pragma unroll
            for (u32 k = 0; k < LZ_HC_BULK; k++) {
                const u32 p = base + k * 64u + lane;
                const bool w = (state[k] & 1u) && d[k] != 0u && p - (m[k] - d[k]) <= LZ_MAX_DIST_LZ4;
                state[k] = (state[k] & ~1u) | (w ? 1u : 0u);
                any = any || w;
                m[k] = w ? m[k] - d[k] : m[k];
            }
            if (!lz_ballot(any)) break;
            const bool full = LZ_HC_PREPASS && a < LZ_HC_PRE_STEPS;          // uniform: measure the candidates of this step
            u64 cA[LZ_HC_BULK], cB[LZ_HC_BULK];
            #pragma unroll
            for (u32 k = 0; k < LZ_HC_BULK; k++) {                   // (a candidate lies below its position: its 16 bytes are inside the block when the position's are)
                const u32 p = base + k * 64u + lane;
                if (full) { cA[k] = lz_ld64(src + m[k]); cB[k] = lz_ld64(src + (p + 16u <= n ? m[k] + 8u : m[k])); }
                else      { cA[k] = lz_ld32(src + m[k]); cB[k] = 0; }
                d[k] = hc.prev[m[k]];
            }
            if (a == 0u) {                                       // the second link, for lz_hc_search's two-at-a-time walk
                #pragma unroll
                for (u32 k = 0; k < LZ_HC_BULK; k++) {
                    const u32 p = base + k * 64u + lane;
                    const u32 sum = (p - m[k]) + d[k];
                    if ((state[k] & 1u) && d[k] != 0u && sum <= LZ_MAX_DIST_LZ4) two[k] |= sum << 16;
                }
            }
            #pragma unroll
            for (u32 k = 0; k < LZ_HC_BULK; k++) {
                const u32 p = base + k * 64u + lane;
                const bool walking = state[k] & 1u;
                const bool ok = walking && p - m[k] >= LZ_MIN_OFFSET && (u32)cA[k] == (u32)pA[k];             // :73
                if (full) {
                    if (ok) {
                        // the sub-block the position belongs to bounds its matches (matchlimit = E - LASTLITERALS, lz_parse_hashchain);
                        // a position the parse never searches from (>= mflimit) gets room 0
                        const u32 S0 = p & ~(LZ_SUBBLOCK - 1u);
                        const u32 E = S0 + LZ_SUBBLOCK < n ? S0 + LZ_SUBBLOCK : n;
                        const u32 maxFwd = (E >= LZ_MFLIMIT && p < E - LZ_MFLIMIT) ? E - LZ_LASTLITERALS - p : 0u;
                        const u64 x = pA[k] ^ cA[k], y = pB[k] ^ cB[k];
                        const u32 seen = p + 16u <= n ? 16u : 8u;    // bytes the loads may compare (the second was clamped near the end of the block)
                        const u32 common = x ? lz_ctz64(x) >> 3 : (seen > 8u && y) ? 8u + (lz_ctz64(y) >> 3) : seen;
                        const u32 mlt = common < maxFwd ? common : maxFwd;
                        state[k] |= 2u;
                        if (common >= seen && maxFwd > seen) state[k] = (state[k] | 4u) & ~1u;       // longer than what was fetched: the parse decides
                        else if (mlt > (best[k] & 0xFFFFu)) best[k] = mlt | ((p - m[k]) << 16);     // strictly longer: the earlier candidate wins ties
                    }
                } else if (ok) {
                    state[k] = (state[k] | 2u | (LZ_HC_PREPASS ? 4u : 0u)) & ~1u;                      // found by the 4-byte test alone: not measured
                } else if (LZ_HC_PREPASS && walking && (state[k] & 2u)) state[k] = (state[k] | 4u) & ~1u;   // a hit is known; what is left could beat it
            }
        }
        u64 mine = 0;
        #pragma unroll
        for (u32 k = 0; k < LZ_HC_BULK; k++) {
            const u32 p = base + k * 64u + lane;
            const u64 w = lz_ballot((state[k] & 2u) != 0u); mine = lane == k ? w : mine;
            if (p < nIns) {
                hc.chain2[p] = two[k];
                hc.best[p] = (!LZ_HC_PREPASS || (state[k] & 4u)) ? LZ_HC_BEST_OPEN : best[k];
            }
        }
        if (lane < LZ_HC_BULK && base + lane * 64u < nIns) hc.hits[(base >> 6) + lane] = mine;
    }
    lz_wave_sync();
}

// One search of the chain that starts at X (uniform).  wider == false: Lizard_InsertAndFindBestMatch
// (longest = 0 on entry, iLow unused); wider == true: Lizard_InsertAndGetWiderMatch with backward
// extension down to iLow.  Returns the new longest; ref/start change only when it grew.
//
// Up to 64 chain candidates are measured at once, one per lane: 4-byte test, then 24 more bytes forward
// and 8 bytes backward with lane-local loads.  Lanes whose comparison is still open after that (long
// matches) are finished one at a time with the wave-wide helpers, skipping those that cannot reach the
// best length any more.  "First strictly longer candidate in chain order" == maximum length, lowest lane.
LZ_DEV u32 lz_hc_search(const u8* src, u32 nBlock, const LzHc& hc, u32 X, u32 iLow, u32 iHigh, u32 longest, bool wider, u32& ref, u32& start, LzStreams& st)
{
    const u32 lane = lz_lane();
    const u32 first4 = lz_ld32(src + X);
    const u32 maxFwd = iHigh - X;                                // X + 4 < iHigh for every caller
    u32 m = X, left = hc.searchNum;                              // uniform
    bool more = true;
    while (more && left) {
        const u32 batch = left < 64u ? left : 64u;
        u32 cand = 0, cnt = 0, c4 = 0;                           // c4: the candidate's 4 test bytes, requested as soon as it is known
        for (u32 j = 0; j < batch; j += 2u) {                    // the walk is serial; one word brings two links
            const u32 w2 = lz_uniform(hc.chain2[m]);
            const u32 d1 = w2 & 0xFFFFu, d2 = w2 >> 16;
            if (!d1) { more = false; break; }
            const u32 m1 = m - d1;
            if (X - m1 > LZ_MAX_DIST_LZ4) { more = false; break; }
            if (lane == j) { cand = m1; c4 = lz_ld32(src + m1); }
            cnt++;
            if (j + 1u >= batch) { m = m1; break; }
            if (!d2) { more = false; break; }                    // no second link, or farther than any window reaches
            const u32 m2 = m - d2;
            if (X - m2 > LZ_MAX_DIST_LZ4) { more = false; break; }
            if (lane == j + 1u) { cand = m2; c4 = lz_ld32(src + m2); }
            cnt++;
            m = m2;
        }
        left -= cnt;
        LZ_PROF(st, 8);                                          // (instrumented build) chain walk
        // The 4-byte test first (its bytes were requested during the walk): few candidates pass it, and what they read next
        // lies in the line the test fetched.
        u32 mlt = 0, bk = 0;
        bool open = false;                                       // my comparison needs the wave-wide helpers
        const bool ok = lane < cnt && X - cand >= LZ_MIN_OFFSET && c4 == first4;                 // :73 / :146
        if (ok) {
            u32 f = 4u;
            bool eq = true;
            #pragma unroll
            for (u32 it = 0; it < 3u; it++) {
                if (eq && f < maxFwd) {
                    const u64 x = lz_ld64(src + X + f) ^ lz_ld64(src + cand + f);
                    u32 c8 = x ? lz_ctz64(x) >> 3 : 8u;
                    const u32 room = maxFwd - f;
                    c8 = c8 < room ? c8 : room;
                    f += c8; eq = c8 == 8u;
                } else eq = false;
            }
            open = eq;
            if (wider) {
                const u32 lim = (X - iLow) < cand ? (X - iLow) : cand;        // :150 both bounds
                if (lim) {
                    if (cand >= 8u) {                                          // X > cand >= 8
                        const u64 x = lz_ld64(src + X - 8u) ^ lz_ld64(src + cand - 8u);
                        const u32 e8 = x ? lz_clz64(x) >> 3 : 8u;
                        bk = e8 < lim ? e8 : lim;
                        open = open || (e8 == 8u && lim > 8u);
                    } else open = true;
                }
            }
            mlt = f + bk;
        }
        LZ_PROF(st, 9);                                          // per-lane measurement
        // best of the lanes that are already exact: max length, lowest lane on ties
        const u32 key = (ok && !open) ? (mlt << 6) | (63u - lane) : 0u;
        const u32 top = lz_readlane(lz_wave_reduce_max(key), 63u);
        u32 bestMl = top >> 6, bestLane = 63u - (top & 63u), bestBack = 0;
        bool bestOpen = false;
        u64 om = lz_ballot(open);
        while (om) {
            const u32 j = lz_ctz64(om);
            om &= om - 1ull;
            const u32 c = lz_readlane(cand, j);
            const u32 cap = maxFwd + (wider ? ((X - iLow) < c ? (X - iLow) : c) : 0u);
            const u32 need = bestMl > longest ? bestMl : longest + 1u;         // what j must reach to matter
            if (cap < need || (cap == bestMl && bestMl > 0u && j > bestLane)) continue;
            u32 mj = 4u + lz_count_fwd(src, X + 4u, c + 4u, iHigh);
            u32 bj = 0;
            if (wider) { bj = lz_count_back(src, X, c, iLow); mj += bj; }      // :150-152
            if (mj > bestMl || (mj == bestMl && j < bestLane)) { bestMl = mj; bestLane = j; bestBack = bj; bestOpen = true; }
        }
        LZ_PROF(st, 10);                                         // selection + wave-wide leftovers
        if (bestMl > longest) {
            const u32 c = lz_readlane(cand, bestLane);
            const u32 back = bestOpen ? bestBack : lz_readlane(bk, bestLane);
            longest = bestMl; ref = c - back; start = X - back;
        }
    }
    return longest;
}

// One search of a noChain level (Lizard_InsertAndFindBestMatchNoChain nochain.h:28-80, wider == false; Lizard_InsertAndGetWiderMatchNoChain
// :83-143, wider == true): ONE candidate, the head of X's bucket = X - prev[X], and every caller has read X's hit bit first — the
// candidate lies inside the window, at least MIN_OFFSET back, and agrees with X in 4 bytes.  Nothing is left to test before the
// lengths are measured, so forward count and backward extension come out of ONE memory trip (lz_count_both) where the general search
// spends three in a row (chain word, 4-byte test, measurement); cw = X's chain word, out of the parse's register window when X lies
// in it.  The guards below only matter for a caller without the hit bit (LZ_HC_SKIP_NOHIT 0 builds).
LZ_DEV u32 lz_nc_search(const u8* src, u32 cw, u32 X, u32 iLow, u32 iHigh, u32 longest, bool wider, u32& ref, u32& start)
{
    const u32 d1 = cw & 0xFFFFu;
    if (d1 < LZ_MIN_OFFSET) return longest;                      // no head inside the window (0; :49 / :103), or closer than MIN_OFFSET (:53 / :107)
    const u32 m = X - d1;
    u32 f, b;
    lz_count_both(src, X, m, iHigh, wider ? iLow : X, f, b);     // (anchor X: the first search does not extend backwards)
    if (f < 4u) return longest;                                  // :55 / :110
    if (f + b > longest) { longest = f + b; ref = m - b; start = X - b; }      // :57-58 / :112-120
    return longest;
}

// One search at searchNum NC = 2 / 4 / 8 (levels 13 / 14 / 15 and 34 / 35 / 36; hashchain.h:45-107 / :109-185 with NC attempts): the
// candidates come out of packed chain words (prev | two links at once << 16: two candidates per word; the first word is X's, out of
// the parse's register window), and all of them are measured side by side — 64 / NC lanes each, 8 bytes forward and one byte backward
// per lane — in ONE memory trip behind the walk's NC / 2 - 1 dependent words, where the general search walks, tests 4 bytes and measures
// one after the other.  The hit bit (read first by every caller) says that at least one of them passes the 4-byte test; the others
// cost loads that run beside it, not trips.  "First strictly longer in chain order" = candidates in order, strictly longer wins.
template <int NC>
LZ_DEV u32 lz_hcN_search(const u8* src, const LzHc& hc, u32 cw, u32 X, u32 iLow, u32 iHigh, u32 longest, bool wider, u32& ref, u32& start)
{
    static_assert(NC == 2 || NC == 4 || NC == 8, "two candidates per chain word; 64 / NC lanes per candidate");
    constexpr u32 G = 64u / NC;                                  // lanes per candidate
    const u32 lane = lz_lane(), k = lane / G, li = lane % G;
    // the walk (uniform): distance of candidate c from X, 0 = none (and none behind it): hashchain.h:70 / :100-102 per step
    u32 dist[NC];
    {
        u32 base = 0, w = cw;
        bool more = true;
        #pragma unroll
        for (u32 c = 0; c < (u32)NC; c += 2u) {
            u32 a = 0, b2 = 0;
            if (more) {
                const u32 e1 = w & 0xFFFFu, e2 = w >> 16;
                if (e1 && base + e1 <= LZ_MAX_DIST_LZ4) { a = base + e1; if (e2 && base + e2 <= LZ_MAX_DIST_LZ4) b2 = base + e2; }
                more = b2 != 0u;
                if (c + 2u < (u32)NC && more) { w = lz_uniform(hc.chain2[X - b2]); base = b2; }
            }
            dist[c] = a; dist[c + 1u] = b2;
        }
    }
    if (!dist[0]) return longest;                                // no head inside the window
    u32 d = 0;
    #pragma unroll
    for (u32 c = 0; c < (u32)NC; c++) d = k == c ? dist[c] : d;
    const bool live = d >= LZ_MIN_OFFSET;                        // :75 / :143 (a missing candidate has d = 0)
    const u32 m = X - (live ? d : 0u);
    const u32 i = 8u * li, j = li + 1u;
    const bool inF = live && X + i < iHigh, inB = live && wider && X >= iLow + j && m >= j;
    const u64 x = lz_ld64(src + (inF ? X + i : X)) ^ lz_ld64(src + (inF ? m + i : X));
    const u32 bp = src[inB ? X - j : X], bm = src[inB ? m - j : X];
    u32 cq = 0;
    if (inF) { const u32 room = iHigh - (X + i); cq = x ? lz_ctz64(x) >> 3 : 8u; cq = cq < room ? cq : room; }
    lz_converge();
    const u64 stop = lz_ballot(cq < 8u), ne = lz_ballot(!(inB && bp == bm));
    #pragma unroll
    for (u32 c = 0; c < (u32)NC; c++) {                          // chain order
        const u32 dc = dist[c];
        if (dc < LZ_MIN_OFFSET) continue;                        // uniform
        const u32 mc = X - dc;
        const u32 sf = (u32)(stop >> (G * c)) & (u32)((1ull << G) - 1ull), nf = (u32)(ne >> (G * c)) & (u32)((1ull << G) - 1ull);
        u32 f, b;
        if (sf) { const u32 t = (u32)__builtin_ctz(sf); f = 8u * t + lz_readlane(cq, G * c + t); }
        else f = 8u * G + lz_count_fwd(src, X + 8u * G, mc + 8u * G, iHigh);
        if (f < 4u) continue;                                    // :73 / :146
        if (nf) b = (u32)__builtin_ctz(nf);
        else b = G + lz_count_back(src, X - G, mc - G, iLow);
        if (f + b > longest) { longest = f + b; ref = mc - b; start = X - b; }      // :76-80 / :150-158
    }
    return longest;
}

// The hit bit of position X (uniform): out of the 4 096 positions of bits the outer loop holds in registers when X lies there,
// else one word from memory.  A search at a position whose bit is clear — the first one or a "wider" one (hashchain.h:45-107,
// :109-186: both test the 4 bytes AT the position, :73 / :146) — finds nothing: no walk, no candidate bytes, no memory trip.
#ifndef LZ_HC_SKIP_NOHIT
#define LZ_HC_SKIP_NOHIT 1
#endif
LZ_DEV bool lz_hc_hit_at(const LzHc& hc, u64 bm, u32 bmBase, u32 X)
{
    const u32 wi = X >> 6, li = wi - bmBase;
    u64 w;
    if (li < 64u) w = ((u64)lz_readlane((u32)(bm >> 32), li) << 32) | lz_readlane((u32)bm, li);
    else          w = lz_uniform64(hc.hits[wi]);
    return (w >> (X & 63u)) & 1ull;
}

// Sub-block [S,E) of the block at src (hashchain.h:188-369).  The chain must have been built for the block.
// NCAND 1: the kernels of levels 12 / 32 / 33 (nochain.h:146-318: one candidate per search, lz_nc_search); NCAND 2 / 4 / 8: levels
// 13 / 14 / 15 and twins (searchNum = NCAND: the candidates measured side by side, lz_hcN_search); NCAND 0: the chain walk of
// lz_hc_search (levels 16 / 17: 16 / 256 candidates).  Kernels of their own so that no side carries another's registers (the hashChain
// kernels spill already).
template <int NCAND>
LZ_DEV void lz_parse_hashchain(const u8* src, u32 nBlock, u32 S, u32 E, const LzHc& hc, LzStreams& st)
{
    constexpr bool NOCHAIN = NCAND == 1;
    const u32 lane = lz_lane();
    int anchor = (int)S, ip = (int)S + 1;                        // uniform; :201
    const int mflimit = (int)E - (int)LZ_MFLIMIT, matchlimit = (int)E - (int)LZ_LASTLITERALS;
    int ml = 0, ml2 = 0, ml3 = 0, ml0 = 0, start0 = 0;
    u32 ref = 0, ref2 = 0, ref3 = 0, ref0 = 0, start2 = 0, start3 = 0, dummy = 0;
    u32 bmBase = 0xFFFF0000u;                                    // first word of the hit bits held in bm (none yet)
    u64 bm = 0;
    // The first-search results of 128 positions (best[bwBase + lane], best[bwBase + 64 + lane]) ride in two registers: one coalesced
    // trip per 128 positions instead of a dependent one per sequence.
    u32 bwBase = 0xFFFF0000u, bwA = 0, bwB = 0;
    // noChain levels: the chain words of 128 positions ride in two registers the same way (lz_nc_search)
    u32 cwBase = 0xFFFF0000u, cwA = 0, cwB = 0;
    auto ncWord = [&](u32 X) -> u32 {
        const u32 o = X - cwBase;
        return o < 64u ? lz_readlane(cwA, o) : o < 128u ? lz_readlane(cwB, o - 64u) : lz_uniform(hc.chain2[X]);
    };
    for (;;) {
        // ---------------- :204-206: first position with any match: a scan of the hit bits ----------------
        for (;;) {
            if (ip >= mflimit) goto tail;
            const u32 wi = (u32)ip >> 6;
            if (wi - bmBase >= 64u) { bmBase = wi; bm = hc.hits[bmBase + lane]; }       // 4096 positions of bits, one word per lane
            const u32 li = wi - bmBase;
            const u64 w = (((u64)lz_readlane((u32)(bm >> 32), li) << 32) | lz_readlane((u32)bm, li)) >> ((u32)ip & 63u);
            if (w) { ip += (int)lz_ctz64(w); if (ip >= mflimit) goto tail; break; }
            ip = (int)((wi + 1u) << 6);
        }
        LZ_PROF(st, 0);
        {
            u32 bw = LZ_HC_BEST_OPEN;                            // the first search, decided ahead of the parse
            if (!NOCHAIN && LZ_HC_PREPASS && hc.pre) {
                if ((u32)ip - bwBase >= 128u) {
                    bwBase = (u32)ip & ~63u;
                    const u32 last = nBlock - 1u, ia = bwBase + lane, ib = bwBase + 64u + lane;
                    bwA = hc.best[ia < last ? ia : last]; bwB = hc.best[ib < last ? ib : last];
                }
                const u32 o = (u32)ip - bwBase;
                bw = o < 64u ? lz_readlane(bwA, o) : lz_readlane(bwB, o - 64u);
            }
            if (bw != LZ_HC_BEST_OPEN) { LZ_STAT(19); ml = (int)(bw & 0xFFFFu); ref = (u32)ip - (bw >> 16); }
            else if constexpr (NCAND != 0) {
                LZ_STAT(20);
                if ((u32)ip - cwBase >= 128u) {
                    cwBase = (u32)ip & ~63u;
                    const u32 last = nBlock - 1u, ia = cwBase + lane, ib = cwBase + 64u + lane;
                    cwA = hc.chain2[ia < last ? ia : last]; cwB = hc.chain2[ib < last ? ib : last];
                }
                if constexpr (NCAND == 1) ml = (int)lz_nc_search(src, ncWord((u32)ip), (u32)ip, (u32)ip, (u32)matchlimit, 0u, false, ref, dummy);
                else                      ml = (int)lz_hcN_search<(NCAND > 1 ? NCAND : 2)>(src, hc, ncWord((u32)ip), (u32)ip, (u32)ip, (u32)matchlimit, 0u, false, ref, dummy);
            }
            else { LZ_STAT(20); ml = (int)lz_hc_search(src, nBlock, hc, (u32)ip, (u32)ip, (u32)matchlimit, 0u, false, ref, dummy, st); }
        }
        LZ_PROF(st, 1);
        if (ml < 16) LZ_STAT(25); else if (ml < 24) LZ_STAT(26); else if (ml < 32) LZ_STAT(27); else LZ_STAT(28);
        start0 = ip; ref0 = ref; ml0 = ml;                                                        // :209
    search2:
        if (ip + ml < mflimit) LZ_STAT(21);
        if (ip + ml < mflimit && (!LZ_HC_SKIP_NOHIT || lz_hc_hit_at(hc, bm, bmBase, (u32)(ip + ml - 2))))   // :212-214
            { LZ_STAT(22); ml2 = NCAND == 1 ? (int)lz_nc_search(src, ncWord((u32)(ip + ml - 2)), (u32)(ip + ml - 2), (u32)(ip + 1), (u32)matchlimit, (u32)ml, true, ref2, start2)
                                 : NCAND > 1 ? (int)lz_hcN_search<(NCAND > 1 ? NCAND : 2)>(src, hc, ncWord((u32)(ip + ml - 2)), (u32)(ip + ml - 2), (u32)(ip + 1), (u32)matchlimit, (u32)ml, true, ref2, start2)
                                            : (int)lz_hc_search(src, nBlock, hc, (u32)(ip + ml - 2), (u32)(ip + 1), (u32)matchlimit, (u32)ml, true, ref2, start2, st); }
        else ml2 = ml;
        LZ_PROF(st, 2);
        if (ml2 == ml) {                                                                          // :216-219
            lz_seq_push(st, (u32)(ip - anchor), (u32)ml, (u32)ip - ref); ip += ml; anchor = ip;
            continue;
        }
        if (start0 < ip) {                                                                        // :221-227
            if ((int)start2 < ip + ml0) { ip = start0; ref = ref0; ml = ml0; }
        }
        if ((int)start2 - ip < 3) {                                                               // :230-235
            ml = ml2; ip = (int)start2; ref = ref2;
            goto search2;
        }
    search3:
        if ((int)start2 - ip < LZ_HC_OPTIMAL_ML) {                                                // :243-260
            int new_ml = ml;
            if (new_ml > LZ_HC_OPTIMAL_ML) new_ml = LZ_HC_OPTIMAL_ML;
            if (ip + new_ml > (int)start2 + ml2 - 4) {
                new_ml = (int)start2 - ip + ml2 - 4;
                if (new_ml < 4 && !NOCHAIN) {                                                  // hashchain.h:257-260; nochain.h:210 has no such exit
                    lz_seq_push(st, (u32)(ip - anchor), (u32)ml, (u32)ip - ref); ip += ml; anchor = ip;
                    continue;
                }
            }
            const int correction = new_ml - ((int)start2 - ip);
            if (correction > 0) { start2 += (u32)correction; ref2 += (u32)correction; ml2 -= correction; }
        }
        if ((int)start2 + ml2 < mflimit) LZ_STAT(23);
        if ((int)start2 + ml2 < mflimit && (!LZ_HC_SKIP_NOHIT || lz_hc_hit_at(hc, bm, bmBase, start2 + (u32)ml2 - 3u)))   // :263-265
            { LZ_STAT(24); ml3 = NCAND == 1 ? (int)lz_nc_search(src, ncWord(start2 + (u32)ml2 - 3u), start2 + (u32)ml2 - 3u, start2, (u32)matchlimit, (u32)ml2, true, ref3, start3)
                                 : NCAND > 1 ? (int)lz_hcN_search<(NCAND > 1 ? NCAND : 2)>(src, hc, ncWord(start2 + (u32)ml2 - 3u), start2 + (u32)ml2 - 3u, start2, (u32)matchlimit, (u32)ml2, true, ref3, start3)
                                            : (int)lz_hc_search(src, nBlock, hc, start2 + (u32)ml2 - 3u, start2, (u32)matchlimit, (u32)ml2, true, ref3, start3, st); }
        else ml3 = ml2;
        LZ_PROF(st, 3);
        if (ml3 == ml2) {                                                                         // :267-275
            if ((int)start2 < ip + ml) ml = (int)start2 - ip;
            lz_seq_push(st, (u32)(ip - anchor), (u32)ml, (u32)ip - ref); ip += ml; anchor = ip;
            ip = (int)start2;
            lz_seq_push(st, (u32)(ip - anchor), (u32)ml2, (u32)ip - ref2); ip += ml2; anchor = ip;
            continue;
        }
        if ((int)start3 < ip + ml + 3) {                                                          // :277-305
            if ((int)start3 >= ip + ml) {
                if ((int)start2 < ip + ml) {
                    const int correction = ip + ml - (int)start2;
                    start2 += (u32)correction; ref2 += (u32)correction; ml2 -= correction;
                    if (ml2 < 4) { start2 = start3; ref2 = ref3; ml2 = ml3; }
                }
                lz_seq_push(st, (u32)(ip - anchor), (u32)ml, (u32)ip - ref); ip += ml; anchor = ip;
                ip = (int)start3; ref = ref3; ml = ml3;
                start0 = (int)start2; ref0 = ref2; ml0 = ml2;
                goto search2;
            }
            start2 = start3; ref2 = ref3; ml2 = ml3;
            goto search3;
        }
        if ((int)start2 < ip + ml) {                                                              // :311-338
            if ((int)start2 - ip < 15) {
                if (ml > LZ_HC_OPTIMAL_ML) ml = LZ_HC_OPTIMAL_ML;
                if (ip + ml > (int)start2 + ml2 - 4) {
                    ml = (int)start2 - ip + ml2 - 4;
                    if (ml < 4) {
                        lz_seq_push(st, (u32)(ip - anchor), (u32)ml, (u32)ip - ref); ip += ml; anchor = ip;
                        ip = (int)start3; ref = ref3; ml = ml3;
                        start0 = (int)start2; ref0 = ref2; ml0 = ml2;
                        goto search2;
                    }
                }
                const int correction = ml - ((int)start2 - ip);
                if (correction > 0) { start2 += (u32)correction; ref2 += (u32)correction; ml2 -= correction; }
            } else {
                ml = (int)start2 - ip;
            }
        }
        lz_seq_push(st, (u32)(ip - anchor), (u32)ml, (u32)ip - ref); ip += ml; anchor = ip;       // :339
        ip = (int)start2; ref = ref2; ml = ml2;
        start2 = start3; ref2 = ref3; ml2 = ml3;
        goto search3;
    }
tail:
    if (st.nseq & (LZ_SEQ_RING - 1u)) lz_seq_flush(st);
    st.lastLits = E - (u32)anchor; st.nlit += E - (u32)anchor;   // lizard_compress_lz4.h:74-86
}
