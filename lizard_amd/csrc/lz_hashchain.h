// lz_hashchain.h — hashChain parser (levels 13-17 / 34-38, fastLZ4 codewords) on one wavefront.
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
//     exactly what Insert recorded for X itself.  So the chain is built AHEAD of the parse, 64 positions
//     per step (phase A, lz_hc_build), into prev[p] = distance to the previous head (0 = none inside the
//     64 KiB window), one u16 per block position; the searches of phase B never touch the head table.
//     (Positions the reference never inserts — the tail after the last search — are invisible: a search at
//     X only ever follows links that start below X.  The reference never searches below nextToUpdate —
//     every search position exceeds the previous one, see the walk through :207-339 in DESIGN.md — and
//     tests/test_oracle pins that on the corpus.)
//   * chainTable is a 2^16 ring in the reference; entries are only read inside the 65535 window where
//     the ring has no aliasing, so a full per-position array is observably the same.  A link the reference
//     clamps to 65535 because it is longer (":30") ends its walk one step later (":92/:172" leave the
//     window); here it is stored as 0 = end of chain.  A true distance of exactly 65535 stays usable.
//   * Head table: 2^18 u32 slots per wave in global memory, never cleared between blocks: a slot holds
//     (epoch << 22 | position) and counts as empty when the epoch is not the current block's (10-bit epoch
//     per wave slot, table re-zeroed every 1023 blocks; blocks up to 4 MiB).
//   * Phase B: the outer "ip++ until a position has a match" loop (:204-206) runs 64 consecutive positions
//     per round, each lane walking its own chain until the first candidate that passes the reference's
//     tests (any such candidate makes ml >= 4 > 0); the first such lane is the position the reference
//     stops at.  From there on the LZ4HC-style arbitration is a serial chain per sequence and runs as
//     wave-uniform code; each search collects up to 64 chain candidates (one per lane), filters them in
//     parallel on the 4-byte test and measures the survivors with the wave-wide compare helpers.  The
//     reference's "first strictly longer match wins" over the chain order is "maximum length, earliest
//     on ties"; its one-byte pre-checks (:73, :146) are necessary conditions of "strictly longer" and
//     therefore unobservable.
//
// Included from lz_block.h after the shared helpers.
#pragma once

#define LZ_HC_HASHLOG   18
#define LZ_HC_TAGLOG    11                      // round tag array: 2 KiB of LDS (aliases the Huffman workspace)
#define LZ_HC_NONE      0x80000000u             // "no head": p - NONE is >= 8 and > 65535 for every block position
#define LZ_HC_EPOCHS    1024u
#define LZ_HC_OPTIMAL_ML 18                     // (ML_MASK_LZ4-1)+MINMATCH, hashchain.h:3
// Per-wave slot in global memory: head table, 64 bytes of persistent metadata (word 0 = epoch of the last
// block), then prev[] with one u16 per position of the largest block.  The host zeroes a slot once.
#define LZ_HC_HEAD_BYTES (4u << LZ_HC_HASHLOG)
#define LZ_HC_SLOT_BYTES(maxBlock) ((size_t)LZ_HC_HEAD_BYTES + 64u + 2u * (size_t)(maxBlock) + 64u)

struct LzHc {
    u32* head;          // global: 2^18 epoch-tagged slots
    u16* prev;          // global: per block position, distance to the previous head of its bucket (0 = none)
    u8*  tag;           // LDS: 2^LZ_HC_TAGLOG bytes
    u32  epoch;         // uniform, 1..1023
    u32  searchNum;     // uniform
};

template <int SEARCHLEN>
LZ_DEV u32 lz_hc_hash(u64 bytes)
{
    if constexpr (SEARCHLEN == 4) return ((u32)bytes * 2654435761u) >> (32 - LZ_HC_HASHLOG);     // lizard_compress.c:87-88
    else return lz_hash5<LZ_HC_HASHLOG>(bytes);                                                   // :90-91
}

// Binds a wave's slot and opens a new epoch (all lanes call).
LZ_DEV void lz_hc_begin(LzHc& hc, void* slotMem, u8* tag, u32 searchNum)
{
    hc.head = (u32*)slotMem;
    u32* meta = hc.head + (1u << LZ_HC_HASHLOG);
    hc.prev = (u16*)(meta + 16);
    hc.tag = tag;
    hc.searchNum = searchNum;
    u32 epoch = lz_uniform(meta[0]) + 1u;
    if (epoch >= LZ_HC_EPOCHS) {
        uint4 z; z.x = z.y = z.z = z.w = 0u;
        for (u32 i = lz_lane() * 4u; i < (1u << LZ_HC_HASHLOG); i += 256u) *(uint4*)(hc.head + i) = z;
        epoch = 1u;
    }
    lz_wave_sync();
    if (lz_lane() == 0) meta[0] = epoch;
    lz_converge();
    hc.epoch = epoch;
}

// Phase A: Lizard_Insert (hashchain.h:13-43) for every position of the block that has 8 readable bytes.
// One step = 64 consecutive positions.  Lanes of a step that share a bucket see each other's conditional
// head updates in position order: such groups (found through the tag array) are replayed with scalar
// code; the last lane of every group stores the bucket's final head.
template <int SEARCHLEN>
LZ_DEV void lz_hc_build(const u8* src, u32 n, const LzHc& hc)
{
    const u32 lane = lz_lane();
    const u64 laneBit = 1ull << lane;
    const u64 lanesAbove = ~(laneBit | (laneBit - 1ull));
    const u32 tagMask = (1u << LZ_HC_TAGLOG) - 1u;
    const u32 nIns = n >= 8u ? n - 7u : 0u;
    for (u32 base = 0; base < nIns; base += 64u) {
        const u32 p = base + lane;
        const bool valid = p < nIns;
        u32 h = 0, hp = LZ_HC_NONE;
        if (valid) {
            h = lz_hc_hash<SEARCHLEN>(lz_ld64(src + p));
            const u32 e = hc.head[h];
            if ((e >> 22) == hc.epoch) hp = e & 0x3FFFFFu;
            hc.tag[h & tagMask] = (u8)lane;
        }
        lz_lds_sync();
        const bool lost = valid && hc.tag[h & tagMask] != (u8)lane;
        u64 pend = lz_ballot(lost);
        u64 grp = laneBit;
        u32 seen = hp;                                           // head as my own insertion sees it
        u32 after = (p - hp >= LZ_MIN_OFFSET) ? p : hp;          // :38 when alone in the bucket
        while (pend) {
            const u32 f = lz_ctz64(pend);
            const u32 hv = lz_readlane(h, f);
            const bool mine = valid && h == hv;
            const u64 g = lz_ballot(mine);
            u32 t = lz_readlane(hp, f);                          // bucket head before this step (uniform)
            for (u64 m = g; m; m &= m - 1ull) {
                const u32 k = lz_ctz64(m), pk = base + k;
                if (lane == k) seen = t;
                t = (pk - t >= LZ_MIN_OFFSET) ? pk : t;
                if (lane == k) after = t;
            }
            if (mine) grp = g;
            pend &= ~g;
        }
        lz_lds_sync();                                           // tag reads done before the next step's writes
        if (valid) {
            const u32 d = p - seen;
            hc.prev[p] = (u16)(d <= LZ_MAX_DIST_LZ4 ? d : 0u);   // :27-31
            if ((grp & lanesAbove) == 0) hc.head[h] = (hc.epoch << 22) | after;
        }
        lz_wave_sync();                                          // head stores before the next step's loads
    }
}

// One search of the chain that starts at X (uniform).  wider == false: Lizard_InsertAndFindBestMatch
// (longest = 0 on entry, iLow unused); wider == true: Lizard_InsertAndGetWiderMatch with backward
// extension down to iLow.  Returns the new longest; ref/start change only when it grew.
//
// Up to 64 chain candidates are measured at once, one per lane: 4-byte test, then 24 more bytes forward
// and 8 bytes backward with lane-local loads.  Lanes whose comparison is still open after that (long
// matches) are finished one at a time with the wave-wide helpers, skipping those that cannot reach the
// best length any more.  "First strictly longer candidate in chain order" == maximum length, lowest lane.
LZ_DEV u32 lz_hc_search(const u8* src, const LzHc& hc, u32 X, u32 iLow, u32 iHigh, u32 longest, bool wider, u32& ref, u32& start, LzStreams& st)
{
    const u32 lane = lz_lane();
    const u32 first4 = lz_ld32(src + X);
    const u32 maxFwd = iHigh - X;                                // X + 4 < iHigh for every caller
    u32 m = X, left = hc.searchNum;                              // uniform
    bool more = true;
    while (more && left) {
        const u32 batch = left < 64u ? left : 64u;
        u32 cand = 0, cnt = 0;
        for (u32 j = 0; j < batch; j++) {                        // the walk itself is serial: one link per step
            const u32 d = lz_uniform((u32)hc.prev[m]);
            if (!d) { more = false; break; }
            m -= d;
            if (X - m > LZ_MAX_DIST_LZ4) { more = false; break; }
            if (lane == j) cand = m;
            cnt++;
        }
        left -= cnt;
        LZ_PROF(st, 8);                                          // (instrumented build) chain walk
        const bool ok = lane < cnt && X - cand >= LZ_MIN_OFFSET && lz_ld32(src + cand) == first4;   // :73 / :146
        u32 mlt = 0, bk = 0;
        bool open = false;                                       // my comparison needs the wave-wide helpers
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

// Sub-block [S,E) of the block at src (hashchain.h:188-369).  The chain must have been built for the block.
LZ_DEV void lz_parse_hashchain(const u8* src, u32 S, u32 E, const LzHc& hc, LzStreams& st)
{
    const u32 lane = lz_lane();
    int anchor = (int)S, ip = (int)S + 1;                        // uniform; :201
    const int mflimit = (int)E - (int)LZ_MFLIMIT, matchlimit = (int)E - (int)LZ_LASTLITERALS;
    int ml = 0, ml2 = 0, ml3 = 0, ml0 = 0, start0 = 0;
    u32 ref = 0, ref2 = 0, ref3 = 0, ref0 = 0, start2 = 0, start3 = 0, dummy = 0;
    for (;;) {
        // ---------------- :204-206: first position with any match, 64 positions per round ----------------
        for (;;) {
            if (ip >= mflimit) goto tail;
            const u32 p = (u32)ip + lane;
            const bool valid = (int)p < mflimit;
            // Wave-uniform loop over chain steps.  A step costs one round trip: the candidate's 4 bytes and
            // its own link are fetched together.  Lanes above the lowest lane that already has a match stop
            // walking — the reference never reaches their positions.
            bool hit = false, walking = valid;
            const u32 first4 = lz_ld32(src + (valid ? p : S));
            u32 m = valid ? p : S;
            u32 d = hc.prev[m];
            for (u32 a = 0; a < hc.searchNum; a++) {
                walking = walking && d != 0u && p - (m - d) <= LZ_MAX_DIST_LZ4;
                if (!lz_ballot(walking)) break;
                m = walking ? m - d : m;
                const u32 c4 = lz_ld32(src + m);
                d = hc.prev[m];
                if (walking && p - m >= LZ_MIN_OFFSET && c4 == first4) { hit = true; walking = false; }
                const u64 hm = lz_ballot(hit);
                if (hm) walking = walking && lane < lz_ctz64(hm);
            }
            const u64 okMask = lz_ballot(hit);
            if (okMask) { ip += (int)lz_ctz64(okMask); break; }
            ip += (int)lz_popc64(lz_ballot(valid));
        }
        LZ_PROF(st, 0);
        ml = (int)lz_hc_search(src, hc, (u32)ip, (u32)ip, (u32)matchlimit, 0u, false, ref, dummy, st);
        LZ_PROF(st, 1);
        start0 = ip; ref0 = ref; ml0 = ml;                                                        // :209
    search2:
        if (ip + ml < mflimit)                                                                    // :212-214
            ml2 = (int)lz_hc_search(src, hc, (u32)(ip + ml - 2), (u32)(ip + 1), (u32)matchlimit, (u32)ml, true, ref2, start2, st);
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
                if (new_ml < 4) {
                    lz_seq_push(st, (u32)(ip - anchor), (u32)ml, (u32)ip - ref); ip += ml; anchor = ip;
                    continue;
                }
            }
            const int correction = new_ml - ((int)start2 - ip);
            if (correction > 0) { start2 += (u32)correction; ref2 += (u32)correction; ml2 -= correction; }
        }
        if ((int)start2 + ml2 < mflimit)                                                          // :263-265
            ml3 = (int)lz_hc_search(src, hc, start2 + (u32)ml2 - 3u, start2, (u32)matchlimit, (u32)ml2, true, ref3, start3, st);
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
