// variants/lz_fast128.h — rounds of 128 slots for the fast parser (levels 10 / 30, LDS tables): built, bit-exact, measured and NOT
// kept in round 4 (14.5 % slower).  Compiled only into -DLZ_FAST_128=1 tuning builds (lz_block.h includes this
// file under that switch); not part of the shipped kernels and not hashed into bench.py's kernel_source_sha16.
#pragma once
// ---- rounds of 128 slots (levels 10 / 30, LDS tables; round 4) ----
// A third of the 64-slot rounds end without a winner and the run goes on with the next 64 slots, whose positions were known all
// along.  Here every lane takes TWO slots of the run — slot l (half A) and slot 64 + l (half B): both are hashed together, both
// entries are exchanged into the table in ONE LDS trip (LzTab::xchg2: all of A, then all of B — exactly the order of the
// reference's walk), both candidates are tested and their bytes requested in ONE batch.  The first accepting slot in A wins as
// before (several sequences per round included); if A has none, the first accepting slot in B wins and the whole of A has
// happened.  What never happened is taken back as in the 64-slot form: B's lanes restore first (everything they received except
// entries of B's own lanes — told by their age, positions ascend), then A's undone lanes by the existing rule, so that a slot
// which an undone A lane had written ends with that lane's restore.  A run that needs more than 64 slots costs one round
// instead of two.
// MEASURED AND NOT KEPT (profiles/r04k_rounds_of_128_slots_rejected.txt): bit-exact (emulator corpus + soaks, GPU: every block of
// the bench batches), but level 10 169.8 instead of 198.6 GB/s, level 30 132.2 instead of 151.1, and a wave alone on its CU
// (256 x 4 MiB) 25.0 instead of 27.3: 22 % fewer rounds do not pay for hashing, exchanging, testing and settling a second slot per
// lane in EVERY round — the parse is bound by the instructions of a round, not by its LDS and memory trips.  Variant builds only
// (make variant NAME=r128 DEFS=-DLZ_FAST_128=1).
#ifndef LZ_FAST_128
#define LZ_FAST_128 0
#endif
#if LZ_FAST_128
struct LzHalf {                                                      // one slot of a lane: what the accept test and the winner need
    u32 p, h, mine, e, age, ep, fwd, cbk;
    bool valid, cand, ok, have24;
};
// candidate measurement from the fetched bytes (fast.h:97,100,102), as in lz_parse_fast
LZ_DEV void lz_half_measure(LzHalf& x, u64 bytes, u64 cA, u64 cB, u64 pB, u64 cC, u64 pC, u64 cZ, u64 pZ, bool haveBack, u32 matchlimit)
{
    x.ok = x.cand && (u32)cA == (u32)bytes;
    const u64 a = bytes ^ cA, y = pB ^ cB, y2 = pC ^ cC, z = pZ ^ cZ;
    const u32 seen = x.have24 ? 24u : 16u;
    const u32 common = a ? lz_ctz64(a) >> 3 : y ? 8u + (lz_ctz64(y) >> 3) : (x.have24 && y2) ? 16u + (lz_ctz64(y2) >> 3) : seen;
    const u32 room = matchlimit - x.p;
    x.fwd = 0xFFFFu;
    if (common < seen || room <= seen) x.fwd = common < room ? common : room;
    x.cbk = !haveBack ? 0u : z ? lz_clz64(z) >> 3 : 8u;
}
template <int HASHLOG>
LZ_DEV void lz_parse_fast128(const u8* src, u32 S, u32 E, const LzTab& table, LzStreams& st)
{
    constexpr bool kChain = LZ_FAST_CHAIN;
    constexpr u32 kTrash = 1u << HASHLOG;
    const u32 lane = lz_lane();
    const u64 laneBit = 1ull << lane;
    u32 anchor = S;                                                  // uniform
    if (E - S < LZ_MFLIMIT + 1u) { st.lastLits = E - S; st.nlit += E - S; return; }     // fast.h:63
    const u32 mflimit = E - LZ_MFLIMIT, matchlimit = E - LZ_LASTLITERALS;
    const u32 lowPos = S > LZ_MAX_DIST_LZ4 ? S - LZ_MAX_DIST_LZ4 : 0u;                  // fast.h:57-58

    if (S >= st.sweepAt) { lz_tab_sweep<HASHLOG>(table, S, false); st.sweepAt = S + LzTab::kSweepEvery; table.sync(); }
    if (lane == 0) { const u64 b0 = lz_ld64(src + S); table.set(lz_hash5<HASHLOG>(b0), table.entry(S, (u32)b0)); }   // fast.h:66
    table.sync();

    u32 ip = S + 1u, special = 0;                                    // uniform
    // my two slots of the coming round, prepared one round ahead (or right after a match)
    u32 pNa, pNb; bool vNa, vNb, poNa, poNb; u64 bNa, bNb;
    lz_slot_pos(ip, 0u, lane, mflimit, pNa, vNa, poNa);
    lz_slot_pos(ip, 0u, 64u + lane, mflimit, pNb, vNb, poNb);
    bNa = lz_ld64(src + (vNa ? pNa : S)); bNb = lz_ld64(src + (vNb ? pNb : S));
    for (;;) {
        u32 v0 = 0;                                                  // uniform: slots consumed by earlier rounds of this run
        u32 P = 0, M = 0, ml = 0, back = 0;                          // uniform: winner position, candidate, lengths
        for (;;) {
            LZ_PROF(st, 3);
            LzHalf A, B;
            A.p = pNa; A.valid = vNa; B.p = pNb; B.valid = vNb;
            const bool putOnly = poNa;                               // slot 0 of a run behind a match (half A only)
            const u64 bytesA = bNa, bytesB = bNb;
            u32 aheadA, aheadB;
            lz_slot_pos(ip, special, v0 + 128u + lane, mflimit, pNa, vNa, poNa);
            lz_slot_pos(ip, special, v0 + 192u + lane, mflimit, pNb, vNb, poNb);
            aheadA = vNa ? pNa : S; aheadB = vNb ? pNb : S;
            {   // keep every live slot younger than 2^17 positions (see LzTab): before anything of this round is in the table
                const u32 p0 = lz_readlane(A.p, 0);
                if (p0 >= st.sweepAt) { lz_tab_sweep<HASHLOG>(table, p0, false); st.sweepAt = p0 + LzTab::kSweepEvery; table.sync(); }
            }
            A.h = lz_hash5<HASHLOG>(bytesA); A.mine = table.entry(A.p, (u32)bytesA);
            B.h = lz_hash5<HASHLOG>(bytesB); B.mine = table.entry(B.p, (u32)bytesB);
            // fast.h:86-88 for 128 visits in one LDS trip, in the reference's order
            table.xchg2(A.valid ? A.h : kTrash, A.mine, B.valid ? B.h : kTrash, B.mine, A.e, B.e);
            // accept test, fast.h:90-97 (check bits first)
            A.age = table.age(A.p, A.e); A.ep = A.p - A.age;
            B.age = table.age(B.p, B.e); B.ep = B.p - B.age;
            A.cand = A.valid && !putOnly && table.sameCheck(A.e, A.mine) && A.age >= LZ_MIN_OFFSET && A.age <= LZ_MAX_DIST_LZ4 && A.age <= A.p - lowPos;
            B.cand = B.valid && table.sameCheck(B.e, B.mine) && B.age >= LZ_MIN_OFFSET && B.age <= LZ_MAX_DIST_LZ4 && B.age <= B.p - lowPos;
            A.have24 = A.p + 24u <= E; B.have24 = B.p + 24u <= E;
            u64 acA = LZ_ANY64, acB = LZ_ANY64, apB = LZ_ANY64, acC = LZ_ANY64, apC = LZ_ANY64, acZ = LZ_ANY64, apZ = LZ_ANY64;
            u64 bcA = LZ_ANY64, bcB = LZ_ANY64, bpB = LZ_ANY64, bcC = LZ_ANY64, bpC = LZ_ANY64, bcZ = LZ_ANY64, bpZ = LZ_ANY64;
            const bool backA = A.cand && A.ep >= 8u, backB = B.cand && B.ep >= 8u;
            if (A.cand) {
                const u32 zb = backA ? 8u : 0u, fc = A.have24 ? 16u : 0u;
                acA = lz_ld64(src + A.ep); acB = lz_ld64(src + A.ep + 8u); apB = lz_ld64(src + A.p + 8u);
                acC = lz_ld64(src + (A.ep + fc)); apC = lz_ld64(src + (A.p + fc));
                acZ = lz_ld64(src + (A.ep - zb)); apZ = lz_ld64(src + (A.p - zb));
            }
            if (B.cand) {
                const u32 zb = backB ? 8u : 0u, fc = B.have24 ? 16u : 0u;
                bcA = lz_ld64(src + B.ep); bcB = lz_ld64(src + B.ep + 8u); bpB = lz_ld64(src + B.p + 8u);
                bcC = lz_ld64(src + (B.ep + fc)); bpC = lz_ld64(src + (B.p + fc));
                bcZ = lz_ld64(src + (B.ep - zb)); bpZ = lz_ld64(src + (B.p - zb));
            }
            bNa = lz_ld64(src + aheadA); bNb = lz_ld64(src + aheadB);   // the next round's bytes (consumed only if no slot accepts)
            LZ_PROF(st, 0);
            lz_half_measure(A, bytesA, acA, acB, apB, acC, apC, acZ, apZ, backA, matchlimit);
            lz_half_measure(B, bytesB, bcA, bcB, bpB, bcC, bpC, bcZ, bpZ, backB, matchlimit);
            lz_pin(A.fwd); lz_pin(A.cbk); lz_pin(B.fwd); lz_pin(B.cbk);
            const u64 okA = lz_ballot(A.ok), okB = lz_ballot(B.ok);      // uniform
            const u64 validA = lz_ballot(A.valid), validB = lz_ballot(B.valid);
            u32 w = 0;
            u64 commitA = validA, commitB = validB, deadMask = 0;
            if (okA) {
                commitB = 0;                                             // half B never happened
                w = lz_ctz64(okA); commitA = validA & (~0ull >> (63u - w));
                if constexpr (kChain) {
                    while (v0 == 0) {                                    // first round of a run: consecutive positions behind the winner (see lz_parse_fast)
                        const u32 Pw = lz_readlane(A.p, w), fw = lz_readlane(A.fwd, w);
                        if (fw == 0xFFFFu) break;
                        const u32 Mw = lz_readlane(A.ep, w);
                        const u32 bk = lz_back_from(lz_readlane(A.cbk, w), Pw, Mw, anchor);
                        if (bk == 0xFFFFu) break;
                        const u32 ipn = Pw + fw, l1 = w + fw;
                        if (ipn > mflimit || l1 > 63u) break;
                        const u64 from1 = ~0ull << l1;
                        const u64 ok2 = okA & from1;
                        if (!ok2) break;
                        const u32 w2 = lz_ctz64(ok2);
                        const u64 put2 = 1ull << (l1 - 2u);
                        const u64 dead2 = deadMask | ((from1 ^ (~0ull << (w + 1u))) & ~put2);
                        const u64 readers = put2 | (from1 & (~0ull >> (63u - w2)));
                        const bool stale = A.age <= lane && ((dead2 >> (lane - A.age)) & 1ull);
                        if (lz_ballot(stale) & readers) break;
                        lz_seq_push(st, Pw - bk - anchor, fw + bk, Pw - Mw);
                        anchor = ipn;
                        deadMask = dead2; commitA |= readers; w = w2;
                    }
                }
            } else if (okB) {
                w = lz_ctz64(okB); commitB = validB & (~0ull >> (63u - w));
            }
            // settle: first half B's undone lanes, then half A's (a slot an undone A lane had written ends with that lane's restore)
            if (okA | okB) {
                bool restoreB;
                const u32 pFirstB = lz_readlane(B.p, 0);
                if (okA) restoreB = B.valid && !(B.age <= B.p - pFirstB);                       // everything but entries of B's own lanes
                else { const u32 pw = lz_readlane(B.p, w); restoreB = B.valid && !(commitB & laneBit) && !(B.p > pw && B.age < B.p - pw); }
                table.set(restoreB ? B.h : kTrash, B.e);
                if (okA) {
                    const u32 pw = lz_readlane(A.p, w);
                    const bool eUndone = (A.p > pw && A.age < A.p - pw) || (kChain && A.age <= lane && ((deadMask >> (lane - A.age)) & 1ull));
                    const bool restoreA = A.valid && !(commitA & laneBit) && !eUndone;
                    table.set(restoreA ? A.h : kTrash, A.e);
                }
            }
            table.sync();
            LZ_PROF(st, 1);
            if (okA) {
                P = lz_readlane(A.p, w); M = lz_readlane(A.ep, w);
                ml = lz_readlane(A.fwd, w); back = lz_back_from(lz_readlane(A.cbk, w), P, M, anchor);
                break;
            }
            if (okB) {
                P = lz_readlane(B.p, w); M = lz_readlane(B.ep, w);
                ml = lz_readlane(B.fwd, w); back = lz_back_from(lz_readlane(B.cbk, w), P, M, anchor);
                break;
            }
            if (validA != ~0ull || validB != ~0ull) goto tail;           // ran into mflimit without a match
            v0 += 128u;
        }
        // ---------------- extend (as lz_parse_fast) ----------------
        if (ml == 0xFFFFu) ml = 4u + lz_count_fwd(src, P + 4u, M + 4u, matchlimit);            // fast.h:100
        if (back == 0xFFFFu) back = lz_count_back(src, P, M, anchor);                           // fast.h:102
        {
            const u32 ipn = P + ml;                                      // sweeps that fall due inside the match (see lz_parse_fast)
            while (ipn >= st.sweepAt) {
                const u32 q = st.sweepAt > P ? st.sweepAt : P + 1u;
                lz_tab_sweep<HASHLOG>(table, q, false); st.sweepAt = q + LzTab::kSweepEvery; table.sync();
            }
        }
        P -= back; M -= back; ml += back;
        ip = P + ml;
        LZ_PROF(st, 2);
        special = 1u;
        lz_slot_pos(ip, 1u, lane, mflimit, pNa, vNa, poNa);
        lz_slot_pos(ip, 1u, 64u + lane, mflimit, pNb, vNb, poNb);
        if (ip > mflimit) { vNa = false; vNb = false; }                  // fast.h:143: there is no next run
        bNa = lz_ld64(src + (vNa ? pNa : S)); bNb = lz_ld64(src + (vNb ? pNb : S));
        lz_seq_push(st, P - anchor, ml, P - M);                          // fast.h:138
        anchor = ip;
        if (ip > mflimit) goto tail;                                     // fast.h:143
    }
tail:
    if (st.nseq & (LZ_SEQ_RING - 1u)) lz_seq_flush(st);
    st.lastLits = E - anchor; st.nlit += E - anchor;                     // fast.h:187-190
    LZ_PROF(st, 3);
}
#endif   // LZ_FAST_128
