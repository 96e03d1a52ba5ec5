// lz_fastbig.h — fastBig parser (levels 20 / 40: LIZv1 codewords, hashLog 14, windowLog 22) on one wavefront.
//
// Bit-exact with reference lib/lizard_parser_fastbig.h:36-175 on a zeroed state; parameters lizard_common.h:248, :270
// (windowLog 22, hashLog 14, minMatchLongOff 16).  The reference's fastBig is its fast parser (lizard_parser_fast.h, levels 11 / 31:
// the same visit schedule, the same unconditional get-then-put, the same post-match steps) with three differences:
//   * the window is 4 MiB, so a candidate may lie anywhere in a block of the benchmark sizes (fastbig.h:85), and
//   * a candidate 65 536 or more bytes back only counts when the match — forward count behind the first four bytes PLUS the
//     backward extension — is at least 16 long (:92-98; the post-match probe :143-146 has no backward extension), and
//   * sequences are LIZv1 codewords (:127), always with an explicit offset (a fastBig match never has offset 0).
//
// Wave mapping: the rounds of lz_parse_fast (lz_block.h) — 64 slots of the visit schedule per round, one per lane, the first
// accepting lane is the reference's match, slots behind it never happened — over the u32 slots of lz_pricefast.h's global-memory
// table form (LzTab32G: position mod 2^24 + 8 check bits, a sweep every 2^22 positions, exact for any block size): 64 KiB per wave
// in a global-memory slot, same-slot lanes of a round found through an LDS tag array, nothing stored before the round is settled.
// What is new is the accept test: a lane whose candidate is 65 536 or more back needs its lengths before the ballot.  They come out
// of the bytes every candidate lane fetches anyway (24 forward, 8 backward); a lane whose lengths those bytes do not decide is
// "undecided", and the undecided lanes in front of the first accepting one are measured with the wave-wide helpers, in lane order,
// before the winner is taken (rare: long offsets with borderline lengths).
// Several sequences out of one round, as at levels 10 / 11 (lz_block.h, "More than one sequence per round"): in the first round of a
// run the lanes behind the winner hold the consecutive positions behind it, so when the winner's lengths come out of its fetched
// bytes the reference's next steps — put(ip-2), the probe of ip (fastbig.h:133-160), the visits ip+1, ... of the next run — are
// those lanes, provided none of them took its slot's value from a lane inside the match.  What differs from level 11: behind a
// sequence the anchor has moved, and with it the backward room of every long-offset lane, so the accept test of the lanes from the new
// ip on is made again with the new anchor (the lane of ip itself is a post-match probe: no backward extension); an undecided lane in
// front of the next accepting one ends the chain.
#ifndef LZ_FASTBIG_CHAIN
#define LZ_FASTBIG_CHAIN 1
#endif
// Sequences go through lz_pricefast.h's LIZv1 list (lz_seq_push_liz, lz_seq_sizes_liz, lz_encode_lizv1).
//
// Slot codes in LDS (`codes`, optional: 4 bits per slot, 8 KiB per wave at hashLog 14).  A round reads 64 slots and every slot is a
// 128-byte line of its own: with sixteen waves per CU that is what the level costs (394 GB of fabric traffic per 4 GiB of input,
// 5.2 TB/s, profiles/r06fin2_*).  Most of those reads find an entry that cannot match.  The code of a slot is 0 while nothing was put
// there in this block, else the low four bits of the entry's check bits (0 written as 15); a probe whose own code differs never
// reads the slot — its entry, if any, fails the check-bit test (:95 below) — and a slot with code 0 is never read at all, so the
// table in global memory needs no clearing between blocks.  Exact: the code is written by the lane that writes the slot.
// Included from lz_block.h behind lz_pricefast.h.
#pragma once

LZ_DEV u32 lz_fb_code(u32 chk8) { const u32 c = chk8 & 15u; return c ? c : 15u; }
LZ_DEV u32 lz_fb_code_get(const u32* codes, u32 h) { return (codes[h >> 3] >> ((h & 7u) * 4u)) & 15u; }
LZ_DEV void lz_fb_code_put(u32* codes, u32 h, u32 c) { lz_lds_mskor(&codes[h >> 3], 15u << ((h & 7u) * 4u), c << ((h & 7u) * 4u)); }
template <int HASHLOG> LZ_DEV void lz_fb_codes_fresh(u32* codes) { for (u32 i = lz_lane(); i < (1u << HASHLOG) / 8u; i += 64u) codes[i] = 0u; lz_lds_sync(); }

template <int HASHLOG, int TAGLOG>
LZ_DEV void lz_parse_fastbig(const u8* src, u32 S, u32 E, const LzTab32G& table, u8* tag, u32* codes, LzStreams& st)
{
    typedef LzTab32G TAB;
    const u32 lane = lz_lane();
    const u64 laneBit = 1ull << lane;
    const u64 lanesBelow = laneBit - 1ull;
    const u32 maxDist = (1u << 22) - 1u;                             // windowLog 22
    const u32 tagMask = (1u << TAGLOG) - 1u;
    u32 anchor = S;                                                  // uniform
    if (E - S < LZ_MFLIMIT + 1u) { st.lastLits = E - S; return; }    // fastbig.h:58
    const u32 mflimit = E - LZ_MFLIMIT, matchlimit = E - LZ_LASTLITERALS;
    const u32 lowPos = S > maxDist ? S - maxDist : 0u;               // fastbig.h:53: lowLimit is fixed at sub-block entry

    if (S >= st.sweepAt) { table.sync(); lz_pf_tab_sweep<HASHLOG>(table, S); st.sweepAt = S + LZ_PF_SWEEP_EVERY; }
    table.sync();
    if (lane == 0) {                                                 // fastbig.h:61
        const u64 b0 = lz_ld64(src + S);
        const u32 h0 = lz_hash5<HASHLOG>(b0), c0 = TAB::chkOf((u32)b0);
        table.set(h0, TAB::make(S, c0));
        if (codes) lz_fb_code_put(codes, h0, lz_fb_code(c0));
    }
    lz_converge();
    table.sync();
    lz_lds_sync();

    u32 ip = S + 1u;        // uniform: run start, or (special == 1) the post-match probe position
    u32 special = 0;        // uniform
    u32 pNext; bool validNext, putOnlyNext;                          // my slot of the coming round, prepared one round ahead (lz_parse_fast)
    u64 nextBytes;
    lz_slot_pos(ip, 0u, lane, mflimit, pNext, validNext, putOnlyNext);
    nextBytes = lz_ld64(src + (validNext ? pNext : S));
    for (;;) {
        u32 v0 = 0;         // uniform: slots consumed by earlier rounds of this run
        u32 P = 0, M = 0, ml = 0, back = 0;   // uniform: winner position, candidate, forward length (from P), backward extension
        for (;;) {
            const u32 p = pNext; const bool valid = validNext, putOnly = putOnlyNext;
            const u64 bytes = nextBytes;
            u32 pAhead;
            lz_slot_pos(ip, special, v0 + 64u + lane, mflimit, pAhead, validNext, putOnlyNext);
            pNext = pAhead;
            if (!validNext) pAhead = S;                              // any readable address
            {
                const u32 p0 = lz_readlane(p, 0);
                if (p0 >= st.sweepAt) { table.sync(); lz_pf_tab_sweep<HASHLOG>(table, p0); st.sweepAt = p0 + LZ_PF_SWEEP_EVERY; table.sync(); }
            }
            const u32 first4 = (u32)bytes;
            const u32 h = lz_hash5<HASHLOG>(bytes);
            const u32 myChk = TAB::chkOf(first4);
            const u32 mine = TAB::make(p, myChk);
            u32 e;                                                   // fastbig.h:81: the slot before this round
            if (codes) {                                             // (uniform) only the lanes whose slot's code is theirs read the table
                e = TAB::dead(p);
                if (valid && !putOnly && lz_fb_code_get(codes, h) == lz_fb_code(myChk)) e = table.get(h, p);
                lz_converge();
            } else e = table.get(valid ? h : 0u, p);
            u64 grp = laneBit;                                       // lanes of this round on my table slot
            u32 jPrev = 64u;                                         // the lane my `e` came from (64 = the table)
            {
                const u32 ti = h & tagMask;
                if (valid) tag[ti] = (u8)lane;
                lz_lds_sync();
                const bool lost = valid && tag[ti] != (u8)lane;
                lz_lds_sync();                                       // reads done before the next round's writes
                u64 pend = lz_ballot(lost);
                if (pend) {
                    while (pend) {
                        const u32 f = lz_ctz64(pend);
                        const u32 hv = lz_readlane(h, f);
                        const bool same = valid && h == hv;
                        const u64 g = lz_ballot(same);
                        if (same) grp = g;
                        pend &= ~g;
                    }
                    const u64 prev = grp & lanesBelow;
                    const u32 j = prev ? 63u - lz_clz64(prev) : lane;
                    const u32 ej = lz_shfl(mine, j);                 // the put of the closest lower lane on my slot (:83, unconditional)
                    if (prev) { e = ej; jPrev = j; }
                }
            }
            // candidate test, fastbig.h:85, :90 (check bits first: they decide whether any bytes are fetched)
            const u32 age = TAB::age(p, TAB::pos(e));
            const u32 ep = p - age;
            const bool cand = valid && !putOnly && TAB::chk(e) == myChk && age >= LZ_MIN_OFFSET && age <= maxDist && age <= p - lowPos;
            u64 cA = LZ_ANY64, cB = LZ_ANY64, pB = LZ_ANY64, cC = LZ_ANY64, pC = LZ_ANY64, cZ = LZ_ANY64, pZ = LZ_ANY64;
            const bool haveBack = cand && ep >= 8u;                  // then p >= 16 as well
            const bool have24 = p + 24u <= E;
            if (cand) {                                              // one batch, straight-line (p + 16 <= E - 4)
                const u32 zb = haveBack ? 8u : 0u, fc = have24 ? 16u : 0u;
                cA = lz_ld64(src + ep); cB = lz_ld64(src + ep + 8u); pB = lz_ld64(src + p + 8u);
                cC = lz_ld64(src + (ep + fc)); pC = lz_ld64(src + (p + fc));
                cZ = lz_ld64(src + (ep - zb)); pZ = lz_ld64(src + (p - zb));
            }
            nextBytes = lz_ld64(src + pAhead);                       // next round of this run (consumed only if no lane accepts)
            // lengths from the batch: forward from p (exact when the difference or the limit lies inside the fetched bytes, else
            // 0xFFFF = at least `seen`), equal bytes among the 8 behind (cbk)
            const bool ok4 = cand && (u32)cA == first4;              // fastbig.h:91
            u32 fwd = 0xFFFFu, cbk = 0u;
            const u32 seen = have24 ? 24u : 16u;
            {
                const u64 x = bytes ^ cA, y = pB ^ cB, y2 = pC ^ cC, z = pZ ^ cZ;
                const u32 common = x ? lz_ctz64(x) >> 3 : y ? 8u + (lz_ctz64(y) >> 3) : (have24 && y2) ? 16u + (lz_ctz64(y2) >> 3) : seen;
                const u32 room = matchlimit - p;                     // p < matchlimit for every valid slot
                if (common < seen || room <= seen) fwd = common < room ? common : room;
                cbk = !haveBack ? 0u : z ? lz_clz64(z) >> 3 : 8u;
            }
            lz_pin(fwd); lz_pin(cbk);
            // the long-offset rule, fastbig.h:96 / :145: (forward count behind the first 4 bytes) + (backward extension) >= 16.
            // The post-match probe (slot 1 of a special run) does not extend backwards.
            u32 probeLane = (special != 0u && v0 == 0u) ? 1u : 64u;  // uniform: the lane that is a post-match probe
            const bool noBack = lane == probeLane;
            const u32 roomB = noBack ? 0u : ((p - anchor) < ep ? (p - anchor) : ep);   // :94 both bounds (anchor <= p for every probing slot)
            const bool backExact = roomB <= cbk || (haveBack && cbk < 8u);
            const u32 backLB = backExact ? (cbk < roomB ? cbk : roomB) : (haveBack ? 8u : 0u);
            const u32 fwdLB = fwd != 0xFFFFu ? fwd : seen;
            const bool shortOff = age < LZ_16BIT_OFFSET;
            const bool accept = ok4 && (shortOff || fwdLB - 4u + backLB >= LZ_MM_LONGOFF);
            const bool undecided = ok4 && !accept && !(fwd != 0xFFFFu && backExact);
            u64 okMask = lz_ballot(accept);                          // uniform
            u64 undMask = lz_ballot(undecided);
            // (emulator build: LZ_STAT 1 a long-offset lane accepted from the fetched bytes, 2 refused from them, 3 / 4 an undecided lane
            //  measured and accepted / refused, 5 a winner that is the post-match probe behind a long offset, 6 a long-offset winner
            //  with a backward extension, 7 a sequence pushed from inside a round)
            if (lz_ballot(accept && !shortOff)) LZ_STAT(1);
            if (lz_ballot(ok4 && !accept && !undecided)) LZ_STAT(2);
            u32 exLane = 64u, exF = 0, exB = 0;                      // a winner measured below: its exact lengths
            while (undMask) {                                        // undecided lanes in front of the first accepting one, in lane order
                const u32 j = lz_ctz64(undMask);
                if (okMask && j > lz_ctz64(okMask)) break;
                undMask &= undMask - 1ull;
                const u32 Pj = lz_readlane(p, j), Mj = lz_readlane(ep, j);
                const bool nbj = j == probeLane;
                const u32 f = 4u + lz_count_fwd(src, Pj + 4u, Mj + 4u, matchlimit);
                const u32 b = nbj ? 0u : lz_count_back(src, Pj, Mj, anchor);
                if (f - 4u + b >= LZ_MM_LONGOFF) { LZ_STAT(3); okMask |= 1ull << j; exLane = j; exF = f; exB = b; break; }
                LZ_STAT(4);
            }
            const u64 validMask = lz_ballot(valid);                  // uniform, a prefix of lanes
            u32 w = 0;
            u64 commit = validMask;
            if (okMask) {
                w = lz_ctz64(okMask); commit = validMask & (~0ull >> (63u - w));
#if LZ_FASTBIG_CHAIN
                u64 deadMask = 0;                                    // lanes inside the matches of chained sequences
                while (v0 == 0u) {                                   // first round of a run: consecutive positions behind the winner
                    const u32 Pw = lz_readlane(p, w), Mw = lz_readlane(ep, w);
                    const u32 fw = w == exLane ? exF : lz_readlane(fwd, w);
                    if (fw == 0xFFFFu) break;
                    const u32 bk = w == probeLane ? 0u : w == exLane ? exB : lz_back_from(lz_readlane(cbk, w), Pw, Mw, anchor);
                    if (bk == 0xFFFFu) break;
                    const u32 ipn = Pw + fw, l1 = w + fw;            // fastbig.h:128: ip behind the sequence, and its lane
                    if (ipn > mflimit || l1 > 63u) break;
                    const u64 from1 = ~0ull << l1;
                    // the accept test of the lanes from ip on, with the anchor at ip (:94: ip + back > anchor) and lane l1 as the
                    // post-match probe (:143-146: no backward extension)
                    const u32 rb2 = lane == l1 ? 0u : ((p - ipn) < ep ? (p - ipn) : ep);      // (lanes below l1: garbage, masked off)
                    const bool be2 = rb2 <= cbk || (haveBack && cbk < 8u);
                    const u32 bl2 = be2 ? (cbk < rb2 ? cbk : rb2) : (haveBack ? 8u : 0u);
                    const bool acc2 = ok4 && (shortOff || fwdLB - 4u + bl2 >= LZ_MM_LONGOFF);
                    const bool und2 = ok4 && !acc2 && !(fwd != 0xFFFFu && be2);
                    const u64 ok2 = lz_ballot(acc2) & from1;
                    if (!ok2) break;
                    const u32 w2 = lz_ctz64(ok2);                    // the next accepting lane: probe of ip or a visit of the next run
                    if (lz_ballot(und2) & from1 & (~0ull >> (63u - w2))) break;    // its lengths would decide: left to the next round
                    const u64 put2 = 1ull << (l1 - 2u);              // put(ip-2), fastbig.h:133
                    const u64 dead2 = deadMask | ((from1 ^ (~0ull << (w + 1u))) & ~put2);
                    const u64 readers = put2 | (from1 & (~0ull >> (63u - w2)));
                    // a lane whose slot value came from the registers of a lane inside the match read a put that never happened
                    const bool stale = jPrev < 64u && ((dead2 >> jPrev) & 1ull);
                    if (lz_ballot(stale) & readers) break;
                    LZ_STAT(7);
                    if (Pw - Mw >= LZ_16BIT_OFFSET) { if (w == probeLane) LZ_STAT(5); else if (bk) LZ_STAT(6); }
                    lz_seq_push_liz(st, Pw - bk - anchor, fw + bk, Pw - Mw);      // fastbig.h:127
                    anchor = ipn; probeLane = l1; exLane = 64u;
                    deadMask = dead2; commit |= readers; w = w2;
                }
#endif
            }
            // settle: the last committed lane of every table slot stores its entry; slots behind the winner never happened
            {
                const u64 c = grp & commit;
                if (valid && (c >> lane) == 1ull) { table.set(h, mine); if (codes) lz_fb_code_put(codes, h, lz_fb_code(myChk)); }
                lz_converge();
            }
            table.sync();
            lz_lds_sync();
            if (okMask) {
                P = lz_readlane(p, w); M = lz_readlane(ep, w);
                if (w == exLane) { ml = exF; back = exB; }
                else {
                    ml = lz_readlane(fwd, w);
                    back = w == probeLane ? 0u : lz_back_from(lz_readlane(cbk, w), P, M, anchor);
                    if (ml == 0xFFFFu) ml = 4u + lz_count_fwd(src, P + 4u, M + 4u, matchlimit);     // fastbig.h:93
                    if (back == 0xFFFFu) back = lz_count_back(src, P, M, anchor);                    // :95
                }
                if (P - M >= LZ_16BIT_OFFSET) { if (w == probeLane) LZ_STAT(5); else if (back) LZ_STAT(6); }
                break;
            }
            if (validMask != ~0ull) goto tail;                       // ran into mflimit without a match (:79)
            v0 += 64u;
        }
        P -= back; M -= back; ml += back;                            // :99-100
        ip = P + ml;
        special = 1u;
        lz_slot_pos(ip, 1u, lane, mflimit, pNext, validNext, putOnlyNext);
        if (ip > mflimit) validNext = false;                         // fastbig.h:130: there is no next run
        nextBytes = lz_ld64(src + (validNext ? pNext : S));
        lz_seq_push_liz(st, P - anchor, ml, P - M);                  // fastbig.h:127 (encoded later, in parallel)
        anchor = ip;
        if (ip > mflimit) goto tail;                                 // :130
    }
tail:
    if (st.nseq & (LZ_SEQ_RING - 1u)) lz_seq_flush(st);
    st.lastLits = E - anchor;                                        // fastbig.h:166-169
}
