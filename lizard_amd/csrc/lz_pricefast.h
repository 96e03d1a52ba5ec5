// lz_pricefast.h — priceFast parser + LIZv1 token encoder on one wavefront (levels 21/41, 22/42).
//
// Bit-exact with reference lib/lizard_parser_pricefast.h:132-249 (+ Lizard_FindMatchFast :3-87,
// Lizard_FindMatchFaster :90-128) and lib/lizard_compress_liz.h:43-179, zero-initialised state.
//
// Wave mapping.  priceFast probes EVERY position until one yields a match, so a round is simply the
// next 64 consecutive positions, one per lane.  Per lane: repeat-offset test first (it wins and hides
// the hash candidate, pricefast.h:19-31), else the hash candidate with the long-offset length rule.
// The table update is conditional (":170-171": put unless the slot already holds a position less than
// 8 bytes back), so when several lanes of a round share a slot the in-order view of each lane is
// obtained by walking that group's members in lane order with scalar code; every lane also records
// the slot value AFTER its own update, and the last same-slot lane up to the winner stores it.
// Everything after the winner (back-extension, the one-step lazy re-search at ip+ml-2 and its overlap
// arbitration, encoding) is a serial chain per sequence and runs as wave-uniform scalar code around
// the wave-parallel byte-compare / copy helpers.
//
// Included from lz_block.h after the shared helpers.
#pragma once

#define LZ_16BIT_OFFSET 65536u      // LIZARD_MAX_16BIT_OFFSET, lizard_common.h:83
#define LZ_MM_LONGOFF   16u         // MM_LONGOFF, lizard_common.h:84 (minMatchLongOff of levels 20-29/40-49)

// LIZv1 sequence (reference lib/lizard_compress_liz.h:43-165). M == P encodes "repeat last offset".
// Updates last_off like the reference (liz.h:119,135). All lanes call; values uniform.
LZ_DEV void lz_emit_lizv1(const u8* src, u32 anchor, u32 P, u32 ml, u32 M, LzStreams& st, u32& last_off)
{
    const u32 lane = lz_lane();
    const u32 L = P - anchor, off = P - M;
    const bool longOff = off >= LZ_16BIT_OFFSET;
    u32 extLw, extLn, extMw, extMn, token;
    lz_len_ext(L >= 7u, L - 7u, extLw, extLn);
    const u32 litTok = L >= 7u ? 7u : L;
    if (longOff) {                                               // liz.h:96-121
        const u32 m = ml - LZ_MM_LONGOFF;
        lz_len_ext(m >= 31u, m - 31u, extMw, extMn);
        token = m >= 31u ? 31u : m;
    } else {                                                     // liz.h:122-148
        lz_len_ext(ml >= 15u, ml - 15u, extMw, extMn);
        token = litTok | (off == 0 ? 128u : 0u) | ((ml >= 15u ? 15u : ml) << 3);
    }
    // literals-stream record: [literal-length escape][literals][match-length escape]
    const u32 oExtM = extLn + L, R = oExtM + extMn;
    u8* out = st.lit + st.nlit;
    for (u32 i = lane; i < R; i += 64u) {
        u32 b;
        if (i < extLn)       b = extLw >> (8u * i);
        else if (i < oExtM)  b = src[anchor + (i - extLn)];
        else                 b = extMw >> (8u * (i - oExtM));
        out[i] = (u8)b;
    }
    st.nlit += R;
    if (lane == 0) {
        u8* f = st.flags + st.nflags;
        if (longOff) {
            if (L > 0) { f[0] = (u8)(litTok | 128u); f[1] = (u8)token; }   // literal-only token first, liz.h:83-93
            else f[0] = (u8)token;
            lz_st24(st.off24 + st.noff24, off);
        } else {
            f[0] = (u8)token;
            if (off != 0) lz_st16(st.off16 + st.noff16, off);
        }
    }
    lz_converge();
    if (longOff) { st.nflags += (L > 0) ? 2u : 1u; st.noff24 += 3u; last_off = off; }
    else { st.nflags += 1u; if (off != 0) { st.noff16 += 2u; last_off = off; } }
}

// Sub-block [S,E) of the block at src. windowLog 22 / minMatchLongOff 16 are the level-21/22 values
// (lizard_common.h:249-250).  table: 2^HASHLOG positions (LZ_EMPTY = never written); tag: 2^TAGLOG bytes.
// Table: 2^HASHLOG slots of 24 bits (u16 + u8 arrays, LzTab without check bits) holding block-relative
// positions, LZ_EMPTY24 when never written: 48 KiB of LDS at HASHLOG 14; or the same values in u32 slots
// (LzTab32) when the table lives in global memory.
// Positions must stay below 2^24 - 1: blocks up to 16 MiB (the launcher refuses larger ones at these levels).
#define LZ_EMPTY24 0xFFFFFFu
template <int HASHLOG, int TAGLOG, class TAB>
LZ_DEV void lz_parse_pricefast(const u8* src, u32 S, u32 E, const TAB& table, u8* tag, LzStreams& st)
{
    const u32 lane = lz_lane();
    const u64 laneBit = 1ull << lane;
    const u64 lanesBelow = laneBit - 1ull;
    const u32 maxDist = (1u << 22) - 1u;
    u32 anchor = S;                                              // uniform
    u32 last_off = 0;                                            // uniform; Lizard_initBlock, lizard_compress.c:137
    if (E - S < LZ_MFLIMIT + 1u) { lz_emit_last_literals(src, anchor, E, st); return; }
    const u32 mflimit = E - LZ_MFLIMIT, matchlimit = E - LZ_LASTLITERALS;
    u32 ip = S + 1u;                                             // uniform, pricefast.h:155
    for (;;) {
        // ---------------- search: 64 consecutive positions per round ----------------
        u32 P = 0, M = 0;
        for (;;) {
            if (ip >= mflimit) goto tail;                        // pricefast.h:158
            const u32 p = ip + lane;
            const bool valid = p < mflimit;
            const u32 lowPos = p > maxDist ? p - maxDist : 0u;   // pricefast.h:11-13, per probe
            u32 h = 0, e = LZ_EMPTY24, first4 = 0;
            if (valid) {
                const u64 bytes = lz_ld64(src + p);
                first4 = (u32)bytes;
                h = lz_hash5<HASHLOG>(bytes);
                e = lz_tab_get(table, h);                        // pricefast.h:160,168 (old value)
                tag[h & ((1u << TAGLOG) - 1u)] = (u8)lane;
            }
            lz_wave_sync();
            const bool lost = valid && tag[h & ((1u << TAGLOG) - 1u)] != (u8)lane;
            u64 pend = lz_ballot(lost);
            u64 grp = laneBit;
            u32 tAfter = (e >= p || p >= e + LZ_MIN_OFFSET) ? p : e;     // pricefast.h:170-171 when alone in the slot
            while (pend) {                                       // same-slot lanes: replay the puts in lane order
                const u32 f = lz_ctz64(pend);
                const u32 hv = lz_readlane(h, f);
                const bool mine = valid && h == hv;
                const u64 g = lz_ballot(mine);
                u32 t = lz_readlane(e, f);                       // slot value before this round (uniform)
                for (u64 m = g; m; m &= m - 1ull) {
                    const u32 k = lz_ctz64(m), pk = ip + k;
                    if (lane == k) e = t;
                    t = (t >= pk || pk >= t + LZ_MIN_OFFSET) ? pk : t;
                    if (lane == k) tAfter = t;
                }
                if (mine) grp = g;
                pend &= ~g;
            }
            // Lizard_FindMatchFast, pricefast.h:3-87
            bool rep = false, hashOk = false;
            if (valid) {
                if (last_off >= LZ_MIN_OFFSET && p >= last_off && p - last_off >= lowPos)
                    rep = lz_ld32(src + p - last_off) == first4;                          // :19-31, returns at once
                if (!rep && e < p && e >= lowPos && p - e >= LZ_MIN_OFFSET && lz_ld32(src + e) == first4) {   // :63-67
                    if (p - e < LZ_16BIT_OFFSET) hashOk = true;
                    else                                                                  // :69: needs ml >= minMatchLongOff
                        hashOk = p + 16u <= matchlimit && lz_ld32(src + p + 4) == lz_ld32(src + e + 4)
                              && lz_ld64(src + p + 8) == lz_ld64(src + e + 8);
                }
            }
            const u64 okMask = lz_ballot(rep || hashOk);
            const u64 validMask = lz_ballot(valid);
            u32 w = 0;
            u64 commit = validMask;
            if (okMask) { w = lz_ctz64(okMask); commit = validMask & (~0ull >> (63u - w)); }
            if ((commit & laneBit) && (grp & commit & ~(lanesBelow | laneBit)) == 0) lz_tab_set(table, h, tAfter);
            lz_wave_sync();
            if (okMask) {
                P = lz_readlane(p, w);
                M = lz_readlane(rep ? p - last_off : e, w);
                break;
            }
            ip += lz_popc64(validMask);                          // "ip++" for every probed position, :173
        }
        // ---------------- winner: lengths, lazy re-search, encode ----------------
        LZ_PROF(st, 0);                                          // search rounds
        {
            u32 ml, back0;
            lz_count_both(src, P, M, matchlimit, anchor, ml, back0);                      // both lengths, one round trip
            u32 ml2 = 0, start2 = 0, ref2 = 0, ref = M, back2 = 0;
            ip = P;
            if (ip - ref == last_off) { ref = ip; goto encode; }                          // :174 -> repeat offset, no lazy step
            ip -= back0; ref -= back0; ml += back0;                                       // :176-182
        search:
            LZ_PROF(st, 2);                                      // winner lengths / arbitration
            if (ip + ml >= mflimit) goto encode;                                          // :185
            start2 = ip + ml - 2u;
            {
                const u32 h2 = lz_hash5<HASHLOG>(lz_ld64(src + start2));
                const u32 e2 = lz_tab_get(table, h2);
                const u32 low2 = start2 > maxDist ? start2 - maxDist : 0u;
                ml2 = 0; back2 = 0;
                if (e2 < start2 && e2 >= low2 && start2 - e2 >= LZ_MIN_OFFSET) {                            // :106-110
                    u32 mlt;                                                      // 4-byte test, length and :195-201 in one round trip
                    lz_count_both(src, start2, e2, matchlimit, ip, mlt, back2);
                    if (mlt >= 4u && (mlt >= LZ_MM_LONGOFF || start2 - e2 < LZ_16BIT_OFFSET)) { ml2 = mlt; ref2 = e2; }   // :112
                }
                lz_wave_sync();
                if (lane == 0 && (e2 >= start2 || start2 >= e2 + LZ_MIN_OFFSET)) lz_tab_set(table, h2, start2);   // :190-191
                lz_wave_sync();
            }
            LZ_PROF(st, 1);                                      // lazy re-search (table get/set, candidate, count)
            if (!ml2) goto encode;
            start2 -= back2; ref2 -= back2; ml2 += back2;                                 // :195-201
            if (ml2 <= ml) { ml2 = 0; goto encode; }                                      // :203
            if (start2 <= ip) { ip = start2; ref = ref2; ml = ml2; ml2 = 0; goto encode; }            // :205-210
            if (start2 - ip < 3u) { ip = start2; ref = ref2; ml = ml2; ml2 = 0; goto search; }        // :212-217
            if (start2 < ip + ml) {                                                       // :219-228
                const u32 correction = ml - (start2 - ip);
                start2 += correction; ref2 += correction; ml2 -= correction;
                if (ml2 < 3u) ml2 = 0;
                if (ml2 < LZ_MM_LONGOFF && start2 - ref2 >= LZ_16BIT_OFFSET) ml2 = 0;
            }
        encode:
            LZ_PROF(st, 2);
            lz_emit_lizv1(src, anchor, ip, ml, ref, st, last_off);                        // :231
            LZ_PROF(st, 3);                                      // LIZv1 encode into the staging areas
            ip += ml; anchor = ip;
            if (ml2) { ip = start2; ref = ref2; ml = ml2; ml2 = 0; goto search; }         // :233-238
        }
    }
tail:
    lz_emit_last_literals(src, anchor, E, st);
}
