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
// arbitration) is a serial chain per sequence and runs as wave-uniform scalar code around the
// wave-parallel byte-compare helpers.
//
// Encoding is not on the serial path (same structure as the fast parser, lz_block.h): the parse appends
// (literal run, match length, offset | 0 = repeat) to a sequence list through an LDS ring and keeps the four
// stream sizes up to date, so the container's raw-fallback rules (lizard_compress.c:201,228) are decided
// before any output exists; lz_encode_lizv1 then writes tokens, literals and both offset streams 64
// sequences per step straight into their final place in dst (staging only in front of the Huffman stage).
//
// Memory latency structure (round 3): ONE exposed memory trip per sequence.
//   * Every table form carries check bits (a hash of the 4 bytes at the stored position) beside the position: a candidate whose
//     check differs from the probing position's cannot pass the reference's 4-byte test (pricefast.h:67 / :109) and is never
//     fetched — with hashLog 14 a 256 KiB block puts 16 positions into every bucket, so nearly every probe finds a false one.
//   * The lanes whose hash candidate survives fetch, in ONE batch together with the repeat-offset candidate of every lane
//     (contiguous, so a line or two per load), everything the winner needs: 24 bytes forward at candidate and position and 8
//     backward — the 4-byte tests, the long-offset length rule (:69), the winner's forward length (< 24) and its backward
//     extension (< 8) all come out of that one trip.
//   * The source bytes of the round live in a register window of 128 positions (A: position ip + lane, B: ip + 64 + lane, 8
//     bytes each).  The next round starts at most 64 positions further in the common case, so its bytes are a lane shift of
//     the window (ds_bpermute) and only the far half is requested from memory — one round ahead of its use.  The lazy step's
//     probe position (ip + ml - 2) is read out of the same window, and its candidate is fetched only if its check bits agree.
// Included from lz_block.h after the shared helpers.
#pragma once

#define LZ_16BIT_OFFSET 65536u      // LIZARD_MAX_16BIT_OFFSET, lizard_common.h:83
#define LZ_MM_LONGOFF   16u         // MM_LONGOFF, lizard_common.h:84 (minMatchLongOff of levels 20-29/40-49)

// ---- hash tables of the priceFast levels (ours; only the parse RESULT is pinned by the reference) ----
// What a slot tells a probe at position p is the AGE of its entry, age(p, e) = how far back the stored position lies: the
// reference's tests (:63-65 "matchIndex < current, >= lowLimit, >= MIN_OFFSET back", :170-171 "put unless < MIN_OFFSET back")
// are tests on the age.  A slot that was never written, or whose entry has left the 4 MiB window, is dead: its age is
// above maxDistance, it fails every test like the reference's zeroed slot fails "e >= lowLimit", and it is always overwritten.
// Every form carries check bits beside the position (a hash of the 4 bytes there).
// The u32 forms keep the position modulo 2^24 and are exact for ANY block size: a sweep every 2^22 positions re-stamps the
// slots that are dead by then as "exactly 2^23 old" (what a fresh table holds), so no age ever reaches 2^24.
//   LzTab24c  LDS, u16 + u8 arrays: 18-bit position + 6 check bits, blocks <= 256 KiB — the benchmark configuration —
//             48 KiB at hashLog 14: three tables per CU
//   LzTab32G / LzTab32L   u32 slots, position mod 2^24 + 8 check bits, any block size: in LDS (64 KiB at hashLog 14, two per
//             CU) for blocks above 256 KiB, in a global-memory slot (one sector per access) for the waves without an LDS
//             table and for hashLog 18 (levels 22/42)
#define LZ_EMPTY24 0xFFFFFFu
#define LZ_EMPTY18 0x3FFFFu
struct LzTab24c {
    LZ_LDS u16* lo; LZ_LDS u8* hi;
    static constexpr u32 kEmpty = LZ_EMPTY18;
    static constexpr bool kSpecPut = true;
    LZ_DEVM static u32 pos(u32 raw) { return raw & 0x3FFFFu; }
    LZ_DEVM static u32 chk(u32 raw) { return raw >> 18; }
    LZ_DEVM static u32 chkOf(u32 first4) { return (first4 * 2654435761u) >> 26; }
    LZ_DEVM static u32 make(u32 p, u32 c) { return p | (c << 18); }
    LZ_DEVM void specPut(u32 h, u32 p) const { lo[h] = (u16)p; }
    LZ_DEVM bool specLost(u32 h, u32 p) const { return lo[h] != (u16)p; }
    static constexpr bool kSweeps = false;                       // full positions: blocks <= 256 KiB
    LZ_DEVM static u32 age(u32 p, u32 e) { return p - e; }       // empty (0x3FFFF, never reached by p) and nothing else wraps: dead
    LZ_DEVM static u32 dead(u32) { return LZ_EMPTY18; }
    LZ_DEVM u32  get(u32 h, u32) const { return (u32)lo[h] | ((u32)hi[h] << 16); }
    LZ_DEVM void set(u32 h, u32 v) const { lo[h] = (u16)v; hi[h] = (u8)(v >> 16); }
    LZ_DEVM void sync() const { lz_lds_sync(); }
};
#define LZ_TAB24C_BYTES(HASHLOG) (3u << (HASHLOG))
struct LzTab32G {                                                // slots in a global-memory slot of the wave
    LZ_GLOBAL u32* w;
    static constexpr u32 kEmpty = LZ_EMPTY24;
    static constexpr bool kSpecPut = false;
    LZ_DEVM static u32 pos(u32 raw) { return raw & 0xFFFFFFu; }
    LZ_DEVM static u32 chk(u32 raw) { return raw >> 24; }
    LZ_DEVM static u32 chkOf(u32 first4) { return (first4 * 2654435761u) >> 24; }
    LZ_DEVM static u32 make(u32 p, u32 c) { return (p & 0xFFFFFFu) | (c << 24); }
    static constexpr bool kSweeps = true;
    LZ_DEVM static u32 age(u32 p, u32 e) { return (p - e) & 0xFFFFFFu; }
    LZ_DEVM static u32 dead(u32 p) { return (p - (1u << 23)) & 0xFFFFFFu; }       // 2^23 old at position p
    // Occupancy summary (LDS, optional; levels 22/42 with their 2^18 slots): as LzTabWide::occ in lz_block.h
    u32* occ = nullptr;
    u32 occShift = 0;
    LZ_DEVM void specPut(u32, u32) const {}
    LZ_DEVM bool specLost(u32, u32) const { return false; }
    LZ_DEVM u32  get(u32 h, u32 p) const
    {
        if (!occ) return w[h];
        const u32 b = h >> occShift;
        const bool oc = (occ[b >> 5] >> (b & 31u)) & 1u;
        const u32 v = w[oc ? h : 0u];
        return oc ? v : dead(p);
    }
    LZ_DEVM u32  raw(u32 i) const { return w[i]; }               // sweeps: the slot itself, no summary, no side effects
    LZ_DEVM void setRaw(u32 i, u32 v) const { w[i] = v; }
    LZ_DEVM void set(u32 h, u32 v) const
    {
        w[h] = v;
        if (occ) { const u32 b = h >> occShift; lz_lds_atomic_or(&occ[b >> 5], 1u << (b & 31u)); }
    }
    LZ_DEVM void sync() const { lz_table_sync(); }
};
struct LzTab32L {                                                // the same slots in LDS
    LZ_LDS u32* w;
    static constexpr u32 kEmpty = LZ_EMPTY24;
    static constexpr bool kSpecPut = true;
    LZ_DEVM static u32 pos(u32 raw) { return raw & 0xFFFFFFu; }
    LZ_DEVM static u32 chk(u32 raw) { return raw >> 24; }
    LZ_DEVM static u32 chkOf(u32 first4) { return (first4 * 2654435761u) >> 24; }
    LZ_DEVM static u32 make(u32 p, u32 c) { return (p & 0xFFFFFFu) | (c << 24); }
    static constexpr bool kSweeps = true;
    LZ_DEVM static u32 age(u32 p, u32 e) { return (p - e) & 0xFFFFFFu; }
    LZ_DEVM static u32 dead(u32 p) { return (p - (1u << 23)) & 0xFFFFFFu; }
    LZ_DEVM u32  raw(u32 i) const { return w[i]; }
    LZ_DEVM void setRaw(u32 i, u32 v) const { w[i] = v; }
    // speculative put of the low half only: the position's low 16 bits tell the lanes of a round apart (they differ by < 64)
    LZ_DEVM void specPut(u32 h, u32 p) const { ((LZ_LDS u16*)w)[2u * h] = (u16)p; }
    LZ_DEVM bool specLost(u32 h, u32 p) const { return ((LZ_LDS u16*)w)[2u * h] != (u16)p; }
    LZ_DEVM u32  get(u32 h, u32) const { return w[h]; }
    LZ_DEVM void set(u32 h, u32 v) const { w[h] = v; }
    LZ_DEVM void sync() const { lz_lds_sync(); }
};
template <int HASHLOG> LZ_DEV void lz_pf_tab_fresh(const LzTab32L& t) { for (u32 i = lz_lane(); i < (1u << HASHLOG); i += 64u) t.w[i] = LzTab32L::dead(0); }
// Re-stamp the slots whose entry has left the window at position Ps (age > maxDistance): every 2^22 positions (blocks above 4 MiB only)
#define LZ_PF_SWEEP_EVERY (1u << 22)
template <int HASHLOG, class TAB>
LZ_DEV void lz_pf_tab_sweep(const TAB& t, u32 Ps)
{
    for (u32 i = lz_lane(); i < (1u << HASHLOG); i += 64u) {
        const u32 v = t.raw(i);
        if (TAB::age(Ps, TAB::pos(v)) > (1u << 22) - 1u) t.setRaw(i, TAB::dead(Ps));
    }
}
template <int HASHLOG> LZ_DEV void lz_pf_tab_fresh(const LzTab32G& t)
{
    for (u32 i = lz_lane(); i < (1u << HASHLOG); i += 64u) t.w[i] = LzTab32G::dead(0);
    if (t.occ) for (u32 i = lz_lane(); i < (((1u << HASHLOG) >> t.occShift) >> 5); i += 64u) t.occ[i] = 0u;
}
template <int HASHLOG> LZ_DEV void lz_pf_tab_fresh(const LzTab24c& t)
{
    for (u32 i = lz_lane(); i < (1u << HASHLOG) / 2u; i += 64u) ((LZ_LDS u32*)t.lo)[i] = 0xFFFFFFFFu;
    for (u32 i = lz_lane(); i < (1u << HASHLOG) / 4u; i += 64u) ((LZ_LDS u32*)t.hi)[i] = 0x03030303u;     // position bits 16-17 set, check bits 0
}

// ---- sequence list (LIZv1): L < 2^18, ml < 2^18, off < 2^24 (0 = repeat the last offset) ----
// The parse pushes the list entry and nothing else (round 6; it used to keep the four stream sizes up to date with every sequence:
// ~55 scalar instructions on the one chain that bounds the level, a quarter of what a sequence pushed from registers costs).  The
// sizes the container's rules need (lizard_compress.c:201,228) are summed up over the finished list, 64 sequences per lane step.
LZ_DEV void lz_seq_push_liz(LzStreams& st, u32 L, u32 ml, u32 off)
{
    if (lz_lane() == 0) st.ring[st.nseq & (LZ_SEQ_RING - 1u)] = (u64)L | ((u64)ml << 18) | ((u64)off << 36);
    lz_converge();
    st.nseq += 1u;
    if ((st.nseq & (LZ_SEQ_RING - 1u)) == 0) lz_seq_flush(st);
}
// Stream sizes of a finished LIZv1 sequence list, exactly as the encoder will produce them (reference lib/lizard_compress_liz.h:43-165).
LZ_DEV void lz_seq_sizes_liz(LzStreams& st)
{
    u32 lit = 0, fl = 0, o16 = 0, o24 = 0;
    lz_wave_sync();                                              // the list's last burst
    for (u32 base = 0; base < st.nseq; base += 256u) {           // four loads in flight: a step is one memory trip long
        u64 q[4];
        #pragma unroll
        for (u32 k = 0; k < 4u; k++) { const u32 i = base + 64u * k + lz_lane(); q[k] = lz_ldq_s(&st.seq[i < st.nseq ? i : st.nseq - 1u]); }
        #pragma unroll
        for (u32 k = 0; k < 4u; k++) {
            const u32 i = base + 64u * k + lz_lane();
            const u32 L = (u32)q[k] & 0x3FFFFu, ml = (u32)(q[k] >> 18) & 0x3FFFFu, off = (u32)(q[k] >> 36);
            const bool longOff = off >= LZ_16BIT_OFFSET;
            const u32 m = longOff ? ml - LZ_MM_LONGOFF : ml, mSat = longOff ? 31u : 15u;
            if (i < st.nseq) {
                lit += lz_ext_len(L >= 7u, L - 7u) + L + lz_ext_len(m >= mSat, m - mSat);
                fl += (longOff && L > 0u) ? 2u : 1u;             // liz.h:83-93: literal-only token in front
                o16 += (!longOff && off != 0u) ? 2u : 0u;
                o24 += longOff ? 3u : 0u;
            }
        }
    }
    st.nlit = st.lastLits + lz_wave_reduce_add(lit);
    st.nflags = lz_wave_reduce_add(fl);
    st.noff16 = lz_wave_reduce_add(o16);
    st.noff24 = lz_wave_reduce_add(o24);
}

// Wave-parallel LIZv1 encoder (reference lib/lizard_compress_liz.h:43-179) over the sequence list of the sub-block
// starting at src+S: 64 sequences per step, one per lane.  Four wave prefix sums place every lane's output: bytes
// in the literals stream, literal source position, and (packed into one scan) tokens / off16 bytes / off24 bytes.
// Token forms (lizard_decompress_liz.h:1-6): [0_MMMM_LLL] 16-bit offset, [1_MMMM_LLL] repeat offset, 0..31 =
// 24-bit offset with ml-16 (31 = escape), preceded by a literal-only [1_0000_LLL] when the sequence has literals.
// Length escapes go to the LITERALS stream (SURVEY finding 4).  Trailing literals are appended raw.
template <int PAIRS>
LZ_DEV void lz_encode_lizv1(const u8* src, u32 S, const LzStreams& st, u8* litOut, u8* flagsOut, u8* off16Out, u8* off24Out)
{
    const u32 lane = lz_lane();
    u32 srcPos = S, outPos = 0, fPos = 0, o16Pos = 0, o24Pos = 0;        // uniform carries
    u64 qNext = st.nseq ? st.seq[lane < st.nseq ? lane : st.nseq - 1u] : 0ull;    // one step ahead, as in lz_encode_lz4
    for (u32 base = 0; base < st.nseq; base += 64u) {
        const u32 cnt = st.nseq - base < 64u ? st.nseq - base : 64u;
        u32 L = 0, ml = 0, off = 0, R = 0, adv = 0, packed = 0, tok = 0, litTok = 0;
        u32 extLw = 0, extLn = 0, extMw = 0, extMn = 0;
        bool longOff = false;
        const u64 q = qNext;
        { const u32 nx = base + 64u + lane; qNext = st.seq[nx < st.nseq ? nx : st.nseq - 1u]; }
        if (lane < cnt) {
            L = (u32)q & 0x3FFFFu; ml = (u32)(q >> 18) & 0x3FFFFu; off = (u32)(q >> 36);
            longOff = off >= LZ_16BIT_OFFSET;
            litTok = L >= 7u ? 7u : L;
            lz_len_ext(L >= 7u, L - 7u, extLw, extLn);
            if (longOff) {                                               // liz.h:96-121
                const u32 m = ml - LZ_MM_LONGOFF;
                lz_len_ext(m >= 31u, m - 31u, extMw, extMn);
                tok = m >= 31u ? 31u : m;
                packed = (L > 0u ? 2u : 1u) | (3u << 16);
            } else {                                                     // liz.h:122-148
                lz_len_ext(ml >= 15u, ml - 15u, extMw, extMn);
                tok = litTok | (off == 0u ? 128u : 0u) | ((ml >= 15u ? 15u : ml) << 3);
                packed = 1u | (off != 0u ? 2u << 8 : 0u);
            }
            R = extLn + L + extMn; adv = L + ml;
        }
        const u32 myOut = outPos + lz_wave_scan_excl_add(R);
        const u32 mySrc = srcPos + lz_wave_scan_excl_add(adv);
        const u32 pk = lz_wave_scan_excl_add(packed);
        if (lane < cnt) {
            u8* f = flagsOut + fPos + (pk & 255u);
            if (longOff) {
                if (L > 0u) { f[0] = (u8)(litTok | 128u); f[1] = (u8)tok; } else f[0] = (u8)tok;
                lz_st24(off24Out + o24Pos + (pk >> 16), off);
            } else {
                f[0] = (u8)tok;
                if (off != 0u) lz_st16(off16Out + o16Pos + ((pk >> 8) & 255u), off);
            }
            u8* r = litOut + myOut;
            for (u32 k = 0; k < extLn; k++) r[k] = (u8)(extLw >> (8u * k));
            r += extLn + L;
            for (u32 k = 0; k < extMn; k++) r[k] = (u8)(extMw >> (8u * k));
        }
        lz_copy_literal_runs<PAIRS>(src, litOut, mySrc, myOut + extLn, L);
        srcPos = lz_readlane(mySrc + adv, 63u);                          // lanes >= cnt hold zeros
        outPos = lz_readlane(myOut + R, 63u);
        const u32 pkEnd = lz_readlane(pk + packed, 63u);
        fPos += pkEnd & 255u; o16Pos += (pkEnd >> 8) & 255u; o24Pos += pkEnd >> 16;
    }
    lz_copy(litOut + outPos, src + srcPos, st.lastLits);                 // liz.h:168-179
}

// Sub-block [S,E) of the block at src. windowLog 22 / minMatchLongOff 16 are the level-21/22 values
// (lizard_common.h:249-250).  table: 2^HASHLOG slots (TAB::kEmpty = never written); tag: 2^TAGLOG bytes of LDS (global tables only).
// Positions must stay below TAB::kEmpty (the launcher picks the table form by block size).
#define LZ_PF_UNRESOLVED 0xFFFFu
#define LZ_PF_UNRESOLVED4 0xFFFEu    // only the 4-byte test is known to hold (a repeat-offset candidate tested again inside a round)
#ifndef LZ_PF_W0
#define LZ_PF_W0 64u                  // round width over a global-memory table right after a match.  Rounds 3: 32 (the lanes behind a
#endif                                // winner were probed for nothing, +2.8 %); with several sequences per round those lanes are the next
                                      // stretch: 64 is +3.7 % at level 21, +2.6 % at 41, +5.8 % on 4 MiB blocks (profiles/r04u_*)
// ---- several sequences out of one round (round 4) ----
// A round has paid for 64 consecutive positions: every lane holds its slot's value as the reference's serial walk would have found
// it (the same-slot replay), its hash candidate's bytes and its repeat-offset test.  The reference's next steps behind a winner are
// all positions of this round as long as the sequence ends inside it:
//   * the lazy step (pricefast.h:184-228) probes start2 = ip + ml - 2 — a lane of the round: its slot value, check bits, candidate
//     bytes (24 forward, 8 backward) are in registers, so Lizard_FindMatchFaster (:90-128) costs no table access and no memory trip,
//     and its conditional put (:190-191) is that lane's own put: the lane is simply committed;
//   * the next search (:158-173) probes ip' = end of the sequence, ip' + 1, ... — the lanes from ip' - ip0 on.  Their hash side is
//     unchanged; their repeat-offset side (:19-31) belongs to the OLD last_off, so when the sequence changed it the three loads at
//     p - last_off are issued again (contiguous: a line or two) and the lane's tests are redone — the only trip a chained sequence
//     costs.  A repeat-offset match changes nothing and costs none.
// What a lane found in its slot is exact iff every lane below it on the same slot happened (the replay walks ALL same-slot lanes in
// lane order = position order = the reference's order).  Lanes inside a match never happen, so a lane of the next stretch — or a
// lazy lane — with a same-slot lane below it that did not happen is "stale": the chain stops there and the next round starts at
// that position with a table settled for exactly the lanes that happened (the settle rule needs nothing else: the last committed
// lane of a slot stores what it left there).  Lengths that do not resolve from the fetched bytes (>= 24 forward, >= 8 backward), a
// lazy position beyond the round, or a lazy lane whose forward count was taken against its repeat candidate fall back to the
// memory-based steps below, after which the round is over.
#ifndef LZ_PF_CHAIN
#define LZ_PF_CHAIN 1
#endif
// backward run of a candidate pair from the 8 bytes fetched behind both (cb equal bytes; nothing fetched: haveBack false, cb 0),
// limited to roomB: exact, or LZ_PF_UNRESOLVED when it may go on past the fetched bytes.  Wave-uniform.
LZ_DEV u32 lz_pf_back_from(u32 cb, bool haveBack, u32 roomB)
{
    return (roomB <= cb || (haveBack && cb < 8u)) ? (cb < roomB ? cb : roomB) : LZ_PF_UNRESOLVED;
}
// A lane's tests from its fetched bytes: the repeat-offset candidate (r*) wins and hides the hash candidate (c*), pricefast.h:19-31;
// the hash candidate needs the 4-byte test (:67) and, behind a long offset, minMatchLongOff bytes (:69: 16 <= 24 fetched bytes).
// fwd = forward length against whichever candidate counts, exact when the difference (or the limit) lies inside the fetched bytes.
LZ_DEV void lz_pf_measure(bool repCand, bool hashCand, u64 bytes, u64 pB, u64 pC, u64 rA, u64 rB, u64 rC, u64 cA, u64 cB, u64 cC,
                          bool have24, u32 room, u32 dist, bool& rep, bool& hashOk, u32& fwd)
{
    const u32 first4 = (u32)bytes;
    rep = repCand && (u32)rA == first4;
    const u64 x = bytes ^ (rep ? rA : cA), y = pB ^ (rep ? rB : cB), y2 = pC ^ (rep ? rC : cC);
    const u32 seen = have24 ? 24u : 16u;
    const u32 common = x ? lz_ctz64(x) >> 3 : y ? 8u + (lz_ctz64(y) >> 3) : (have24 && y2) ? 16u + (lz_ctz64(y2) >> 3) : seen;
    fwd = LZ_PF_UNRESOLVED;
    if (common < seen || room <= seen) fwd = common < room ? common : room;
    hashOk = !rep && hashCand && (u32)cA == first4
          && (dist < LZ_16BIT_OFFSET || (have24 && common >= LZ_MM_LONGOFF && room >= LZ_MM_LONGOFF));
}

#if LZ_PF_CHAIN
template <int HASHLOG, int TAGLOG, class TAB>
LZ_DEV void lz_parse_pricefast(const u8* src, u32 S, u32 E, const TAB& table, u8* tag, LzStreams& st)
{
    const u32 lane = lz_lane();
    const u64 laneBit = 1ull << lane;
    const u64 lanesBelow = laneBit - 1ull;
    const u32 maxDist = (1u << 22) - 1u;
    const u32 tagMask = (1u << TAGLOG) - 1u;
    u32 anchor = S;                                              // uniform
    u32 last_off = 0;                                            // uniform; Lizard_initBlock, lizard_compress.c:137
    if (E - S < LZ_MFLIMIT + 1u) { st.lastLits = E - S; return; }
    const u32 mflimit = E - LZ_MFLIMIT, matchlimit = E - LZ_LASTLITERALS;
    u32 ip = S + 1u;                                             // uniform, pricefast.h:155
    constexpr bool kNarrow = !TAB::kSpecPut && LZ_PF_W0 < 64u;   // round width over a global-memory table: see the unchained form
    u32 W = kNarrow ? LZ_PF_W0 : 64u;                            // uniform
    u32 winBase = ip;                                            // uniform: register window, wA = 8 bytes at winBase + lane, wB = at winBase + 64 + lane
    u64 wA, wB;
    { const u32 qa = ip + lane, qb = ip + 64u + lane; wA = lz_ld64(src + (qa < mflimit ? qa : S)); wB = lz_ld64(src + (qb < mflimit ? qb : S)); }
    for (;;) {
        // ---------------- one round: 64 consecutive positions ----------------
        if (ip >= mflimit) goto tail;                            // pricefast.h:158
        if constexpr (TAB::kSweeps) if (ip >= st.sweepAt) { table.sync(); lz_pf_tab_sweep<HASHLOG>(table, ip); st.sweepAt = ip + LZ_PF_SWEEP_EVERY; table.sync(); }
        if (ip != winBase) {                                     // move the window (see the unchained form)
            const u32 d = ip - winBase;                          // uniform
            const u32 qa = ip + lane, qb = ip + 64u + lane;
            if (d == 64u) wA = wB;
            else if (d < 64u) {
                const u32 sl = (d + lane) & 63u;
                const u32 al = lz_shfl((u32)wA, sl), ah = lz_shfl((u32)(wA >> 32), sl);
                const u32 bl = lz_shfl((u32)wB, sl), bh = lz_shfl((u32)(wB >> 32), sl);
                wA = d + lane < 64u ? ((u64)al | ((u64)ah << 32)) : ((u64)bl | ((u64)bh << 32));
            }
            else wA = lz_ld64(src + (qa < mflimit ? qa : S));
            wB = lz_ld64(src + (qb < mflimit ? qb : S));
            winBase = ip;
        }
        const u32 ip0 = ip;                                      // uniform: lane k of this round is position ip0 + k
        LZ_STAT(55);
        const u32 p = ip0 + lane;
        const bool valid = p < mflimit && (!kNarrow || lane < W);
        const u32 lowPos = p > maxDist ? p - maxDist : 0u;       // pricefast.h:11-13, per probe
        const u64 bytes = wA;
        const u32 first4 = (u32)bytes;
        const u32 h = lz_hash5_plain<HASHLOG>(bytes);
        const u32 myChk = TAB::chkOf(first4);
        u32 e, ec;                                               // pricefast.h:160,168 (old value; garbage when !valid); its check bits
        { const u32 raw = table.get(kNarrow && !valid ? 0u : h, p); e = TAB::pos(raw); ec = TAB::chk(raw); }
        bool lost;                                               // same-slot lanes of this round (see the unchained form)
        if constexpr (TAB::kSpecPut) {
            lz_lds_sync();
            if (valid) table.specPut(h, p);
            lz_lds_sync();
            lost = valid && table.specLost(h, p);
        } else {
            if (valid) tag[h & tagMask] = (u8)lane;
            lz_lds_sync();
            lost = valid && tag[h & tagMask] != (u8)lane;
            lz_lds_sync();
        }
        LZ_PROF(st, 8);                                          // (instrumented build) round: window, hash, table read
        const u32 eOld = e, ecOld = ec;
        u64 pend = lz_ballot(lost);
        const bool anyGroups = pend != 0;                        // uniform: some lanes of this round share a slot
        u64 grp = laneBit;
        const bool putAlone = TAB::age(p, e) - 1u >= LZ_MIN_OFFSET - 1u;     // pricefast.h:170-171 when alone in the slot
        u32 tAfter = putAlone ? p : e, tcAfter = putAlone ? myChk : ec;
        while (pend) {                                           // same-slot lanes: replay the puts in lane order
            const u32 f = lz_ctz64(pend);
            const u32 hv = lz_readlane(h, f);
            const bool mine = valid && h == hv;
            const u64 g = lz_ballot(mine);
            // Alone on its slot after all: a false alarm of the tag array (global-memory tables: 64 lanes meet in 2^TAGLOG tag bytes,
            // about two unrelated pairs per round) — the lane's view and what it leaves behind are already those of a lane alone.
            // (Round 6: such turns were most of this phase, 10 % of the global-table waves' time.)
            if ((g & (g - 1ull)) == 0) { pend &= ~g; continue; }
            u32 t = lz_readlane(e, f);
            u32 tc = lz_readlane(ec, f);
            for (u64 m = g; m; m &= m - 1ull) {
                const u32 k = lz_ctz64(m), pk = ip0 + k;
                const u32 ck = lz_readlane(myChk, k);
                if (lane == k) { e = t; ec = tc; }
                const bool put = TAB::age(pk, t) - 1u >= LZ_MIN_OFFSET - 1u;
                t = put ? TAB::pos(TAB::make(pk, 0u)) : t; tc = put ? ck : tc;
                if (lane == k) { tAfter = t; tcAfter = tc; }
            }
            if (mine) grp = g;
            pend &= ~g;
        }
        LZ_PROF(st, 9);                                          // round: same-slot replay
        // candidates (Lizard_FindMatchFast, pricefast.h:3-87) and one batch of loads for both kinds
        const u32 ageE = TAB::age(p, e);
        const bool hashCand = valid && ageE >= LZ_MIN_OFFSET && ageE <= (p > maxDist ? maxDist : p) && ec == myChk;   // :63-65 + check bits
        e = p - ageE;                                            // the candidate's position in the block (hashCand lanes)
        const bool have24 = p + 24u <= E;
        const u32 fc = have24 ? 16u : 0u;
        const u32 pp = valid ? p : S;
        const u32 room = matchlimit - p;                         // p < matchlimit for every valid slot
        const u64 pB = lz_ld64(src + pp + 8u), pC = lz_ld64(src + pp + fc);
        u64 rA, rB, rC;
        {
            const bool repCand = valid && last_off >= LZ_MIN_OFFSET && p >= last_off && p - last_off >= lowPos;   // :19
            const u32 rp = repCand ? p - last_off : S;
            rA = lz_ld64(src + rp); rB = lz_ld64(src + rp + 8u); rC = lz_ld64(src + rp + fc);
        }
        u64 cA = 0, cB = 0, cC = 0, cZ = 0, pZ = 0;
        const bool haveBack = hashCand && e >= 8u;               // then p >= 16 as well
        if (hashCand) {                                          // one batch, straight-line
            const u32 zb = haveBack ? 8u : 0u;
            cA = lz_ld64(src + e); cB = lz_ld64(src + e + 8u); cC = lz_ld64(src + e + fc);
            cZ = lz_ld64(src + (e - zb)); pZ = lz_ld64(src + (p - zb));
        }
        lz_converge();
        bool rep, hashOk;
        u32 fwd;
        lz_pf_measure(valid && last_off >= LZ_MIN_OFFSET && p >= last_off && p - last_off >= lowPos, hashCand, bytes, pB, pC, rA, rB, rC,
                      cA, cB, cC, have24, room, p - e, rep, hashOk, fwd);
        u32 cb;                                                  // equal bytes in the 8 behind position and hash candidate (+ 16: something was fetched)
        { const u64 z = pZ ^ cZ; cb = !haveBack ? 0u : 16u + (z ? lz_clz64(z) >> 3 : 8u); }
        lz_pin(fwd); lz_pin(cb);                                 // computed here, under this batch's counted wait
        // What the scalar steps below read from a lane, packed so that one v_readlane brings all of it:
        //   hinfo  the hash side, which does not depend on last_off: forward length against the hash candidate (16 bits) | cb << 16 |
        //          hashCand << 21 | "known" << 22 (a lane whose forward count was taken against its repeat candidate has no hash side
        //          yet) | "passes when no repeat candidate hides it" << 23
        //   winfo  the side that counts if the lane wins: forward length | cb << 16 | "repeat candidate" << 21
        // A changed last_off only needs the 4-byte repeat test again: winfo is rebuilt from hinfo.
        bool hashKnown = !rep || !hashCand;
        u32 hinfo = fwd | (cb << 16) | ((u32)hashCand << 21) | ((u32)hashKnown << 22) | ((u32)hashOk << 23);
        u32 winfo = fwd | (cb << 16) | ((u32)rep << 21);
        u64 okMask = lz_ballot(rep || hashOk);
        const u64 validMask = lz_ballot(valid);                  // a prefix of lanes
        LZ_PROF(st, 10);                                         // round: candidate bytes, tests

        // ---------------- the sequences of this round ----------------
        u64 commit = 0;                                          // uniform: lanes whose probe + put happened
        u32 c = 0;                                               // uniform: first lane of the stretch being searched
        u32 slow = 0;                                            // uniform: 1 = winner lengths from memory, 2 = lazy step from memory
        u32 P = 0, M = 0, ml = 0, back0 = 0;
        u32 ml2 = 0, start2 = 0, ref2 = 0, ref = 0, back2 = 0;
        u32 measuredOff = last_off;                              // the last_off the lanes' repeat tests belong to
        for (;;) {
            const u64 seg = ~0ull << c;
            const u64 ok2 = okMask & seg;
            const u32 w = ok2 ? lz_ctz64(ok2) : 63u;
            const u64 readers = validMask & seg & (~0ull >> (63u - w));     // the probes up to the winner: all of them happen
            if (c && anyGroups) {
                // (Working a stale lane's view out again from the lanes that did happen does not pay: same-slot lanes of one round
                //  are repeats of the same bytes, so 9 times out of 10 the corrected value IS a candidate — whose bytes were never
                //  requested.  Measured under emulation, round 4.)
                const bool staleLane = (grp & lanesBelow & ~(commit | readers)) != 0;
                if (lz_ballot(staleLane) & readers) { LZ_STAT(32); ip = ip0 + c; if constexpr (kNarrow) W = LZ_PF_W0; break; }
            }
            commit |= readers;
            if (!ok2) {                                          // "ip++" for every probed position, :173
                if (c) LZ_STAT(33);
                ip = ip0 + lz_popc64(validMask);
                if constexpr (kNarrow) { if (c) W = LZ_PF_W0; W = W < 32u ? 2u * W : 64u; }
                break;
            }
            if constexpr (kNarrow) W = LZ_PF_W0;
            // ---- winner: lane w ----
            const u32 wi = lz_readlane(winfo, w);
            P = ip0 + w;
            M = ((wi >> 21) & 1u) ? P - last_off : lz_readlane(e, w);
            ml = wi & 0xFFFFu;
            ip = P; ref = M;
            const bool repMatch = ip - ref == last_off;          // :174 -> repeat offset: no backward extension, no lazy step
            { const u32 cbw = (wi >> 16) & 31u; back0 = repMatch ? 0u : lz_pf_back_from(cbw & 15u, cbw >= 16u, (P - anchor) < M ? (P - anchor) : M); }   // :176-180
            if (ml >= LZ_PF_UNRESOLVED4 || back0 == LZ_PF_UNRESOLVED) { LZ_STAT(34); slow = 1u; break; }
            if (c) LZ_STAT(35);                                  // a winner found in a later stretch of the round
            if (repMatch) ref = ip;
            else { ip -= back0; ref -= back0; ml += back0; }    // :176-182
            bool lazy = !repMatch;
            for (;;) {                                           // this winner's sequence, and the second match's when one is kept (:233-238)
                ml2 = 0;
                while (lazy) {                                   // "search:", :184-228
                    if (ip + ml >= mflimit) break;               // :185
                    start2 = ip + ml - 2u;
                    const u32 ls = start2 - ip0;                 // its lane (above every lane that happened so far)
                    if (ls > 63u || !((validMask >> ls) & 1ull)) { LZ_STAT(37); slow = 2u; break; }
                    const u32 hi = lz_readlane(hinfo, ls);
                    if (!((hi >> 22) & 1u)) { LZ_STAT(38); slow = 2u; break; }                        // its forward count belongs to a repeat candidate
                    if (anyGroups && (lz_readlane64(grp, ls) & ((1ull << ls) - 1ull) & ~commit) != 0) { LZ_STAT(52); slow = 2u; break; }   // stale: see above
                    u32 mlt = 0, e2 = 0, b2 = 0;
                    bool found = false;
                    if ((hi >> 21) & 1u) {                       // :106-110 (check bits differ: :109 fails)
                        mlt = hi & 0xFFFFu;
                        if (mlt == LZ_PF_UNRESOLVED) { LZ_STAT(39); slow = 2u; break; }
                        e2 = lz_readlane(e, ls);
                        if (mlt >= 4u && (mlt >= LZ_MM_LONGOFF || start2 - e2 < LZ_16BIT_OFFSET)) {     // :109 4 bytes (room >= 5 here), :112
                            { const u32 cbl = (hi >> 16) & 31u; b2 = lz_pf_back_from(cbl & 15u, cbl >= 16u, (start2 - ip) < e2 ? (start2 - ip) : e2); }   // :195-201
                            if (b2 == LZ_PF_UNRESOLVED) { LZ_STAT(40); slow = 2u; break; }
                            found = true;
                        }
                    }
                    commit |= 1ull << ls;                        // :190-191: the lane's own (conditional) put
                    LZ_STAT(41);
                    if (!found) break;
                    LZ_STAT(42);
                    ml2 = mlt + b2; ref2 = e2 - b2; start2 -= b2;                                  // :195-201
                    if (ml2 <= ml) { LZ_STAT(43); ml2 = 0; break; }                                 // :203
                    if (start2 <= ip) { LZ_STAT(44); ip = start2; ref = ref2; ml = ml2; ml2 = 0; break; }        // :205-210
                    if (start2 - ip < 3u) { LZ_STAT(45); ip = start2; ref = ref2; ml = ml2; ml2 = 0; continue; } // :212-217
                    if (start2 < ip + ml) {                                                         // :219-228
                        const u32 correction = ml - (start2 - ip);
                        LZ_STAT(46);
                        start2 += correction; ref2 += correction; ml2 -= correction;
                        if (ml2 < 3u) ml2 = 0;
                        if (ml2 < LZ_MM_LONGOFF && start2 - ref2 >= LZ_16BIT_OFFSET) ml2 = 0;
                    }
                    if (ml2) LZ_STAT(47);
                    break;
                }
                if (slow) break;
                {
                    const u32 off = ip - ref;                    // 0 = repeat (ref == ip), liz.h:122
                    lz_seq_push_liz(st, ip - anchor, ml, off);   // :231 (encoded later, in parallel)
                    if (off != 0u) last_off = off;               // liz.h:119,135
                }
                ip += ml; anchor = ip;
                LZ_STAT(48);                                     // a sequence pushed from registers
                if (!ml2) break;
                LZ_STAT(49);
                ip = start2; ref = ref2; ml = ml2; lazy = true;  // :233-238
            }
            if (slow) break;
            LZ_PROF(st, 2);                                      // winner lengths / arbitration / push from registers
            // ---- the next search starts at ip: the lanes from ip - ip0 on ----
            c = ip - ip0;
            if (c > 63u || !((validMask >> c) & 1ull)) break;    // beyond the round (or its width, or mflimit): the next round starts at ip
            if constexpr (TAB::kSweeps) if (ip >= st.sweepAt) break;
            if (last_off != measuredOff) {                       // their repeat-offset side again (:19-31): the 4-byte test decides
                LZ_STAT(50);
                const bool repCand = valid && lane >= c && last_off >= LZ_MIN_OFFSET && p >= last_off && p - last_off >= lowPos;
                const u64 rN = lz_ld64(src + (repCand ? p - last_off : S));
                rep = repCand && (u32)rN == first4;
                const bool needHash = !rep && hashCand && lane >= c && !hashKnown;
                if (lz_ballot(needHash)) {                       // (rare) the hash side of lanes that had a repeat match when the round was measured
                    bool r0, h0; u32 f0;
                    LZ_STAT(54);
                    lz_pf_measure(false, hashCand, bytes, pB, pC, 0, 0, 0, cA, cB, cC, have24, room, p - e, r0, h0, f0);
                    if (needHash) { hashKnown = true; hinfo = f0 | (cb << 16) | (1u << 21) | (1u << 22) | ((u32)h0 << 23); }
                }
                winfo = rep ? (LZ_PF_UNRESOLVED4 | (cb << 16) | (1u << 21)) : (hinfo & 0x1FFFFFu);
                okMask = lz_ballot(rep || (hinfo >> 22) == 3u);  // known and passing
                measuredOff = last_off;
                LZ_PROF(st, 12);                                 // the next stretch's repeat-offset side again
            }
        }
        // settle the table for exactly the lanes that happened
        if constexpr (TAB::kSpecPut) {
            // the last committed lane of every slot group stores the slot's final value; a group none of whose lanes happened is
            // restored by its first lane (whose eOld is the value from before the round)
            const u64 cm = grp & commit;
            const bool writer = valid && (cm ? (cm >> lane) == 1ull : (grp & lanesBelow) == 0);
            if (writer) table.set(h, cm ? TAB::make(tAfter, tcAfter) : TAB::make(eOld, ecOld));
        } else {
            if ((commit & laneBit) && (grp & commit & ~(lanesBelow | laneBit)) == 0) table.set(h, TAB::make(tAfter, tcAfter));
        }
        table.sync();
        LZ_PROF(st, 11);                                         // round: table put
        if (!slow) continue;

        // ---------------- the steps that need memory (then the round is over) ----------------
        LZ_PROF(st, 0);
        {
            if (slow == 2u) goto search;
            if (ml == LZ_PF_UNRESOLVED) ml = 24u + lz_count_fwd(src, P + 24u, M + 24u, matchlimit);   // (unresolved: 24 bytes were seen and agree)
            else if (ml == LZ_PF_UNRESOLVED4) ml = 4u + lz_count_fwd(src, P + 4u, M + 4u, matchlimit);
            ref = M;
            ip = P;
            if (ip - ref == last_off) { ref = ip; goto encode; }                          // :174
            if (back0 == LZ_PF_UNRESOLVED) back0 = lz_count_back(src, P, M, anchor);
            ip -= back0; ref -= back0; ml += back0;                                       // :176-182
        search:
            LZ_PROF(st, 13);                                     // memory-based steps: winner lengths
            ml2 = 0;
            if (ip + ml >= mflimit) goto encode;                                          // :185
            start2 = ip + ml - 2u;
            {
                const u32 i2 = start2 - winBase;                 // uniform: the 8 bytes at start2 out of the register window when it reaches
                u64 b2;
                if (i2 < 128u) {
                    const u32 lo = i2 < 64u ? lz_readlane((u32)wA, i2 & 63u) : lz_readlane((u32)wB, i2 & 63u);
                    const u32 hi = i2 < 64u ? lz_readlane((u32)(wA >> 32), i2 & 63u) : lz_readlane((u32)(wB >> 32), i2 & 63u);
                    b2 = (u64)lo | ((u64)hi << 32);
                } else b2 = lz_ld64(src + start2);
                const u32 h2 = lz_hash5_plain<HASHLOG>(b2);
                const u32 raw2 = table.get(h2, start2);
                const u32 c2 = TAB::chk(raw2), chk2 = TAB::chkOf((u32)b2);
                const u32 age2 = TAB::age(start2, TAB::pos(raw2)), e2 = start2 - age2;
                ml2 = 0; back2 = 0;
                if (age2 >= LZ_MIN_OFFSET && age2 <= (start2 > maxDist ? maxDist : start2) && c2 == chk2) {   // :106-110
                    u32 mlt;
                    lz_count_both(src, start2, e2, matchlimit, ip, mlt, back2);
                    if (mlt >= 4u && (mlt >= LZ_MM_LONGOFF || start2 - e2 < LZ_16BIT_OFFSET)) { ml2 = mlt; ref2 = e2; }   // :112
                }
                table.sync();
                if (lane == 0 && age2 - 1u >= LZ_MIN_OFFSET - 1u) table.set(h2, TAB::make(start2, chk2));   // :190-191
                table.sync();
            }
            LZ_PROF(st, 1);
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
            LZ_PROF(st, 13);
            {
                const u32 off = ip - ref;
                lz_seq_push_liz(st, ip - anchor, ml, off);
                if (off != 0u) last_off = off;
            }
            LZ_STAT(51);                                         // a sequence pushed by the memory-based steps
            LZ_PROF(st, 3);
            ip += ml; anchor = ip;
            if (ml2) { ip = start2; ref = ref2; ml = ml2; ml2 = 0; goto search; }         // :233-238
        }
    }
tail:
    if (st.nseq & (LZ_SEQ_RING - 1u)) lz_seq_flush(st);
    st.lastLits = E - anchor;                                    // liz.h:168-179
}
#else    // LZ_PF_CHAIN == 0: one sequence per round, lazy step from memory (rounds 2-3; A/B builds)
template <int HASHLOG, int TAGLOG, class TAB>
LZ_DEV void lz_parse_pricefast(const u8* src, u32 S, u32 E, const TAB& table, u8* tag, LzStreams& st)
{
    const u32 lane = lz_lane();
    const u64 laneBit = 1ull << lane;
    const u64 lanesBelow = laneBit - 1ull;
    const u32 maxDist = (1u << 22) - 1u;
    const u32 tagMask = (1u << TAGLOG) - 1u;
    u32 anchor = S;                                              // uniform
    u32 last_off = 0;                                            // uniform; Lizard_initBlock, lizard_compress.c:137
    if (E - S < LZ_MFLIMIT + 1u) { st.lastLits = E - S; return; }
    const u32 mflimit = E - LZ_MFLIMIT, matchlimit = E - LZ_LASTLITERALS;
    u32 ip = S + 1u;                                             // uniform, pricefast.h:155
    // Round width (see lz_parse_fast): rounds over a table in global memory start LZ_PF_W0 positions wide after a match and
    // double after every round without a winner — the positions behind a winner are 128-byte lines fetched for nothing.
    constexpr bool kNarrow = !TAB::kSpecPut && LZ_PF_W0 < 64u;
    u32 W = kNarrow ? LZ_PF_W0 : 64u;                            // uniform
    // Register window over the source: wA = the 8 bytes at winBase + lane, wB = at winBase + 64 + lane (positions from mflimit
    // on are never probed; their lanes hold the bytes at S).  A round probes winBase + lane, i.e. wA.
    u32 winBase = ip;                                            // uniform
    u64 wA, wB;
    { const u32 qa = ip + lane, qb = ip + 64u + lane; wA = lz_ld64(src + (qa < mflimit ? qa : S)); wB = lz_ld64(src + (qb < mflimit ? qb : S)); }
    for (;;) {
        // ---------------- search: 64 consecutive positions per round ----------------
        u32 P = 0, M = 0, ml = 0, back0 = 0;
        for (;;) {
            if (ip >= mflimit) goto tail;                        // pricefast.h:158
            if constexpr (TAB::kSweeps) if (ip >= st.sweepAt) { table.sync(); lz_pf_tab_sweep<HASHLOG>(table, ip); st.sweepAt = ip + LZ_PF_SWEEP_EVERY; table.sync(); }
            // move the window to ip: up to 64 positions further its near half is a lane shift of what is already here and only the
            // far half is requested (it is used a round from now: by the lazy step, or as the next round's near half)
            if (ip != winBase) {
                const u32 d = ip - winBase;                      // uniform
                const u32 qa = ip + lane, qb = ip + 64u + lane;
                if (d == 64u) wA = wB;
                else if (d < 64u) {
                    const u32 sl = (d + lane) & 63u;
                    const u32 al = lz_shfl((u32)wA, sl), ah = lz_shfl((u32)(wA >> 32), sl);
                    const u32 bl = lz_shfl((u32)wB, sl), bh = lz_shfl((u32)(wB >> 32), sl);
                    wA = d + lane < 64u ? ((u64)al | ((u64)ah << 32)) : ((u64)bl | ((u64)bh << 32));
                }
                else wA = lz_ld64(src + (qa < mflimit ? qa : S));
                wB = lz_ld64(src + (qb < mflimit ? qb : S));
                winBase = ip;
            }
            const u32 p = ip + lane;
            const bool valid = p < mflimit && (!kNarrow || lane < W);
            const u32 lowPos = p > maxDist ? p - maxDist : 0u;   // pricefast.h:11-13, per probe
            const u64 bytes = wA;
            const u32 first4 = (u32)bytes;
            const u32 h = lz_hash5_plain<HASHLOG>(bytes);
            const u32 myChk = TAB::chkOf(first4);
            u32 e, ec;                                           // pricefast.h:160,168 (old value; garbage when !valid); its check bits
            // (the lanes beyond a narrow round's width all read slot 0: one line instead of one each)
            { const u32 raw = table.get(kNarrow && !valid ? 0u : h, p); e = TAB::pos(raw); ec = TAB::chk(raw); }
            // Which lanes of this round share a table slot?  LDS tables: every lane stores the low half of its position
            // speculatively and reads the slot back — a lane that does not find its own value shares the slot with a later
            // lane (positions of a round differ by < 2^16); exact, no extra memory, settled after the winner is known.
            // Global tables: nothing is stored before the winner is known; lanes meet in a small LDS tag array (false
            // alarms only cost a turn of the replay loop).
            bool lost;
            if constexpr (TAB::kSpecPut) {
                lz_lds_sync();                                   // every lane has read before any lane stores
                if (valid) table.specPut(h, p);
                lz_lds_sync();
                lost = valid && table.specLost(h, p);
            } else {
                if (valid) tag[h & tagMask] = (u8)lane;
                lz_lds_sync();
                lost = valid && tag[h & tagMask] != (u8)lane;
                lz_lds_sync();                                   // tag reads done before the next round's writes
            }
            LZ_PROF(st, 8);                                      // (instrumented build) round: window, hash, table read
            const u32 eOld = e, ecOld = ec;
            u64 pend = lz_ballot(lost);
            u64 grp = laneBit;
            const bool putAlone = TAB::age(p, e) - 1u >= LZ_MIN_OFFSET - 1u;     // pricefast.h:170-171 when alone in the slot: put unless 1..7 back
            u32 tAfter = putAlone ? p : e, tcAfter = putAlone ? myChk : ec;
            while (pend) {                                       // same-slot lanes: replay the puts in lane order
                const u32 f = lz_ctz64(pend);
                const u32 hv = lz_readlane(h, f);
                const bool mine = valid && h == hv;
                const u64 g = lz_ballot(mine);
                u32 t = lz_readlane(e, f);                       // slot value before this round (uniform)
                u32 tc = lz_readlane(ec, f);                     // ... and its check bits
                for (u64 m = g; m; m &= m - 1ull) {
                    const u32 k = lz_ctz64(m), pk = ip + k;
                    const u32 ck = lz_readlane(myChk, k);
                    if (lane == k) { e = t; ec = tc; }
                    const bool put = TAB::age(pk, t) - 1u >= LZ_MIN_OFFSET - 1u;
                    t = put ? TAB::pos(TAB::make(pk, 0u)) : t; tc = put ? ck : tc;
                    if (lane == k) { tAfter = t; tcAfter = tc; }
                }
                if (mine) grp = g;
                pend &= ~g;
            }
            LZ_PROF(st, 9);                                      // round: same-slot replay
            // Lizard_FindMatchFast, pricefast.h:3-87.  Candidates: the repeat offset (:19; it wins and hides the hash candidate) and
            // the hash candidate (:63-65) if its check bits agree (else the 4-byte test of :67 fails).  One batch of loads serves
            // both: the repeat candidates of the 64 lanes are contiguous, the hash candidates are few.
            const bool repCand = valid && last_off >= LZ_MIN_OFFSET && p >= last_off && p - last_off >= lowPos;
            const u32 ageE = TAB::age(p, e);                     // e < p, e >= lowLimit, >= MIN_OFFSET back (:63-65) as tests on the age
            const bool hashCand = valid && ageE >= LZ_MIN_OFFSET && ageE <= (p > maxDist ? maxDist : p) && ec == myChk;
            e = p - ageE;                                        // the candidate's position in the block (meaningful for hashCand lanes)
            const bool have24 = p + 24u <= E;                    // third 8 bytes readable inside the sub-block (p + 16 <= E - 5 always)
            const u32 fc = have24 ? 16u : 0u;
            const u32 pp = valid ? p : S, rp = repCand ? p - last_off : S;
            const u64 rA = lz_ld64(src + rp), rB = lz_ld64(src + rp + 8u), rC = lz_ld64(src + rp + fc);
            const u64 pB = lz_ld64(src + pp + 8u), pC = lz_ld64(src + pp + fc);
            u64 cA = 0, cB = 0, cC = 0, cZ = 0, pZ = 0;
            const bool haveBack = hashCand && e >= 8u;           // then p >= 16 as well
            if (hashCand) {                                      // one batch, straight-line
                const u32 zb = haveBack ? 8u : 0u;
                cA = lz_ld64(src + e); cB = lz_ld64(src + e + 8u); cC = lz_ld64(src + e + fc);
                cZ = lz_ld64(src + (e - zb)); pZ = lz_ld64(src + (p - zb));
            }
            lz_converge();
            const bool rep = repCand && (u32)rA == first4;                                                  // :19-31
            u32 fwd = LZ_PF_UNRESOLVED, bwd = LZ_PF_UNRESOLVED;  // exact when the difference (or the limit) lies inside the fetched bytes
            bool hashOk;
            {
                const u64 x = bytes ^ (rep ? rA : cA), y = pB ^ (rep ? rB : cB), y2 = pC ^ (rep ? rC : cC), z = pZ ^ cZ;
                const u32 seen = have24 ? 24u : 16u;
                const u32 common = x ? lz_ctz64(x) >> 3 : y ? 8u + (lz_ctz64(y) >> 3) : (have24 && y2) ? 16u + (lz_ctz64(y2) >> 3) : seen;
                const u32 room = matchlimit - p;                 // p < matchlimit for every valid slot
                if (common < seen || room <= seen) fwd = common < room ? common : room;
                // :67 the 4-byte test, :69 a long offset needs ml >= minMatchLongOff (16 <= 24 fetched bytes: decided here)
                hashOk = !rep && hashCand && (u32)cA == first4
                      && (p - e < LZ_16BIT_OFFSET || (have24 && common >= LZ_MM_LONGOFF && room >= LZ_MM_LONGOFF));
                const u32 roomB = (p - anchor) < e ? (p - anchor) : e;                                      // :176-180
                const u32 cb = !haveBack ? 0u : z ? lz_clz64(z) >> 3 : 8u;
                if (roomB <= cb || (haveBack && cb < 8u)) bwd = cb < roomB ? cb : roomB;
            }
            lz_pin(fwd); lz_pin(bwd);                            // computed here, under this batch's counted wait
            const u64 okMask = lz_ballot(rep || hashOk);
            const u64 validMask = lz_ballot(valid);
            LZ_PROF(st, 10);                                     // round: candidate bytes, tests
            u32 w = 0;
            u64 commit = validMask;
            if (okMask) { w = lz_ctz64(okMask); commit = validMask & (~0ull >> (63u - w)); }
            if constexpr (TAB::kSpecPut) {
                // settle: the last committed lane of every slot group stores the slot's final value; a group none of whose
                // lanes happened (all after the winner) is restored by its first lane (whose `e` is still the old value)
                const u64 c = grp & commit;
                const bool writer = valid && (c ? (c >> lane) == 1ull : (grp & lanesBelow) == 0);
                if (writer) table.set(h, c ? TAB::make(tAfter, tcAfter) : TAB::make(eOld, ecOld));
            } else {
                if ((commit & laneBit) && (grp & commit & ~(lanesBelow | laneBit)) == 0) table.set(h, TAB::make(tAfter, tcAfter));
            }
            table.sync();
            LZ_PROF(st, 11);                                     // round: table put
            if (okMask) {
                P = lz_readlane(p, w);
                M = lz_readlane(rep ? p - last_off : e, w);
                ml = lz_readlane(fwd, w); back0 = lz_readlane(bwd, w);
                break;
            }
            ip += lz_popc64(validMask);                          // "ip++" for every probed position, :173
            if constexpr (kNarrow) W = W < 32u ? 2u * W : 64u;
        }
        if constexpr (kNarrow) W = LZ_PF_W0;
        // ---------------- winner: lengths, lazy re-search, sequence push ----------------
        LZ_PROF(st, 0);                                          // search rounds
        {
            if (ml == LZ_PF_UNRESOLVED) ml = 24u + lz_count_fwd(src, P + 24u, M + 24u, matchlimit);   // (unresolved: 24 bytes were seen and agree)
            u32 ml2 = 0, start2 = 0, ref2 = 0, ref = M, back2 = 0;
            ip = P;
            if (ip - ref == last_off) { ref = ip; goto encode; }                          // :174 -> repeat offset, no lazy step
            if (back0 == LZ_PF_UNRESOLVED) back0 = lz_count_back(src, P, M, anchor);
            ip -= back0; ref -= back0; ml += back0;                                       // :176-182
        search:
            LZ_PROF(st, 2);                                      // winner lengths / arbitration
            if (ip + ml >= mflimit) goto encode;                                          // :185
            start2 = ip + ml - 2u;
            {
                // the 8 bytes at start2: out of the register window when it reaches that far (start2 < mflimit - 2 here)
                const u32 i2 = start2 - winBase;                 // uniform
                u64 b2;
                if (i2 < 128u) {
                    const u32 lo = i2 < 64u ? lz_readlane((u32)wA, i2 & 63u) : lz_readlane((u32)wB, i2 & 63u);
                    const u32 hi = i2 < 64u ? lz_readlane((u32)(wA >> 32), i2 & 63u) : lz_readlane((u32)(wB >> 32), i2 & 63u);
                    b2 = (u64)lo | ((u64)hi << 32);
                } else b2 = lz_ld64(src + start2);
                const u32 h2 = lz_hash5_plain<HASHLOG>(b2);
                const u32 raw2 = table.get(h2, start2);
                const u32 c2 = TAB::chk(raw2), chk2 = TAB::chkOf((u32)b2);
                const u32 age2 = TAB::age(start2, TAB::pos(raw2)), e2 = start2 - age2;
                ml2 = 0; back2 = 0;
                if (age2 >= LZ_MIN_OFFSET && age2 <= (start2 > maxDist ? maxDist : start2) && c2 == chk2) {   // :106-110 (check bits differ: :109 fails)
                    u32 mlt;                                                      // 4-byte test, length and :195-201 in one round trip
                    lz_count_both(src, start2, e2, matchlimit, ip, mlt, back2);
                    if (mlt >= 4u && (mlt >= LZ_MM_LONGOFF || start2 - e2 < LZ_16BIT_OFFSET)) { ml2 = mlt; ref2 = e2; }   // :112
                }
                table.sync();
                if (lane == 0 && age2 - 1u >= LZ_MIN_OFFSET - 1u) table.set(h2, TAB::make(start2, chk2));   // :190-191
                table.sync();
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
            {
                const u32 off = ip - ref;                                                 // 0 = repeat (ref == ip), liz.h:122
                lz_seq_push_liz(st, ip - anchor, ml, off);                                // :231 (encoded later, in parallel)
                if (off != 0u) last_off = off;                                            // liz.h:119,135
            }
            LZ_PROF(st, 3);                                      // sequence push
            ip += ml; anchor = ip;
            if (ml2) { ip = start2; ref = ref2; ml = ml2; ml2 = 0; goto search; }         // :233-238
        }
    }
tail:
    if (st.nseq & (LZ_SEQ_RING - 1u)) lz_seq_flush(st);
    st.lastLits = E - anchor;                                    // liz.h:168-179
}
#endif   // LZ_PF_CHAIN
