// lz_block.h — device-side Lizard block compression for ONE wavefront per API block (gfx950).
//
// What is implemented here (bit-exact with reference inikep/lizard 1.0, 64-bit build, zero-initialised
// match-finder state — the oracle of SURVEY.md §0.2):
//   * fastSmall / fast greedy parser   (reference lib/lizard_parser_fastsmall.h:34-189,
//                                        lib/lizard_parser_fast.h:41-196)          levels 10/30, 11/31
//   * fastLZ4 token encoder            (reference lib/lizard_compress_lz4.h:3-86)
//   * sub-block container              (reference lib/lizard_compress.c:141-250, 472-547)
//   * lz_compress_block: one API block through any of the parsers (this file, lz_pricefast.h, lz_hashchain.h)
//     and, for levels >= 30, the huff0 stage (lz_huf.h)
//
// How the serial parse is mapped onto a 64-lane wave (this is a re-design, not a translation):
//   The reference walks one position at a time: hash, table get, table put, test candidate.  Between
//   two accepted matches the positions it will visit are known in advance (the skip schedule of
//   fast.h:75-82 restarts after every match), so ONE ROUND evaluates the next 64 visits at once:
//     lane l -> visit v0+l -> position p, 8 source bytes (loaded one round ahead), hash5, table slot.
//   Two visits of one round can hit the same table slot; the reference would have shown the later one the
//   earlier one's position.  With the table in LDS that order comes from the hardware: the get and the put of a
//   round are ONE pair of returning LDS atomics, and the lanes of one DS atomic that hit the same dword are served in
//   lane order (LzTab::xchg; the library checks that property when a device's context is created).  With the table in
//   global memory (LzTabWide: nothing is stored before the winner is known) such rounds are found through a small LDS
//   tag array; only then a short loop over the clashing hash values rebuilds, per lane, the mask of same-slot lanes,
//   from which the in-order predecessor (and later the in-order LAST writer) follow with clz/ctz on ballot masks.
//   Each lane then applies the reference's accept test to its candidate; the first accepting lane
//   (ctz of the ballot) is the match the reference would have taken; lanes up to and including it
//   commit their table puts, later lanes are discarded (the reference never reached them).
//   The parser only appends (literal run, match length, offset) to a sequence list; tokens and literals are
//   written afterwards by a wave-parallel pass (lz_encode_lz4), straight into dst when there is no Huffman stage.
//   The reference's post-match steps (fast.h:143-165: put(ip-2), probe of ip) are slots 0 and 1 of the next
//   run's first round: with anchor == ip the probe is exactly a search visit that cannot extend backwards.
//
// All cross-lane traffic goes through lz_wave.h.  Variables documented "uniform" hold the same value
// in every lane (they are derived from ballots/readlanes and live in SGPRs).
#pragma once
#include "lz_wave.h"
#include "lz_huf.h"

#define LZ_SUBBLOCK       (1u << 17)        // LIZARD_BLOCK_SIZE, reference lib/lizard_compress.h:122
#define LZ_SUBBLOCK_PAD   (LZ_SUBBLOCK + 32)
#define LZ_EMPTY          0xFFFFFFFFu       // table entry that fails "e < cur" (reference state: 0 < lowLimit)
#define LZ_MFLIMIT        20u               // MFLIMIT = WILDCOPYLENGTH + MINMATCH, lizard_common.h:77-79
#define LZ_LASTLITERALS   16u
#define LZ_MIN_OFFSET     8u                // LIZARD_FAST_MIN_OFFSET, lizard_parser_fast.h:1
#define LZ_MAX_DIST_LZ4   65535u            // (1 << windowLog) - 1, windowLog 16, lizard_common.h:223,237

// ---- per-wave stream staging (global-memory scratch; reference keeps these in Lizard_stream_t) ----
struct LzStreams {
    u8* lit;    u8* flags;  u8* off16;  u8* off24;      // scratch bases (each LZ_SUBBLOCK_PAD bytes)
    u32 nlit;   u32 nflags; u32 noff16; u32 noff24;     // uniform byte counts of the current sub-block
    // fast parser: the serial parse only appends (literal run, match length, offset) to a sequence list;
    // the streams are produced afterwards by a wave-parallel pass (lz_encode_lz4)
    u64* seq;   u32 nseq;                               // list base (scratch), uniform count
    u64* ring;                                          // LDS: LZ_SEQ_RING most recent sequences, flushed to `seq` in
                                                        // coalesced bursts — gfx950 counts stores in vmcnt, so a global
                                                        // store per sequence would sit in front of every later load wait
    u32 lastLits;                                       // uniform: trailing literals of the sub-block (fast.h:187-190)
    u32 sweepAt;                                        // uniform: position at which the table is swept next (tables with positions modulo a power of two)
#ifdef LZ_PROFILE
    u64 prof_last; u64 prof[16];                         // shader-clock deltas per phase (profile builds only)
#endif
};
// Phase profiling exists only in -DLZ_PROFILE builds of the library (lizard_amd/variants/prof), never in
// the shipped liblizard_amd.so: LZ_PROF(st, k) adds the shader clocks since the previous mark to slot k.
#ifdef LZ_PROFILE
#ifdef LZ_HPROF_FINE      // the Huffman stage's fine marks take slots 0..4 (lz_huf.h): the parsers' marks there are dropped
#define LZ_PROF(st, k) do { const u64 t_ = __builtin_readcyclecounter(); (st).prof[(k) < 5 ? 15 : (k)] += t_ - (st).prof_last; (st).prof_last = t_; (st).prof[15] = t_; } while (0)
#else
#define LZ_PROF(st, k) do { const u64 t_ = __builtin_readcyclecounter(); (st).prof[k] += t_ - (st).prof_last; (st).prof_last = t_; (st).prof[15] = t_; } while (0)
#endif
#else
#define LZ_PROF(st, k) ((void)0)
#endif
#define LZ_SCRATCH_BYTES (5u * LZ_SUBBLOCK_PAD)
// Scratch slot of one wave (global memory): the sequence list of the current sub-block (8 B per sequence), then a
// literals staging area and a flags staging area that only the Huffman levels use (levels without Huffman write
// every stream straight into dst).  fastLZ4 codewords: matches are >= 4 bytes, at most 131072/4 sequences (256 KiB);
// LIZv1: a trimmed second match may be 3 bytes long (pricefast.h:225), at most 131072/3 sequences (352 KiB).
#define LZ_SEQ_BYTES     (1u << 18)
#define LZ_SEQ_BYTES_LIZ (352u << 10)
#define LZ_SEQ_RING      32u                            // sequences buffered in LDS (256 B per wave)

LZ_DEV void lz_streams_bind(LzStreams& st, u8* scratch, bool lz4Codewords, u64* ring)
{
    st.ring = ring;
    st.seq = (u64*)scratch;
    st.lit = scratch + (lz4Codewords ? LZ_SEQ_BYTES : LZ_SEQ_BYTES_LIZ); st.flags = st.lit + LZ_SUBBLOCK_PAD;
    st.off16 = st.off24 = st.flags;                                                // offset streams are never staged
    st.nlit = st.nflags = st.noff16 = st.noff24 = 0;
    st.nseq = 0; st.lastLits = 0;
}

// 64-bit-build hash of the reference: hash5 over an 8-byte little-endian read
// (reference lib/lizard_compress.c:77,90-91; chosen by lizard_parser_fastsmall.h:4-9 / fast.h:7-12).
// ((u * prime5bytes) << 24) >> (64 - HASHLOG) is the top of u * (prime5bytes << 24): only the high word of that product is
// needed, written out as its three partial products so that the one over the first four bytes (LZ_HASH5_KHI below) can be
// shared with the tables' check bits (lz_check_mix) — 32-bit multiplies are quarter-rate instructions and the parse is
// issue-bound.  The fifth byte's term is an 8 x 8-bit product.
#define LZ_HASH5_K   (889523592379ULL << 24)
#define LZ_HASH5_KLO ((u32)LZ_HASH5_K)                 /* 0xBB000000 */
#define LZ_HASH5_KHI ((u32)(LZ_HASH5_K >> 32))
template <int HASHLOG>
LZ_DEV u32 lz_hash5(u64 u)
{
    const u32 lo = (u32)u, b4 = (u32)(u >> 32) & 0xFFu;
    const u32 hi = lz_mulhi(lo, LZ_HASH5_KLO) + lo * LZ_HASH5_KHI + (lz_opaque(lz_mul24(b4, LZ_HASH5_KLO >> 24)) << 24);
    return hi >> (32 - HASHLOG);
}
// the plain form (one 64-bit multiply = three 32-bit ones).  lz_pricefast.h keeps it: with the form above its kernels need 22
// more VGPRs (92 -> 114, 133 -> 160) and levels 21 / 41 lose 1.3 % / 4.5 % (profiles/r03r_*)
template <int HASHLOG>
LZ_DEV u32 lz_hash5_plain(u64 u) { return (u32)(((u * 889523592379ULL) << 24) >> (64 - HASHLOG)); }
// 32 mixed bits of the first four bytes at a position; the tables keep the top few as check bits beside the position.  Any
// function of those four bytes serves (equal bytes => equal bits, fast.h:97 is the test that counts); this one costs nothing,
// it is a partial product of the hash.
LZ_DEV u32 lz_check_mix(u32 first4) { return first4 * LZ_HASH5_KHI; }

// Offset of visit v from the run start and the step taken after it, closed form of
// "step = searchMatchNb++ >> Lizard_skipTrigger" (reference lizard_parser_fast.h:75-82):
// s_0 = 1, s_v = (63+v)>>6, f(v) = sum_{j<v} s_j.
LZ_DEV u32 lz_visit_off(u32 v)
{
    u32 q = (v - 1u) >> 6, t = (v - 1u) & 63u;                              // offsets stay below the block size: 24-bit factors
    return v == 0 ? 0u : 1u + lz_mul24(32u * q + t, q + 1u);
}
LZ_DEV u32 lz_visit_step(u32 v) { return v == 0 ? 1u : (63u + v) >> 6; }

// Common-prefix length of src[a..] and src[b..] (b < a) with a+i < limit — reference
// lib/lizard_common.h:475-490 (its 8/4/2/1-byte stepping is unobservable). 512 bytes per round: lane i
// compares the 8 bytes at offset 8i.  Reads up to 7 bytes past `limit`; every caller's limit is at
// least 16 bytes before the end of the block (matchlimit = E - 16).
LZ_DEV u32 lz_count_fwd(const u8* src, u32 a, u32 b, u32 limit)
{
    const u32 lane = lz_lane();
    u32 n = 0;                                              // uniform
    for (;;) {
        const u32 i = n + 8u * lane;
        u32 c = 0;                                          // equal bytes in my 8, clamped to the limit
        if (a + i < limit) {
            const u64 x = lz_ld64(src + (a + i)) ^ lz_ld64(src + (b + i));
            const u32 room = limit - (a + i);
            c = x ? lz_ctz64(x) >> 3 : 8u;
            c = c < room ? c : room;
        }
        const u64 stop = lz_ballot(c < 8u);
        if (stop) { const u32 f = lz_ctz64(stop); return n + 8u * f + lz_readlane(c, f); }
        n += 512u;
    }
}

// Backward extension: largest k with P-i >= anchor, M-i >= 0 and src[P-i] == src[M-i] for 1<=i<=k
// (reference lizard_parser_fast.h:102; lowPrefixPtr is the block start for independent blocks).
LZ_DEV u32 lz_count_back(const u8* src, u32 P, u32 M, u32 anchor)
{
    const u32 lane = lz_lane();
    u32 n = 0;                                              // uniform
    for (;;) {
        const u32 i = n + lane + 1u;
        const bool eq = (P >= anchor + i) && (M >= i) && (src[(P - i)] == src[(M - i)]);
        const u64 ne = lz_ballot(!eq);
        if (ne) return n + lz_ctz64(ne);
        n += 64;
    }
}

// Forward count from offset 0 and backward extension of one candidate pair (P > M, uniform) with their
// loads in flight together: one round trip instead of two whenever the match is shorter than 512 bytes and
// the extension shorter than 64.  fwd = lz_count_fwd(src, P, M, limit), back = lz_count_back(src, P, M, anchor).
LZ_DEV void lz_count_both(const u8* src, u32 P, u32 M, u32 limit, u32 anchor, u32& fwd, u32& back)
{
    const u32 lane = lz_lane();
    const u32 i = 8u * lane, j = lane + 1u;
    const bool inF = P + i < limit, inB = (P >= anchor + j) && (M >= j);
    const u64 x = lz_ld64(src + (inF ? P + i : P)) ^ lz_ld64(src + (inF ? M + i : M));
    const u32 bp = src[inB ? P - j : P], bm = src[inB ? M - j : P];
    u32 c = 0;
    if (inF) {
        const u32 room = limit - (P + i);
        c = x ? lz_ctz64(x) >> 3 : 8u;
        c = c < room ? c : room;
    }
    const u64 stop = lz_ballot(c < 8u);
    const u64 ne = lz_ballot(!(inB && bp == bm));
    if (stop) { const u32 f = lz_ctz64(stop); fwd = 8u * f + lz_readlane(c, f); }
    else fwd = 512u + lz_count_fwd(src, P + 512u, M + 512u, limit);
    if (ne) back = lz_ctz64(ne);
    else back = 64u + lz_count_back(src, P - 64u, M - 64u, anchor);
}

// Wave-wide byte copy (all lanes call; n uniform). dst/src need no alignment.
LZ_DEV void lz_copy(u8* dst, const u8* src, u32 n)
{
    const u32 lane = lz_lane();
    const u32 n4 = n & ~3u;
    for (u32 i = lane * 4u; i < n4; i += 256u) lz_st32_s(dst + i, lz_ld32_s(src + i));
    for (u32 i = n4 + lane; i < n; i += 64u) lz_st8_s(dst + i, lz_ld8_s(src + i));
}

// Length escape shared by all codewords (reference lib/lizard_compress_lz4.h:21-23), as a packed
// little-endian word + byte count: v<254 -> [v]; v<65536 -> [254, lo, hi]; else [255, b0, b1, b2].
LZ_DEV void lz_len_ext(bool present, u32 v, u32& word, u32& nbytes)
{
    if (!present)          { word = 0; nbytes = 0; }
    else if (v < 254u)     { word = v; nbytes = 1; }
    else if (v < 65536u)   { word = 254u | (v << 8); nbytes = 3; }
    else                   { word = 255u | (v << 8); nbytes = 4; }
}

// Bytes a fastLZ4 sequence adds to the literals stream (reference lib/lizard_compress_lz4.h:18-56):
// literal-length escape + literals + LE16 offset + match-length escape.
LZ_DEV u32 lz_ext_len(bool present, u32 v) { return !present ? 0u : v < 254u ? 1u : v < 65536u ? 3u : 4u; }
LZ_DEV u32 lz_lz4_record_bytes(u32 L, u32 mlc)
{
    return lz_ext_len(L >= 15u, L - 15u) + L + 2u + lz_ext_len(mlc >= 15u, mlc - 15u);
}

// Serial side of the encoder: one 8-byte LDS store per sequence, one coalesced global burst per
// LZ_SEQ_RING sequences.  L < 2^18, ml-4 < 2^18, offset < 2^16.
LZ_DEV void lz_seq_flush(LzStreams& st)
{
    const u32 pending = st.nseq & (LZ_SEQ_RING - 1u) ? st.nseq & (LZ_SEQ_RING - 1u) : (st.nseq ? LZ_SEQ_RING : 0u);
    lz_lds_sync();
    if (lz_lane() < pending) lz_stq_s(&st.seq[st.nseq - pending + lz_lane()], st.ring[lz_lane()]);
    lz_lds_sync();
}
// LEAN (the producers of lz_split.h): only the list entry; the stream sizes the container's rules need (nflags = one token per
// sequence, nlit = the records' bytes) are summed up by the CONSUMER, 64 sequences per step (lz_seq_sizes), instead of ~15 scalar
// instructions per sequence on the producer's serial chain.
template <bool LEAN = false>
LZ_DEV void lz_seq_push(LzStreams& st, u32 L, u32 ml, u32 off)
{
    const u32 mlc = ml - 4u;
    if (lz_lane() == 0) st.ring[st.nseq & (LZ_SEQ_RING - 1u)] = (u64)L | ((u64)mlc << 18) | ((u64)off << 36);
    lz_converge();
    st.nseq += 1u;
    if constexpr (!LEAN) { st.nflags += 1u; st.nlit += lz_lz4_record_bytes(L, mlc); }
    if ((st.nseq & (LZ_SEQ_RING - 1u)) == 0) lz_seq_flush(st);
}
// nflags / nlit of a finished fastLZ4 sequence list (what lz_seq_push keeps up to date when it is not LEAN): wave-parallel
LZ_DEV void lz_seq_sizes(LzStreams& st)
{
    u32 sum = 0;
    for (u32 base = 0; base < st.nseq; base += 256u) {        // four loads in flight: a step is one memory trip long
        u64 q[4];
        #pragma unroll
        for (u32 k = 0; k < 4u; k++) { const u32 i = base + 64u * k + lz_lane(); q[k] = lz_ldq_s(&st.seq[i < st.nseq ? i : st.nseq - 1u]); }
        #pragma unroll
        for (u32 k = 0; k < 4u; k++) {
            const u32 i = base + 64u * k + lz_lane();
            sum += i < st.nseq ? lz_lz4_record_bytes((u32)q[k] & 0x3FFFFu, (u32)(q[k] >> 18) & 0x3FFFFu) : 0u;
        }
    }
    st.nflags = st.nseq;
    st.nlit = st.lastLits + lz_wave_reduce_add(sum);
}

// Literal runs of one encode step (64 sequences, one per lane): run of lane j = L bytes from src + mySrc to
// litOut + litAt.  Vector-memory cost on this path is per instruction, not per byte, so one load/store pair serves
// EIGHT runs: lane l works on run 8g + (l >> 3) with sub-lane s = l & 7.  Runs of >= 8 bytes move 8 bytes per
// sub-lane (the last piece is pulled back to end exactly at the run's end; overlapping pieces rewrite identical
// bytes), shorter runs one byte per sub-lane.  All loads of the step are issued before the first store (loads and
// stores return through one in-order counter on gfx950).  Lanes without a sequence pass L = 0.
// PAIRS: long runs in flight per iteration of the loop at the end, two per pair: 4 in the consumers of lz_split.h (registers to spare),
// 2 in the one-wave kernels, which sit at the 128-VGPR limit of a 16-wave workgroup.
template <int PAIRS>
LZ_DEV void lz_copy_literal_runs(const u8* src, u8* litOut, u32 mySrc, u32 litAt, u32 L)
{
    const u32 lane = lz_lane();
    {
        u64 w8[8]; u32 w1[8];
        #pragma unroll
        for (u32 g = 0; g < 8u; g++) {
            const u32 jr = 8u * g + (lane >> 3), sl = lane & 7u;
            const u32 a = lz_shfl(mySrc, jr), nj = lz_shfl(L, jr);
            const u32 n64 = nj < 64u ? nj : 64u;              // bytes beyond 64 are copied below
            const u32 k = 8u * sl + 8u <= n64 ? 8u * sl : n64 - 8u;
            w8[g] = (n64 >= 8u && 8u * sl < n64) ? lz_ld64_s(src + a + k) : 0ull;
            w1[g] = (n64 < 8u && sl < n64) ? lz_ld8_s(src + a + sl) : 0u;
        }
        #pragma unroll
        for (u32 g = 0; g < 8u; g++) {
            const u32 jr = 8u * g + (lane >> 3), sl = lane & 7u;
            const u32 o = lz_shfl(litAt, jr), nj = lz_shfl(L, jr);
            const u32 n64 = nj < 64u ? nj : 64u;
            const u32 k = 8u * sl + 8u <= n64 ? 8u * sl : n64 - 8u;
            if (n64 >= 8u && 8u * sl < n64) lz_st64_s(litOut + o + k, w8[g]);
            if (n64 < 8u && sl < n64) lz_st8_s(litOut + o + sl, w1[g]);
        }
    }
    // runs longer than 64 bytes (on the benchmark data the mean run is 76 bytes, so these are common): the rest of them.
    // Eight runs per iteration, their loads issued side by side before the first store: lanes 0..31 work on one run, lanes 32..63
    // on the next, 16 bytes per lane (512 bytes of a run per pass), four such pairs.  One run at a time was a memory trip per long
    // run — on the benchmark data ~14 of a step's 64 runs carry 4/5 of its literal bytes — and the largest single item of the
    // consumers' encode pass (round 6).
    for (u64 longRuns = lz_ballot(L > 64u); longRuns; ) {
        const bool upper = lane >= 32u;
        u32 nj[PAIRS], a[PAIRS], o[PAIRS];                                 // per lane: the run of my half in pair q
        u32 most = 0;                                          // uniform: the longest of the (up to) 2 x PAIRS runs
        #pragma unroll
        for (u32 q = 0; q < (u32)PAIRS; q++) {
            const u32 j0 = longRuns ? lz_ctz64(longRuns) : 0u;
            const u32 n0 = longRuns ? lz_readlane(L, j0) : 0u, a0 = lz_readlane(mySrc, j0), o0 = lz_readlane(litAt, j0);
            longRuns &= longRuns - 1ull;                       // (0 stays 0)
            const u32 j1 = longRuns ? lz_ctz64(longRuns) : 0u;
            const u32 n1 = longRuns ? lz_readlane(L, j1) : 0u, a1 = lz_readlane(mySrc, j1), o1 = lz_readlane(litAt, j1);
            longRuns &= longRuns - 1ull;
            nj[q] = upper ? n1 : n0; a[q] = upper ? a1 : a0; o[q] = upper ? o1 : o0;
            most = n0 > most ? n0 : most; most = n1 > most ? n1 : most;
        }
        const u32 sl = 16u * (lane & 31u);
        for (u32 k0 = 64u; k0 < most; k0 += 512u) {            // uniform trip count
            const u32 k = k0 + sl;
            lz_u128 w[PAIRS];
            #pragma unroll
            for (u32 q = 0; q < (u32)PAIRS; q++) {
                const u32 kk = k + 16u <= nj[q] ? k : nj[q] - 16u;          // the last piece is pulled back to end at the run's end (runs here are > 64 bytes)
                if (k < nj[q]) w[q] = lz_ld128(src + a[q] + kk); else { w[q].lo = 0; w[q].hi = 0; }
            }
            #pragma unroll
            for (u32 q = 0; q < (u32)PAIRS; q++) {
                const u32 kk = k + 16u <= nj[q] ? k : nj[q] - 16u;
                if (k < nj[q]) lz_st128(litOut + o[q] + kk, w[q]);
            }
        }
    }
}

// Wave-parallel fastLZ4 encoder (reference lib/lizard_compress_lz4.h:3-86) over the sequence list of the
// sub-block starting at src+S: 64 sequences per step, one per lane.  Tokens go to flagsOut (one coalesced
// 64-byte store per step); record offsets inside litOut come from a wave prefix sum of the record sizes,
// literal source positions from a prefix sum of literals+match lengths; every lane writes its own
// escape/offset bytes, then the literal runs of the step are copied 8 sequences at a time (loads of 8
// runs in flight before the first store).  Trailing literals are appended raw.
template <int PAIRS>
LZ_DEV void lz_encode_lz4(const u8* src, u32 S, const LzStreams& st, u8* litOut, u8* flagsOut)
{
    const u32 lane = lz_lane();
    u32 srcPos = S, outPos = 0;                                   // uniform carries
    // the next step's sequences are requested one step ahead (clamped, unconditional): a step is a chain of scans and stores
    u64 qNext = st.nseq ? lz_ldq_s(&st.seq[lane < st.nseq ? lane : st.nseq - 1u]) : 0ull;
    for (u32 base = 0; base < st.nseq; base += 64u) {
        const u32 cnt = st.nseq - base < 64u ? st.nseq - base : 64u;
        u32 L = 0, mlc = 0, off = 0, R = 0, adv = 0;
        const u64 q = qNext;
        { const u32 nx = base + 64u + lane; qNext = lz_ldq_s(&st.seq[nx < st.nseq ? nx : st.nseq - 1u]); }
        if (lane < cnt) {
            L = (u32)q & 0x3FFFFu; mlc = (u32)(q >> 18) & 0x3FFFFu; off = (u32)(q >> 36);
            R = lz_lz4_record_bytes(L, mlc); adv = L + mlc + 4u;
        }
        const u32 myOut = outPos + lz_wave_scan_excl_add(R);
        const u32 mySrc = srcPos + lz_wave_scan_excl_add(adv);
        u32 extLw, extLn, extMw, extMn;
        lz_len_ext(L >= 15u, L - 15u, extLw, extLn);
        lz_len_ext(mlc >= 15u, mlc - 15u, extMw, extMn);
        if (lane < cnt) {
            lz_st8_s(flagsOut + base + lane, (L >= 15u ? 15u : L) | ((mlc >= 15u ? 15u : mlc) << 4));
            u8* r = litOut + myOut;
            for (u32 k = 0; k < extLn; k++) lz_st8_s(r + k, extLw >> (8u * k));
            r += extLn + L;
            lz_st16_s(r, off);
            for (u32 k = 0; k < extMn; k++) lz_st8_s(r + 2u + k, extMw >> (8u * k));
        }
        lz_copy_literal_runs<PAIRS>(src, litOut, mySrc, myOut + extLn, L);
        srcPos = lz_readlane(mySrc + adv, 63u);                   // lanes >= cnt hold adv == R == 0
        outPos = lz_readlane(myOut + R, 63u);
    }
    lz_copy(litOut + outPos, src + srcPos, st.lastLits);          // lizard_compress_lz4.h:74-86
}

// ---- fastSmall / fast parser over one sub-block [S,E) of the block at `src` ----------------------
// Hash table (ours; only the parse RESULT is pinned by the reference): 2^HASHLOG slots of 24 bits, kept
// as a u16 array + a u8 array so that a level-10 table is 12 KiB of LDS (13 waves per CU instead of 10
// with u32 slots — the parse is latency-bound and throughput scales with resident waves):
//   bits  0..16  position mod 2^17.  The reference only ever accepts candidates at distance <= 65535
//                (fast.h:90), so 17 bits are exact as long as no live slot gets older than 2^17-1: a sweep
//                at least every 2^15 positions re-stamps every slot older than 65535 as "exactly 65536
//                old" (dead for the reference from then on, and also the encoding of an empty slot).
//   bits 17..23  check hash of the 4 bytes at that position.  A candidate whose check bits differ from
//                the probing position's cannot pass the reference's 4-byte equality test (fast.h:97), so
//                its bytes are never fetched: most rounds issue no candidate gather at all.  Equal check
//                bits prove nothing; those lanes still load and compare the real bytes.
struct LzTab {
    LZ_LDS u16* lo; LZ_LDS u8* hi;
    static constexpr bool kSweeps = true;
    static constexpr u32  kSweepEvery = 32768u;                      // positions between two sweeps
    static constexpr bool kTagDedup = false;
    static constexpr bool kXchg = true;                              // get + put of a round in ONE LDS trip, in lane order (lz_lds_mskor_rtn2)
    // put `ent` into slot h and return the slot's previous value (as seen after the puts of all lower lanes of this instruction)
    LZ_DEVM u32 xchg(u32 h, u32 ent) const
    {
        const u32 sl = (h & 1u) * 16u, sh = (h & 3u) * 8u;
        u32 ol, oh;
        lz_lds_mskor_rtn2((LZ_LDS u32*)lo + (h >> 1), 0xFFFFu << sl, (ent & 0xFFFFu) << sl, (LZ_LDS u32*)hi + (h >> 2), 0xFFu << sh, ((ent >> 16) & 0xFFu) << sh, ol, oh);
        return ((ol >> sl) & 0xFFFFu) | (((oh >> sh) & 0xFFu) << 16);
    }
    // two slots per lane in one LDS trip: slot ha first (all lanes), then slot hb (all lanes)
    LZ_DEVM void xchg2(u32 ha, u32 ea, u32 hb, u32 eb, u32& oa, u32& ob) const
    {
        const u32 sla = (ha & 1u) * 16u, sha = (ha & 3u) * 8u, slb = (hb & 1u) * 16u, shb = (hb & 3u) * 8u;
        u32 al, ah, bl, bh;
        lz_lds_mskor_rtn4((LZ_LDS u32*)lo + (ha >> 1), 0xFFFFu << sla, (ea & 0xFFFFu) << sla, (LZ_LDS u32*)hi + (ha >> 2), 0xFFu << sha, ((ea >> 16) & 0xFFu) << sha,
                          (LZ_LDS u32*)lo + (hb >> 1), 0xFFFFu << slb, (eb & 0xFFFFu) << slb, (LZ_LDS u32*)hi + (hb >> 2), 0xFFu << shb, ((eb >> 16) & 0xFFu) << shb,
                          al, ah, bl, bh);
        oa = ((al >> sla) & 0xFFFFu) | (((ah >> sha) & 0xFFu) << 16);
        ob = ((bl >> slb) & 0xFFFFu) | (((bh >> shb) & 0xFFu) << 16);
    }
    LZ_DEVM u32  entry(u32 p, u32 first4) const { return (p & 0x1FFFFu) | (lz_check_mix(first4) >> 25 << 17); }
    LZ_DEVM u32  get(u32 h) const { return (u32)lo[h] | ((u32)hi[h] << 16); }
    LZ_DEVM u32  get(u32 h, u32) const { return get(h); }
    LZ_DEVM void set(u32 h, u32 ent) const { lo[h] = (u16)ent; hi[h] = (u8)(ent >> 16); }
    // age of a slot seen from position p (0..131071); usable iff 8 <= age <= 65535 (+ the lowLimit rule)
    LZ_DEVM u32  age(u32 p, u32 ent) const { return (p - ent) & 0x1FFFFu; }
    LZ_DEVM bool sameCheck(u32 a, u32 b) const { return ((a ^ b) >> 17) == 0; }
    // positions of one round differ by < 2^16, so the low halves alone tell the writers of a slot apart
    LZ_DEVM bool lostPut(u32 h, u32 mine) const { return lo[h] != (u16)mine; }
    LZ_DEVM void sync() const { lz_lds_sync(); }
};
// One extra slot (index 2^HASHLOG, "trash") lets lanes that must not store do so anyway, branch-free.
// (both arrays dword-aligned, with room for the trash slot's dword: the exchanges are dword atomics)
#define LZ_TAB_BYTES(HASHLOG) ((3u << (HASHLOG)) + 8u)
template <int HASHLOG> LZ_DEV LzTab lz_tab_bind(void* mem) { LzTab t; t.lo = (LZ_LDS u16*)mem; t.hi = (LZ_LDS u8*)mem + (2u << HASHLOG) + 4u; return t; }
// Re-stamp every slot that is dead at position Ps (age >= 65536); with `fresh`: all empty.
template <int HASHLOG>
LZ_DEV void lz_tab_sweep(const LzTab& t, u32 Ps, bool fresh)
{
    const u32 dead = (Ps + 65536u) & 0x1FFFFu;
    for (u32 i = lz_lane(); i < (1u << HASHLOG); i += 64u)
        if (fresh || t.age(Ps, t.get(i)) >= 65536u) t.set(i, dead);
}
template <int HASHLOG> LZ_DEV void lz_tab_fresh(const LzTab& t) { lz_tab_sweep<HASHLOG>(t, 0, true); }

// Wide variant for tables that do not fit LDS (hashLog 18: levels 11/31; 1 MiB per wave in global memory —
// 4 GiB over the resident waves, so every access is a random HBM sector): u32 slots, bits 0..21 = position mod 2^22,
// bits 22..31 = check hash.  Like LzTab's 17 bits, 22 bits are exact for any block size because the reference only
// accepts distances <= 65535 (fast.h:90): a sweep every 2^20 positions re-stamps every slot that is dead by then (age
// >= 65536) as "exactly 2^21 old", which is also what a fresh table holds — no slot ever gets older than 2^22 - 1.
// (A 256 KiB or 4 MiB block never sweeps.)  Cross-lane ordering needs the full wave sync.  Table accesses are what a round
// costs here, so same-slot lanes of a round are found through a small LDS tag array (as in lz_pricefast.h) instead of
// put + read-back, and nothing is stored speculatively: one gather and one scatter per round.
#ifndef LZ_WIDE_TAGLOG
#define LZ_WIDE_TAGLOG 11
#endif
struct LzTabWide {
    LZ_GLOBAL u32* w;
    u8* tag;                                                                     // LDS, tagMask + 1 bytes
    u32 tagMask = (1u << LZ_WIDE_TAGLOG) - 1u;
    // Occupancy summary (LDS, optional): bit (h >> occShift) is set once a slot of its group has been written in this block.
    // A probe whose bit is clear finds a dead slot without touching the table: with 2^18 slots and at most 2^17 insertions
    // per 256 KiB block most probes do — and a probe is a random 128-byte line.
    u32* occ = nullptr;
    u32 occShift = 0;
    static constexpr bool kSweeps = true;
    static constexpr u32  kSweepEvery = 1u << 20;
    static constexpr bool kTagDedup = true;
    static constexpr bool kXchg = false;
    LZ_DEVM u32 xchg(u32, u32) const { return 0; }
    LZ_DEVM static u32 dead(u32 p) { return (p - (1u << 21)) & 0x3FFFFFu; }      // a slot value that is 2^21 old at position p
    LZ_DEVM u32  entry(u32 p, u32 first4) const { return (p & 0x3FFFFFu) | (lz_check_mix(first4) >> 22 << 22); }
    LZ_DEVM u32  get(u32 h, u32 p) const
    {
        if (!occ) return w[h];
        const u32 b = h >> occShift;
        const bool oc = (occ[b >> 5] >> (b & 31u)) & 1u;
        const u32 v = w[oc ? h : 0u];                                            // unconditional load (slot 0 for the lanes that need none)
        return oc ? v : dead(p);
    }
    LZ_DEVM void set(u32 h, u32 ent) const
    {
        w[h] = ent;
        if (occ) { const u32 b = h >> occShift; lz_lds_atomic_or(&occ[b >> 5], 1u << (b & 31u)); }   // (the trash slot sets the spare bit)
    }
    LZ_DEVM u32  age(u32 p, u32 ent) const { return (p - ent) & 0x3FFFFFu; }     // dead slots: >= 65536
    LZ_DEVM bool sameCheck(u32 a, u32 b) const { return ((a ^ b) >> 22) == 0; }
    LZ_DEVM bool lostPut(u32 h, u32 mine) const { return w[h] != mine; }
    LZ_DEVM void sync() const { lz_table_sync(); }
};
#define LZ_TABWIDE_BYTES(HASHLOG) ((4u << (HASHLOG)) + 64u)
// Re-stamp every slot that is dead at position Ps (with `fresh`: every slot), 16 bytes per lane; the occupancy summary stays
// as it is (fresh: cleared) — a re-stamped slot is as good as a never-written one.
template <int HASHLOG>
LZ_DEV void lz_tab_sweep(const LzTabWide& t, u32 Ps, bool fresh)
{
    typedef u32 u32x4 __attribute__((vector_size(16)));
    const u32 d = LzTabWide::dead(Ps);
    for (u32 i = lz_lane() * 4u; i < (1u << HASHLOG) + 4u; i += 256u) {         // (aligned base; the 4 words behind the table are the trash slot's)
        u32x4 v = { d, d, d, d };
        if (!fresh) {
            const u32x4 o = *(LZ_GLOBAL u32x4*)(t.w + i);
            for (int k = 0; k < 4; k++) v[k] = t.age(Ps, o[k]) >= 65536u ? d : o[k];
        }
        *(LZ_GLOBAL u32x4*)(t.w + i) = v;
    }
    if (fresh && t.occ) for (u32 i = lz_lane(); i < (((1u << HASHLOG) >> t.occShift) >> 5) + 1u; i += 64u) t.occ[i] = 0u;
}
template <int HASHLOG> LZ_DEV void lz_tab_fresh(const LzTabWide& t) { lz_tab_sweep<HASHLOG>(t, 0, true); }

// Position handled by `slot` of the current run.  A run that follows a match ("special") spends its first
// two slots on the reference's post-match steps: slot 0 = put(ip-2) only (fast.h:146), slot 1 = the
// probe of ip itself (fast.h:149-165; identical to a search visit that cannot extend backwards because
// anchor == ip), slots >= 2 = visits 0.. of the search run starting at ip+1 (fast.h:184).
LZ_DEV void lz_slot_pos(u32 ip, u32 special, u32 slot, u32 mflimit, u32& p, bool& valid, bool& putOnly)
{
    const bool post = special && slot < 2u;                              // ip <= mflimit guaranteed (fast.h:143)
    const u32 v = slot - 2u * special;                                   // (garbage when post; selected away)
    const u32 pv = ip + special + lz_visit_off(v);
    p = post ? ip - 2u + 2u * slot : pv;
    valid = post || pv + lz_visit_step(v) <= mflimit;                    // fast.h:84, tested before the probe
    putOnly = post && slot == 0;
}

// Backward extension of a winner (fast.h:102) from the bytes its lane fetched: cb = equal bytes in the 8 before position and
// candidate (0 when the candidate starts less than 8 bytes into the block: nothing was fetched).  Exact when the run ends inside
// those bytes or at the limit (anchor, start of the block), else 0xFFFF = unresolved (lz_count_back finishes it).  Wave-uniform.
LZ_DEV u32 lz_back_from(u32 cb, u32 P, u32 M, u32 anchor)
{
    const u32 roomB = (P - anchor) < M ? (P - anchor) : M;
    return (roomB <= cb || (M >= 8u && cb < 8u)) ? (cb < roomB ? cb : roomB) : 0xFFFFu;
}

// Memory-latency structure of a round (the parse is latency-bound: ~70 % of wave time is s_waitcnt):
//   * the 8 source bytes of every lane are loaded ONE ROUND AHEAD (nextBytes), for the slots the run
//     will reach if the current round finds no match, and right after a match for the first round of
//     the next run, before the sequence is encoded — so encoding overlaps them;
//   * lanes whose candidate survives the check bits fetch, in one batch, everything the winner needs:
//     16 bytes forward at candidate and position (resolves match lengths < 16 without another trip)
//     and 8 bytes backward (resolves backward extensions < 8).
//
// Round width.  With a table in LDS a probe costs nothing and every round takes all 64 slots.  With a table in global memory
// (LzTabWide) every probed slot is a random 128-byte line, and the slots behind the round's winner are probed for nothing —
// the winner is lane 13 on average on the bench data, two thirds of a round's lines.  Those runs start LZ_WIDE_W0 slots wide
// and double the width after every round without a winner; which slots a round covers changes nothing in what is committed.
// (Level 11: 2.75e9 -> 1.63e9 line reads per 4 GiB at width 16, 38 -> 41 GB/s; 32 is the better start once rounds chain, below.)
#ifndef LZ_WIDE_W0
#define LZ_WIDE_W0 32u
#endif
//
// More than one sequence per round.  A round costs a table trip and, with a candidate, a memory trip into the 64 KiB behind the
// position (L2 holds 8 KiB per resident wave: those are fabric trips); the wave waits through both, and that — not the
// instruction count — is what a round costs (profiles/r03r_w_*: 60 vector instructions fewer per round changed nothing).  In the first round of a run the lanes behind the winner hold the consecutive positions P+1, P+2, ... and have
// already exchanged their entries and fetched their candidates' bytes.  When the winner's lengths are known from its batch, the
// reference's next steps are all in those lanes: put(ip-2), the probe of ip (fast.h:146-165) and the visits ip+1, ip+2, ... of the
// next run (step 1 for 64 visits) — provided none of them found in its slot the put of a lane in between, i.e. of a position
// inside the match, which the reference never inserts.  Positions written in this round are recognised by their age, so that test
// is one comparison per lane (global-memory tables store nothing before the round is settled: there a lane's entry came from
// another lane's registers, and the test is whether that lane lies inside the match).  If it holds up to the next accepting
// lane, the first sequence is pushed and the round goes on with that lane as its winner; the lanes inside the match are taken
// back like the lanes behind the last winner.  On the bench data 30 % of the sequences come out of a round that already
// produced one (2 387 rounds per 256 KiB block instead of 3 046): level 10 172 -> 199 GB/s, level 30 134 -> 151
// (profiles/r03y_*).  Measured and not kept (profiles/r03z_*): going on INSIDE the next run when no further lane accepts (the next
// round then starts in the schedule's step-2 stretch and can chain nothing: -2..4 %), and serving the next run's first bytes out
// of a 128-position register window instead of a load (-3 %: that load hits L1/L2, four lane shifts cost more).
#ifndef LZ_FAST_CHAIN
#define LZ_FAST_CHAIN 1
#endif
// (Measured in round 5 and not kept, profiles/r05j_m_*: 32 bytes forward in the candidate batch instead of 24 — two more loads per
//  candidate lane — is 5.3 % SLOWER at level 10 (191.9 vs 202.7 GB/s) although it halves the matches that need a second trip; 16
//  bytes forward is the same as 24 within 0.3 %.)
template <int HASHLOG, class TAB, bool LEAN = false>
LZ_DEV void lz_parse_fast(const u8* src, u32 S, u32 E, const TAB& table, LzStreams& st)
{
    constexpr bool kChain = (TAB::kXchg || TAB::kTagDedup) && LZ_FAST_CHAIN;   // several sequences out of one round (see below)
    constexpr bool kNarrow = TAB::kTagDedup && LZ_WIDE_W0 < 64u;
    constexpr u32  kW0 = kNarrow ? LZ_WIDE_W0 : 64u;
    const u32 lane = lz_lane();
    const u64 laneBit = 1ull << lane;
    const u64 lanesBelow = laneBit - 1ull;
    u32 anchor = S;                                                  // uniform
    if (E - S < LZ_MFLIMIT + 1u) { st.lastLits = E - S; st.nlit += E - S; return; }     // fast.h:63
    const u32 mflimit = E - LZ_MFLIMIT, matchlimit = E - LZ_LASTLITERALS;
    // fast.h:57-58 in block-relative positions: lowLimit is fixed at sub-block entry
    const u32 lowPos = S > LZ_MAX_DIST_LZ4 ? S - LZ_MAX_DIST_LZ4 : 0u;

    if constexpr (TAB::kSweeps) if (S >= st.sweepAt) { lz_tab_sweep<HASHLOG>(table, S, false); st.sweepAt = S + TAB::kSweepEvery; table.sync(); }
    if (lane == 0) { const u64 b0 = lz_ld64(src + S); table.set(lz_hash5<HASHLOG>(b0), table.entry(S, (u32)b0)); }   // fast.h:66
    table.sync();

    u32 ip = S + 1u;        // uniform: run start, or (special==1) the post-match probe position
    u32 special = 0;        // uniform
    // my slot of the coming round — position, flags and source bytes — is prepared one round ahead (for the
    // slots the run reaches if the current round finds no match) or right after a match (first round of the
    // next run): the visit-schedule arithmetic and the source load are off the round's critical path
    u32 pNext; bool validNext, putOnlyNext;
    u64 nextBytes;
    lz_slot_pos(ip, 0u, lane, mflimit, pNext, validNext, putOnlyNext);
    if constexpr (kNarrow) validNext = validNext && lane < kW0;
    nextBytes = lz_ld64(src + (validNext ? pNext : S));
    u32 W = kW0;            // uniform: slots of the round about to run
    for (;;) {
        // ---------------- search: rounds of 64 slots until a lane accepts ----------------
        u32 v0 = 0;         // uniform: slots consumed by earlier rounds of this run
        u32 P = 0, M = 0, ml = 0, back = 0;   // uniform: winner position, candidate, lengths
        for (;;) {
            LZ_PROF(st, 3);                                              // loop glue, sequence push
            const u32 p = pNext; const bool valid = validNext, putOnly = putOnlyNext;
            const u64 bytes = nextBytes;
            u32 pAhead;                                                  // my slot's position in the next round of this run
            {
                lz_slot_pos(ip, special, v0 + W + lane, mflimit, pAhead, validNext, putOnlyNext);
                if constexpr (kNarrow) validNext = validNext && lane < (W < 32u ? 2u * W : 64u);
                pNext = pAhead;
                if (!validNext) pAhead = S;                              // any readable address
            }
            if constexpr (TAB::kSweeps) {   // keep every live slot younger than 2^17 positions (see LzTab)
                const u32 p0 = lz_readlane(p, 0);
                if (p0 >= st.sweepAt) { lz_tab_sweep<HASHLOG>(table, p0, false); st.sweepAt = p0 + TAB::kSweepEvery; table.sync(); }
            }
            // (slots past mflimit hold stale bytes and a meaningless hash; `valid` keeps them out of every decision
            //  and their stores go to the trash slot — the round itself is branch-free up to the candidate batch)
            const u32 first4 = (u32)bytes;
            const u32 h = lz_hash5<HASHLOG>(bytes);
            const u32 mine = table.entry(p, first4);
            u32 e;                                                       // fast.h:86: the slot as this visit finds it
            u64 grp = laneBit;                                           // lanes of this round on my table slot (not needed with kXchg)
            u32 eOld = 0;
            u32 jPrev = 64u;                                             // kTagDedup: the lane my `e` came from (64 = the table)
            if constexpr (TAB::kXchg) {
                // fast.h:86-88 for all 64 visits in one LDS trip: every lane exchanges its entry into its slot; lanes that share a
                // slot are served in lane order, so each gets back what the reference's serial walk would have found there.
                // Visits after the winner never happened: they are taken back once the winner is known (below).
                e = table.xchg(valid ? h : (1u << HASHLOG), mine);
            } else {
            e = table.get(h, p);                                         // value before this round
            bool lost;
            if constexpr (TAB::kTagDedup) {                              // same-slot lanes found through LDS; nothing stored yet
                const u32 ti = h & table.tagMask;
                if (valid) table.tag[ti] = (u8)lane;
                lz_lds_sync();
                lost = valid && table.tag[ti] != (u8)lane;
                lz_lds_sync();                                           // reads done before the next round's writes
            } else {
                lz_converge();                                           // every lane has read before any lane puts
                table.set(valid ? h : (1u << HASHLOG), mine);            // speculative put (fast.h:88); undone below if needed
                table.sync();
                // two slots of this round on one table slot: the later must see the earlier's put, in order
                lost = valid && table.lostPut(h, mine);
            }
            u64 pend = lz_ballot(lost);                                  // uniform
            eOld = e;
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
                const u32 ej = lz_shfl(mine, j);
                if (prev) { e = ej; jPrev = j; }
            }
            }
            // accept test, fast.h:90-97 (check bits first: they decide whether any bytes are fetched)
            const u32 age = table.age(p, e);
            const u32 ep = p - age;
            const bool cand = valid && !putOnly && table.sameCheck(e, mine) && age >= LZ_MIN_OFFSET && age <= LZ_MAX_DIST_LZ4
                           && age <= p - lowPos;
            u64 cA = LZ_ANY64, cB = LZ_ANY64, pB = LZ_ANY64, cC = LZ_ANY64, pC = LZ_ANY64, cZ = LZ_ANY64, pZ = LZ_ANY64;   // (read by `cand` lanes only)
            const bool haveBack = cand && ep >= 8u;                      // then p >= 16 as well
            const bool have24 = p + 24u <= E;                            // third 8 bytes readable inside the sub-block
            if (cand) {                                                  // one batch, straight-line (p + 16 <= E - 5)
                const u32 zb = haveBack ? 8u : 0u, fc = have24 ? 16u : 0u;
                cA = lz_ld64(src + ep); cB = lz_ld64(src + ep + 8u); pB = lz_ld64(src + p + 8u);
                cC = lz_ld64(src + (ep + fc)); pC = lz_ld64(src + (p + fc));     // (32-bit offsets from one scalar base: no 64-bit address math)
                cZ = lz_ld64(src + (ep - zb)); pZ = lz_ld64(src + (p - zb));
            }
            // source bytes for the next round of this run (consumed only if no lane accepts).  Always
            // issued and assigned unconditionally: no select forces the load to complete inside this
            // round and the vmcnt arithmetic of the batch above stays exact.
            nextBytes = lz_ld64(src + pAhead);
            LZ_PROF(st, 0);                                              // round part A: bytes wait, hash, LDS, filter, loads issued
            // match lengths from the batch, already clamped like the reference's counts (fast.h:100,102):
            // exact when the difference (or the limit) lies inside the fetched bytes, else 0xFFFF = unresolved
            bool ok = false;
            u32 fwd = 0xFFFFu, cbk = 0u;
#ifdef LZ_SKIP_NOCAND
            if (lz_ballot(cand))                                         // most rounds have no candidate at all: no batch, nothing to measure
#endif
            {
                ok = cand && (u32)cA == first4;                          // fast.h:97
                const u64 x = bytes ^ cA, y = pB ^ cB, y2 = pC ^ cC, z = pZ ^ cZ;
                const u32 seen = have24 ? 24u : 16u;
                const u32 common = x ? lz_ctz64(x) >> 3 : y ? 8u + (lz_ctz64(y) >> 3) : (have24 && y2) ? 16u + (lz_ctz64(y2) >> 3) : seen;
                const u32 room = matchlimit - p;                         // p < matchlimit for every valid slot
                if (common < seen || room <= seen) fwd = common < room ? common : room;
                cbk = !haveBack ? 0u : z ? lz_clz64(z) >> 3 : 8u;        // (lz_back_from turns it into the winner's backward length)
            }
            lz_pin(fwd); lz_pin(cbk);                                    // computed here, under this batch's counted wait
            const u64 okMask = lz_ballot(ok);                            // uniform
            const u64 validMask = lz_ballot(valid);                      // uniform, a prefix of lanes
            u32 w = 0;
            u64 commit = validMask;
            u64 deadMask = 0;                                            // lanes inside the matches of chained sequences
            if (okMask) {
                w = lz_ctz64(okMask); commit = validMask & (~0ull >> (63u - w));
                if constexpr (kChain) {
                    while (v0 == 0) {                                    // first round of a run: consecutive positions behind the winner
                        const u32 Pw = lz_readlane(p, w), fw = lz_readlane(fwd, w);
                        if (fw == 0xFFFFu) break;
                        const u32 Mw = lz_readlane(ep, w);
                        const u32 bk = lz_back_from(lz_readlane(cbk, w), Pw, Mw, anchor);
                        if (bk == 0xFFFFu) break;
                        const u32 ipn = Pw + fw, l1 = w + fw;            // fast.h:141: ip behind the sequence, and its lane
                        if (ipn > mflimit || l1 > 63u) break;
                        const u64 from1 = ~0ull << l1;
                        const u64 ok2 = okMask & from1;
                        if (!ok2) break;
                        const u32 w2 = lz_ctz64(ok2);                    // the next accepting lane: probe of ip or a visit of the next run
                        const u64 put2 = 1ull << (l1 - 2u);              // put(ip-2), fast.h:146
                        const u64 dead2 = deadMask | ((from1 ^ (~0ull << (w + 1u))) & ~put2);
                        const u64 readers = put2 | (from1 & (~0ull >> (63u - w2)));
                        // my slot's entry was written by the lane `age` below me (consecutive positions): a put that never happened?
                        // (global tables: nothing was stored yet, the entry came from that lane's registers)
                        const bool stale = TAB::kXchg ? (age <= lane && ((dead2 >> (lane - age)) & 1ull)) : (jPrev < 64u && ((dead2 >> jPrev) & 1ull));
                        if (lz_ballot(stale) & readers) break;
                        lz_seq_push<LEAN>(st, Pw - bk - anchor, fw + bk, Pw - Mw);      // fast.h:138
                        anchor = ipn;
                        deadMask = dead2; commit |= readers; w = w2;
                    }
                }
            }
            // settle the table slots: slots after the winner never happened (the reference stopped there)
            if constexpr (TAB::kXchg) {
                // Of the undone lanes on one slot, the lowest one holds in `e` what the slot must go back to (the entry of the last
                // lane that did happen, or the value from before the round); it is the one whose `e` was not written by an undone
                // lane: entries of this round are told from older ones by their age (positions of a round ascend).
                const u32 pw = okMask ? lz_readlane(p, w) : 0u;
                const bool eUndone = (p > pw && age < p - pw) || (kChain && age <= lane && ((deadMask >> (lane - age)) & 1ull));
                const bool restore = okMask != 0 && valid && !(commit & laneBit) && !eUndone;
                table.set(restore ? h : (1u << HASHLOG), e);
            } else if constexpr (TAB::kTagDedup) {                       // last committed slot of every group stores, once
                const u64 c = grp & commit;
                table.set((valid && (c >> lane) == 1ull) ? h : (1u << HASHLOG), mine);
            } else {
                const u64 c = grp & commit;                              // committed slots on my table slot
                const bool single = grp == laneBit;
                const bool undo = single ? !(commit & laneBit)           // my own put did not happen
                                         : (c == 0 && (grp & lanesBelow) == 0);   // whole group undone: its first lane restores
                const bool redo = !single && c != 0 && (c >> lane) == 1ull;       // last committed slot of the group wins
                table.set((valid && (undo || redo)) ? h : (1u << HASHLOG), undo ? eOld : mine);
            }
            table.sync();
            LZ_PROF(st, 1);                                              // round part B: candidate wait, ballots, slot settle
            if (okMask) {
                P = lz_readlane(p, w); M = lz_readlane(ep, w);
                ml = lz_readlane(fwd, w); back = lz_back_from(lz_readlane(cbk, w), P, M, anchor);
                break;
            }
            if (validMask != (kNarrow ? ~0ull >> (64u - W) : ~0ull)) goto tail;      // ran into mflimit without a match
            v0 += W;
            if constexpr (kNarrow) W = W < 32u ? 2u * W : 64u;
        }
        // ---------------- extend ----------------
        if (ml == 0xFFFFu) ml = 4u + lz_count_fwd(src, P + 4u, M + 4u, matchlimit);            // fast.h:100
        if (back == 0xFFFFu) back = lz_count_back(src, P, M, anchor);                           // fast.h:102
        if constexpr (TAB::kSweeps) {
            // A match moves ip by up to a sub-block at once (128 KiB - 1), four sweep intervals of the 17-bit table, and the round's
            // own check below only sweeps once, at the position it finds itself at: a slot stamped "65536 old" by the last sweep would
            // by then be more than 2^17 old and read as young again.  Nothing is inserted inside a match, so the sweeps that fall due on
            // the way are made up here, at the positions they were due (never below the winner's: every entry in the table is older).
            const u32 ipn = P + ml;
            while (ipn >= st.sweepAt) {
                const u32 q = st.sweepAt > P ? st.sweepAt : P + 1u;
                lz_tab_sweep<HASHLOG>(table, q, false); st.sweepAt = q + TAB::kSweepEvery; table.sync();
            }
        }
        P -= back; M -= back; ml += back;
        ip = P + ml;
        LZ_PROF(st, 2);                                                  // extension
        // first round of the next run: its source bytes are requested before the sequence is pushed — right behind a match nothing
        // else hides that trip (+2 % at level 10, +2.7 % at level 30, profiles/r03f_*).  (Requesting them from inside the round,
        // as soon as the winner's forward length is known, is slower: 159.3 vs 171.6 GB/s, profiles/r03g_* — a conditional load of
        // a loop-carried value makes every round's wait a vmcnt(0).)
        special = 1u;
        lz_slot_pos(ip, 1u, lane, mflimit, pNext, validNext, putOnlyNext);
        if constexpr (kNarrow) { W = kW0; validNext = validNext && lane < kW0; }
        if (ip > mflimit) validNext = false;                             // (fast.h:143: there is no next run; any readable address)
        nextBytes = lz_ld64(src + (validNext ? pNext : S));
        lz_seq_push<LEAN>(st, P - anchor, ml, P - M);                    // fast.h:138 (encoded later, in parallel)
        anchor = ip;
        if (ip > mflimit) goto tail;                                     // fast.h:143
    }
tail:
    if (st.nseq & (LZ_SEQ_RING - 1u)) lz_seq_flush(st);
    st.lastLits = E - anchor; st.nlit += E - anchor;                     // fast.h:187-190
    LZ_PROF(st, 3);
}


// Rounds of 128 slots (two per lane; round 4, measured 14.5 % slower, not kept) live in variants/lz_fast128.h: -DLZ_FAST_128=1 builds only.
#ifndef LZ_FAST_128
#define LZ_FAST_128 0
#endif
#if LZ_FAST_128
#include "variants/lz_fast128.h"
#endif

LZ_DEV void lz_st24(u8* p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); p[2] = (u8)(v >> 16); }

#include "lz_pricefast.h"   // priceFast parser + LIZv1 encoder (uses the helpers above)
#include "lz_fastbig.h"     // fastBig parser (levels 20 / 40): the fast parser's rounds over a 4 MiB window, LIZv1 sequence list
// LDS pools of a workgroup (Huffman workspaces; chain-build regions of the hashChain levels).  A wave needs its LZ_HUF_WS_WORDS of LDS only while it entropy-codes a sub-block
// (about a third of its time at level 30), so the W waves of a workgroup share K < W workspaces and the LDS this frees
// holds more hash tables.  A wave holding a workspace never waits for anything else: no deadlock; waves that find the pool
// empty sleep and poll.  mask == nullptr: the wave owns `base` outright.
struct LzHufPool { u32* base; u32* mask; u32 count; u32 stride; };   // stride in words
LZ_DEV u32* lz_pool_acquire(const LzHufPool& pool, u32& slot)
{
    slot = 0;
    if (!pool.mask) return pool.base;
    for (;;) {
        lz_converge();
        const u32 freeBits = ~lz_lds_poll_u(pool.mask) & ((1u << pool.count) - 1u);
        if (freeBits) {
            const u32 bit = freeBits & (0u - freeBits);
            const u32 old = lz_readlane(lz_lds_atomic_or_rtn(pool.mask, lz_lane() == 0 ? bit : 0u), 0);   // branch-free claim, like lz_claim_index
            if (!(old & bit)) { slot = bit; lz_lds_sync(); return pool.base + (31u - (u32)__builtin_clz(bit)) * pool.stride; }
        } else lz_sleep();
    }
}
LZ_DEV void lz_pool_release(const LzHufPool& pool, u32 slot)
{
    if (!pool.mask) return;
    lz_lds_sync();                                               // my last workspace accesses are done
    lz_lds_atomic_and(pool.mask, lz_lane() == 0 ? ~slot : 0xFFFFFFFFu);
    lz_converge();
}

#include "lz_hashchain.h"   // hashChain parser (fastLZ4 codewords through the sequence list)

// ---- sub-block container (reference lib/lizard_compress.c:141-250) -------------------------------

// Lizard_writeBlock (reference lib/lizard_compress.c:186-250) over the sequence list of one sub-block.  The stream
// sizes are known from the parse, so the raw-fallback rule of :201 is decided before a single output byte exists;
// without Huffman also :228, and every stream is encoded straight into its final place in dst (order: len (always
// empty), off16, off24, flags, literals; each LE24 length + bytes).  With Huffman the two candidate streams
// (huffType = LITERALS + FLAGS, lizard_compress.c:374-377) are encoded into the staging areas first (the entropy
// stage needs them contiguous); the offset streams still go straight to dst.  LIZ: LIZv1 codewords (priceFast),
// else fastLZ4 codewords (whose off16/off24 streams are always empty).
template <bool HUF, bool LIZ, int PAIRS = 2>
LZ_DEV u32 lz_write_subblock_seq(const u8* src, u32 S, u32 E, u8* op, LzStreams& st, const LzHufPool& pool)
{
    const u32 n = E - S, sum = st.nflags + st.nlit + st.noff16 + st.noff24;
    bool raw = st.nlit < LZ_LASTLITERALS || sum + 16u > n;             // lizard_compress.c:201
    u32 total = 16u + sum;
    if (!HUF) raw = raw || total + total / 32u + 512u > n;             // :228 (sizes are final without Huffman)
    if (!raw) {
        lz_wave_sync();                                                // sequence list written by lane 0
        u8* const p16 = op + 4u;                                       // header byte, LE24 0 = empty `len` stream (:203-207)
        u8* const p24 = p16 + 3u + st.noff16;                          // :209
        u8* const pf = p24 + 3u + st.noff24;                           // :212
        if (lz_lane() == 0) { op[0] = 0; lz_st24(op + 1, 0); lz_st24(p16, st.noff16); lz_st24(p24, st.noff24); }
        lz_converge();
        if constexpr (!HUF) {
            u8* const pl = pf + 3u + st.nflags;
            if (lz_lane() == 0) { lz_st24(pf, st.nflags); lz_st24(pl, st.nlit); }        // :215, :221
            lz_converge();
            if constexpr (LIZ) lz_encode_lizv1<PAIRS>(src, S, st, pl + 3u, pf + 3u, p16 + 3u, p24 + 3u);
            else               lz_encode_lz4<PAIRS>(src, S, st, pl + 3u, pf + 3u);
        } else {
            if constexpr (LIZ) lz_encode_lizv1<PAIRS>(src, S, st, st.lit, st.flags, p16 + 3u, p24 + 3u);
            else               lz_encode_lz4<PAIRS>(src, S, st, st.lit, st.flags);
            lz_wave_sync();
            u32 hf = 0, hl = 0, slot;
            u8* q = pf;
            u32* const ws = lz_pool_acquire(pool, slot);                                // held for the entropy stage only
            q += lz_put_stream_huf(q, st.flags, st.nflags, ws, &hf LZ_HPROF_ARG(st));   // LIZARD_FLAG_FLAGS = 2
            q += lz_put_stream_huf(q, st.lit, st.nlit, ws, &hl LZ_HPROF_ARG(st));       // LIZARD_FLAG_LITERALS = 1
            lz_pool_release(pool, slot);
            total = (u32)(q - op);
            if (lz_lane() == 0) op[0] = (u8)(hl * 1u + hf * 2u);
            lz_converge();
            raw = total + total / 32u + 512u > n;                      // :228
        }
    }
    if (raw) {
        lz_wave_sync();
        if (lz_lane() == 0) { op[0] = 128; lz_st24(op + 1, n); }       // LIZARD_FLAG_UNCOMPRESSED, :239-244
        lz_converge();
        lz_copy(op + 4, src + S, n);
        return n + 4u;
    }
    return total;
}

// ---- one API block: reference Lizard_compress_extState on a zeroed state (lizard_compress.c:583) ----
// dst must have room for Lizard_compressBound(n) bytes. Returns the compressed size (uniform).
// seqRing:  LZ_SEQ_RING u64 of LDS.
// tabKind / tableMem — where and how this wave keeps its hash table:
//   LZ_TABKIND_LDS     fast parser: LZ_TAB_BYTES(HASHLOG) bytes of LDS (24-bit slots with check bits, LzTab);
//                      priceFast: 4 << HASHLOG bytes of LDS, u32 slots = 24-bit position + 8 check bits (LzTab32L, blocks < 16 MiB)
//   LZ_TABKIND_GLOBAL  u32 slots in global memory: fast parser LZ_TABWIDE_BYTES(HASHLOG) bytes, 16-byte aligned
//                      (LzTabWide, any block size; the only form for HASHLOG > 14); priceFast 4 << HASHLOG bytes (LzTab32G)
//   LZ_TABKIND_LDS18   priceFast only: LZ_TAB24C_BYTES(HASHLOG) bytes of LDS, 18-bit position + 6 check bits (LzTab24c, blocks <= 256 KiB)
// AUX:      priceFast -> TAGLOG of the round tag array (ws holds 2^TAGLOG bytes of LDS).
//           hashChain -> searchLength (4 or 5); tableMem = the wave's LZ_HC_SLOT_BYTES slot (global; nothing in it needs clearing).
// PARSER: 0 = fastSmall/fast + fastLZ4 codewords, 1 = priceFast + LIZv1 codewords, 2 = hashChain / noChain + fastLZ4 codewords,
//         3 = fastBig + LIZv1 codewords (AUX = TAGLOG; the table is the wave's global-memory slot of 4 << HASHLOG bytes, LzTab32G).
#define LZ_PARSER_FAST      0
#define LZ_PARSER_PRICEFAST 1
#define LZ_PARSER_HASHCHAIN 2
#define LZ_PARSER_FASTBIG   3
#define LZ_TABKIND_LDS      0u
#define LZ_TABKIND_GLOBAL   1u
#define LZ_TABKIND_LDS18    2u
template <int PARSER, int HASHLOG, int AUX, bool HUF>
LZ_DEV u32 lz_compress_block(const u8* src, u32 n, u8* dst, u32 level, void* tableMem, u8* ws, u8* scratch, u64* seqRing,
                             u32 tabKind = LZ_TABKIND_LDS, u32* hufPoolBase = nullptr, u32* hufPoolMask = nullptr, u32 hufPoolCount = 0,
                             const LzHufPool* hcPool = nullptr, u32 maxBlock = 0, u32* wideOcc = nullptr, u32 wideOccLog = 0, u32 wideTagLog = LZ_WIDE_TAGLOG)
{
    const u32 lane = lz_lane();
    LzStreams st;
    constexpr bool kLiz = PARSER == LZ_PARSER_PRICEFAST || PARSER == LZ_PARSER_FASTBIG;    // LIZv1 codewords
    lz_streams_bind(st, scratch, !kLiz, seqRing);
#ifdef LZ_PROFILE
    st.prof_last = __builtin_readcyclecounter();
    for (int k = 0; k < 16; k++) st.prof[k] = 0;
#endif
    // fast parser: 24-bit LDS slots up to hashLog 14, wide u32 slots in global memory above
    constexpr bool kWide = PARSER == LZ_PARSER_FAST && HASHLOG > 14;
    LzTab tab = lz_tab_bind<kWide ? 1 : HASHLOG>(tableMem);
    LzTabWide tabw; tabw.w = (LZ_GLOBAL u32*)tableMem; tabw.tag = ws; tabw.tagMask = (1u << wideTagLog) - 1u;
    if constexpr (PARSER == LZ_PARSER_FAST && HASHLOG > 14) { tabw.occ = wideOcc; tabw.occShift = wideOcc ? (u32)HASHLOG - wideOccLog : 0u; }
    // priceFast table forms (lz_pricefast.h): u32 slots in global memory / in LDS, or the packed 18 + 6 bit LDS form
    LzTab32G pf32g; pf32g.w = (LZ_GLOBAL u32*)tableMem;
    if constexpr (PARSER == LZ_PARSER_PRICEFAST && HASHLOG > 14) { pf32g.occ = wideOcc; pf32g.occShift = wideOcc ? (u32)HASHLOG - wideOccLog : 0u; }
    LzTab32L pf32l; pf32l.w = (LZ_LDS u32*)tableMem;
    LzTab24c pf24c; pf24c.lo = (LZ_LDS u16*)tableMem; pf24c.hi = (LZ_LDS u8*)tableMem + (2u << (kWide ? 1 : HASHLOG));
    LzHc hc;
    if constexpr (PARSER == LZ_PARSER_HASHCHAIN) {
        const u32 row = (level >= 30u ? level - 21u : level) - 13u;                  // lizard_common.h:240-244, :264-268
        const bool noChain = level == 12u || level == 32u || level == 33u;           // :239, :262-263: one candidate per search
        lz_hc_begin(hc, tableMem, maxBlock, noChain ? 1u : row == 4u ? 256u : 2u << row);
        hc.noChain = noChain;
        // AUX: the level's searchLength (4 or 5), or 6 = the kernels of the noChain levels (hash5, nochain.h:4), or 7 / 8 / 9 = the
        // kernels of levels 13 / 14 / 15 and twins (hash5, searchNum 2 / 4 / 8: lz_hcN_search)
        static_assert(AUX >= 4 && AUX <= 9, "hashChain: searchLength 4 / 5, 6 = noChain, 7 / 8 / 9 = searchNum 2 / 4 / 8");
        lz_hc_build<(AUX >= 6 ? 5 : AUX), HASHLOG>(src, n, hc, *hcPool, st);
        // first searches decided ahead of the parse at levels 16/17 / 37/38 (searchLength 4: searchNum 16 / 256, three times the
        // sequences of level 13); at 13-15 the plain hit pass is the faster one (profiles/r04y_*)
        constexpr bool kPre = LZ_HC_PREPASS && AUX == 4;
        hc.pre = kPre;
        if constexpr (kPre) lz_hc_hits(src, n, hc); else lz_hc_hits_plain(src, n, hc);
    }
    else if constexpr (PARSER == LZ_PARSER_FASTBIG) {            // with slot codes in LDS (wideOcc) a slot is only read after this block wrote it
        if (wideOcc) lz_fb_codes_fresh<HASHLOG>(wideOcc); else lz_pf_tab_fresh<HASHLOG>(pf32g);
        st.sweepAt = LZ_PF_SWEEP_EVERY;
    }
    else if constexpr (kWide) { lz_tab_fresh<HASHLOG>(tabw); st.sweepAt = LzTabWide::kSweepEvery; }
    else if constexpr (PARSER == LZ_PARSER_FAST) {
        if (tabKind == LZ_TABKIND_GLOBAL) { lz_tab_fresh<HASHLOG>(tabw); st.sweepAt = LzTabWide::kSweepEvery; }
        else { lz_tab_fresh<HASHLOG>(tab); st.sweepAt = LzTab::kSweepEvery; }
    }
    else if (tabKind == LZ_TABKIND_GLOBAL) { lz_pf_tab_fresh<HASHLOG>(pf32g); st.sweepAt = LZ_PF_SWEEP_EVERY; }
    else if (tabKind == LZ_TABKIND_LDS18)  lz_pf_tab_fresh<HASHLOG>(pf24c);
    else                                   { lz_pf_tab_fresh<HASHLOG>(pf32l); st.sweepAt = LZ_PF_SWEEP_EVERY; }
    lz_wave_sync();
    LZ_PROF(st, 6);                                           // table init
    if (lane == 0) dst[0] = (u8)level;                        // lizard_compress.c:488
    lz_converge();
    u32 op = 1u;                                              // uniform
    for (u32 pos = 0; pos < n; ) {                            // lizard_compress.c:494
        const u32 part = (n - pos) < LZ_SUBBLOCK ? (n - pos) : LZ_SUBBLOCK;
        st.nlit = st.nflags = st.noff16 = st.noff24 = 0;      // Lizard_initBlock, :130-138
        st.nseq = 0; st.lastLits = 0;
        if constexpr (PARSER == LZ_PARSER_HASHCHAIN) lz_parse_hashchain<(AUX == 6 ? 1 : AUX == 7 ? 2 : AUX == 8 ? 4 : AUX == 9 ? 8 : 0)>(src, n, pos, pos + part, hc, st);
        else if constexpr (PARSER == LZ_PARSER_FASTBIG) lz_parse_fastbig<HASHLOG, AUX>(src, pos, pos + part, pf32g, ws, wideOcc, st);
        else if constexpr (kWide)                    lz_parse_fast<HASHLOG>(src, pos, pos + part, tabw, st);
        else if constexpr (PARSER == LZ_PARSER_FAST) {
            if (tabKind == LZ_TABKIND_GLOBAL) lz_parse_fast<HASHLOG>(src, pos, pos + part, tabw, st);
#if LZ_FAST_128
            else if constexpr (HASHLOG <= 14) lz_parse_fast128<HASHLOG>(src, pos, pos + part, tab, st);
#endif
            else                              lz_parse_fast<HASHLOG>(src, pos, pos + part, tab, st);
        }
        else if (tabKind == LZ_TABKIND_GLOBAL) lz_parse_pricefast<HASHLOG, AUX>(src, pos, pos + part, pf32g, ws, st);
        else if (tabKind == LZ_TABKIND_LDS18)  lz_parse_pricefast<HASHLOG, AUX>(src, pos, pos + part, pf24c, ws, st);
        else                                   lz_parse_pricefast<HASHLOG, AUX>(src, pos, pos + part, pf32l, ws, st);
        if constexpr (kLiz) lz_seq_sizes_liz(st);             // the stream sizes the priceFast parse does not keep (lz_seq_push_liz)
        {   // Huffman workspace: the wave's own (it doubles as the parser's tag array), or one borrowed from the workgroup's pool
            LzHufPool pool; pool.base = hufPoolMask ? hufPoolBase : (u32*)ws; pool.mask = hufPoolMask; pool.count = hufPoolCount; pool.stride = LZ_HUF_WS_WORDS;
            op += lz_write_subblock_seq<HUF, kLiz>(src, pos, pos + part, dst + op, st, pool);
        }
        LZ_PROF(st, 5);                                       // container: encode pass / Huffman
        lz_wave_sync();                                       // scratch is reused by the next sub-block
        LZ_PROF(st, 4);                                       // draining the sub-block's stores
        pos += part;
    }
#ifdef LZ_PROFILE
    if (lane == 0) { u64* pr = (u64*)(scratch + LZ_SCRATCH_BYTES - 128u); for (int k = 0; k < 15; k++) pr[k] += st.prof[k]; }   // per-wave totals at the tail of its scratch slot
    lz_converge();
#endif
    return op;
}
