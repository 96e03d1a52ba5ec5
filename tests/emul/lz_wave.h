// tests/emul/lz_wave.h — TEST INFRASTRUCTURE ONLY: a 64-lane SIMT emulator for x86-64 hosts.
//
// It provides the same names as lizard_amd/csrc/lz_wave.h (the gfx950 one) so that the *unmodified*
// kernel bodies (lz_block.h, lz_huf.h, ...) can be compiled with g++ and executed lane-for-lane on a
// CPU in `pytest -m "not gpu"`, where there is no GPU.  It is never part of the product library and
// cannot be reached from it: the product includes lizard_amd/csrc/lz_wave.h, this file is only found
// through tests/emul's own -I path (see tests/emul/build.py).
//
// Model: each lane is a stackful coroutine (hand-rolled x86-64 context switch).  A lane runs until it
// reaches a cross-lane operation (ballot / readlane / shfl / wave_sync) and parks; when all 64 lanes
// have parked at the SAME operation the scheduler computes the results and resumes them.  Between
// two cross-lane operations the lanes run one after another in a pseudo-random order that changes at
// every phase, so code that silently depends on which lane wins a same-address store, or that reads
// another lane's store without an lz_wave_sync() in between, produces run-to-run differences and
// fails the parity tests.  Parking at different operations (divergent control flow around a
// cross-lane op) aborts.
#ifndef LZ_WAVE_H_
#define LZ_WAVE_H_   /* shared guard: the first lz_wave.h seen (gfx950 or test emulator) wins */
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

struct uint4 { uint32_t x, y, z, w; };
#define LZ_DEV static inline
#define LZ_DEVM inline
#define LZ_DEV_NOINLINE static __attribute__((noinline))
#define LZ_WAVE 64
#define LZ_LDS                 /* address-space qualifier of LDS pointers on the device; nothing here */
#define LZ_GLOBAL

namespace lzemu {

extern "C" void lzemu_ctx_switch(void** save_sp, void* load_sp);

enum Op { OP_NONE = 0, OP_BALLOT, OP_READLANE, OP_UNIFORM, OP_SHFL, OP_SYNC, OP_MSKOR2, OP_ATOM1, OP_DONE };

struct Wave {
    void*  lane_sp[LZ_WAVE];
    void*  sched_sp;
    u8*    stacks;
    int    cur;                 // lane currently running
    int    op[LZ_WAVE];         // operation each lane is parked at
    u64    arg0[LZ_WAVE];       // per-lane operands
    u64    arg1[LZ_WAVE];
    u64    res[LZ_WAVE];        // per-lane results
    u32*   xp[2][LZ_WAVE];      // OP_MSKOR2 operands: pointers, masks, values; results in xo
    u32    xm[2][LZ_WAVE], xv[2][LZ_WAVE], xo[2][LZ_WAVE];
    u32    rng;
    void (*entry)(void*);
    void*  entry_arg;
    u64    n_ops;
};

extern thread_local Wave* g_wave;

static inline void park(int op)
{
    Wave* w = g_wave;
    int me = w->cur;
    w->op[me] = op;
    lzemu_ctx_switch(&w->lane_sp[me], w->sched_sp);
}

void run_wave(void (*entry)(void*), void* arg, u32 seed);

}  // namespace lzemu

LZ_DEV u32 lz_lane() { return (u32)lzemu::g_wave->cur; }

LZ_DEV u64 lz_ballot(bool pred)
{
    lzemu::Wave* w = lzemu::g_wave; int me = w->cur;
    w->arg0[me] = pred ? 1 : 0;
    lzemu::park(lzemu::OP_BALLOT);
    return w->res[me];
}

LZ_DEV u32 lz_readlane(u32 v, u32 src)
{
    lzemu::Wave* w = lzemu::g_wave; int me = w->cur;
    w->arg0[me] = v; w->arg1[me] = src;
    lzemu::park(lzemu::OP_READLANE);
    return (u32)w->res[me];
}
LZ_DEV u64 lz_readlane64(u64 v, u32 src) { return (u64)lz_readlane((u32)v, src) | ((u64)lz_readlane((u32)(v >> 32), src) << 32); }

LZ_DEV u32 lz_uniform(u32 v)
{
    lzemu::Wave* w = lzemu::g_wave; int me = w->cur;
    w->arg0[me] = v;
    lzemu::park(lzemu::OP_UNIFORM);
    return (u32)w->res[me];
}

LZ_DEV u64 lz_uniform64(u64 v) { return (u64)lz_uniform((u32)v) | ((u64)lz_uniform((u32)(v >> 32)) << 32); }

// x and dst are wave-uniform; no other lane's state is needed to emulate v_writelane
LZ_DEV u32 lz_writelane(u32 v, u32 x, u32 dst) { return lz_lane() == dst ? x : v; }
LZ_DEV void lz_writelane2(u32& a, u32 xa, u32& b, u32 xb, u32 dst) { if (lz_lane() == dst) { a = xa; b = xb; } }

LZ_DEV u32 lz_shfl(u32 v, u32 srcLane)
{
    lzemu::Wave* w = lzemu::g_wave; int me = w->cur;
    w->arg0[me] = v; w->arg1[me] = srcLane & 63u;
    lzemu::park(lzemu::OP_SHFL);
    return (u32)w->res[me];
}

LZ_DEV void lz_wave_sync() { lzemu::park(lzemu::OP_SYNC); }

LZ_DEV void lz_lds_sync() { lzemu::park(lzemu::OP_SYNC); }
LZ_DEV void lz_table_sync() { lzemu::park(lzemu::OP_SYNC); }
LZ_DEV void lz_pin(u32& x) { (void)x; }
// event counters (coverage of the parsers' paths under test): LZ_STAT(i) counts once per wave; read with emul_stats()
extern "C" unsigned long long lzemu_stats[64];
#define LZ_STAT(i) do { if (lz_lane() == 0) __atomic_add_fetch(&lzemu_stats[i], 1ull, __ATOMIC_RELAXED); } while (0)
LZ_DEV u32 lz_mul24(u32 a, u32 b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
LZ_DEV u32 lz_opaque(u32 x) { return x; }
LZ_DEV u32 lz_mulhi(u32 a, u32 b) { return (u32)(((u64)a * b) >> 32); }
#define LZ_ANY64 ((u64)0)
LZ_DEV void lz_converge() { lzemu::park(lzemu::OP_SYNC); }   // all lanes must arrive together

LZ_DEV u32 lz_ctz64(u64 m) { return (u32)__builtin_ctzll(m); }
LZ_DEV u32 lz_clz64(u64 m) { return (u32)__builtin_clzll(m); }
LZ_DEV u32 lz_popc64(u64 m) { return (u32)__builtin_popcountll(m); }

// Words shared between waves: several emulated waves run on OS threads of their own (emul_compress_split), so these are real atomics
// and a sleeping wave gives its core away.
LZ_DEV void lz_lds_atomic_add(u32* p, u32 v) { __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
LZ_DEV void lz_lds_atomic_or(u32* p, u32 v) { __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
LZ_DEV u32 lz_lds_atomic_or_rtn(u32* p, u32 v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
LZ_DEV void lz_lds_atomic_and(u32* p, u32 v) { __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
LZ_DEV u32 lz_lds_poll(const u32* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
LZ_DEV void lz_sleep() { sched_yield(); }
// one read per wave (another wave's thread may change the word between two lanes' turns)
LZ_DEV u32 lz_lds_poll_u(const u32* p) { u32 v = 0; if (lz_lane() == 0) v = lz_lds_poll(p); return lz_readlane(v, 0); }
LZ_DEV u32 lz_lds_claim(u32* counter)
{
    u32 old = 0;
    if (lz_lane() == 0) old = __atomic_fetch_add(counter, 1u, __ATOMIC_SEQ_CST);
    return lz_readlane(old, 0);
}
LZ_DEV u32 lz_claim_index(u32* counter) { return lz_lds_claim(counter); }
LZ_DEV void lz_lds_store(u32* p, u32 v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
LZ_DEV void lz_publish_release() { lzemu::park(lzemu::OP_SYNC); __atomic_thread_fence(__ATOMIC_SEQ_CST); }   // every lane's stores are done
LZ_DEV void lz_publish_acquire() { __atomic_thread_fence(__ATOMIC_SEQ_CST); lzemu::park(lzemu::OP_SYNC); }
LZ_DEV u32 lz_ld_shared_u32(const u32* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
LZ_DEV void lz_st_shared_u32(u32* p, u32 v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
// returning masked exchanges: the hardware serves the lanes of one instruction in ascending lane order — the scheduler does too
LZ_DEV void lz_lds_mskor_rtn2(u32* pa, u32 ma, u32 va, u32* pb, u32 mb, u32 vb, u32& oa, u32& ob)
{
    lzemu::Wave* w = lzemu::g_wave; int me = w->cur;
    w->xp[0][me] = pa; w->xm[0][me] = ma; w->xv[0][me] = va;
    w->xp[1][me] = pb; w->xm[1][me] = mb; w->xv[1][me] = vb;
    lzemu::park(lzemu::OP_MSKOR2);
    oa = w->xo[0][me]; ob = w->xo[1][me];
}
// two slots per lane: the a-pair of every lane, then the b-pair of every lane (the LDS unit executes the instructions in order)
LZ_DEV void lz_lds_mskor_rtn4(u32* pa, u32 ma, u32 va, u32* pb, u32 mb, u32 vb, u32* pc, u32 mc, u32 vc, u32* pd, u32 md, u32 vd, u32& oa, u32& ob, u32& oc, u32& od)
{
    lz_lds_mskor_rtn2(pa, ma, va, pb, mb, vb, oa, ob);
    lz_lds_mskor_rtn2(pc, mc, vc, pd, md, vd, oc, od);
}
LZ_DEV void lz_lds_mskor(u32* p, u32 mask, u32 val) { *p = (*p & ~mask) | val; }
// returning exchange (kind 0) / add (kind 1), all lanes take part, served in ascending lane order
LZ_DEV u32 lz_lds_atom1(u32* p, u32 v, u32 kind)
{
    lzemu::Wave* w = lzemu::g_wave; int me = w->cur;
    w->xp[0][me] = p; w->xm[0][me] = kind; w->xv[0][me] = v;
    lzemu::park(lzemu::OP_ATOM1);
    return w->xo[0][me];
}
LZ_DEV u32 lz_lds_xchg_rtn(u32* p, u32 v) { return lz_lds_atom1(p, v, 0u); }
LZ_DEV u32 lz_lds_add_rtn(u32* p, u32 v) { return lz_lds_atom1(p, v, 1u); }


LZ_DEV u32 lz_wave_reduce_add(u32 v)
{
    for (u32 d = 32; d > 0; d >>= 1) v += lz_shfl(v, lz_lane() ^ d);
    return lz_uniform(v);
}
LZ_DEV u32 lz_wave_reduce_max(u32 v)
{
    for (u32 d = 32; d > 0; d >>= 1) { const u32 o = lz_shfl(v, lz_lane() ^ d); v = o > v ? o : v; }
    return lz_uniform(v);
}
LZ_DEV u32 lz_wave_scan_excl_add(u32 v)
{
    const u32 lane = lz_lane();
    u32 incl = v;
    for (u32 d = 1; d < 64; d <<= 1) { const u32 o = lz_shfl(incl, lane - d); if (lane >= d) incl += o; }
    return incl - v;
}

LZ_DEV u32 lz_ld32(const u8* p) { u32 v; memcpy(&v, p, 4); return v; }
LZ_DEV u64 lz_ld64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
LZ_DEV void lz_st16(u8* p, u32 v) { u16 x = (u16)v; memcpy(p, &x, 2); }
LZ_DEV void lz_st32(u8* p, u32 v) { memcpy(p, &v, 4); }
LZ_DEV void lz_st64(u8* p, u64 v) { memcpy(p, &v, 8); }
struct lz_u128 { u64 lo, hi; };
LZ_DEV lz_u128 lz_ld128(const u8* p) { lz_u128 v; memcpy(&v, p, 16); return v; }
LZ_DEV void lz_st128(u8* p, lz_u128 v) { memcpy(p, &v, 16); }
LZ_DEV u64 lz_ld64_s(const u8* p) { return lz_ld64(p); }
LZ_DEV u32 lz_ld32_s(const u8* p) { return lz_ld32(p); }
LZ_DEV u8  lz_ld8_s(const u8* p) { return *p; }
LZ_DEV void lz_st64_s(u8* p, u64 v) { lz_st64(p, v); }
LZ_DEV void lz_st32_s(u8* p, u32 v) { lz_st32(p, v); }
LZ_DEV void lz_st16_s(u8* p, u32 v) { lz_st16(p, v); }
LZ_DEV void lz_st8_s(u8* p, u32 v) { *p = (u8)v; }
LZ_DEV u64 lz_ldq_s(const u64* p) { return *p; }
LZ_DEV void lz_stq_s(u64* p, u64 v) { *p = v; }
#endif  /* LZ_WAVE_H_ */
