// lz_wave.h — gfx950 wavefront primitives used by the Lizard block kernels.
//
// One Lizard API block is parsed by ONE 64-lane wavefront (the LZ77 parse is a serial dependency
// chain; parallelism inside a block is the 64 speculative probes of one search round, parallelism
// across blocks is thousands of resident waves).  Everything the kernels need from the machine is
// named here so the kernel bodies (lz_block.h, lz_huf.h) read as algorithm, and so the test-only SIMT
// emulator (tests/emul/lz_wave.h) can supply the same names on a CPU to run the very same kernel
// bodies under pytest -m "not gpu".  This file is the gfx950 one: it is what hipcc sees.
#ifndef LZ_WAVE_H_
#define LZ_WAVE_H_   /* shared guard: the first lz_wave.h seen (gfx950 or test emulator) wins */
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define LZ_DEV __device__ __forceinline__
#define LZ_DEVM __device__ __forceinline__       /* member functions */
#define LZ_DEV_NOINLINE __device__ __noinline__
#define LZ_WAVE 64
// A pointer into LDS, typed as such: accesses through it are DS instructions with a 32-bit address.  A table pointer that is
// picked at run time ("this wave's table is in LDS, that one's in global memory") is a generic pointer to the compiler, and
// accesses through a generic pointer are FLAT instructions — 64-bit addresses in VGPR pairs, and counted in vmcnt AND lgkmcnt, so
// that every wait for one of them also waits for all global loads in flight.  The table forms that live in LDS cast once.
#define LZ_LDS __attribute__((address_space(3)))
#define LZ_GLOBAL __attribute__((address_space(1)))            /* likewise for a table in the wave's global-memory slot: global_*, not flat_* */

// lane index inside the wavefront (workgroups are launched with blockDim.x a multiple of 64)
LZ_DEV u32 lz_lane() { return threadIdx.x & 63u; }

// 64-bit lane mask of `pred` over the active lanes (wave-uniform result, lives in an SGPR pair)
LZ_DEV u64 lz_ballot(bool pred) { return __ballot(pred); }

// value of `v` in lane `src` where `src` is wave-uniform: v_readlane_b32, result is scalar
LZ_DEV u32 lz_readlane(u32 v, u32 src) { return (u32)__builtin_amdgcn_readlane((int)v, (int)src); }
LZ_DEV u64 lz_readlane64(u64 v, u32 src) { return (u64)lz_readlane((u32)v, src) | ((u64)lz_readlane((u32)(v >> 32), src) << 32); }

// value held by the first active lane, as a scalar.  Used to pin wave-uniform state into SGPRs.
LZ_DEV u32 lz_uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
LZ_DEV u64 lz_uniform64(u64 v) { return (u64)lz_uniform((u32)v) | ((u64)lz_uniform((u32)(v >> 32)) << 32); }

// `v` with lane `dst` (wave-uniform) replaced by the wave-uniform value x.  Together with lz_readlane this makes a VGPR
// a 64-entry table that wave-uniform (scalar) code reads and writes without touching memory.  (Compare + select: two
// VALU ops, like s_mov m0 + v_writelane_b32, which needs M0 for its second scalar operand on gfx9.)
LZ_DEV u32 lz_writelane(u32 v, u32 x, u32 dst) { return lz_lane() == dst ? x : v; }

// Two register tables updated at one (wave-uniform) lane with wave-uniform values: s_mov m0 + two v_writelane_b32 (M0 as the lane
// select is the one form in which gfx9 lets v_writelane take a second scalar operand) instead of two compare-and-select pairs.
LZ_DEV void lz_writelane2(u32& a, u32 xa, u32& b, u32 xb, u32 dst)
{
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                 : "+v"(a), "+v"(b) : "s"(xa), "s"(xb), "s"(dst) : "m0");
}

// arbitrary cross-lane gather (ds_bpermute_b32): every lane names its own source lane
LZ_DEV u32 lz_shfl(u32 v, u32 srcLane) { return (u32)__builtin_amdgcn_ds_bpermute((int)(srcLane << 2), (int)v); }

// Ordering point for LDS / global traffic between the lanes of ONE wave: all earlier stores of every
// lane are visible to later loads of every lane.  Lanes of a wave execute DS/VMEM instructions in
// program order, so no s_barrier is needed; the fence stops the compiler from forwarding a lane's
// own store to its later load (or hoisting loads) across the point and drains lgkmcnt/vmcnt.
LZ_DEV void lz_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Re-convergence point. LLVM may jump-thread two `if (lane == 0)` regions into separate per-lane paths
// (seen on ROCm 7.2: the claim loop below split lane 0 from lanes 1..63 around its back-edge, so 63
// lanes spun forever); a convergent, side-effecting, non-duplicable marker after every single-lane
// region and at the top of every wave-uniform loop forces the paths to re-join first. No ISA emitted.
LZ_DEV void lz_converge() { __builtin_amdgcn_wave_barrier(); }

// Claim the next work item from a device-wide counter: ONE atomic per wave, result wave-uniform.
// Deliberately branch-free (every lane takes part, only lane 0 adds 1): no single-lane region sits
// next to the loop back-edge of the caller.
LZ_DEV u32 lz_claim_index(u32* counter)
{
    const u32 old = atomicAdd(counter, lz_lane() == 0 ? 1u : 0u);
    return lz_readlane(old, 0);
}

// Ordering point for LDS traffic only, between the lanes of ONE wave.  DS instructions of a wave execute
// in issue order, so the hardware needs nothing; the wavefront-scope fences only stop the compiler from
// reordering LDS accesses across the point or forwarding a lane's own store to its later load.  Unlike
// lz_wave_sync() this does not drain vmcnt, so pending global stores (stream output) stay in flight.
LZ_DEV void lz_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Ordering point between the lanes of ONE wave for a table in GLOBAL memory (lane A's put, lane B's later get of the same slot).
// Default: the draining form (lz_wave_sync: vmcnt(0) behind the puts).  The AMDGPU memory model orders the operations of one
// wavefront without it (LLVM AMDGPUUsage, gfx90a/gfx942: "fence acq_rel, wavefront: none"), and -DLZ_TABLE_SYNC_DRAIN=0 builds that
// form — measured on levels 21/41/22/42/11/31 it changes nothing (46.09 vs 46.05, 29.3 vs 30.1, 37.4 vs 38.0 GB/s,
// profiles/r03d_*: the puts are acknowledged quickly and the other waves of the SIMD fill the wait), so the conservative form stays.
#ifndef LZ_TABLE_SYNC_DRAIN
#define LZ_TABLE_SYNC_DRAIN 1
#endif
LZ_DEV void lz_table_sync()
{
#if LZ_TABLE_SYNC_DRAIN
    lz_wave_sync();
#else
    lz_lds_sync();
#endif
}

// Pin a per-lane value: it must be computed HERE (the optimiser may not sink its computation into a
// later conditional block).  Used to keep arithmetic on just-loaded data next to the counted
// s_waitcnt of its own load batch instead of behind a later, conservative vmcnt(0).
LZ_DEV void lz_pin(u32& x) { asm volatile("" : "+v"(x)); }
#define LZ_STAT(i) ((void)0)                                   /* event counters exist in the test emulator only (tests/emul/lz_wave.h) */
// product of two values below 2^24: one full-rate v_mul_u32_u24 (v_mul_lo_u32 is a quarter-rate instruction)
LZ_DEV u32 lz_mul24(u32 a, u32 b) { return (u32)__umul24(a, b); }
// keeps the optimiser from folding a cheap form back into the expensive one it was written to avoid
LZ_DEV u32 lz_opaque(u32 x) { asm("" : "+v"(x)); return x; }
LZ_DEV u32 lz_mulhi(u32 a, u32 b) { return __umulhi(a, b); }
// initial value of a register that only the lanes which later assign it ever read: some valid value, whatever the register
// holds — no fill instruction on the device (the emulator's lanes hold 0)
LZ_DEV u64 lz_any64() { u64 x; asm volatile("" : "=v"(x)); return x; }
#define LZ_ANY64 lz_any64()

LZ_DEV u32 lz_ctz64(u64 m) { return (u32)__builtin_ctzll(m); }         // m != 0
LZ_DEV u32 lz_clz64(u64 m) { return (u32)__builtin_clzll(m); }         // m != 0
LZ_DEV u32 lz_popc64(u64 m) { return (u32)__builtin_popcountll(m); }

// LDS atomics (histograms, bit-string assembly); results unused -> ds_add_u32 / ds_or_b32 without return
LZ_DEV void lz_lds_atomic_add(u32* p, u32 v) { atomicAdd(p, v); }
LZ_DEV void lz_lds_atomic_or(u32* p, u32 v) { atomicOr(p, v); }
// LDS words shared by the waves of a workgroup (the Huffman workspace pool): returning OR, AND, and a load the
// optimiser may not hoist out of a polling loop.
LZ_DEV u32 lz_lds_atomic_or_rtn(u32* p, u32 v) { return atomicOr(p, v); }
LZ_DEV void lz_lds_atomic_and(u32* p, u32 v) { atomicAnd(p, v); }
LZ_DEV u32 lz_lds_poll(const u32* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
LZ_DEV void lz_sleep() { __builtin_amdgcn_s_sleep(8); }
// the same word as every lane of the wave sees it in ONE read (wave-uniform)
LZ_DEV u32 lz_lds_poll_u(const u32* p) { return lz_uniform(lz_lds_poll(p)); }
// Words shared between the WAVES of a workgroup (lz_split.h: mailboxes, free masks, counters).
//   lz_lds_claim       take the next ticket of an LDS counter: one atomic per wave, result wave-uniform (branch-free like lz_claim_index)
//   lz_lds_store       store of a word other waves poll (lz_lds_poll)
//   lz_publish_release / lz_publish_acquire   around such a word when it announces GLOBAL-memory data written by this wave / to be read
//                      by this wave: all earlier stores of every lane have completed (agent scope) before the word is stored; the
//                      vector L1 is invalidated after the word was seen, so reused buffers are re-read from L2
//   lz_ld_shared_u32 / lz_st_shared_u32        a global-memory word handed between waves (never served from the scalar cache)
LZ_DEV u32 lz_lds_claim(u32* counter)
{
    const u32 old = atomicAdd(counter, lz_lane() == 0 ? 1u : 0u);
    return lz_readlane(old, 0);
}
LZ_DEV void lz_lds_store(u32* p, u32 v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
LZ_DEV void lz_publish_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_wave_barrier(); }
LZ_DEV void lz_publish_acquire() { __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
LZ_DEV u32 lz_ld_shared_u32(const u32* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
LZ_DEV void lz_st_shared_u32(u32* p, u32 v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
// Masked bit-field store into an LDS dword, atomic per lane: *p = (*p & ~mask) | val  (ds_mskor_b32).  Lanes of one
// instruction may target different fields of the same dword.  val must lie inside mask.
LZ_DEV void lz_lds_mskor(u32* p, u32 mask, u32 val)
{
    asm volatile("ds_mskor_b32 %0, %1, %2" :: "v"((u32)(uintptr_t)p), "v"(mask), "v"(val) : "memory");
}

// Wave-wide reductions / exclusive prefix sum over all 64 lanes (every lane must call).
// DPP form (no LDS round trips): Hillis-Steele inside each row of 16 lanes with row_shr:1/2/4/8, then the
// two row broadcasts that carry a row's total into the following rows (row_bcast:15 to rows 1 and 3,
// row_bcast:31 to rows 2 and 3).  Lanes whose DPP source does not exist keep `old` = the identity 0.
template <int CTRL, int ROW_MASK>
LZ_DEV u32 lz_dpp0(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false); }
LZ_DEV u32 lz_wave_scan_incl_add(u32 v)
{
    v += lz_dpp0<0x111, 0xf>(v);      // row_shr:1
    v += lz_dpp0<0x112, 0xf>(v);      // row_shr:2
    v += lz_dpp0<0x114, 0xf>(v);      // row_shr:4
    v += lz_dpp0<0x118, 0xf>(v);      // row_shr:8
    v += lz_dpp0<0x142, 0xa>(v);      // row_bcast:15 -> rows 1, 3
    v += lz_dpp0<0x143, 0xc>(v);      // row_bcast:31 -> rows 2, 3
    return v;
}
LZ_DEV u32 lz_wave_scan_excl_add(u32 v) { return lz_wave_scan_incl_add(v) - v; }
LZ_DEV u32 lz_wave_reduce_add(u32 v) { return lz_readlane(lz_wave_scan_incl_add(v), 63u); }
LZ_DEV u32 lz_wave_reduce_max(u32 v)  // running max of unsigned values (identity 0), total in lane 63
{
    u32 o;
    o = lz_dpp0<0x111, 0xf>(v); v = o > v ? o : v;
    o = lz_dpp0<0x112, 0xf>(v); v = o > v ? o : v;
    o = lz_dpp0<0x114, 0xf>(v); v = o > v ? o : v;
    o = lz_dpp0<0x118, 0xf>(v); v = o > v ? o : v;
    o = lz_dpp0<0x142, 0xa>(v); v = o > v ? o : v;
    o = lz_dpp0<0x143, 0xc>(v); v = o > v ? o : v;
    return lz_readlane(v, 63u);
}

// Unaligned little-endian loads/stores from global memory.  gfx950 global/flat accesses have no
// alignment requirement (unaligned access mode), so these compile to single dword/dwordx2 ops.
struct __attribute__((packed, aligned(1))) lz_u16u { u16 v; };
struct __attribute__((packed, aligned(1))) lz_u32u { u32 v; };
struct __attribute__((packed, aligned(1))) lz_u64u { u64 v; };
LZ_DEV u32 lz_ld32(const u8* p) { return reinterpret_cast<const lz_u32u*>(p)->v; }
LZ_DEV u64 lz_ld64(const u8* p) { return reinterpret_cast<const lz_u64u*>(p)->v; }
LZ_DEV void lz_st16(u8* p, u32 v) { reinterpret_cast<lz_u16u*>(p)->v = (u16)v; }
LZ_DEV void lz_st32(u8* p, u32 v) { reinterpret_cast<lz_u32u*>(p)->v = v; }
LZ_DEV void lz_st64(u8* p, u64 v) { reinterpret_cast<lz_u64u*>(p)->v = v; }
// 16 bytes at once (one dwordx4 access, no alignment requirement either)
struct lz_u128 { u64 lo, hi; };
struct __attribute__((packed, aligned(1))) lz_u128u { u64 lo, hi; };
LZ_DEV lz_u128 lz_ld128(const u8* p) { const lz_u128u* q = reinterpret_cast<const lz_u128u*>(p); lz_u128 r; r.lo = q->lo; r.hi = q->hi; return r; }
LZ_DEV void lz_st128(u8* p, lz_u128 v) { lz_u128u* q = reinterpret_cast<lz_u128u*>(p); q->lo = v.lo; q->hi = v.hi; }
// Two masked bit-field EXCHANGES in LDS in one round trip (ds_mskor_rtn_b32 twice, one wait): *pa = (*pa & ~ma) | va and the
// same for b; oa / ob receive the dwords as they were before this lane's update.  The lanes of one DS atomic instruction that
// hit the same dword are served in ascending lane order on gfx950 (tests/lds_atomic_order.hip: 0 violations in
// 2.5 M colliding wave-instructions; pinned by tests/test_gpu_parity.py::test_lds_atomics_are_served_in_lane_order): a later lane
// of a round sees an earlier lane's update — the in-order view of a serial hash-table walk, for free.
LZ_DEV void lz_lds_mskor_rtn2(LZ_LDS u32* pa, u32 ma, u32 va, LZ_LDS u32* pb, u32 mb, u32 vb, u32& oa, u32& ob)
{
    asm volatile("ds_mskor_rtn_b32 %0, %2, %3, %4\n\tds_mskor_rtn_b32 %1, %5, %6, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(oa), "=&v"(ob)
                 : "v"((u32)(uintptr_t)pa), "v"(ma), "v"(va), "v"((u32)(uintptr_t)pb), "v"(mb), "v"(vb) : "memory");
}

// The same for TWO slots per lane in one round trip: a's pair, then b's pair, one wait.  The LDS unit executes the four instructions
// in order, so every lane's b-exchange sees all 64 a-exchanges (128 exchanges in the order a0..a63, b0..b63).
LZ_DEV void lz_lds_mskor_rtn4(LZ_LDS u32* pa, u32 ma, u32 va, LZ_LDS u32* pb, u32 mb, u32 vb, LZ_LDS u32* pc, u32 mc, u32 vc, LZ_LDS u32* pd, u32 md, u32 vd,
                              u32& oa, u32& ob, u32& oc, u32& od)
{
    asm volatile("ds_mskor_rtn_b32 %0, %4, %5, %6\n\tds_mskor_rtn_b32 %1, %7, %8, %9\n\tds_mskor_rtn_b32 %2, %10, %11, %12\n\tds_mskor_rtn_b32 %3, %13, %14, %15\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(oa), "=&v"(ob), "=&v"(oc), "=&v"(od)
                 : "v"((u32)(uintptr_t)pa), "v"(ma), "v"(va), "v"((u32)(uintptr_t)pb), "v"(mb), "v"(vb),
                   "v"((u32)(uintptr_t)pc), "v"(mc), "v"(vc), "v"((u32)(uintptr_t)pd), "v"(md), "v"(vd) : "memory");
}

// Returning exchange / add on an LDS dword; every lane of the wave takes part (lanes with nothing to do aim at a spare word).
// Lane order as above: lane l receives what the closest lower lane of the same dword left behind — lz_lds_add_rtn with 1 hands
// out consecutive slots in lane order (a stable partition step), lz_lds_xchg_rtn is "read the head, become the head".
LZ_DEV u32 lz_lds_xchg_rtn(u32* p, u32 v)
{
    u32 o;
    asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(o) : "v"((u32)(uintptr_t)p), "v"(v) : "memory");
    return o;
}
LZ_DEV u32 lz_lds_add_rtn(u32* p, u32 v)
{
    u32 o;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(o) : "v"((u32)(uintptr_t)p), "v"(v) : "memory");
    return o;
}

// One-pass stream traffic (sequence list, the encode pass's literal re-reads, the output streams): with
// -DLZ_NT_STREAMS these carry the `nt` hint so that they do not displace the hash tables that live in L2.
#ifdef LZ_NT_STREAMS
typedef u64 __attribute__((aligned(1))) lz_u64nt;
typedef u32 __attribute__((aligned(1))) lz_u32nt;
typedef u16 __attribute__((aligned(1))) lz_u16nt;
LZ_DEV u64 lz_ld64_s(const u8* p) { return __builtin_nontemporal_load((const lz_u64nt*)p); }
LZ_DEV u32 lz_ld32_s(const u8* p) { return __builtin_nontemporal_load((const lz_u32nt*)p); }
LZ_DEV u8  lz_ld8_s(const u8* p) { return __builtin_nontemporal_load(p); }
LZ_DEV void lz_st64_s(u8* p, u64 v) { __builtin_nontemporal_store(v, (lz_u64nt*)p); }
LZ_DEV void lz_st32_s(u8* p, u32 v) { __builtin_nontemporal_store(v, (lz_u32nt*)p); }
LZ_DEV void lz_st16_s(u8* p, u32 v) { __builtin_nontemporal_store((u16)v, (lz_u16nt*)p); }
LZ_DEV void lz_st8_s(u8* p, u32 v) { __builtin_nontemporal_store((u8)v, p); }
LZ_DEV u64 lz_ldq_s(const u64* p) { return __builtin_nontemporal_load(p); }
LZ_DEV void lz_stq_s(u64* p, u64 v) { __builtin_nontemporal_store(v, p); }
#else
LZ_DEV u64 lz_ld64_s(const u8* p) { return lz_ld64(p); }
LZ_DEV u32 lz_ld32_s(const u8* p) { return lz_ld32(p); }
LZ_DEV u8  lz_ld8_s(const u8* p) { return *p; }
LZ_DEV void lz_st64_s(u8* p, u64 v) { lz_st64(p, v); }
LZ_DEV void lz_st32_s(u8* p, u32 v) { lz_st32(p, v); }
LZ_DEV void lz_st16_s(u8* p, u32 v) { lz_st16(p, v); }
LZ_DEV void lz_st8_s(u8* p, u32 v) { *p = (u8)v; }
LZ_DEV u64 lz_ldq_s(const u64* p) { return *p; }
LZ_DEV void lz_stq_s(u64* p, u64 v) { *p = v; }
#endif
#endif  /* LZ_WAVE_H_ */
