// Device check (tests/test_gpu_parity.py::test_lds_atomics_are_served_in_lane_order): do the lanes of ONE wave64 DS atomic instruction that hit the same LDS dword get
// processed in ascending lane order?  ds_wrxchg_rtn_b32, ds_mskor_rtn_b32 and ds_add_rtn_u32, random address patterns with many collisions.
// Prints the number of violations (a returned value that is not the value stored by the closest lower lane on the same word /
// bit-field, or the initial value when there is none).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const unsigned* addr, unsigned* bad, int trials)
{
    __shared__ unsigned s[256];
    const unsigned lane = threadIdx.x;
    unsigned nbad = 0;
    for (int t = 0; t < trials; t++) {
        for (unsigned i = lane; i < 256; i += 64) s[i] = 0xABCD0000u + i;
        __syncthreads();
        const unsigned a = addr[t * 64 + lane] & 255u;
        // expected: value of the closest lower lane with the same address, else initial
        unsigned expect = 0xABCD0000u + a;
        for (unsigned l = 0; l < 64; l++) { const unsigned al = __shfl(a, l); if (l < lane && al == a) expect = 0x1000u + l + (unsigned)t * 64u; }
        unsigned old;
        const unsigned val = 0x1000u + lane + (unsigned)t * 64u, off = a * 4u;
        asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(old) : "v"((unsigned)(size_t)s + off), "v"(val) : "memory");
        if (old != expect) nbad++;
        __syncthreads();
        // 16-bit fields through mskor: slot index a2 in 0..511, two per dword
        for (unsigned i = lane; i < 256; i += 64) s[i] = 0;
        __syncthreads();
        const unsigned a2 = addr[t * 64 + lane] & 511u;
        unsigned exp2 = 0;
        for (unsigned l = 0; l < 64; l++) { const unsigned al = __shfl(a2, l); if (l < lane && al == a2) exp2 = (l + 1u + (unsigned)t) & 0xFFFFu; }
        const unsigned sh = (a2 & 1u) * 16u, v2 = ((lane + 1u + (unsigned)t) & 0xFFFFu) << sh, m2 = 0xFFFFu << sh;
        unsigned old2;
        asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n s_waitcnt lgkmcnt(0)" : "=v"(old2) : "v"((unsigned)(size_t)s + (a2 >> 1) * 4u), "v"(m2), "v"(v2) : "memory");
        if (((old2 >> sh) & 0xFFFFu) != exp2) nbad++;
        __syncthreads();
        // returning add of 1: consecutive slots in lane order (the stable partition step of lz_hc_build)
        for (unsigned i = lane; i < 256; i += 64) s[i] = 1000u * i;
        __syncthreads();
        unsigned exp3 = 1000u * a;
        for (unsigned l = 0; l < 64; l++) { const unsigned al = __shfl(a, l); if (l < lane && al == a) exp3++; }
        unsigned old3;
        asm volatile("ds_add_rtn_u32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(old3) : "v"((unsigned)(size_t)s + off), "v"(1u) : "memory");
        if (old3 != exp3) nbad++;
        __syncthreads();
    }
    atomicAdd(bad, nbad);
}
int main()
{
    const int trials = 20000;
    unsigned* h = (unsigned*)malloc(trials * 64 * 4);
    srand(1);
    for (int t = 0; t < trials; t++) {
        const int mode = t % 5, span = mode == 0 ? 1 : mode == 1 ? 4 : mode == 2 ? 32 : mode == 3 ? 64 : 512;
        for (int l = 0; l < 64; l++) h[t * 64 + l] = (unsigned)(rand() % span) * (mode == 2 ? 32u : 1u) + (mode == 4 ? 0u : (unsigned)(rand() % 2) * 0u);
    }
    unsigned *d, *bad, hb = 0;
    (void)hipMalloc((void**)&d, trials * 64 * 4); (void)hipMalloc((void**)&bad, 4);
    (void)hipMemcpy(d, h, trials * 64 * 4, hipMemcpyHostToDevice); (void)hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d, bad, trials);
    (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("lds atomic lane-order violations: %u (of %d trials x 3 ops x 64 workgroups)\n", hb, trials);
    return hb != 0;
}
