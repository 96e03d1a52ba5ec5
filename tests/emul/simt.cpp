// tests/emul/simt.cpp — TEST INFRASTRUCTURE ONLY: scheduler of the 64-lane SIMT emulator (lz_wave.h).
#include "lz_wave.h"

namespace lzemu {

thread_local Wave* g_wave = nullptr;

// Minimal x86-64 SysV context switch: save callee-saved registers on the current stack, publish the
// stack pointer, adopt the other stack and restore its registers.
__asm__(
    ".text\n"
    ".globl lzemu_ctx_switch\n"
    ".type lzemu_ctx_switch,@function\n"
    "lzemu_ctx_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size lzemu_ctx_switch,.-lzemu_ctx_switch\n");

static void trampoline()
{
    Wave* w = g_wave;
    w->entry(w->entry_arg);
    w = g_wave;
    w->op[w->cur] = OP_DONE;
    lzemu_ctx_switch(&w->lane_sp[w->cur], w->sched_sp);
    abort();   // a finished lane is never resumed
}

static u32 xorshift(u32* s) { u32 x = *s; x ^= x << 13; x ^= x >> 17; x ^= x << 5; *s = x; return x; }

void run_wave(void (*entry)(void*), void* arg, u32 seed)
{
    const size_t STACK = 192 * 1024;
    Wave* w = (Wave*)calloc(1, sizeof(Wave));
    Wave* saved = g_wave;
    w->stacks = (u8*)malloc(STACK * LZ_WAVE);
    w->entry = entry; w->entry_arg = arg; w->rng = seed ? seed : 0x9E3779B9u;
    for (int l = 0; l < LZ_WAVE; l++) {
        uintptr_t top = ((uintptr_t)(w->stacks + STACK * (l + 1))) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;                 // fake return address of trampoline (keeps rsp % 16 == 8 at entry)
        *--sp = (void*)&trampoline;      // popped by `ret` in lzemu_ctx_switch
        for (int i = 0; i < 6; i++) *--sp = nullptr;
        w->lane_sp[l] = (void*)sp;
        w->op[l] = OP_NONE;
    }
    g_wave = w;
    int order[LZ_WAVE];
    for (;;) {
        for (int i = 0; i < LZ_WAVE; i++) order[i] = i;
        for (int i = LZ_WAVE - 1; i > 0; i--) { int j = (int)(xorshift(&w->rng) % (u32)(i + 1)); int t = order[i]; order[i] = order[j]; order[j] = t; }
        for (int i = 0; i < LZ_WAVE; i++) {
            int l = order[i];
            if (w->op[l] == OP_DONE) continue;
            w->cur = l;
            lzemu_ctx_switch(&w->sched_sp, w->lane_sp[l]);
        }
        int op = w->op[0];
        for (int l = 1; l < LZ_WAVE; l++)
            if (w->op[l] != op) { fprintf(stderr, "lzemu: divergent cross-lane op: lane 0 at %d, lane %d at %d\n", op, l, w->op[l]); abort(); }
        if (op == OP_DONE) break;
        w->n_ops++;
        switch (op) {
        case OP_BALLOT: {
            u64 m = 0;
            for (int l = 0; l < LZ_WAVE; l++) m |= (w->arg0[l] & 1) << l;
            for (int l = 0; l < LZ_WAVE; l++) w->res[l] = m;
            break; }
        case OP_READLANE: {
            u64 s = w->arg1[0];
            for (int l = 1; l < LZ_WAVE; l++)
                if (w->arg1[l] != s) {
                    fprintf(stderr, "lzemu: lz_readlane with a non-uniform lane index (lane 0: %llu, lane %d: %llu; values", (unsigned long long)s, l, (unsigned long long)w->arg1[l]);
                    for (int k = 0; k < LZ_WAVE; k += 8) fprintf(stderr, " [%d]=%llu", k, (unsigned long long)w->arg1[k]);
                    fprintf(stderr, ")\n"); abort(); }
            if (s >= LZ_WAVE) { fprintf(stderr, "lzemu: lz_readlane index %llu out of range\n", (unsigned long long)s); abort(); }
            for (int l = 0; l < LZ_WAVE; l++) w->res[l] = w->arg0[s];
            break; }
        case OP_UNIFORM: {
            for (int l = 1; l < LZ_WAVE; l++)
                if (w->arg0[l] != w->arg0[0]) { fprintf(stderr, "lzemu: lz_uniform on a non-uniform value (lane 0: %llu, lane %d: %llu)\n", (unsigned long long)w->arg0[0], l, (unsigned long long)w->arg0[l]); abort(); }
            for (int l = 0; l < LZ_WAVE; l++) w->res[l] = w->arg0[0];
            break; }
        case OP_SHFL:
            for (int l = 0; l < LZ_WAVE; l++) w->res[l] = w->arg0[w->arg1[l]];
            break;
        case OP_SYNC:
            break;
        case OP_MSKOR2:
            for (int k = 0; k < 2; k++)
                for (int l = 0; l < LZ_WAVE; l++) {          // one instruction at a time, lanes in ascending order
                    const u32 old = *w->xp[k][l];
                    *w->xp[k][l] = (old & ~w->xm[k][l]) | w->xv[k][l];
                    w->xo[k][l] = old;
                }
            break;
        case OP_ATOM1:
            for (int l = 0; l < LZ_WAVE; l++) {              // lanes in ascending order
                const u32 old = *w->xp[0][l];
                *w->xp[0][l] = w->xm[0][l] ? old + w->xv[0][l] : w->xv[0][l];
                w->xo[0][l] = old;
            }
            break;
        default:
            fprintf(stderr, "lzemu: bad op %d\n", op); abort();
        }
        for (int l = 0; l < LZ_WAVE; l++) w->op[l] = OP_NONE;
    }
    g_wave = saved;
    free(w->stacks);
    free(w);
}

}  // namespace lzemu
