// Hand-managed vector-memory queue (gfx950): inline-asm loads that hipcc does not see, the s_waitcnt vmcnt(N) that releases
// them, LDS-DMA, compile-time loops.  Shared by w8a8.hip (config 3's unrolled K loop) and w4_gemm256.hip (the 256 x 256 int4 GEMM).
// Rules of use (cdna_hip_programming.md 5.7): loads complete in issue order, so "X has landed" = "at most N younger loads are
// outstanding"; a register written by gload*() is handed to the compiler only through vm_wait_imm<N>(reg) (the "+v" tie keeps
// every use behind the wait); drain with vm_wait_imm<0>() before the registers' last use goes out of scope.
#pragma once
#include <utility>

#include "ql_common.h"

namespace ql {

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void vm_wait_imm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait_imm(i32x4& w) {          // ... tied to the fragment it releases: its MFMA cannot move above
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w) : "n"(N) : "memory");
}
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}) - the wait counts are
// template arguments (a run-time switch over 41 immediates per wait made the loop body too large for hipcc to unroll)
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// 64 lanes x 16 bytes: global (wave-uniform 64-bit base + per-lane byte offset) -> 1 KB of LDS at `lds_dst` + 16 lane.  M0 holds
// the LDS address and is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(unsigned lds_dst, unsigned voff, unsigned long long base) {
    unsigned keep;
    // s_nop 4: when hipcc materialises the base with v_readfirstlane (it does whenever it cannot prove the chunk index uniform), a
    // VALU write of an SGPR needs 5 wait states before a VMEM instruction reads it, and the hazard pass does not look inside asm
    // (the stamped build faulted on addresses with a stale high half until these were added)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_dst), "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void gload16(i32x4& dst, unsigned voff, unsigned long long base) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ unsigned long long sgpr64(unsigned long long v) {   // wave-uniform value -> an SGPR pair
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

}  // namespace ql
