// woq_prefill_common.h -- LDS-DMA / LDS-read helpers shared by the prefill GEMM (woq_prefill.hip) and its skinny form
// (woq_skinny.hip).  Everything here is internal to those two translation units.
#pragma once
#include "woq_common.h"

namespace mi355q
{
namespace
{

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

constexpr int PF_WAVES = 4;

// LDS-DMA from inline asm (see context_attn.hip): after __builtin_amdgcn_global_load_lds hipcc makes every following
// LDS read wait for vmcnt(0), i.e. the compute of tile kt waited for the prefetch of tile kt + 1 that had just been
// issued.  lds_off: wave-uniform LDS byte address; lane l's 16 bytes land at lds_off + 16 l.
// The source is base (SGPR pair) + a 32-bit per-lane byte offset: half the address registers of a 64-bit pointer.
__device__ __forceinline__ void pf_dma16(const void* base, uint32_t voff, uint32_t lds_off)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_off)
                 : "memory", "m0");
}

template <int I>
struct PfSlot
{
    static constexpr int value = I;
};

typedef uint32_t pf_u32x4_t __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ void pf_lds_read(pf_u32x4_t& d, uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// d is usable after this statement (N = LDS reads issued after the one that fills d)
template <int N>
__device__ __forceinline__ void pf_lds_wait(pf_u32x4_t& d)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N));
}
template <int N, int I = 0, typename F>
__device__ __forceinline__ void pf_static_for(F&& f)
{
    if constexpr (I < N)
    {
        f(PfSlot<I>{});
        pf_static_for<N, I + 1>(f);
    }
}

} // namespace
} // namespace mi355q
