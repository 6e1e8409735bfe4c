// woq_gemm.hip -- weight-only quantized GEMV / GEMM: fp16|bf16 activations x int4|int8 weights
// (group-wise AWQ/GPTQ scale(+zero) or per-channel scale).  Replaces
//   kernels/weightOnlyBatchedGemv/{kernel.h,kernelLauncher.cu}  (m < 5, CUDA cores) and
//   kernels/cutlass_kernels/fpA_intB_gemm/*                      (m >= 5, dequant-in-mainloop)
// of the reference with one gfx950 design:
//   * weights live in the GFX950 layout (preprocess.cpp): 1 KiB tiles of 16 columns x 128 (int4) or
//     64 (int8) rows; one 16-byte load per lane is already the B fragment of 4 (2) consecutive
//     v_mfma_f32_16x16x32 k-steps -- no LDS, no shuffles;
//   * decode (m <= 32): the persistent kernel below.  One 16-wave workgroup per CU stages its activation rows in
//     LDS once and walks its n-blocks; wave w owns a contiguous k-tile range and keeps two chunks of 2 KiB of
//     weights in flight while it computes a third (ring of register images, nontemporal buffer loads with SGPR tile
//     offsets, issued unconditionally so the compiler's static vmcnt bookkeeping keeps them in flight).  int4 runs in
//     the INTEGER domain: P = sum a*u on the MFMA (u = q + 8 unpacked exactly through the fp8 converter), S1 = sum a
//     from a per-block LDS table, scale / zero applied to the group's partial sums -- about 8 VALU operations per
//     scale group instead of ~2.8 per weight; int8 dequantises with the exact fma.  fp32 accumulators are reduced
//     through LDS per n-block (one barrier) while the next n-blocks' weights are already in flight;
//   * the bit-exact dequantisation w = round16(fma(q, scale, zero)) (kernel.h:364-365,
//     mma_tensorop_dequantizer.h:244-252) backs mi355q_weight_only_dequantize and the prefill GEMM (woq_prefill.hip);
//   * optional split-K across workgroups (fp32 slabs in the workspace + a small reduce kernel) when the activation
//     slab of a workgroup would not fit LDS or N/16 workgroups cannot fill the chip.
#include "woq_common.h"
#include <cstdlib>
#include <utility>

namespace mi355q
{

namespace
{


// The weight side of CH consecutive k-tiles of one n-block, issued as one burst of independent,
// branch-free buffer loads: per-lane byte offset in a VGPR (set once), the tile / group offset in an
// SGPR -- no per-load vector address arithmetic.  Weights: CH KiB contiguous per wave, nontemporal
// (streamed once); scales / zeros come from L2.
// NSZ = (scale, zero) pairs per tile: 2 only for int4 with group size 64 (a 128-row tile spans two).
template <int CH, int NSZ>
struct GemvChunk
{
    uint4 w[CH];
    uint16_t s[CH][NSZ], z[CH][NSZ];
};

struct GemvSrc
{
    __amdgpu_buffer_rsrc_t w, s, z;
    uint32_t w_voff, sz_voff; // lane * 16, col * 2
    int group_shift;          // log2(group), or 31 for per-channel (every k maps to group 0)
    uint32_t row_bytes;       // n * 2: stride between groups in scales / zeros
};

template <int WBITS, bool ZERO, int CH, int NSZ>
__device__ __forceinline__ void gemv_load_chunk(GemvChunk<CH, NSZ>& c, const GemvSrc& src, int kt0, int kt_last)
{
    constexpr int KTILE = Geo<WBITS>::KTILE;
#pragma unroll
    for (int i = 0; i < CH; ++i)
    {
        const int kt = min(kt0 + i, kt_last); // tiles past the end re-read the last one; zeroed in compute
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(src.w, src.w_voff, kt * 1024, 2 /* nt */);
        c.w[i] = make_uint4(v.x, v.y, v.z, v.w);
#pragma unroll
        for (int h = 0; h < NSZ; ++h)
        {
            const uint32_t goff = (uint32_t) ((kt * KTILE + h * 64) >> src.group_shift) * src.row_bytes;
            c.s[i][h] = __builtin_amdgcn_raw_buffer_load_b16(src.s, src.sz_voff, goff, 0);
            c.z[i][h] = ZERO ? __builtin_amdgcn_raw_buffer_load_b16(src.z, src.sz_voff, goff, 0) : (uint16_t) 0;
        }
    }
}

template <typename T, int WBITS, int ST>
__device__ __forceinline__ f32x4_t gemv_step(const uint4& w, const uint4& av, float s, float z, f32x4_t acc)
{
    return mfma16<T>(av, dequant_step<T, WBITS, ST>(w, s, z), acc);
}

// floats of the S1 table of a block: one per (64-row unit of its k range, row of its 16 * mc rows), rounded so the
// activation slab behind it stays 16-byte aligned; sized for the finest unit (group size 64) whatever the group is
__host__ __device__ constexpr int gemv_s1_floats(int krange, int mc)
{
    return ((krange + 63) / 64) * 16 * mc;
}

// One int4 k-tile (STEPS k-steps of 32 rows) in the integer domain.  Per scale group g of the tile:
//     P  = sum_k a[m,k] * u[k,n]      (MFMA, u = q + 8 exact in 16 bits, fp32 accumulate)
//     S1 = sum_k a[m,k]               (depends on the activations only: summed once per block while they are staged,
//                                      `s1frag(mc, h)` reads the lane's 4 rows of it from LDS)
//     acc += scale * (P - 8 * S1) + zero * S1   ==   sum_k a * (q * scale + zero)
// i.e. the scale / zero are applied to the group's partial sums (12 VALU ops per group) instead of
// to every weight.  Unlike kernel.h:364-365 / mma_tensorop_dequantizer.h the dequantised weight is
// not rounded to 16 bits before the multiply, so for general activations results differ from the
// exact-dequant path by about half an output ulp (closer to the real-number result).  Written in this
// order -- not as scale * P + (zero - 8 * scale) * S1, whose second factor is rounded -- a one-hot
// activation row gives P - 8 S1 = q and zero * S1 = zero exactly, i.e. fma(q, scale, zero) itself:
// the reference's identity-activation gate (test_weight_only_quant_matmul.py:44-53) recovers the
// dequantised weights bit for bit through this kernel too.
template <typename T, int NSZ, int STEPS, int MC, typename AF, typename SF>
__device__ __forceinline__ void gemv_tile_int4(
    const uint4& w, f32x4_t (&acc)[MC], const float* s, const float* z, AF afrag, SF s1frag)
{
    const uint32_t* wd = reinterpret_cast<const uint32_t*>(&w);
    constexpr int PER = STEPS / NSZ; // k-steps per scale group
#pragma unroll
    for (int h = 0; h < NSZ; ++h)
    {
        f32x4_t P[MC];
#pragma unroll
        for (int mc = 0; mc < MC; ++mc)
            P[mc] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = h * PER; st < (h + 1) * PER; ++st)
        {
            const auto u = unpack8_u4<T>(wd[st]); // unpacked once, used by every 16-row chunk of m
#pragma unroll
            for (int mc = 0; mc < MC; ++mc)
                P[mc] = mfma16<T>(afrag(mc, st), u, P[mc]);
        }
#pragma unroll
        for (int mc = 0; mc < MC; ++mc)
        {
            const f32x4_t S1 = s1frag(mc, h);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[mc][r] = fmaf(s[h], fmaf(-8.f, S1[r], P[mc][r]), fmaf(z[h], S1[r], acc[mc][r]));
        }
    }
}

// `afrag(mc, kt, st)` returns the activation fragment of k-step st of tile kt for m-chunk mc (LDS).
template <typename T, int WBITS, int CH, int NSZ, int MC, typename AF, typename SF>
__device__ __forceinline__ void gemv_compute_chunk(
    const GemvChunk<CH, NSZ>& c, f32x4_t (&acc)[MC], int kt0, int kt_end, int kt_last, AF afrag, SF s1frag)
{
#pragma unroll
    for (int i = 0; i < CH; ++i)
    {
        // wave-uniform branch: a tile past the wave's range holds unrelated weights and its
        // activation fragment may lie outside the staged LDS slab (NaN garbage), so it is skipped
        if (kt0 + i >= kt_end)
            continue;
        const int kt = kt0 + i;
        float s[2], z[2];
#pragma unroll
        for (int h = 0; h < NSZ; ++h)
        {
            s[h] = T16<T>::to_float(c.s[i][h]);
            z[h] = T16<T>::to_float(c.z[i][h]);
        }
        if constexpr (WBITS == 4)
            gemv_tile_int4<T, NSZ, 4, MC>(
                c.w[i], acc, s, z, [&](int mc, int st) { return afrag(mc, kt, st); },
                [&](int mc, int h) { return s1frag(mc, kt, h); });
        else
        {
            const auto w0 = dequant_step<T, 8, 0>(c.w[i], s[0], z[0]);
            const auto w1 = dequant_step<T, 8, 1>(c.w[i], s[0], z[0]);
#pragma unroll
            for (int mc = 0; mc < MC; ++mc)
            {
                acc[mc] = mfma16<T>(afrag(mc, kt, 0), w0, acc[mc]);
                acc[mc] = mfma16<T>(afrag(mc, kt, 1), w1, acc[mc]);
            }
        }
    }
}

// step(0) || step(1) || ... : one trip round the register ring, stops at the first position that reports "done"
template <typename F, int... B>
__device__ __forceinline__ bool gemv_ring_round(F& step, std::integer_sequence<int, B...>)
{
    return (step(std::integral_constant<int, B>{}) || ...);
}

// Persistent decode kernel.  grid: x = n-group, y = 16-row chunk of m, z = k segment; block =
// NWAVES waves.  A block stages its activation rows [rows][k segment] into LDS once (row stride
// kseg*2+32 bytes: conflict-free ds_read_b128 fragments) and then walks the n-blocks x, x+gridDim.x,
// ...: wave w owns a contiguous range of the segment's k-tiles, cut into chunks of CH tiles, and
// runs ONE ping-pong load pipeline over the whole (n-block, chunk) sequence, so the next n-block's
// weights are already in flight while the current one is reduced (LDS, one 16x16 tile per n-block)
// and stored -- the HBM queue never drains between n-blocks.
// MC = 16-row chunks of m handled by one block (1, 2): for 16 < m <= 32 the weights are still streamed ONCE and
// every unpacked fragment feeds MC MFMAs (the prefill GEMM's 2-barrier tile loop is latency-bound at these sizes).
// NBUF = register images in the ring (NBUF - 1 chunks of CH tiles in flight during a chunk's compute).
// WPN ("wave per n-block"): instead of splitting the block's k segment over the waves, wave w takes n-block 16 s + w
// of the block's current set s over the WHOLE segment -- no cross-wave reduction, no barrier after the staging, the
// accumulators go straight to the output (or the split-K slab).  For 16 < m <= 32, where the activation slab forces
// short k segments and the per-n-block reduction (two barriers for one tile per wave) was most of the time.
template <typename T, int WBITS, bool ZERO, int NWAVES, int CH, int NSZ, int MC, int NBUF, bool WPN = false>
__global__ __launch_bounds__(NWAVES * 64) void woq_gemv_kernel(WoqArgs a)
{
    constexpr int KTILE = Geo<WBITS>::KTILE;
    // LDS: red[RB][MC][NWAVES][64][4] floats | s1[units][16 * MC] floats | activation rows
    const int RB = a.red_buffers;          // reduce buffers: two (one barrier per n-block) when LDS has the room
    constexpr int UNIT = KTILE / NSZ;      // k rows that share one (scale, zero): the granule of S1
    constexpr int RED_FLOATS = MC * NWAVES * 64 * 4;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* red = reinterpret_cast<float*>(smem);
    float* s1_lds = red + RB * RED_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * 16 * MC;
    const int kh = (lane >> 4) * 8;
    const int rows = min(16 * MC, a.m - m0);
    const int nblocks_all = a.n >> 4;
    const int nblocks = WPN ? (nblocks_all + NWAVES - 1) / NWAVES : nblocks_all; // WPN: sets of NWAVES n-blocks
    const int nb_stride = gridDim.x;
    // Workgroups go to the 8 XCDs round-robin, so blocks with equal blockIdx.x % 8 share an L2.  Give each XCD a
    // contiguous range of n-blocks per round: a 128-byte line of scales / zeros covers 4 n-blocks, and with
    // neighbouring n-blocks on different XCDs every line was fetched from HBM by 4 L2s (PMC: 147 MB read per launch
    // at 28672 x 8192 against 125 MB algorithmic).
    int nb_first;
    {
        const int q = nb_stride / 8, r = nb_stride % 8, x = blockIdx.x % 8;
        nb_first = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + blockIdx.x / 8;
    }

    // k segment of this block, and this wave's contiguous tile range in it
    const int seg = (a.ktiles + a.ksplit - 1) / a.ksplit; // 0 when k == 64 on int4: half tile only
    const int seg_begin = blockIdx.z * seg;
    const int seg_end = min(a.ktiles, seg_begin + seg);
    const int tpw = WPN ? seg : (seg + NWAVES - 1) / NWAVES; // tiles per wave (block-uniform)
    const int cpw = max(1, (tpw + CH - 1) / CH);   // chunks per wave per n-block (block-uniform)
    const int kt_begin = WPN ? seg_begin : seg_begin + wave * tpw;
    const int kt_end = min(seg_end, kt_begin + tpw); // may be <= kt_begin: this wave only helps reduce
    const int kt_hi = max(seg_end - 1, 0);           // clamp for dead tiles (a valid tile of the segment)
    const bool half_tile = WBITS == 4 && (a.k & 127) && blockIdx.z == a.ksplit - 1;
    const int k_lo = seg_begin * KTILE;
    const int k_hi = half_tile ? a.k : seg_end * KTILE;

    const size_t run_bytes = (size_t) a.k * 16 * WBITS / 8;
    const size_t sz_bytes = (size_t) (a.k / a.group) * a.n * 2;
    GemvSrc src;
    src.s = make_rsrc(a.scales, sz_bytes);
    src.z = make_rsrc(ZERO ? a.zeros : a.scales, sz_bytes);
    src.w_voff = lane * 16;
    src.group_shift = a.group_shift;
    src.row_bytes = a.n * 2;
    const __amdgpu_buffer_rsrc_t s_rsrc = src.s, z_rsrc = src.z;
    const __amdgpu_buffer_rsrc_t s_dead = make_rsrc(a.scales, 0), z_dead = make_rsrc(a.scales, 0);
    // `live == false` (past the block's last n-block): the same loads against zero-sized descriptors -- they return 0
    // without touching memory.  The loads must not sit under a branch: the compiler counts vmcnt statically, and a
    // path that issues nothing forces `s_waitcnt vmcnt(0)` on the chunk computed next, i.e. a wait for the chunk
    // that was requested a moment ago (no prefetch at all).
    auto issue = [&](GemvChunk<CH, NSZ>& c, int nb, int ck, bool live)
    {
        if constexpr (WPN)
        {
            nb = nb * NWAVES + wave;
            live = live && nb < nblocks_all;
            nb = live ? nb : 0;
        }
        src.w = make_rsrc(a.qweight + (size_t) nb * run_bytes, live ? run_bytes : 0);
        src.s = live ? s_rsrc : s_dead;
        src.z = live ? z_rsrc : z_dead;
        src.sz_voff = (nb * 16 + (lane & 15)) * 2;
        gemv_load_chunk<WBITS, ZERO, CH, NSZ>(c, src, kt_begin + ck * CH, kt_hi);
    };

    // rows >= m are clamped to the last valid row: their MFMA output rows are computed, never stored
    const int lds_stride = (k_hi - k_lo) * 2 + 32;
    const int units = (k_hi - k_lo + UNIT - 1) / UNIT;
    uint8_t* a_lds = reinterpret_cast<uint8_t*>(s1_lds + gemv_s1_floats(k_hi - k_lo, MC));

    GemvChunk<CH, NSZ> ring[NBUF];
    int l_nb = nb_first, l_ck = 0; // next chunk to load
    auto load_next = [&](GemvChunk<CH, NSZ>& c)
    {
        const bool live = l_nb < nblocks;
        issue(c, live ? l_nb : 0, l_ck, live);
        if (++l_ck == cpw)
        {
            l_ck = 0;
            l_nb += nb_stride;
        }
    };
    // weights first: their HBM latency overlaps the activation staging below
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
        load_next(ring[b]);

    {
        // stage the rows; int4: the same pass sums every UNIT of k per row (S1 of gemv_tile_int4) -- a thread adds its
        // 8 values, the UNIT / 8 neighbouring lanes that hold one unit combine through shuffles
        constexpr int VPU = UNIT / 8;                        // 16-byte vectors per unit (16 or 8)
        const int vec_per_row = (k_hi - k_lo) / 8;           // multiple of 8 (k % 64 == 0)
        const int vpr_pad = (vec_per_row + VPU - 1) / VPU * VPU;
        const int total = rows * vpr_pad;
        // SB vectors per thread and trip, all loads first (clamped addresses, no branch): with one load per trip a
        // block of 16 rows paid 8 L2 round trips one after the other (M = 16 at 4096 x 4096: 10 us against 5 at M = 1)
        constexpr int SB = 4;
        for (int base = tid - lane; base < total; base += SB * NWAVES * 64)
        {
            uint4 x[SB], sc[SB];
            int r[SB], v[SB];
            bool valid[SB];
#pragma unroll
            for (int b = 0; b < SB; ++b)
            {
                const int i = base + b * NWAVES * 64 + lane;
                r[b] = i / vpr_pad;
                v[b] = i - r[b] * vpr_pad;
                valid[b] = i < total && v[b] < vec_per_row;
                const int rc = valid[b] ? r[b] : 0, vc = valid[b] ? v[b] : 0;
                x[b] = *reinterpret_cast<const uint4*>(a.act + (size_t) (m0 + rc) * a.k + k_lo + vc * 8);
            }
            if (a.act_scale)
            {
#pragma unroll
                for (int b = 0; b < SB; ++b)
                    sc[b] = *reinterpret_cast<const uint4*>(a.act_scale + k_lo + (valid[b] ? v[b] : 0) * 8);
            }
#pragma unroll
            for (int b = 0; b < SB; ++b)
            {
                if (base + b * NWAVES * 64 >= total) // wave-uniform: nothing left in this trip
                    break;
                float part = 0.f;
                if (a.act_scale)
                {
                    // AWQ pre-quant scale fused into the staging: act * scale rounded to T, the values the separate
                    // mi355q_apply_per_channel_scale pass (preQuantScaleKernel.cu) would have written
                    uint32_t* xp = reinterpret_cast<uint32_t*>(&x[b]);
                    const uint32_t* sp = reinterpret_cast<const uint32_t*>(&sc[b]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        xp[j] = pack2<T>(T16<T>::lo(xp[j]) * T16<T>::lo(sp[j]), T16<T>::hi(xp[j]) * T16<T>::hi(sp[j]));
                }
                if (valid[b])
                {
                    *reinterpret_cast<uint4*>(a_lds + r[b] * lds_stride + v[b] * 16) = x[b];
                    if constexpr (WBITS == 4)
                    {
                        const uint32_t* xd = reinterpret_cast<const uint32_t*>(&x[b]);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            part += T16<T>::to_float((uint16_t) (xd[j] & 0xffffu)) + T16<T>::to_float((uint16_t) (xd[j] >> 16));
                    }
                }
                if constexpr (WBITS == 4)
                {
                    part += __shfl_xor(part, 1);
                    part += __shfl_xor(part, 2);
                    part += __shfl_xor(part, 4);
                    if constexpr (VPU == 16)
                        part += __shfl_xor(part, 8);
                    if (valid[b] && (v[b] & (VPU - 1)) == 0)
                        s1_lds[(v[b] / VPU) * (16 * MC) + r[b]] = part;
                }
            }
        }
        if constexpr (WBITS == 4) // rows past m: never stored, keep them finite
            for (int i = tid; i < units * 16 * MC; i += NWAVES * 64)
                if ((i & (16 * MC - 1)) >= rows)
                    s1_lds[i] = 0.f;
    }
    __syncthreads();
    const uint8_t* a_lane[MC];
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
        a_lane[mc] = a_lds + min(16 * mc + (lane & 15), rows - 1) * lds_stride + (kh - k_lo) * 2;
    auto afrag = [&](int mc, int kt, int st) -> uint4
    { return *reinterpret_cast<const uint4*>(a_lane[mc] + (kt * KTILE + st * 32) * 2); };
    // the lane's 4 output rows (C/D map: row = 4 * (lane >> 4) + reg) of S1 for scale unit h of tile kt
    const float* s1_lane = s1_lds + (lane >> 4) * 4 - seg_begin * NSZ * 16 * MC;
    auto s1frag = [&](int mc, int kt, int h) -> f32x4_t
    { return *reinterpret_cast<const f32x4_t*>(s1_lane + ((kt * NSZ + h) * MC + mc) * 16); };

    f32x4_t acc[MC];
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
        acc[mc] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // end of an n-block: (half tile,) cross-wave reduce, bias, store
    int it = 0; // n-blocks this block has finished (the same sequence in every wave)
    auto flush = [&](int nb)
    {
        if constexpr (WPN)
            nb = nb * NWAVES + wave;
        if (WPN && nb >= nblocks_all)
            return;
        if (half_tile && (WPN || wave == 0))
        {
            // int4 with K % 128 == 64: the run ends with a 512-byte half tile (2 k-steps)
            const int col = nb * 16 + (lane & 15);
            const uint8_t* run = a.qweight + (size_t) nb * run_bytes;
            const uint2 hw = *reinterpret_cast<const uint2*>(run + (size_t) a.ktiles * 1024 + lane * 8);
            const size_t si = (size_t) ((a.ktiles * KTILE) >> a.group_shift) * a.n + col;
            const float s = T16<T>::to_float(a.scales[si]);
            const float z = ZERO ? T16<T>::to_float(a.zeros[si]) : 0.f;
            const uint4 w = make_uint4(hw.x, hw.y, 0, 0);
            if constexpr (WBITS == 4)
                gemv_tile_int4<T, 1, 2, MC>(
                    w, acc, &s, &z, [&](int mc, int st) { return afrag(mc, a.ktiles, st); },
                    [&](int mc, int) { return s1frag(mc, a.ktiles, 0); });
        }
        if constexpr (WPN)
        {
            // this wave owns the n-block: C/D map col = lane & 15, row = (lane >> 4) * 4 + reg
            const int on = lane & 15;
            const float bias = (a.ksplit == 1 && a.bias) ? T16<T>::to_float(a.bias[nb * 16 + on]) : 0.f;
            const float alpha = (a.ksplit == 1 && a.alpha) ? a.alpha[0] : 1.f;
#pragma unroll
            for (int mc = 0; mc < MC; ++mc)
            {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                {
                    const int om = 16 * mc + (lane >> 4) * 4 + r;
                    if (om < rows)
                    {
                        const size_t oi = (size_t) (m0 + om) * a.n + nb * 16 + on;
                        if (a.ksplit == 1)
                            a.out[oi] = T16<T>::from_float(acc[mc][r] * alpha + bias);
                        else
                            a.slabs[(size_t) blockIdx.z * a.m * a.n + oi] = acc[mc][r];
                    }
                }
                acc[mc] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            return;
        }
        // RB == 2: this n-block's buffer was last read two n-blocks ago, and every reader has passed the barrier of the
        // n-block in between -- one barrier per n-block.  (Tried: no barrier at all -- every wave counts itself into
        // the slot with an LDS atomic and the last arriver sums the 16 partials.  Slower, 27.5 us against 24.9 at
        // 28672 x 8192: the reduction then always lands on the wave that is already last.)
        float* red_nb = red;
        if (RB == 2)
        {
            red_nb = red + (it & 1) * RED_FLOATS;
            ++it;
        }
        else
            __syncthreads(); // the previous n-block's readers are done with `red`
#pragma unroll
        for (int mc = 0; mc < MC; ++mc)
        {
            *reinterpret_cast<f32x4_t*>(red_nb + ((mc * NWAVES + wave) * 64 + lane) * 4) = acc[mc];
            acc[mc] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        // MC x (16 x 16) outputs: C/D map col = lane & 15, row = (lane >> 4) * 4 + reg
        if (tid < 256 * MC)
        {
            const int mc = tid >> 8, om16 = (tid >> 4) & 15, on = tid & 15;
            const int om = 16 * mc + om16;
            const int src_lane = (om16 >> 2) * 16 + on, reg = om16 & 3;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w)
                v += red_nb[((mc * NWAVES + w) * 64 + src_lane) * 4 + reg];
            if (om < rows)
            {
                const size_t oi = (size_t) (m0 + om) * a.n + nb * 16 + on;
                if (a.ksplit == 1)
                {
                    if (a.alpha)
                        v *= a.alpha[0];
                    if (a.bias)
                        v += T16<T>::to_float(a.bias[nb * 16 + on]);
                    a.out[oi] = T16<T>::from_float(v);
                }
                else
                    a.slabs[(size_t) blockIdx.z * a.m * a.n + oi] = v;
            }
        }
    };

    // ring of NBUF register images: NBUF - 1 chunks are in flight while one is computed (a wave that holds a single
    // chunk in flight during its compute phase runs at one chunk per max(compute, HBM latency under load))
    int c_nb = nb_first, c_ck = 0; // chunk being computed (held by ring[0] at the loop head)
    // one ring position; the positions are expanded statically (fold below) so every image stays in registers
    auto step = [&](auto pos) -> bool
    {
        constexpr int b = decltype(pos)::value;
        if (c_nb >= nblocks)
            return true;
        load_next(ring[(b + NBUF - 1) % NBUF]);
        gemv_compute_chunk<T, WBITS, CH, NSZ, MC>(ring[b], acc, kt_begin + c_ck * CH, kt_end, kt_hi, afrag, s1frag);
        if (++c_ck == cpw)
        {
            flush(c_nb);
            c_ck = 0;
            c_nb += nb_stride;
        }
        return false;
    };
    while (!gemv_ring_round(step, std::make_integer_sequence<int, NBUF>{}))
    {
    }
}

template <typename T>
__global__ void woq_splitk_reduce_kernel(
    const float* slabs, const uint16_t* bias, const float* alpha, uint16_t* out, int m, int n, int ksplit)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t) m * n)
        return;
    float v = 0.f;
    for (int s = 0; s < ksplit; ++s)
        v += slabs[(size_t) s * m * n + i];
    if (alpha)
        v *= alpha[0];
    if (bias)
        v += T16<T>::to_float(bias[i % n]);
    out[i] = T16<T>::from_float(v);
}

template <typename T, int WBITS, int ST>
__device__ __forceinline__ void dequant_store_step(
    const uint4& w, const WoqArgs& a, bool zero, int k0, int col, int lane)
{
    const size_t si = (size_t) (k0 >> a.group_shift) * a.n + col;
    const float s = T16<T>::to_float(a.scales[si]);
    const float z = zero ? T16<T>::to_float(a.zeros[si]) : 0.f;
    const uint4 f = dequant_step<T, WBITS, ST>(w, s, z);
    const uint32_t* p = reinterpret_cast<const uint32_t*>(&f);
#pragma unroll
    for (int j = 0; j < 8; ++j)
    {
        const int k = k0 + (lane >> 4) * 8 + j;
        a.out[(size_t) k * a.n + col] = (uint16_t) (p[j >> 1] >> (16 * (j & 1)));
    }
}

// Dense [k, n] dequantised weights from the GFX950 layout (test / tooling path; the reference's
// tests recover them with an identity matmul, test_weight_only_quant_matmul.py:44-53).
// grid: x = n-block, y = groups of 4 k-tiles (incl. the optional half tile); one wave per tile.
template <typename T, int WBITS, bool ZERO>
__global__ void woq_dequant_kernel(WoqArgs a)
{
    constexpr int KTILE = Geo<WBITS>::KTILE;
    const int lane = threadIdx.x & 63;
    const int kt = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nb = blockIdx.x;
    const bool half = WBITS == 4 && (a.k & 127) && kt == a.ktiles;
    if (kt > a.ktiles || (kt == a.ktiles && !half))
        return;
    const uint8_t* run = a.qweight + (size_t) nb * a.k * 16 * WBITS / 8;
    uint4 w;
    if (half)
    {
        const uint2 hw = *reinterpret_cast<const uint2*>(run + (size_t) kt * 1024 + lane * 8);
        w = make_uint4(hw.x, hw.y, 0, 0);
    }
    else
        w = *reinterpret_cast<const uint4*>(run + (size_t) kt * 1024 + lane * 16);
    const int col = nb * 16 + (lane & 15);
    const int k0 = kt * KTILE;
    dequant_store_step<T, WBITS, 0>(w, a, ZERO, k0, col, lane);
    dequant_store_step<T, WBITS, 1>(w, a, ZERO, k0 + 32, col, lane);
    if constexpr (WBITS == 4)
    {
        if (!half)
        {
            dequant_store_step<T, 4, 2>(w, a, ZERO, k0 + 64, col, lane);
            dequant_store_step<T, 4, 3>(w, a, ZERO, k0 + 96, col, lane);
        }
    }
}

// FP8: the W4A8 form (apply_per_channel_scale<T_in, __nv_fp8_e4m3>, preQuantScaleKernel.cu:26-79): the T-rounded
// product is converted to OCP e4m3 (round to nearest even, saturating); stored here as the same value in T (every
// e4m3 value is exact in fp16 and bf16), which is what the GEMM kernels read.  scale may be null then (no smoothing).
template <typename T, bool FP8 = false>
__global__ void per_channel_scale_kernel(uint16_t* out, const uint16_t* act, const uint16_t* scale, int rows, int cols)
{
    // 8 elements (16 B) per thread
    const size_t i = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= (size_t) rows * cols)
        return;
    const int c = (int) (i % cols);
    const uint4 av = *reinterpret_cast<const uint4*>(act + i);
    const uint32_t one2 = pack2<T>(1.f, 1.f);
    const uint4 sv = scale ? *reinterpret_cast<const uint4*>(scale + c) : make_uint4(one2, one2, one2, one2);
    const uint32_t* ap = reinterpret_cast<const uint32_t*>(&av);
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(&sv);
    uint4 r;
    uint32_t* rp = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
        rp[j] = pack2<T>(T16<T>::lo(ap[j]) * T16<T>::lo(sp[j]), T16<T>::hi(ap[j]) * T16<T>::hi(sp[j]));
        if constexpr (FP8)
        {
            const float lo = fminf(fmaxf(T16<T>::lo(rp[j]), -448.f), 448.f), hi = fminf(fmaxf(T16<T>::hi(rp[j]), -448.f), 448.f);
            const int q = __builtin_amdgcn_cvt_pk_fp8_f32(lo, hi, 0, false);
            rp[j] = pack2<T>(__builtin_amdgcn_cvt_f32_fp8(q, 0), __builtin_amdgcn_cvt_f32_fp8(q, 1));
        }
    }
    *reinterpret_cast<uint4*>(out + i) = r;
}

WoqArgs make_args(const mi355q_weight_only_params& p)
{
    MQ_CHECK(p.act_type == MI355Q_HALF || p.act_type == MI355Q_BF16, "Unsupported data type %d", p.act_type);
    MQ_CHECK(p.weight_type == MI355Q_WEIGHT_INT4 || p.weight_type == MI355Q_WEIGHT_INT8, "weight_type %d", p.weight_type);
    MQ_CHECK(p.qweight && p.scales && p.out, "null tensor");
    MQ_CHECK(p.n > 0 && p.k > 0 && p.m >= 0, "bad m/n/k");
    MQ_CHECK(p.k % 64 == 0 && p.n % 16 == 0, "k must be a multiple of 64 and n of 16 (k=%d n=%d)", p.k, p.n);
    const int group = p.group_size > 0 ? p.group_size : p.k;
    // fpA_intB_gemm_template.h:136-191: fine-grained => group size 64 or 128 (any multiple of 64 is
    // accepted here, like the SM90 path)
    MQ_CHECK(group % 64 == 0 && p.k % group == 0, "group_size %d must be a multiple of 64 dividing k=%d", group, p.k);
    WoqArgs a{};
    a.qweight = reinterpret_cast<const uint8_t*>(p.qweight);
    a.scales = reinterpret_cast<const uint16_t*>(p.scales);
    a.zeros = reinterpret_cast<const uint16_t*>(p.zeros);
    a.act = reinterpret_cast<const uint16_t*>(p.in);
    a.bias = reinterpret_cast<const uint16_t*>(p.bias);
    a.alpha = p.alpha;
    a.out = reinterpret_cast<uint16_t*>(p.out);
    a.m = p.m;
    a.n = p.n;
    a.k = p.k;
    a.group = group;
    if (p.group_size > 0)
    {
        MQ_CHECK((group & (group - 1)) == 0, "group_size %d must be a power of two", group);
        a.group_shift = 0;
        while ((1 << a.group_shift) < group)
            ++a.group_shift;
    }
    else
        a.group_shift = 31;
    a.ktiles = p.k / (p.weight_type == MI355Q_WEIGHT_INT4 ? 128 : 64);
    a.ksplit = 1;
    return a;
}

// Launch geometry of the decode kernel (see woq_gemv_kernel).
struct GemvPlan
{
    int ksplit, nwaves, ngroups, mc, red_buffers;
    bool wpn; // wave-per-n-block mode (see woq_gemv_kernel)
    size_t lds_bytes;
};

// Two 16-row chunks per block at most (wave-per-n-block mode, see the kernel): the block's activation slab
// [rows][k segment] must fit LDS, and with four chunks the k segments get so short that the fp32 split-K slabs and
// their reduce cost more than the prefill GEMM (m = 64 at 28672 x 8192: 51 + 13 us against 54 + 7 us).
constexpr int GEMV_MAX_M = 32;

int gemv_mc(int m)
{
    return m > 16 ? 2 : 1;
}

constexpr int GEMV_CUS = 256;
constexpr size_t GEMV_LDS_MAX = 160 * 1024 - 512;

size_t gemv_lds_bytes(int m, int k, int ktile, int ktiles, int ksplit, int nwaves, int red_buffers)
{
    const int seg = (ktiles + std::max(1, ksplit) - 1) / std::max(1, ksplit);
    const int krange = seg * ktile + (k - ktiles * ktile); // + the int4 half tile
    const int mc = gemv_mc(m);
    return (size_t) red_buffers * mc * nwaves * 64 * 4 * sizeof(float) + (size_t) gemv_s1_floats(krange, mc) * sizeof(float)
        + (size_t) std::min(16 * mc, m) * (krange * 2 + 32);
}

GemvPlan plan_gemv(int m, int n, int k, int wbits)
{
    const int ktile = wbits == 4 ? 128 : 64;
    const int ktiles = k / ktile;
    GemvPlan p{};
    p.mc = gemv_mc(m);
    const int nblocks = n / 16, mchunks = std::max(1, (m + 16 * p.mc - 1) / (16 * p.mc));
    p.nwaves = 16; // one 16-wave block per CU: all of a CU's waves finish an n-block together
    // k segments: (a) the activation slab of a segment must fit LDS, (b) when the n-blocks alone leave
    // most CUs idle, split K across blocks (fp32 slabs + a reduce launch, so only if really idle)
    static const int lab_wpn = lab_env("MI355Q_GEMV_WPN") ? atoi(lab_env("MI355Q_GEMV_WPN")) : -1; // lab: 0 / 1 force
    const int sets = (nblocks + p.nwaves - 1) / p.nwaves;
    // wave-per-n-block mode: two 16-row chunks (whose slab forces short k segments anyway) and enough sets of 16
    // n-blocks to give every CU a few
    p.wpn = lab_wpn >= 0 ? ((lab_wpn == 1 && p.mc >= 2) || lab_wpn == 2) : p.mc >= 2;
    if (p.wpn)
    {
        p.red_buffers = 0;
        int ks = 1;
        while (ks < ktiles && gemv_lds_bytes(m, k, ktile, ktiles, ks, p.nwaves, 0) > GEMV_LDS_MAX)
            ++ks;
        while (sets * mchunks * ks < GEMV_CUS && ktiles / (ks + 1) >= 4)
            ++ks;
        const int seg = std::max(1, (ktiles + ks - 1) / ks);
        p.ksplit = std::max(1, (ktiles + seg - 1) / seg);
        const int slots = std::max(1, GEMV_CUS / (mchunks * p.ksplit));
        const int iters = (sets + slots - 1) / slots;
        p.ngroups = (sets + iters - 1) / iters;
        p.lds_bytes = gemv_lds_bytes(m, k, ktile, ktiles, p.ksplit, p.nwaves, 0);
        MQ_CHECK(p.lds_bytes <= GEMV_LDS_MAX, "k=%d too large for the decode kernel's LDS activation slab", k);
        return p;
    }
    // two reduce buffers (one barrier per n-block) unless a single one saves a k split, i.e. the fp32 slabs and the
    // reduce launch (m = 16 at k = 4096: 128 KiB of rows + 32 KiB of buffers do not fit, + 16 KiB do)
    auto min_split = [&](int rb)
    {
        int s = 1;
        while (s < ktiles && gemv_lds_bytes(m, k, ktile, ktiles, s, p.nwaves, rb) > GEMV_LDS_MAX)
            ++s;
        return s;
    };
    p.red_buffers = (p.mc == 1 && min_split(2) == min_split(1)) ? 2 : 1;
    int ks = min_split(p.red_buffers);
    const int wgs = nblocks * mchunks;
    if (wgs < GEMV_CUS / 2)
    {
        int fill = (GEMV_CUS + wgs - 1) / wgs;
        fill = std::min(fill, std::max(1, ktiles / 16)); // keep >= 16 tiles per block
        ks = std::max(ks, std::min(fill, 16));
    }
    const int seg = std::max(1, (ktiles + ks - 1) / ks);
    p.ksplit = std::max(1, (ktiles + seg - 1) / seg); // no empty trailing segment
    // n-groups: one block per CU, n-blocks dealt round-robin, every block the same count (+-1)
    const int slots = std::max(1, GEMV_CUS / (mchunks * p.ksplit));
    const int iters = (nblocks + slots - 1) / slots;
    p.ngroups = (nblocks + iters - 1) / iters;
    p.lds_bytes = gemv_lds_bytes(m, k, ktile, ktiles, p.ksplit, p.nwaves, p.red_buffers);
    MQ_CHECK(p.lds_bytes <= GEMV_LDS_MAX, "k=%d too large for the decode kernel's LDS activation slab", k);
    return p;
}

template <typename T, int WBITS, bool ZERO, int NWAVES, int CH, int MC, int NBUF, bool WPN = false>
void launch_gemv_variant(const WoqArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
    // two (scale, zero) pairs per tile only when a 128-row int4 tile spans two groups of 64
    const bool split = WBITS == 4 && a.group == 64;
    auto kern = split ? woq_gemv_kernel<T, WBITS, ZERO, NWAVES, CH, (WBITS == 4 ? 2 : 1), MC, NBUF, WPN>
                      : woq_gemv_kernel<T, WBITS, ZERO, NWAVES, CH, 1, MC, NBUF, WPN>;
    if (lds_bytes > 64 * 1024)
    {
        MQ_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) GEMV_LDS_MAX));
    }
    hipLaunchKernelGGL(kern, grid, dim3(NWAVES * 64), lds_bytes, stream, a);
}

template <typename T, int WBITS, bool ZERO>
void launch_gemv(const WoqArgs& a, hipStream_t stream)
{
    const GemvPlan p = plan_gemv(a.m, a.n, a.k, WBITS);
    MQ_CHECK(p.ksplit == a.ksplit || a.ksplit == 1, "internal: plan mismatch");
    GemvPlan q = p;
    if (a.ksplit != p.ksplit) // no workspace for the slabs: one segment (must still fit LDS)
    {
        q.ksplit = 1;
        q.red_buffers = 1;
        q.wpn = false;
        q.ngroups = std::min(p.wpn ? GEMV_CUS : p.ngroups, a.n / 16);
        q.lds_bytes = gemv_lds_bytes(a.m, a.k, Geo<WBITS>::KTILE, a.ktiles, 1, p.nwaves, 1);
        MQ_CHECK(q.lds_bytes <= GEMV_LDS_MAX,
            "this m/k needs the split-K workspace (mi355q_weight_only_gemm_workspace_size)");
    }
    WoqArgs ka = a;
    ka.red_buffers = q.red_buffers;
    dim3 grid(q.ngroups, (a.m + 16 * q.mc - 1) / (16 * q.mc), q.ksplit);
    // ring geometry: 3 register images of 2 tiles (measured best of {2x4, 3x4, 4x2, 6x2, 3x2, 4x1, 6x1, 8x1} at
    // 28672 x 8192 and no worse elsewhere); MI355Q_GEMV_RING = 10 * images + tiles selects the lab alternatives
    static const int lab = lab_env("MI355Q_GEMV_RING") ? atoi(lab_env("MI355Q_GEMV_RING")) : 0;
    if (q.wpn && q.mc == 2)
        launch_gemv_variant<T, WBITS, ZERO, 16, 2, 2, 3, true>(ka, grid, q.lds_bytes, stream);
    else if (q.wpn)
        launch_gemv_variant<T, WBITS, ZERO, 16, 2, 1, 3, true>(ka, grid, q.lds_bytes, stream);
    else if (q.mc == 2 && lab == 32)
        launch_gemv_variant<T, WBITS, ZERO, 16, 2, 2, 3>(ka, grid, q.lds_bytes, stream);
    else if (q.mc == 2)
        launch_gemv_variant<T, WBITS, ZERO, 16, 2, 2, 2>(ka, grid, q.lds_bytes, stream);
    else if (lab == 24)
        launch_gemv_variant<T, WBITS, ZERO, 16, 4, 1, 2>(ka, grid, q.lds_bytes, stream);
    else if (lab == 41)
        launch_gemv_variant<T, WBITS, ZERO, 16, 1, 1, 4>(ka, grid, q.lds_bytes, stream);
    else
        launch_gemv_variant<T, WBITS, ZERO, 16, 2, 1, 3>(ka, grid, q.lds_bytes, stream);
    if (a.ksplit > 1)
    {
        const size_t total = (size_t) a.m * a.n;
        hipLaunchKernelGGL((woq_splitk_reduce_kernel<T>), dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, stream,
            a.slabs, a.bias, a.alpha, a.out, a.m, a.n, a.ksplit);
    }
}

template <typename T, int WBITS, bool ZERO>
void launch_dequant(const WoqArgs& a, hipStream_t stream)
{
    dim3 grid(a.n / 16, (a.ktiles + 1 + 3) / 4);
    hipLaunchKernelGGL((woq_dequant_kernel<T, WBITS, ZERO>), grid, dim3(256), 0, stream, a);
}

#define MQ_DISPATCH_WOQ(FN, p, a, stream)                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        const bool bf = (p).act_type == MI355Q_BF16, i4 = (p).weight_type == MI355Q_WEIGHT_INT4;                       \
        const bool z = (p).zeros != nullptr;                                                                           \
        if (bf && i4 && z) FN<bf16_t, 4, true>(a, stream);                                                             \
        else if (bf && i4) FN<bf16_t, 4, false>(a, stream);                                                            \
        else if (bf && z) FN<bf16_t, 8, true>(a, stream);                                                              \
        else if (bf) FN<bf16_t, 8, false>(a, stream);                                                                  \
        else if (i4 && z) FN<f16_t, 4, true>(a, stream);                                                               \
        else if (i4) FN<f16_t, 4, false>(a, stream);                                                                   \
        else if (z) FN<f16_t, 8, true>(a, stream);                                                                     \
        else FN<f16_t, 8, false>(a, stream);                                                                           \
    } while (0)

} // namespace

int woq_choose_ksplit(int m, int n, int k, int weight_type)
{
    return plan_gemv(m, n, k, weight_type == MI355Q_WEIGHT_INT4 ? 4 : 8).ksplit;
}

size_t woq_gemm_workspace_size(int m, int n, int k, int weight_type)
{
    // worst case over all m' <= m of ksplit(m') * m' * n fp32 slabs
    size_t best = 0;
    for (int mm = 1; mm <= m; mm = (mm < 64 || m - mm < 16) ? mm + 1 : mm + 16) // every m' up to 64, then 16s, then m
        for (int pre = 0; pre < 2; ++pre) // the route can depend on whether the call carries a pre-quant scale
        {
            const bool decode = woq_gemm_fuses_act_scale(mm, n, k, weight_type, pre != 0);
            if (decode)
            {
                const LightPlan lp = woq_light_plan(mm, n, k, weight_type);
                if (lp.applies)
                {
                    best = std::max(best, lp.workspace_bytes);
                    continue;
                }
            }
            else
                best = std::max(best, woq_skinny_workspace_size(mm, n, k, weight_type));
            const int s = decode ? woq_choose_ksplit(mm, n, k, weight_type) : woq_prefill_ksplit(mm, n, k, weight_type);
            if (s > 1)
                best = std::max(best, (size_t) s * mm * n * sizeof(float));
        }
    return best;
}

void woq_splitk_reduce(const float* slabs, const void* bias, const float* alpha, void* out, int m, int n, int ksplit,
    bool bf16, hipStream_t stream)
{
    const size_t total = (size_t) m * n;
    const dim3 grid((unsigned) ((total + 255) / 256));
    if (bf16)
        hipLaunchKernelGGL((woq_splitk_reduce_kernel<bf16_t>), grid, dim3(256), 0, stream, slabs,
            (const uint16_t*) bias, alpha, (uint16_t*) out, m, n, ksplit);
    else
        hipLaunchKernelGGL((woq_splitk_reduce_kernel<f16_t>), grid, dim3(256), 0, stream, slabs,
            (const uint16_t*) bias, alpha, (uint16_t*) out, m, n, ksplit);
}

// Which kernel family takes (m, n, k): true = the decode kernels (light / persistent GEMV), which apply the pre-quant
// scale themselves; false = the MFMA GEMMs over staged activation tiles (skinny form / ping-pong prefill), which want
// the activations smoothed beforehand.
//   m <= 16: decode.  m > 32: the GEMMs (where the shape allows).
//   16 < m <= 32 (tools/lab_skinny.py, MI355Q_GEMV_MAX_M=16 against 32): the persistent kernel's two 16-row chunks
//   lose 3 - 20 % to the skinny form / the ping-pong at every decode-sized weight (4096 x 4096: 13.8 -> 11.0 us,
//   22016 x 4096: 29.1 -> 23.5, 3584 x 8192: 17.6 -> 14.5) and win only beyond ~64 MB of weights (28672 x 8192: 45.5
//   against 50 us).  With a pre-quant scale the GEMMs need the extra smoothing launch, which eats the difference: those
//   calls stay on the decode kernel.
bool woq_gemm_fuses_act_scale(int m, int n, int k, int weight_type, bool has_act_scale)
{
    static const int lab_max_m = lab_env("MI355Q_GEMV_MAX_M") ? atoi(lab_env("MI355Q_GEMV_MAX_M")) : 0;
    if (!woq_prefill_supported(m, n, k) || lab_env("MI355Q_NO_PREFILL"))
        return true;
    if (lab_max_m > 0)
        return m <= lab_max_m;
    if (m <= 16)
        return true;
    if (m > GEMV_MAX_M)
        return false;
    const size_t wbytes = (size_t) n * k / (weight_type == MI355Q_WEIGHT_INT4 ? 2 : 1);
    return has_act_scale || wbytes > ((size_t) 64 << 20);
}

void weight_only_gemm(const mi355q_weight_only_params& p, void* workspace, size_t workspace_bytes, hipStream_t stream)
{
    WoqArgs a = make_args(p);
    MQ_CHECK(p.in, "null activation");
    if (p.m == 0)
        return;
    if (!woq_gemm_fuses_act_scale(p.m, p.n, p.k, p.weight_type, p.act_scale != nullptr))
    {
        MQ_CHECK(p.act_scale == nullptr, "pre-quant scale must be applied with mi355q_apply_per_channel_scale first");
        // m > 64: dequantise-in-registers MFMA GEMM (woq_prefill.hip); the decode kernel below would
        // stream the weights once per 64 rows
        const int ks = woq_prefill_ksplit(p.m, p.n, p.k, p.weight_type);
        const size_t need = (size_t) ks * p.m * p.n * sizeof(float);
        float* slabs = (ks > 1 && workspace && workspace_bytes >= need) ? reinterpret_cast<float*>(workspace) : nullptr;
        WoqArgsHost h{a.qweight, a.scales, a.zeros, a.act, a.bias, a.out, a.m, a.n, a.k, a.group, a.group_shift, a.ktiles,
            p.act_type == MI355Q_BF16, p.weight_type == MI355Q_WEIGHT_INT4, slabs, a.alpha};
        if (weight_only_skinny_gemm(h, workspace, workspace_bytes, stream)) // m <= 64: one burst per workgroup
            return;
        weight_only_prefill_gemm(h, stream);
        if (slabs)
            woq_splitk_reduce(slabs, p.bias, p.alpha, p.out, p.m, p.n, ks, p.act_type == MI355Q_BF16, stream);
        MQ_HIP(hipGetLastError());
        return;
    }
    {
        // m <= 16 at decode-sized weights: the light kernel (woq_gemv_small.hip), one launch whatever the k split
        const LightPlan lp = woq_light_plan(p.m, p.n, p.k, p.weight_type);
        if (lp.applies)
        {
            weight_only_gemv_light(p, lp, workspace, workspace_bytes, stream);
            return;
        }
    }
    a.act_scale = static_cast<const uint16_t*>(p.act_scale); // decode kernel: applied while the rows are staged
    a.ksplit = woq_choose_ksplit(p.m, p.n, p.k, p.weight_type);
    if (a.ksplit > 1)
    {
        const size_t need = (size_t) a.ksplit * p.m * p.n * sizeof(float);
        if (workspace && workspace_bytes >= need)
            a.slabs = reinterpret_cast<float*>(workspace);
        else
            a.ksplit = 1; // correct without workspace, just fewer workgroups
    }
    MQ_DISPATCH_WOQ(launch_gemv, p, a, stream);
    MQ_HIP(hipGetLastError());
}

void weight_only_dequantize(const mi355q_weight_only_params& p, hipStream_t stream)
{
    WoqArgs a = make_args(p);
    MQ_DISPATCH_WOQ(launch_dequant, p, a, stream);
    MQ_HIP(hipGetLastError());
}

void apply_per_channel_scale_fp8(void* out, const void* act, const void* scale, int rows, int cols, int act_type, hipStream_t stream)
{
    MQ_CHECK(out && act, "null tensor");
    MQ_CHECK(act_type == MI355Q_HALF || act_type == MI355Q_BF16, "Unsupported data type %d", act_type);
    MQ_CHECK(cols % 8 == 0, "cols must be a multiple of 8 (preQuantScaleKernel.cu:86-89), got %d", cols);
    if (rows == 0)
        return;
    const size_t vecs = (size_t) rows * cols / 8;
    dim3 grid((unsigned) ((vecs + 255) / 256));
    if (act_type == MI355Q_BF16)
        hipLaunchKernelGGL((per_channel_scale_kernel<bf16_t, true>), grid, dim3(256), 0, stream, (uint16_t*) out,
            (const uint16_t*) act, (const uint16_t*) scale, rows, cols);
    else
        hipLaunchKernelGGL((per_channel_scale_kernel<f16_t, true>), grid, dim3(256), 0, stream, (uint16_t*) out,
            (const uint16_t*) act, (const uint16_t*) scale, rows, cols);
    MQ_HIP(hipGetLastError());
}

void apply_per_channel_scale(void* out, const void* act, const void* scale, int rows, int cols, int act_type, hipStream_t stream)
{
    MQ_CHECK(out && act && scale, "null tensor");
    MQ_CHECK(act_type == MI355Q_HALF || act_type == MI355Q_BF16, "Unsupported data type %d", act_type);
    MQ_CHECK(cols % 8 == 0, "cols must be a multiple of 8 (preQuantScaleKernel.cu:86-89), got %d", cols);
    if (rows == 0)
        return;
    const size_t vecs = (size_t) rows * cols / 8;
    dim3 grid((unsigned) ((vecs + 255) / 256));
    if (act_type == MI355Q_BF16)
        hipLaunchKernelGGL((per_channel_scale_kernel<bf16_t>), grid, dim3(256), 0, stream, (uint16_t*) out,
            (const uint16_t*) act, (const uint16_t*) scale, rows, cols);
    else
        hipLaunchKernelGGL((per_channel_scale_kernel<f16_t>), grid, dim3(256), 0, stream, (uint16_t*) out,
            (const uint16_t*) act, (const uint16_t*) scale, rows, cols);
    MQ_HIP(hipGetLastError());
}

} // namespace mi355q
