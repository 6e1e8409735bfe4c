// woq_gemm.hip -- weight-only quantized GEMV / GEMM: fp16|bf16 activations x int4|int8 weights
// (group-wise AWQ/GPTQ scale(+zero) or per-channel scale).  Replaces
//   kernels/weightOnlyBatchedGemv/{kernel.h,kernelLauncher.cu}  (m < 5, CUDA cores) and
//   kernels/cutlass_kernels/fpA_intB_gemm/*                      (m >= 5, dequant-in-mainloop)
// of the reference with one gfx950 design:
//   * weights live in the GFX950 layout (preprocess.cpp): 1 KiB tiles of 16 columns x 128 (int4) or
//     64 (int8) rows; one 16-byte load per lane is already the B fragment of 4 (2) consecutive
//     v_mfma_f32_16x16x32 k-steps -- no LDS, no shuffles;
//   * dequantisation in registers with the reference's single-rounding semantic
//     w = round16(fma(q, scale, zero)) (kernel.h:364-365, mma_tensorop_dequantizer.h:244-252):
//     v_cvt_f32_i32 (SDWA sign-extending byte select) + v_pk_fma_f32 + v_cvt_pk_{bf16,f16}_f32;
//     the same device function backs mi355q_weight_only_dequantize, i.e. the bit-exact gate checks
//     exactly the values the MFMA consumes;
//   * decode (m <= 16 per block row): one workgroup = one 16-column n-block, 4 waves split K in
//     chunks of 4 tiles; every wave issues a whole chunk (4 KiB of weights + activations + scales)
//     as one burst of branch-free loads and keeps the next chunk in flight while it dequantises;
//     fp32 MFMA accumulators are reduced through LDS.  The launch is HBM/VALU co-limited: the
//     dequant costs ~2.8 VALU instructions per weight, so all 256 CUs must take part (N/16 blocks);
//   * optional split-K across workgroups (fp32 slabs in the workspace + a tiny reduce kernel) when
//     N/16 workgroups cannot fill the chip.
#include "kernels.h"
#include <cstdlib>

namespace mi355q
{

namespace
{

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int GEMV_CH = 4; // k-tiles per chunk

struct WoqArgs
{
    const uint8_t* qweight;
    const uint16_t* scales;
    const uint16_t* zeros;
    const uint16_t* act;
    const uint16_t* bias;
    uint16_t* out;
    float* slabs; // [ksplit][m][n] fp32 partials when ksplit > 1
    int m, n, k;
    int group;  // group size in k; == k for per-channel
    int group_shift; // log2(group); 31 for per-channel (k >> 31 == 0)
    int ktiles; // full k-tiles: k / KTILE
    int ksplit;
};

template <typename T>
__device__ __forceinline__ f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c);
template <>
__device__ __forceinline__ f32x4_t mfma16<bf16_t>(const uint4& a, const uint4& b, f32x4_t c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(
        __builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mfma16<f16_t>(const uint4& a, const uint4& b, f32x4_t c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(
        __builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ float sbyte(uint32_t w, int i)
{
    return (float) (int8_t) (w >> (8 * i)); // v_cvt_f32_i32_sdwa sext(BYTE_i)
}

// 8 signed int4 (one dword, nibble j = element j) -> 8 dequantised 16-bit values (4 packed dwords).
// lo/hi bytes carry 16*q exactly, so the fma uses scale/16 (a power-of-two rescale: exact).
template <typename T>
__device__ __forceinline__ uint4 dequant8_int4(uint32_t w, float s16, float z)
{
    const uint32_t lo = (w << 4) & 0xf0f0f0f0u; // byte i = 16 * q[2i]
    const uint32_t hi = w & 0xf0f0f0f0u;        // byte i = 16 * q[2i+1]
    uint4 r;
    r.x = pack2<T>(fmaf(sbyte(lo, 0), s16, z), fmaf(sbyte(hi, 0), s16, z));
    r.y = pack2<T>(fmaf(sbyte(lo, 1), s16, z), fmaf(sbyte(hi, 1), s16, z));
    r.z = pack2<T>(fmaf(sbyte(lo, 2), s16, z), fmaf(sbyte(hi, 2), s16, z));
    r.w = pack2<T>(fmaf(sbyte(lo, 3), s16, z), fmaf(sbyte(hi, 3), s16, z));
    return r;
}

// 8 signed int8 (two dwords) -> 8 dequantised 16-bit values
template <typename T>
__device__ __forceinline__ uint4 dequant8_int8(uint32_t w0, uint32_t w1, float s, float z)
{
    uint4 r;
    r.x = pack2<T>(fmaf(sbyte(w0, 0), s, z), fmaf(sbyte(w0, 1), s, z));
    r.y = pack2<T>(fmaf(sbyte(w0, 2), s, z), fmaf(sbyte(w0, 3), s, z));
    r.z = pack2<T>(fmaf(sbyte(w1, 0), s, z), fmaf(sbyte(w1, 1), s, z));
    r.w = pack2<T>(fmaf(sbyte(w1, 2), s, z), fmaf(sbyte(w1, 3), s, z));
    return r;
}

template <int WBITS>
struct Geo
{
    static constexpr int KTILE = WBITS == 4 ? 128 : 64;
    static constexpr int KSTEPS = KTILE / 32;      // MFMA k-steps per tile (4 or 2)
    static constexpr int NSZ = WBITS == 4 ? 2 : 1; // (scale, zero) pairs a tile may span (group 64 on int4)
};

// k-step ST of the lane fragment `w` -> dequantised B operand.  s / z: this k-step's group.
template <typename T, int WBITS, int ST>
__device__ __forceinline__ uint4 dequant_step(const uint4& w, float s, float z)
{
    const uint32_t* wd = reinterpret_cast<const uint32_t*>(&w);
    if constexpr (WBITS == 4)
        return dequant8_int4<T>(wd[ST], s * 0.0625f, z);
    else
        return dequant8_int8<T>(wd[ST * 2], wd[ST * 2 + 1], s, z);
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// SRD over `bytes` bytes at p (wave-uniform inputs only, so hipcc never wraps the loads in a
// waterfall loop); out-of-range offsets return 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes)
{
    const uint64_t u = (uint64_t) p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t) (u >> 32));
    const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t) (bytes > 0xffffffffull ? 0xffffffffull : bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void*) (((uint64_t) hi << 32) | lo), 0, nb, 0x00020000);
}

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

// `afrag(kt, st)` returns the activation fragment of k-step st of tile kt (LDS or global).
template <typename T, int WBITS, int CH, int NSZ, typename AF>
__device__ __forceinline__ void gemv_compute_chunk(
    const GemvChunk<CH, NSZ>& c, f32x4_t& acc, int kt0, int kt_end, int kt_last, AF afrag)
{
#pragma unroll
    for (int i = 0; i < CH; ++i)
    {
        const bool live = kt0 + i < kt_end; // wave-uniform; a dead tile gets scale = zero = 0 -> W = 0
        const int kt = min(kt0 + i, kt_last);
        float s[2], z[2];
#pragma unroll
        for (int h = 0; h < NSZ; ++h)
        {
            s[h] = live ? T16<T>::to_float(c.s[i][h]) : 0.f;
            z[h] = live ? T16<T>::to_float(c.z[i][h]) : 0.f;
        }
        if constexpr (WBITS == 4)
        {
            constexpr int H1 = NSZ - 1;
            acc = gemv_step<T, 4, 0>(c.w[i], afrag(kt, 0), s[0], z[0], acc);
            acc = gemv_step<T, 4, 1>(c.w[i], afrag(kt, 1), s[0], z[0], acc);
            acc = gemv_step<T, 4, 2>(c.w[i], afrag(kt, 2), s[H1], z[H1], acc);
            acc = gemv_step<T, 4, 3>(c.w[i], afrag(kt, 3), s[H1], z[H1], acc);
        }
        else
        {
            acc = gemv_step<T, 8, 0>(c.w[i], afrag(kt, 0), s[0], z[0], acc);
            acc = gemv_step<T, 8, 1>(c.w[i], afrag(kt, 1), s[0], z[0], acc);
        }
    }
}

// grid: x = n-block (16 columns), y = 16-row chunk of m, z = k split.  block = NWAVES waves; wave w
// owns the chunks w, w+NWAVES, ... (CH k-tiles each) of the block's k range (ping-pong register
// sets).  LDS_A: the block's activation rows [rows][k range] are staged once into LDS (row stride
// k*2+32 bytes: conflict-free ds_read_b128 fragments) so the per-tile activation traffic leaves
// the vector-memory path, which is then used by the weight stream alone.
template <typename T, int WBITS, bool ZERO, int NWAVES, bool LDS_A, int NSZ>
__global__ __launch_bounds__(NWAVES * 64) void woq_gemv_kernel(WoqArgs a)
{
    constexpr int CH = GEMV_CH;
    constexpr int KTILE = Geo<WBITS>::KTILE;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[]; // red[NWAVES][64][4] floats, then A rows
    float* red = reinterpret_cast<float*>(smem);
    uint8_t* a_lds = smem + NWAVES * 64 * 4 * sizeof(float);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = blockIdx.x;
    const int m0 = blockIdx.y * 16;
    const int col = nb * 16 + (lane & 15);
    const int kh = (lane >> 4) * 8;
    const int rows = min(16, a.m - m0);

    const int per = (a.ktiles + a.ksplit - 1) / a.ksplit;
    const int kt_begin = blockIdx.z * per;
    const int kt_end = min(a.ktiles, kt_begin + per);
    const int kt_last = max(kt_end - 1, 0);
    const bool half_tile = WBITS == 4 && (a.k & 127) && blockIdx.z == a.ksplit - 1;
    // k range [k_lo, k_hi) this block touches (incl. the half tile)
    const int k_lo = kt_begin * KTILE;
    const int k_hi = half_tile ? a.k : kt_end * KTILE;

    const size_t run_bytes = (size_t) a.k * 16 * WBITS / 8;
    const uint8_t* run = a.qweight + (size_t) nb * run_bytes;
    GemvSrc src;
    src.w = make_rsrc(run, run_bytes);
    const size_t sz_bytes = (size_t) (a.k / a.group) * a.n * 2;
    src.s = make_rsrc(a.scales, sz_bytes);
    src.z = make_rsrc(ZERO ? a.zeros : a.scales, sz_bytes);
    src.w_voff = lane * 16;
    src.sz_voff = col * 2;
    src.group_shift = a.group_shift;
    src.row_bytes = a.n * 2;
    // rows >= m are clamped to the last valid row: their MFMA output rows are computed, never stored
    const int rloc = min(lane & 15, rows - 1);
    const uint16_t* arow = a.act + (size_t) (m0 + rloc) * a.k + kh;
    const int lds_stride = (k_hi - k_lo) * 2 + 32;

    constexpr int STRIDE = NWAVES * CH;
    GemvChunk<CH, NSZ> ca, cb;
    int kt0 = kt_begin + wave * CH;
    if (kt0 < kt_end) // weights first: their HBM latency overlaps the activation staging below
        gemv_load_chunk<WBITS, ZERO, CH, NSZ>(ca, src, kt0, kt_last);

    if constexpr (LDS_A)
    {
        const int vec_per_row = (k_hi - k_lo) / 8; // 16-byte vectors
        for (int i = tid; i < rows * vec_per_row; i += NWAVES * 64)
        {
            const int r = i / vec_per_row, v = i % vec_per_row;
            *reinterpret_cast<uint4*>(a_lds + r * lds_stride + v * 16)
                = *reinterpret_cast<const uint4*>(a.act + (size_t) (m0 + r) * a.k + k_lo + v * 8);
        }
        __syncthreads();
    }
    const uint8_t* a_lane = a_lds + rloc * lds_stride + (kh - k_lo) * 2;
    auto afrag = [&](int kt, int st) -> uint4
    {
        if constexpr (LDS_A)
            return *reinterpret_cast<const uint4*>(a_lane + (kt * KTILE + st * 32) * 2);
        else
            return *reinterpret_cast<const uint4*>(arow + (size_t) kt * KTILE + st * 32);
    };

    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (kt0 < kt_end)
    {
        while (true)
        {
            // invariant: `ca` holds chunk kt0
            const bool more_b = kt0 + STRIDE < kt_end;
            if (more_b)
                gemv_load_chunk<WBITS, ZERO, CH, NSZ>(cb, src, kt0 + STRIDE, kt_last);
            gemv_compute_chunk<T, WBITS, CH, NSZ>(ca, acc, kt0, kt_end, kt_last, afrag);
            if (!more_b)
                break;
            kt0 += STRIDE;
            const bool more_a = kt0 + STRIDE < kt_end;
            if (more_a)
                gemv_load_chunk<WBITS, ZERO, CH, NSZ>(ca, src, kt0 + STRIDE, kt_last);
            gemv_compute_chunk<T, WBITS, CH, NSZ>(cb, acc, kt0, kt_end, kt_last, afrag);
            if (!more_a)
                break;
            kt0 += STRIDE;
        }
    }

    // int4 with K % 128 == 64: the run ends with a 512-byte half tile (2 k-steps); wave 0 of the
    // last k split takes it
    if (half_tile && wave == 0)
    {
        const int kbase = a.ktiles * KTILE;
        const uint2 hw = *reinterpret_cast<const uint2*>(run + (size_t) a.ktiles * 1024 + lane * 8);
        const size_t si = (size_t) (kbase >> a.group_shift) * a.n + col;
        const float s = T16<T>::to_float(a.scales[si]);
        const float z = ZERO ? T16<T>::to_float(a.zeros[si]) : 0.f;
        const uint4 w = make_uint4(hw.x, hw.y, 0, 0);
        acc = gemv_step<T, 4, 0>(w, afrag(a.ktiles, 0), s, z, acc);
        acc = gemv_step<T, 4, 1>(w, afrag(a.ktiles, 1), s, z, acc);
    }

#pragma unroll
    for (int r = 0; r < 4; ++r)
        red[(wave * 64 + lane) * 4 + r] = acc[r];
    __syncthreads();

    // 16 x 16 outputs: C/D map col = lane & 15, row = (lane >> 4) * 4 + reg
    if (tid < 256)
    {
        const int om = tid >> 4, on = tid & 15;
        const int src_lane = (om >> 2) * 16 + on, reg = om & 3;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w)
            v += red[(w * 64 + src_lane) * 4 + reg];
        if (om < rows)
        {
            const size_t oi = (size_t) (m0 + om) * a.n + nb * 16 + on;
            if (a.ksplit == 1)
            {
                if (a.bias)
                    v += T16<T>::to_float(a.bias[nb * 16 + on]);
                a.out[oi] = T16<T>::from_float(v);
            }
            else
                a.slabs[(size_t) blockIdx.z * a.m * a.n + oi] = v;
        }
    }
}

template <typename T>
__global__ void woq_splitk_reduce_kernel(const float* slabs, const uint16_t* bias, uint16_t* out, int m, int n, int ksplit)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t) m * n)
        return;
    float v = 0.f;
    for (int s = 0; s < ksplit; ++s)
        v += slabs[(size_t) s * m * n + i];
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

template <typename T>
__global__ void per_channel_scale_kernel(uint16_t* out, const uint16_t* act, const uint16_t* scale, int rows, int cols)
{
    // 8 elements (16 B) per thread
    const size_t i = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= (size_t) rows * cols)
        return;
    const int c = (int) (i % cols);
    const uint4 av = *reinterpret_cast<const uint4*>(act + i);
    const uint4 sv = *reinterpret_cast<const uint4*>(scale + c);
    const uint32_t* ap = reinterpret_cast<const uint32_t*>(&av);
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(&sv);
    uint4 r;
    uint32_t* rp = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        rp[j] = pack2<T>(T16<T>::lo(ap[j]) * T16<T>::lo(sp[j]), T16<T>::hi(ap[j]) * T16<T>::hi(sp[j]));
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

template <typename T, int WBITS, bool ZERO, int NWAVES, bool LDS_A>
void launch_gemv_variant(const WoqArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
    // two (scale, zero) pairs per tile only when a 128-row int4 tile spans two groups of 64
    const bool split = WBITS == 4 && a.group == 64;
    auto kern = split ? woq_gemv_kernel<T, WBITS, ZERO, NWAVES, LDS_A, (WBITS == 4 ? 2 : 1)>
                      : woq_gemv_kernel<T, WBITS, ZERO, NWAVES, LDS_A, 1>;
    if (lds_bytes > 64 * 1024)
    {
        MQ_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    }
    hipLaunchKernelGGL(kern, grid, dim3(NWAVES * 64), lds_bytes, stream, a);
}

template <typename T, int WBITS, bool ZERO>
void launch_gemv(const WoqArgs& a, hipStream_t stream)
{
    dim3 grid(a.n / 16, (a.m + 15) / 16, a.ksplit);
    constexpr int KTILE = Geo<WBITS>::KTILE;
    const int per = (a.ktiles + a.ksplit - 1) / a.ksplit;
    const int krange = per * KTILE + (WBITS == 4 ? (a.k & 127) : 0);
    const size_t red_bytes = 8 * 64 * 4 * sizeof(float);
    const size_t a_bytes = (size_t) std::min(16, a.m) * (krange * 2 + 32);
    // activation rows in LDS when they fit beside a second resident block (<= 72 KiB), or, for a
    // taller slab, when one block per CU is all the grid asks for anyway
    const bool lds_a = red_bytes + a_bytes <= 72 * 1024
        || (red_bytes + a_bytes <= 150 * 1024 && grid.x * grid.y * grid.z <= 320);
    // 8 waves when a block owns enough tiles to keep them all busy
    const bool w8 = per >= 32;
    if (lds_a && w8)
        launch_gemv_variant<T, WBITS, ZERO, 8, true>(a, grid, red_bytes + a_bytes, stream);
    else if (lds_a)
        launch_gemv_variant<T, WBITS, ZERO, 4, true>(a, grid, red_bytes + a_bytes, stream);
    else if (w8)
        launch_gemv_variant<T, WBITS, ZERO, 8, false>(a, grid, red_bytes, stream);
    else
        launch_gemv_variant<T, WBITS, ZERO, 4, false>(a, grid, red_bytes, stream);
    if (a.ksplit > 1)
    {
        const size_t total = (size_t) a.m * a.n;
        hipLaunchKernelGGL((woq_splitk_reduce_kernel<T>), dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, stream,
            a.slabs, a.bias, a.out, a.m, a.n, a.ksplit);
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

// Split-K heuristic: only when the n-blocks alone leave most of the chip idle -- the reduce is a
// second launch and costs more than a half-empty grid does.
int woq_choose_ksplit(int m, int n, int k, int weight_type)
{
    const int ktiles = k / (weight_type == MI355Q_WEIGHT_INT4 ? 128 : 64);
    const int wgs = (n / 16) * ((m + 15) / 16);
    if (wgs >= 128)
        return 1;
    int s = (256 + wgs - 1) / wgs;
    s = std::min(s, std::max(1, ktiles / 16)); // keep >= 16 tiles per block
    return std::max(1, std::min(s, 16));
}

size_t woq_gemm_workspace_size(int m, int n, int k, int weight_type)
{
    // worst case over all m' <= m of ksplit(m') * m' * n fp32 slabs
    size_t best = 0;
    for (int mm = 1; mm <= m; mm = mm < 16 ? mm + 1 : mm + 16)
    {
        const int s = woq_choose_ksplit(mm, n, k, weight_type);
        if (s > 1)
            best = std::max(best, (size_t) s * mm * n * sizeof(float));
    }
    return best;
}

void weight_only_gemm(const mi355q_weight_only_params& p, void* workspace, size_t workspace_bytes, hipStream_t stream)
{
    WoqArgs a = make_args(p);
    MQ_CHECK(p.in, "null activation");
    if (p.m == 0)
        return;
    MQ_CHECK(p.act_scale == nullptr, "pre-quant scale must be applied with mi355q_apply_per_channel_scale first");
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
