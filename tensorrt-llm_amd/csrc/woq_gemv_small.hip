// woq_gemv_small.hip -- the decode GEMV for m <= 16: light workgroups, nothing staged, one launch.
// Replaces kernels/weightOnlyBatchedGemv/kernelLauncher.cu:122-232 + kernel.h:294-468 of the reference (its m < 5
// CUDA-core GEMV) at the shapes a decode step really has (N = K = 4096 ... 8192 x 3584: 4-25 MB of weights, a
// 1-4 us HBM floor), where the persistent kernel of woq_gemm.hip is bound by its own fixed costs.
//
// What bounds a launch this small (tools/lab_skel.hip, 8 MiB of weights, 256 workgroups): an empty launch in a
// stream of launches is 1.65 us; streaming the weights, an LDS reduce and the store make it 3.0 us; and every
// vector-memory instruction a wave issues -- live, out of range or against an empty descriptor alike -- holds the
// CU's address / return path for ~16 cycles: 28 loads per wave (the first form of this kernel: 4 weight, 8 scale /
// zero, 16 activation-fragment loads) cost 1.5 us per launch with no byte moved.  So the design counts instructions:
//   * workgroup = 16 waves = ONE n-block (16 columns) x one k segment; wave w takes tiles (round * 16 + w) * CH ..
//     of the segment; every load of a round is issued up front, branch-free (tiles past the segment read through
//     zero-sized descriptors), NBUF = 2 register images prefetch the next round under the current one's arithmetic.
//     cfg1 (4096 x 4096 int4) is one round of two tiles per wave;
//   * weights: one 16-byte load per lane and tile (the GFX950 layout's B fragments of a whole k-tile);
//   * scales / zeros: ONE 2-byte load each per round -- lane group h fetches the pair of slot h (tile, 64-row half) and
//     the lanes read their tile's pair back with ds_bpermute;
//   * activations, m <= 8 (MF 0 / 1): v_mfma_f32_4x4x4_16B with A broadcast inside each group of four blocks
//     (cbsz = 2, abid = q): lane 16 h + 4 q + i holds row i's 8 values of k-step q, k-group h, so ONE 16-byte load per
//     lane is the A operand of all 8 MFMAs of a tile (4 k-steps x 2 halves; 4 loads per tile in the 16x16x32 form),
//     rows >= m read as zero through the descriptor's range check; a lane's 4 accumulators are rows 0..3 of its
//     column over its own k-group, the 4 k-groups are summed by two ds_bpermute exchanges at the very end;
//     m <= 16 (MF 2): v_mfma_f32_16x16x32, 4 fragment loads per tile;
//   * arithmetic: with every byte requested in the first microsecond the data of all waves lands together and the
//     SIMDs' instruction issue is what is left -- the exact dequantisation (23 VALU operations per 8 int4 weights) was
//     0.9 us of a 5.4 us launch.  int4 therefore runs in the INTEGER domain like the persistent kernel (light_group);
//     int8 keeps the exact w = round16(fma(q, scale, zero)) (kernel.h:364-365);
//   * 16 partial tiles are summed through LDS (one barrier), bias added, rows < m stored;
//   * split-K (few n-blocks: the TP slices 8192 -> 1280 ...) stays inside the launch: every segment's workgroup
//     publishes its fp32 tile write-through, counts itself on the n-block's arrival word and the workgroup whose
//     arrival completes the count sums the tiles in segment order (deterministic) and stores -- no slab reduce launch.
//     Hand-off per MI355X_MICROARCH.md (Valid forms, row "one lane of each storing workgroup ... the workgroup whose
//     add came last"): sc1 payload stores, every storing wave drains vmcnt, workgroup barrier, ONE lane's agent-scope
//     read-modify-write, barrier, sc1 loads.  The arrival words are arrival_ring.hip's (zero at launch, one per 128-byte
//     line, put back to zero by the last arrival's own wrapping increment).
#include "woq_common.h"
#include <algorithm>
#include <atomic>

namespace mi355q
{

namespace
{


struct LightArgs
{
    WoqArgs a;
    int seg_tiles;      // k-tiles per segment
    int rounds;         // chunk rounds per workgroup: ceil(seg_tiles / (LW * CH))
    float* records;     // [n-block][segment][16 x 16] fp32 partial tiles (split-K only)
    uint32_t* arrivals; // one arrival word per n-block (split-K only): arrival_ring.hip, ARRIVAL_WORD_STRIDE apart
    int abl;            // lab build only (MI355Q_GEMV_LIGHT_ABL): ablation bits, see the kernel
};

// MF: MFMA form.  0: 4x4x4, rows 0..3; 1: 4x4x4, two passes (rows 0..7); 2: 16x16x32 (rows 0..15)
template <int WBITS, int MF>
struct LightGeo
{
    static constexpr int KS = Geo<WBITS>::KSTEPS;
    static constexpr int AV = MF == 2 ? KS : MF + 1; // 16-byte activation loads per tile
    static constexpr int PV = MF == 2 ? KS : 1;      // 16-byte pre-quant-scale loads per tile
};

template <int WBITS, int CH, int NSZ, bool PRE, int MF>
struct LightImage
{
    static constexpr int SZL = (CH * NSZ + 3) / 4; // (scale, zero) loads per round: 4 slots each
    uint4 w[CH];
    uint32_t s[SZL], z[SZL];
    uint4 act[CH][LightGeo<WBITS, MF>::AV];
    uint4 pre[PRE ? CH : 1][LightGeo<WBITS, MF>::PV];
};

template <typename T>
__device__ __forceinline__ uint4 scale_round(const uint4& x, const uint4& s)
{
    // AWQ pre-quant scale: act * scale rounded to T, the values preQuantScaleKernel.cu would have written
    const uint32_t* xp = reinterpret_cast<const uint32_t*>(&x);
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(&s);
    uint4 r;
    uint32_t* rp = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        rp[j] = pack2<T>(T16<T>::lo(xp[j]) * T16<T>::lo(sp[j]), T16<T>::hi(xp[j]) * T16<T>::hi(sp[j]));
    return r;
}

// v_mfma_f32_4x4x4_16B: 16 blocks of (4 x 4) += (4 x 4)(4 x 4); with cbsz = 2 the A block `Q` of every group of four
// blocks (16 lanes) is broadcast to the group's four blocks.  Lane 4 b + j: A = row j's 4 k-values, B = column j's 4
// k-values, D[0..3] = rows 0..3 of column j of block b.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
template <typename T, int Q>
__device__ __forceinline__ f32x4_t mfma4(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, f32x4_t c)
{
    const uint2 a = make_uint2(a0, a1), b = make_uint2(b0, b1);
    if constexpr (T16<T>::is_bf16)
        return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), c, 2, Q, 0);
    else
        return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4_t, a), __builtin_bit_cast(f16x4_t, b), c, 2, Q, 0);
}

// one MFMA step of the lane's operands: A = activations (act[ps] halves for the 4x4x4 form, act[ST] for 16x16x32),
// B = `wf` (8 values of the lane's column)
template <typename T, int MF, int ST>
__device__ __forceinline__ void light_mma(const uint4& wf, const uint4* act, f32x4_t* acc)
{
    if constexpr (MF == 2)
        acc[0] = mfma16<T>(act[ST], wf, acc[0]);
    else
    {
#pragma unroll
        for (int ps = 0; ps <= MF; ++ps)
        {
            acc[ps] = mfma4<T, ST>(act[ps].x, act[ps].y, wf.x, wf.y, acc[ps]);
            acc[ps] = mfma4<T, ST>(act[ps].z, act[ps].w, wf.z, wf.w, acc[ps]);
        }
    }
}

// k-steps [ST, ST + N) of one tile for ONE (scale, zero) pair.
//   int8: exact dequantisation w = round16(fma(q, s, z)) of the lane's 8 weights, then the MFMA(s);
//   int4: the integer domain of woq_gemm.hip (gemv_tile_int4): P = sum a * u on the MFMA with u = q + 8 unpacked exactly
//         through the fp8 converter (7 VALU operations per 8 weights instead of 23), S1 = sum a through the same MFMA
//         against ones, then acc += s * (P - 8 S1) + z * S1 == sum a * (q s + z): the weight is not rounded to 16 bits
//         before the multiply (results within half an output ulp of the rounded-weight result; a one-hot activation
//         row gives fma(q, s, z) exactly, see gemv_tile_int4).
template <typename T, int WBITS, int MF, int ST, int N>
__device__ __forceinline__ void light_group(const uint4& w, float s, float z, const uint4* act, f32x4_t* acc)
{
    if constexpr (WBITS == 8)
    {
        light_mma<T, MF, ST>(dequant_step<T, 8, ST>(w, s, z), act, acc);
        if constexpr (N == 2)
            light_mma<T, MF, ST + 1>(dequant_step<T, 8, ST + 1>(w, s, z), act, acc);
    }
    else
    {
        const uint32_t* wd = reinterpret_cast<const uint32_t*>(&w);
        f32x4_t P[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, S1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const uint4 ones = ones8<T>();
        light_mma<T, MF, ST>(unpack8_u4<T>(wd[ST]), act, P);
        light_mma<T, MF, ST>(ones, act, S1);
        if constexpr (N >= 2)
        {
            light_mma<T, MF, ST + 1>(unpack8_u4<T>(wd[ST + 1]), act, P);
            light_mma<T, MF, ST + 1>(ones, act, S1);
        }
        if constexpr (N == 4)
        {
            light_mma<T, MF, ST + 2>(unpack8_u4<T>(wd[ST + 2]), act, P);
            light_mma<T, MF, ST + 2>(ones, act, S1);
            light_mma<T, MF, ST + 3>(unpack8_u4<T>(wd[ST + 3]), act, P);
            light_mma<T, MF, ST + 3>(ones, act, S1);
        }
#pragma unroll
        for (int ps = 0; ps < (MF == 1 ? 2 : 1); ++ps)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[ps][r] = fmaf(s, fmaf(-8.f, S1[ps][r], P[ps][r]), fmaf(z, S1[ps][r], acc[ps][r]));
    }
}

// one tile (KSU k-steps: the full tile, or 2 for the int4 half tile); NS (scale, zero) pairs over the KSU k-steps
template <typename T, int WBITS, int MF, bool PRE, int NS, int KSU>
__device__ __forceinline__ void light_tile(const uint4& w, const float* s, const float* z, const uint4* act_in,
    const uint4* pre, f32x4_t* acc)
{
    uint4 act[4];
#pragma unroll
    for (int i = 0; i < LightGeo<WBITS, MF>::AV; ++i)
        act[i] = PRE ? scale_round<T>(act_in[i], pre[MF == 2 ? i : 0]) : act_in[i];
    if constexpr (NS == 1)
        light_group<T, WBITS, MF, 0, KSU>(w, s[0], z[0], act, acc);
    else
    {
        light_group<T, WBITS, MF, 0, KSU / 2>(w, s[0], z[0], act, acc);
        light_group<T, WBITS, MF, KSU / 2, KSU / 2>(w, s[1], z[1], act, acc);
    }
}

typedef __attribute__((address_space(1))) uint32_t gu32;
typedef __attribute__((address_space(1))) float gf32;

// LW: waves per workgroup (16: one workgroup per CU at N = K = 4096; 8: two can share a CU at the larger shapes)
template <typename T, int WBITS, int NSZ, int CH, int NBUF, bool PRE, int MF, int LW>
__global__ __launch_bounds__(LW * 64) void woq_gemv_light_kernel(LightArgs p)
{
    constexpr int KTILE = Geo<WBITS>::KTILE;
    constexpr int KS = Geo<WBITS>::KSTEPS;
    constexpr int AV = LightGeo<WBITS, MF>::AV;
    constexpr int ROWS = MF == 2 ? 16 : 4 * (MF + 1);
    const WoqArgs& a = p.a;
    __shared__ float red[MF == 2 ? LW * 256 : LW * ROWS * 16];
    __shared__ int last_flag;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MI355Q_LAB
    // ablation (timing only, results wrong): 1 no arithmetic, 2 no cross-wave reduce, 4 / 8 / 16 activation / scale+zero
    // / weight loads against zero-sized descriptors (issued, no memory traffic), 32 exit at once, 64 exit before the loads
    const int abl = p.abl;
    if (abl & 32)
        return;
#else
    constexpr int abl = 0;
#endif
    // workgroups b and b + 8 share an XCD (and its L2): give each XCD a contiguous range of units, so the four
    // n-blocks that share a 128-byte line of scales / zeros fetch it into one L2 only
    int unit;
    {
        const int total = gridDim.x, q = total / 8, r = total % 8, x = blockIdx.x % 8;
        unit = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + blockIdx.x / 8;
    }
    const int ks = unit % a.ksplit;
    const int nb = unit / a.ksplit;
    const int seg_begin = ks * p.seg_tiles;
    const int seg_end = min(a.ktiles, seg_begin + p.seg_tiles);

    const size_t run_bytes = (size_t) a.k * 16 * WBITS / 8;
    const uint8_t* run = a.qweight + (size_t) nb * run_bytes;
    const uint32_t w_limit = (uint32_t) seg_end * 1024u;
    const uint32_t sz_bytes = (uint32_t) ((size_t) (a.k / a.group) * a.n * 2);
    const uint32_t act_bytes = (uint32_t) ((size_t) a.m * a.k * 2);
    const uint32_t w_voff = lane * 16;
    const uint32_t row_bytes = a.n * 2;
    const int hg = lane >> 4; // k-group of the lane inside a k-step (rows 8 hg .. 8 hg + 7)
    constexpr uint32_t DEAD = 0x7fffffffu; // beyond every descriptor used here
    // activation / pre-quant-scale fragment offsets of the lane (bytes; without the tile's k offset)
    uint32_t a_voff[MF == 1 ? 2 : 1], p_voff;
    if constexpr (MF == 2)
    {
        a_voff[0] = ((lane & 15) * a.k + hg * 8) * 2; // row lane & 15, one load per k-step
        p_voff = hg * 16;
    }
    else
    {
        const int i = lane & 3, q = (lane >> 2) & 3; // row i (+ 4 per pass) of k-step q
        const uint32_t koff = (32 * q + 8 * hg) * 2;
#pragma unroll
        for (int ps = 0; ps <= MF; ++ps)
            a_voff[ps] = q < KS ? (uint32_t) (4 * ps + i) * a.k * 2 + koff : DEAD;
        p_voff = q < KS ? koff : DEAD;
    }

    using Image = LightImage<WBITS, CH, NSZ, PRE, MF>;
    auto issue = [&](Image& im, int round)
    {
        const int kt0 = seg_begin + (round * LW + wave) * CH;
#pragma unroll
        for (int i = 0; i < CH; ++i)
        {
            const int kt = kt0 + i;
            const bool live = kt < seg_end; // wave-uniform
            const __amdgpu_buffer_rsrc_t wr = make_rsrc(run, (live && !(abl & 16)) ? w_limit : 0);
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wr, w_voff, kt * 1024, 2 /* nt: streamed once */);
            im.w[i] = make_uint4(v.x, v.y, v.z, v.w);
            const __amdgpu_buffer_rsrc_t ar = make_rsrc(a.act, (live && !(abl & 4)) ? act_bytes : 0);
            const __amdgpu_buffer_rsrc_t pr
                = make_rsrc(PRE ? a.act_scale : a.act, (PRE && live && !(abl & 4)) ? (uint32_t) a.k * 2 : 0);
            if constexpr (MF == 2)
            {
#pragma unroll
                for (int st = 0; st < KS; ++st)
                {
                    const uint32_t koff = (uint32_t) (kt * KTILE + st * 32) * 2;
                    const u32x4_t x = __builtin_amdgcn_raw_buffer_load_b128(ar, a_voff[0], koff, 0);
                    im.act[i][st] = make_uint4(x.x, x.y, x.z, x.w);
                    if constexpr (PRE)
                    {
                        const u32x4_t y = __builtin_amdgcn_raw_buffer_load_b128(pr, p_voff, koff, 0);
                        im.pre[i][st] = make_uint4(y.x, y.y, y.z, y.w);
                    }
                }
            }
            else
            {
                const uint32_t koff = (uint32_t) (kt * KTILE) * 2;
#pragma unroll
                for (int ps = 0; ps <= MF; ++ps)
                {
                    const u32x4_t x = __builtin_amdgcn_raw_buffer_load_b128(ar, a_voff[ps], koff, 0);
                    im.act[i][ps] = make_uint4(x.x, x.y, x.z, x.w);
                }
                if constexpr (PRE)
                {
                    const u32x4_t y = __builtin_amdgcn_raw_buffer_load_b128(pr, p_voff, koff, 0);
                    im.pre[i][0] = make_uint4(y.x, y.y, y.z, y.w);
                }
            }
        }
        // (scale, zero) pairs of the round: slot j = tile * NSZ + half; lane group hg fetches slots hg, hg + 4, ...
        const __amdgpu_buffer_rsrc_t sr = make_rsrc(a.scales, (abl & 8) ? 0 : sz_bytes);
        const __amdgpu_buffer_rsrc_t zr = make_rsrc(a.zeros ? a.zeros : a.scales, (a.zeros && !(abl & 8)) ? sz_bytes : 0);
#pragma unroll
        for (int l = 0; l < Image::SZL; ++l)
        {
            const int slot = 4 * l + hg;
            const int kt = kt0 + slot / NSZ;
            const bool ok = slot < CH * NSZ && kt < seg_end;
            const uint32_t voff = ok
                ? (uint32_t) ((kt * KTILE + (slot % NSZ) * 64) >> a.group_shift) * row_bytes + (nb * 16 + (lane & 15)) * 2
                : DEAD;
            im.s[l] = __builtin_amdgcn_raw_buffer_load_b16(sr, voff, 0, 0);
            im.z[l] = __builtin_amdgcn_raw_buffer_load_b16(zr, voff, 0, 0);
        }
    };

    f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto compute = [&](const Image& im)
    {
        if (abl & 1)
        {
#pragma unroll
            for (int i = 0; i < CH; ++i)
            {
                uint32_t x = im.w[i].x ^ im.w[i].y ^ im.w[i].z ^ im.w[i].w ^ im.s[0] ^ im.z[0];
#pragma unroll
                for (int st = 0; st < AV; ++st)
                    x ^= im.act[i][st].x ^ im.act[i][st].y ^ im.act[i][st].z ^ im.act[i][st].w;
                acc[0][0] += __uint_as_float(x & 0x3fffffffu);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < CH; ++i)
        {
            float s[NSZ], z[NSZ];
#pragma unroll
            for (int h = 0; h < NSZ; ++h)
            {
                const int slot = i * NSZ + h;
                const int src = 16 * (slot & 3) + (lane & 15);
                s[h] = T16<T>::to_float((uint32_t) __shfl((int) im.s[slot >> 2], src) & 0xffffu);
                z[h] = T16<T>::to_float((uint32_t) __shfl((int) im.z[slot >> 2], src) & 0xffffu);
            }
            light_tile<T, WBITS, MF, PRE, NSZ, KS>(im.w[i], s, z, im.act[i], im.pre[PRE ? i : 0], acc);
        }
    };

    // sched_barrier: left alone, hipcc's scheduler trades the burst for registers and interleaves the loads with the
    // arithmetic (9-11 loads in flight instead of a whole round)
    Image img[NBUF];
    if (abl & 64)
    {
        if (w_voff + a_voff[0] + p_voff + row_bytes + w_limit + sz_bytes + act_bytes == 0x12345u)
            a.out[0] = 1;
        return;
    }
    if constexpr (NBUF == 1)
    {
        for (int r = 0; r < p.rounds; ++r) // one image: no prefetch across rounds (one round at the decode shapes)
        {
            issue(img[0], r);
            __builtin_amdgcn_sched_barrier(0);
            compute(img[0]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    else
    {
        // two images in rotation; the issue of round r + 1 is unconditional (past the last round every tile is dead:
        // zero-sized descriptors, no memory traffic), so no path forces vmcnt(0) on the round computed next
        issue(img[0], 0);
        __builtin_amdgcn_sched_barrier(0);
        for (int r = 0; r < p.rounds; r += 2)
        {
            issue(img[1], r + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(img[0]);
            __builtin_amdgcn_sched_barrier(0);
            issue(img[0], r + 2);
            __builtin_amdgcn_sched_barrier(0);
            compute(img[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // int4 with K % 128 == 64: the run ends with a 512-byte half tile (2 k-steps); last segment, wave 0
    if (WBITS == 4 && (a.k & 127) && ks == a.ksplit - 1 && wave == 0)
    {
        const int col = nb * 16 + (lane & 15);
        const int kbase = a.ktiles * KTILE;
        const uint2 hw = *reinterpret_cast<const uint2*>(run + (size_t) a.ktiles * 1024 + lane * 8);
        const size_t si = (size_t) (kbase >> a.group_shift) * a.n + col;
        const float s = T16<T>::to_float(a.scales[si]);
        const float z = a.zeros ? T16<T>::to_float(a.zeros[si]) : 0.f;
        const uint4 w = make_uint4(hw.x, hw.y, 0, 0);
        uint4 av[4], pv[4];
        if constexpr (MF == 2)
        {
#pragma unroll
            for (int st = 0; st < 2; ++st)
            {
                const int k0 = kbase + st * 32 + hg * 8, row = lane & 15;
                av[st] = row < a.m ? *reinterpret_cast<const uint4*>(a.act + (size_t) row * a.k + k0) : make_uint4(0, 0, 0, 0);
                if constexpr (PRE)
                    pv[st] = *reinterpret_cast<const uint4*>(a.act_scale + k0);
            }
        }
        else
        {
            const int i = lane & 3, q = (lane >> 2) & 3;
            const int k0 = kbase + 32 * q + 8 * hg;
#pragma unroll
            for (int ps = 0; ps <= MF; ++ps)
                av[ps] = (q < 2 && 4 * ps + i < a.m) ? *reinterpret_cast<const uint4*>(a.act + (size_t) (4 * ps + i) * a.k + k0)
                                                     : make_uint4(0, 0, 0, 0);
            if constexpr (PRE)
                pv[0] = q < 2 ? *reinterpret_cast<const uint4*>(a.act_scale + k0) : make_uint4(0, 0, 0, 0);
        }
        if constexpr (WBITS == 4)
            light_tile<T, 4, MF, PRE, 1, 2>(w, &s, &z, av, pv, acc);
    }

    // ---- LW partial tiles -> one
    float v = 0.f;
    int om, on;
    bool mine;
    if (abl & 2)
    {
        if ((lane & 15) < a.m && wave == 0)
            a.out[(size_t) (lane & 15) * a.n + nb * 16 + (lane >> 2)] = T16<T>::from_float(acc[0][0] + acc[0][1] + acc[1][2]);
        return;
    }
    if constexpr (MF == 2)
    {
        // C/D map: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
        for (int r = 0; r < 4; ++r)
            red[(wave * 4 + r) * 64 + lane] = acc[0][r];
        __syncthreads();
        const int t_lane = tid & 63, t_reg = (tid >> 6) & 3;
        om = 4 * (t_lane >> 4) + t_reg, on = t_lane & 15;
        mine = tid < 256 && om < a.m;
        if (tid < 256)
#pragma unroll
            for (int w = 0; w < LW; ++w)
                v += red[(w * 4 + t_reg) * 64 + t_lane];
    }
    else
    {
        // a lane holds rows 0..3 (per pass) of column lane & 15 over its k-group: sum the 4 k-groups first -- all the
        // exchanges of a step issued together (written one value at a time hipcc waits for each ds_bpermute in turn)
        float x[2][4];
#pragma unroll
        for (int ps = 0; ps <= MF; ++ps)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                x[ps][r] = __shfl_xor(acc[ps][r], 16);
#pragma unroll
        for (int ps = 0; ps <= MF; ++ps)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[ps][r] += x[ps][r];
#pragma unroll
        for (int ps = 0; ps <= MF; ++ps)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                x[ps][r] = __shfl_xor(acc[ps][r], 32);
        if (lane < 16)
        {
#pragma unroll
            for (int ps = 0; ps <= MF; ++ps)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[(wave * ROWS + 4 * ps + r) * 16 + lane] = acc[ps][r] + x[ps][r];
        }
        __syncthreads();
        om = tid >> 4, on = tid & 15;
        mine = tid < ROWS * 16 && om < a.m;
        if (tid < ROWS * 16)
#pragma unroll
            for (int w = 0; w < LW; ++w)
                v += red[(w * ROWS + om) * 16 + on];
    }
    const size_t oi = (size_t) om * a.n + nb * 16 + on;
    if (a.ksplit == 1)
    {
        if (mine)
        {
            if (a.alpha)
                v *= a.alpha[0];
            if (a.bias)
                v += T16<T>::to_float(a.bias[nb * 16 + on]);
            a.out[oi] = T16<T>::from_float(v);
        }
        return;
    }

    // ---- split-K inside the launch (header comment): publish, count, the completing workgroup sums
    gf32* rec = (gf32*) (p.records + ((size_t) nb * a.ksplit + ks) * 256);
    if (mine)
        __hip_atomic_store(rec + om * 16 + on, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // sc1: write-through
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains before the barrier
    __syncthreads();
    if (tid == 0)
    {
        // wrapping increment: the arrival that finds ksplit - 1 is the last and leaves the word at zero again
        gu32* word = (gu32*) (p.arrivals + (size_t) nb * ARRIVAL_WORD_STRIDE);
        last_flag = __builtin_amdgcn_atomic_inc32(word, (uint32_t) (a.ksplit - 1), __ATOMIC_RELAXED, "agent")
            == (uint32_t) (a.ksplit - 1);
    }
    __syncthreads();
    if (!last_flag)
        return;
    if (mine)
    {
        // sc1 loads, 8 segments in flight; summed in segment order: the same sum whichever workgroup arrives last
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(p.records + (size_t) nb * a.ksplit * 256, (uint32_t) a.ksplit * 1024);
        const uint32_t off = (uint32_t) (om * 16 + on) * 4;
        float sum = 0.f;
        for (int s0 = 0; s0 < a.ksplit; s0 += 8)
        {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) // past the last segment: beyond the descriptor, reads 0
                x[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, off + (uint32_t) (s0 + j) * 1024, 0, 16));
#pragma unroll
            for (int j = 0; j < 8; ++j)
                sum += x[j];
        }
        if (a.alpha)
            sum *= a.alpha[0];
        if (a.bias)
            sum += T16<T>::to_float(a.bias[nb * 16 + on]);
        a.out[oi] = T16<T>::from_float(sum);
    }
}

template <typename T, int WBITS, int NSZ, int CH, int NBUF, int MF, int LW>
void launch_variant(LightArgs& la, int grid, hipStream_t stream)
{
    la.rounds = (la.seg_tiles + LW * CH - 1) / (LW * CH);
    if (la.a.act_scale)
    {
        if constexpr (MF == 2 && !(CH == 2 && NBUF == 1)) // the 16-row form with pre-quant scale fragments: registers
        {                                                   // for one image of two tiles, rounds without prefetch
            la.rounds = (la.seg_tiles + LW * 2 - 1) / (LW * 2);
            hipLaunchKernelGGL((woq_gemv_light_kernel<T, WBITS, NSZ, 2, 1, true, 2, LW>), dim3(grid), dim3(LW * 64), 0, stream, la);
        }
        else
            hipLaunchKernelGGL((woq_gemv_light_kernel<T, WBITS, NSZ, CH, NBUF, true, MF, LW>), dim3(grid), dim3(LW * 64), 0, stream, la);
    }
    else
        hipLaunchKernelGGL((woq_gemv_light_kernel<T, WBITS, NSZ, CH, NBUF, false, MF, LW>), dim3(grid), dim3(LW * 64), 0, stream, la);
}

template <typename T, int WBITS, int NSZ, int MF, int LW>
void launch_mf(LightArgs& la, int grid, hipStream_t stream)
{
    const int per_wave = (la.seg_tiles + LW - 1) / LW;
    if (per_wave <= 2)
        launch_variant<T, WBITS, NSZ, 2, 1, MF, LW>(la, grid, stream);
    else if constexpr (MF >= 1) // two images do not fit the 128 registers that let 16 waves share a CU:
        launch_variant<T, WBITS, NSZ, 4, 1, MF, LW>(la, grid, stream); // rounds of one 4-tile image without prefetch
    else if (per_wave <= 4)
        launch_variant<T, WBITS, NSZ, 4, 1, MF, LW>(la, grid, stream);
    else
        launch_variant<T, WBITS, NSZ, 2, 2, MF, LW>(la, grid, stream);
}

template <typename T, int WBITS, int NSZ>
void launch_shape(LightArgs& la, int grid, hipStream_t stream)
{
    // 16 waves when one workgroup per CU covers the launch and every wave still gets two tiles (cfg1: 4.7 against
    // 4.9 us); 8 waves otherwise, so that two workgroups share a CU (12288 x 4096: 7.8 against 8.9 us)
    const bool wide = grid <= 320 && la.seg_tiles >= 24 && la.seg_tiles <= 32; // longer segments: 4096 x 11008 7.9 (8 waves) against 8.9 us
    if (la.a.m <= 4)
        wide ? launch_mf<T, WBITS, NSZ, 0, 16>(la, grid, stream) : launch_mf<T, WBITS, NSZ, 0, 8>(la, grid, stream);
    else if (la.a.m <= 8)
        wide ? launch_mf<T, WBITS, NSZ, 1, 16>(la, grid, stream) : launch_mf<T, WBITS, NSZ, 1, 8>(la, grid, stream);
    else
        wide ? launch_mf<T, WBITS, NSZ, 2, 16>(la, grid, stream) : launch_mf<T, WBITS, NSZ, 2, 8>(la, grid, stream);
}

} // namespace

// Geometry of the light kernel for (m, n, k): does it apply, how K is cut.
LightPlan woq_light_plan(int m, int n, int k, int weight_type)
{
    LightPlan pl{};
    const int ktile = weight_type == MI355Q_WEIGHT_INT4 ? 128 : 64;
    const int ktiles = k / ktile, nblocks = n / 16;
    static const int lab = lab_env("MI355Q_GEMV_LIGHT") ? atoi(lab_env("MI355Q_GEMV_LIGHT")) : -1; // lab: 0 off, 1 always
    const size_t wbytes = (size_t) n * k * (weight_type == MI355Q_WEIGHT_INT4 ? 1 : 2) / 2;
    // where it wins (tools/lab_gemv_light.sh, tools/lab_splitk.py): every n-block count that fills the chip without a k
    // split, and fewer n-blocks when K can be cut into segments of >= 16 tiles (1280 x 8192: 5.4 us against the
    // persistent kernel's 7.9; 8-tile segments lose); m > 8 streams 4 activation fragments per tile and loses beyond ~24 MB
    pl.applies = m >= 1 && m <= 16 && ktiles >= 1 && (size_t) 16 * k * 2 < (1ull << 30)
        && (nblocks >= 192 || ktiles >= 32)
        && wbytes <= (m <= 8 ? LIGHT_MAX_WEIGHT_BYTES : LIGHT_MAX_WEIGHT_BYTES_16ROWS);
    if (lab == 0)
        pl.applies = false;
    else if (lab == 1)
        pl.applies = m >= 1 && m <= 16 && ktiles >= 1;
    if (!pl.applies)
        return pl;
    // split K only when the n-blocks alone leave most of the chip idle: towards 512 workgroups, never below 16 tiles
    // per workgroup (2560 x 8192: 2 / 4 / 8 segments take 7.1 / 6.5 / 9.1 us)
    int ks = 1;
    if (nblocks < 192)
        ks = std::max(1, std::min({(512 + nblocks - 1) / nblocks, ktiles / 16, 32}));
    static const int lab_ks = lab_env("MI355Q_GEMV_LIGHT_KS") ? atoi(lab_env("MI355Q_GEMV_LIGHT_KS")) : 0;
    if (lab_ks > 0)
        ks = std::min(lab_ks, ktiles);
    pl.seg_tiles = (ktiles + ks - 1) / ks;
    pl.ksplit = (ktiles + pl.seg_tiles - 1) / pl.seg_tiles; // no empty trailing segment
    pl.workspace_bytes = pl.ksplit > 1 ? (size_t) nblocks * pl.ksplit * 1024 : 0;
    return pl;
}

void weight_only_gemv_light(const mi355q_weight_only_params& p, LightPlan pl, void* workspace, size_t workspace_bytes,
    hipStream_t stream)
{
    MQ_CHECK(pl.applies, "internal: light GEMV plan does not apply");
    const bool int4 = p.weight_type == MI355Q_WEIGHT_INT4;
    const int group = p.group_size > 0 ? p.group_size : p.k;
    LightArgs la{};
    WoqArgs& a = la.a;
    a.qweight = reinterpret_cast<const uint8_t*>(p.qweight);
    a.scales = reinterpret_cast<const uint16_t*>(p.scales);
    a.zeros = reinterpret_cast<const uint16_t*>(p.zeros);
    a.act = reinterpret_cast<const uint16_t*>(p.in);
    a.act_scale = reinterpret_cast<const uint16_t*>(p.act_scale);
    a.bias = reinterpret_cast<const uint16_t*>(p.bias);
    a.alpha = p.alpha;
    a.out = reinterpret_cast<uint16_t*>(p.out);
    a.m = p.m, a.n = p.n, a.k = p.k;
    a.group = group;
    a.group_shift = 31;
    if (p.group_size > 0)
    {
        a.group_shift = 0;
        while ((1 << a.group_shift) < group)
            ++a.group_shift;
    }
    a.ktiles = p.k / (int4 ? 128 : 64);
    const int nblocks = p.n / 16;
    if (pl.ksplit > 1 && !(workspace && workspace_bytes >= pl.workspace_bytes && ((uintptr_t) workspace & 255) == 0))
    {
        // no room for the partial tiles: one segment per n-block (correct, fewer workgroups)
        pl.ksplit = 1;
        pl.seg_tiles = a.ktiles;
    }
    a.ksplit = pl.ksplit;
    la.seg_tiles = pl.seg_tiles;
    if (pl.ksplit > 1)
    {
        la.records = reinterpret_cast<float*>(workspace);
        la.arrivals = arrival_lines((size_t) nblocks, stream);
    }
    static const int lab_abl = lab_env("MI355Q_GEMV_LIGHT_ABL") ? atoi(lab_env("MI355Q_GEMV_LIGHT_ABL")) : 0;
    la.abl = lab_abl;
    const int grid = nblocks * pl.ksplit;
    const bool bf = p.act_type == MI355Q_BF16;
    const bool split = int4 && group == 64; // a 128-row int4 tile spans two groups
    if (bf && int4 && split) launch_shape<bf16_t, 4, 2>(la, grid, stream);
    else if (bf && int4) launch_shape<bf16_t, 4, 1>(la, grid, stream);
    else if (bf) launch_shape<bf16_t, 8, 1>(la, grid, stream);
    else if (int4 && split) launch_shape<f16_t, 4, 2>(la, grid, stream);
    else if (int4) launch_shape<f16_t, 4, 1>(la, grid, stream);
    else launch_shape<f16_t, 8, 1>(la, grid, stream);
    MQ_HIP(hipGetLastError());
}

} // namespace mi355q
