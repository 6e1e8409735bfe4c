// woq_prefill.hip -- weight-only quantized GEMM for prefill-sized M (m > 16): fp16|bf16 activations x
// int4|int8 weights, dequantised in registers and fed to v_mfma_f32_16x16x32_{bf16,f16}.  Replaces
//   kernels/cutlass_kernels/fpA_intB_gemm/fpA_intB_gemm.h:54-137 (CutlassFpAIntBGemmRunner::gemm, the
//   dequant-in-mainloop tensor-op GEMM the plugins pick for m >= 5).
// Design (MFMA-bound):
//   * output tile TM (= MT*16: 32 / 64 / 128 rows of m) x TN (= NB*64: 128 / 256 columns) per workgroup
//     of 4 waves (or 256 x 256 on 8 waves for large int4 problems, template parameter NW); wave w owns the NB 16-column n-blocks NB*w .. NB*w+NB-1 of the tile and ALL its m
//     rows, so every weight is dequantised exactly once per workgroup (23 VALU ops per 8 weights) and
//     reused by MT MFMAs, and every activation fragment read from LDS feeds NB MFMAs (NB = 4 keeps the
//     LDS read pipe at half the MFMA pipe's time; NB = 2 would tie them);
//   * optional split-K over grid.y for small m (few output tiles): fp32 slabs + the reduce kernel of
//     woq_gemm.hip;
//   * weights come straight from the GFX950 layout (preprocess.cpp) into registers -- one 16-byte load
//     per lane is the fragment of a whole k-tile, no LDS, double-buffered one k-tile ahead;
//   * dequantisation is the bit-exact one: w = round16(fma(q, scale, zero)) (woq_common.h), i.e. the MFMA
//     consumes exactly the values mi355q_weight_only_dequantize returns;
//   * the activation tile [TM][k-tile] goes through an LDS double buffer filled by LDS-DMA
//     (global_load_lds_dwordx4; the XOR swizzle is applied to the per-lane source address so that the
//     16 rows x one 16-byte chunk of a ds_read_b128 fragment cover all 64 banks once).  The DMA is issued from
//     inline asm and the weight loads carry no branch, so tile kt + 1 really is in flight while tile kt is computed:
//     with the builtin DMA (and the half-tile branch inside the weight loads) hipcc put s_waitcnt vmcnt(0) in front
//     of every tile's first LDS read -- load, wait, compute, in series;
//   * the dequantised weights are the MFMA *A* operand and the activations the *B* operand, so a lane
//     ends up with 4 consecutive output columns of one row: 8-byte stores, bias fused.
#include "woq_prefill_common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace mi355q
{

namespace
{

struct PrefillPlan
{
    int mt, nb, ksplit;
};

// 256 x 256 tiles on 8 waves (one workgroup per CU) against the 128-row tiles (two per CU): the big tile runs about
// 1.15x faster per unit of work (tools/lab_pf256.py), either pays for a partly filled last round of workgroups.
bool prefer_tile256(int64_t w256, int64_t w128)
{
    if (w256 <= 0)
        return false;
    const double r256 = (double) w256 / 256.0, r128 = (double) w128 / 512.0;
    return 1.15 * r256 / std::ceil(r256) > r128 / std::ceil(r128);
}

PrefillPlan plan_prefill(int m, int n, int k, int wbits)
{
    PrefillPlan p{};
    p.mt = m <= 32 ? 2 : (m <= 64 ? 4 : 8);
    // 256 x 256 tiles on 8 waves once they alone fill the chip (one workgroup per CU); int4 only (the int8 form of that
    // tile spills registers)
    const bool lab_no256 = lab_env("MI355Q_PF_NO256") != nullptr; // (read per call: A/B in one process)
    if (wbits == 4 && m >= 256 && n % 256 == 0 && ((m + 255) / 256) * (n / 256) >= 256 && !lab_no256
        && prefer_tile256((int64_t) ((m + 255) / 256) * (n / 256), (int64_t) ((m + 127) / 128) * (n / 256)))
    {
        p.mt = 16;
        p.nb = 2;
        p.ksplit = 1;
        return p;
    }
    const int tiles_m = (m + p.mt * 16 - 1) / (p.mt * 16);
    // 256-column tiles (4 n-blocks per wave) once they alone fill the chip; else 128 columns
    // (int8 weights: the 4-n-block instantiation spills ~140 registers per lane into scratch inside the k loop and ran
    // at 120 TFLOP/s -- tools/lab_w8_prefill.py; 2 n-blocks fit)
    p.nb = (wbits == 4 && p.mt == 8 && n % 256 == 0 && tiles_m * (n / 256) >= 256) ? 4 : 2;
    const int wgs = tiles_m * (n / (p.nb * 64));
    const int nk = (k + (wbits == 4 ? 127 : 63)) / (wbits == 4 ? 128 : 64);
    // split K while the output tiles leave most CUs idle (fp32 slabs + reduce launch): >= 4 k-tiles per split
    int ks = 1;
    // as many splits as still fit the chip in one residency (2 workgroups per CU): a third partial round costs more
    // than the extra split gains (sweep in tools/lab_prefill_sweep.sh: m = 128 at 28672 x 8192, 2 splits 97 us,
    // 3 splits 107 us incl. the reduce)
    if (wgs < 256)
        ks = std::max(1, std::min(std::min((p.mt == 8 ? 512 : 768) / wgs, nk / 4), 16)); // 64-row tiles: 3 per CU
    const int per = (nk + ks - 1) / ks;
    p.ksplit = (nk + per - 1) / per;
    return p;
}

// grid.x = tiles_m * tiles_n (remapped XCD-contiguous, m fastest: neighbours share the weight panel)
// NW: waves per workgroup.  4: tile (MT*16) x (NB*64), two workgroups per CU.  8: tile 256 x 256 (MT = 16, NB = 2), one
// workgroup per CU (128 KiB of LDS for the int4 k-tile): every dequantised weight fragment then feeds 16 MFMAs instead
// of 8, which halves the vector-ALU work per MFMA (worth 5 - 13 % once the activation fragment reads stopped being
// serialised, see compute()).
// DEPTH: tiles of the k loop in flight + 1.  2 = ping-pong (one tile ahead: enough when a tile's MFMAs outlast a load);
// 4 = a ring with three tiles ahead for the shortest tiles (MT = 2, i.e. 32 rows per workgroup: a tile is 16 MFMAs per
// wave, far shorter than an HBM round trip, and with one tile in flight per workgroup the launch is latency-bound).  At
// MT = 4 the ring's 64 KiB of LDS leave 2 workgroups per CU where the ping-pong has 5, and the ping-pong wins (grouped
// MoE launch, 64 rows per expert: 167 against 192 us, tools/lab_moe_depth.py).
// INTD (int4 experts of the grouped MoE launch at decode: per-column scales, no zeros, <= 64-row tiles of a weight
// stream): the integer domain of the decode GEMV (gemv_tile_int4, woq_gemm.hip) with the scale deferred to the
// epilogue.  At these heights a dequantised fragment feeds 2 - 4 MFMAs and the exact dequantisation's 23 vector-ALU
// operations per 8 weights are the bound, next to HBM; here the nibbles go to the MFMA as u = 0 .. 15 (7 operations per
// 8 weights), P = sum a u and S1 = sum a accumulate over the workgroup's whole k range (S1 by one more MFMA against
// ones per k-step, m-block j on wave j % NW, shared through LDS at the end), and out = s (P - 8 S1) once -- for a
// one-hot activation row that is s q exactly.  No scale registers in the ring: 107 registers, 4 workgroups per CU (the
// per-group form of the same arithmetic needed 160 and ran 3: the launch lives on bytes in flight).
template <typename T, int WBITS, bool ZERO, int NSZ, int MT, int NB, int DEPTH = 2, int NW = PF_WAVES, bool INTD = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2))) void woq_prefill_kernel(WoqArgs a, int tiles_m, int tiles_n)
{
    static_assert(!INTD || (!ZERO && NSZ == 1), "integer domain with deferred scales: per-column, no zeros");
    constexpr int PF_TN = NB * 16 * NW;
    constexpr int KTILE = Geo<WBITS>::KTILE;   // 128 (int4) / 64 (int8) rows of k per weight tile
    constexpr int KSTEPS = Geo<WBITS>::KSTEPS; // MFMA k-steps per tile
    constexpr int TM = MT * 16;
    constexpr int RB = KTILE * 2;              // bytes per activation row in LDS: 256 / 128
    constexpr int CHUNKS = RB / 16;            // 16-byte chunks per row: 16 / 8
    constexpr int ROWS_PER_DMA = 1024 / RB;    // 4 / 8
    constexpr int BUF_BYTES = TM * RB;
    constexpr int DMA_PER_TILE = BUF_BYTES / 1024;
    constexpr int DMA_PER_WAVE = (DMA_PER_TILE + NW - 1) / NW;
    static_assert(DEPTH == 2 || DMA_PER_TILE % NW == 0, "the ring counts its loads: every wave must issue the same number");
    __shared__ __attribute__((aligned(1024))) uint8_t lds[DEPTH * BUF_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, x = bid % 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
    }
    int tn = bid / tiles_m;
    int tm = bid % tiles_m;
    if (a.pf_group_m > 0)
    {
        // Rasterisation for the tall tiles.  The workgroups an XCD holds at one time are consecutive here; with m fastest
        // over the whole launch they are one column of m-tiles: one weight panel shared, every activation panel (4 x the
        // bytes of a weight panel per tile: 16-bit against 4-bit) fetched by one workgroup only and fetched again by the
        // XCD's next column.  Groups of G m-tiles x all n-tiles, m fastest inside: an XCD's 32 resident tiles are G
        // activation panels x 32 / G weight panels.
        const int G = a.pf_group_m;
        const int gsz = G * tiles_n;
        const int grp = bid / gsz, rem = bid - grp * gsz;
        const int gm = min(G, tiles_m - grp * G); // the last group may be shorter
        tm = __builtin_amdgcn_readfirstlane(grp * G + rem % gm);
        tn = __builtin_amdgcn_readfirstlane(rem / gm);
    }
    int moe_row0 = 0; // grouped form: the expert's first row in the expert-sorted order
    if (a.moe_rows)
    {
        // grouped form: m-tile index over all experts -> (expert, its m-tile); then this workgroup is an ordinary tile
        // of that expert's GEMM.  Everything here is wave-uniform (scalar loads); tiles past the last expert's leave.
        int64_t row0 = 0;
        int e = 0, rows = 0, skip = 0;
        if (a.moe_tail_mod)
        {
            // the tail part of a two-part launch (below): 256 / TM tiles per expert, indexed directly -- most of these
            // workgroups find nothing to do, and two scalar loads are all they should cost
            constexpr int PER = 256 / TM > 0 ? 256 / TM : 1;
            e = tm / PER;
            tm %= PER;
            if (e >= a.moe_experts)
                return;
            row0 = e ? a.moe_rows[e - 1] : 0;
            rows = (int) (a.moe_rows[e] - row0);
            skip = rows & ~255;
            if (tm * TM >= rows - skip)
                return;
        }
        else
        {
            int tiles_before = 0;
            for (; e < a.moe_experts; ++e)
            {
                const int64_t end = a.moe_rows[e];
                rows = (int) (end - row0);
                if (a.moe_full_only) // the main part of a two-part launch ends at the expert's last whole tile
                    rows -= rows % TM;
                const int t = (rows + TM - 1) / TM;
                if (tm < tiles_before + t)
                    break;
                tiles_before += t;
                row0 = end;
            }
            if (e == a.moe_experts)
                return;
            tm -= tiles_before;
        }
        row0 += skip;
        a.m = rows - skip;
        moe_row0 = (int) row0;
        if (!a.moe_row_map)
            a.act += row0 * a.k;
        a.out += row0 * (a.glu_inter ? a.glu_inter : a.n);
        a.slabs += row0 * a.n;
        a.qweight += (size_t) e * a.moe_wbytes;
        a.scales += (size_t) e * a.n;
        if (a.bias)
            a.bias += (size_t) e * a.n;
    }
    const int m0 = tm * TM, n0 = tn * PF_TN;

    // swizzle of the 16-byte chunk index by row: 16 consecutive rows x one logical chunk -> 256 distinct bytes
    auto sw = [](int row) { return CHUNKS == 16 ? (row & 15) : ((row >> 1) & 7); };

    // ---- activation staging (LDS-DMA).  DMA d covers rows [d*ROWS_PER_DMA, +ROWS_PER_DMA); lane -> (row, stored chunk)
    const int drow = lane / CHUNKS, dchunk = lane % CHUNKS;
    uint32_t aoff[DMA_PER_WAVE]; // byte offset of the lane's 16 bytes in tile 0, from a.act (host checks m * k * 2 < 4 GiB)
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i)
    {
        const int d = wave * DMA_PER_WAVE + i;
        const int row = d * ROWS_PER_DMA + drow;
        // rows past m re-read the last row; never stored
        uint32_t src_row = (uint32_t) min(m0 + row, a.m - 1);
        if (a.moe_row_map) // gathered on the fly: the token row behind this expert-order row
            src_row = (uint32_t) (a.moe_row_map[moe_row0 + (int) src_row] % a.moe_row_mod);
        aoff[i] = src_row * (uint32_t) a.k * 2u + (dchunk ^ sw(row)) * 16;
    }
    // k range of this workgroup (split-K over grid.y); the int4 half tile belongs to the last split
    const bool half_tile = WBITS == 4 && (a.k & 127);
    const int nk_all = a.ktiles + (half_tile ? 1 : 0);
    const int per = (nk_all + a.ksplit - 1) / a.ksplit;
    const int kt_lo = blockIdx.y * per;
    const int nk = min(nk_all, kt_lo + per);
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t) (uintptr_t) (lds_void_t*) lds);
    const uint16_t* act_base = a.act;
    auto stage = [&](int kt, int buf) // a full tile
    {
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i)
        {
            const int d = wave * DMA_PER_WAVE + i;
            if (DMA_PER_TILE % NW != 0 && d >= DMA_PER_TILE)
                continue;
            pf_dma16(act_base, aoff[i] + kt * (KTILE * 2), lds_base + buf * BUF_BYTES + d * 1024);
        }
    };
    auto stage_half = [&](int kt, int buf) // the int4 half tile: chunks past k are staged from a valid address, never read
    {
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i)
        {
            const int d = wave * DMA_PER_WAVE + i;
            if (DMA_PER_TILE % NW != 0 && d >= DMA_PER_TILE)
                continue;
            const int achunk = dchunk ^ sw(d * ROWS_PER_DMA + drow);
            const uint32_t back = achunk * 8 >= 64 ? achunk * 16 : 0;
            pf_dma16(act_base, aoff[i] - back + kt * (KTILE * 2), lds_base + buf * BUF_BYTES + d * 1024);
        }
    };

    // ---- weights: this wave's two n-blocks, one k-tile per 16-byte load, one tile ahead
    const size_t run_bytes = (size_t) a.k * 16 * WBITS / 8;
    // the wave's n-blocks: NB adjacent ones, or -- gated epilogue -- NB / 2 of the linear half and the matching NB / 2
    // of the gate half (the workgroup then covers PF_TN / 2 output columns)
    int nbi[NB];
    {
        const int nb0 = (n0 >> 4) + wave * NB, lin0 = (n0 >> 5) + wave * (NB / 2);
#pragma unroll
        for (int b = 0; b < NB; ++b)
            nbi[b] = !a.glu_inter ? nb0 + b : (b < NB / 2 ? lin0 + b : (a.glu_inter >> 4) + lin0 + (b - NB / 2));
    }
    struct WTile
    {
        uint4 w[NB];
        uint32_t s[NB][NSZ], z[NB][NSZ]; // 16-bit values, zero-extended (not loaded with INTD: applied in the epilogue)
    };
    auto load_sz = [&](WTile& t, int kt)
    {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int h = 0; h < NSZ; ++h)
            {
                const int k = min(kt * KTILE + h * 64, a.k - 1);
                if constexpr (!INTD)
                {
                    const size_t si = (size_t) (k >> a.group_shift) * a.n + nbi[b] * 16 + (lane & 15);
                    t.s[b][h] = a.scales[si];
                    t.z[b][h] = ZERO ? a.zeros[si] : (uint16_t) 0;
                }
            }
    };
    auto load_w = [&](WTile& t, int kt) // a full tile: no branch between the loads
    {
#pragma unroll
        for (int b = 0; b < NB; ++b)
            t.w[b] = *reinterpret_cast<const uint4*>(a.qweight + (size_t) nbi[b] * run_bytes + (size_t) kt * 1024 + lane * 16);
        load_sz(t, kt);
    };
    auto load_w_half = [&](WTile& t, int kt) // the 512-byte trailing half tile of int4 with k % 128 == 64: 8 bytes per lane
    {
#pragma unroll
        for (int b = 0; b < NB; ++b)
        {
            const uint2 hw = *reinterpret_cast<const uint2*>(a.qweight + (size_t) nbi[b] * run_bytes + (size_t) kt * 1024 + lane * 8);
            t.w[b] = make_uint4(hw.x, hw.y, 0, 0);
        }
        load_sz(t, kt);
    };
    // "the loads into t have landed" for the compiler's own bookkeeping: placed right after the explicit
    // s_waitcnt vmcnt(0) of a tile boundary.  Without it hipcc (which does not see that wait, and does not count the
    // asm DMAs) puts a vmcnt in front of the tile's first use that also waits for part of the NEXT tile's prefetch.
    auto landed = [&](WTile& t)
    {
#pragma unroll
        for (int b = 0; b < NB; ++b)
        {
            asm volatile("" : "+v"(t.w[b].x), "+v"(t.w[b].y), "+v"(t.w[b].z), "+v"(t.w[b].w));
#pragma unroll
            for (int h = 0; h < NSZ; ++h)
            {
                if constexpr (!INTD)
                    asm volatile("" : "+v"(t.s[b][h]), "+v"(t.z[b][h]));
            }
        }
    };

    f32x4_t acc[NB][MT];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < MT; ++j)
            acc[b][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = lane >> 4;
    // per m-tile j the fragment row is j*16 + frow: sw(row) only depends on frow (CHUNKS == 16) or on
    // frow and nothing else either ((row >> 1) & 7 with row = 16j + frow) -> hoisted
    const int fsw = sw(frow);
    const uint8_t* frag_base = lds + frow * RB;

    // full tiles [kt_lo, nfull) run in the ping-pong loop; the int4 half tile (if it falls into this split) afterwards
    const int nfull = min(nk, a.ktiles);
    WTile wt[DEPTH];
    if constexpr (DEPTH == 2)
    {
        if (kt_lo < nfull)
        {
            stage(kt_lo, 0);
            load_w(wt[0], kt_lo);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        landed(wt[0]);
        __syncthreads();
    }

    constexpr int S1N = INTD ? (MT + NW - 1) / NW : 1; // m-blocks whose S1 this wave accumulates: j = wave + NW i
    f32x4_t s1all[S1N];
#pragma unroll
    for (int i = 0; i < S1N; ++i)
        s1all[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto compute_deferred = [&](const WTile& t, int buf, int steps)
    {
        const uint8_t* sa = frag_base + buf * BUF_BYTES;
        const uint4 ones = ones8<T>();
        pf_static_for<KSTEPS>(
            [&](auto stc)
            {
                constexpr int st = decltype(stc)::value;
                if (st >= steps)
                    return;
                uint4 wf[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b)
                {
                    const uint32_t* wd = reinterpret_cast<const uint32_t*>(&t.w[b]);
                    if constexpr (WBITS == 4)
                        wf[b] = unpack8_u4<T>(wd[st]);
                    else // int8: q itself, exact in 16 bits; no offset, hence no S1
                        wf[b] = cvt8_i8<T>(wd[2 * st], wd[2 * st + 1]);
                }
                const int coff = ((st * 4 + fk) ^ fsw) << 4;
                constexpr int AFW = NB == 2 ? 4 : 2;
                constexpr int AF = MT - 1 < AFW ? MT - 1 : AFW, R = AF + 1;
                const uint32_t abase = (uint32_t) (uintptr_t) (lds_void_t*) sa + (uint32_t) coff;
                pf_u32x4_t av[R];
                pf_static_for<AF>([&](auto jc) { pf_lds_read<decltype(jc)::value * 16 * RB>(av[decltype(jc)::value % R], abase); });
                pf_static_for<MT>(
                    [&](auto jc)
                    {
                        constexpr int j = decltype(jc)::value;
                        if constexpr (j + AF < MT)
                            pf_lds_read<(j + AF) * 16 * RB>(av[(j + AF) % R], abase);
                        constexpr int issued = j + AF < MT ? j + 1 + AF : MT;
                        pf_lds_wait<issued - (j + 1)>(av[j % R]);
                        const uint4 af = __builtin_bit_cast(uint4, av[j % R]);
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            acc[b][j] = mfma16<T>(wf[b], af, acc[b][j]);
                        if constexpr (WBITS == 4 && INTD) // (the lambda's body is compiled for every instantiation)
                            if (j % NW == wave) // wave-uniform
                                s1all[j / NW] = mfma16<T>(ones, af, s1all[j / NW]);
                    });
            });
    };
    auto compute_exact = [&](const WTile& t, int buf, int steps)
    {
        float s[NB][NSZ], z[NB][NSZ];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int h = 0; h < NSZ; ++h)
            {
                s[b][h] = T16<T>::to_float((uint16_t) t.s[b][h]);
                z[b][h] = T16<T>::to_float((uint16_t) t.z[b][h]);
            }
        const uint8_t* sa = frag_base + buf * BUF_BYTES;
#pragma unroll
        for (int st = 0; st < KSTEPS; ++st)
        {
            if (st >= steps)
                break;
            constexpr int H1 = NSZ - 1;
            const int h = (WBITS == 4 && st >= 2) ? H1 : 0;
            uint4 wf[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b)
            {
                if (st == 0)
                    wf[b] = dequant_step<T, WBITS, 0>(t.w[b], s[b][h], z[b][h]);
                else if (st == 1)
                    wf[b] = dequant_step<T, WBITS, 1>(t.w[b], s[b][h], z[b][h]);
                else if (st == 2)
                    wf[b] = dequant_step<T, WBITS, 2 % KSTEPS>(t.w[b], s[b][h], z[b][h]);
                else
                    wf[b] = dequant_step<T, WBITS, 3 % KSTEPS>(t.w[b], s[b][h], z[b][h]);
            }
            const int coff = ((st * 4 + fk) ^ fsw) << 4;
            // Activation fragments run AF m-blocks ahead of the MFMAs that use them, in a ring of AF + 1 register quads,
            // read and waited for from inline asm: left to itself hipcc reads every fragment into the SAME quad right in
            // front of its MFMAs (ds_read, lgkmcnt(0), NB MFMAs, ds_read ...), so every group of NB MFMAs starts with an
            // exposed LDS round trip -- that, not the dequantisation, kept the matrix pipe at half rate.  The LDS returns
            // in order: "fragment j has landed" = at most (reads issued after it) outstanding.
            // (distance: about one LDS round trip of MFMAs -- 4 groups of NB = 2, 2 of NB = 4; 1 where the second
            // scale / zero set of group size 64 leaves no registers)
            constexpr int AFW = NB == 2 ? 4 : (NSZ == 2 ? 1 : 2);
            constexpr int AF = MT - 1 < AFW ? MT - 1 : AFW, R = AF + 1;
            const uint32_t abase = (uint32_t) (uintptr_t) (lds_void_t*) sa + (uint32_t) coff;
            pf_u32x4_t av[R];
            pf_static_for<AF>([&](auto jc) { pf_lds_read<decltype(jc)::value * 16 * RB>(av[decltype(jc)::value % R], abase); });
            pf_static_for<MT>(
                [&](auto jc)
                {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (j + AF < MT)
                        pf_lds_read<(j + AF) * 16 * RB>(av[(j + AF) % R], abase);
                    constexpr int issued = j + AF < MT ? j + 1 + AF : MT;
                    pf_lds_wait<issued - (j + 1)>(av[j % R]);
                    const uint4 af = __builtin_bit_cast(uint4, av[j % R]);
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        acc[b][j] = mfma16<T>(wf[b], af, acc[b][j]);
                });
        }
    };

    auto compute = [&](const WTile& t, int buf, int steps)
    {
        if constexpr (INTD)
            compute_deferred(t, buf, steps);
        else
            compute_exact(t, buf, steps);
    };

    // one tile boundary: everything issued for the next tile (DMA and weights) has landed, all waves are done with
    // the buffer that is staged next
    auto boundary = [&](WTile& next)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        landed(next);
        __syncthreads();
    };
    if constexpr (DEPTH == 2)
    {
        for (int kt = kt_lo; kt < nfull; kt += 2)
        {
            if (kt + 1 < nfull)
            {
                stage(kt + 1, 1);
                load_w(wt[1], kt + 1);
            }
            compute(wt[0], 0, KSTEPS);
            boundary(wt[1]);
            if (kt + 1 >= nfull)
                break;
            if (kt + 2 < nfull)
            {
                stage(kt + 2, 0);
                load_w(wt[0], kt + 2);
            }
            compute(wt[1], 1, KSTEPS);
            boundary(wt[0]);
        }
    }
    else
    {
        // the ring.  Vector-memory operations a wave issues per tile, in issue order: DMA_PER_WAVE LDS-DMAs, then the
        // weight and scale / zero loads.  vmcnt retires in order, so "tile kt has landed" = at most (tiles issued
        // after it) x L operations outstanding.  hipcc cannot see the DMAs and adds its own, somewhat stricter wait
        // in front of landed(); the explicit one is what covers the DMAs.
        constexpr int L = DMA_PER_WAVE + NB * (1 + (INTD ? 0 : NSZ * (ZERO ? 2 : 1)));
        static_assert((DEPTH - 2) * L < 64, "vmcnt immediate");
        auto issue = [&](auto slot, int kt)
        {
            constexpr int s = decltype(slot)::value;
            stage(kt, s);
            load_w(wt[s], kt);
        };
        auto step = [&](auto slot, int kt)
        {
            constexpr int s = decltype(slot)::value;
            const int later = min(DEPTH - 2, nfull - 1 - kt); // tiles issued after tile kt
            if (later >= 2)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L) : "memory");
            else if (later == 1)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            landed(wt[s]);
            __syncthreads(); // every wave's share of tile kt is in LDS; every wave is done with tile kt - 1's buffer
            if (kt + DEPTH - 1 < nfull)
                issue(PfSlot<(s + DEPTH - 1) % DEPTH>{}, kt + DEPTH - 1);
            compute(wt[s], s, KSTEPS);
        };
        static_assert(DEPTH == 4, "the ring is written out for 4 slots");
        if (kt_lo < nfull)
            issue(PfSlot<0>{}, kt_lo);
        if (kt_lo + 1 < nfull)
            issue(PfSlot<1>{}, kt_lo + 1);
        if (kt_lo + 2 < nfull)
            issue(PfSlot<2>{}, kt_lo + 2);
        for (int kt = kt_lo; kt < nfull; kt += 4)
        {
            step(PfSlot<0>{}, kt);
            if (kt + 1 < nfull)
                step(PfSlot<1>{}, kt + 1);
            if (kt + 2 < nfull)
                step(PfSlot<2>{}, kt + 2);
            if (kt + 3 < nfull)
                step(PfSlot<3>{}, kt + 3);
        }
        __syncthreads(); // the half tile below restages buffer 0
    }
    if (WBITS == 4 && nk > nfull) // the half tile: both buffers are free after the last boundary
    {
        stage_half(a.ktiles, 0);
        load_w_half(wt[0], a.ktiles);
        boundary(wt[0]);
        compute(wt[0], 0, 2);
    }

    // epilogue.  D[row = n][col = m]: lane holds n = 4*(l>>4) + {0..3}, m = l & 15.
    const int em = lane & 15, en = (lane >> 4) * 4;
    if constexpr (INTD)
    {
        // the deferred scales: acc holds P = sum a u over this workgroup's k range, the waves' s1all the matching
        // sum a (every D row of an MFMA against ones is S1[m]: lanes 0 .. 15 hold it).  Shared through LDS -- the
        // k loop's buffers are free after its last barrier --, then out (or this split's slab) = s (P - 8 S1).
        float s1[MT];
        if constexpr (WBITS == 4)
        {
            float* s1x = reinterpret_cast<float*>(lds);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < S1N; ++i)
                if (wave + NW * i < MT && lane < 16)
                    s1x[(wave + NW * i) * 16 + lane] = s1all[i][0];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < MT; ++j)
                s1[j] = s1x[j * 16 + em];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
        {
            const uint2 sv2 = *reinterpret_cast<const uint2*>(a.scales + (size_t) nbi[b] * 16 + en);
            const float sv[4] = {T16<T>::lo(sv2.x), T16<T>::hi(sv2.x), T16<T>::lo(sv2.y), T16<T>::hi(sv2.y)};
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[b][j][i] = WBITS == 4 ? sv[i] * fmaf(-8.f, s1[j], acc[b][j][i]) : sv[i] * acc[b][j][i];
        }
    }
    const float alpha = (a.alpha && a.ksplit == 1) ? a.alpha[0] : 1.f; // out = alpha * acc + bias
    auto bias4 = [&](int n, float (&bias)[4])
    {
        bias[0] = bias[1] = bias[2] = bias[3] = 0.f;
        if (a.bias && a.ksplit == 1)
        {
            const uint2 bv = *reinterpret_cast<const uint2*>(a.bias + n);
            bias[0] = T16<T>::lo(bv.x);
            bias[1] = T16<T>::hi(bv.x);
            bias[2] = T16<T>::lo(bv.y);
            bias[3] = T16<T>::hi(bv.y);
        }
    };
    if (a.glu_inter) // wave-uniform; the host only asks for it without split-K
    {
        // the reference's two steps in one: the GEMM's T-rounded outputs (bias added, identity activation), then
        // out = T(linear * T(act(gate))) -- the values and roundings of moe_gated_act_kernel
#pragma unroll
        for (int b = 0; b < NB / 2; ++b)
        {
            const int n = nbi[b] * 16 + en, ng = nbi[b + NB / 2] * 16 + en;
            float bl[4], bg[4];
            bias4(n, bl);
            bias4(ng, bg);
#pragma unroll
            for (int j = 0; j < MT; ++j)
            {
                const int m = m0 + j * 16 + em;
                if (m >= a.m)
                    continue;
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                {
                    const float lin = T16<T>::to_float(T16<T>::from_float(acc[b][j][i] * alpha + bl[i]));
                    const float gate = T16<T>::to_float(T16<T>::from_float(acc[b + NB / 2][j][i] * alpha + bg[i]));
                    o[i] = lin * T16<T>::to_float(T16<T>::from_float(woq_activation(gate, a.glu_fn)));
                }
                *reinterpret_cast<uint2*>(a.out + (size_t) m * a.glu_inter + n) = make_uint2(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]));
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
    {
        const int n = nbi[b] * 16 + en;
        float bias[4];
        bias4(n, bias);
#pragma unroll
        for (int j = 0; j < MT; ++j)
        {
            const int m = m0 + j * 16 + em;
            if (m >= a.m)
                continue;
            const size_t oi = (size_t) m * a.n + n;
            if (a.ksplit == 1)
                *reinterpret_cast<uint2*>(a.out + oi)
                    = make_uint2(pack2<T>(woq_activation(acc[b][j][0] * alpha + bias[0], a.act_fn),
                                     woq_activation(acc[b][j][1] * alpha + bias[1], a.act_fn)),
                        pack2<T>(woq_activation(acc[b][j][2] * alpha + bias[2], a.act_fn),
                            woq_activation(acc[b][j][3] * alpha + bias[3], a.act_fn)));
            else
                *reinterpret_cast<float4*>(a.slabs + (size_t) blockIdx.y * a.slab_stride + oi)
                    = make_float4(acc[b][j][0], acc[b][j][1], acc[b][j][2], acc[b][j][3]);
        }
    }
}

template <typename T, int WBITS, bool ZERO, int NSZ, int MT, int NB, int NW>
void launch_prefill_mt_grouped(const WoqArgs& a, int tiles_m, int tiles_n, hipStream_t stream);

template <typename T, int WBITS, bool ZERO, int NSZ, int MT, int NB, int NW = PF_WAVES>
void launch_prefill_mt(const WoqArgs& a, hipStream_t stream)
{
    const int tiles_m = (a.m + MT * 16 - 1) / (MT * 16), tiles_n = a.n / (NB * 16 * NW);
    WoqArgs b = a;
    // groups of 4 m-tiles for the 256-row tiles, 8 for the 128-row ones (twice the resident tiles per XCD, half the
    // activation panel: W8A16 4096 x 4096 x 11008 375 -> 348 us, 4096 x 11008 x 4096 430 -> 405; nothing to gain below)
    b.pf_group_m = MT == 16 ? 4 : (MT == 8 ? 8 : 0);
    if (const char* e = lab_env(MT == 16 ? "MI355Q_PF_GROUP_M" : "MI355Q_PF_GROUP_M8"))
        b.pf_group_m = atoi(e);
    return launch_prefill_mt_grouped<T, WBITS, ZERO, NSZ, MT, NB, NW>(b, tiles_m, tiles_n, stream);
}

template <typename T, int WBITS, bool ZERO, int NSZ, int MT, int NB, int NW>
void launch_prefill_mt_grouped(const WoqArgs& a, int tiles_m, int tiles_n, hipStream_t stream)
{
    if constexpr (!ZERO && NSZ == 1 && NW == PF_WAVES && (WBITS == 4 || MT >= 8))
    {
        // per-channel scales without zeros (weightOnlyQuantMatmulPlugin's forms): the scale is the same for every k, so
        // it is deferred to the epilogue.  int8: the values go to the MFMA as they are (12 vector-ALU operations per 8
        // weights instead of 20, no scale registers through the loop), out = s * sum a q.  int4: the integer-domain
        // tiles of the grouped MoE launch, out = s (P - 8 S1).  One-hot rows give s q exactly either way.  (Not the
        // 256 x 256 tile: MFMA-bound, see moe_weight_only_gemm.)
        if (a.group_shift == 31 && !lab_env("MI355Q_W8_EXACT"))
        {
            hipLaunchKernelGGL((woq_prefill_kernel<T, WBITS, ZERO, NSZ, MT, NB, 2, NW, true>), dim3(tiles_m * tiles_n, a.ksplit),
                dim3(NW * 64), 0, stream, a, tiles_m, tiles_n);
            return;
        }
    }
    // (the 4-slot ring of the grouped MoE launch was measured here too, m = 40 - 64 at seven Llama shapes: 2 - 17 %
    // SLOWER than the ping-pong -- these launches are split over K into 4-tile pieces, a ring never fills)
    hipLaunchKernelGGL((woq_prefill_kernel<T, WBITS, ZERO, NSZ, MT, NB, 2, NW>), dim3(tiles_m * tiles_n, a.ksplit),
        dim3(NW * 64), 0, stream, a, tiles_m, tiles_n);
}

template <typename T, int WBITS, bool ZERO>
void launch_prefill(const WoqArgs& a, hipStream_t stream)
{
    constexpr int NSZ2 = WBITS == 4 ? 2 : 1;
    const bool split = WBITS == 4 && a.group == 64;
    const PrefillPlan p = plan_prefill(a.m, a.n, a.k, WBITS);
#define MQ_PF(MTV, NBV)                                                                                                \
    do                                                                                                                 \
    {                                                                                                                  \
        if (split)                                                                                                     \
            launch_prefill_mt<T, WBITS, ZERO, NSZ2, MTV, NBV>(a, stream);                                              \
        else                                                                                                           \
            launch_prefill_mt<T, WBITS, ZERO, 1, MTV, NBV>(a, stream);                                                 \
    } while (0)
    if (p.mt == 16)
    {
        if constexpr (WBITS == 4)
        {
            if (split)
                launch_prefill_mt<T, WBITS, ZERO, NSZ2, 16, 2, 8>(a, stream);
            else
                launch_prefill_mt<T, WBITS, ZERO, 1, 16, 2, 8>(a, stream);
        }
    }
    else if (p.mt == 2)
        MQ_PF(2, 2);
    else if (p.mt == 4)
        MQ_PF(4, 2);
    else if (p.nb == 2 || WBITS != 4)
        MQ_PF(8, 2);
    else
    {
        if constexpr (WBITS == 4)
            MQ_PF(8, 4);
    }
#undef MQ_PF
}


} // namespace

bool woq_prefill_supported(int m, int n, int k)
{
    return m > 16 && n % 128 == 0 && k % 64 == 0;
}

int woq_prefill_ksplit(int m, int n, int k, int weight_type)
{
    return plan_prefill(m, n, k, weight_type == MI355Q_WEIGHT_INT4 ? 4 : 8).ksplit;
}

void weight_only_prefill_gemm(const WoqArgsHost& h, hipStream_t stream)
{
    WoqArgs a{};
    a.qweight = h.qweight;
    a.scales = h.scales;
    a.zeros = h.zeros;
    a.act = h.act;
    a.bias = h.bias;
    a.alpha = h.alpha;
    a.out = h.out;
    a.m = h.m;
    a.n = h.n;
    a.k = h.k;
    a.group = h.group;
    a.group_shift = h.group_shift;
    a.ktiles = h.ktiles;
    a.act_fn = 5; // identity
    MQ_CHECK((uint64_t) h.m * (uint64_t) h.k * 2 < (1ull << 32),
        "activation matrix of %d x %d exceeds the prefill kernel's 32-bit byte offsets (4 GiB)", h.m, h.k);
    a.ksplit = h.slabs ? plan_prefill(h.m, h.n, h.k, h.int4 ? 4 : 8).ksplit : 1;
    a.slabs = h.slabs;
    a.slab_stride = (size_t) h.m * h.n;
    const bool bf = h.bf16, i4 = h.int4, z = h.zeros != nullptr;
    if (bf && i4 && z) launch_prefill<bf16_t, 4, true>(a, stream);
    else if (bf && i4) launch_prefill<bf16_t, 4, false>(a, stream);
    else if (bf && z) launch_prefill<bf16_t, 8, true>(a, stream);
    else if (bf) launch_prefill<bf16_t, 8, false>(a, stream);
    else if (i4 && z) launch_prefill<f16_t, 4, true>(a, stream);
    else if (i4) launch_prefill<f16_t, 4, false>(a, stream);
    else if (z) launch_prefill<f16_t, 8, true>(a, stream);
    else launch_prefill<f16_t, 8, false>(a, stream);
    MQ_HIP(hipGetLastError());
}

// Tile height of the grouped MoE launch.  The host sees only the total row count, but with real routing the experts'
// row counts scatter around the average (about a binomial: sigma^2 ~ avg (1 - 1/experts)), and a workgroup tile that the
// average fills exactly is half wasted on every expert that got a few rows more.  Expected padded rows per expert
// E[ceil(r / TM)] TM for r ~ N(avg, sigma^2), weighted by how well each tile height runs.  The weights were re-measured on
// the WHOLE LAYER under random routing once the integer-domain tiles and the grouped tile walk existed
// (tools/lab_moe_tile_layer.py; time per padded row): 64-row tiles 1.0, 128-row 0.98, the 256 x 256 / 8-wave tile 1.15
// -- but the tall tile only from 768 rows per expert: below that its few tiles fill the chip badly in the w2 GEMM and
// half of its rows are padding (192 / 384 / 512 / 640 rows per expert: 64-row tiles 754 / 1315 / 1730 / 2066 us against
// 1050 / 1720 / 1960 / 2212; 768 rows: 2450 against 2414; 1024 rows: 3220 against 3020).  The earlier weights (0.87 / 1.0 / 1.16) came from the bare GEMM with perfectly
// balanced experts, which flatters tall tiles.  32-row tiles (with their deeper ring) while the launch is a weight
// stream (<= 24 rows per expert).
double moe_expected_padded_rows(double avg, double sigma, int tm)
{
    auto cdf = [&](double x) { return 0.5 * (1.0 + std::erf((x - avg) / (sigma * 1.4142135623730951))); };
    const int j0 = std::max(1, (int) std::floor((avg - 5 * sigma) / tm)), j1 = (int) std::ceil((avg + 5 * sigma) / tm) + 1;
    double e = 0.0, mass = 0.0;
    for (int j = j0; j <= j1; ++j)
    {
        const double pj = cdf((double) j * tm) - cdf((double) (j - 1) * tm);
        e += pj * j * tm;
        mass += pj;
    }
    return mass > 0 ? e / mass : avg;
}

int moe_pick_tile(int64_t total_rows, int experts, bool allow256)
{
    const double avg = (double) total_rows / experts;
    if (avg <= 24.0)
        return 2;
    const double sigma = std::sqrt(std::max(1.0, avg * (1.0 - 1.0 / std::max(2, experts))));
    const double c4 = moe_expected_padded_rows(avg, sigma, 64), c8 = moe_expected_padded_rows(avg, sigma, 128) / 0.98;
    const double c16 = allow256 && avg >= 768.0 ? moe_expected_padded_rows(avg, sigma, 256) / 1.15 : 1e30;
    if (c16 < c8 && c16 < c4)
        return 16;
    return c4 < c8 ? 4 : 8;
}

// The expert GEMMs of a mixture-of-experts layer with weight-only quantised experts, as ONE launch
// (MoeGemmRunner<T, uint8_t | cutlass::uint4b_t>::moeGemmBiasAct, kernels/cutlass_kernels/moe_gemm/moe_gemm_kernels.h:51-58,
// moe_gemm_kernels_template.h:401-470; the reference runs a CUTLASS grouped GEMM whose problem sizes a kernel derives
// from total_rows_before_expert).  Here the prefill kernel's m-tile index simply runs over all experts: a workgroup
// scans the cumulative row counts (device memory, <= 64 scalar loads), finds its expert and is from then on a tile of
// that expert's GEMM.  The grid is sized for the worst case sum_e ceil(rows_e / TM) <= ceil(total / TM) + experts;
// surplus workgroups leave at once.
int moe_weight_only_gemm(const mi355q_moe_gemm_params& p, hipStream_t stream, bool reduce, const int32_t* row_map, int row_mod,
    int glu_fn)
{
    if (p.total_rows <= 0)
        return 1;
    MQ_CHECK(p.A && p.B && p.weight_scales && p.C && p.total_rows_before_expert, "null tensor");
    MQ_CHECK(p.act_type == MI355Q_HALF || p.act_type == MI355Q_BF16, "Unsupported data type %d", p.act_type);
    MQ_CHECK(p.weight_type == MI355Q_WEIGHT_INT4 || p.weight_type == MI355Q_WEIGHT_INT8, "weight_type %d", p.weight_type);
    MQ_CHECK(p.num_experts >= 1 && p.num_experts <= 1024, "num_experts %d", p.num_experts);
    MQ_CHECK(p.gemm_n > 0 && p.gemm_n % 128 == 0 && p.gemm_k > 0 && p.gemm_k % 64 == 0,
        "gemm_n must be a multiple of 128 and gemm_k of 64 (n=%lld k=%lld)", (long long) p.gemm_n, (long long) p.gemm_k);
    MQ_CHECK(p.activation_type == 0 || p.activation_type == 1 || p.activation_type == 2 || p.activation_type == 5,
        "activation_type %d: gelu (0), relu (1), silu (2), identity (5); the gated forms run identity here and "
        "doGatedActivation afterwards, as in the reference (moe_kernels.cu)", p.activation_type);
    MQ_CHECK((uint64_t) p.total_rows * (uint64_t) p.gemm_k * 2 < (1ull << 32),
        "activation matrix of %lld x %lld exceeds the kernel's 32-bit byte offsets (4 GiB)", (long long) p.total_rows,
        (long long) p.gemm_k);
    const bool i4 = p.weight_type == MI355Q_WEIGHT_INT4;
    WoqArgs a{};
    a.qweight = static_cast<const uint8_t*>(p.B);
    a.scales = static_cast<const uint16_t*>(p.weight_scales);
    a.act = static_cast<const uint16_t*>(p.A);
    a.bias = static_cast<const uint16_t*>(p.biases);
    a.out = static_cast<uint16_t*>(p.C);
    a.m = (int) p.total_rows;
    a.n = (int) p.gemm_n;
    a.k = (int) p.gemm_k;
    a.group = a.k; // per-column scales [experts, n]
    a.group_shift = 31;
    a.ktiles = a.k / (i4 ? 128 : 64);
    a.ksplit = 1;
    a.act_fn = p.activation_type;
    // tile height from the average rows per expert (the host does not know the split)
    const bool allow256 = i4 && p.gemm_n % 256 == 0 && !lab_env("MI355Q_PF_NO256");
    int mt = moe_pick_tile(p.total_rows, p.num_experts, allow256);
    if (const char* e = lab_env("MI355Q_MOE_MT")) // lab: force the tile height (2 / 4 / 8 m-blocks)
        mt = atoi(e);
    const bool want256 = mt == 16;
    if (want256)
        mt = 8; // (the split-K estimate below and the fallback when the 256-row tile's rounds fill the chip badly)
    // split K over grid.y while the tiles leave most CUs idle (the w2 GEMM has few n-tiles and a long k: at 512 tokens
    // its 256 live workgroups each walked 112 k-tiles and the GEMM took 3x the w1|w3 GEMM's time per flop): fp32 slabs in
    // the caller's workspace + the reduce launch; only without bias / activation (moeGemm), whose epilogue the reduce
    // kernel then is.  Workgroups counted at the tile height actually used (+ half a tile per expert for the ragged ends).
    {
        const int ktiles = (int) (p.gemm_k / (i4 ? 128 : 64));
        const int64_t wgs = (p.gemm_n / 128) * ((p.total_rows + mt * 16 - 1) / (mt * 16) + (p.num_experts + 1) / 2);
        int ks = 1;
        if (p.splitk_workspace && !p.biases && p.activation_type == 5 && wgs < 768 && ktiles >= 16)
        {
            // (one resident round of workgroups, 4 per CU: with the integer-domain tiles 4 splits beat 8 at the Mixtral
            // decode shape, 51 vs 55 us, tools/lab_moe_ks.sh -- and the finalize step reads half the slabs)
            ks = (int) std::min<int64_t>(std::min<int64_t>(8, ktiles / 8), (1024 + wgs - 1) / wgs);
            const size_t per = (size_t) p.total_rows * p.gemm_n * sizeof(float);
            ks = (int) std::min<size_t>(ks, p.splitk_workspace_bytes / per);
            if (const char* e = lab_env("MI355Q_MOE_KS")) // lab: force the split
                ks = std::max(1, std::min(atoi(e), (int) (p.splitk_workspace_bytes / per)));
        }
        if (ks > 1)
        {
            a.ksplit = ks;
            a.slabs = static_cast<float*>(p.splitk_workspace);
            a.slab_stride = (size_t) p.total_rows * p.gemm_n;
        }
    }
    a.moe_rows = p.total_rows_before_expert;
    a.moe_row_map = row_map;
    a.moe_row_mod = row_mod;
    if (glu_fn >= 0)
    {
        MQ_CHECK(a.ksplit == 1 && p.activation_type == 5 && p.gemm_n % 512 == 0,
            "internal: fused gated activation needs an unsplit identity GEMM over 2 x (a multiple of 256) columns");
        a.glu_inter = a.n / 2;
        a.glu_fn = glu_fn;
    }
    a.moe_experts = p.num_experts;
    a.moe_wbytes = (size_t) p.gemm_n * p.gemm_k * (i4 ? 1 : 2) / 2;
    const bool bf = p.act_type == MI355Q_BF16;
    // prefill-sized experts (int4): 256 x 256 tiles on 8 waves, as in the dense launch
    if (want256 && a.ksplit == 1
        && prefer_tile256(((p.total_rows + 255) / 256 + p.num_experts / 2) * (a.n / 256),
            ((p.total_rows + 127) / 128 + p.num_experts / 2) * (a.n / 128)))
    {
        mt = 16;
        // Two parts.  Every expert ends in a ragged tile (half empty on average: with 4096 tokens over 8 experts, 4 of the
        // ~36 tiles held a handful of rows and cost a whole 256-row tile of MFMAs each).  The 256-row launch takes the
        // experts' WHOLE tiles only; a second launch of 64-row integer-domain tiles takes each expert's remaining
        // rows (< 256: at most 4 tiles per expert; tiles past an expert's end leave at once).  Disjoint output rows.
        // (tools/lab_moe_two_parts.py, whole layer, uniform routing, one part -> two: 1024 rows per expert 3.05 -> 2.92 ms,
        // 2048 rows 5.65 -> 5.57; 768 rows 2.28 -> 2.38, 384 rows 1.25 -> 1.35.  Part of the gain at 1024 is the w2 GEMM's
        // whole tiles filling exactly two rounds of the chip; below ~4 whole tiles per expert the main launch fills its
        // last round badly on its own and the ragged tiles were hiding in that slack: from 1024)
        const bool two_parts = p.total_rows >= (int64_t) 1024 * p.num_experts && !lab_env("MI355Q_MOE_ONE_PART");
        const int tm16 = (int) (two_parts ? p.total_rows / 256 : (p.total_rows + 255) / 256 + p.num_experts), tn16 = a.n / 256;
        a.pf_group_m = lab_env("MI355Q_PF_GROUP_M") ? atoi(lab_env("MI355Q_PF_GROUP_M")) : 4; // (tiles of one expert are adjacent m-tiles)
        a.moe_full_only = two_parts ? 1 : 0;
        // (the exact dequantisation here: this tile is MFMA-bound, and the integer-domain form's extra MFMAs against ones
        // cost it 7 % -- 4096 tokens, fc1 1.86 vs 2.00 ms, tools/lab_moe_intd.py)
        if (tm16 > 0)
        {
            if (bf)
                hipLaunchKernelGGL((woq_prefill_kernel<bf16_t, 4, false, 1, 16, 2, 2, 8>), dim3(tm16 * tn16, 1), dim3(512), 0, stream,
                    a, tm16, tn16);
            else
                hipLaunchKernelGGL((woq_prefill_kernel<f16_t, 4, false, 1, 16, 2, 2, 8>), dim3(tm16 * tn16, 1), dim3(512), 0, stream,
                    a, tm16, tn16);
        }
        if (two_parts)
        {
            a.moe_full_only = 0;
            a.pf_group_m = 0; // (groups of 2 / 4 / 8 for the tail tiles measured: nothing, 2950 vs 2940 us for the layer)
            a.moe_tail_mod = 256; // (the kernel's tail indexing is written for 256)
            // (64-row tail tiles: 128-row ones, two per expert, measured slower -- layer at 4096 tokens 2.96 vs 2.92 ms)
            const int tmt = p.num_experts * 4, tnt = a.n / 128;
            if (bf)
                hipLaunchKernelGGL((woq_prefill_kernel<bf16_t, 4, false, 1, 4, 2, 2, PF_WAVES, true>), dim3(tmt * tnt, 1),
                    dim3(PF_WAVES * 64), 0, stream, a, tmt, tnt);
            else
                hipLaunchKernelGGL((woq_prefill_kernel<f16_t, 4, false, 1, 4, 2, 2, PF_WAVES, true>), dim3(tmt * tnt, 1),
                    dim3(PF_WAVES * 64), 0, stream, a, tmt, tnt);
        }
        MQ_HIP(hipGetLastError());
        return 1;
    }
    const int tiles_m = (int) ((p.total_rows + mt * 16 - 1) / (mt * 16)) + p.num_experts;
    const int tiles_n = a.n / 128;
    // (the grouped tile walk below the 256-row tile: +4 % at 512 and 2048 tokens, -5 % at 1024 -- not taken)
    a.pf_group_m = lab_env("MI355Q_PF_GROUP_M_MOE") ? atoi(lab_env("MI355Q_PF_GROUP_M_MOE")) : 0;
#define MQ_MOE(T, WB, MTV)                                                                                             \
    do                                                                                                                 \
    {                                                                                                                  \
        if (!lab_env("MI355Q_MOE_EXACT"))                                                                              \
            hipLaunchKernelGGL((woq_prefill_kernel<T, WB, false, 1, MTV, 2, (MTV <= 2 ? 4 : 2), PF_WAVES, true>),      \
                dim3(tiles_m * tiles_n, a.ksplit), dim3(PF_WAVES * 64), 0, stream, a, tiles_m, tiles_n);               \
        else if (MTV <= 2 && !lab_env("MI355Q_PF_DEPTH2"))                                                             \
            hipLaunchKernelGGL((woq_prefill_kernel<T, WB, false, 1, MTV, 2, (MTV <= 2 ? 4 : 2)>), dim3(tiles_m * tiles_n, a.ksplit), \
                dim3(PF_WAVES * 64), 0, stream, a, tiles_m, tiles_n);                                                  \
        else                                                                                                           \
            hipLaunchKernelGGL((woq_prefill_kernel<T, WB, false, 1, MTV, 2>), dim3(tiles_m * tiles_n, a.ksplit),       \
                dim3(PF_WAVES * 64), 0, stream, a, tiles_m, tiles_n);                                                  \
    } while (0)
#define MQ_MOE_MT(T, WB)                                                                                               \
    do                                                                                                                 \
    {                                                                                                                  \
        if (mt == 2)                                                                                                   \
            MQ_MOE(T, WB, 2);                                                                                          \
        else if (mt == 4)                                                                                              \
            MQ_MOE(T, WB, 4);                                                                                          \
        else                                                                                                           \
            MQ_MOE(T, WB, 8);                                                                                          \
    } while (0)
    if (bf && i4) MQ_MOE_MT(bf16_t, 4);
    else if (bf) MQ_MOE_MT(bf16_t, 8);
    else if (i4) MQ_MOE_MT(f16_t, 4);
    else MQ_MOE_MT(f16_t, 8);
#undef MQ_MOE_MT
#undef MQ_MOE
    if (a.ksplit > 1 && reduce) // rows past the last expert's hold whatever the workspace held: nobody reads them
        woq_splitk_reduce(a.slabs, nullptr, nullptr, p.C, (int) p.total_rows, (int) p.gemm_n, a.ksplit, bf, stream);
    MQ_HIP(hipGetLastError());
    return a.ksplit;
}

} // namespace mi355q
