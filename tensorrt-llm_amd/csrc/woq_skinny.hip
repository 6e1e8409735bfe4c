// woq_skinny.hip -- the skinny form of the weight-only GEMM (16 < m <= 64 at decode-sized weights): one burst of loads
// per workgroup, int4 in the integer domain, split-K closed inside the launch.  Replaces, for those shapes, the path
// through kernels/cutlass_kernels/fpA_intB_gemm/fpA_intB_gemm.h:54-137 (CutlassFpAIntBGemmRunner::gemm with its
// split-k workspace).  DESIGN.md 4.3c.
#include "woq_prefill_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace mi355q
{

namespace
{

// ---------------------------------------------------------------------------------------------------------------
// The skinny form: 16 < m <= 64 rows against decode-sized weights (a serving step of a few dozen sequences, the
// tensor-parallel slices of cfg4).  Such a launch is a weight stream with a little MFMA work attached, and the
// ping-pong above, cut into 4-tile pieces over K to fill the chip, spends one exposed HBM round trip per tile and a
// second launch on the slab reduce.  Here a workgroup is ONE piece of P k-tiles of one TM x 128 output tile and
// issues EVERYTHING it will ever read in one burst -- the activation tiles by LDS-DMA first, then the P weight
// fragments and their scales -- so a workgroup's life is one round trip, P tiles of MFMAs, one store.  Split-K is
// closed inside the launch like the light GEMV's (woq_gemv_small.hip): fp32 partial tiles written through, one
// wrapping increment on the tile's arrival word, and the workgroup whose arrival completes the count sums the pieces
// in piece order (deterministic) and runs the epilogue.
// Issue order matters for the waits: the DMAs (invisible to hipcc) all come before the weight loads, vmcnt retires in
// order, so "tile t's weights have landed" -- the wait hipcc places itself -- implies this wave's DMAs have; one
// barrier after the first tile's weights then covers every wave's share of every activation tile.
template <typename T, int WBITS, bool ZERO, int NSZ, int MT, int P>
__global__ __launch_bounds__(256) void woq_skinny_kernel(
    WoqArgs a, int tiles_m, int tiles_n, int pieces, float* partials, uint32_t* arrive, unsigned long long* stamps)
{
#ifdef MI355Q_LAB
#define SK_STAMP(i)                                                                                                    \
    do                                                                                                                 \
    {                                                                                                                  \
        if (stamps && threadIdx.x == 0)                                                                                \
            stamps[(size_t) blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime();                                \
    } while (0)
#else
#define SK_STAMP(i)                                                                                                    \
    do                                                                                                                 \
    {                                                                                                                  \
    } while (0)
#endif
    SK_STAMP(0);
    constexpr int NB = 2, NW = 4;
    constexpr int KTILE = Geo<WBITS>::KTILE, KSTEPS = Geo<WBITS>::KSTEPS;
    constexpr int TM = MT * 16;
    constexpr int RB = KTILE * 2, CHUNKS = RB / 16, ROWS_PER_DMA = 1024 / RB;
    constexpr int BUF_BYTES = TM * RB;
    constexpr int DMA_PER_TILE = BUF_BYTES / 1024, DMA_PER_WAVE = DMA_PER_TILE / NW;
    static_assert(DMA_PER_TILE % NW == 0 && DMA_PER_WAVE >= 1, "activation tile does not split over the waves");
    static_assert(P * BUF_BYTES <= 64 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(1024))) uint8_t lds[P * BUF_BYTES];
    __shared__ float s1tab[WBITS == 4 ? P * NSZ * MT * 16 : 1]; // integer domain: activation sums per (tile, group, m-block, row)
    __shared__ int last_flag;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = tiles_m * tiles_n, nwg = tiles * pieces;
    // XCD x (= blockIdx.x % 8) takes a contiguous range of (tile, piece), piece fastest: the pieces of one output tile
    // -- and the m-tiles over one weight panel -- meet in one L2, which is where the partial tiles are exchanged
    int lin = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, x = lin % 8;
        lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + lin / 8;
    }
    const int tile = lin / pieces, piece = lin % pieces;
    const int tm = tile % tiles_m, tn = tile / tiles_m;
    const int m0 = tm * TM, n0 = tn * (NB * 16 * NW);
    const int kt0 = piece * P;

    auto sw = [](int row) { return CHUNKS == 16 ? (row & 15) : ((row >> 1) & 7); };
    const int drow = lane / CHUNKS, dchunk = lane % CHUNKS;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t) (uintptr_t) (lds_void_t*) lds);
    // ---- the burst, part 1: activation tiles (tiles past K re-read the last one; never computed)
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i)
    {
        const int d = wave * DMA_PER_WAVE + i;
        const int row = d * ROWS_PER_DMA + drow;
        const uint32_t aoff = (uint32_t) min(m0 + row, a.m - 1) * (uint32_t) a.k * 2u + (dchunk ^ sw(row)) * 16;
#pragma unroll
        for (int t = 0; t < P; ++t)
            pf_dma16(a.act, aoff + (uint32_t) min(kt0 + t, a.ktiles - 1) * (KTILE * 2), lds_base + t * BUF_BYTES + d * 1024);
    }
    // ---- part 2: this wave's two n-blocks of every tile, then the scales / zeros
    const size_t run_bytes = (size_t) a.k * 16 * WBITS / 8;
    const int nb0 = (n0 >> 4) + wave * NB;
    const uint8_t* run0 = a.qweight + (size_t) nb0 * run_bytes;
    const int col0 = nb0 * 16 + (lane & 15);
    uint4 w[P][NB];
#pragma unroll
    for (int t = 0; t < P; ++t)
    {
        const int kt = min(kt0 + t, a.ktiles - 1);
#pragma unroll
        for (int b = 0; b < NB; ++b)
            w[t][b] = *reinterpret_cast<const uint4*>(run0 + (size_t) b * run_bytes + (size_t) kt * 1024 + lane * 16);
    }
    // scales / zeros: a vector-memory instruction costs the CU the same ~16 cycles whether it moves 2 bytes or 1 KiB
    // (DESIGN 4.2), and one load per (tile, n-block, group) was half of all this wave issues; they are fetched for
    // several (tile, group) units per instruction and handed round by ds_bpermute when the tile is computed.
    constexpr bool INTD = WBITS == 4; // int4: integer-domain arithmetic (below); int8: the exact dequantisation
    constexpr int UNITS = P * NSZ, NI = INTD ? 1 : UNITS / 4;
    static_assert(UNITS % 4 == 0 && UNITS <= 16, "units per load instruction");
    // exact path: the lane needs column lane & 15 (an A-operand row); row q = lane >> 4 fetches unit 4 i + q
    uint32_t szp[NI][NB]; // scale | zero << 16
    // integer domain: the lane needs the four output rows 4 q .. 4 q + 3 of the n-block (D layout), the same for every
    // lane of its row; lane c of the row fetches unit c (8 bytes of scales, 8 of zeros)
    uint2 sd[NB], zd[NB];
    if constexpr (INTD)
    {
        const int u = min(lane & 15, UNITS - 1);
        const int kt = min(kt0 + u / NSZ, a.ktiles - 1);
        const size_t row = (size_t) ((kt * KTILE + (u % NSZ) * 64) >> a.group_shift) * a.n;
#pragma unroll
        for (int b = 0; b < NB; ++b)
        {
            const size_t si = row + (size_t) (nb0 + b) * 16 + 4 * (lane >> 4);
            sd[b] = *reinterpret_cast<const uint2*>(a.scales + si);
            zd[b] = ZERO ? *reinterpret_cast<const uint2*>(a.zeros + si) : make_uint2(0u, 0u);
        }
    }
    else
    {
#pragma unroll
        for (int i = 0; i < NI; ++i)
        {
            const int u = 4 * i + (lane >> 4);
            const int kt = min(kt0 + u / NSZ, a.ktiles - 1);
            const size_t row = (size_t) ((kt * KTILE + (u % NSZ) * 64) >> a.group_shift) * a.n;
#pragma unroll
            for (int b = 0; b < NB; ++b)
            {
                const uint32_t sv = a.scales[row + col0 + b * 16];
                const uint32_t zv = ZERO ? a.zeros[row + col0 + b * 16] : (uint16_t) 0;
                szp[i][b] = sv | (zv << 16);
            }
        }
    }
    SK_STAMP(1);
    f32x4_t acc[NB][MT];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < MT; ++j)
            acc[b][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fk = lane >> 4;
    const int fsw = sw(frow);
    const uint32_t frag_base = (uint32_t) (uintptr_t) (lds_void_t*) lds + frow * RB;

    // the first tile's weights have landed => this wave's DMAs have (issued earlier, retired in order)
#pragma unroll
    for (int b = 0; b < NB; ++b)
        asm volatile("" : "+v"(w[0][b].x), "+v"(w[0][b].y), "+v"(w[0][b].z), "+v"(w[0][b].w));
    SK_STAMP(2);
    __syncthreads();
    SK_STAMP(3);

    if constexpr (INTD)
    {
        // Integer domain, like the decode GEMV (woq_gemm.hip): the nibbles go to the MFMA as the integers u = 0 .. 15
        // (7 vector-ALU operations per 8 weights instead of the exact dequantisation's 23 -- at these heights a
        // dequantised fragment feeds MT <= 4 MFMAs and the 23 were the bound, tools/lab_skinny_stamps.py), and the
        // group's scale and zero meet the accumulator once per group: acc += s (P - 8 S1) + z S1, P = sum a u,
        // S1 = sum a (in this order a one-hot activation row gives fma(q, s, z) itself: the identity-activation gate
        // stays bit-exact, see gemv_tile_int4).  S1 depends on the activations only: wave j computes it for m-block j of every (tile, group)
        // with one MFMA against ones per k-step, through a small LDS table.
        constexpr int KG = KSTEPS / NSZ; // k-steps per group
        if (wave < MT)
        {
#pragma unroll
            for (int t = 0; t < P; ++t)
#pragma unroll
                for (int h = 0; h < NSZ; ++h)
                {
                    f32x4_t r = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int g = 0; g < KG; ++g)
                    {
                        const int st = h * KG + g;
                        const uint4 af = *reinterpret_cast<const uint4*>(
                            lds + t * BUF_BYTES + (wave * 16 + frow) * RB + (((st * 4 + fk) ^ fsw) << 4));
                        r = mfma16<T>(ones8<T>(), af, r);
                    }
                    if (lane < 16)
                        s1tab[((t * NSZ + h) * MT + wave) * 16 + lane] = r[0];
                }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < P; ++t)
        {
            if (kt0 + t >= a.ktiles) // wave-uniform
                break;
            pf_static_for<NSZ>(
                [&](auto hc)
                {
                    constexpr int h = decltype(hc)::value;
                    const int u = t * NSZ + h;
                    f32x4_t pacc[NB][MT];
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int j = 0; j < MT; ++j)
                            pacc[b][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    pf_static_for<KG>(
                        [&](auto gc)
                        {
                            constexpr int st = h * KG + decltype(gc)::value;
                            uint4 wf[NB];
#pragma unroll
                            for (int b = 0; b < NB; ++b)
                                wf[b] = unpack8_u4<T>(reinterpret_cast<const uint32_t*>(&w[t][b])[st]);
                            const uint32_t abase = frag_base + t * BUF_BYTES + ((uint32_t) ((st * 4 + fk) ^ fsw) << 4);
                            constexpr int AF = MT - 1 < 3 ? MT - 1 : 3, R = AF + 1;
                            pf_u32x4_t av[R];
                            pf_static_for<AF>(
                                [&](auto jc) { pf_lds_read<decltype(jc)::value * 16 * RB>(av[decltype(jc)::value % R], abase); });
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
                                        pacc[b][j] = mfma16<T>(wf[b], af, pacc[b][j]);
                                });
                        });
                    float s1[MT];
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        s1[j] = s1tab[(u * MT + j) * 16 + (lane & 15)];
                    const int src = ((lane & 48) + u) << 2;
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                    {
                        const uint32_t s01 = (uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) sd[b].x);
                        const uint32_t s23 = (uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) sd[b].y);
                        const uint32_t z01 = ZERO ? (uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) zd[b].x) : 0u;
                        const uint32_t z23 = ZERO ? (uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) zd[b].y) : 0u;
                        const float sv[4] = {T16<T>::lo(s01), T16<T>::hi(s01), T16<T>::lo(s23), T16<T>::hi(s23)};
                        const float zv[4] = {T16<T>::lo(z01), T16<T>::hi(z01), T16<T>::lo(z23), T16<T>::hi(z23)};
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < MT; ++j)
                                acc[b][j][i] = fmaf(sv[i], fmaf(-8.f, s1[j], pacc[b][j][i]), fmaf(zv[i], s1[j], acc[b][j][i]));
                    }
                });
        }
    }
    else
    {
#pragma unroll
    for (int t = 0; t < P; ++t)
    {
        if (kt0 + t >= a.ktiles) // wave-uniform
            break;
        float s[NB][NSZ], z[NB][NSZ];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int h = 0; h < NSZ; ++h)
            {
                const int u = t * NSZ + h; // compile-time after unrolling
                const uint32_t v = (uint32_t) __builtin_amdgcn_ds_bpermute(((u & 3) * 16 + (lane & 15)) << 2, (int) szp[u >> 2][b]);
                s[b][h] = T16<T>::to_float((uint16_t) (v & 0xffffu));
                z[b][h] = T16<T>::to_float((uint16_t) (v >> 16));
            }
        pf_static_for<KSTEPS>(
            [&](auto stc)
            {
                constexpr int st = decltype(stc)::value;
                constexpr int h = 0;
                uint4 wf[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    wf[b] = dequant_step<T, WBITS, st>(w[t][b], s[b][h], z[b][h]);
                const uint32_t abase = frag_base + t * BUF_BYTES + ((uint32_t) ((st * 4 + fk) ^ fsw) << 4);
                // activation fragments one LDS round trip ahead of their MFMAs (see compute() of the prefill kernel)
                constexpr int AF = MT - 1 < 3 ? MT - 1 : 3, R = AF + 1;
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
            });
    }
    }

    SK_STAMP(4);
#ifdef MI355Q_LAB
    if ((a.red_buffers & 2) && pieces > 1) // lab ablation: no publish / reduce
    {
        if (acc[0][0][0] == 1.2345f)
            a.out[0] = 1;
        return;
    }
#endif
    if (pieces > 1)
    {
        // ---- publish the fp32 tile (lane-linear: (b, j) -> 4 KiB, 16 bytes per thread), count, the last arrival sums
        float* mine = partials + ((size_t) tile * pieces + piece) * (TM * 128);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < MT; ++j)
            {
                const __amdgpu_buffer_rsrc_t pr = make_rsrc(mine, (size_t) TM * 128 * 4);
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(u32x4_t, acc[b][j]), pr, (uint32_t) ((b * MT + j) * 256 + tid) * 16, 0, 16 /* sc1 */);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        SK_STAMP(5);
        if (tid == 0)
        {
            typedef __attribute__((address_space(1))) uint32_t gu32_t;
            gu32_t* word = (gu32_t*) (arrive + (size_t) tile * ARRIVAL_WORD_STRIDE);
            last_flag = __builtin_amdgcn_atomic_inc32(word, (uint32_t) (pieces - 1), __ATOMIC_RELAXED, "agent")
                == (uint32_t) (pieces - 1);
        }
        __syncthreads();
        SK_STAMP(6);
        if (!last_flag)
            return;
        // acquire at agent scope (the arrival word was the release's other half), then ordinary loads: the pieces were
        // written through by workgroups that -- with the mapping above -- sit on this XCD, so the lines are in this L2;
        // were they not, the invalidate makes the loads fetch them from memory, where the write-through put them
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(partials + (size_t) tile * pieces * (TM * 128), (size_t) pieces * TM * 128 * 4);
        constexpr int RQ = 8; // loads in flight per (b, j)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int j = 0; j < MT; ++j)
            {
                f32x4_t sum = f32x4_t{0.f, 0.f, 0.f, 0.f};
                const uint32_t off = (uint32_t) ((b * MT + j) * 256 + tid) * 16;
                for (int p0 = 0; p0 < pieces; p0 += RQ)
                {
                    u32x4_t x[RQ];
#pragma unroll
                    for (int q = 0; q < RQ; ++q) // past the last piece: beyond the descriptor, reads 0
                        x[q] = __builtin_amdgcn_raw_buffer_load_b128(rr, off, (uint32_t) (p0 + q) * (TM * 128 * 4), 0);
#pragma unroll
                    for (int q = 0; q < RQ; ++q)
                        sum += __builtin_bit_cast(f32x4_t, x[q]);
                }
                acc[b][j] = sum;
            }
        SK_STAMP(7);
    }

    // epilogue.  D[row = n][col = m]: lane holds n = 4*(l>>4) + {0..3}, m = l & 15.
    const int em = lane & 15, en = (lane >> 4) * 4;
    const float alpha = a.alpha ? a.alpha[0] : 1.f; // out = alpha * acc + bias
#pragma unroll
    for (int b = 0; b < NB; ++b)
    {
        const int n = (nb0 + b) * 16 + en;
        float bias[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias)
        {
            const uint2 bv = *reinterpret_cast<const uint2*>(a.bias + n);
            bias[0] = T16<T>::lo(bv.x);
            bias[1] = T16<T>::hi(bv.x);
            bias[2] = T16<T>::lo(bv.y);
            bias[3] = T16<T>::hi(bv.y);
        }
#pragma unroll
        for (int j = 0; j < MT; ++j)
        {
            const int m = m0 + j * 16 + em;
            if (m >= a.m)
                continue;
            *reinterpret_cast<uint2*>(a.out + (size_t) m * a.n + n)
                = make_uint2(pack2<T>(woq_activation(acc[b][j][0] * alpha + bias[0], a.act_fn),
                                 woq_activation(acc[b][j][1] * alpha + bias[1], a.act_fn)),
                    pack2<T>(woq_activation(acc[b][j][2] * alpha + bias[2], a.act_fn),
                        woq_activation(acc[b][j][3] * alpha + bias[3], a.act_fn)));
        }
    }
}

#ifdef MI355Q_LAB
unsigned long long* g_skinny_stamps = nullptr;
#endif

struct SkinnyPlan
{
    bool applies;
    int mt, p, tiles_m, tiles_n, pieces;
    size_t workspace_bytes;
};

// Where the skinny form applies (tools/lab_skinny.py): 16 < m <= 64, whole k-tiles, 128-column tiles, and a launch that
// is ONE resident round of workgroups (<= 512: two per CU) with at most 16 pieces per tile -- beyond that the
// ping-pong's many rounds overlap their phases by themselves and the partial tiles cost more than its slab reduce.
// Always 32-row tiles (two m-tiles read the weight panel through one L2): the partial tile is 16 KiB and the last
// arrival fetches every piece of it in one round of loads.  8 k-tiles per piece when that still gives >= 200
// workgroups, else 4.
SkinnyPlan plan_skinny(int m, int n, int k, int wbits)
{
    SkinnyPlan s{};
    const int ktile = wbits == 4 ? 128 : 64;
    if (m <= 16 || m > 64 || n % 128 != 0 || k % ktile != 0 || k < ktile || lab_env("MI355Q_NO_SKINNY"))
        return s;
    s.mt = 2;
    s.tiles_m = (m + 31) / 32;
    s.tiles_n = n / 128;
    const int nk = k / ktile, tiles = s.tiles_m * s.tiles_n;
    s.p = tiles * ((nk + 7) / 8) >= 200 ? 8 : 4;
    bool forced = false;
    if (const char* e = lab_env("MI355Q_SKINNY_MT"))
        s.mt = atoi(e), s.tiles_m = (m + s.mt * 16 - 1) / (s.mt * 16), forced = true;
    if (const char* e = lab_env("MI355Q_SKINNY_P"))
        s.p = atoi(e), forced = true;
    if (s.mt == 4)
        s.p = 4;
    s.pieces = (nk + s.p - 1) / s.p;
    s.applies = forced || (s.tiles_m * s.tiles_n * s.pieces <= 512 && s.pieces <= 16);
    s.workspace_bytes = s.pieces > 1 ? (size_t) s.tiles_m * s.tiles_n * s.pieces * (s.mt * 16 * 128 * 4) : 0;
    return s;
}

template <typename T, int WBITS, bool ZERO, int NSZ>
void launch_skinny(const WoqArgs& a, const SkinnyPlan& s, float* partials, uint32_t* arrive, hipStream_t stream)
{
    const dim3 grid(s.tiles_m * s.tiles_n * s.pieces);
    unsigned long long* stamps = nullptr;
#ifdef MI355Q_LAB
    if (lab_env("MI355Q_SKINNY_STAMPS"))
    {
        if (!g_skinny_stamps)
        {
            MQ_HIP(hipHostMalloc((void**) &g_skinny_stamps, 8192 * 16 * sizeof(unsigned long long), hipHostMallocMapped));
            std::memset(g_skinny_stamps, 0, 8192 * 16 * sizeof(unsigned long long));
        }
        if (grid.x <= 8192)
            stamps = g_skinny_stamps;
    }
#endif
#define MQ_SKINNY(MTV, PV)                                                                                             \
    hipLaunchKernelGGL((woq_skinny_kernel<T, WBITS, ZERO, NSZ, MTV, PV>), grid, dim3(256), 0, stream, a, s.tiles_m,    \
        s.tiles_n, s.pieces, partials, arrive, stamps)
#ifdef MI355Q_LAB
    if (s.mt == 4) // (64-row tiles with 4 k-tiles: measured and left to the lab)
        MQ_SKINNY(4, 4);
    else
#endif
    if (s.p == 8)
        MQ_SKINNY(2, 8);
    else
        MQ_SKINNY(2, 4);
#undef MQ_SKINNY
}

} // namespace

#ifdef MI355Q_LAB
extern "C" const unsigned long long* mi355q_lab_skinny_stamps()
{
    return g_skinny_stamps;
}
#endif

size_t woq_skinny_workspace_size(int m, int n, int k, int weight_type)
{
    const SkinnyPlan s = plan_skinny(m, n, k, weight_type == MI355Q_WEIGHT_INT4 ? 4 : 8);
    return s.applies ? s.workspace_bytes : 0;
}

// true: launched.  false: the form does not apply (shape, or a workspace too small for the partial tiles)
bool weight_only_skinny_gemm(const WoqArgsHost& h, void* workspace, size_t workspace_bytes, hipStream_t stream)
{
    const SkinnyPlan s = plan_skinny(h.m, h.n, h.k, h.int4 ? 4 : 8);
    if (!s.applies || (uint64_t) h.m * (uint64_t) h.k * 2 >= (1ull << 32))
        return false;
    if (s.pieces > 1 && !(workspace && workspace_bytes >= s.workspace_bytes && ((uintptr_t) workspace & 255) == 0))
        return false;
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
    a.red_buffers = lab_env("MI355Q_SKINNY_ABL") ? atoi(lab_env("MI355Q_SKINNY_ABL")) : 0;
    float* partials = reinterpret_cast<float*>(workspace);
    uint32_t* arrive = s.pieces > 1 ? arrival_lines((size_t) s.tiles_m * s.tiles_n, stream) : nullptr;
    const bool bf = h.bf16, i4 = h.int4, z = h.zeros != nullptr;
    const bool split = i4 && h.group == 64;
    if (i4 && split)
    {
        if (bf && z) launch_skinny<bf16_t, 4, true, 2>(a, s, partials, arrive, stream);
        else if (bf) launch_skinny<bf16_t, 4, false, 2>(a, s, partials, arrive, stream);
        else if (z) launch_skinny<f16_t, 4, true, 2>(a, s, partials, arrive, stream);
        else launch_skinny<f16_t, 4, false, 2>(a, s, partials, arrive, stream);
    }
    else if (i4)
    {
        if (bf && z) launch_skinny<bf16_t, 4, true, 1>(a, s, partials, arrive, stream);
        else if (bf) launch_skinny<bf16_t, 4, false, 1>(a, s, partials, arrive, stream);
        else if (z) launch_skinny<f16_t, 4, true, 1>(a, s, partials, arrive, stream);
        else launch_skinny<f16_t, 4, false, 1>(a, s, partials, arrive, stream);
    }
    else
    {
        if (bf && z) launch_skinny<bf16_t, 8, true, 1>(a, s, partials, arrive, stream);
        else if (bf) launch_skinny<bf16_t, 8, false, 1>(a, s, partials, arrive, stream);
        else if (z) launch_skinny<f16_t, 8, true, 1>(a, s, partials, arrive, stream);
        else launch_skinny<f16_t, 8, false, 1>(a, s, partials, arrive, stream);
    }
    MQ_HIP(hipGetLastError());
    return true;
}

} // namespace mi355q
