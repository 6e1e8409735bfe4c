// preprocess.cpp -- host-side weight tooling behind the thop `trtllm::*` ops
// (cpp/tensorrt_llm/thop/weightOnlyQuantOp.cpp:116-343) and
// cpp/tensorrt_llm/kernels/cutlass_kernels/cutlass_preprocessors.cpp:161-729.
//
// The reference selects the physical weight layout by GPU architecture (SM80: permuted + column
// interleaved + biased; SM90: plain column-major).  gfx950 gets its own: MFMA-fragment-major tiles
// (see DESIGN.md "GFX950 weight layout").  Every layout is expressed here as ONE index map
// (k, n) -> (element slot, value transform), used for both directions, so reference-preprocessed
// checkpoints can be ingested (unprocess REF_SM80 -> ROW_MAJOR -> GFX950).
#include "common.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace mi355q
{

namespace
{

struct Geometry
{
    size_t K, N;
    int bits;
};

inline int get_elt(const uint8_t* p, size_t idx, int bits)
{
    if (bits == 8)
        return p[idx];
    uint8_t b = p[idx >> 1];
    return (idx & 1) ? (b >> 4) : (b & 0xf);
}

inline void set_elt(uint8_t* p, size_t idx, int bits, int v)
{
    if (bits == 8)
    {
        p[idx] = (uint8_t) v;
        return;
    }
    uint8_t& b = p[idx >> 1];
    b = (idx & 1) ? (uint8_t) ((b & 0x0f) | ((v & 0xf) << 4)) : (uint8_t) ((b & 0xf0) | (v & 0xf));
}

// ---- REF_SM80: where does logical element (k, n) live, in element units?
struct Sm80Map
{
    int bits, ew, rpg, interleave, vt;
    size_t K, N, vr_total;
    std::vector<int> inv_perm; // source row (within group) -> destination row
    explicit Sm80Map(const Geometry& g)
        : bits(g.bits)
        , K(g.K)
        , N(g.N)
    {
        ew = 32 / bits;           // elements per 32-bit word
        rpg = 8 * (16 / bits);    // rows per permutation group (16 int8 / 32 int4)
        interleave = bits == 8 ? 2 : 4;
        vt = 64 / ew;             // words per 64-row tile
        vr_total = K / ew;
        inv_perm.resize(rpg);
        for (int t = 0; t < rpg; ++t)
        {
            int src = 8 * ((t % ew) / 2) + t % 2 + 2 * (t / ew); // cutlass_preprocessors.cpp:209-210
            inv_perm[src] = t;
        }
    }
    size_t slot(size_t k, size_t n) const
    {
        size_t r = k / rpg * rpg + inv_perm[k % rpg];           // after the row permutation
        size_t vr = r / ew;                                      // word row in the column-major view
        int e = (int) (r % ew);
        size_t word = (n / interleave) * (vr_total * interleave) + interleave * (vr / vt * vt)
            + vt * (n % interleave) + vr % vt;                   // :516-527
        int d = bits == 8 ? (e == 1 ? 2 : (e == 2 ? 1 : e))      // [e3 e2 e1 e0] -> [e3 e1 e2 e0]
                          : ((e & 1) ? 4 + (e >> 1) : (e >> 1)); // [e7..e0] -> [e7 e5 e3 e1 e6 e4 e2 e0]
        return word * ew + d;
    }
};

// ---- GFX950: MFMA-fragment-major tiles.
//   n-block = 16 columns; the k-tiles of an n-block are contiguous ("tile run").
//   int4: k-tile = 128 rows = 1 KiB = 64 lanes x 16 B.  Lane l = h*16 + c owns column c and, in
//         dword s (k-step s of 32 rows), the 8 consecutive rows k = 8h .. 8h+7 -- the B operand
//         fragment of v_mfma_f32_16x16x32_{bf16,f16}: one 16-byte load per lane feeds 4 MFMAs with
//         no shuffle.  Row 8h+j sits at nibble j < 4 ? 2j : 2(j-4)+1 and is stored biased (q + 8),
//         so `w & 0x0f0f0f0f` / `(w >> 4) & 0x0f0f0f0f` are rows 0..3 / 4..7 as bytes 0..15, which
//         v_cvt_scalef32_pk_{bf16,f16}_fp8 turns into 16-bit floats two at a time (woq_gemm.hip).  When K % 128 == 64 the run
//         ends with a 512-byte half tile (64 rows: lane l owns 8 B at l*8, dwords s = 0, 1).
//   int8: k-tile = 64 rows = 1 KiB.  Lane l owns dwords (s*2 + j/4), byte j%4 for k-step s = 0, 1.
//   int8 values stay signed two's complement; byte count == the row-major tensor's.
struct Gfx950Map
{
    int bits;
    size_t K, N, run_bytes;
    explicit Gfx950Map(const Geometry& g)
        : bits(g.bits)
        , K(g.K)
        , N(g.N)
    {
        run_bytes = K * 16 * bits / 8; // one n-block: K rows x 16 columns
    }
    size_t slot(size_t k, size_t n) const
    {
        const size_t nb = n / 16, c = n % 16;
        const size_t base = nb * run_bytes;
        if (bits == 4)
        {
            const size_t kt = k / 128, r = k % 128;
            const size_t s = r / 32, h = (r % 32) / 8, j = r % 8;
            const size_t lane = h * 16 + c;
            const bool half = kt == K / 128; // trailing 64-row half tile
            const size_t byte = base + kt * 1024 + (half ? lane * 8 : lane * 16) + s * 4;
            return byte * 2 + (j < 4 ? 2 * j : 2 * (j - 4) + 1);
        }
        const size_t kt = k / 64, r = k % 64;
        const size_t s = r / 32, h = (r % 32) / 8, j = r % 8;
        const size_t lane = h * 16 + c;
        return base + kt * 1024 + lane * 16 + (s * 2 + j / 4) * 4 + j % 4;
    }
};

void check_shape(const size_t shape[3], int weight_type, int layout)
{
    MQ_CHECK(weight_type == MI355Q_WEIGHT_INT8 || weight_type == MI355Q_WEIGHT_INT4, "Invalid quant_type %d",
        weight_type);
    const int bits = weight_type == MI355Q_WEIGHT_INT8 ? 8 : 4;
    const size_t K = shape[1], N = shape[2];
    MQ_CHECK(shape[0] >= 1 && K >= 1 && N >= 1, "empty shape");
    switch (layout)
    {
    case MI355Q_LAYOUT_ROW_MAJOR: MQ_CHECK(N * bits % 8 == 0, "N must fill whole bytes"); break;
    case MI355Q_LAYOUT_REF_SM80:
        // cutlass_preprocessors.cpp:190-195, 258-261, 498-505
        MQ_CHECK(K % (8 * (16 / bits)) == 0 && K % 64 == 0,
            "Number of rows of quantized matrix must be a multiple of 64 (got %zu)", K);
        MQ_CHECK(N % 8 == 0 && (N * bits / 8) % 32 == 0 && (K * bits / 8) % 32 == 0,
            "Number of bytes for rows and cols must be a multiple of 32 (K=%zu N=%zu)", K, N);
        break;
    case MI355Q_LAYOUT_REF_SM90:
        MQ_CHECK((N * bits / 8) % 32 == 0 && (K * bits / 8) % 32 == 0,
            "Number of bytes for rows and cols must be a multiple of 32 (K=%zu N=%zu)", K, N);
        break;
    case MI355Q_LAYOUT_GFX950:
        MQ_CHECK(K % 64 == 0 && N % 16 == 0, "GFX950 layout needs K %% 64 == 0 and N %% 16 == 0 (K=%zu N=%zu)", K, N);
        break;
    default: MQ_CHECK(false, "unknown layout %d", layout);
    }
}

// forward == true : src ROW_MAJOR -> dst `layout`;  false: src `layout` -> dst ROW_MAJOR
void transform(int8_t* dst8, const int8_t* src8, const size_t shape[3], int weight_type, int layout, bool forward)
{
    check_shape(shape, weight_type, layout);
    const int bits = weight_type == MI355Q_WEIGHT_INT8 ? 8 : 4;
    const size_t E = shape[0], K = shape[1], N = shape[2];
    const size_t mat_bytes = K * N * bits / 8;
    const Geometry g{K, N, bits};
    const int mask = bits == 8 ? 0xff : 0xf;
    // signed -> unsigned: SM80 both widths (:386-432); GFX950 int4 only (u = q + 8 feeds the fp8-style unpack)
    const int bias = layout == MI355Q_LAYOUT_REF_SM80 ? (bits == 8 ? 128 : 8) : (layout == MI355Q_LAYOUT_GFX950 && bits == 4 ? 8 : 0);
    const Sm80Map sm80(g);
    const Gfx950Map gfx(g);
    std::vector<uint8_t> tmp(mat_bytes);
    for (size_t e = 0; e < E; ++e)
    {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(src8) + e * mat_bytes;
        uint8_t* dst = reinterpret_cast<uint8_t*>(dst8) + e * mat_bytes;
        if (layout == MI355Q_LAYOUT_ROW_MAJOR)
        {
            std::memmove(dst, src, mat_bytes);
            continue;
        }
        std::memset(tmp.data(), 0, mat_bytes);
        for (size_t k = 0; k < K; ++k)
            for (size_t n = 0; n < N; ++n)
            {
                size_t rm = k * N + n, slot;
                switch (layout)
                {
                case MI355Q_LAYOUT_REF_SM80: slot = sm80.slot(k, n); break;
                case MI355Q_LAYOUT_REF_SM90: slot = n * K + k; break;
                default: slot = gfx.slot(k, n); break;
                }
                if (forward)
                    set_elt(tmp.data(), slot, bits, (get_elt(src, rm, bits) + bias) & mask);
                else
                    set_elt(tmp.data(), rm, bits, (get_elt(src, slot, bits) - bias) & mask);
            }
        std::memcpy(dst, tmp.data(), mat_bytes);
    }
}

} // namespace

void preprocess_weights(int8_t* dst, const int8_t* src, const size_t shape[3], int weight_type, int layout)
{
    MQ_CHECK(dst && src && shape, "null argument");
    transform(dst, src, shape, weight_type, layout, true);
}

void unprocess_weights(int8_t* dst, const int8_t* src, const size_t shape[3], int weight_type, int layout)
{
    MQ_CHECK(dst && src && shape, "null argument");
    transform(dst, src, shape, weight_type, layout, false);
}

void unpack_int4(int8_t* dst, const int8_t* src, size_t n)
{
    MQ_CHECK(dst && src, "null argument");
    for (size_t i = 0; i < n; ++i)
    {
        int8_t b = src[i];
        dst[2 * i] = (int8_t) ((int8_t) (b << 4) >> 4);
        dst[2 * i + 1] = (int8_t) (b >> 4);
    }
}

void pack_int4(int8_t* dst, const int8_t* src, size_t n)
{
    MQ_CHECK(dst && src, "null argument");
    for (size_t i = 0; i < n; ++i)
    {
        int a = src[2 * i], b = src[2 * i + 1];
        MQ_CHECK(a >= -8 && a <= 7 && b >= -8 && b <= 7, "Value in unpacked tensor not in int4 range");
        dst[i] = (int8_t) ((a & 0x0f) | (b << 4));
    }
}

void symmetric_quantize(int8_t* processed, int8_t* unprocessed, float* scales, const float* weight,
    const size_t shape[3], int weight_type, int layout)
{
    MQ_CHECK(processed, "Processed quantized tensor is NULL");
    MQ_CHECK(scales, "Scale output pointer is NULL");
    MQ_CHECK(weight, "Input weight pointer is NULL");
    check_shape(shape, weight_type, layout);
    const int bits = weight_type == MI355Q_WEIGHT_INT8 ? 8 : 4;
    const size_t E = shape[0], K = shape[1], N = shape[2];
    const size_t row_bytes = N * bits / 8;
    std::vector<int8_t> own;
    if (!unprocessed)
    {
        own.resize(E * K * row_bytes);
        unprocessed = own.data();
    }
    std::memset(unprocessed, 0, E * K * row_bytes);
    const float range = 1.f / (float) (1 << (bits - 1));
    const float lo = bits == 8 ? -128.f : -8.f, hi = bits == 8 ? 127.f : 7.f;
    std::vector<float> col_scale(N);
    for (size_t e = 0; e < E; ++e)
    {
        const float* w = weight + e * K * N;
        uint8_t* q = reinterpret_cast<uint8_t*>(unprocessed) + e * K * row_bytes;
        std::fill(col_scale.begin(), col_scale.end(), 0.f);
        for (size_t k = 0; k < K; ++k)
            for (size_t n = 0; n < N; ++n)
                col_scale[n] = std::max(col_scale[n], std::fabs(w[k * N + n]));
        for (size_t n = 0; n < N; ++n)
        {
            col_scale[n] *= range;
            scales[e * N + n] = col_scale[n];
        }
        for (size_t k = 0; k < K; ++k)
            for (size_t n = 0; n < N; ++n)
            {
                float s = std::round(w[k * N + n] / col_scale[n]); // half away from zero, :692,:708
                if (s != s)
                    s = 0.f;
                s = std::max(lo, std::min(hi, s));
                set_elt(q, k * N + n, bits, (int) s & (bits == 8 ? 0xff : 0xf));
            }
    }
    transform(processed, unprocessed, shape, weight_type, layout, true);
}

} // namespace mi355q
