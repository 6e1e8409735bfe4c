/*
 * oracle.c -- CPU restatement of the reference's integer / bit-level algorithms for the
 * quantized-inference hot path.  TEST INFRASTRUCTURE ONLY: nothing under tensorrt-llm_amd/
 * (the product) may link, load or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Pinning status: the reference's own implementation cannot be compiled here (it needs the
 * un-vendored 3rdparty/cutlass submodule, CUDA and TensorRT headers) and its Python cannot be
 * imported (needs `tensorrt`).  The restatement is therefore pinned by the known-answer vectors the
 * reference itself spells out (tests/test_oracle_golden.py):
 *   - row permutation maps            cutlass_preprocessors.cpp:157-160
 *   - register relayouts              cutlass_preprocessors.cpp:393-401, 434-441
 *   - fp8 scale KAT                   tests/quantization/test_fp8_quantization.py:45-80
 * Results none of those cover (e.g. full preprocess output on random data) are "parity unpinned"
 * beyond step-wise agreement with the maps above.
 *
 * All paths below are relative to /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_INT8 1 /* WeightTypeId::INT8, weightOnlyQuantMatmulPlugin.h:38-42 */
#define ORC_INT4 2

/* ---------------------------------------------------------------- 16-bit float helpers */

/* float -> bf16 bits, round-to-nearest-even, NaN stays NaN (quiet). */
static uint16_t f32_to_bf16_bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u)
        return (uint16_t) ((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t) (u >> 16);
}

static float bf16_bits_to_f32(uint16_t h)
{
    uint32_t u = (uint32_t) h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* float -> IEEE binary16 bits, round-to-nearest-even (handles subnormals / overflow). */
static uint16_t f32_to_f16_bits(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t absx = x & 0x7fffffffu;
    if (absx > 0x7f800000u)
        return (uint16_t) (sign | 0x7e00u);
    if (absx >= 0x47800000u) /* >= 65536 -> inf (65520 rounds to inf too, handled below) */
        return (uint16_t) (sign | 0x7c00u);
    if (absx < 0x33000000u) /* < 2^-25 -> 0 */
        return (uint16_t) sign;
    int32_t exp = (int32_t) (absx >> 23) - 127;
    uint32_t man = (absx & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t half;
    if (exp < -14)
    {
        shift = 13 + (-14 - exp); /* subnormal */
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        half = man >> shift;
        if (rem > halfway || (rem == halfway && (half & 1u)))
            half++;
        return (uint16_t) (sign | half);
    }
    shift = 13;
    {
        uint32_t rem = man & 0x1fffu;
        half = ((uint32_t) (exp + 15) << 10) | ((man >> 13) & 0x3ffu);
        if (rem > 0x1000u || (rem == 0x1000u && (half & 1u)))
            half++; /* carries into exponent correctly, up to inf */
    }
    return (uint16_t) (sign | half);
}

static float f16_bits_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t) h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t u;
    if (exp == 0)
    {
        if (man == 0)
            u = sign;
        else
        {
            int e = -1;
            do
            {
                man <<= 1;
                e++;
            } while (!(man & 0x400u));
            u = sign | ((uint32_t) (127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    }
    else if (exp == 31)
        u = sign | 0x7f800000u | (man << 13);
    else
        u = sign | ((exp + 112u) << 23) | (man << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* dtype codes follow nvinfer1::DataType: kHALF = 1, kBF16 = 7 */
static uint16_t to16(float f, int dtype)
{
    return dtype == 7 ? f32_to_bf16_bits(f) : f32_to_f16_bits(f);
}

static float from16(uint16_t h, int dtype)
{
    return dtype == 7 ? bf16_bits_to_f32(h) : f16_bits_to_f32(h);
}

void orc_f32_to_16(const float* src, uint16_t* dst, size_t n, int dtype)
{
    for (size_t i = 0; i < n; ++i)
        dst[i] = to16(src[i], dtype);
}

void orc_16_to_f32(const uint16_t* src, float* dst, size_t n, int dtype)
{
    for (size_t i = 0; i < n; ++i)
        dst[i] = from16(src[i], dtype);
}

/* ---------------------------------------------------------------- int4 pack / unpack */

/* cpp/tensorrt_llm/thop/weightOnlyQuantOp.cpp:274-306: low nibble = element 2i (sign-extended),
 * high nibble = element 2i+1 (arithmetic shift). */
void orc_unpack_int4(const int8_t* packed, int8_t* unpacked, size_t num_packed)
{
    for (size_t i = 0; i < num_packed; ++i)
    {
        uint8_t b = (uint8_t) packed[i];
        int lo = b & 0xf, hi = b >> 4;
        unpacked[2 * i] = (int8_t) (lo >= 8 ? lo - 16 : lo);
        unpacked[2 * i + 1] = (int8_t) (hi >= 8 ? hi - 16 : hi);
    }
}

/* cpp/tensorrt_llm/thop/weightOnlyQuantOp.cpp:308-343. Returns -1 when a value is outside [-8,7]
 * (the reference raises via TORCH_CHECK). */
int orc_pack_int4(const int8_t* unpacked, int8_t* packed, size_t num_packed)
{
    for (size_t i = 0; i < num_packed; ++i)
    {
        int a = unpacked[2 * i], b = unpacked[2 * i + 1];
        if (a < -8 || a > 7 || b < -8 || b > 7)
            return -1;
        packed[i] = (int8_t) ((a & 0xf) | ((b & 0xf) << 4));
    }
    return 0;
}

/* ---------------------------------------------------------------- element accessors on packed data */

static int get_elt(const uint8_t* base, size_t idx, int bits)
{
    if (bits == 8)
        return base[idx];
    uint8_t b = base[idx >> 1];
    return (idx & 1) ? (b >> 4) : (b & 0xf);
}

static void set_elt(uint8_t* base, size_t idx, int bits, int v)
{
    if (bits == 8)
    {
        base[idx] = (uint8_t) v;
        return;
    }
    uint8_t* p = base + (idx >> 1);
    if (idx & 1)
        *p = (uint8_t) ((*p & 0x0f) | ((v & 0xf) << 4));
    else
        *p = (uint8_t) ((*p & 0xf0) | (v & 0xf));
}

/* ---------------------------------------------------------------- SM80 layout pipeline, step by step */

/* Step 1. cutlass_preprocessors.cpp:161-222.  Rows are permuted inside groups of 16 (int8) or
 * 32 (int4) rows: destination row r of a group takes source row
 *   8*((r % E)/2) + r%2 + 2*(r / E),  E = elements per 32-bit register (4 or 8). */
void orc_permute_B_rows(const uint8_t* src, uint8_t* dst, size_t rows, size_t cols, int bits)
{
    int elts_per_reg = 32 / bits;
    size_t rows_per_group = (size_t) 8 * (16 / bits);
    size_t row_bytes = cols * bits / 8;
    for (size_t r = 0; r < rows; ++r)
    {
        size_t base = r / rows_per_group * rows_per_group, t = r % rows_per_group;
        size_t s = 8 * ((t % elts_per_reg) / 2) + t % 2 + 2 * (t / elts_per_reg);
        memcpy(dst + r * row_bytes, src + (base + s) * row_bytes, row_bytes);
    }
}

/* Step 2. cutlass_preprocessors.cpp:228-384: [rows, cols] row-major -> [cols, rows] row-major,
 * element-wise (nibble-aware).  The reference does it in 64x64 tiles for speed; the result is the
 * plain transpose. */
void orc_subbyte_transpose(const uint8_t* src, uint8_t* dst, size_t rows, size_t cols, int bits)
{
    memset(dst, 0, rows * cols * bits / 8);
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c)
            set_elt(dst, c * rows + r, bits, get_elt(src, r * cols + c, bits));
}

/* Step 3. cutlass_preprocessors.cpp:481-533 with LayoutDetailsB<..., Sm80> =
 * ColumnMajorTileInterleave<64, interleave> (mixed_gemm_B_layout.h:82-112): the column-major
 * tensor is viewed as 32-bit words; `interleave` adjacent columns are merged into one long "row",
 * alternating tiles of (64 rows) words. */
void orc_interleave_column_major(
    const uint32_t* src, uint32_t* dst, size_t rows, size_t cols, int bits, int rows_per_tile, int interleave)
{
    size_t elts_in_word = 32 / bits;
    size_t vec_rows = rows / elts_in_word;
    size_t vec_rows_per_tile = rows_per_tile / elts_in_word;
    for (size_t c = 0; c < cols; ++c)
    {
        size_t wc = c / interleave;
        for (size_t vr = 0; vr < vec_rows; ++vr)
        {
            size_t base = vr / vec_rows_per_tile * vec_rows_per_tile;
            size_t wr = interleave * base + vec_rows_per_tile * (c % interleave) + vr % vec_rows_per_tile;
            dst[wc * vec_rows * interleave + wr] = src[c * vec_rows + vr];
        }
    }
}

/* Step 4a. cutlass_preprocessors.cpp:386-408: +128 then swap bytes 1,2 of each 32-bit word
 * ([e3 e2 e1 e0] -> [e3 e1 e2 e0]). */
void orc_add_bias_and_interleave_int8(uint8_t* t, size_t num_elts)
{
    for (size_t i = 0; i < num_elts; ++i)
        t[i] = (uint8_t) (t[i] + 128);
    for (size_t b = 0; b + 3 < num_elts; b += 4)
    {
        uint8_t x = t[b + 1];
        t[b + 1] = t[b + 2];
        t[b + 2] = x;
    }
}

/* Step 4b. cutlass_preprocessors.cpp:410-463: +8 on every nibble (signed -> unsigned), then inside
 * each 32-bit word [e7..e0] -> [e7 e5 e3 e1 e6 e4 e2 e0]. */
void orc_add_bias_and_interleave_int4(uint8_t* t, size_t num_elts)
{
    size_t num_bytes = num_elts / 2;
    for (size_t i = 0; i < num_bytes; ++i)
    {
        int lo = ((t[i] & 0xf) + 8) & 0xf, hi = ((t[i] >> 4) + 8) & 0xf;
        t[i] = (uint8_t) (lo | (hi << 4));
    }
    for (size_t w = 0; w < num_bytes / 4; ++w)
    {
        uint32_t reg, out = 0;
        memcpy(&reg, t + 4 * w, 4);
        for (int d = 0; d < 8; ++d)
        {
            int s = d < 4 ? 2 * d : 2 * (d - 4) + 1;
            out |= ((reg >> (4 * s)) & 0xfu) << (4 * d);
        }
        memcpy(t + 4 * w, &out, 4);
    }
}

/* cutlass_preprocessors.cpp:535-584.  arch 80..89: permute + transpose + interleave(64, 2|4) + bias;
 * arch 90: transpose only; arch 70..74: transpose + bias (column-major, no interleave);
 * arch 75..79 behaves like 80 (mixed_gemm_B_layout.h Sm75 specialisation uses the same traits).
 * shape = [experts, rows(K), cols(N)].  Returns 0, or -1 on an unsupported shape / arch. */
int orc_preprocess_weights_for_mixed_gemm(
    const int8_t* src, int8_t* dst, size_t experts, size_t rows, size_t cols, int quant_type, int arch)
{
    int bits = quant_type == ORC_INT8 ? 8 : 4;
    size_t mat_bytes = rows * cols * bits / 8;
    int interleaved = arch >= 75 && arch < 90;
    int interleave = bits == 8 ? 2 : 4;
    if (quant_type != ORC_INT8 && quant_type != ORC_INT4)
        return -1;
    if (arch < 70 || arch > 90)
        return -1;
    if (interleaved && (rows % (8 * (16 / bits)) || cols % 8 || rows % 64))
        return -1;
    if ((rows * bits / 8) % 32 || (cols * bits / 8) % 32) /* transpose vector-width check, :258 */
        return -1;
    uint8_t* a = (uint8_t*) malloc(mat_bytes);
    uint8_t* b = (uint8_t*) malloc(mat_bytes);
    for (size_t e = 0; e < experts; ++e)
    {
        memcpy(a, src + e * mat_bytes, mat_bytes);
        if (interleaved)
        {
            orc_permute_B_rows(a, b, rows, cols, bits);
            memcpy(a, b, mat_bytes);
        }
        orc_subbyte_transpose(a, b, rows, cols, bits);
        memcpy(a, b, mat_bytes);
        if (interleaved)
        {
            orc_interleave_column_major((const uint32_t*) a, (uint32_t*) b, rows, cols, bits, 64, interleave);
            memcpy(a, b, mat_bytes);
        }
        if (arch < 90)
        {
            if (bits == 8)
                orc_add_bias_and_interleave_int8(a, rows * cols);
            else
                orc_add_bias_and_interleave_int4(a, rows * cols);
        }
        memcpy(dst + e * mat_bytes, a, mat_bytes);
    }
    free(a);
    free(b);
    return 0;
}

/* ---------------------------------------------------------------- symmetric quantizer */

/* cutlass_preprocessors.cpp:621-729.  weight [experts, rows, cols] float (the caller converts
 * fp16/bf16 to float first -- exact); per-column scale = absmax / 2^(bits-1) kept in float for the
 * division, returned in `scales` (caller rounds to the 16-bit type as `ComputeType(per_col_max)`
 * does).  q = clip(round(w / scale)) with C `round` (half away from zero).  NaN from 0/0 converts
 * to 0 here (int(NaN) is UB in the reference). */
void orc_symmetric_quantize(const float* weight, int8_t* unprocessed, float* scales, size_t experts, size_t rows,
    size_t cols, int quant_type)
{
    int bits = quant_type == ORC_INT8 ? 8 : 4;
    float qscale = 1.f / (float) (1 << (bits - 1));
    size_t bytes_per_row = cols * bits / 8;
    float lo = bits == 8 ? -128.f : -8.f, hi = bits == 8 ? 127.f : 7.f;
    for (size_t e = 0; e < experts; ++e)
    {
        const float* w = weight + e * rows * cols;
        float* sc = scales + e * cols;
        int8_t* q = unprocessed + e * rows * bytes_per_row;
        for (size_t c = 0; c < cols; ++c)
            sc[c] = 0.f;
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c)
            {
                float a = fabsf(w[r * cols + c]);
                if (a > sc[c])
                    sc[c] = a;
            }
        for (size_t c = 0; c < cols; ++c)
            sc[c] *= qscale;
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c)
            {
                float s = roundf(w[r * cols + c] / sc[c]);
                if (!(s == s))
                    s = 0.f;
                s = s < lo ? lo : (s > hi ? hi : s);
                int v = (int) s;
                if (bits == 8)
                    q[r * cols + c] = (int8_t) v;
                else
                    set_elt((uint8_t*) q, r * cols + c, 4, v & 0xf);
            }
    }
}

/* ---------------------------------------------------------------- dequantisation (the bit-exact gate) */

/* Single-rounding definition of the dequantised weight: the device paths compute
 * fma(q, scale, zero) with ONE rounding to the activation type
 * (kernels/weightOnlyBatchedGemv/kernel.h:364-365, mma_tensorop_dequantizer.h:244-252).
 * Here: w = round16( fmaf(float(q), scale, zero) ).  q*scale is exact in fp32 (<= 8+11 bits), so the
 * fp32 fma differs from an infinitely precise fused result only when |zero| < 2^-12 |q*scale| AND
 * q*scale sits exactly on a 16-bit rounding tie; the GPU kernels use the same v_fma_f32 + RNE.
 * q: int8 [K, N] (unpacked), scale/zero: 16-bit [K/group, N] (zero may be NULL), out: 16-bit [K, N].
 * group == K  <=> per-channel. */
void orc_dequant_fma(const int8_t* q, const uint16_t* scale, const uint16_t* zero, uint16_t* out, size_t K, size_t N,
    size_t group, int dtype)
{
    for (size_t k = 0; k < K; ++k)
        for (size_t n = 0; n < N; ++n)
        {
            float s = from16(scale[(k / group) * N + n], dtype);
            float z = zero ? from16(zero[(k / group) * N + n], dtype) : 0.f;
            out[k * N + n] = to16(fmaf((float) q[k * N + n], s, z), dtype);
        }
}

/* The torch test reference rounds twice: q.to(dtype) * scale, then += zero
 * (tests/quantization/test_weight_only_groupwise_quant_matmul.py:190-195). */
void orc_dequant_torch(const int8_t* q, const uint16_t* scale, const uint16_t* zero, uint16_t* out, size_t K, size_t N,
    size_t group, int dtype)
{
    for (size_t k = 0; k < K; ++k)
        for (size_t n = 0; n < N; ++n)
        {
            float s = from16(scale[(k / group) * N + n], dtype);
            float w = from16(to16((float) q[k * N + n] * s, dtype), dtype);
            if (zero)
                w = from16(to16(w + from16(zero[(k / group) * N + n], dtype), dtype), dtype);
            out[k * N + n] = to16(w, dtype);
        }
}

/* Activation smoothing, kernels/preQuantScaleKernel.cu:26-90 / torch.mul in the act dtype
 * (test_weight_only_groupwise_quant_matmul.py:205-207): one rounding of the exact product. */
void orc_apply_per_channel_scale(
    const uint16_t* act, const uint16_t* pre, uint16_t* out, size_t M, size_t K, int dtype)
{
    for (size_t m = 0; m < M; ++m)
        for (size_t k = 0; k < K; ++k)
            out[m * K + k] = to16(from16(act[m * K + k], dtype) * from16(pre[k], dtype), dtype);
}

/* C[M,N] (float) = A16[M,K] @ W16[K,N], fp32 inputs widened exactly, accumulation in double so the
 * oracle is order-independent.  Used as the ground truth for tolerance-gated GEMM outputs
 * (tests/quantization/_utils.py:63-67). */
void orc_matmul_16(const uint16_t* A, const uint16_t* W, float* C, size_t M, size_t N, size_t K, int dtype)
{
    float* wf = (float*) malloc(K * N * sizeof(float));
    for (size_t i = 0; i < K * N; ++i)
        wf[i] = from16(W[i], dtype);
#pragma omp parallel for
    for (size_t m = 0; m < M; ++m)
    {
        double* acc = (double*) calloc(N, sizeof(double));
        for (size_t k = 0; k < K; ++k)
        {
            double a = from16(A[m * K + k], dtype);
            const float* wr = wf + k * N;
            for (size_t n = 0; n < N; ++n)
                acc[n] += a * wr[n];
        }
        for (size_t n = 0; n < N; ++n)
            C[m * N + n] = (float) acc[n];
        free(acc);
    }
    free(wf);
}

/* ---------------------------------------------------------------- SmoothQuant int8 GEMM */

/* int32 accumulators of A[M,K] (row-major) x B[N,K] ("TN", smoothQuantGemmPlugin.cpp:224-230),
 * tests/quantization/_utils.py:112-118.  Exact. */
void orc_gemm_s8s8s32(const int8_t* A, const int8_t* B, int32_t* C, size_t M, size_t N, size_t K)
{
#pragma omp parallel for
    for (size_t m = 0; m < M; ++m)
        for (size_t n = 0; n < N; ++n)
        {
            const int8_t* a = A + m * K;
            const int8_t* b = B + n * K;
            int32_t acc = 0;
            for (size_t k = 0; k < K; ++k)
                acc += (int32_t) a[k] * (int32_t) b[k];
            C[m * N + n] = acc;
        }
}

/* Epilogue, epilogue_per_row_per_col_scale.h:320-347: out = cvt( float(acc) * (scale_col * scale_row) ),
 * scalars broadcast when not per-channel / per-token (:211-219).  out_f32 receives the fp32 product
 * (the caller casts: fp16 RNE, or rint for int32 -- _utils.py:132-134). */
void orc_sq_epilogue(const int32_t* acc, const float* scale_tok, const float* scale_ch, float* out_f32, size_t M,
    size_t N, int per_token, int per_channel)
{
    for (size_t m = 0; m < M; ++m)
        for (size_t n = 0; n < N; ++n)
        {
            float sr = scale_tok[per_token ? m : 0], sc = scale_ch[per_channel ? n : 0];
            out_f32[m * N + n] = (float) acc[m * N + n] * (sc * sr);
        }
}

/* kernels/quantization.cu:93-127 + _utils.py:145-150: per-row absmax (floored at T(1e-6),
 * quantization.cu:99 -- the caller passes 1e-6 rounded to the activation type), scale = max/127,
 * q = sat(rint(x * (127 / max))) -- the kernel multiplies by the reciprocal scale
 * `127.f / rowMax` (:111-114) and converts with cvt.rni.sat.s8.f32 (round-half-even, saturating). */
void orc_quantize_per_token(const float* x, int8_t* q, float* scale, size_t M, size_t K, float amax_floor)
{
    for (size_t m = 0; m < M; ++m)
    {
        float amax = amax_floor;
        for (size_t k = 0; k < K; ++k)
        {
            float a = fabsf(x[m * K + k]);
            if (a > amax)
                amax = a;
        }
        float s = 127.f / amax;
        for (size_t k = 0; k < K; ++k)
        {
            float v = nearbyintf(x[m * K + k] * s);
            v = v < -128.f ? -128.f : (v > 127.f ? 127.f : v);
            q[m * K + k] = (int8_t) v;
        }
        scale[m] = amax / 127.f;
    }
}

/* ---------------------------------------------------------------- decode attention (C / OpenMP)
 * The masked-MHA math of oracle.decode_attention (decoderMaskedMultiheadAttentionTemplate.h:1755-1813, 1988-2036,
 * 2246-2313, Logit_value_fma :954-999) for a 16-bit paged cache with NeoX RoPE, as plain loops: bench.py's cpu_baseline
 * leg times this one (the Python version spends its time in the interpreter).  pool: 16-bit elements; offsets
 * [B][2][max_blocks] element offsets of the K / V blocks ([Hkv][tokens_per_block][Dh] each, kvCacheUtils.h:101-152).
 * seq_lens include the current token; the new K/V are NOT appended (update_cache = False).  out: fp32 [B][Hq*Dh]. */
void orc_decode_attention_16(const uint16_t* qkv, const uint16_t* pool, const int64_t* offsets, const int32_t* seq_lens,
    float* out, int B, int Hq, int Hkv, int D, int max_blocks, int tpb, int dtype, int rope_neox, float q_scaling)
{
    const int grp = Hq / Hkv;
    const float inv_sqrt = 1.0f / (sqrtf((float) D) * q_scaling);
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < Hq; ++h)
        {
            const int hk = h / grp;
            const int tl = seq_lens[b] - 1;
            const uint16_t* row = qkv + (size_t) b * (Hq + 2 * Hkv) * D;
            float* q = (float*) malloc(sizeof(float) * 3 * D);
            float *k = q + D, *v = k + D;
            for (int d = 0; d < D; ++d)
            {
                q[d] = from16(row[h * D + d], dtype);
                k[d] = from16(row[(Hq + hk) * D + d], dtype);
                v[d] = from16(row[(Hq + Hkv + hk) * D + d], dtype);
            }
            if (rope_neox)
            {
                const int half = D / 2;
                for (int i = 0; i < half; ++i)
                {
                    const float ang = (float) tl / powf(10000.0f, (float) (2 * i) / (float) D);
                    const float c = cosf(ang), s = sinf(ang);
                    const float qa = q[i], qb = q[i + half], ka = k[i], kb = k[i + half];
                    q[i] = from16(to16(c * qa - s * qb, dtype), dtype);
                    q[i + half] = from16(to16(c * qb + s * qa, dtype), dtype);
                    k[i] = from16(to16(c * ka - s * kb, dtype), dtype);
                    k[i + half] = from16(to16(c * kb + s * ka, dtype), dtype);
                }
            }
            float* logit = (float*) malloc(sizeof(float) * (tl + 1));
            float mx = -INFINITY;
            for (int t = 0; t <= tl; ++t)
            {
                double acc = 0.0;
                if (t < tl)
                {
                    const uint16_t* kr = pool + offsets[((size_t) b * 2 + 0) * max_blocks + t / tpb]
                        + ((size_t) hk * tpb + t % tpb) * D;
                    for (int d = 0; d < D; ++d)
                        acc += (double) q[d] * (double) from16(kr[d], dtype);
                }
                else
                    for (int d = 0; d < D; ++d)
                        acc += (double) q[d] * (double) k[d];
                logit[t] = (float) acc * inv_sqrt;
                mx = logit[t] > mx ? logit[t] : mx;
            }
            double* o = (double*) calloc(D, sizeof(double));
            float sum = 0.f;
            for (int t = 0; t <= tl; ++t)
            {
                const float p = expf(logit[t] - mx);
                sum += p;
                if (t < tl)
                {
                    const uint16_t* vr = pool + offsets[((size_t) b * 2 + 1) * max_blocks + t / tpb]
                        + ((size_t) hk * tpb + t % tpb) * D;
                    for (int d = 0; d < D; ++d)
                        o[d] += (double) p * (double) from16(vr[d], dtype);
                }
                else
                    for (int d = 0; d < D; ++d)
                        o[d] += (double) p * (double) v[d];
            }
            for (int d = 0; d < D; ++d)
                out[((size_t) b * Hq + h) * D + d] = (float) o[d] / (sum + 1e-6f);
            free(o);
            free(logit);
            free(q);
        }
}
