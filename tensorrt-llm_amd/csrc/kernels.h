// kernels.h -- internal C++ interface between the plugin layer and the HIP kernels.
#pragma once
#include "common.h"

namespace mi355q
{

// preprocess.cpp (host)
void preprocess_weights(int8_t* dst, const int8_t* src, const size_t shape[3], int weight_type, int layout);
void unprocess_weights(int8_t* dst, const int8_t* src, const size_t shape[3], int weight_type, int layout);
void unpack_int4(int8_t* dst, const int8_t* src, size_t numPacked);
void pack_int4(int8_t* dst, const int8_t* src, size_t numPacked);
void symmetric_quantize(int8_t* processed, int8_t* unprocessed, float* scales, const float* weight,
    const size_t shape[3], int weight_type, int layout);

// woq_gemm.hip
int woq_choose_ksplit(int m, int n, int k, int weight_type);
size_t woq_gemm_workspace_size(int m, int n, int k, int weight_type);
void weight_only_gemm(const mi355q_weight_only_params& p, void* workspace, size_t workspace_bytes, hipStream_t stream);
bool woq_gemm_fuses_act_scale(int m, int n, int k, int weight_type, bool has_act_scale); // true: a decode kernel takes the call and applies p.act_scale itself (no separate smoothing pass)
void weight_only_dequantize(const mi355q_weight_only_params& p, hipStream_t stream);

// woq_gemv_small.hip: the light decode GEMV (m <= 16, one launch, in-launch split-K)
struct LightPlan
{
    bool applies;
    int ksplit, seg_tiles; // k segments per n-block, k-tiles per segment
    size_t workspace_bytes; // partial tiles + arrival words (split-K only)
};
// above these the persistent kernel of woq_gemm.hip (16-wave workgroups that stay resident, LDS activation slab) wins
constexpr size_t LIGHT_MAX_WEIGHT_BYTES = 160u << 20, LIGHT_MAX_WEIGHT_BYTES_16ROWS = 24u << 20;
LightPlan woq_light_plan(int m, int n, int k, int weight_type);
void weight_only_gemv_light(const mi355q_weight_only_params& p, LightPlan pl, void* workspace, size_t workspace_bytes,
    hipStream_t stream);

// arrival_ring.hip: zero-initialised, self-resetting arrival words of in-launch reductions, one per 128-byte line
constexpr int ARRIVAL_WORD_STRIDE = 32; // uint32 words between two arrival words
size_t arrival_lines_max();
// device pointer (current device) to `count` zero words, ARRIVAL_WORD_STRIDE apart, that no launch able to overlap with
// one on `stream` (another stream's, another captured graph's) counts on
uint32_t* arrival_lines(size_t count, hipStream_t stream);
void arrival_words_prepare(); // resolve the current device's array now (plugin configure), not inside the first enqueue

// woq_prefill.hip
struct WoqArgsHost
{
    const uint8_t* qweight;
    const uint16_t *scales, *zeros, *act, *bias;
    uint16_t* out;
    int m, n, k, group, group_shift, ktiles;
    bool bf16, int4;
    float* slabs; // fp32 [ksplit][m][n] workspace, or null (then no split-K)
    const float* alpha; // device [1] or null
};
bool woq_prefill_supported(int m, int n, int k);
int woq_prefill_ksplit(int m, int n, int k, int weight_type);
void woq_splitk_reduce(const float* slabs, const void* bias, const float* alpha, void* out, int m, int n, int ksplit,
    bool bf16, hipStream_t stream);
void weight_only_prefill_gemm(const WoqArgsHost& h, hipStream_t stream);
// the skinny form (16 < m <= 64, one burst per workgroup, split-K closed inside the launch); false = does not apply
size_t woq_skinny_workspace_size(int m, int n, int k, int weight_type);
bool weight_only_skinny_gemm(const WoqArgsHost& h, void* workspace, size_t workspace_bytes, hipStream_t stream);
// returns the split-K count used; reduce = false leaves the fp32 slabs in p.splitk_workspace ([splits][total_rows][gemm_n])
// for the caller to sum (run_moe's finalize step does), instead of launching the reduce kernel into p.C
// row_map: p.A holds the UN-permuted rows and row r of the expert order is row row_map[r] % row_mod of it (the layer's
// dest_to_src map: no gather pass); null = p.A is in expert order
// glu_fn >= 0: gemm_n = 2 inter columns [linear | gate]; p.C receives [total_rows, inter] = linear * act(gate) with the
// reference's roundings (doGatedActivationKernel), act = woq_activation code glu_fn (2 SiLU, 0 GELU); needs inter % 256 == 0
int moe_weight_only_gemm(const mi355q_moe_gemm_params& p, hipStream_t stream, bool reduce = true,
    const int32_t* row_map = nullptr, int row_mod = 1, int glu_fn = -1);
size_t moe_workspace_size(const mi355q_moe_params& p);
void run_moe(const mi355q_moe_params& p, hipStream_t stream);
void apply_per_channel_scale(
    void* out, const void* act, const void* scale, int rows, int cols, int act_type, hipStream_t stream);
// the W4A8 form: T(e4m3(T(act * scale))), scale may be null
void apply_per_channel_scale_fp8(
    void* out, const void* act, const void* scale, int rows, int cols, int act_type, hipStream_t stream);
// plugins.cpp: smoothing (if act_scale) + GEMM, carving p.workspace
void weight_only_gemm_entry(const mi355q_weight_only_params& p, hipStream_t stream);

// gemm8.hip
// workspace: gemm8_workspace_size(m, n, k) bytes let shapes whose 256 x 256 tiles do not fill whole rounds of the CUs
// run the persistent / stream-K kernel; null or smaller = one tile per workgroup (same results for int8)
size_t gemm8_workspace_size(int m, int n, int k);
size_t gemm8_workspace_size_max(int n, int k); // bound over every m
void int8_gemm(const int8_t* A, const int8_t* B, int per_token, int per_channel, const float* alphaCol,
    const float* alphaRow, void* C, int out_type, int m, int n, int k, void* workspace, size_t workspace_bytes,
    hipStream_t stream);
void fp8_gemm(const void* A, const void* B, const float* scaleA, const float* scaleB, void* C, int out_type, int m,
    int n, int k, void* workspace, size_t workspace_bytes, hipStream_t stream);
void per_token_quantization(
    int8_t* dst, const void* src, int64_t rows, int64_t cols, float* scales, int src_type, hipStream_t stream);
void quantize_e4m3(void* dst, float* scales, const void* src, int64_t rows, int64_t cols, int src_type, int mode,
    hipStream_t stream);

// norm_quant.hip
void rms_norm(void* out, const void* input, const void* gamma, const void* beta, float eps, int tokens, int hidden,
    const float* scale, float* dynamic_scale, int8_t* out_quant, int type, hipStream_t stream);
void layer_norm(void* out, const void* input, const void* gamma, const void* beta, float eps, int tokens, int hidden,
    bool use_diff_of_squares, const float* scale, float* dynamic_scale, int8_t* out_quant, int type, hipStream_t stream);
void quantize_tensor(int8_t* dst, const void* src, int64_t size, const float* scale, int type, hipStream_t stream);
bool row_quant_supported(int64_t cols);
void per_token_quantization_rows(int8_t* dst, const void* src, int64_t rows, int64_t cols, float* scales, int type, hipStream_t stream);
void gated_silu(const mi355q_gated_act_params& p, hipStream_t stream);

// allreduce.hip
void allreduce_params_deserialize(
    mi355q_allreduce_params& out, const void* const* buffer_ptrs, size_t tpSize, size_t tpRank, uint32_t flag_value);
void custom_all_reduce(mi355q_allreduce_params& params, void* data, size_t elts, int dtype, int strat, hipStream_t stream);
void custom_all_gather(mi355q_allreduce_params& params, const void* in, void* out, size_t bytes_per_rank, hipStream_t stream);
void multi_gpu_barrier_launch(const mi355q_allreduce_params& params, hipStream_t stream);
size_t allreduce_barrier_bytes();
void allreduce_alloc_timeout_flag(uint32_t** flag);
void allreduce_free_timeout_flag(uint32_t* flag);
void ipc_alloc(size_t bytes, void** ptr);
void ipc_free(void* ptr);
void ipc_get_handle(void* ptr, uint8_t* handle);
void ipc_open_handle(const uint8_t* handle, void** ptr);
void ipc_close_handle(void* ptr);
// rccl_comm.cpp
void comm_get_unique_id(uint8_t* id);
void comm_init_rank(const int32_t* group, int n, const uint8_t* id, int group_rank);
bool comm_exists(const int32_t* group, int n);
void comm_destroy(const int32_t* group, int n);
void comm_all_reduce(const int32_t* group, int n, const void* in, void* out, size_t count, int dtype, hipStream_t stream);
void comm_all_gather(const int32_t* group, int n, const void* in, void* out, size_t count, int dtype, hipStream_t stream);

// decode_attn.hip
size_t decode_attention_workspace_size(const mi355q_decode_attention_params& q);
void masked_multihead_attention(const mi355q_decode_attention_params& q, hipStream_t stream);
size_t decode_attention_static_scratch_bytes();
void masked_multihead_attention_static(const mi355q_decode_attention_params& q, hipStream_t stream);

// context_attn.hip
size_t context_attention_workspace_size(const mi355q_context_attention_params& q);
void context_attention(const mi355q_context_attention_params& q, hipStream_t stream);

} // namespace mi355q
