/*
 * mi355q.h -- C ABI of libmi355q.so: MI355X (gfx950) implementation of TensorRT-LLM's
 * quantized-inference hot path behind the reference's own plugin / launcher contracts.
 *
 * Every entry point names the reference interface it replaces (paths relative to the reference
 * tree).  All pointers in `inputs` / `outputs` / `workspace` are DEVICE pointers unless the
 * reference marks the tensor HOST_* (GPTAttention); `stream` is a hipStream_t passed as void*.
 * Functions return MI355Q_OK (0) or a negative status; the message is in mi355q_last_error()
 * (thread-local).  No entry point allocates device memory; enqueue calls never synchronise.
 */
#ifndef MI355Q_H
#define MI355Q_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

    /* ---- status: replaces TLLM_CHECK / TllmException + caughtError (plugins/common/plugin.cpp) */
    typedef enum
    {
        MI355Q_OK = 0,
        MI355Q_ERR_INVALID_ARGUMENT = -1, /* TLLM_CHECK failure */
        MI355Q_ERR_UNSUPPORTED = -2,      /* TLLM_THROW("... unsupported ...") */
        MI355Q_ERR_HIP = -3,              /* check_cuda_error */
        MI355Q_ERR_INTERNAL = -4
    } mi355q_status;

    const char* mi355q_last_error(void);
    const char* mi355q_version(void);

    /* ---- nvinfer1::DataType / Dims / PluginTensorDesc / PluginField, layout-compatible subsets */
    enum
    {
        MI355Q_FLOAT = 0,
        MI355Q_HALF = 1,
        MI355Q_INT8 = 2,
        MI355Q_INT32 = 3,
        MI355Q_BOOL = 4,
        MI355Q_UINT8 = 5,
        MI355Q_FP8 = 6,
        MI355Q_BF16 = 7,
        MI355Q_INT64 = 8
    };

    typedef struct
    {
        int32_t nbDims;
        int32_t d[8];
    } mi355q_dims; /* nvinfer1::Dims */

    typedef struct
    {
        mi355q_dims dims;
        int32_t type;   /* nvinfer1::DataType */
        int32_t format; /* nvinfer1::TensorFormat, 0 = kLINEAR */
        float scale;
    } mi355q_tensor_desc; /* nvinfer1::PluginTensorDesc */

    typedef struct
    {
        mi355q_tensor_desc desc;
        mi355q_dims min;
        mi355q_dims max;
    } mi355q_dynamic_tensor_desc; /* nvinfer1::DynamicPluginTensorDesc */

    enum
    {
        MI355Q_FIELD_FLOAT16 = 0,
        MI355Q_FIELD_FLOAT32 = 1,
        MI355Q_FIELD_FLOAT64 = 2,
        MI355Q_FIELD_INT8 = 3,
        MI355Q_FIELD_INT16 = 4,
        MI355Q_FIELD_INT32 = 5
    }; /* nvinfer1::PluginFieldType */

    typedef struct
    {
        const char* name;
        const void* data;
        int32_t type;
        int32_t length;
    } mi355q_plugin_field; /* nvinfer1::PluginField */

    /* ---- plugin objects: IPluginCreator::createPlugin / deserializePlugin + IPluginV2DynamicExt.
     * Names and field names are the reference's:
     *   "WeightOnlyGroupwiseQuantMatmul" {type_id, quant_algo, group_size}
     *        plugins/weightOnlyGroupwiseQuantMatmulPlugin/weightOnlyGroupwiseQuantMatmulPlugin.cpp:581-648
     *   "WeightOnlyQuantMatmul"          {type_id, weight_type_id}
     *        plugins/weightOnlyQuantMatmulPlugin/weightOnlyQuantMatmulPlugin.cpp:409-460
     *   "SmoothQuantGemm"                {has_per_channel_scaling, has_per_token_scaling, type_id}
     *        plugins/smoothQuantGemmPlugin/smoothQuantGemmPlugin.cpp:314-370
     *   "QuantizePerToken"               {} plugins/quantizePerTokenPlugin/quantizePerTokenPlugin.cpp
     *   "GPTAttention"                   {layer_idx, num_heads, ... is_medusa_enabled}
     *        plugins/gptAttentionCommon/gptAttentionCommon.cpp:1561-1593, gptAttentionPlugin.cpp:700-790
     *   "QuantizeTensor"                 {} plugins/quantizeTensorPlugin/quantizeTensorPlugin.cpp
     *   "RmsnormQuantization"            {eps, dyn_act_scaling, type_id}
     *        plugins/rmsnormQuantizationPlugin/rmsnormQuantizationPlugin.cpp:227-290
     *   "LayernormQuantization"          {eps, use_diff_of_squares, dyn_act_scaling, type_id}
     *        plugins/layernormQuantizationPlugin/layernormQuantizationPlugin.cpp:233-302
     *   "AllReduce"                      {group, type_id, strategy, counter} (+ optional "rank")
     *        plugins/ncclPlugin/allreducePlugin.cpp:141-203, 286-350 (one-shot / two-shot push kernels; RING and AUTO
     *        beyond the comm buffers = ncclAllReduce on the group's communicator, mi355q_comm_init_rank)
     *   "AllGather"                      {group, type_id} plugins/ncclPlugin/allgatherPlugin.cpp:90-110, 190-230
     *        (one input: ncclAllGather; with the AllReduce workspace table as second input: the push kernel)
     *   "Fp8Gemm"  (no reference plugin: TensorRT fuses Q/DQ+matmul) {type_id}
     * Weight tensors handed to enqueue are in the layout mi355q_preprocess_weights_for_mixed_gemm
     * produces for MI355Q_LAYOUT_GFX950 (the arch-selected layout, as the reference selects per SM). */
    typedef struct mi355q_plugin mi355q_plugin;

    /* IPluginCreator::createPlugin(name, PluginFieldCollection*) */
    mi355q_status mi355q_plugin_create(
        const char* name, const mi355q_plugin_field* fields, int32_t nbFields, mi355q_plugin** out);
    /* IPluginCreator::deserializePlugin(name, data, length) */
    mi355q_status mi355q_plugin_deserialize(const char* name, const void* data, size_t length, mi355q_plugin** out);
    /* IPluginV2::destroy / clone / getPluginType / getNbOutputs */
    void mi355q_plugin_destroy(mi355q_plugin* p);
    mi355q_status mi355q_plugin_clone(const mi355q_plugin* p, mi355q_plugin** out);
    const char* mi355q_plugin_type(const mi355q_plugin* p);
    int32_t mi355q_plugin_nb_outputs(const mi355q_plugin* p);
    /* IPluginV2DynamicExt::getOutputDimensions (constant dims only) */
    mi355q_status mi355q_plugin_get_output_dimensions(
        const mi355q_plugin* p, int32_t outputIndex, const mi355q_dims* inputs, int32_t nbInputs, mi355q_dims* out);
    /* IPluginV2DynamicExt::supportsFormatCombination; returns 1 / 0, negative on error */
    int32_t mi355q_plugin_supports_format_combination(
        const mi355q_plugin* p, int32_t pos, const mi355q_tensor_desc* inOut, int32_t nbInputs, int32_t nbOutputs);
    /* IPluginV2Ext::getOutputDataType; returns the DataType code, negative on error */
    int32_t mi355q_plugin_get_output_data_type(
        const mi355q_plugin* p, int32_t index, const int32_t* inputTypes, int32_t nbInputs);
    /* IPluginV2DynamicExt::configurePlugin */
    mi355q_status mi355q_plugin_configure(mi355q_plugin* p, const mi355q_dynamic_tensor_desc* in, int32_t nbInputs,
        const mi355q_dynamic_tensor_desc* out, int32_t nbOutputs);
    /* IPluginV2DynamicExt::getWorkspaceSize */
    size_t mi355q_plugin_get_workspace_size(const mi355q_plugin* p, const mi355q_tensor_desc* inputs, int32_t nbInputs,
        const mi355q_tensor_desc* outputs, int32_t nbOutputs);
    /* IPluginV2DynamicExt::enqueue -- the hot call.  Returns 0 on success like the reference. */
    int32_t mi355q_plugin_enqueue(mi355q_plugin* p, const mi355q_tensor_desc* inputDesc,
        const mi355q_tensor_desc* outputDesc, const void* const* inputs, void* const* outputs, void* workspace,
        void* stream);
    /* IPluginV2::getSerializationSize / serialize */
    size_t mi355q_plugin_get_serialization_size(const mi355q_plugin* p);
    mi355q_status mi355q_plugin_serialize(const mi355q_plugin* p, void* buffer);
    /* getPluginCreators(int32_t&) (plugins/api/tllmPlugin.h:47-56): NULL-terminated list of names */
    const char* const* mi355q_get_plugin_names(int32_t* nbPlugins);

    /* ---- host-side weight tooling: cpp/tensorrt_llm/thop/weightOnlyQuantOp.cpp (CPU tensors) */
    enum
    {
        MI355Q_WEIGHT_INT8 = 1, /* WeightTypeId::INT8  weightOnlyQuantMatmulPlugin.h:38-42 */
        MI355Q_WEIGHT_INT4 = 2
    };
    enum
    {
        MI355Q_LAYOUT_ROW_MAJOR = 0,         /* unprocessed [K, N] (int4: [K, N/2], low nibble = even column) */
        MI355Q_LAYOUT_REF_SM80 = 1,          /* ColumnMajorTileInterleave<64, 2|4> + permute + bias */
        MI355Q_LAYOUT_REF_SM90 = 2,          /* plain column-major, signed */
        MI355Q_LAYOUT_GFX950 = 3             /* this library's MFMA-fragment-major layout */
    };
    /* thop preprocess_weights_for_mixed_gemm (weightOnlyQuantOp.cpp:116-141) /
     * cutlass_preprocessors.cpp:535-584.  shape = {experts, K, N} in ELEMENTS; src is ROW_MAJOR;
     * dst receives `layout`.  HOST pointers. */
    mi355q_status mi355q_preprocess_weights_for_mixed_gemm(
        int8_t* dst, const int8_t* src, const size_t shape[3], int32_t weight_type, int32_t layout);
    /* Inverse: any layout -> ROW_MAJOR (ingest reference-preprocessed checkpoints).  HOST pointers. */
    mi355q_status mi355q_unprocess_weights_for_mixed_gemm(
        int8_t* dst, const int8_t* src, const size_t shape[3], int32_t weight_type, int32_t layout);
    /* thop unpack_int4_packed_tensor_to_int8 / pack_int8_tensor_to_packed_int4 (:274-343). HOST. */
    mi355q_status mi355q_unpack_int4_packed_tensor_to_int8(int8_t* dst, const int8_t* src, size_t numPacked);
    mi355q_status mi355q_pack_int8_tensor_to_packed_int4(int8_t* dst, const int8_t* src, size_t numPacked);
    /* symmetric_quantize<float-scales, float-weights> (cutlass_preprocessors.cpp:621-729).  HOST.
     * unprocessed may be NULL.  processed is written in `layout`. */
    mi355q_status mi355q_symmetric_quantize(int8_t* processed, int8_t* unprocessed, float* scales,
        const float* weight, const size_t shape[3], int32_t weight_type, int32_t layout);

    /* ---- kernel launchers (layer 2 of the reference), device pointers */

    /* kernels/weightOnlyBatchedGemv/common.h:63-106 WeightOnlyParams + kernelLauncher.cu:122
     * weight_only_batched_gemv_launcher, and CutlassFpAIntBGemmRunner::gemm
     * (kernels/cutlass_kernels/fpA_intB_gemm/fpA_intB_gemm.h:54-137): one struct, any m. */
    typedef struct
    {
        const void* qweight;   /* MI355Q_LAYOUT_GFX950 */
        const void* scales;    /* [k/group, n] (group-wise) or [n] (per-channel), act dtype */
        const void* zeros;     /* [k/group, n] additive, act dtype, or NULL */
        const void* in;        /* [m, k] act dtype */
        const void* act_scale; /* [k] pre-quant scale or NULL */
        const void* bias;      /* [n] or NULL */
        void* out;             /* [m, n] act dtype */
        int32_t m, n, k;
        int32_t group_size;    /* 64 / 128, or 0 == per-channel */
        int32_t weight_type;   /* MI355Q_WEIGHT_INT8 / INT4 */
        int32_t act_type;      /* MI355Q_HALF / MI355Q_BF16 */
        void* workspace;       /* smoothed activations (m*k, 256-B aligned) + split-K slabs; may be NULL
                                  when act_scale is NULL (split-K is then skipped) */
        size_t workspace_bytes;
        /* W4A8-AWQ (quant_algo bit FP8_ALPHA, weightOnlyGroupwiseQuantMatmulPlugin.cpp:148-168, 376-418; all zero / NULL
         * = the plain W4A16 / W8A16 GEMM): the activation, after the pre-quant scale, is rounded through OCP e4m3
         * (apply_per_channel_scale<T, __nv_fp8_e4m3>, preQuantScaleKernel.cu:26-79) and the result is
         * out = alpha * (sum_k a8 * w) + bias.  alpha is read on the device (the reference copies it to the host with a
         * blocking cudaMemcpy, :377-380).  The workspace must hold the m * k converted activations. */
        const float* alpha;    /* device [1] or NULL */
        int32_t act_fp8;       /* 0 / 1 */
    } mi355q_weight_only_params;
    /* CutlassFpAIntBGemmRunner::getWorkspaceSize + the plugin's smoothed-activation area
     * (weightOnlyGroupwiseQuantMatmulPlugin.cpp:330-332) */
    size_t mi355q_weight_only_gemm_workspace_size(int32_t max_m, int32_t n, int32_t k, int32_t weight_type);
    mi355q_status mi355q_weight_only_gemm(const mi355q_weight_only_params* params, void* stream);
    /* Dequantise the GFX950 layout back to a dense [k, n] act-dtype matrix on the device
     * (the reference tests do this with an identity matmul, test_weight_only_quant_matmul.py:44-53). */
    mi355q_status mi355q_weight_only_dequantize(const mi355q_weight_only_params* params, void* stream);

    /* kernels/cutlass_kernels/moe_gemm/moe_gemm_kernels.h:51-58 MoeGemmRunner<T, WeightType>::moeGemmBiasAct /
     * moeGemm with weight-only quantised experts (WeightType uint8_t / cutlass::uint4b_t: moe_gemm_kernels_*_uint{4,8}.cu):
     * for every expert e, C[rows_e, :] = act(A[rows_e, :] x dequant(B[e]) + biases[e]), rows sorted by expert,
     * rows_e = [total_rows_before_expert[e-1], total_rows_before_expert[e]) (inclusive cumulative counts on the
     * device, computeTotalRowsBeforeExpert in kernels/mixtureOfExperts/moe_kernels.cu).  One launch. */
    typedef struct
    {
        const void* A;             /* [total_rows, gemm_k] act dtype */
        const void* B;             /* [num_experts] preprocessed weights (MI355Q_LAYOUT_GFX950 of each [k, n] matrix) */
        const void* weight_scales; /* [num_experts, gemm_n] act dtype (per-column scales) */
        const void* biases;        /* [num_experts, gemm_n] act dtype or NULL (moeGemm) */
        void* C;                   /* [total_rows, gemm_n] act dtype */
        const int64_t* total_rows_before_expert; /* device [num_experts] */
        int64_t total_rows, gemm_n, gemm_k;
        int32_t num_experts;
        int32_t activation_type;   /* ActivationType (moe_gemm_kernels.h:27-36): 0 Gelu, 1 Relu, 2 Silu, 5 Identity */
        int32_t weight_type;       /* MI355Q_WEIGHT_INT8 / INT4 */
        int32_t act_type;          /* MI355Q_HALF / MI355Q_BF16 */
        /* optional: fp32 scratch that lets the GEMM split K over workgroups while its output tiles leave most CUs idle
         * (few rows, or few n-tiles and a long k) and it has neither bias nor activation; splits * total_rows * gemm_n * 4
         * bytes are used, at most 8 splits, fewer if the scratch is smaller; NULL = never */
        void* splitk_workspace;
        size_t splitk_workspace_bytes;
    } mi355q_moe_gemm_params;
    mi355q_status mi355q_moe_gemm_bias_act(const mi355q_moe_gemm_params* params, void* stream);

    /* kernels/mixtureOfExperts/moe_kernels.h CutlassMoeFCRunner<T, uint8_t | uint4b_t>::runMoe (moe_kernels.cu:922-1035),
     * the body of the MixtureOfExperts plugin (plugins/mixtureOfExperts/mixtureOfExpertsPlugin.cpp:388-431) for
     * weight-only quantised experts: softmax + top-k routing, the sort of the (token, k) pairs by expert, the gather,
     * expert GEMM 1 (+ bias, activation; Swiglu / Geglu: fc1 * act(gate) over a [.., 2 inter] GEMM), expert GEMM 2, and
     * the un-permute / bias / routing-scale / k-way sum.  Dense (unquantised) experts are not built. */
    typedef struct
    {
        const void* input;               /* act dtype [num_rows, hidden_size] */
        const float* gating_output;      /* fp32 [num_rows, num_experts] router logits */
        const void* fc1_expert_weights;  /* [experts_per_node] preprocessed [hidden, inter (x2 if gated)] (MI355Q_LAYOUT_GFX950) */
        const void* fc1_scales;          /* act dtype [experts_per_node, inter (x2)] */
        const void* fc1_expert_biases;   /* act dtype [experts_per_node, inter (x2)] or NULL */
        const void* fc2_expert_weights;  /* [experts_per_node] preprocessed [inter, hidden] */
        const void* fc2_scales;          /* act dtype [experts_per_node, hidden] */
        const void* fc2_expert_biases;   /* act dtype [experts_per_node, hidden] or NULL; added on tp_rank 0 only */
        const uint8_t* finished;         /* bool [num_rows] or NULL: finished rows are routed nowhere */
        void* output;                    /* act dtype [num_rows, hidden_size] */
        void* workspace;                 /* mi355q_moe_workspace_size bytes */
        size_t workspace_bytes;
        int32_t num_rows, hidden_size, inter_size, num_experts, k;
        int32_t activation_type;         /* ActivationType (moe_gemm_kernels.h:27-36): 0 Gelu 1 Relu 2 Silu 3 Swiglu 4 Geglu 5 Identity */
        int32_t act_type;                /* MI355Q_HALF / MI355Q_BF16 */
        int32_t weight_type;             /* MI355Q_WEIGHT_INT8 / INT4 */
        int32_t ep_size, ep_rank;        /* expert parallelism (MOEParallelismConfig): this rank owns experts
                                            [ep_rank * num_experts / ep_size, +num_experts / ep_size); 0 / 1 = none */
        int32_t tp_size, tp_rank;        /* tensor parallelism: inter_size is the local slice; bias 2 on rank 0 */
        int32_t normalization_mode;      /* MOEExpertScaleNormalizationMode: 0 NONE, 1 RENORMALIZE */
    } mi355q_moe_params;
    size_t mi355q_moe_workspace_size(const mi355q_moe_params* params);
    mi355q_status mi355q_run_moe(const mi355q_moe_params* params, void* stream);

    /* kernels/preQuantScaleKernel.cu:26-136 apply_per_channel_scale_kernel_launcher */
    mi355q_status mi355q_apply_per_channel_scale(
        void* out, const void* act, const void* scale, int32_t rows, int32_t cols, int32_t act_type, void* stream);

    /* kernels/cutlass_kernels/int8_gemm/int8_gemm.h:43-89 CutlassInt8GemmRunner<T>::gemm:
     * C[m,n] = cvt(float(sum_k A[m,k]*B[n,k]) * (alphaCol[n] * alphaRow[m])) */
    mi355q_status mi355q_int8_gemm(const int8_t* A, const int8_t* B, int32_t per_token, int32_t per_channel,
        const float* alphaCol, const float* alphaRow, void* C, int32_t out_type, int32_t m, int32_t n, int32_t k,
        void* stream);

    /* The same with the runner's workspace arguments (int8_gemm.h:54-58 gemm(..., workspacePtr, workspaceBytes, stream),
     * getWorkspaceSize(m, n, k) :63): the reference's CUTLASS kernels use it for serial split-k, here it holds the partial
     * tiles of the persistent / stream-K kernel that shapes whose 256 x 256 tiles leave the last round of CUs part-filled
     * take (cfg3: 864 tiles = 3.375 rounds).  mi355q_gemm8_workspace_size: bytes for this m, n, k (0 = the shape does
     * not use one; at most 64 MiB); with a null / smaller workspace the call runs one tile per workgroup.  int8 results
     * are bit-identical either way; fp8 sums its fp32 partial sums in a fixed order.  Workspace 256-byte aligned. */
    size_t mi355q_gemm8_workspace_size(int32_t m, int32_t n, int32_t k);
    mi355q_status mi355q_int8_gemm_ws(const int8_t* A, const int8_t* B, int32_t per_token, int32_t per_channel,
        const float* alphaCol, const float* alphaRow, void* C, int32_t out_type, int32_t m, int32_t n, int32_t k,
        void* workspace, size_t workspaceBytes, void* stream);
    mi355q_status mi355q_fp8_gemm_ws(const void* A, const void* B, const float* scaleA, const float* scaleB, void* C,
        int32_t out_type, int32_t m, int32_t n, int32_t k, void* workspace, size_t workspaceBytes, void* stream);

    /* kernels/quantization.cu:118-127 invokePerTokenQuantization */
    mi355q_status mi355q_per_token_quantization(int8_t* dst, const void* src, int64_t numRows, int64_t numCols,
        float* scalePtr, int32_t src_type, void* stream);

    /* kernels/quantization.cu:69-86 invokeQuantization(dst, src, size, scalePtr, stream, maxGridSize): static per-tensor
     * int8 quantisation dst = cvt.rni.sat(src * *scalePtr); size must be a multiple of 4 (QuantizeTensor plugin). */
    mi355q_status mi355q_quantize_tensor(int8_t* dst, const void* src, int64_t size, const float* scalePtr,
        int32_t src_type, void* stream);

    /* The gated activation of the SmoothQuant MLP between the gate/up GEMMs and the down projection, in one pass:
     * tensorrt_llm/quantization/layers.py:922-941 SmoothQuantGatedMLP.forward (inter = silu(fc(x)); y = inter * gate(x);
     * y = y / smoother; quantize_tensor(y, scale) | quantize_per_token(y)) and tensorrt_llm/functional.py:736-752 swiglu
     * (the fused [rows, 2 * cols] layout: gate_in = input, act_in = input + cols, both strides 2 * cols).  Every step
     * is rounded to T like the reference's layer boundaries.  out (T) and/or out_quant (int8): with dynamic_scale the
     * row is quantised with 127 / amax(row) and dynamic_scale[row] = amax / 127 (quantization.cu:93-127), else with
     * scale != NULL by *scale (quantization.cu:30-86); smoother may be NULL.  cols % 8 == 0, cols <= 32768. */
    typedef struct mi355q_gated_act_params
    {
        const void* act_in;   /* T [rows, cols], row stride ld_act elements: goes through SiLU */
        const void* gate_in;  /* T [rows, cols], row stride ld_gate elements: the multiplier */
        const void* smoother; /* T [cols] or NULL */
        void* out;            /* T [rows, cols] or NULL */
        int8_t* out_quant;    /* int8 [rows, cols] or NULL */
        const float* scale;   /* static per-tensor scale or NULL */
        float* dynamic_scale; /* fp32 [rows] out, or NULL */
        int32_t rows, cols;
        int64_t ld_act, ld_gate;
        int32_t type; /* MI355Q_FLOAT / MI355Q_HALF / MI355Q_BF16 */
    } mi355q_gated_act_params;
    mi355q_status mi355q_gated_silu(const mi355q_gated_act_params* params, void* stream);

    /* kernels/rmsnormKernels.cu:176-206 invokeGeneralRmsNorm(out, input, gamma, beta, eps, tokens, hidden_dim, stream,
     * scale, dynamic_scale, normed_output_quant): out = T(x * rsqrt(mean(x^2) + eps) * gamma + beta); with
     * dynamic_scale != NULL the row is quantised with 127 / amax(row) and dynamic_scale[row] = amax / 127, else with
     * scale != NULL by *scale, into normed_output_quant (int8); with neither, `out` receives the T result.
     * type: MI355Q_FLOAT / MI355Q_HALF (the reference's) / MI355Q_BF16.  beta may be NULL. */
    mi355q_status mi355q_rms_norm(void* out, const void* input, const void* gamma, const void* beta, float eps,
        int32_t tokens, int32_t hidden_dim, const float* scale, float* dynamic_scale, int8_t* normed_output_quant,
        int32_t type, void* stream);
    /* kernels/layernormKernels.cu:230-262 invokeGeneralLayerNorm(..., use_diff_of_squares, scale, dynamic_scale,
     * normed_output_quant): same with (x - mean) and Var[x]; use_diff_of_squares selects E[x^2] - E[x]^2. */
    mi355q_status mi355q_layer_norm(void* out, const void* input, const void* gamma, const void* beta, float eps,
        int32_t tokens, int32_t hidden_dim, int32_t use_diff_of_squares, const float* scale, float* dynamic_scale,
        int8_t* normed_output_quant, int32_t type, void* stream);

    /* FP8 e4m3 (OCP) GEMM: C[m,n] = cvt((sum_k A[m,k]*B[n,k]) * (scaleA*scaleB)); replaces the
     * TensorRT Q/DQ fusion / cuBLASLt fp8 path (plugins/gemmPlugin/gemmPlugin.cpp:193-198). */
    mi355q_status mi355q_fp8_gemm(const void* A, const void* B, const float* scaleA, const float* scaleB, void* C,
        int32_t out_type, int32_t m, int32_t n, int32_t k, void* stream);
    /* thop/fp8Op.cpp quantize_e4m3_{activation,weight,per_tensor}: mode 0 per-row (scales [rows]),
     * 1 per-col (scales [cols]), 2 per-tensor (scales must hold TWO floats: [0] receives the scale,
     * [1] is scratch for the amax pass). */
    mi355q_status mi355q_quantize_e4m3(void* dst, float* scales, const void* src, int64_t rows, int64_t cols,
        int32_t src_type, int32_t mode, void* stream);

    /* kernels/decoderMaskedMultiheadAttention.h:76-244 Multihead_attention_params +
     * masked_multihead_attention(params, KVBlockArray, ...): one generation step over a paged cache. */
    typedef struct
    {
        const void* qkv;            /* [batch, (Hq+2Hkv)*Dh] act dtype; q | k | v */
        const void* qkv_bias;       /* [(Hq+2Hkv)*Dh] or NULL */
        void* out;                  /* [batch, Hq*Dh] act dtype */
        const int32_t* sequence_lengths; /* device [batch]: kv length INCLUDING the current token */
        const int64_t* block_pointers;   /* device [batch, 2, max_blocks_per_seq] block addresses (KVBlockArray::data) */
        const float* kv_scale_orig_quant; /* device [1] or NULL */
        const float* kv_scale_quant_orig; /* device [1] or NULL */
        void* workspace;            /* mi355q_decode_attention_workspace_size bytes */
        size_t workspace_bytes;
        int32_t batch, num_heads, num_kv_heads, head_size;
        int32_t max_blocks_per_seq, tokens_per_block;
        int32_t max_seq_len;        /* upper bound of sequence_lengths (host knows it: max past kv len + 1) */
        int32_t act_type;           /* MI355Q_HALF / MI355Q_BF16 */
        int32_t kv_type;            /* act_type, MI355Q_INT8 or MI355Q_FP8 */
        int32_t position_embedding_type; /* PositionEmbeddingType, kernels/gptKernels.h:40-50 */
        int32_t rotary_embedding_dim;
        float rotary_embedding_base;
        int32_t rotary_embedding_scale_type; /* RotaryScalingType :52-57 */
        float rotary_embedding_scale;
        int32_t rotary_embedding_max_positions;
        float q_scaling;
        int32_t multi_block_mode;   /* 0: heuristic, >0: force that many kv splits */
        /* ---- cache addressing beyond the identity map (kernels/kvCacheUtils.h:36-153, 241-338; all zero / NULL =
         * paged cache, no window, no sink tokens, beam width 1, i.e. what the fields above describe on their own) */
        int32_t cyclic_attention_window_size; /* Multihead_attention_params::cyclic_attention_window_size
                                       (decoderMaskedMultiheadAttention.h:120-124); 0 = no window.  With a window the step
                                       attends to cache slots [0, min(kv length, window)) and writes the new token at
                                       KVBlockArray::getKVTokenIdx(kv length) (kvCacheUtils.h:109-117) */
        int32_t sink_token_length;  /* KVBlockArray::mSinkTokens: slots that are never overwritten */
        int32_t beam_width;         /* 0 or 1: no beams.  > 1: `batch` counts batch x beam rows (batch_beam_idx) */
        int32_t max_attention_window_size; /* row stride of cache_indirection (:1873-1879) */
        const int32_t* cache_indirection; /* device [batch / beam_width, beam_width, max_attention_window_size]: the
                                       beam a cached token of this (sequence, beam) was written by (:1879, :2076, :2441) */
        const int32_t* input_lengths; /* device [batch]: context lengths; with beams the first input_lengths[b] tokens
                                       are read from beam 0 (:1497-1498, :1929) */
        const void* kv_linear_buffer; /* NULL, or the contiguous cache of KVLinearBuffer (kvCacheUtils.h:241-338):
                                       [batch, 2, Hkv, linear_max_seq_len, Dh] in kv_type; block_pointers is then unused */
        int32_t linear_max_seq_len;
        /* ALiBi (position_embedding_type 3 / 4; decoderMaskedMultiheadAttentionTemplate.h:1891-1895, :2026): act_type
         * [num_heads] slopes, score += slope[h] * (t - tlength); NULL otherwise */
        const void* alibi_slopes;
        /* position_shift_enabled (StreamingLLM; Multihead_attention_params::position_shift_enabled,
         * decoderMaskedMultiheadAttention.h, POS_SHIFT of decoderMaskedMultiheadAttentionTemplate.h:1500, 1745-1752,
         * 1918-1927, 2065, 2386; invokeShiftKCache, unfusedAttentionKernels.cu:1776-1947, called from
         * gptAttentionCommon.cpp:1276-1290): the K cache holds keys WITHOUT the position embedding; every step the live
         * keys -- the sink tokens and the last window - sink tokens -- are rotated to their position INSIDE the cache
         * (0 .. kv length - 1 in age order), the query and the new key to position min(kv length, window), and the new
         * key is stored un-rotated.  RoPE types only; 16-bit caches, beam width 1.  Needs the larger workspace
         * mi355q_decode_attention_workspace_size reports (the rotated keys of the step: the reference's
         * shift_k_cache, :1260-1262). */
        int32_t position_shift_enabled;
        /* relative attention bias (position_embedding_type 5, kRELATIVE; Multihead_attention_params::
         * relative_attention_bias / relative_attention_bias_stride / max_distance, Template.h:1469-1476, 1773-1797,
         * 1960-1985, 2026): act_type values added to the scaled scores.  max_distance == 0: the explicit tensor
         * [num_heads, stride, stride], entry [h][kv length][t]; max_distance > 0: the table [num_heads, stride = num_buckets]
         * and the bucket of (t - kv length) computed on the fly.  NULL otherwise. */
        const void* relative_attention_bias;
        int32_t relative_attention_bias_stride;
        int32_t max_distance;
    } mi355q_decode_attention_params;
    size_t mi355q_decode_attention_workspace_size(const mi355q_decode_attention_params* params);
    mi355q_status mi355q_masked_multihead_attention(const mi355q_decode_attention_params* params, void* stream);
    /* No counterpart in the reference.  The same generation step for callers that cannot allocate and have no
     * workspace to hand over (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc):
     * params->workspace / workspace_bytes are ignored; the split-KV partials go to a static device array inside the
     * library (mi355q_decode_attention_static_scratch_size() bytes per GPU, held until the process ends).  Calls no
     * hipMalloc / hipFree; a call whose partials would not fit returns MI355Q_ERR_INVALID_ARGUMENT.  All calls on
     * one GPU share that array: calls that overlap on one GPU (two streams) race -- serialise them.  The first call
     * on each device resolves the array's address (hipGetSymbolAddress); make it outside a graph capture. */
    size_t mi355q_decode_attention_static_scratch_size(void);
    mi355q_status mi355q_masked_multihead_attention_static(const mi355q_decode_attention_params* params, void* stream);

    /* The context (prefill) phase of GPTAttention: plugins/gptAttentionCommon/gptAttentionCommon.cpp:641-1133
     * enqueueContext = invokeAddFusedQKVBiasTranspose / invokeApplyBiasRopeUpdateKVCache
     * (kernels/unfusedAttentionKernels.h) + causal attention among the input tokens
     * (contextFusedMultiHeadAttention/fmhaRunner.h, or the unfused cuBLAS + invokeMaskedSoftmax path :1000-1120)
     * + the K / V append into the KVBlockArray cache.  One call does all three; a following
     * mi355q_masked_multihead_attention step with sequence_lengths = context_lengths + 1 continues the sequence. */
    typedef struct
    {
        const void* qkv;            /* remove_padding: [num_tokens, (Hq+2Hkv)*Dh], sequences back to back;
                                       else [batch, max_context_length, (Hq+2Hkv)*Dh] */
        const void* qkv_bias;       /* [(Hq+2Hkv)*Dh] or NULL */
        void* out;                  /* same token indexing, [.., Hq*Dh]; rows past a sequence's length are not written */
        const int32_t* context_lengths;  /* device [batch] */
        const int64_t* block_pointers;   /* device [batch, 2, max_blocks_per_seq] or NULL (no cache append) */
        const float* kv_scale_orig_quant; /* device [1]; needed for an 8-bit cache */
        void* workspace;            /* mi355q_context_attention_workspace_size bytes */
        size_t workspace_bytes;
        int32_t batch, num_heads, num_kv_heads, head_size;
        int32_t max_blocks_per_seq, tokens_per_block;
        int32_t max_context_length; /* upper bound of context_lengths (host_context_lengths' max) */
        int32_t num_tokens;         /* remove_padding: total rows of qkv (sum of context_lengths) */
        int32_t remove_padding;     /* 0 / 1 */
        int32_t act_type;           /* MI355Q_HALF / MI355Q_BF16 */
        int32_t kv_type;            /* act_type, MI355Q_INT8 or MI355Q_FP8 */
        int32_t position_embedding_type; /* 0 learned-absolute (none applied here), 1 RoPE GPT-J, 2 RoPE GPT-NeoX */
        int32_t rotary_embedding_dim;
        float rotary_embedding_base;
        int32_t rotary_embedding_scale_type;
        float rotary_embedding_scale;
        int32_t rotary_embedding_max_positions;
        float q_scaling;
        const void* kv_linear_buffer; /* NULL, or the contiguous cache of KVLinearBuffer (kvCacheUtils.h:241-338):
                                       [batch, 2, Hkv, linear_max_seq_len, Dh] in kv_type; block_pointers is then unused */
        int32_t linear_max_seq_len;
        /* chunked context (use_paged_context_fmha, gptAttentionCommon.cpp:825-864; kv_seq_lengths of
         * unfusedAttentionKernels_2_template.h:252-256): device [batch] tokens of each sequence INCLUDING this call's, or
         * NULL (= context_lengths: nothing cached yet).  The call's tokens take positions sequence_lengths -
         * context_lengths .. and attend to every earlier token through the cache (16-bit caches only, as the reference). */
        const int32_t* sequence_lengths;
        int32_t max_kv_length;      /* host bound of sequence_lengths (max_past_kv_len); 0 = max_context_length */
        /* sliding window attention (gptKernels.cu:174-177: key t visible to query a iff a - window <= t <= a) and the
         * cyclic cache the append follows (kvCacheUtils.h:71-117; the sink tokens and the last window - sink tokens are
         * kept, unfusedAttentionKernels_2_template.h:336-340).  0 = no window. */
        int32_t cyclic_attention_window_size;
        int32_t sink_token_length;
        /* ALiBi (position_embedding_type 3 / 4): T [num_heads] slopes, score += slope[h] * (t - a)
         * (unfusedAttentionKernels.cu:212-218); NULL otherwise */
        const void* alibi_slopes;
        /* position_shift_enabled: the attention of this call is unchanged, but K goes to the cache WITHOUT the position
         * embedding (k_to_cache = POS_SHIFT ? k_wo_pos : k, unfusedAttentionKernels_2_template.h:284-347), for
         * generation steps with mi355q_decode_attention_params.position_shift_enabled */
        int32_t position_shift_enabled;
        /* relative attention bias (position_embedding_type 5, kRELATIVE; invokeAddRelativeAttentionBiasUnaligned,
         * unfusedAttentionKernels.cu:1688-1751, called with bidirectional = false from gptAttentionCommon.cpp:1006-1016
         * on the scores already scaled by qk_scale, :816-817): T values added to the scaled scores.  max_distance == 0:
         * the explicit tensor [num_heads, stride, stride], entry [h][query position][key position] (the reference indexes it with
         * cyclic_attention_window_size as the row length, which is the tensor's own for every consistent caller; the
         * plugin passes the tensor's dims.d[1]); max_distance > 0: the table
         * [num_heads, stride = num_buckets] and the one-directional bucket of (query position - key position) computed
         * on the fly.  NULL otherwise. */
        const void* relative_attention_bias;
        int32_t relative_attention_bias_stride;
        int32_t max_distance;
    } mi355q_context_attention_params;
    size_t mi355q_context_attention_workspace_size(const mi355q_context_attention_params* params);
    mi355q_status mi355q_context_attention(const mi355q_context_attention_params* params, void* stream);

    enum
    {
        MI355Q_AR_RING = 0,
        MI355Q_AR_ONESHOT = 1,
        MI355Q_AR_TWOSHOT = 2,
        MI355Q_AR_AUTO = 3
    }; /* AllReduceStrategyType, kernels/customAllReduceKernels.h */
    /* AllreducePlugin::selectImplementation (plugins/ncclPlugin/allreducePlugin.cpp:111-139) */
    int32_t mi355q_allreduce_select_implementation(size_t messageBytes, int32_t worldSize, size_t maxWorkspaceBytes);
    /* utils::customAllReduceUtils::getMaxRequiredWorkspaceSize (common/customAllReduceUtils.h:30-37) */
    size_t mi355q_allreduce_max_workspace_size(int32_t worldSize);

#define MI355Q_MAX_RANKS_PER_NODE 8 /* kernels/customAllReduceKernels.h:33 */
#define MI355Q_IPC_HANDLE_BYTES 64  /* sizeof(hipIpcMemHandle_t) */
#define MI355Q_AR_MAX_WORKGROUPS 256 /* one per CU (the reference: MAX_ALL_REDUCE_BLOCKS = 24, customAllReduceKernels.h:32) */
    /* bytes of ONE barrier-flag array (barrier-in or barrier-out) per rank: flags are per (phase, source rank,
     * workgroup), followed by the rank's own epoch and arrival words (the kernels derive the flag value and the
     * ping / pong parity of a call from that device-resident epoch, so plugins that replay the same build-time
     * `counter` on every pass -- an engine, a hipGraph -- stay synchronised); replaces IPC_BARRIERS_SIZE_PER_GPU of
     * tensorrt_llm/_ipc_utils.py.  The arrays must be zeroed once, at allocation (mi355q_ipc_alloc does). */
    size_t mi355q_allreduce_barrier_bytes(void);
    /* kernels/customAllReduceKernels.h:44-58 AllReduceParams, plus what the MI355X kernels add (trailing fields):
     * the input where it lies (they push it to the peers themselves: no staging copy), the byte size of a comm buffer
     * (checked against the message), a host-visible word that receives 1 when a peer never arrived (or NULL), the
     * wall-clock bound of every wait in milliseconds (0 = 30000), and BOTH sets of comm buffers: the kernels choose
     * ping or pong by their own epoch, barrier_flag / peer_comm_buffer_ptrs keep the reference's meaning and are not
     * read by them.  All ranks must issue the same sequence of collectives on these buffers, one stream at a time.
     * A workgroup that waits for its peers keeps its place on the CU, so all ranks' workgroups of a collective must be
     * resident together: always true with one rank per GPU; ranks that share a GPU (one-GPU rehearsals) set
     * max_workgroups so that ranks x max_workgroups x 256 threads fit the chip (e.g. 64 for 8 ranks). */
    typedef struct
    {
        size_t elts_total;
        size_t elts_per_rank;
        size_t elts_per_block;
        size_t rank_offset;
        size_t ranks_per_node, rank, local_rank;
        uint32_t barrier_flag;
        uint32_t* peer_barrier_ptrs_in[MI355Q_MAX_RANKS_PER_NODE];
        uint32_t* peer_barrier_ptrs_out[MI355Q_MAX_RANKS_PER_NODE];
        void* peer_comm_buffer_ptrs[MI355Q_MAX_RANKS_PER_NODE];
        void* local_output_buffer_ptr;
        uint32_t* timeout_flag;
        const void* local_input_buffer_ptr;
        size_t comm_buffer_bytes; /* 0 = mi355q_allreduce_max_workspace_size(ranks_per_node) */
        uint32_t timeout_ms;
        void* peer_comm_buffer_ptrs_ping[MI355Q_MAX_RANKS_PER_NODE];
        void* peer_comm_buffer_ptrs_pong[MI355Q_MAX_RANKS_PER_NODE];
        uint32_t max_workgroups; /* 0 = MI355Q_AR_MAX_WORKGROUPS; lower it when several ranks share one GPU */
    } mi355q_allreduce_params;
    /* AllReduceParams::deserialize(buffer, tpSize, tpRank, flag_value) (customAllReduceKernels.cu:398-425):
     * buffer_ptrs = HOST array [4 * tpSize]: ping buffers | pong buffers | barrier-in | barrier-out, one per rank */
    mi355q_status mi355q_allreduce_params_deserialize(mi355q_allreduce_params* out, const void* const* buffer_ptrs,
        size_t tpSize, size_t tpRank, uint32_t flag_value);
    /* customAllReduce(params, data, elts, size_per_elem, dataType, strat, stream) (:427-454): one-shot / two-shot sum of
     * params->local_input_buffer_ptr over the ranks into `data` (in place allowed).  The reference has the caller copy
     * its input into peer_comm_buffer_ptrs[rank] first (allreducePlugin.cpp:195-199); here the kernel pushes it into
     * the peers' buffers itself.  dtype MI355Q_FLOAT / HALF / BF16; strategy MI355Q_AR_ONESHOT / TWOSHOT.  Capacity:
     * one-shot ranks * roundup(bytes, 256) <= comm_buffer_bytes, two-shot 2 * bytes <= comm_buffer_bytes. */
    /* plugins/ncclPlugin/allgatherPlugin.cpp:90-110 (ncclAllGather(in, out, size, ...)): output = the ranks' inputs back
     * to back in rank order, `bytes_per_rank` each (a multiple of 16), through the same peer buffers and flags as the
     * one-shot all-reduce (params as for mi355q_custom_all_reduce; ranks * roundup(bytes_per_rank, 256) must fit a comm
     * buffer).  Push-based, one launch, bounded waits (on a timeout the output is filled with NaN bytes). */
    mi355q_status mi355q_custom_all_gather(
        mi355q_allreduce_params* params, const void* input, void* output, size_t bytes_per_rank, void* stream);
    mi355q_status mi355q_custom_all_reduce(mi355q_allreduce_params* params, void* data, size_t elts, int32_t dtype,
        int32_t strategy, void* stream);
    /* invokeMultiGpuBarrier (:391-394) over peer_barrier_ptrs_out */
    mi355q_status mi355q_multi_gpu_barrier(const mi355q_allreduce_params* params, void* stream);
    /* No counterpart in the reference (its flag spins are unbounded): a zeroed host-visible record of FOUR words for
     * mi355q_allreduce_params.timeout_flag / the AllReduce plugin's workspace table; the kernels set word 0 to 1 when a
     * wait for a peer ran out of time (their output is then NaN) and leave in words 1..3 which wait it was
     * (phase << 16 | peer rank << 8 | workgroup & 255, the flag value waited for, the value found); the host may read
     * it at any time.  A caller that passes its own memory must provide those 16 bytes. */
    mi355q_status mi355q_allreduce_alloc_timeout_flag(uint32_t** flag);
    mi355q_status mi355q_allreduce_free_timeout_flag(uint32_t* flag);
    /* IpcMemory (tensorrt_llm/_ipc_utils.py:66-133; cpp/tensorrt_llm/runtime/ipcUtils.cpp): set-up time helpers --
     * allocate a zeroed fine-grained device buffer, export / import it across the ranks' processes.  These are the
     * only entry points that allocate device memory. */
    mi355q_status mi355q_ipc_alloc(size_t bytes, void** ptr);
    mi355q_status mi355q_ipc_free(void* ptr);
    mi355q_status mi355q_ipc_get_handle(void* ptr, uint8_t handle[MI355Q_IPC_HANDLE_BYTES]);
    mi355q_status mi355q_ipc_open_handle(const uint8_t handle[MI355Q_IPC_HANDLE_BYTES], void** ptr);
    mi355q_status mi355q_ipc_close_handle(void* ptr);

    /* ---- RCCL communicators for the RING strategy of "AllReduce" and for "AllGather" without a workspace table:
     * getCommMap / initCommMap (plugins/common/plugin.cpp:44-87).  The reference creates a communicator per rank group
     * in AllreducePlugin::initialize -- the group's first rank calls ncclGetUniqueId and ships the id over MPI, every
     * rank calls ncclCommInitRank -- and its enqueue issues ncclAllReduce / ncclAllGather on it
     * (allreducePlugin.cpp:181-185, allgatherPlugin.cpp:90-110).  Without MPI the id travels by the host's own means
     * (torch.distributed in mi355q/ipc.py): get_unique_id on one rank, init_rank on every rank of the group
     * (group_rank = this rank's position in the sorted group; the device that is current at the call is the rank's
     * GPU; a group that already has a communicator is left alone, as :54).  librccl.so.1 is loaded at the first of
     * these calls, not at library load.  comm_destroy = AllreducePlugin::terminate (:254-267). */
#define MI355Q_COMM_UNIQUE_ID_BYTES 128 /* sizeof(ncclUniqueId) */
    mi355q_status mi355q_comm_get_unique_id(uint8_t id[MI355Q_COMM_UNIQUE_ID_BYTES]);
    mi355q_status mi355q_comm_init_rank(const int32_t* group, int32_t group_size,
        const uint8_t id[MI355Q_COMM_UNIQUE_ID_BYTES], int32_t group_rank);
    int32_t mi355q_comm_exists(const int32_t* group, int32_t group_size); /* 1 / 0, negative on error */
    mi355q_status mi355q_comm_destroy(const int32_t* group, int32_t group_size);

#ifdef __cplusplus
}
#endif
#endif /* MI355Q_H */
