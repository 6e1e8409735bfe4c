// plugins.cpp -- host side above the C ABI: the four TensorRT plugin contracts of the hot path,
// restated without TensorRT types (IPluginV2DynamicExt -> PluginBase), plus the C entry points.
// Input order, flag compaction, shape arithmetic and error behaviour follow the reference files
// cited at each class.
#include "common.h"
#include "kernels.h"
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <vector>

namespace mi355q
{

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg)
{
    g_last_error = msg;
}

std::string fmtstr(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return std::string(buf);
}

namespace
{

// plugins/common/plugin.h read/write helpers
template <typename T>
void write(char*& p, const T& v)
{
    std::memcpy(p, &v, sizeof(T));
    p += sizeof(T);
}

template <typename T>
void read(const char*& p, T& v)
{
    std::memcpy(&v, p, sizeof(T));
    p += sizeof(T);
}

struct GemmDims // plugins/common/gemmPluginProfiler.h GemmDims
{
    int32_t minM = -1, maxM = -1, n = -1, k = -1;
    bool isInitialized() const
    {
        return minM >= 0 && maxM >= 0 && n >= 0 && k >= 0;
    }
};

struct FieldReader
{
    const mi355q_plugin_field* f;
    int n;
    const mi355q_plugin_field* find(const char* name) const
    {
        for (int i = 0; i < n; ++i)
            if (f[i].name && std::strcmp(f[i].name, name) == 0)
                return &f[i];
        return nullptr;
    }
    bool get_int(const char* name, int64_t& out) const
    {
        const mi355q_plugin_field* x = find(name);
        if (!x || !x->data)
            return false;
        switch (x->type)
        {
        case MI355Q_FIELD_INT8: out = *static_cast<const int8_t*>(x->data); return true;
        case MI355Q_FIELD_INT16: out = *static_cast<const int16_t*>(x->data); return true;
        case MI355Q_FIELD_INT32: out = *static_cast<const int32_t*>(x->data); return true;
        case MI355Q_FIELD_FLOAT32: out = (int64_t) * static_cast<const float*>(x->data); return true;
        default: return false;
        }
    }
    int req_int(const char* name) const
    {
        int64_t v;
        MQ_CHECK(get_int(name, v), "plugin field '%s' missing", name);
        return (int) v;
    }
    int opt_int(const char* name, int dflt) const
    {
        int64_t v;
        return get_int(name, v) ? (int) v : dflt;
    }
    float opt_float(const char* name, float dflt) const
    {
        const mi355q_plugin_field* x = find(name);
        if (!x || !x->data)
            return dflt;
        if (x->type == MI355Q_FIELD_FLOAT32)
            return *static_cast<const float*>(x->data);
        if (x->type == MI355Q_FIELD_FLOAT64)
            return (float) *static_cast<const double*>(x->data);
        int64_t v;
        return get_int(name, v) ? (float) v : dflt;
    }
};

} // namespace

struct PluginBase
{
    virtual ~PluginBase() = default;
    virtual const char* type() const = 0;
    virtual int nbOutputs() const
    {
        return 1;
    }
    virtual PluginBase* clone() const = 0;
    virtual void getOutputDimensions(int idx, const mi355q_dims* in, int nbIn, mi355q_dims* out) const = 0;
    virtual bool supportsFormatCombination(int pos, const mi355q_tensor_desc* io, int nbIn, int nbOut) const = 0;
    virtual int getOutputDataType(int idx, const int32_t* inTypes, int nbIn) const = 0;
    virtual void configure(const mi355q_dynamic_tensor_desc*, int, const mi355q_dynamic_tensor_desc*, int) {}
    virtual size_t workspaceSize(const mi355q_tensor_desc*, int, const mi355q_tensor_desc*, int) const
    {
        return 0;
    }
    virtual void enqueue(const mi355q_tensor_desc* inDesc, const mi355q_tensor_desc* outDesc, const void* const* inputs,
        void* const* outputs, void* workspace, hipStream_t stream)
        = 0;
    virtual size_t serializationSize() const = 0;
    virtual void serialize(char* buf) const = 0;
};

namespace
{

inline size_t align_up(size_t x, size_t a)
{
    return (x + a - 1) / a * a;
}

int rows_of(const mi355q_dims& d)
{
    return (int) dims_prod(d, 0, d.nbDims - 1);
}

// ===================================================================== WeightOnlyGroupwiseQuantMatmul
// plugins/weightOnlyGroupwiseQuantMatmulPlugin/weightOnlyGroupwiseQuantMatmulPlugin.{h,cpp}
constexpr int BIAS = 1, ZERO = 2, PRE_QUANT_SCALE = 4, FP8_ALPHA = 8; // :28-34
constexpr int FP16_INT4_RATIO = 4;                                    // weightOnlyQuantMatmulPlugin.h:48
[[maybe_unused]] constexpr int FP16_INT8_RATIO = 2;

struct WeightOnlyGroupwiseQuantMatmulPlugin : PluginBase
{
    int32_t mType, mQuantAlgo, mGroupSize;
    GemmDims mDims;
    int mPreQuantScaleInputIdx, mWeightInputIdx, mScalesInputIdx, mZerosInputIdx, mBiasesInputIdx, mAlphaInputIdx;
    size_t m_workspaceMaxSize = 0;
    int m_cfgMaxM = -1, m_cfgN = -1, m_cfgK = -1; // maxima of the last configurePlugin (what the workspace was sized for)

    WeightOnlyGroupwiseQuantMatmulPlugin(int type, int quant_algo, int group_size)
    {
        init(type, quant_algo, group_size);
    }
    void init(int type, int quant_algo, int group_size) // :132-229
    {
        mType = type;
        mQuantAlgo = quant_algo;
        mGroupSize = group_size;
        mPreQuantScaleInputIdx = (quant_algo & PRE_QUANT_SCALE) ? 1 : 0;
        mWeightInputIdx = mPreQuantScaleInputIdx + 1;
        mScalesInputIdx = mWeightInputIdx + 1;
        mZerosInputIdx = (quant_algo & ZERO) ? mScalesInputIdx + 1 : mScalesInputIdx;
        mBiasesInputIdx = (quant_algo & BIAS) ? mZerosInputIdx + 1 : mZerosInputIdx;
        mAlphaInputIdx = (quant_algo & FP8_ALPHA) ? mBiasesInputIdx + 1 : mBiasesInputIdx;
        // FP8_ALPHA (W4A8-AWQ) is Hopper-only in the reference (:148-154); built here: e4m3-rounded activations through
        // the W4A16 kernels (every e4m3 value is exact in fp16 / bf16), alpha applied in their epilogues
        if (mType != MI355Q_HALF && mType != MI355Q_BF16)
            MQ_UNSUPPORTED("Unsupported data type");
        MQ_CHECK(group_size > 0 && group_size % 64 == 0, "group_size must be a multiple of 64 (64 or 128), got %d",
            group_size);
    }
    const char* type() const override
    {
        return "WeightOnlyGroupwiseQuantMatmul";
    }
    PluginBase* clone() const override
    {
        return new WeightOnlyGroupwiseQuantMatmulPlugin(*this);
    }
    void getOutputDimensions(int idx, const mi355q_dims* in, int nbIn, mi355q_dims* out) const override // :243-281
    {
        MQ_CHECK(nbIn == mAlphaInputIdx + 1, "nbInputs %d != %d", nbIn, mAlphaInputIdx + 1);
        MQ_CHECK(idx == 0, "outputIndex");
        const int nbDimsA = in[0].nbDims;
        MQ_CHECK(nbDimsA >= 2 && in[mWeightInputIdx].nbDims == 2, "ranks");
        *out = in[0];
        out->d[nbDimsA - 1] = in[mWeightInputIdx].d[1] * FP16_INT4_RATIO;
    }
    bool supportsFormatCombination(int pos, const mi355q_tensor_desc* io, int, int) const override // :283-308
    {
        MQ_CHECK(pos < mAlphaInputIdx + 2, "pos");
        if ((mQuantAlgo & FP8_ALPHA) && pos == mAlphaInputIdx)
            return io[pos].type == MI355Q_FLOAT && io[pos].format == 0; // :293-296
        return io[pos].type == mType && io[pos].format == 0;
    }
    int getOutputDataType(int idx, const int32_t*, int) const override
    {
        MQ_CHECK(idx == 0, "index");
        return mType;
    }
    void configure(const mi355q_dynamic_tensor_desc* in, int, const mi355q_dynamic_tensor_desc*, int) override // :310-333
    {
        const int minM = rows_of(in[0].min), maxM = rows_of(in[0].max);
        const int maxK = in[0].max.d[in[0].max.nbDims - 1];
        const int maxN = in[mWeightInputIdx].max.d[1] * FP16_INT4_RATIO;
        if (!mDims.isInitialized())
            mDims = {minM, maxM, maxN / FP16_INT4_RATIO, maxK};
        m_workspaceMaxSize = mi355q_weight_only_gemm_workspace_size(maxM, maxN, maxK, MI355Q_WEIGHT_INT4);
        m_cfgMaxM = maxM, m_cfgN = maxN, m_cfgK = maxK;
    }
    size_t workspaceSize(const mi355q_tensor_desc*, int, const mi355q_tensor_desc*, int) const override
    {
        return m_workspaceMaxSize;
    }
    void enqueue(const mi355q_tensor_desc* inDesc, const mi355q_tensor_desc*, const void* const* inputs,
        void* const* outputs, void* workspace, hipStream_t stream) override // :341-517
    {
        const int m = rows_of(inDesc[0].dims);
        const int n = inDesc[mWeightInputIdx].dims.d[1];
        const int k = inDesc[0].dims.d[inDesc[0].dims.nbDims - 1];
        mi355q_weight_only_params p{};
        p.qweight = inputs[mWeightInputIdx];
        p.scales = inputs[mScalesInputIdx];
        p.zeros = (mQuantAlgo & ZERO) ? inputs[mZerosInputIdx] : nullptr;
        p.bias = (mQuantAlgo & BIAS) ? inputs[mBiasesInputIdx] : nullptr;
        p.act_scale = (mQuantAlgo & PRE_QUANT_SCALE) ? inputs[mPreQuantScaleInputIdx] : nullptr;
        if (mQuantAlgo & FP8_ALPHA)
        {
            p.alpha = static_cast<const float*>(inputs[mAlphaInputIdx]);
            p.act_fp8 = 1;
        }
        p.in = inputs[0];
        p.out = outputs[0];
        p.m = m;
        p.n = n * FP16_INT4_RATIO; // real_n, :429
        p.k = k;
        p.group_size = mGroupSize;
        p.weight_type = MI355Q_WEIGHT_INT4;
        p.act_type = mType;
        p.workspace = workspace;
        // only what getWorkspaceSize promised (the size configurePlugin computed for maxM); without a configure call
        // nothing is claimed and the kernels run their workspace-free form (ksplit = 1)
        if (m_cfgMaxM >= 0)
            MQ_CHECK(m <= m_cfgMaxM && p.n <= m_cfgN && k <= m_cfgK,
                "enqueue shape m=%d n=%d k=%d outside the configured maxima m=%d n=%d k=%d", m, p.n, k, m_cfgMaxM, m_cfgN, m_cfgK);
        p.workspace_bytes = workspace ? m_workspaceMaxSize : 0;
        weight_only_gemm_entry(p, stream);
    }
    size_t serializationSize() const override // :552-559 (minus the profiler's tactic table)
    {
        return sizeof(int32_t) * 3 + sizeof(GemmDims);
    }
    void serialize(char* buf) const override
    {
        write(buf, mType);
        write(buf, mQuantAlgo);
        write(buf, mGroupSize);
        write(buf, mDims);
    }
};

// ===================================================================== WeightOnlyQuantMatmul
// plugins/weightOnlyQuantMatmulPlugin/weightOnlyQuantMatmulPlugin.{h,cpp}
struct WeightOnlyQuantMatmulPlugin : PluginBase
{
    int32_t mType, mWeightTypeId;
    GemmDims mDims;
    size_t m_workspaceMaxSize = 0;
    int m_cfgMaxM = -1, m_cfgN = -1, m_cfgK = -1; // maxima of the last configurePlugin (what the workspace was sized for)
    WeightOnlyQuantMatmulPlugin(int type, int weightTypeId)
        : mType(type)
        , mWeightTypeId(weightTypeId)
    {
        MQ_CHECK(weightTypeId == MI355Q_WEIGHT_INT8 || weightTypeId == MI355Q_WEIGHT_INT4, "weight_type_id %d",
            weightTypeId);
        if (mType != MI355Q_HALF && mType != MI355Q_BF16)
            MQ_UNSUPPORTED("Unsupported data type");
    }
    int ratio() const
    {
        return mWeightTypeId == MI355Q_WEIGHT_INT4 ? 2 : 1; // INT8_INT4_RATIO, :317-320
    }
    const char* type() const override
    {
        return "WeightOnlyQuantMatmul";
    }
    PluginBase* clone() const override
    {
        return new WeightOnlyQuantMatmulPlugin(*this);
    }
    void getOutputDimensions(int idx, const mi355q_dims* in, int nbIn, mi355q_dims* out) const override // :168-205
    {
        MQ_CHECK(nbIn == 3 && idx == 0, "nbInputs / outputIndex");
        MQ_CHECK(in[0].nbDims >= 2 && in[1].nbDims == 2, "ranks");
        *out = in[0];
        out->d[in[0].nbDims - 1] = in[1].d[1] * ratio();
    }
    bool supportsFormatCombination(int pos, const mi355q_tensor_desc* io, int, int) const override // :207-241
    {
        switch (pos)
        {
        case 0: return io[0].type == mType && io[0].format == 0;
        case 1: return io[1].type == MI355Q_INT8 && io[1].format == 0; // int4 weights are reported as kINT8
        case 2: return io[2].type == mType && io[2].format == 0;
        case 3: return io[3].type == mType && io[3].format == 0;
        default: MQ_CHECK(false, "pos %d", pos); return false;
        }
    }
    int getOutputDataType(int idx, const int32_t*, int) const override
    {
        MQ_CHECK(idx == 0, "index");
        return mType;
    }
    void configure(const mi355q_dynamic_tensor_desc* in, int, const mi355q_dynamic_tensor_desc*, int) override // :243-266
    {
        const int minM = rows_of(in[0].min), maxM = rows_of(in[0].max);
        const int maxK = in[0].max.d[in[0].max.nbDims - 1];
        const int maxN = in[1].max.d[1] * ratio();
        if (!mDims.isInitialized())
            mDims = {minM, maxM, maxN / ratio(), maxK};
        m_workspaceMaxSize = mi355q_weight_only_gemm_workspace_size(maxM, maxN, maxK, mWeightTypeId);
        m_cfgMaxM = maxM, m_cfgN = maxN, m_cfgK = maxK;
    }
    size_t workspaceSize(const mi355q_tensor_desc*, int, const mi355q_tensor_desc*, int) const override
    {
        return m_workspaceMaxSize;
    }
    void enqueue(const mi355q_tensor_desc* inDesc, const mi355q_tensor_desc*, const void* const* inputs,
        void* const* outputs, void* workspace, hipStream_t stream) override // :273-347
    {
        mi355q_weight_only_params p{};
        p.m = rows_of(inDesc[0].dims);
        p.n = inDesc[1].dims.d[1] * ratio();
        p.k = inDesc[0].dims.d[inDesc[0].dims.nbDims - 1];
        p.in = inputs[0];
        p.qweight = inputs[1];
        p.scales = inputs[2];
        p.out = outputs[0];
        p.group_size = 0; // PER_COLUMN_SCALE_ONLY
        p.weight_type = mWeightTypeId;
        p.act_type = mType;
        p.workspace = workspace;
        if (m_cfgMaxM >= 0)
            MQ_CHECK(p.m <= m_cfgMaxM && p.n <= m_cfgN && p.k <= m_cfgK,
                "enqueue shape m=%d n=%d k=%d outside the configured maxima m=%d n=%d k=%d", p.m, p.n, p.k, m_cfgMaxM, m_cfgN, m_cfgK);
        p.workspace_bytes = workspace ? m_workspaceMaxSize : 0; // what getWorkspaceSize promised, nothing more
        weight_only_gemm_entry(p, stream);
    }
    size_t serializationSize() const override // :384-399
    {
        return sizeof(int32_t) * 2 + sizeof(GemmDims);
    }
    void serialize(char* buf) const override
    {
        write(buf, mType);
        write(buf, mWeightTypeId);
        write(buf, mDims);
    }
};

// ===================================================================== SmoothQuantGemm
// plugins/smoothQuantGemmPlugin/smoothQuantGemmPlugin.{h,cpp}; QuantMode bits common/quantization.h:44-91
struct SmoothQuantGemmPlugin : PluginBase
{
    int32_t mQuantMode; // perChannelScaling = 1<<3, perTokenScaling = 1<<4
    int32_t mType;
    GemmDims mDims;
    SmoothQuantGemmPlugin(int quantMode, int type)
        : mQuantMode(quantMode)
        , mType(type)
    {
        // init(), :106-130
        if (mType != MI355Q_HALF && mType != MI355Q_FLOAT && mType != MI355Q_INT32 && mType != MI355Q_BF16)
            MQ_UNSUPPORTED("Support for the requested SmoothQuant output type is missing");
    }
    bool perChannel() const
    {
        return mQuantMode & (1 << 3);
    }
    bool perToken() const
    {
        return mQuantMode & (1 << 4);
    }
    const char* type() const override
    {
        return "SmoothQuantGemm";
    }
    PluginBase* clone() const override
    {
        return new SmoothQuantGemmPlugin(*this);
    }
    void getOutputDimensions(int idx, const mi355q_dims* in, int nbIn, mi355q_dims* out) const override // :143-166
    {
        MQ_CHECK(nbIn == 4 && idx == 0, "nbInputs / outputIndex");
        MQ_CHECK(in[0].nbDims >= 2 && in[1].nbDims == 2, "ranks");
        *out = in[0];
        out->d[in[0].nbDims - 1] = in[1].d[0];
    }
    bool supportsFormatCombination(int pos, const mi355q_tensor_desc* io, int, int) const override // :168-190
    {
        switch (pos)
        {
        case 0:
        case 1: return io[pos].type == MI355Q_INT8 && io[pos].format == 0;
        case 2:
        case 3: return io[pos].type == MI355Q_FLOAT && io[pos].format == 0;
        case 4: return io[pos].type == mType && io[pos].format == 0;
        default: MQ_CHECK(false, "pos %d", pos); return false;
        }
    }
    int getOutputDataType(int idx, const int32_t*, int) const override
    {
        MQ_CHECK(idx == 0, "index");
        return mType;
    }
    void configure(const mi355q_dynamic_tensor_desc* in, int, const mi355q_dynamic_tensor_desc*, int) override
    {
        if (!mDims.isInitialized())
            mDims = {rows_of(in[0].min), rows_of(in[0].max), in[1].max.d[0], in[0].max.d[in[0].max.nbDims - 1]};
    }
    // :200-203 (m_workspaceMaxSize = the runner's split-k workspace): here the partial tiles of the stream-K kernel, a
    // bound that holds for every m the profile allows
    size_t workspaceSize(const mi355q_tensor_desc* in, int nbIn, const mi355q_tensor_desc*, int) const override
    {
        if (nbIn < 2 || in[0].dims.nbDims < 1)
            return 0;
        return gemm8_workspace_size_max(in[1].dims.d[0], in[0].dims.d[in[0].dims.nbDims - 1]);
    }
    void enqueue(const mi355q_tensor_desc* inDesc, const mi355q_tensor_desc*, const void* const* inputs,
        void* const* outputs, void* workspace, hipStream_t stream) override // :220-247
    {
        const int m = rows_of(inDesc[0].dims);
        const int n = inDesc[1].dims.d[0];
        const int k = inDesc[0].dims.d[inDesc[0].dims.nbDims - 1];
        int8_gemm(static_cast<const int8_t*>(inputs[0]), static_cast<const int8_t*>(inputs[1]), perToken(),
            perChannel(), static_cast<const float*>(inputs[3]), static_cast<const float*>(inputs[2]), outputs[0], mType,
            m, n, k, workspace, workspace ? gemm8_workspace_size_max(n, k) : 0, stream);
    }
    size_t serializationSize() const override // :288-296
    {
        return sizeof(int32_t) * 2 + sizeof(GemmDims);
    }
    void serialize(char* buf) const override
    {
        write(buf, mQuantMode);
        write(buf, mType);
        write(buf, mDims);
    }
};

// ===================================================================== QuantizePerToken
// plugins/quantizePerTokenPlugin/quantizePerTokenPlugin.cpp: in [.., K] fp16/bf16/fp32 ->
// out0 int8 [.., K], out1 fp32 [.., 1]
struct QuantizePerTokenPlugin : PluginBase
{
    const char* type() const override
    {
        return "QuantizePerToken";
    }
    int nbOutputs() const override
    {
        return 2;
    }
    PluginBase* clone() const override
    {
        return new QuantizePerTokenPlugin(*this);
    }
    void getOutputDimensions(int idx, const mi355q_dims* in, int nbIn, mi355q_dims* out) const override
    {
        MQ_CHECK(nbIn == 1 && idx < 2, "nbInputs / outputIndex");
        *out = in[0];
        if (idx == 1)
            out->d[in[0].nbDims - 1] = 1;
    }
    bool supportsFormatCombination(int pos, const mi355q_tensor_desc* io, int, int) const override
    {
        switch (pos)
        {
        case 0:
            return (io[0].type == MI355Q_FLOAT || io[0].type == MI355Q_HALF || io[0].type == MI355Q_BF16)
                && io[0].format == 0;
        case 1: return io[1].type == MI355Q_INT8 && io[1].format == 0;
        case 2: return io[2].type == MI355Q_FLOAT && io[2].format == 0;
        default: MQ_CHECK(false, "pos %d", pos); return false;
        }
    }
    int getOutputDataType(int idx, const int32_t*, int) const override
    {
        MQ_CHECK(idx < 2, "index");
        return idx == 0 ? MI355Q_INT8 : MI355Q_FLOAT;
    }
    void enqueue(const mi355q_tensor_desc* inDesc, const mi355q_tensor_desc*, const void* const* inputs,
        void* const* outputs, void*, hipStream_t stream) override
    {
        const int m = rows_of(inDesc[0].dims);
        const int k = inDesc[0].dims.d[inDesc[0].dims.nbDims - 1];
        per_token_quantization(
            static_cast<int8_t*>(outputs[0]), inputs[0], m, k, static_cast<float*>(outputs[1]), inDesc[0].type, stream);
    }
    size_t serializationSize() const override
    {
        return 0;
    }
    void serialize(char*) const override {}
};

// ===================================================================== QuantizeTensor
// plugins/quantizeTensorPlugin/quantizeTensorPlugin.cpp: inputs activation [M(*), K] (float / half), scale [1, 1] fp32;
// output int8 [M(*), K]
struct QuantizeTensorPlugin : PluginBase
{
    const char* type() const override
    {
        return "QuantizeTensor";
    }
    PluginBase* clone() const override
    {
        return new QuantizeTensorPlugin(*this);
    }
    void getOutputDimensions(int idx, const mi355q_dims* in, int nbIn, mi355q_dims* out) const override // :51-66
    {
        MQ_CHECK(nbIn == 2 && idx < 1, "nbInputs / outputIndex");
        *out = in[0];
    }
    bool supportsFormatCombination(int pos, const mi355q_tensor_desc* io, int, int) const override // :68-85
    {
        switch (pos)
        {
        case 0:
            return (io[0].type == MI355Q_FLOAT || io[0].type == MI355Q_HALF || io[0].type == MI355Q_BF16)
                && io[0].format == 0;
        case 1: return io[1].type == MI355Q_FLOAT && io[1].format == 0;
        case 2: return io[2].type == MI355Q_INT8 && io[2].format == 0;
        default: MQ_CHECK(false, "pos %d", pos); return false;
        }
    }
    int getOutputDataType(int idx, const int32_t*, int nbIn) const override
    {
        MQ_CHECK(nbIn == 2 && idx == 0, "nbInputs / index");
        return MI355Q_INT8;
    }
    void enqueue(const mi355q_tensor_desc* inDesc, const mi355q_tensor_desc*, const void* const* inputs,
        void* const* outputs, void*, hipStream_t stream) override // :99-127
    {
        const int64_t numElts = dims_prod(inDesc[0].dims, 0, inDesc[0].dims.nbDims);
        quantize_tensor(static_cast<int8_t*>(outputs[0]), inputs[0], numElts, static_cast<const float*>(inputs[1]),
            inDesc[0].type, stream);
    }
    size_t serializationSize() const override
    {
        return 0;
    }
    void serialize(char*) const override {}
};

// ===================================================================== RmsnormQuantization / LayernormQuantization
// plugins/rmsnormQuantizationPlugin/rmsnormQuantizationPlugin.cpp, plugins/layernormQuantizationPlugin/...:
// inputs: input [M(*), N], weight [N], bias [N] (activation type), scale_to_int [1] fp32;
// outputs: int8 [M(*), N], and with dyn_act_scaling the per-token scales [M(*), 1] fp32
struct NormQuantizationPlugin : PluginBase
{
    bool mLayer; // LayerNorm (else RMSNorm)
    float mEps;
    bool mUseDiffOfSquares, mDynActScaling;
    int32_t mType;
    NormQuantizationPlugin(bool layer, float eps, bool diffOfSquares, bool dyn, int type)
        : mLayer(layer)
        , mEps(eps)
        , mUseDiffOfSquares(diffOfSquares)
        , mDynActScaling(dyn)
        , mType(type)
    {
        // the reference enqueues kHALF and kFLOAT only (rmsnormQuantizationPlugin.cpp:149-162); bf16 is an extension
        if (mType != MI355Q_HALF && mType != MI355Q_FLOAT && mType != MI355Q_BF16)
            MQ_UNSUPPORTED("%s: unsupported data type %d", type_name(), mType);
    }
    const char* type_name() const
    {
        return mLayer ? "LayernormQuantization" : "RmsnormQuantization";
    }
    const char* type() const override
    {
        return type_name();
    }
    int nbOutputs() const override
    {
        return 1 + (mDynActScaling ? 1 : 0);
    }
    PluginBase* clone() const override
    {
        return new NormQuantizationPlugin(*this);
    }
    void getOutputDimensions(int idx, const mi355q_dims* in, int, mi355q_dims* out) const override // :60-87
    {
        MQ_CHECK(idx == 0 || (idx == 1 && mDynActScaling), "outputIndex %d", idx);
        *out = in[0];
        if (idx == 1)
            out->d[in[0].nbDims - 1] = 1;
    }
    bool supportsFormatCombination(int pos, const mi355q_tensor_desc* io, int nbIn, int) const override // :89-112
    {
        const int totalPoses = 6 + (mDynActScaling ? 1 : 0);
        MQ_CHECK(0 <= pos && pos < totalPoses, "pos %d", pos);
        MQ_CHECK(nbIn == 4, "nbInputs %d", nbIn);
        if (pos < 3)
            return io[pos].type == mType && io[pos].format == 0;
        if (pos == 3)
            return io[pos].type == MI355Q_FLOAT && io[pos].format == 0;
        if (pos == 4)
            return io[pos].type == MI355Q_INT8 && io[pos].format == 0;
        return io[pos].type == MI355Q_FLOAT && io[pos].format == 0;
    }
    int getOutputDataType(int idx, const int32_t*, int) const override
    {
        MQ_CHECK((mDynActScaling && idx < 2) || (!mDynActScaling && idx == 0), "index %d", idx);
        return idx == 0 ? MI355Q_INT8 : MI355Q_FLOAT;
    }
    void enqueue(const mi355q_tensor_desc* inDesc, const mi355q_tensor_desc*, const void* const* inputs,
        void* const* outputs, void*, hipStream_t stream) override // :125-165
    {
        const int m = rows_of(inDesc[0].dims);
        const int n = inDesc[1].dims.d[0];
        MQ_CHECK(inDesc[0].dims.d[inDesc[0].dims.nbDims - 1] == n, "weight length %d != hidden size", n);
        const float* scale = static_cast<const float*>(inputs[3]);
        float* dynamic_scale = mDynActScaling ? static_cast<float*>(outputs[1]) : nullptr;
        int8_t* output = static_cast<int8_t*>(outputs[0]);
        if (mLayer)
            layer_norm(nullptr, inputs[0], inputs[1], inputs[2], mEps, m, n, mUseDiffOfSquares, scale, dynamic_scale,
                output, mType, stream);
        else
            rms_norm(nullptr, inputs[0], inputs[1], inputs[2], mEps, m, n, scale, dynamic_scale, output, mType, stream);
    }
    size_t serializationSize() const override // float eps, [bool use_diff_of_squares,] bool dyn, int32 type
    {
        return sizeof(float) + (mLayer ? 1 : 0) + 1 + sizeof(int32_t);
    }
    void serialize(char* buf) const override
    {
        write(buf, mEps);
        if (mLayer)
            write(buf, mUseDiffOfSquares);
        write(buf, mDynActScaling);
        write(buf, mType);
    }
};

// ===================================================================== AllReduce
// plugins/ncclPlugin/allreducePlugin.{h,cpp}: inputs: tensor (type_id); when strategy != RING a HOST int64 tensor
// [4 * tp_size] of peer pointers (ping | pong | barrier-in | barrier-out, CustomAllReduceHelper.allocate_workspace,
// tensorrt_llm/plugin/plugin.py:386-403).  Output: the summed tensor.
struct AllReducePlugin : PluginBase
{
    std::vector<int32_t> mGroup;
    int32_t mType;
    int8_t mStrategy;
    int32_t mCounter;
    int32_t mRank; // this process's rank inside the group; -1 = $RANK % nRanks at enqueue (the reference asks MPI)
    AllReducePlugin(std::vector<int32_t> group, int type, int strategy, int counter, int rank)
        : mGroup(std::move(group))
        , mType(type)
        , mStrategy((int8_t) strategy)
        , mCounter(counter)
        , mRank(rank)
    {
        MQ_CHECK(mStrategy >= MI355Q_AR_RING && mStrategy <= MI355Q_AR_AUTO, "strategy %d", (int) mStrategy);
        if (mType != MI355Q_FLOAT && mType != MI355Q_HALF && mType != MI355Q_BF16)
            MQ_UNSUPPORTED("AllReduce: unsupported data type %d", mType);
    }
    const char* type() const override
    {
        return "AllReduce";
    }
    PluginBase* clone() const override
    {
        return new AllReducePlugin(*this);
    }
    void getOutputDimensions(int idx, const mi355q_dims* in, int, mi355q_dims* out) const override // :76-80
    {
        MQ_CHECK(idx == 0, "outputIndex %d", idx);
        *out = in[0];
    }
    bool supportsFormatCombination(int pos, const mi355q_tensor_desc* io, int nbIn, int) const override // :82-98
    {
        if (mStrategy != MI355Q_AR_RING && pos == 1 && nbIn == 2)
            return io[pos].type == MI355Q_INT64 && io[pos].format == 0;
        return io[pos].type == mType && io[pos].format == 0;
    }
    int getOutputDataType(int idx, const int32_t* inTypes, int) const override
    {
        MQ_CHECK(idx == 0, "index %d", idx);
        return inTypes[0];
    }
    void enqueue(const mi355q_tensor_desc* inDesc, const mi355q_tensor_desc*, const void* const* inputs,
        void* const* outputs, void*, hipStream_t stream) override // :141-203
    {
        const int64_t size = dims_prod(inDesc[0].dims, 0, inDesc[0].dims.nbDims);
        const size_t sizePerElem = dtype_size(mType);
        int runtimeStrategy = mStrategy;
        if (runtimeStrategy == MI355Q_AR_RING) // :181-185
        {
            comm_all_reduce(mGroup.data(), (int) mGroup.size(), inputs[0], outputs[0], (size_t) size, mType, stream);
            return;
        }
        // the workspace table: 4 pointers per rank (NUM_POINTERS_PER_RANK), optionally followed by the entries this
        // library's CustomAllReduceHelper appends: the byte size of a comm buffer, the host-visible timeout record, and
        // (third, optional) the workgroup bound for ranks that share one GPU
        const int len = inDesc[1].dims.d[0];
        const int nRanks = (int) mGroup.size();
        MQ_CHECK(nRanks >= 2 && len >= 4 * nRanks && len <= 4 * nRanks + 3 && len != 4 * nRanks + 1,
            "all_reduce_workspace length %d for %d ranks", len, nRanks);
        const int64_t* table = static_cast<const int64_t*>(inputs[1]);
        int myRank = mRank;
        if (myRank < 0)
        {
            const char* e = getenv("RANK");
            MQ_CHECK(e, "AllReduce: no 'rank' field and no RANK in the environment");
            myRank = atoi(e);
        }
        myRank %= nRanks;
        mi355q_allreduce_params params;
        allreduce_params_deserialize(params, static_cast<const void* const*>(inputs[1]), nRanks, myRank, mCounter);
        params.local_input_buffer_ptr = inputs[0]; // pushed to the peers by the kernel: no staging copy (allreducePlugin.cpp:195-199)
        params.comm_buffer_bytes = mi355q_allreduce_max_workspace_size(nRanks);
        if (len >= 4 * nRanks + 2)
        {
            if (len == 4 * nRanks + 3)
                params.max_workgroups = (uint32_t) table[4 * nRanks + 2];
            params.comm_buffer_bytes = (size_t) table[4 * nRanks];
            params.timeout_flag = reinterpret_cast<uint32_t*>(table[4 * nRanks + 1]);
            if (params.timeout_flag)
                MQ_CHECK(*static_cast<volatile uint32_t*>(params.timeout_flag) == 0,
                    "AllReduce: an earlier all-reduce timed out waiting for a peer rank (its output was poisoned with NaN)");
        }
        if (mStrategy == MI355Q_AR_AUTO) // selectImplementation, :111-139, for the real buffer size
            runtimeStrategy = mi355q_allreduce_select_implementation(size * sizePerElem, nRanks, params.comm_buffer_bytes);
        if (runtimeStrategy == MI355Q_AR_RING) // AUTO beyond the comm buffers (:115-118)
        {
            comm_all_reduce(mGroup.data(), (int) mGroup.size(), inputs[0], outputs[0], (size_t) size, mType, stream);
            return;
        }
        custom_all_reduce(params, outputs[0], size, mType, runtimeStrategy, stream);
    }
    size_t serializationSize() const override // :268-284
    {
        return sizeof(int32_t) + sizeof(int8_t) + sizeof(int32_t) + sizeof(int32_t) * mGroup.size();
    }
    void serialize(char* buf) const override
    {
        write(buf, mType);
        write(buf, mStrategy);
        write(buf, mCounter);
        for (int32_t g : mGroup)
            write(buf, g);
    }
};

// ===================================================================== AllGather
// plugins/ncclPlugin/allgatherPlugin.{h,cpp}: input 0 the local tensor (type_id); output = the group's tensors back to
// back along dim 0 (:66-72, :96-110).  The reference's form -- one input, ncclAllGather on the group's communicator --
// is what a created-without-"strategy" or deserialized instance runs (RCCL through rccl_comm.cpp).  An instance created
// with the extra field strategy = ONESHOT takes the AllReduce plugin's HOST int64 workspace table as a second input and
// gathers through the push kernel on those IPC buffers (the flag value comes from the buffers' device-resident epoch:
// no counter is needed, so there is nothing a reference-built network would have to supply).  The serialized form is
// the reference's (type, group): it does not carry the strategy, a deserialized instance is the RCCL form.
struct AllGatherPlugin : PluginBase
{
    std::vector<int32_t> mGroup;
    int32_t mType;
    int32_t mStrategy, mRank;
    AllGatherPlugin(std::vector<int32_t> group, int type, int strategy, int rank)
        : mGroup(std::move(group))
        , mType(type)
        , mStrategy(strategy)
        , mRank(rank)
    {
        MQ_CHECK(dtype_size(mType) > 0, "AllGather: data type %d", mType);
        MQ_CHECK(mStrategy == MI355Q_AR_RING || mStrategy == MI355Q_AR_ONESHOT, "AllGather: strategy %d (RING or ONESHOT)", mStrategy);
    }
    const char* type() const override
    {
        return "AllGather";
    }
    PluginBase* clone() const override
    {
        return new AllGatherPlugin(*this);
    }
    void getOutputDimensions(int idx, const mi355q_dims* in, int, mi355q_dims* out) const override // :66-72
    {
        MQ_CHECK(idx == 0, "outputIndex %d", idx);
        *out = in[0];
        out->d[0] *= (int) mGroup.size();
    }
    bool supportsFormatCombination(int pos, const mi355q_tensor_desc* io, int nbIn, int) const override // :74-79
    {
        if (mStrategy != MI355Q_AR_RING && pos == 1 && nbIn == 2)
            return io[pos].type == MI355Q_INT64 && io[pos].format == 0;
        return io[pos].type == mType && io[pos].format == 0;
    }
    int getOutputDataType(int idx, const int32_t* inTypes, int) const override
    {
        MQ_CHECK(idx == 0, "index %d", idx);
        return inTypes[0];
    }
    void enqueue(const mi355q_tensor_desc* inDesc, const mi355q_tensor_desc*, const void* const* inputs,
        void* const* outputs, void*, hipStream_t stream) override // :90-110
    {
        const int64_t size = dims_prod(inDesc[0].dims, 0, inDesc[0].dims.nbDims);
        const int nRanks = (int) mGroup.size();
        if (mStrategy == MI355Q_AR_RING) // ncclAllGather(inputs[0], outputs[0], size, dtype, comm, stream), as bytes
        {
            comm_all_gather(mGroup.data(), nRanks, inputs[0], outputs[0], (size_t) size * dtype_size(mType), MI355Q_INT8, stream);
            return;
        }
        const int len = inDesc[1].dims.d[0];
        MQ_CHECK(nRanks >= 2 && len >= 4 * nRanks && len <= 4 * nRanks + 3 && len != 4 * nRanks + 1,
            "all_reduce_workspace length %d for %d ranks", len, nRanks);
        const int64_t* table = static_cast<const int64_t*>(inputs[1]);
        int myRank = mRank;
        if (myRank < 0)
        {
            const char* e = getenv("RANK");
            MQ_CHECK(e, "AllGather: no 'rank' field and no RANK in the environment");
            myRank = atoi(e);
        }
        myRank %= nRanks;
        mi355q_allreduce_params params;
        allreduce_params_deserialize(params, static_cast<const void* const*>(inputs[1]), nRanks, myRank, 0);
        params.comm_buffer_bytes = mi355q_allreduce_max_workspace_size(nRanks);
        if (len >= 4 * nRanks + 2)
        {
            if (len == 4 * nRanks + 3)
                params.max_workgroups = (uint32_t) table[4 * nRanks + 2];
            params.comm_buffer_bytes = (size_t) table[4 * nRanks];
            params.timeout_flag = reinterpret_cast<uint32_t*>(table[4 * nRanks + 1]);
            if (params.timeout_flag)
                MQ_CHECK(*static_cast<volatile uint32_t*>(params.timeout_flag) == 0,
                    "AllGather: an earlier collective timed out waiting for a peer rank");
        }
        const size_t bytes = (size_t) size * dtype_size(mType);
        if (bytes % 16 != 0 || (size_t) nRanks * ((bytes + 255) / 256 * 256) > params.comm_buffer_bytes)
        {
            comm_all_gather(mGroup.data(), nRanks, inputs[0], outputs[0], bytes, MI355Q_INT8, stream); // does not fit: RCCL
            return;
        }
        custom_all_gather(params, inputs[0], outputs[0], bytes, stream);
    }
    size_t serializationSize() const override // :147-150: type, then the group
    {
        return sizeof(int32_t) + sizeof(int32_t) * mGroup.size();
    }
    void serialize(char* buf) const override
    {
        write(buf, mType);
        for (int32_t g : mGroup)
            write(buf, g);
    }
};

// ===================================================================== MixtureOfExperts
// plugins/mixtureOfExperts/mixtureOfExpertsPlugin.{h,cpp} for weight-only quantised experts (weight type INT8, int4 by
// the quant mode): inputs hidden states, fp32 routing logits, expert weights 1 / 2, [biases 1 / 2], [finished], quant
// scales 1 / 2 (index compaction :186-243 of the header); output = the input's shape.
struct MixtureOfExpertsPlugin : PluginBase
{
    int32_t mNumExperts, mK, mHidden, mInter, mActivation, mType, mWeightType, mQuantMode;
    bool mUseFinished, mUseBias;
    int32_t mTPSize, mTPRank, mParallelismMode, mNormalizationMode;
    static constexpr int32_t INT4_WEIGHTS = 1 << 0, INT8_WEIGHTS = 1 << 1; // common/quantization.h QuantMode bits
    MixtureOfExpertsPlugin(int experts, int k, int hidden, int inter, int act, int type, int wtype, int qmode, bool finished,
        bool bias, int tp_size, int tp_rank, int pmode, int nmode)
        : mNumExperts(experts)
        , mK(k)
        , mHidden(hidden)
        , mInter(inter)
        , mActivation(act)
        , mType(type)
        , mWeightType(wtype)
        , mQuantMode(qmode)
        , mUseFinished(finished)
        , mUseBias(bias)
        , mTPSize(tp_size)
        , mTPRank(tp_rank)
        , mParallelismMode(pmode)
        , mNormalizationMode(nmode)
    {
        // init() :156-200: (half | bf16) x INT8-typed weights with an int4 / int8 weight-only quant mode
        if (!((mType == MI355Q_HALF || mType == MI355Q_BF16) && mWeightType == MI355Q_INT8 && hasExpertQuantScales()))
            MQ_UNSUPPORTED("Could not construct the mixture of experts plugin with the requested input combination (built: "
                           "half / bf16 activations with int8 / int4 weight-only experts; type %d, weight type %d, quant mode %d)",
                mType, mWeightType, mQuantMode);
        MQ_CHECK(mActivation >= 0 && mActivation <= 5, "activation_type %d", mActivation);
        MQ_CHECK(mParallelismMode >= 0 && mParallelismMode <= 2, "parallelism_mode %d", mParallelismMode);
    }
    bool int4() const
    {
        return (mQuantMode & INT4_WEIGHTS) != 0;
    }
    bool hasExpertQuantScales() const
    {
        return (mQuantMode & (INT4_WEIGHTS | INT8_WEIGHTS)) != 0;
    }
    bool gated() const
    {
        return mActivation == 3 || mActivation == 4;
    }
    int bias1Idx() const { return 3 + mUseBias; }
    int bias2Idx() const { return bias1Idx() + mUseBias; }
    int finishedIdx() const { return bias2Idx() + mUseFinished; }
    int scale1Idx() const { return finishedIdx() + 1; }
    int scale2Idx() const { return scale1Idx() + 1; }
    int nbInputs() const { return scale2Idx() + 1; }
    const char* type() const override
    {
        return "MixtureOfExperts";
    }
    PluginBase* clone() const override
    {
        return new MixtureOfExpertsPlugin(*this);
    }
    void getOutputDimensions(int idx, const mi355q_dims* in, int, mi355q_dims* out) const override
    {
        MQ_CHECK(idx == 0, "outputIndex %d", idx);
        *out = in[0];
    }
    bool supportsFormatCombination(int pos, const mi355q_tensor_desc* io, int nbIn, int nbOut) const override // :222-254
    {
        MQ_CHECK(nbIn == nbInputs() && nbOut == 1 && pos >= 0 && pos < nbIn + nbOut, "MixtureOfExperts: %d inputs expected", nbInputs());
        if (io[pos].format != 0)
            return false;
        if (pos == 2 || pos == 3)
            return io[pos].type == mWeightType;
        if (mUseFinished && pos == finishedIdx())
            return io[pos].type == MI355Q_BOOL;
        if (pos == 1)
            return io[pos].type == MI355Q_FLOAT;
        return io[pos].type == mType;
    }
    int getOutputDataType(int idx, const int32_t*, int) const override
    {
        MQ_CHECK(idx == 0, "index %d", idx);
        return mType;
    }
    int numTokens(const mi355q_tensor_desc* in) const // getNumTokens :336-347
    {
        const int nd = in[0].dims.nbDims;
        MQ_CHECK(nd == 2 || nd == 3, "hidden_state dimension should be either 2 [b*s, hidden], or 3 [b, s, hidden]");
        return nd == 3 ? in[0].dims.d[0] * in[0].dims.d[1] : in[0].dims.d[0];
    }
    mi355q_moe_params params(int tokens) const
    {
        mi355q_moe_params p{};
        p.num_rows = tokens;
        p.hidden_size = mHidden;
        p.inter_size = mInter;
        p.num_experts = mNumExperts;
        p.k = mK;
        p.activation_type = mActivation;
        p.act_type = mType;
        p.weight_type = int4() ? MI355Q_WEIGHT_INT4 : MI355Q_WEIGHT_INT8;
        p.ep_size = mParallelismMode == 1 ? mTPSize : 1; // getParallelismConfig :356-368
        p.ep_rank = mParallelismMode == 1 ? mTPRank : 0;
        p.tp_size = mParallelismMode == 2 ? mTPSize : 1;
        p.tp_rank = mParallelismMode == 2 ? mTPRank : 0;
        p.normalization_mode = mNormalizationMode;
        return p;
    }
    size_t workspaceSize(const mi355q_tensor_desc* in, int, const mi355q_tensor_desc*, int) const override
    {
        const mi355q_moe_params p = params(numTokens(in));
        return moe_workspace_size(p);
    }
    void enqueue(const mi355q_tensor_desc* inDesc, const mi355q_tensor_desc*, const void* const* inputs, void* const* outputs,
        void* workspace, hipStream_t stream) override // :370-431
    {
        mi355q_moe_params p = params(numTokens(inDesc));
        const auto& w1 = inDesc[2].dims;
        const auto& w2 = inDesc[3].dims;
        const int per_node = mNumExperts / p.ep_size;
        const int packed = int4() ? 2 : 1; // getWeightPackedElements
        MQ_CHECK(w1.nbDims == 3 && w1.d[0] == per_node && w2.nbDims == 3 && w2.d[0] == per_node, "expert weights must be [experts, ., .]");
        // in weight-only mode the shape is [experts, in, out (packed)] (getGemmShapeInnerDimIndex :255-259)
        MQ_CHECK(w1.d[1] == mHidden && w1.d[2] * packed == mInter * (gated() ? 2 : 1), "expert weights 1 shape");
        MQ_CHECK(w2.d[1] == mInter && w2.d[2] * packed == mHidden, "expert weights 2 shape");
        p.input = inputs[0];
        p.gating_output = static_cast<const float*>(inputs[1]);
        p.fc1_expert_weights = inputs[2];
        p.fc2_expert_weights = inputs[3];
        p.fc1_expert_biases = mUseBias ? inputs[bias1Idx()] : nullptr;
        p.fc2_expert_biases = mUseBias ? inputs[bias2Idx()] : nullptr;
        p.finished = mUseFinished ? static_cast<const uint8_t*>(inputs[finishedIdx()]) : nullptr;
        p.fc1_scales = inputs[scale1Idx()];
        p.fc2_scales = inputs[scale2Idx()];
        p.output = outputs[0];
        p.workspace = workspace;
        p.workspace_bytes = workspace ? moe_workspace_size(p) : 0;
        run_moe(p, stream);
    }
    size_t serializationSize() const override // :88-94 without the profiler's tactics and GemmDims
    {
        return 8 * sizeof(int32_t) + 2 * sizeof(bool) + 4 * sizeof(int32_t);
    }
    void serialize(char* buf) const override
    {
        write(buf, mNumExperts);
        write(buf, mK);
        write(buf, mHidden);
        write(buf, mInter);
        write(buf, mActivation);
        write(buf, mType);
        write(buf, mWeightType);
        write(buf, mQuantMode);
        write(buf, mUseFinished);
        write(buf, mUseBias);
        write(buf, mTPSize);
        write(buf, mTPRank);
        write(buf, mParallelismMode);
        write(buf, mNormalizationMode);
    }
};

// ===================================================================== Fp8Gemm (no reference plugin)
// inputs: 0 A e4m3 [.., K]; 1 B e4m3 [N, K]; 2 scale_a fp32 [1]; 3 scale_b fp32 [1]
struct Fp8GemmPlugin : PluginBase
{
    int32_t mType;
    explicit Fp8GemmPlugin(int type)
        : mType(type)
    {
        if (mType != MI355Q_HALF && mType != MI355Q_FLOAT && mType != MI355Q_BF16)
            MQ_UNSUPPORTED("Unsupported fp8 GEMM output type");
    }
    const char* type() const override
    {
        return "Fp8Gemm";
    }
    PluginBase* clone() const override
    {
        return new Fp8GemmPlugin(*this);
    }
    void getOutputDimensions(int idx, const mi355q_dims* in, int nbIn, mi355q_dims* out) const override
    {
        MQ_CHECK(nbIn == 4 && idx == 0, "nbInputs / outputIndex");
        *out = in[0];
        out->d[in[0].nbDims - 1] = in[1].d[0];
    }
    bool supportsFormatCombination(int pos, const mi355q_tensor_desc* io, int, int) const override
    {
        if (pos < 2)
            return io[pos].type == MI355Q_FP8 && io[pos].format == 0;
        if (pos < 4)
            return io[pos].type == MI355Q_FLOAT && io[pos].format == 0;
        return io[pos].type == mType && io[pos].format == 0;
    }
    int getOutputDataType(int, const int32_t*, int) const override
    {
        return mType;
    }
    size_t workspaceSize(const mi355q_tensor_desc* in, int nbIn, const mi355q_tensor_desc*, int) const override
    {
        if (nbIn < 2 || in[0].dims.nbDims < 1)
            return 0;
        return gemm8_workspace_size_max(in[1].dims.d[0], in[0].dims.d[in[0].dims.nbDims - 1]);
    }
    void enqueue(const mi355q_tensor_desc* inDesc, const mi355q_tensor_desc*, const void* const* inputs,
        void* const* outputs, void* workspace, hipStream_t stream) override
    {
        const int m = rows_of(inDesc[0].dims);
        const int n = inDesc[1].dims.d[0];
        const int k = inDesc[0].dims.d[inDesc[0].dims.nbDims - 1];
        fp8_gemm(inputs[0], inputs[1], static_cast<const float*>(inputs[2]), static_cast<const float*>(inputs[3]),
            outputs[0], mType, m, n, k, workspace, workspace ? gemm8_workspace_size_max(n, k) : 0, stream);
    }
    size_t serializationSize() const override
    {
        return sizeof(int32_t);
    }
    void serialize(char* buf) const override
    {
        write(buf, mType);
    }
};

// ===================================================================== GPTAttention
// plugins/gptAttentionPlugin/gptAttentionPlugin.{h,cpp} + gptAttentionCommon/gptAttentionCommon.cpp
struct GptAttentionFields // constructor arguments, gptAttentionPlugin.h:73-84
{
    int32_t layer_idx = 0, num_heads = 0, num_kv_heads = 0, head_size = 0, unidirectional = 1;
    float q_scaling = 1.f;
    int32_t position_embedding_type = 0, rotary_embedding_dim = 0;
    float rotary_embedding_base = 10000.f;
    int32_t rotary_embedding_scale_type = 0;
    float rotary_embedding_scale = 1.f;
    int32_t rotary_embedding_max_positions = 0, tp_size = 1, tp_rank = 0, unfuse_qkv_gemm = 0;
    int32_t context_fmha_type = 0, multi_block_mode = 0, enable_xqa = 0, kv_cache_quant_mode = 0;
    int32_t remove_input_padding = 0, mask_type = 1, paged_kv_cache = 0, tokens_per_block = 0;
    int32_t type = MI355Q_HALF, max_context_length = 0, qkv_bias_enabled = 0, cross_attention = 0;
    int32_t max_distance = 0, pos_shift_enabled = 0, dense_context_fmha = 0, use_paged_context_fmha = 0;
    int32_t use_cache = 1, is_medusa_enabled = 0;
};

struct GPTAttentionPlugin : PluginBase
{
    GptAttentionFields f;
    enum IdxEntry // gptAttentionPlugin.h:150-177
    {
        QKV_TENSOR,
        K_TENSOR,
        V_TENSOR,
        SEQUENCE_LENGTH,
        HOST_PAST_KEY_VALUE_LENGTHS,
        HOST_MAX_ATTENTION_WINDOW,
        HOST_SINK_TOKEN_LENGTH,
        CONTEXT_LENGTHS,
        CACHE_INDIR,
        REQUEST_TYPES,
        KV_CACHE_BLOCK_POINTERS,
        HOST_KV_CACHE_BLOCK_POINTERS,
        PAST_KEY_VALUE,
        KV_CACHE_QUANTIZATION_SCALE,
        KV_CACHE_DEQUANTIZATION_SCALE,
        ALIBI_SLOPES,
        RELATIVE_ATTENTION_BIAS,
        CROSS_QKV,
        CROSS_QKV_LENGTH,
        ENCODER_INPUT_LENGTH,
        HOST_CONTEXT_LENGTH,
        QKV_BIAS_TENSOR,
        MEDUSA_PACKED_MASK,
        MEDUSA_POSITION_OFFSETS,
        ENUM_SIZE
    };
    int mEntryIdx[ENUM_SIZE];

    explicit GPTAttentionPlugin(const GptAttentionFields& fields)
        : f(fields)
    {
        MQ_CHECK(f.num_heads > 0 && f.head_size > 0, "num_heads / head_size");
        if (f.num_kv_heads <= 0)
            f.num_kv_heads = f.num_heads;
        MQ_CHECK(f.num_heads % f.num_kv_heads == 0, "num_heads %% num_kv_heads");
        if (f.type != MI355Q_HALF && f.type != MI355Q_BF16)
            MQ_UNSUPPORTED("GPTAttention: only fp16 / bf16 activations are built (type_id %d)", f.type);
        initEntryIdx();
    }
    bool hasKvQuant() const
    {
        return f.kv_cache_quant_mode & ((1 << 6) | (1 << 7)); // int8KvCache | fp8KvCache
    }
    bool isEntryUsed(int e) const // gptAttentionPlugin.cpp:67-97
    {
        const bool cache = f.use_cache;
        switch (e)
        {
        case QKV_TENSOR: return true;
        case K_TENSOR:
        case V_TENSOR: return f.unfuse_qkv_gemm;
        case SEQUENCE_LENGTH:
        case HOST_PAST_KEY_VALUE_LENGTHS: return cache;
        case HOST_MAX_ATTENTION_WINDOW:
        case HOST_SINK_TOKEN_LENGTH:
        case CONTEXT_LENGTHS: return true;
        case CACHE_INDIR: return cache;
        case REQUEST_TYPES: return true;
        case KV_CACHE_BLOCK_POINTERS:
        case HOST_KV_CACHE_BLOCK_POINTERS: return cache && f.paged_kv_cache;
        case PAST_KEY_VALUE: return cache && !f.paged_kv_cache;
        case KV_CACHE_QUANTIZATION_SCALE:
        case KV_CACHE_DEQUANTIZATION_SCALE: return cache && hasKvQuant();
        case ALIBI_SLOPES: return f.position_embedding_type == 3 || f.position_embedding_type == 4;
        case RELATIVE_ATTENTION_BIAS: return f.position_embedding_type == 5;
        case CROSS_QKV:
        case CROSS_QKV_LENGTH:
        case ENCODER_INPUT_LENGTH: return f.cross_attention;
        case HOST_CONTEXT_LENGTH: return f.remove_input_padding;
        case QKV_BIAS_TENSOR: return f.qkv_bias_enabled;
        case MEDUSA_PACKED_MASK:
        case MEDUSA_POSITION_OFFSETS: return f.is_medusa_enabled;
        default: return false;
        }
    }
    void initEntryIdx() // :99-108
    {
        int idx = 0;
        for (int i = 0; i < ENUM_SIZE; ++i)
        {
            mEntryIdx[i] = idx;
            idx += isEntryUsed(i);
        }
    }
    int getIdx(int e) const
    {
        MQ_CHECK(isEntryUsed(e), "getIdx() should not be used with entry %d", e);
        return mEntryIdx[e];
    }
    int nbInputsExpected() const
    {
        int n = 0;
        for (int i = 0; i < ENUM_SIZE; ++i)
            n += isEntryUsed(i);
        return n;
    }
    int hiddenDimIndex() const
    {
        return f.remove_input_padding ? 1 : 2;
    }
    int kvType() const
    {
        if (f.kv_cache_quant_mode & (1 << 6))
            return MI355Q_INT8;
        if (f.kv_cache_quant_mode & (1 << 7))
            return MI355Q_FP8;
        return f.type;
    }
    const char* type() const override
    {
        return "GPTAttention";
    }
    int nbOutputs() const override // present_key_value only for the non-paged cache
    {
        return (f.paged_kv_cache || !f.use_cache) ? 1 : 2;
    }
    PluginBase* clone() const override
    {
        return new GPTAttentionPlugin(*this);
    }
    void getOutputDimensions(int idx, const mi355q_dims* in, int, mi355q_dims* out) const override // :152-164
    {
        MQ_CHECK(idx == 0 || (!f.paged_kv_cache && f.use_cache && idx == 1), "outputIndex %d", idx);
        if (idx == 0)
        {
            *out = in[getIdx(QKV_TENSOR)];
            out->d[hiddenDimIndex()] = f.head_size * f.num_heads;
        }
        else
            *out = in[getIdx(PAST_KEY_VALUE)];
    }
    bool supportsFormatCombination(int pos, const mi355q_tensor_desc* io, int nbIn, int) const override // :166-230
    {
        auto is = [&](int e) { return isEntryUsed(e) && pos == mEntryIdx[e]; };
        if (is(CONTEXT_LENGTHS) || is(REQUEST_TYPES) || is(HOST_MAX_ATTENTION_WINDOW) || is(HOST_SINK_TOKEN_LENGTH)
            || is(MEDUSA_PACKED_MASK) || is(MEDUSA_POSITION_OFFSETS) || is(SEQUENCE_LENGTH)
            || is(HOST_PAST_KEY_VALUE_LENGTHS) || is(CACHE_INDIR) || is(HOST_CONTEXT_LENGTH)
            || is(ENCODER_INPUT_LENGTH) || is(CROSS_QKV_LENGTH))
            return io[pos].type == MI355Q_INT32;
        if (is(KV_CACHE_QUANTIZATION_SCALE) || is(KV_CACHE_DEQUANTIZATION_SCALE))
            return io[pos].type == MI355Q_FLOAT && io[pos].format == 0;
        if (is(KV_CACHE_BLOCK_POINTERS) || is(HOST_KV_CACHE_BLOCK_POINTERS))
            return io[pos].type == MI355Q_INT64 && io[pos].format == 0;
        if (is(PAST_KEY_VALUE) || (!f.paged_kv_cache && f.use_cache && pos == nbIn + 1))
            return io[pos].type == kvType() && io[pos].format == 0;
        return io[pos].type == f.type && io[pos].format == 0;
    }
    int getOutputDataType(int idx, const int32_t* inTypes, int) const override // :300-311
    {
        MQ_CHECK(idx == 0 || idx == 1, "index");
        return idx == 0 ? inTypes[0] : inTypes[getIdx(PAST_KEY_VALUE)];
    }
    mi355q_decode_attention_params baseParams(int batch, int max_blocks, int max_seq) const
    {
        mi355q_decode_attention_params p{};
        p.batch = batch;
        p.num_heads = f.num_heads;
        p.num_kv_heads = f.num_kv_heads;
        p.head_size = f.head_size;
        p.max_blocks_per_seq = max_blocks;
        p.tokens_per_block = f.tokens_per_block;
        p.max_seq_len = max_seq;
        p.act_type = f.type;
        p.kv_type = kvType();
        p.position_embedding_type = f.position_embedding_type;
        p.rotary_embedding_dim = f.rotary_embedding_dim;
        p.rotary_embedding_base = f.rotary_embedding_base;
        p.rotary_embedding_scale_type = f.rotary_embedding_scale_type;
        p.rotary_embedding_scale = f.rotary_embedding_scale;
        p.rotary_embedding_max_positions = f.rotary_embedding_max_positions;
        p.q_scaling = f.q_scaling;
        p.multi_block_mode = 0;
        p.position_shift_enabled = f.pos_shift_enabled ? 1 : 0; // mPosShiftEnabled, gptAttentionCommon.cpp:1276-1290, :1345
        return p;
    }
    mi355q_context_attention_params contextParams(int batch, int max_blocks, int max_len, int tokens) const
    {
        mi355q_context_attention_params p{};
        p.batch = batch;
        p.num_heads = f.num_heads;
        p.num_kv_heads = f.num_kv_heads;
        p.head_size = f.head_size;
        p.max_blocks_per_seq = max_blocks;
        p.tokens_per_block = f.tokens_per_block;
        p.max_context_length = max_len;
        p.num_tokens = tokens;
        p.remove_padding = f.remove_input_padding;
        p.act_type = f.type;
        p.kv_type = kvType();
        p.position_embedding_type = f.position_embedding_type;
        p.rotary_embedding_dim = f.rotary_embedding_dim;
        p.rotary_embedding_base = f.rotary_embedding_base;
        p.rotary_embedding_scale_type = f.rotary_embedding_scale_type;
        p.rotary_embedding_scale = f.rotary_embedding_scale;
        p.rotary_embedding_max_positions = f.rotary_embedding_max_positions;
        p.q_scaling = f.q_scaling;
        p.position_shift_enabled = f.pos_shift_enabled ? 1 : 0; // mPosShiftEnabled, gptAttentionCommon.cpp:834, :1276-1290, :1345
        return p;
    }
    size_t workspaceSize(const mi355q_tensor_desc* in, int, const mi355q_tensor_desc*, int) const override
    {
        // max(context, generation) like gptAttentionPlugin.cpp:312-333: the context needs the processed q|k|v
        // rows (getWorkspaceSizeForContext, gptAttentionCommon.cpp:497-560), the generation the multi-block
        // partials (:1231-1274)
        const int batch = in[getIdx(CONTEXT_LENGTHS)].dims.d[0];
        const auto& qd = in[getIdx(QKV_TENSOR)].dims;
        const int tokens = f.remove_input_padding ? qd.d[0] : qd.d[0] * qd.d[1];
        const int max_len = f.remove_input_padding ? std::min<int>(tokens, std::max(f.max_context_length, 1)) : qd.d[1];
        mi355q_context_attention_params c = contextParams(batch, 0, max_len, tokens);
        size_t ctx = mi355q_context_attention_workspace_size(&c);
        size_t gen = 0;
        if (f.use_cache && f.paged_kv_cache)
        {
            const auto& bp = in[getIdx(KV_CACHE_BLOCK_POINTERS)].dims;
            const int max_blocks = bp.d[bp.nbDims - 1];
            mi355q_decode_attention_params p = baseParams(batch, max_blocks, max_blocks * f.tokens_per_block);
            gen = mi355q_decode_attention_workspace_size(&p);
        }
        else if (f.use_cache)
        {
            const auto& kv = in[getIdx(PAST_KEY_VALUE)].dims; // [B, 2, Hkv, S, Dh]
            mi355q_decode_attention_params p = baseParams(batch, 1, kv.d[3]);
            p.kv_linear_buffer = reinterpret_cast<const void*>(uintptr_t(256)); // sizes only: marks the linear layout
            p.linear_max_seq_len = kv.d[3];
            gen = mi355q_decode_attention_workspace_size(&p);
        }
        return std::max(ctx, gen);
    }
    void enqueue(const mi355q_tensor_desc* inDesc, const mi355q_tensor_desc* outDesc, const void* const* inputs,
        void* const* outputs, void* workspace, hipStream_t stream) override // :336-393, :396-656
    {
        const int nbSeq = inDesc[getIdx(CONTEXT_LENGTHS)].dims.d[0];
        const int32_t* reqTypes = static_cast<const int32_t*>(inputs[getIdx(REQUEST_TYPES)]);
        const auto& qkvDims = inDesc[getIdx(QKV_TENSOR)].dims;
        int nbContext = 0, contextTokenIdxEnd = 0;
        for (int i = 0; i < nbSeq && reqTypes[i] == 0; ++i)
        {
            ++nbContext;
            contextTokenIdxEnd += f.remove_input_padding
                ? static_cast<const int32_t*>(inputs[getIdx(HOST_CONTEXT_LENGTH)])[i]
                : qkvDims.d[1];
        }
        for (int i = nbContext; i < nbSeq; ++i)
            MQ_CHECK(reqTypes[i] == 1, "host_request_types: context requests must come first (seq %d)", i);
        if (nbContext != 0 && nbContext != nbSeq)
            MQ_CHECK(f.remove_input_padding && f.paged_kv_cache, "mixed requests require remove_input_padding and paged_kv_cache");
        if (nbSeq == 0)
            return;
        if (f.cross_attention || f.is_medusa_enabled || f.unfuse_qkv_gemm)
            MQ_UNSUPPORTED("GPTAttention: cross attention / medusa / unfused qkv are not built");
        if (f.use_cache && !f.paged_kv_cache)
        {
            // the contiguous cache is output 1, aliased to the past_key_value input by the engine; copy when it is not
            // (gptAttentionPlugin.cpp:524-537)
            const void* past = inputs[getIdx(PAST_KEY_VALUE)];
            if (past != outputs[1])
            {
                const size_t bytes = (size_t) dims_prod(outDesc[1].dims, 0, outDesc[1].dims.nbDims) * dtype_size(kvType());
                MQ_HIP(hipMemcpyAsync(outputs[1], past, bytes, hipMemcpyDeviceToDevice, stream));
            }
        }
        const int hidden = qkvDims.d[hiddenDimIndex()];
        MQ_CHECK(hidden == (f.num_heads + 2 * f.num_kv_heads) * f.head_size, "qkv hidden size %d", hidden);
        const int outHidden = f.num_heads * f.head_size;
        const size_t esz = dtype_size(f.type);
        if (nbContext > 0)
            enqueueContext(0, nbContext, 0, contextTokenIdxEnd, inDesc, outDesc, inputs, outputs, workspace, stream);
        if (nbSeq - nbContext > 0)
        {
            const int tokens = f.remove_input_padding ? qkvDims.d[0] - contextTokenIdxEnd : qkvDims.d[0] * qkvDims.d[1];
            enqueueGeneration(nbContext, nbSeq - nbContext, tokens, inDesc, outDesc,
                static_cast<const char*>(inputs[getIdx(QKV_TENSOR)]) + (size_t) contextTokenIdxEnd * hidden * esz,
                static_cast<char*>(outputs[0]) + (size_t) contextTokenIdxEnd * outHidden * esz, inputs, outputs, workspace,
                stream);
        }
    }
    // block table of this layer, first sequence seqBeg (:506-519)
    const int64_t* blockPointers(const mi355q_tensor_desc* inDesc, const void* const* inputs, int seqBeg, int& max_blocks) const
    {
        const auto& bpDims = inDesc[getIdx(KV_CACHE_BLOCK_POINTERS)].dims; // [layers, B, 2, max_blocks]
        max_blocks = bpDims.d[bpDims.nbDims - 1];
        const int64_t seqStride = dims_prod(bpDims, 2, bpDims.nbDims);
        const int64_t layerOffset = (int64_t) f.layer_idx * bpDims.d[1] * seqStride;
        return static_cast<const int64_t*>(inputs[getIdx(KV_CACHE_BLOCK_POINTERS)]) + layerOffset + seqBeg * seqStride;
    }
    // the contiguous cache of sequence seqBeg onwards: output 1 = [B, 2, Hkv, S, Dh] (gptAttentionPlugin.cpp:524-529)
    void* linearCache(const mi355q_tensor_desc* outDesc, void* const* outputs, int seqBeg, int& max_seq) const
    {
        const auto& d = outDesc[1].dims;
        MQ_CHECK(d.nbDims == 5 && d.d[1] == 2 && d.d[2] == f.num_kv_heads && d.d[4] == f.head_size,
            "present_key_value must be [batch, 2, num_kv_heads, max_seq_len, head_size]");
        max_seq = d.d[3];
        return static_cast<char*>(outputs[1]) + (size_t) seqBeg * dims_prod(d, 1, d.nbDims) * dtype_size(kvType());
    }
    // the context phase's window: a context longer than cyclic_attention_window_size is masked to the sliding window
    // and appended cyclically (gptAttentionCommon.cpp:867-873, unfusedAttentionKernels_2_template.h:336-340)
    void contextWindow(const mi355q_tensor_desc* inDesc, const void* const* inputs, mi355q_context_attention_params& c) const
    {
        const auto& indir = inDesc[getIdx(CACHE_INDIR)].dims;
        if (indir.d[1] != 1)
            MQ_UNSUPPORTED("GPTAttention context: beam_width %d (the context phase runs with beam width 1)", (int) indir.d[1]);
        c.cyclic_attention_window_size = static_cast<const int32_t*>(inputs[getIdx(HOST_MAX_ATTENTION_WINDOW)])[f.layer_idx];
        c.sink_token_length = static_cast<const int32_t*>(inputs[getIdx(HOST_SINK_TOKEN_LENGTH)])[0];
    }
    void enqueueContext(int seqBeg, int nb, int tokenBeg, int nbTokens, const mi355q_tensor_desc* inDesc,
        const mi355q_tensor_desc* outDesc, const void* const* inputs, void* const* outputs, void* workspace,
        hipStream_t stream) // enqueueSome is_context, :565-600
    {
        MQ_CHECK(seqBeg == 0 && tokenBeg == 0, "context requests come first");
        if (f.mask_type != 1)
            MQ_UNSUPPORTED("GPTAttention context: only the causal mask (AttentionMaskType::CAUSAL = 1) is built, got %d", f.mask_type);
        if (f.position_embedding_type < 0 || f.position_embedding_type > 5)
            MQ_UNSUPPORTED("GPTAttention context: position_embedding_type %d is not built (0 - 5)", f.position_embedding_type);
        const auto& qkvDims = inDesc[getIdx(QKV_TENSOR)].dims;
        int max_len;
        if (f.remove_input_padding)
        {
            const int32_t* hl = static_cast<const int32_t*>(inputs[getIdx(HOST_CONTEXT_LENGTH)]);
            max_len = *std::max_element(hl, hl + nb);
        }
        else
            max_len = qkvDims.d[1];
        if (f.max_context_length > 0)
            MQ_CHECK(max_len <= f.max_context_length, "max_context_q_len %d > max_context_length %d", max_len, f.max_context_length);
        int max_blocks = 0;
        mi355q_context_attention_params c = contextParams(nb, 0, max_len, nbTokens);
        if (f.use_cache)
        {
            contextWindow(inDesc, inputs, c);
            if (f.paged_kv_cache)
            {
                c.block_pointers = blockPointers(inDesc, inputs, seqBeg, max_blocks);
                c.max_blocks_per_seq = max_blocks;
            }
            else
                c.kv_linear_buffer = linearCache(outDesc, outputs, seqBeg, c.linear_max_seq_len);
            if (f.use_paged_context_fmha) // chunked context (:825-864): the earlier chunks are read back from the cache
            {
                MQ_CHECK(f.paged_kv_cache, "use_paged_context_fmha needs paged_kv_cache");
                const int32_t* kvl = static_cast<const int32_t*>(inputs[getIdx(HOST_PAST_KEY_VALUE_LENGTHS)]) + seqBeg;
                c.sequence_lengths = static_cast<const int32_t*>(inputs[getIdx(SEQUENCE_LENGTH)]) + seqBeg;
                c.max_kv_length = *std::max_element(kvl, kvl + nb); // max_context_kv_len, gptAttentionPlugin.cpp:558-563
            }
            if (hasKvQuant())
                c.kv_scale_orig_quant = static_cast<const float*>(inputs[getIdx(KV_CACHE_QUANTIZATION_SCALE)]);
        }
        c.qkv = inputs[getIdx(QKV_TENSOR)];
        c.qkv_bias = f.qkv_bias_enabled ? inputs[getIdx(QKV_BIAS_TENSOR)] : nullptr;
        if (f.position_embedding_type == 3 || f.position_embedding_type == 4)
            c.alibi_slopes = inputs[getIdx(ALIBI_SLOPES)];
        if (f.position_embedding_type == 5) // isRelativePosition(), gptAttentionPlugin.cpp:590-596; gptAttentionCommon.cpp:1006-1016
        {
            c.relative_attention_bias = inputs[getIdx(RELATIVE_ATTENTION_BIAS)];
            c.relative_attention_bias_stride = inDesc[getIdx(RELATIVE_ATTENTION_BIAS)].dims.d[1]; // max_seq_len or num_buckets
            c.max_distance = f.max_distance;
        }
        c.out = outputs[0];
        c.context_lengths = static_cast<const int32_t*>(inputs[getIdx(CONTEXT_LENGTHS)]) + seqBeg;
        c.workspace = workspace;
        c.workspace_bytes = workspace ? mi355q_context_attention_workspace_size(&c) : 0;
        context_attention(c, stream);
    }
    void enqueueGeneration(int seqBeg, int nb, int tokens, const mi355q_tensor_desc* inDesc,
        const mi355q_tensor_desc* outDesc, const void* qkv, void* out, const void* const* inputs, void* const* outputs,
        void* workspace, hipStream_t stream) // enqueueSome !is_context, :606-656
    {
        MQ_CHECK(f.use_cache, "KV-cache-less is only supported for context");
        MQ_CHECK(tokens == nb, "Only Medusa mode supports input length > 1 in the generation phase");
        const int32_t* past = static_cast<const int32_t*>(inputs[getIdx(HOST_PAST_KEY_VALUE_LENGTHS)]) + seqBeg;
        const int max_past = *std::max_element(past, past + nb);
        int max_blocks = 0;
        const int64_t* bp = f.paged_kv_cache ? blockPointers(inDesc, inputs, seqBeg, max_blocks) : nullptr;
        mi355q_decode_attention_params p = baseParams(nb, max_blocks, max_past + 1);
        if (!f.paged_kv_cache)
        {
            p.kv_linear_buffer = linearCache(outDesc, outputs, seqBeg, p.linear_max_seq_len);
            p.max_blocks_per_seq = 1;
        }
        // window, sink tokens, beams (:476-489, :612-616): nb counts batch x beam rows
        const auto& indir = inDesc[getIdx(CACHE_INDIR)].dims; // [num_requests, beam_width, max_attention_window_size]
        p.cyclic_attention_window_size = static_cast<const int32_t*>(inputs[getIdx(HOST_MAX_ATTENTION_WINDOW)])[f.layer_idx];
        p.sink_token_length = static_cast<const int32_t*>(inputs[getIdx(HOST_SINK_TOKEN_LENGTH)])[0];
        if (indir.d[1] > 1)
        {
            MQ_CHECK(nb % indir.d[1] == 0 && seqBeg % indir.d[1] == 0, "batch_beam %d (from %d) against beam_width %d", nb,
                seqBeg, (int) indir.d[1]);
            p.beam_width = indir.d[1];
            p.max_attention_window_size = indir.d[2];
            p.cache_indirection = static_cast<const int32_t*>(inputs[getIdx(CACHE_INDIR)]) + (size_t) seqBeg * indir.d[2];
            p.input_lengths = static_cast<const int32_t*>(inputs[getIdx(CONTEXT_LENGTHS)]) + seqBeg;
        }
        p.qkv = qkv;
        p.qkv_bias = f.qkv_bias_enabled ? inputs[getIdx(QKV_BIAS_TENSOR)] : nullptr;
        p.out = out;
        p.sequence_lengths = static_cast<const int32_t*>(inputs[getIdx(SEQUENCE_LENGTH)]) + seqBeg;
        if (f.position_embedding_type == 3 || f.position_embedding_type == 4)
            p.alibi_slopes = inputs[getIdx(ALIBI_SLOPES)];
        if (f.position_embedding_type == 5) // isRelativePosition(), gptAttentionPlugin.cpp:634-640
        {
            p.relative_attention_bias = inputs[getIdx(RELATIVE_ATTENTION_BIAS)];
            p.relative_attention_bias_stride = inDesc[getIdx(RELATIVE_ATTENTION_BIAS)].dims.d[1]; // max_seq_len or num_buckets
            p.max_distance = f.max_distance;
        }
        p.block_pointers = bp;
        if (hasKvQuant())
        {
            p.kv_scale_orig_quant = static_cast<const float*>(inputs[getIdx(KV_CACHE_QUANTIZATION_SCALE)]);
            p.kv_scale_quant_orig = static_cast<const float*>(inputs[getIdx(KV_CACHE_DEQUANTIZATION_SCALE)]);
        }
        p.workspace = workspace;
        p.workspace_bytes = workspace ? mi355q_decode_attention_workspace_size(&p) : 0;
        masked_multihead_attention(p, stream);
    }
    size_t serializationSize() const override
    {
        return sizeof(GptAttentionFields);
    }
    void serialize(char* buf) const override
    {
        write(buf, f);
    }
};

GptAttentionFields parse_attention_fields(const FieldReader& r) // gptAttentionPlugin.cpp:700-790
{
    GptAttentionFields f;
    f.layer_idx = r.opt_int("layer_idx", 0);
    f.num_heads = r.req_int("num_heads");
    f.num_kv_heads = r.opt_int("num_kv_heads", 0);
    f.head_size = r.req_int("head_size");
    f.unidirectional = r.opt_int("unidirectional", 1);
    f.q_scaling = r.opt_float("q_scaling", 1.f);
    f.position_embedding_type = r.opt_int("position_embedding_type", 0);
    f.rotary_embedding_dim = r.opt_int("rotary_embedding_dim", 0);
    f.rotary_embedding_base = r.opt_float("rotary_embedding_base", 10000.f);
    f.rotary_embedding_scale_type = r.opt_int("rotary_embedding_scale_type", 0);
    f.rotary_embedding_scale = r.opt_float("rotary_embedding_scale", 1.f);
    f.rotary_embedding_max_positions = r.opt_int("rotary_embedding_max_positions", 0);
    f.tp_size = r.opt_int("tp_size", 1);
    f.tp_rank = r.opt_int("tp_rank", 0);
    f.unfuse_qkv_gemm = r.opt_int("unfuse_qkv_gemm", 0);
    f.context_fmha_type = r.opt_int("context_fmha_type", 0);
    f.multi_block_mode = r.opt_int("multi_block_mode", 0);
    f.enable_xqa = r.opt_int("enable_xqa", 0);
    f.kv_cache_quant_mode = r.opt_int("kv_cache_quant_mode", 0);
    f.remove_input_padding = r.opt_int("remove_input_padding", 0);
    f.mask_type = r.opt_int("mask_type", 1);
    f.paged_kv_cache = r.opt_int("paged_kv_cache", 0);
    f.tokens_per_block = r.opt_int("tokens_per_block", 0);
    f.type = r.req_int("type_id");
    f.max_context_length = r.opt_int("max_context_length", 0);
    f.qkv_bias_enabled = r.opt_int("qkv_bias_enabled", 0);
    f.cross_attention = r.opt_int("do_cross_attention", 0);
    f.max_distance = r.opt_int("max_distance", 0);
    f.pos_shift_enabled = r.opt_int("pos_shift_enabled", 0);
    f.dense_context_fmha = r.opt_int("dense_context_fmha", 0);
    f.use_paged_context_fmha = r.opt_int("use_paged_context_fmha", 0);
    f.use_cache = r.opt_int("use_cache", 1);
    f.is_medusa_enabled = r.opt_int("is_medusa_enabled", 0);
    return f;
}

const char* const kPluginNames[] = {"WeightOnlyGroupwiseQuantMatmul", "WeightOnlyQuantMatmul", "SmoothQuantGemm",
    "QuantizePerToken", "GPTAttention", "Fp8Gemm", "QuantizeTensor", "RmsnormQuantization", "LayernormQuantization",
    "AllReduce", "AllGather", "MixtureOfExperts", nullptr};

PluginBase* create_plugin(const char* name, const FieldReader& r)
{
    const std::string n(name);
    if (n == "WeightOnlyGroupwiseQuantMatmul")
        return new WeightOnlyGroupwiseQuantMatmulPlugin(r.req_int("type_id"), r.req_int("quant_algo"), r.req_int("group_size"));
    if (n == "WeightOnlyQuantMatmul")
        return new WeightOnlyQuantMatmulPlugin(r.req_int("type_id"), r.req_int("weight_type_id"));
    if (n == "SmoothQuantGemm")
    {
        // createPlugin, smoothQuantGemmPlugin.cpp:335-370: QuantMode from the two booleans
        const int mode = (r.req_int("has_per_channel_scaling") ? (1 << 3) : 0)
            | (r.req_int("has_per_token_scaling") ? (1 << 4) : 0);
        return new SmoothQuantGemmPlugin(mode, r.req_int("type_id"));
    }
    if (n == "QuantizePerToken")
        return new QuantizePerTokenPlugin();
    if (n == "GPTAttention")
        return new GPTAttentionPlugin(parse_attention_fields(r));
    if (n == "Fp8Gemm")
        return new Fp8GemmPlugin(r.req_int("type_id"));
    if (n == "QuantizeTensor")
        return new QuantizeTensorPlugin();
    if (n == "MixtureOfExperts") // createPlugin, mixtureOfExpertsPlugin.cpp:497-556 (every field kINT32)
        return new MixtureOfExpertsPlugin(r.req_int("number_of_experts"), r.req_int("top_k"), r.req_int("expert_hidden_size"),
            r.req_int("expert_inter_size"), r.opt_int("activation_type", 5), r.req_int("type_id"), r.req_int("weight_type_id"),
            r.opt_int("quant_mode", 0), r.opt_int("use_finished", 0) != 0, r.opt_int("use_bias", 0) != 0, r.opt_int("tp_size", 1),
            r.opt_int("tp_rank", 0), r.opt_int("parallelism_mode", 0), r.opt_int("normalization_mode", 0));
    if (n == "AllGather") // createPlugin, allgatherPlugin.cpp:190-230
    {
        const mi355q_plugin_field* g = r.find("group");
        MQ_CHECK(g && g->type == MI355Q_FIELD_INT32 && g->length > 0, "AllGather: field 'group' (int32 list) is missing");
        const int32_t* gi = static_cast<const int32_t*>(g->data);
        return new AllGatherPlugin(std::vector<int32_t>(gi, gi + g->length), r.req_int("type_id"), r.opt_int("strategy", MI355Q_AR_RING),
            r.opt_int("rank", -1));
    }
    if (n == "AllReduce") // createPlugin, allreducePlugin.cpp:306-350
    {
        const mi355q_plugin_field* g = r.find("group");
        MQ_CHECK(g && g->type == MI355Q_FIELD_INT32 && g->length > 0, "AllReduce: field 'group' (int32 list) is missing");
        const int32_t* gi = static_cast<const int32_t*>(g->data);
        return new AllReducePlugin(std::vector<int32_t>(gi, gi + g->length), r.req_int("type_id"),
            r.opt_int("strategy", MI355Q_AR_AUTO), r.opt_int("counter", 0), r.opt_int("rank", -1));
    }
    if (n == "RmsnormQuantization") // createPlugin, rmsnormQuantizationPlugin.cpp:253-290
        return new NormQuantizationPlugin(
            false, r.opt_float("eps", 1e-5f), false, r.req_int("dyn_act_scaling") != 0, r.req_int("type_id"));
    if (n == "LayernormQuantization") // layernormQuantizationPlugin.cpp:260-302
        return new NormQuantizationPlugin(true, r.opt_float("eps", 1e-5f), r.opt_int("use_diff_of_squares", 1) != 0,
            r.req_int("dyn_act_scaling") != 0, r.req_int("type_id"));
    MQ_CHECK(false, "unknown plugin '%s'", name);
    return nullptr;
}

PluginBase* deserialize_plugin(const char* name, const void* data, size_t length)
{
    const std::string n(name);
    const char* d = static_cast<const char*>(data);
    const char* a = d;
    PluginBase* out = nullptr;
    auto need = [&](size_t bytes)
    {
        MQ_CHECK(length == bytes,
            "Expected length (%d) != real length (%d). This is often caused by using different TensorRT-LLM version to "
            "build engine and run engine.",
            (int) length, (int) bytes);
    };
    if (n == "WeightOnlyGroupwiseQuantMatmul")
    {
        need(sizeof(int32_t) * 3 + sizeof(GemmDims));
        int32_t t, q, g;
        read(d, t);
        read(d, q);
        read(d, g);
        auto* p = new WeightOnlyGroupwiseQuantMatmulPlugin(t, q, g);
        read(d, p->mDims);
        out = p;
    }
    else if (n == "WeightOnlyQuantMatmul")
    {
        need(sizeof(int32_t) * 2 + sizeof(GemmDims));
        int32_t t, w;
        read(d, t);
        read(d, w);
        auto* p = new WeightOnlyQuantMatmulPlugin(t, w);
        read(d, p->mDims);
        out = p;
    }
    else if (n == "SmoothQuantGemm")
    {
        need(sizeof(int32_t) * 2 + sizeof(GemmDims));
        int32_t q, t;
        read(d, q);
        read(d, t);
        auto* p = new SmoothQuantGemmPlugin(q, t);
        read(d, p->mDims);
        out = p;
    }
    else if (n == "QuantizePerToken")
    {
        need(0);
        out = new QuantizePerTokenPlugin();
    }
    else if (n == "GPTAttention")
    {
        need(sizeof(GptAttentionFields));
        GptAttentionFields f;
        read(d, f);
        out = new GPTAttentionPlugin(f);
    }
    else if (n == "Fp8Gemm")
    {
        need(sizeof(int32_t));
        int32_t t;
        read(d, t);
        out = new Fp8GemmPlugin(t);
    }
    else if (n == "QuantizeTensor")
    {
        need(0);
        out = new QuantizeTensorPlugin();
    }
    else if (n == "MixtureOfExperts") // mixtureOfExpertsPlugin.cpp:96-128
    {
        int32_t v[8], t[4];
        bool fin, bias;
        for (auto& x : v)
            read(d, x);
        read(d, fin);
        read(d, bias);
        for (auto& x : t)
            read(d, x);
        out = new MixtureOfExpertsPlugin(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], fin, bias, t[0], t[1], t[2], t[3]);
    }
    else if (n == "AllGather") // allgatherPlugin.cpp:36-54: type, then the group to the end
    {
        MQ_CHECK(length >= 4 && (length - 4) % 4 == 0, "Expected length (%d) != real length", (int) length);
        int32_t t;
        read(d, t);
        std::vector<int32_t> group((length - 4) / 4);
        for (auto& g : group)
            read(d, g);
        out = new AllGatherPlugin(group, t, MI355Q_AR_RING, -1);
    }
    else if (n == "AllReduce") // allreducePlugin.cpp:42-62: type, strategy, counter, then the group to the end
    {
        MQ_CHECK(length >= 9 && (length - 9) % 4 == 0, "Expected length (%d) != real length", (int) length);
        int32_t t, c;
        int8_t st;
        read(d, t);
        read(d, st);
        read(d, c);
        std::vector<int32_t> group((length - 9) / 4);
        for (auto& g : group)
            read(d, g);
        out = new AllReducePlugin(group, t, st, c, -1);
    }
    else if (n == "RmsnormQuantization" || n == "LayernormQuantization")
    {
        const bool layer = n == "LayernormQuantization";
        need(sizeof(float) + (layer ? 1 : 0) + 1 + sizeof(int32_t));
        float eps;
        bool diff = false, dyn;
        int32_t t;
        read(d, eps);
        if (layer)
            read(d, diff);
        read(d, dyn);
        read(d, t);
        out = new NormQuantizationPlugin(layer, eps, diff, dyn, t);
    }
    else
        MQ_CHECK(false, "unknown plugin '%s'", name);
    (void) a;
    return out;
}

} // namespace

// shared by the plugins and the C entry point: pre-quant smoothing into the workspace, then GEMM
void weight_only_gemm_entry(const mi355q_weight_only_params& p, hipStream_t stream)
{
    if (p.m == 0)
        return;
    mi355q_weight_only_params q = p;
    char* ws = static_cast<char*>(p.workspace);
    size_t ws_bytes = p.workspace_bytes;
    if (p.act_fp8)
    {
        // W4A8-AWQ (:382-418 with T_out = __nv_fp8_e4m3): smoothing and the e4m3 rounding in one pass, always through
        // the workspace (the decode kernels' fused smoothing cannot round through e4m3)
        const size_t act_bytes = align_up((size_t) p.m * p.k * 2, 256);
        MQ_CHECK(ws && ws_bytes >= act_bytes, "workspace too small for the fp8-rounded activations (%zu < %zu)", ws_bytes,
            act_bytes);
        apply_per_channel_scale_fp8(ws, p.in, p.act_scale, p.m, p.k, p.act_type, stream);
        q.in = ws;
        q.act_scale = nullptr;
        ws += act_bytes;
        ws_bytes -= act_bytes;
    }
    else if (p.act_scale && !woq_gemm_fuses_act_scale(p.m, p.n, p.k, p.weight_type, true))
    {
        // weightOnlyGroupwiseQuantMatmulPlugin.cpp:382-418: smoothed activations go to the workspace (the decode
        // kernels, m <= 32, apply the scale while they stage the rows instead: one launch less)
        const size_t act_bytes = align_up((size_t) p.m * p.k * 2, 256);
        MQ_CHECK(ws && ws_bytes >= act_bytes, "workspace too small for the smoothed activations (%zu < %zu)", ws_bytes,
            act_bytes);
        apply_per_channel_scale(ws, p.in, p.act_scale, p.m, p.k, p.act_type, stream);
        q.in = ws;
        q.act_scale = nullptr;
        ws += act_bytes;
        ws_bytes -= act_bytes;
    }
    weight_only_gemm(q, ws, ws ? ws_bytes : 0, stream);
}

} // namespace mi355q

// ============================================================================== C entry points
using namespace mi355q;

struct mi355q_plugin
{
    std::unique_ptr<PluginBase> impl;
};

extern "C"
{

    const char* mi355q_last_error(void)
    {
        return g_last_error.c_str();
    }

    const char* mi355q_version(void)
    {
        return "mi355q 0.1 (gfx950)";
    }

    mi355q_status mi355q_plugin_create(
        const char* name, const mi355q_plugin_field* fields, int32_t nbFields, mi355q_plugin** out)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(name && out, "null argument");
                FieldReader r{fields, fields ? nbFields : 0};
                std::unique_ptr<PluginBase> p(create_plugin(name, r));
                *out = new mi355q_plugin{std::move(p)};
            });
    }

    mi355q_status mi355q_plugin_deserialize(const char* name, const void* data, size_t length, mi355q_plugin** out)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(name && out && (data || length == 0), "null argument");
                std::unique_ptr<PluginBase> p(deserialize_plugin(name, data, length));
                *out = new mi355q_plugin{std::move(p)};
            });
    }

    void mi355q_plugin_destroy(mi355q_plugin* p)
    {
        delete p;
    }

    mi355q_status mi355q_plugin_clone(const mi355q_plugin* p, mi355q_plugin** out)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(p && out, "null argument");
                *out = new mi355q_plugin{std::unique_ptr<PluginBase>(p->impl->clone())};
            });
    }

    const char* mi355q_plugin_type(const mi355q_plugin* p)
    {
        return p ? p->impl->type() : "";
    }

    int32_t mi355q_plugin_nb_outputs(const mi355q_plugin* p)
    {
        return p ? p->impl->nbOutputs() : 0;
    }

    mi355q_status mi355q_plugin_get_output_dimensions(
        const mi355q_plugin* p, int32_t outputIndex, const mi355q_dims* inputs, int32_t nbInputs, mi355q_dims* out)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(p && inputs && out, "null argument");
                p->impl->getOutputDimensions(outputIndex, inputs, nbInputs, out);
            });
    }

    int32_t mi355q_plugin_supports_format_combination(
        const mi355q_plugin* p, int32_t pos, const mi355q_tensor_desc* inOut, int32_t nbInputs, int32_t nbOutputs)
    {
        int32_t res = 0;
        mi355q_status s = guarded(
            [&]
            {
                MQ_CHECK(p && inOut && pos >= 0 && pos < nbInputs + nbOutputs, "bad argument");
                res = p->impl->supportsFormatCombination(pos, inOut, nbInputs, nbOutputs) ? 1 : 0;
            });
        return s == MI355Q_OK ? res : (int32_t) s;
    }

    int32_t mi355q_plugin_get_output_data_type(
        const mi355q_plugin* p, int32_t index, const int32_t* inputTypes, int32_t nbInputs)
    {
        int32_t res = 0;
        mi355q_status s = guarded(
            [&]
            {
                MQ_CHECK(p, "null argument");
                res = p->impl->getOutputDataType(index, inputTypes, nbInputs);
            });
        return s == MI355Q_OK ? res : (int32_t) s;
    }

    mi355q_status mi355q_plugin_configure(mi355q_plugin* p, const mi355q_dynamic_tensor_desc* in, int32_t nbInputs,
        const mi355q_dynamic_tensor_desc* out, int32_t nbOutputs)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(p && in, "null argument");
                p->impl->configure(in, nbInputs, out, nbOutputs);
                // resolve this GPU's arrival-word array now, not inside the first enqueue (which may run under stream
                // capture); a build machine without a GPU configures plugins too
                int devices = 0;
                if (hipGetDeviceCount(&devices) == hipSuccess && devices > 0)
                    arrival_words_prepare();
                else
                    (void) hipGetLastError();
            });
    }

    size_t mi355q_plugin_get_workspace_size(const mi355q_plugin* p, const mi355q_tensor_desc* inputs, int32_t nbInputs,
        const mi355q_tensor_desc* outputs, int32_t nbOutputs)
    {
        size_t res = 0;
        guarded(
            [&]
            {
                MQ_CHECK(p, "null argument");
                res = p->impl->workspaceSize(inputs, nbInputs, outputs, nbOutputs);
            });
        return res;
    }

    int32_t mi355q_plugin_enqueue(mi355q_plugin* p, const mi355q_tensor_desc* inputDesc,
        const mi355q_tensor_desc* outputDesc, const void* const* inputs, void* const* outputs, void* workspace,
        void* stream)
    {
        return (int32_t) guarded(
            [&]
            {
                MQ_CHECK(p && inputDesc && inputs && outputs, "null argument");
                p->impl->enqueue(inputDesc, outputDesc, inputs, outputs, workspace, static_cast<hipStream_t>(stream));
            });
    }

    size_t mi355q_plugin_get_serialization_size(const mi355q_plugin* p)
    {
        return p ? p->impl->serializationSize() : 0;
    }

    mi355q_status mi355q_plugin_serialize(const mi355q_plugin* p, void* buffer)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(p && (buffer || p->impl->serializationSize() == 0), "null argument");
                p->impl->serialize(static_cast<char*>(buffer));
            });
    }

    const char* const* mi355q_get_plugin_names(int32_t* nbPlugins)
    {
        if (nbPlugins)
            *nbPlugins = (int32_t) (sizeof(kPluginNames) / sizeof(kPluginNames[0])) - 1; // the nullptr terminator
        return kPluginNames;
    }

    mi355q_status mi355q_preprocess_weights_for_mixed_gemm(
        int8_t* dst, const int8_t* src, const size_t shape[3], int32_t weight_type, int32_t layout)
    {
        return guarded([&] { preprocess_weights(dst, src, shape, weight_type, layout); });
    }

    mi355q_status mi355q_unprocess_weights_for_mixed_gemm(
        int8_t* dst, const int8_t* src, const size_t shape[3], int32_t weight_type, int32_t layout)
    {
        return guarded([&] { unprocess_weights(dst, src, shape, weight_type, layout); });
    }

    mi355q_status mi355q_unpack_int4_packed_tensor_to_int8(int8_t* dst, const int8_t* src, size_t numPacked)
    {
        return guarded([&] { unpack_int4(dst, src, numPacked); });
    }

    mi355q_status mi355q_pack_int8_tensor_to_packed_int4(int8_t* dst, const int8_t* src, size_t numPacked)
    {
        return guarded([&] { pack_int4(dst, src, numPacked); });
    }

    mi355q_status mi355q_symmetric_quantize(int8_t* processed, int8_t* unprocessed, float* scales, const float* weight,
        const size_t shape[3], int32_t weight_type, int32_t layout)
    {
        return guarded([&] { symmetric_quantize(processed, unprocessed, scales, weight, shape, weight_type, layout); });
    }

    size_t mi355q_weight_only_gemm_workspace_size(int32_t max_m, int32_t n, int32_t k, int32_t weight_type)
    {
        if (max_m <= 0 || n <= 0 || k <= 0)
            return 0;
        return align_up((size_t) max_m * k * 2, 256) + woq_gemm_workspace_size(max_m, n, k, weight_type);
    }

    mi355q_status mi355q_weight_only_gemm(const mi355q_weight_only_params* params, void* stream)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                weight_only_gemm_entry(*params, static_cast<hipStream_t>(stream));
            });
    }

    mi355q_status mi355q_moe_gemm_bias_act(const mi355q_moe_gemm_params* params, void* stream)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                moe_weight_only_gemm(*params, static_cast<hipStream_t>(stream));
            });
    }

    size_t mi355q_moe_workspace_size(const mi355q_moe_params* params)
    {
        size_t r = 0;
        guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                r = moe_workspace_size(*params);
            });
        return r;
    }

    mi355q_status mi355q_run_moe(const mi355q_moe_params* params, void* stream)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                run_moe(*params, static_cast<hipStream_t>(stream));
            });
    }

    mi355q_status mi355q_weight_only_dequantize(const mi355q_weight_only_params* params, void* stream)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                weight_only_dequantize(*params, static_cast<hipStream_t>(stream));
            });
    }

    mi355q_status mi355q_apply_per_channel_scale(
        void* out, const void* act, const void* scale, int32_t rows, int32_t cols, int32_t act_type, void* stream)
    {
        return guarded(
            [&] { apply_per_channel_scale(out, act, scale, rows, cols, act_type, static_cast<hipStream_t>(stream)); });
    }

    mi355q_status mi355q_int8_gemm(const int8_t* A, const int8_t* B, int32_t per_token, int32_t per_channel,
        const float* alphaCol, const float* alphaRow, void* C, int32_t out_type, int32_t m, int32_t n, int32_t k,
        void* stream)
    {
        return guarded(
            [&]
            {
                int8_gemm(A, B, per_token, per_channel, alphaCol, alphaRow, C, out_type, m, n, k, nullptr, 0,
                    static_cast<hipStream_t>(stream));
            });
    }

    size_t mi355q_gemm8_workspace_size(int32_t m, int32_t n, int32_t k)
    {
        return gemm8_workspace_size(m, n, k);
    }

    mi355q_status mi355q_int8_gemm_ws(const int8_t* A, const int8_t* B, int32_t per_token, int32_t per_channel,
        const float* alphaCol, const float* alphaRow, void* C, int32_t out_type, int32_t m, int32_t n, int32_t k,
        void* workspace, size_t workspaceBytes, void* stream)
    {
        return guarded(
            [&]
            {
                int8_gemm(A, B, per_token, per_channel, alphaCol, alphaRow, C, out_type, m, n, k, workspace,
                    workspaceBytes, static_cast<hipStream_t>(stream));
            });
    }

    mi355q_status mi355q_fp8_gemm_ws(const void* A, const void* B, const float* scaleA, const float* scaleB, void* C,
        int32_t out_type, int32_t m, int32_t n, int32_t k, void* workspace, size_t workspaceBytes, void* stream)
    {
        return guarded(
            [&]
            {
                fp8_gemm(A, B, scaleA, scaleB, C, out_type, m, n, k, workspace, workspaceBytes,
                    static_cast<hipStream_t>(stream));
            });
    }

    mi355q_status mi355q_per_token_quantization(int8_t* dst, const void* src, int64_t numRows, int64_t numCols,
        float* scalePtr, int32_t src_type, void* stream)
    {
        return guarded(
            [&]
            { per_token_quantization(dst, src, numRows, numCols, scalePtr, src_type, static_cast<hipStream_t>(stream)); });
    }

    mi355q_status mi355q_fp8_gemm(const void* A, const void* B, const float* scaleA, const float* scaleB, void* C,
        int32_t out_type, int32_t m, int32_t n, int32_t k, void* stream)
    {
        return guarded([&] { fp8_gemm(A, B, scaleA, scaleB, C, out_type, m, n, k, nullptr, 0, static_cast<hipStream_t>(stream)); });
    }

    mi355q_status mi355q_quantize_e4m3(void* dst, float* scales, const void* src, int64_t rows, int64_t cols,
        int32_t src_type, int32_t mode, void* stream)
    {
        return guarded(
            [&] { quantize_e4m3(dst, scales, src, rows, cols, src_type, mode, static_cast<hipStream_t>(stream)); });
    }

    mi355q_status mi355q_quantize_tensor(
        int8_t* dst, const void* src, int64_t size, const float* scalePtr, int32_t src_type, void* stream)
    {
        return guarded([&] { quantize_tensor(dst, src, size, scalePtr, src_type, static_cast<hipStream_t>(stream)); });
    }

    mi355q_status mi355q_gated_silu(const mi355q_gated_act_params* params, void* stream)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(params, "null params");
                gated_silu(*params, static_cast<hipStream_t>(stream));
            });
    }

    mi355q_status mi355q_rms_norm(void* out, const void* input, const void* gamma, const void* beta, float eps,
        int32_t tokens, int32_t hidden_dim, const float* scale, float* dynamic_scale, int8_t* normed_output_quant,
        int32_t type, void* stream)
    {
        return guarded(
            [&]
            {
                rms_norm(out, input, gamma, beta, eps, tokens, hidden_dim, scale, dynamic_scale, normed_output_quant, type,
                    static_cast<hipStream_t>(stream));
            });
    }

    mi355q_status mi355q_layer_norm(void* out, const void* input, const void* gamma, const void* beta, float eps,
        int32_t tokens, int32_t hidden_dim, int32_t use_diff_of_squares, const float* scale, float* dynamic_scale,
        int8_t* normed_output_quant, int32_t type, void* stream)
    {
        return guarded(
            [&]
            {
                layer_norm(out, input, gamma, beta, eps, tokens, hidden_dim, use_diff_of_squares != 0, scale,
                    dynamic_scale, normed_output_quant, type, static_cast<hipStream_t>(stream));
            });
    }

    size_t mi355q_decode_attention_workspace_size(const mi355q_decode_attention_params* params)
    {
        size_t res = 0;
        guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                res = decode_attention_workspace_size(*params);
            });
        return res;
    }

    mi355q_status mi355q_masked_multihead_attention(const mi355q_decode_attention_params* params, void* stream)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                masked_multihead_attention(*params, static_cast<hipStream_t>(stream));
            });
    }

    size_t mi355q_decode_attention_static_scratch_size(void)
    {
        return decode_attention_static_scratch_bytes();
    }

    mi355q_status mi355q_masked_multihead_attention_static(const mi355q_decode_attention_params* params, void* stream)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                masked_multihead_attention_static(*params, static_cast<hipStream_t>(stream));
            });
    }

    size_t mi355q_context_attention_workspace_size(const mi355q_context_attention_params* params)
    {
        size_t res = 0;
        guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                res = context_attention_workspace_size(*params);
            });
        return res;
    }

    mi355q_status mi355q_context_attention(const mi355q_context_attention_params* params, void* stream)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                context_attention(*params, static_cast<hipStream_t>(stream));
            });
    }

    size_t mi355q_allreduce_max_workspace_size(int32_t worldSize)
    {
        return worldSize <= 2 ? 16 * 1000 * 1000 : 8 * 1000 * 1000;
    }

    mi355q_status mi355q_allreduce_params_deserialize(mi355q_allreduce_params* out, const void* const* buffer_ptrs,
        size_t tpSize, size_t tpRank, uint32_t flag_value)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(out, "null argument");
                allreduce_params_deserialize(*out, buffer_ptrs, tpSize, tpRank, flag_value);
            });
    }

    mi355q_status mi355q_custom_all_gather(
        mi355q_allreduce_params* params, const void* input, void* output, size_t bytes_per_rank, void* stream)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                custom_all_gather(*params, input, output, bytes_per_rank, static_cast<hipStream_t>(stream));
            });
    }

    mi355q_status mi355q_custom_all_reduce(mi355q_allreduce_params* params, void* data, size_t elts, int32_t dtype,
        int32_t strategy, void* stream)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                custom_all_reduce(*params, data, elts, dtype, strategy, static_cast<hipStream_t>(stream));
            });
    }

    size_t mi355q_allreduce_barrier_bytes(void)
    {
        return allreduce_barrier_bytes();
    }

    mi355q_status mi355q_allreduce_alloc_timeout_flag(uint32_t** flag)
    {
        return guarded([&] { allreduce_alloc_timeout_flag(flag); });
    }

    mi355q_status mi355q_allreduce_free_timeout_flag(uint32_t* flag)
    {
        return guarded([&] { allreduce_free_timeout_flag(flag); });
    }

    mi355q_status mi355q_multi_gpu_barrier(const mi355q_allreduce_params* params, void* stream)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(params, "null argument");
                multi_gpu_barrier_launch(*params, static_cast<hipStream_t>(stream));
            });
    }

    mi355q_status mi355q_ipc_alloc(size_t bytes, void** ptr)
    {
        return guarded([&] { ipc_alloc(bytes, ptr); });
    }
    mi355q_status mi355q_ipc_free(void* ptr)
    {
        return guarded([&] { ipc_free(ptr); });
    }
    mi355q_status mi355q_ipc_get_handle(void* ptr, uint8_t handle[MI355Q_IPC_HANDLE_BYTES])
    {
        return guarded(
            [&]
            {
                MQ_CHECK(ptr && handle, "null argument");
                ipc_get_handle(ptr, handle);
            });
    }
    mi355q_status mi355q_ipc_open_handle(const uint8_t handle[MI355Q_IPC_HANDLE_BYTES], void** ptr)
    {
        return guarded(
            [&]
            {
                MQ_CHECK(ptr && handle, "null argument");
                ipc_open_handle(handle, ptr);
            });
    }
    mi355q_status mi355q_ipc_close_handle(void* ptr)
    {
        return guarded([&] { ipc_close_handle(ptr); });
    }

    mi355q_status mi355q_comm_get_unique_id(uint8_t id[MI355Q_COMM_UNIQUE_ID_BYTES])
    {
        return guarded([&] { comm_get_unique_id(id); });
    }
    mi355q_status mi355q_comm_init_rank(
        const int32_t* group, int32_t group_size, const uint8_t id[MI355Q_COMM_UNIQUE_ID_BYTES], int32_t group_rank)
    {
        return guarded([&] { comm_init_rank(group, group_size, id, group_rank); });
    }
    int32_t mi355q_comm_exists(const int32_t* group, int32_t group_size)
    {
        int32_t res = 0;
        const mi355q_status st = guarded([&] { res = comm_exists(group, group_size) ? 1 : 0; });
        return st == MI355Q_OK ? res : (int32_t) st;
    }
    mi355q_status mi355q_comm_destroy(const int32_t* group, int32_t group_size)
    {
        return guarded([&] { comm_destroy(group, group_size); });
    }

    // AllreducePlugin::selectImplementation, plugins/ncclPlugin/allreducePlugin.cpp:111-139
    int32_t mi355q_allreduce_select_implementation(size_t messageBytes, int32_t worldSize, size_t maxWorkspaceBytes)
    {
        // the reference keeps the custom kernels up to the workspace size; the pushed two-shot here needs a scatter and
        // a gather area (2 x message) and one-shot a slot per rank, so the crossover to RING (RCCL) comes earlier
        if (2 * messageBytes > maxWorkspaceBytes)
            return MI355Q_AR_RING;
        if ((size_t) worldSize * ((messageBytes + 255) / 256 * 256) > maxWorkspaceBytes)
            return MI355Q_AR_TWOSHOT;
        if (worldSize <= 2)
            return MI355Q_AR_ONESHOT;
        if (worldSize <= 4)
            return messageBytes < 1 * 1000 * 1000 ? MI355Q_AR_ONESHOT : MI355Q_AR_TWOSHOT;
        return messageBytes < 500 * 1000 ? MI355Q_AR_ONESHOT : MI355Q_AR_TWOSHOT;
    }
}
