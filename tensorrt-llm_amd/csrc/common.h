// common.h -- shared host/device helpers for libmi355q (gfx950 only).
#pragma once
#include "../../include/mi355q.h"
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace mi355q
{

// Error plumbing: the reference throws TllmException from TLLM_CHECK and catches it at the plugin
// boundary (plugins/common/plugin.cpp caughtError).  Same idea, C-ABI friendly.
struct Error : std::runtime_error
{
    mi355q_status status;
    Error(mi355q_status s, const std::string& m)
        : std::runtime_error(m)
        , status(s)
    {
    }
};

std::string fmtstr(const char* fmt, ...);
void set_last_error(const std::string& msg);

#define MQ_CHECK(cond, ...)                                                                                            \
    do                                                                                                                 \
    {                                                                                                                  \
        if (!(cond))                                                                                                   \
            throw ::mi355q::Error(MI355Q_ERR_INVALID_ARGUMENT,                                                         \
                ::mi355q::fmtstr("%s:%d check failed: %s : ", __FILE__, __LINE__, #cond) + ::mi355q::fmtstr(__VA_ARGS__)); \
    } while (0)

#define MQ_UNSUPPORTED(...) throw ::mi355q::Error(MI355Q_ERR_UNSUPPORTED, ::mi355q::fmtstr(__VA_ARGS__))

#define MQ_HIP(expr)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            throw ::mi355q::Error(MI355Q_ERR_HIP,                                                                      \
                ::mi355q::fmtstr("%s:%d HIP error %d (%s) in %s", __FILE__, __LINE__, (int) e_, hipGetErrorString(e_), \
                    #expr));                                                                                           \
    } while (0)

// Run a lambda, map exceptions to status + thread-local message.
template <typename F>
mi355q_status guarded(F&& f) noexcept
{
    try
    {
        f();
        return MI355Q_OK;
    }
    catch (const Error& e)
    {
        set_last_error(e.what());
        return e.status;
    }
    catch (const std::exception& e)
    {
        set_last_error(e.what());
        return MI355Q_ERR_INTERNAL;
    }
}

// Lab switches (measured alternatives) exist only in a `make LAB=1` build: the release library
// reads no environment variable on any call path.
inline const char* lab_env(const char* name)
{
#ifdef MI355Q_LAB
    return std::getenv(name);
#else
    (void) name;
    return nullptr;
#endif
}

inline size_t dtype_size(int t)
{
    switch (t)
    {
    case MI355Q_FLOAT:
    case MI355Q_INT32: return 4;
    case MI355Q_HALF:
    case MI355Q_BF16: return 2;
    case MI355Q_INT64: return 8;
    default: return 1;
    }
}

inline int64_t dims_prod(const mi355q_dims& d, int from, int to)
{
    int64_t p = 1;
    for (int i = from; i < to; ++i)
        p *= d.d[i];
    return p;
}

// ---- device-side 16-bit helpers (bit-level, so the same code serves fp16 and bf16 templates)
struct bf16_t
{
    uint16_t x;
};
struct f16_t
{
    uint16_t x;
};

__device__ __forceinline__ float bf16_bits_to_float(uint32_t b)
{
    return __uint_as_float(b << 16);
}

// RNE float -> bf16 bits; a plain cast compiles to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN.
__device__ __forceinline__ uint16_t float_to_bf16_bits(float f)
{
    __bf16 h = (__bf16) f;
    return __builtin_bit_cast(uint16_t, h);
}

__device__ __forceinline__ float f16_bits_to_float(uint32_t b)
{
    return (float) __builtin_bit_cast(_Float16, (uint16_t) b);
}

__device__ __forceinline__ uint16_t float_to_f16_bits(float f)
{
    _Float16 h = (_Float16) f;
    return __builtin_bit_cast(uint16_t, h);
}

template <typename T>
struct T16;
template <>
struct T16<bf16_t>
{
    static constexpr bool is_bf16 = true;
    static __device__ __forceinline__ float to_float(uint32_t b)
    {
        return bf16_bits_to_float(b);
    }
    static __device__ __forceinline__ uint16_t from_float(float f)
    {
        return float_to_bf16_bits(f);
    }
    // low / high halves of a packed pair
    static __device__ __forceinline__ float lo(uint32_t p)
    {
        return __uint_as_float(p << 16);
    }
    static __device__ __forceinline__ float hi(uint32_t p)
    {
        return __uint_as_float(p & 0xffff0000u);
    }
};
template <>
struct T16<f16_t>
{
    static constexpr bool is_bf16 = false;
    static __device__ __forceinline__ float to_float(uint32_t b)
    {
        return f16_bits_to_float(b);
    }
    static __device__ __forceinline__ uint16_t from_float(float f)
    {
        return float_to_f16_bits(f);
    }
    static __device__ __forceinline__ float lo(uint32_t p)
    {
        return f16_bits_to_float(p & 0xffffu);
    }
    static __device__ __forceinline__ float hi(uint32_t p)
    {
        return f16_bits_to_float(p >> 16);
    }
};

// two floats -> one dword of two 16-bit values (a in the low half), RNE: a single
// v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 on gfx950
template <typename T>
__device__ __forceinline__ uint32_t pack2(float a, float b);
template <>
__device__ __forceinline__ uint32_t pack2<bf16_t>(float a, float b)
{
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
}
template <>
__device__ __forceinline__ uint32_t pack2<f16_t>(float a, float b)
{
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_){a, b}, f16x2_));
}

} // namespace mi355q
