// common.h — shared device/host helpers for libmcn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/mcn.h"

#define MCN_NUM_CU 256                       // compute units of one MI355X (8 XCDs x 32): launch-shape heuristics only

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16_t;                              // fp16 storage: the reference's own low precision (convnet.py:63), with loss scaling
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

#define MCN_WAVE 64

// ---- error plumbing (host) ---------------------------------------------------------------
void mcn_set_error(const char* fmt, ...);
#define MCN_FAIL(code, ...)        \
    do {                           \
        mcn_set_error(__VA_ARGS__); \
        return (code);             \
    } while (0)
#define MCN_CHECK_LAUNCH()                                                         \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) MCN_FAIL(MCN_E_LAUNCH, "%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
    } while (0)

static inline size_t mcn_dtype_size(mcn_dtype t) { return t == MCN_F32 ? 4 : 2; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- device helpers ----------------------------------------------------------------------
template <typename T>
struct VecTraits;
template <>
struct VecTraits<float> {
    static constexpr int CE = 4;  // elements per 16-byte chunk
};
template <>
struct VecTraits<bf16_t> {
    static constexpr int CE = 8;
    typedef bf16x8 V8;
    typedef bf16x4 V4;
};
template <>
struct VecTraits<f16_t> {
    static constexpr int CE = 8;
    typedef f16x8 V8;
    typedef f16x4 V4;
};
// mcn_dtype of a storage type (host side: tile heuristics, workspace sizes)
template <typename T>
struct DtypeOf;
template <>
struct DtypeOf<float> { static constexpr mcn_dtype value = MCN_F32; };
template <>
struct DtypeOf<bf16_t> { static constexpr mcn_dtype value = MCN_BF16; };
template <>
struct DtypeOf<f16_t> { static constexpr mcn_dtype value = MCN_F16; };
static inline bool mcn_dtype_ok(mcn_dtype t) { return t == MCN_F32 || t == MCN_BF16 || t == MCN_F16; }

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <>
__device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }          // round to nearest even, overflow -> inf (as tf.cast)

// sigmoid on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1e-6 relative): the precise expf + IEEE division cost
// ~30 VALU instructions per element, which made the BN+swish kernels VALU-bound instead of HBM-bound.  The reciprocal is the raw
// v_rcp_f32 (1 ulp): __frcp_rn is the correctly ROUNDED reciprocal and expands to the whole division sequence (2 v_div_scale, v_rcp,
// 4 fma, v_div_fmas, v_div_fixup — 10 of the 21 VALU instructions per element of the BN + swish backward, round 5 ISA count).
// exp(-z) = inf gives rcp(inf) = 0, exp(-z) flushed to 0 gives 1: the ends are exact without a fix-up.
__device__ __forceinline__ float fast_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }

// 16-byte chunk <-> fp32 lanes
template <typename T>
struct Chunk;
template <>
struct Chunk<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = f; }
};
template <>
struct Chunk<bf16_t> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = (bf16_t)f; }
};

template <>
struct Chunk<f16_t> {
    static constexpr int N = 8;
    f16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float f) { v[i] = (f16_t)f; }
};

// 16-byte chunk loads of the element-wise / BN / pooling kernels are streaming (non-temporal): every tensor element is read once per
// kernel and should not displace what the GEMMs keep in L2.  MCN_NT_HINT: 1 = non-temporal loads (default), 2 = non-temporal stores,
// 3 = both, 0 = neither.  Same-box A/B of the whole step (B = 256): bf16 22.0-22.3 -> 21.45 ms, fp32 68.65 -> 68.2 ms with 1; stores
// gain nothing (22.1 ms) and 3 is between (21.6 ms).
#ifndef MCN_NT_HINT
#define MCN_NT_HINT 1
#endif
template <typename T>
__device__ __forceinline__ Chunk<T> load_chunk(const T* p) {
    Chunk<T> c;
#if MCN_NT_HINT & 1
    *reinterpret_cast<i32x4*>(&c.v) = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(p));
#else
    *reinterpret_cast<i32x4*>(&c.v) = *reinterpret_cast<const i32x4*>(p);
#endif
    return c;
}
template <typename T>
__device__ __forceinline__ void store_chunk(T* p, const Chunk<T>& c) {
#if MCN_NT_HINT & 2
    __builtin_nontemporal_store(*reinterpret_cast<const i32x4*>(&c.v), reinterpret_cast<i32x4*>(p));
#else
    *reinterpret_cast<i32x4*>(p) = *reinterpret_cast<const i32x4*>(&c.v);
#endif
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// XCD-aware bijective remap of a linear block id: blocks b and b+8 share an XCD (round-robin
// dispatch), so give each XCD a contiguous range of logical tiles (operand panels stay in its L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
