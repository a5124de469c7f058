// Shared device/host helpers for libretinanet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "retinanet_hip.h"

#define RN_API extern "C" __attribute__((visibility("default")))

#define RN_WAVE 64

#define RN_LAUNCH_CHECK()                                 \
    do {                                                  \
        hipError_t e__ = hipGetLastError();               \
        if (e__ != hipSuccess) return (int)e__;           \
    } while (0)

#define RN_HIP(call)                                      \
    do {                                                  \
        hipError_t e__ = (call);                          \
        if (e__ != hipSuccess) return (int)e__;           \
    } while (0)

namespace rn {

// fp16 halves of a packed dword (explicit bit ops; vector bit_casts of half2 are avoided on purpose)
static __device__ __forceinline__ float half_lo(const uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
static __device__ __forceinline__ float half_hi(const uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
static __device__ __forceinline__ uint32_t half_pack(const float a, const float b) {
    const uint32_t lo = __builtin_bit_cast(uint16_t, (_Float16)a), hi = __builtin_bit_cast(uint16_t, (_Float16)b);
    return lo | (hi << 16);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- dtype traits: 16-byte vectors of VEC elements <-> float[VEC] -----------
template <int DT> struct dt;

template <> struct dt<RN_F32> {
    typedef float elem;
    static constexpr int VEC = 4;
    static __device__ __forceinline__ void unpack(const u32x4 v, float (&f)[4]) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ u32x4 pack(const float (&f)[4]) {
        u32x4 v; v.x = __float_as_uint(f[0]); v.y = __float_as_uint(f[1]);
        v.z = __float_as_uint(f[2]); v.w = __float_as_uint(f[3]); return v;
    }
    static __device__ __forceinline__ float ld(const void *p, int64_t i) { return ((const float *)p)[i]; }
    static __device__ __forceinline__ void st(void *p, int64_t i, float v) { ((float *)p)[i] = v; }
};

template <> struct dt<RN_BF16> {
    typedef __bf16 elem;
    static constexpr int VEC = 8;
    static __device__ __forceinline__ void unpack(const u32x4 v, float (&f)[8]) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ uint32_t pk(float a, float b) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        bf2 r; r.x = (__bf16)a; r.y = (__bf16)b;     // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
        return __builtin_bit_cast(uint32_t, r);
    }
    static __device__ __forceinline__ u32x4 pack(const float (&f)[8]) {
        u32x4 v; v.x = pk(f[0], f[1]); v.y = pk(f[2], f[3]); v.z = pk(f[4], f[5]); v.w = pk(f[6], f[7]); return v;
    }
    static __device__ __forceinline__ float ld(const void *p, int64_t i) {
        return __uint_as_float(((uint32_t)((const uint16_t *)p)[i]) << 16);
    }
    static __device__ __forceinline__ void st(void *p, int64_t i, float v) { ((__bf16 *)p)[i] = (__bf16)v; }
};

template <> struct dt<RN_F16> {
    typedef _Float16 elem;
    static constexpr int VEC = 8;
    static __device__ __forceinline__ void unpack(const u32x4 v, float (&f)[8]) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = half_lo(w[i]); f[2 * i + 1] = half_hi(w[i]); }
    }
    static __device__ __forceinline__ uint32_t pk(float a, float b) { return half_pack(a, b); }
    static __device__ __forceinline__ u32x4 pack(const float (&f)[8]) {
        u32x4 v; v.x = pk(f[0], f[1]); v.y = pk(f[2], f[3]); v.z = pk(f[4], f[5]); v.w = pk(f[6], f[7]); return v;
    }
    static __device__ __forceinline__ float ld(const void *p, int64_t i) { return (float)((const _Float16 *)p)[i]; }
    static __device__ __forceinline__ void st(void *p, int64_t i, float v) { ((_Float16 *)p)[i] = (_Float16)v; }
};

// ---- matrix-instruction traits of the two 16-bit element types (same rate on gfx950: v_mfma_f32_{16x16x32,32x32x16}_{bf16,f16}) ----
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int DT> struct mma;
template <> struct mma<RN_BF16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 m16(const frag a, const frag b, const f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 m32(const frag a, const frag b, const f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint16_t dn(const float f) { return (uint16_t)(dt<RN_BF16>::pk(f, 0.0f) & 0xffffu); }      // f32 -> element bits (RNE)
    static __device__ __forceinline__ float lo(const uint32_t w) { return __uint_as_float(w << 16); }                            // halves of a packed dword
    static __device__ __forceinline__ float hi(const uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
};
template <> struct mma<RN_F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4 m16(const frag a, const frag b, const f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x16 m32(const frag a, const frag b, const f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint16_t dn(const float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
    static __device__ __forceinline__ float lo(const uint32_t w) { return half_lo(w); }
    static __device__ __forceinline__ float hi(const uint32_t w) { return half_hi(w); }
};

// ---- wave64 reductions -------------------------------------------------------
static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, RN_WAVE);
    return v;
}
static __device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = __shfl_xor((unsigned)u, o, RN_WAVE), hi = __shfl_xor((unsigned)(u >> 32), o, RN_WAVE);
        v += __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    }
    return v;
}
static __device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, RN_WAVE);
    return v;
}

static inline bool aligned(const void *p, size_t a) { return (((uintptr_t)p) & (a - 1)) == 0; }

// Opt-in for more than 64 KiB of dynamic LDS.  The attribute lives with the DEVICE's copy of the code object, so the "already
// done" note is kept per device (a process that drives a second GPU must set it there too) and per kernel (one `DynLdsOptIn`
// object per kernel instantiation); relaxed atomics: two threads racing here both set the same value, which is harmless.
struct DynLdsOptIn {
    int have[64];
    int ensure(const void *kernel, const int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        const bool slot = dev >= 0 && dev < 64;
        if (slot && __atomic_load_n(&have[dev], __ATOMIC_RELAXED) >= bytes) return RN_OK;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
        if (slot) __atomic_store_n(&have[dev], bytes, __ATOMIC_RELAXED);
        return RN_OK;
    }
};

}  // namespace rn
