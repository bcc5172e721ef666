// Shared device/host helpers for libdycon_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dycon_hip.h"

// ------------------------------------------------------------------ error plumbing
void dycon_set_error(const char* fmt, ...);

#define DYCON_REQUIRE(cond, ...)                      \
    do {                                              \
        if (!(cond)) {                                \
            dycon_set_error(__VA_ARGS__);             \
            return DYCON_ERR_INVALID;                 \
        }                                             \
    } while (0)

#define DYCON_LAUNCH_CHECK()                                                   \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            dycon_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,     \
                            hipGetErrorString(e__));                           \
            return DYCON_ERR_LAUNCH;                                           \
        }                                                                      \
    } while (0)

// dtype dispatch: DT is DYCON_F32 / DYCON_BF16
#define DYCON_DISPATCH(DT, ...)                                  \
    do {                                                         \
        if ((DT) == DYCON_F32) {                                 \
            using T = float;                                     \
            __VA_ARGS__;                                         \
        } else if ((DT) == DYCON_BF16) {                         \
            using T = __hip_bfloat16;                            \
            __VA_ARGS__;                                         \
        } else {                                                 \
            dycon_set_error("bad dtype %d", (int)(DT));          \
            return DYCON_ERR_INVALID;                            \
        }                                                        \
    } while (0)

// ------------------------------------------------------------------ scalar load/store
typedef __hip_bfloat16 bf16;

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16* p) { return __bfloat162float(*p); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16* p, float v) { *p = __float2bfloat16(v); }

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    bf16 h = __float2bfloat16(f);
    return *reinterpret_cast<unsigned short*>(&h);
}

// 16-byte vector of T: 4 floats or 8 bf16
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    float4 v;
    __device__ __forceinline__ float get(int i) const { return (&v.x)[i]; }
    __device__ __forceinline__ void set(int i, float f) { (&v.x)[i] = f; }
};
template <> struct Vec16<bf16> {
    static constexpr int N = 8;
    uint4 v;
    __device__ __forceinline__ float get(int i) const {
        unsigned w = (&v.x)[i >> 1];
        return bf16_bits_to_f32((unsigned short)((i & 1) ? (w >> 16) : (w & 0xffffu)));
    }
    __device__ __forceinline__ void set(int i, float f) {
        unsigned& w = (&v.x)[i >> 1];
        unsigned b = f32_to_bf16_bits(f);
        w = (i & 1) ? ((w & 0x0000ffffu) | (b << 16)) : ((w & 0xffff0000u) | b);
    }
};
template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec16<T>& r) {
    *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}

// ------------------------------------------------------------------ reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block sum for blockDim.x <= 1024 (multiple of 64); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 17 floats */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += smem[i];
    return r;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
