// Shared device helpers for the climb_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CLIMB_OK 0
#define CLIMB_EINVAL (-1)
#define CLIMB_EUNSUPPORTED (-2)

#define CLIMB_DT_F32 0
#define CLIMB_DT_BF16 1

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define LAUNCH_CHECK()                                   \
  do {                                                   \
    hipError_t e__ = hipGetLastError();                  \
    if (e__ != hipSuccess) return (int)e__;              \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-to-nearest-even: the casts lower to the gfx950 hardware converter v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(uint32_t, h);
}

template <typename T> struct Act;
template <> struct Act<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Act<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 4 consecutive elements <-> float4
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  uint2 r = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                     __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  uint2 r;
  r.x = pack_bf16x2(v.x, v.y);
  r.y = pack_bf16x2(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = r;
}

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

// exact (erf) GELU and its derivative: ACT2FN['gelu'] / nn.GELU()
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Fast GELU pair for the bf16 epilogues: erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below bf16 resolution),
// ONE v_exp shared between erf(x/sqrt2) and the Gaussian of gelu'.  The fp32 path keeps erff().
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  const float e = __expf(-z * z);                                    // = exp(-x^2/2)
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  pdf = 0.39894228040143267794f * e;
}
__device__ __forceinline__ float gelu_fast(float x) { float c, p; gelu_parts(x, c, p); return x * c; }
__device__ __forceinline__ float dgelu_fast(float x) { float c, p; gelu_parts(x, c, p); return c + x * p; }

// epilogue codes shared by the f32 and bf16 GEMMs
#define EPI_NONE 0    // C = acc + bias
#define EPI_GELU 1    // aux_out = acc + bias ; C = gelu(aux_out)
#define EPI_RESID 2   // C = acc + bias + aux      (aux is f32 [M,N], leading dim ldaux)
#define EPI_DGELU 3   // C = acc * gelu'(aux)      (aux has C's dtype)
#define EPI_TANH 4    // C = tanh(acc + bias)
#define EPI_SILU 5    // aux_out = acc + bias ; C = silu(aux_out)                  (adapter down-projection)
#define EPI_DSILU 6   // C = acc * silu'(aux)       (aux has C's dtype)             (adapter backward)
#define EPI_RESID2 7  // C = acc + bias + aux (f32) + aux2 (C's operand dtype)     (adapter up-projection + both residuals)

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
  const float sg = 1.0f / (1.0f + __expf(-x));
  return sg * (1.0f + x * (1.0f - sg));
}
