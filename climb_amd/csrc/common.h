// Shared device helpers for the climb_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CLIMB_OK 0
#define CLIMB_EINVAL (-1)
#define CLIMB_EUNSUPPORTED (-2)

#define CLIMB_DT_F32 0
#define CLIMB_DT_BF16 1
#define CLIMB_DT_SPLIT 2      // r06: a (hi, lo) PAIR of 16-bit planes: x ~= hi + lo, hi = rn16(x), lo = rn16(x - hi) (split.hip); the pointer names the hi plane

// The 16-bit operand type of the throughput mode.  One source tree, two libraries: libclimb_hip.so (bf16: 8 significant bits, fp32's range;
// BASELINE configs[1]) and libclimb_hip_f16.so (-DCLIMB_H16_F16: IEEE half, 11 significant bits -- 8x smaller operand rounding at the same
// MFMA rate, but 5 exponent bits: the host scales the loss gradient).  Every conversion goes through the helpers
// below and CLIMB_MFMA_H16; names keep "bf16" (the C ABI's CLIMB_DT_BF16 means "the library's 16-bit type").
#ifndef CLIMB_H16_F16
#define CLIMB_H16_F16 0
#endif
#if CLIMB_H16_F16
typedef _Float16 h16_scalar_t;
#define CLIMB_MFMA_H16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define CLIMB_H16_NAME "fp16"
#define CLIMB_H16_ONE_X2 0x3C003C00u          // two packed 1.0
#else
typedef __bf16 h16_scalar_t;
#define CLIMB_MFMA_H16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define CLIMB_H16_NAME "bf16"
#define CLIMB_H16_ONE_X2 0x3F803F80u
#endif
typedef unsigned short bf16_t;  // raw bits of the 16-bit type
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) h16_scalar_t bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define LAUNCH_CHECK()                                   \
  do {                                                   \
    hipError_t e__ = hipGetLastError();                  \
    if (e__ != hipSuccess) return (int)e__;              \
  } while (0)

#if CLIMB_H16_F16
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// two packed 16-bit values -> floats (lo = bits 0..15)
__device__ __forceinline__ float h16lo_to_f32(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float h16hi_to_f32(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }
#else
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float h16lo_to_f32(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float h16hi_to_f32(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
#endif
// fp32 -> 16 bit, round-to-nearest-even: the casts lower to the gfx950 hardware converters v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32
// (fp16: values beyond +-65504 become inf -- forward activations of this model are far inside, gradients are scaled by the host)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) h16_scalar_t bf16x2_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  h16_scalar_t h = (h16_scalar_t)f;
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
  return make_float4(h16lo_to_f32(r.x), h16hi_to_f32(r.x), h16lo_to_f32(r.y), h16hi_to_f32(r.y));
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  uint2 r;
  r.x = pack_bf16x2(v.x, v.y);
  r.y = pack_bf16x2(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = r;
}

// r06, split operands (split.hip): element type of the hi plane of a (hi, lo) pair -- same size as bf16_t, so row / column arithmetic on a
// `sp16_t*` is that of the 16-bit tensor; the lo plane lies `lo` ELEMENTS behind it.  hi = rn16(x), lo = rn16(x - hi): with bf16 planes
// hi + lo carries 16 significant bits of x (and fp32's range), the three MFMA products hi.hi + hi.lo + lo.hi recover a product to ~2^-16.
struct sp16_t { bf16_t v; };
__device__ __forceinline__ void split_st4(bf16_t* hi, long lo, float4 v) {
  uint2 h, l;
  h.x = pack_bf16x2(v.x, v.y);
  h.y = pack_bf16x2(v.z, v.w);
  l.x = pack_bf16x2(v.x - h16lo_to_f32(h.x), v.y - h16hi_to_f32(h.x));
  l.y = pack_bf16x2(v.z - h16lo_to_f32(h.y), v.w - h16hi_to_f32(h.y));
  *reinterpret_cast<uint2*>(hi) = h;
  *reinterpret_cast<uint2*>(hi + lo) = l;
}
// st4 with a lo-plane offset that only the split type uses
template <typename T> __device__ __forceinline__ void st4x(T* p, long, float4 v) { st4(p, v); }
template <> __device__ __forceinline__ void st4x<sp16_t>(sp16_t* p, long lo, float4 v) { split_st4(reinterpret_cast<bf16_t*>(p), lo, v); }

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

// GELU pair for the bf16 GEMM epilogues (r05: sigmoid form; r01 - r04 used odd degree-15 polynomials of 11 - 14 VALU operations per element, |error|
// <= 6.8e-4 / 6.2e-4, which made the GELU epilogues VALU-bound and was a visible deviation from the reference's erf GELU, HF/modeling_vilt.py:393).
//   gelu(x) ~= x * sigma(x * P(t)),  t = min(x^2, 64),  P(t) = a + b t + c t^2      (the tanh approximation is the two-term member of this family)
// fitted minimax on [-12, 12] against the erf form: |gelu_fast - gelu| <= 2.6e-5 absolute in fp32 arithmetic (27 x closer than the polynomial; below
// the bf16 rounding of every value it is stored as, and below north_star's 1e-3 by itself), and the DERIVATIVE OF THE SAME EXPRESSION
//   gelu'(x) ~= s + x s (1 - s) (a + 3 b t + 5 c t^2),  s = sigma(x P(t))
// is within 1.1e-4 of the exact gelu'.  7 full-rate operations + v_exp_f32 + v_rcp_f32 for the value (the transcendentals cost ~2 extra cycles each
// beside MFMAs: MI355X_MICROARCH.md, "price of one filler"), 5 more for the derivative of the same element.  The clamp of t keeps P positive
// (c < 0: P would change sign at |x| = 11.1) and costs nothing in accuracy (|x| > 8: sigma is 0 or 1 to 1e-12).  Saturation is exact where it
// matters: exp2 overflows to +inf -> rcp 0 -> x * 0 for very negative x, exp2 underflows to 0 -> x * 1 for very positive x.
// The fp32 parity path keeps erff().
#define GELU_SIG_A 1.5950158f
#define GELU_SIG_B 7.4011292e-2f
#define GELU_SIG_C (-7.0303358e-4f)
#define GELU_SIG_NL2E_A (-2.3011212f)        // -log2(e) * {a, b, c}: the exponent is taken in base 2
#define GELU_SIG_NL2E_B (-0.10677572f)
#define GELU_SIG_NL2E_C 1.014263e-3f
#define GELU_SIG_DA GELU_SIG_A               // a, 3 b, 5 c
#define GELU_SIG_DB 0.22203387f
#define GELU_SIG_DC (-3.5151679e-3f)
__device__ __forceinline__ float gelu_sig_t(float x) { return fminf(x * x, 64.0f); }
__device__ __forceinline__ float gelu_sig_s(float x, float t) {          // sigma(x P(t))
  const float p = fmaf(fmaf(GELU_SIG_NL2E_C, t, GELU_SIG_NL2E_B), t, GELU_SIG_NL2E_A);
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
}
__device__ __forceinline__ float gelu_sig_d(float x, float t, float s) {  // derivative, given s
  const float q = fmaf(fmaf(GELU_SIG_DC, t, GELU_SIG_DB), t, GELU_SIG_DA);
  return fmaf(x * q, fmaf(-s, s, s), s);
}
__device__ __forceinline__ float gelu_fast(float x) { return x * gelu_sig_s(x, gelu_sig_t(x)); }
__device__ __forceinline__ float dgelu_fast(float x) {
  const float t = gelu_sig_t(x);
  return gelu_sig_d(x, t, gelu_sig_s(x, t));
}
__device__ __forceinline__ void gelu_fast_pair(float x, float& y, float& dy) {      // value and derivative of one element (shared sigma)
  const float t = gelu_sig_t(x), s = gelu_sig_s(x, t);
  y = x * s;
  dy = gelu_sig_d(x, t, s);
}

// torch.optim.AdamW (decoupled decay), REF/modeling/vilt.py:205-215 -- ONE definition for the flat optimizer pass (optim.hip) and the fused epilogue of
// the grouped weight-gradient launch (gemm_bf16_tnp.hip), so that the two produce the same bits:
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
struct AdamGroup { float lr, wd, beta1, beta2, eps, bc1, bc2, pad; };
__device__ __forceinline__ void adamw_update(float& p, float& m, float& v, float g, const AdamGroup& G, float isb2, float step) {
#pragma clang fp contract(off)      // no FMA contraction: the two kernels that inline this must round identically whatever surrounds the call
  p *= 1.f - G.lr * G.wd;
  m = G.beta1 * m + (1.f - G.beta1) * g;
  v = G.beta2 * v + (1.f - G.beta2) * g * g;
  p -= step * m / (sqrtf(v) * isb2 + G.eps);
}

// epilogue codes shared by the f32 and bf16 GEMMs
#define EPI_NONE 0    // C = acc + bias
#define EPI_GELU 1    // aux_out = acc + bias ; C = gelu(aux_out)
#define EPI_RESID 2   // C = acc + bias + aux      (aux is f32 [M,N], leading dim ldaux)
#define EPI_DGELU 3   // C = acc * gelu'(aux)      (aux has C's dtype)
#define EPI_TANH 4    // C = tanh(acc + bias)
#define EPI_SILU 5    // aux_out = acc + bias ; C = silu(aux_out)                  (adapter down-projection)
#define EPI_DSILU 6   // C = acc * silu'(aux)       (aux has C's dtype)             (adapter backward)
#define EPI_RESID2 7  // C = acc + bias + aux (f32) + aux2 (C's operand dtype)     (adapter up-projection + both residuals)
#define EPI_GELUD 8   // aux_out = gelu'(acc + bias) ; C = gelu(acc + bias)        (r05: the backward multiplies by the SAVED derivative, 16-bit outputs only)
#define EPI_MUL 9     // C = acc * aux               (aux has C's dtype: the derivative EPI_GELUD saved)

// r06, split-operand mode (nt4 epilogues of split.hip's GEMMs; fp32 C / aux):
#define EPI_GELU_SP 10  // C = acc + bias (fp32 pre-activation u);  aux_out (split planes) = gelu(u)
#define EPI_DGELU_SP 11 // aux_out (split planes) = acc * gelu'(aux), aux = fp32 u;  C is not written
// erf-form GELU and its derivative for those epilogues: Abramowitz-Stegun 7.1.26, erfc(z) = (a1 t + .. + a5 t^5) exp(-z^2), t = 1 / (1 + p z), |error| <=
// 1.5e-7 -- fp32's own resolution -- in 17 / 19 operations per element (erff / expf of the exact forms above cost ~40: more than the MFMA time of the window
// a 32 x 32 block is drained in).  Phi(x) for x < 0 is taken as erfc(|x| / sqrt 2) / 2 directly, so the tail has no 1 - (1 - eps) cancellation; the
// derivative shares the exponential: gelu'(x) = Phi(x) + x exp(-x^2 / 2) / sqrt(2 pi).
__device__ __forceinline__ void gelu_as_parts(float x, float& cdf, float& e) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);          // exp(-x^2 / 2) = 2^(-x^2 log2(e) / 2)
  const float q = 0.5f * poly * e;                                        // Phi(-|x|)
  cdf = x < 0.f ? q : 1.0f - q;
}
__device__ __forceinline__ float gelu_as(float x) {
  float cdf, e;
  gelu_as_parts(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float dgelu_as(float x) {
  float cdf, e;
  gelu_as_parts(x, cdf, e);
  return fmaf(x * 0.39894228040143267794f, e, cdf);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
  const float sg = 1.0f / (1.0f + __expf(-x));
  return sg * (1.0f + x * (1.0f - sg));
}
