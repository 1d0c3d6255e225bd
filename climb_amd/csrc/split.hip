// Split-operand arithmetic (r06): the parity-grade fast mode.  Every GEMM operand is a PAIR of 16-bit planes, x ~= hi + lo with hi = rn16(x) and
// lo = rn16(x - hi), and a product is three MFMA passes, hi.hi + hi.lo + lo.hi, accumulated in fp32 -- with bf16 planes 16 significant bits per
// operand (2^-17 relative rounding) at fp32's range, at a third of the 16-bit MFMA rate instead of the 1/16 of v_mfma_f32_32x32x2_f32.  The GEMM
// kernels are the throughput mode's (gemm_bf16_nt4.hip, gemm_bf16.hip, gemm_bf16_tnp.hip) walking a reduction of three products per k-tile; this file holds
// the passes that PRODUCE the planes from fp32 tensors and the C entry points of the split GEMMs.
//
// Layout: a split tensor is [2][rows][ld] 16-bit, the hi plane first; the lo plane of a tensor of `rows` rows lies rows * ld elements behind
// (the CLIMB_DT_SPLIT convention of include/climb_hip.h); the flat weight shadow is [2][total].
#include "gemm_bf16_nt.h"

// y (split) = f(x): mode 0 f = x; 1 f = gelu(x) (exact erf form: HF/modeling_vilt.py:393, ACT2FN["gelu"]); 2 f = x * gelu'(aux)
template <int MODE>
__global__ __launch_bounds__(256) void split_f32_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ y, long ldy, long lo, long n4, int c4,
                                                         const float* __restrict__ aux, long ldaux) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / c4;
    const int c = (int)(i - r * c4) * 4;
    float4 v = ld4(x + r * ldx + c);
    if (MODE == 1) v = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
    if (MODE == 2) {
      const float4 u = ld4(aux + r * ldaux + c);
      v = make_float4(v.x * dgelu_f(u.x), v.y * dgelu_f(u.y), v.z * dgelu_f(u.z), v.w * dgelu_f(u.w));
    }
    split_st4(y + r * ldy + c, lo, v);
  }
}

extern "C" int climb_split_f32(const float* x, long ldx, void* y, long ldy, long lo_off, int M, int C, int mode, const float* aux, long ldaux, void* stream) {
  if (!x || !y || M <= 0 || C <= 0 || (C % 4) || (ldx % 4) || (ldy % 4) || (lo_off % 4) || mode < 0 || mode > 2 || (mode == 2 && (!aux || (ldaux % 4)))) return CLIMB_EINVAL;
  const long n4 = (long)M * (C / 4);
  const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  hipStream_t st = (hipStream_t)stream;
#define SP(MODE_) hipLaunchKernelGGL((split_f32_kernel<MODE_>), dim3(grid), dim3(256), 0, st, x, ldx, (bf16_t*)y, ldy, lo_off, n4, C / 4, aux, ldaux)
  if (mode == 0) SP(0); else if (mode == 1) SP(1); else SP(2);
#undef SP
  LAUNCH_CHECK();
  return CLIMB_OK;
}

int climb_nt4_split_launch(const bf16_t* A, long lda, long a_lo, const bf16_t* B, long ldb, long b_lo, float* C, long ldc, int M, int N, int K, const float* bias,
                           int epi, const void* aux, long ldaux, hipStream_t st, bf16_t* aux_out = nullptr, long ldauxo = 0, long o_lo = 0);
int climb_nt_split_generic(const bf16_t* A, long lda, long a_lo, const bf16_t* B, long ldb, long b_lo, float* C, long ldc, int M, int N, int K, const float* bias,
                           int epi, const float* aux, long ldaux, hipStream_t st);

extern "C" int climb_gemm_bf16_tn(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K, float* dbias, void* stream);

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// C[M,N] (fp32) = epi(A B^T + bias), A = (hi, lo) planes [M,K] (lda; the lo plane a_lo elements behind), B likewise [N,K]: the reduction runs over three
// products per k-tile -- A hi.B hi, A hi.B lo, A lo.B hi.  epi: 0 none, 2 + aux (fp32 residual [M,N]).  K % 64 == 0.
extern "C" int climb_gemm_split_nt(const void* A, long lda, long a_lo, const void* B, long ldb, long b_lo, float* C, long ldc, int M, int N, int K, const float* bias,
                                   int epi, const float* aux, long ldaux, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || (K % GB_BK) || (N % 4) || (lda % 8) || (ldb % 8) || (ldc % 4) || (a_lo % 8) || (b_lo % 8) || !al16(A) || !al16(B) || !al16(C))
    return CLIMB_EINVAL;
  if (epi != EPI_NONE && epi != EPI_RESID) return CLIMB_EINVAL;
  if (epi == EPI_RESID && (!aux || (ldaux % 4))) return CLIMB_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int rc = climb_nt4_split_launch((const bf16_t*)A, lda, a_lo, (const bf16_t*)B, ldb, b_lo, C, ldc, M, N, K, bias, epi, aux, ldaux, st);
  if (rc == CLIMB_OK) { LAUNCH_CHECK(); return CLIMB_OK; }
  if (rc != CLIMB_EUNSUPPORTED) return rc;
  return climb_nt_split_generic((const bf16_t*)A, lda, a_lo, (const bf16_t*)B, ldb, b_lo, C, ldc, M, N, K, bias, epi, aux, ldaux, st);
}

// The MLP's two activation epilogues on the four-wave kernel (HF:393, :397-414): epi 10: C (fp32 [M,N]) = A B^T + bias, out (split planes [M,N], ldo; the lo plane
// o_lo elements behind) = gelu(C); epi 11: out = (A B^T) * gelu'(aux), aux fp32 [M,N], C unused.  GELU in the erf form to 1.5e-7 (common.h: gelu_as).
// Only the shapes climb_gemm_split_nt_takes_act() names; CLIMB_EUNSUPPORTED otherwise (the caller then runs the plain product + climb_split_f32 mode 1 / 2).
extern "C" int climb_gemm_split_nt_takes_act(int M, int N, int K) {
  return (M > 0 && N > 0 && K > 0 && (M % 192) == 0 && (N % 192) == 0 && ((M / 192) % 8) == 0 && (K % GB_BK) == 0 && 3 * K >= 10 * GB_BK) ? 1 : 0;
}
extern "C" int climb_gemm_split_nt_act(const void* A, long lda, long a_lo, const void* B, long ldb, long b_lo, float* C, long ldc, void* out, long ldo, long o_lo, int M, int N,
                                       int K, const float* bias, int epi, const float* aux, long ldaux, void* stream) {
  if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0 || (K % GB_BK) || (N % 4) || (lda % 8) || (ldb % 8) || (ldo % 8) || (a_lo % 8) || (b_lo % 8) || (o_lo % 8) || !al16(A) || !al16(B) ||
      !al16(out))
    return CLIMB_EINVAL;
  if (epi != EPI_GELU_SP && epi != EPI_DGELU_SP) return CLIMB_EINVAL;
  if (epi == EPI_GELU_SP && (!C || (ldc % 4) || !al16(C))) return CLIMB_EINVAL;
  if (epi == EPI_DGELU_SP && (!aux || (ldaux % 4))) return CLIMB_EINVAL;
  if (!climb_gemm_split_nt_takes_act(M, N, K)) return CLIMB_EUNSUPPORTED;
  int rc = climb_nt4_split_launch((const bf16_t*)A, lda, a_lo, (const bf16_t*)B, ldb, b_lo, C, ldc, M, N, K, bias, epi, aux, ldaux, (hipStream_t)stream, (bf16_t*)out, ldo, o_lo);
  if (rc == CLIMB_OK) LAUNCH_CHECK();
  return rc;
}

// C[N,K] (fp32) += A^T B over the tokens, A = (hi, lo) planes [M,N], B = (hi, lo) planes [M,K]: three ordinary weight-gradient launches (hi.hi, hi.lo,
// lo.hi) -- the path of shapes the grouped launch does not take (climb_gemm_bf16_tn_grouped with split problems is the step's).  dbias += column sums of
// A hi + A lo.
extern "C" int climb_gemm_split_tn(const void* A, long lda, long a_lo, const void* B, long ldb, long b_lo, float* C, long ldc, int M, int N, int K, float* dbias,
                                   void* stream) {
  if (!A || !B || !C) return CLIMB_EINVAL;
  const bf16_t* Ah = (const bf16_t*)A;
  const bf16_t* Bh = (const bf16_t*)B;
  int rc = climb_gemm_bf16_tn(Ah, lda, Bh, ldb, C, ldc, M, N, K, dbias, stream);
  if (rc != CLIMB_OK) return rc;
  rc = climb_gemm_bf16_tn(Ah, lda, Bh + b_lo, ldb, C, ldc, M, N, K, nullptr, stream);
  if (rc != CLIMB_OK) return rc;
  return climb_gemm_bf16_tn(Ah + a_lo, lda, Bh, ldb, C, ldc, M, N, K, dbias, stream);
}
