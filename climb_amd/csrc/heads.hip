// Skinny fp32 products of the pooler and the task heads (REF/modeling/vilt.py:179-203, HF modeling_vilt.py:650-663): M = batch rows (<= a few
// hundred), N and K up to a few thousand, exact fp32 on v_mfma_f32_16x16x4_f32.  r04.
//
// Why not climb_gemm_f32: at M = 64 its 64 x 64 tiles give 12 - 49 workgroups that walk K one 16-deep round trip at a time (61 us for the
// pooler), so since r01 the 16-bit mode split K over workgroups with fp32 atomics into a zeroed C -- which forbids every epilogue (tanh, GELU',
// tanh' ran as separate 5 us launches, the bias gradients as K = 1 GEMMs, the zeroing as another launch each): 36 launches and 0.35 ms per step
// between the last encoder layer and the first backward GEMM, for 2.5 GFLOP.  Here
//   * one workgroup owns ALL rows of a 16-column strip of C: N / 16 workgroups (48 ... 196), 16 waves that split K sixteen ways, partial sums
//     meet in LDS in a fixed order -- no atomics, no zeroing, the same result on every run (both arithmetic modes use it);
//   * so the epilogue sees finished values: bias, tanh, x (1 - aux^2), x gelu'(aux), and -- because the strip holds every row -- the COLUMN
//     SUMS of the result (the bias gradient of the layer below) come for free; the column sums of the A operand (bias gradient of the layer
//     above: d(logits)) are dealt over the workgroups by k-block;
//   * operands go global -> registers -> MFMA (16-byte loads along k, 10 - 15 in flight per lane): every workgroup reads all of A
//     (64 x K floats, L2-resident) and its own strip of B once.
// lane (r = lane & 15, q = lane >> 4) of a wave holds, for a 16-deep k-block, k = k0 + 4 q .. + 3 of row r (A: four 16-row blocks; B: column
// r of the strip); MFMA j of the block multiplies the j-th of those (any assignment of k to MFMA slots is valid as long as A and B agree).
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;

#define SK_WAVES 16
#define SK_ROWS 64
#define SK_COLS 16

struct SkinnyArgs {
  const float* A; long lda;               // [M][K], k contiguous
  const float* B; long sbn, sbk;          // B(n, k) = B[n * sbn + k * sbk]
  float* C; long ldc;
  int M, N, K;
  const float* bias;                      // [N] or null
  int epi;                                // 0 none, 1 tanh, 2 x (1 - aux^2), 3 x gelu'(aux)
  const float* aux; long ldaux;           // [M][N] (epi 2, 3)
  float* colsum; float cs_beta;           // [N] or null: colsum[n] = cs_beta * colsum[n] + sum_m C[m][n]
  float* acol; float acol_beta;           // [K] or null: acol[k] = acol_beta * acol[k] + sum_m A[m][k]
};

template <bool KCONTIG, int PROBE>
__global__ __launch_bounds__(64 * SK_WAVES) void skinny_f32_kernel(SkinnyArgs a) {
  __shared__ __attribute__((aligned(16))) float red[SK_WAVES][SK_ROWS * SK_COLS];      // 64 KB
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * SK_COLS, m0 = blockIdx.y * SK_ROWS;
  const int K = a.K;
  const int nkb = (K + 15) >> 4;                              // 16-deep k-blocks, dealt round-robin to the waves
  f32x4 acc[4];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int n = n0 + r;
  const bool nok = n < a.N;
  const float* bp = a.B + (long)(nok ? n : 0) * a.sbn;
  const float* ap[4];
  bool mok[4];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {
    const int m = m0 + 16 * rb + r;
    mok[rb] = m < a.M;
    ap[rb] = a.A + (long)(mok[rb] ? m : 0) * a.lda;
  }
  const bool want_acol = a.acol != nullptr && blockIdx.y == 0;          // (only defined for M <= 64: one row tile)
  // blocks kb = w, w + 16, ...; SK_UNR of them per trip with every load issued before the first MFMA
  auto block = [&](int kb, const f32x4 (&av)[4], const f32x4& bv, bool valid) {
    const int k = kb * 16 + 4 * q;
    if constexpr (PROBE == 1) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb] += av[rb] * bv;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb][j], bv[j], acc[rb], 0, 0, 0);
    }
    // column sums of A: k-block kb belongs to workgroup kb mod gridDim.x; the 16 lanes that share q hold the 64 rows of 4 columns
    if (want_acol && valid && (kb % (int)gridDim.x) == (int)blockIdx.x) {
      f32x4 s = (av[0] + av[1]) + (av[2] + av[3]);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += __shfl_xor(s[j], o, 64);
      }
      if (r == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k + j < K) a.acol[k + j] = (a.acol_beta != 0.f ? a.acol_beta * a.acol[k + j] : 0.f) + s[j];
      }
    }
  };
  const int nfull = K >> 4;                                  // blocks with all 16 k in range
  constexpr int SK_UNR = 2;                                  // (128 registers at 16 waves; four waves per SIMD cover each other's round trips)
  // straight-line trips: every load is unconditional (rows / columns / blocks out of range read a clamped address and are zeroed by a select), so the
  // compiler issues a trip's loads back to back and waits once -- predicated loads put each one in its own basic block behind a vmcnt(0)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int kb = w; kb < nfull; kb += SK_UNR * SK_WAVES) {
    f32x4 av[SK_UNR][4], bv[SK_UNR];
#pragma unroll
    for (int u = 0; u < SK_UNR; ++u) {
      const int kbu = kb + u * SK_WAVES;                      // wave-uniform
      const int k = (kbu < nfull ? kbu : nfull - 1) * 16 + 4 * q;
      if constexpr (PROBE == 2) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) av[u][rb] = (f32x4){1.f, 2.f, 3.f, (float)k};
        bv[u] = (f32x4){1.f, 2.f, 3.f, (float)k};
      } else {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) av[u][rb] = *reinterpret_cast<const f32x4*>(ap[rb] + k);
        if (KCONTIG) bv[u] = *reinterpret_cast<const f32x4*>(bp + k);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[u][j] = bp[(long)(k + j) * a.sbk];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < SK_UNR; ++u) {
      const bool valid = kb + u * SK_WAVES < nfull;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) av[u][rb] = mok[rb] ? av[u][rb] : zero4;
      bv[u] = (nok && valid) ? bv[u] : zero4;
      block(kb + u * SK_WAVES, av[u], bv[u], valid);
    }
  }
  if (nkb > nfull && (nfull % SK_WAVES) == w) {              // the ragged last block (K % 16 != 0): element-wise loads, one wave
    const int k = nfull * 16 + 4 * q;
    f32x4 av[4], bv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool kok = k + j < K;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) av[rb][j] = (kok && mok[rb]) ? ap[rb][k + j] : 0.f;
      bv[j] = (kok && nok) ? bp[(long)(k + j) * a.sbk] : 0.f;
    }
    block(nfull, av, bv, true);
  }
  // D layout of 16x16x4: lane (col = lane & 15, rows 4 * (lane >> 4) .. + 3).  red[w][row][col]
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[w][(16 * rb + 4 * q + i) * SK_COLS + r] = acc[rb][i];
  __syncthreads();
  // 1024 results, 1024 threads: element e = tid -> (row = e >> 4, col = e & 15); fixed summation order over the waves
  float out;
  {
    const int e = threadIdx.x, row = e >> 4, col = e & 15;
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < SK_WAVES; ++ww) v += red[ww][e];
    const int m = m0 + row, nn = n0 + col;
    const bool ok = m < a.M && nn < a.N;
    if (ok) {
      if (a.bias) v += a.bias[nn];
      if (a.epi == 1) v = tanhf(v);
      else if (a.epi == 2) { const float t = a.aux[(long)m * a.ldaux + nn]; v *= 1.f - t * t; }
      else if (a.epi == 3) v *= dgelu_f(a.aux[(long)m * a.ldaux + nn]);
      a.C[(long)m * a.ldc + nn] = v;
    } else v = 0.f;
    out = v;
  }
  if (a.colsum) {
    __syncthreads();
    red[0][threadIdx.x] = out;
    __syncthreads();
    if (threadIdx.x < SK_COLS && n0 + (int)threadIdx.x < a.N && blockIdx.y == 0) {
      float s = 0.f;
      for (int row = 0; row < SK_ROWS; ++row) s += red[0][row * SK_COLS + threadIdx.x];
      float* cp = a.colsum + n0 + threadIdx.x;
      *cp = (a.cs_beta != 0.f ? a.cs_beta * *cp : 0.f) + s;
    }
  }
}

static int g_skinny_probe = 0;
void climb_skinny_set_probe(int v) { g_skinny_probe = v; }
// C[m, n] (ldc) = epi(sum_k A[m * lda + k] * B[n * sbn + k * sbk] + bias[n]);  optional column sums of C and of A (see SkinnyArgs).
// colsum / acol need M <= 64 (one row tile holds every row); A rows and -- for sbk == 1 -- B rows must be 16-byte aligned (lda, sbn % 4 == 0).
extern "C" int climb_skinny_f32(const float* A, long lda, const float* B, long sbn, long sbk, float* C, long ldc, int M, int N, int K, const float* bias,
                                int epi, const float* aux, long ldaux, float* colsum, float colsum_beta, float* acol, float acol_beta, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return CLIMB_EINVAL;
  if (epi < 0 || epi > 3 || ((epi == 2 || epi == 3) && !aux)) return CLIMB_EINVAL;
  if ((colsum || acol) && M > SK_ROWS) return CLIMB_EUNSUPPORTED;
  if ((lda & 3) || (((uintptr_t)A) & 15)) return CLIMB_EUNSUPPORTED;
  const bool kcontig = sbk == 1 && (sbn & 3) == 0 && (((uintptr_t)B) & 15) == 0;
  SkinnyArgs a{A, lda, B, sbn, sbk, C, ldc, M, N, K, bias, epi, aux, ldaux, colsum, colsum_beta, acol, acol_beta};
  dim3 grid((N + SK_COLS - 1) / SK_COLS, (M + SK_ROWS - 1) / SK_ROWS);
#define SK_LAUNCH(KC, P) hipLaunchKernelGGL((skinny_f32_kernel<KC, P>), grid, dim3(64 * SK_WAVES), 0, (hipStream_t)stream, a)
  if (g_skinny_probe == 1) { if (kcontig) SK_LAUNCH(true, 1); else SK_LAUNCH(false, 1); }          // measurement builds (climb_set_option 19): results are wrong
  else if (g_skinny_probe == 2) { if (kcontig) SK_LAUNCH(true, 2); else SK_LAUNCH(false, 2); }
  else if (kcontig) SK_LAUNCH(true, 0);
  else SK_LAUNCH(false, 0);
#undef SK_LAUNCH
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// head LayerNorm + GELU in one pass (REF/modeling/vilt.py:191-193: Linear -> LayerNorm(eps 1e-5) -> GELU -> Linear): zn = LN(z) (kept: the backward's
// gelu' argument) and gz = gelu(zn) (the next product's operand).  One wave per row, the row in registers, statistics as layernorm_fwd_kernel's.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_gelu_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float eps, float* __restrict__ zn, float* __restrict__ gz, long ldy, float* __restrict__ mean_out,
                                                                 float* __restrict__ rstd_out, int M, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (long)row * ldx;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = (c < C) ? ld4(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += v[i].x + v[i].y + v[i].z + v[i].w;
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += a * a + b * b + cc * cc + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      const float4 g = ld4(gamma + c), b = ld4(beta + c);
      const float4 o = make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y, (v[i].z - mean) * rstd * g.z + b.z,
                                   (v[i].w - mean) * rstd * g.w + b.w);
      st4(zn + (long)row * ldy + c, o);
      st4(gz + (long)row * ldy + c, make_float4(gelu_f(o.x), gelu_f(o.y), gelu_f(o.z), gelu_f(o.w)));
    }
  }
}
extern "C" int climb_layernorm_gelu_fwd(const float* x, long ldx, const float* gamma, const float* beta, float eps, float* zn, float* gz, long ldy, float* mean,
                                        float* rstd, int M, int C, void* stream) {
  if (M <= 0 || C <= 0 || (C & 3) || !mean || !rstd) return CLIMB_EINVAL;
  dim3 grid((M + 3) / 4), blk(256);
  if (C <= 768) hipLaunchKernelGGL((layernorm_gelu_fwd_kernel<3>), grid, blk, 0, (hipStream_t)stream, x, ldx, gamma, beta, eps, zn, gz, ldy, mean, rstd, M, C);
  else if (C <= 1536) hipLaunchKernelGGL((layernorm_gelu_fwd_kernel<6>), grid, blk, 0, (hipStream_t)stream, x, ldx, gamma, beta, eps, zn, gz, ldy, mean, rstd, M, C);
  else return CLIMB_EUNSUPPORTED;
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// Rank-M weight-gradient updates of the pooler and the heads (M = batch rows): C[n, k] += sum_m dY[m, n] * X[m, k].  climb_gemm_f32 walks the
// 64 rows in four 16-deep trips (load, LDS, barrier, MFMA); here a 64 x 64 tile's operands go global -> registers in ONE round trip (for a fixed m
// both operands are contiguous along the lanes: coalesced 128-byte loads, 32 + 32 dwords per lane for 64 rows) and straight into
// v_mfma_f32_32x32x2_f32; the tile is then read-modified-written once.  Exact fp32, one fixed summation order.  Measured in the step
// (tools/head_trace.sh): 28.9 / 15.0 / 11.8 us for the 3129 x 1536, 1536 x 768 and 768 x 768 updates against 28.7 / 18.0 / 15.5 -- the big one is
// bound by its 38 MB read-modify-write at 1176 small workgroups, not by the trips.
__global__ __launch_bounds__(256) void rank_update_f32_kernel(const float* __restrict__ dY, long lddy, const float* __restrict__ X, long ldx, float* __restrict__ C,
                                                               long ldc, int M, int N, int K) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.y * 64 + (w >> 1) * 32, k0 = blockIdx.x * 64 + (w & 1) * 32;
  const int n = n0 + l31, k = k0 + l31;
  const bool nok = n < N, kok = k < K;
  const float* yp = dY + (nok ? n : 0);
  const float* xp = X + (kok ? k : 0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int m0 = 0; m0 < M; m0 += 64) {
    float a[32], b[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int m = m0 + 2 * i + half;
      const int mc = m < M ? m : M - 1;
      a[i] = yp[(long)mc * lddy];
      b[i] = xp[(long)mc * ldx];
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const bool mok = m0 + 2 * i + half < M;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32((mok && nok) ? a[i] : 0.f, (mok && kok) ? b[i] : 0.f, acc, 0, 0, 0);
    }
  }
  if (!kok) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int nn = n0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (nn < N) C[(long)nn * ldc + k] += acc[r];
  }
}
extern "C" int climb_rank_update_f32(const float* dY, long lddy, const float* X, long ldx, float* C, long ldc, int M, int N, int K, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !dY || !X || !C) return CLIMB_EINVAL;
  hipLaunchKernelGGL(rank_update_f32_kernel, dim3((K + 63) / 64, (N + 63) / 64), dim3(256), 0, (hipStream_t)stream, dY, lddy, X, ldx, C, ldc, M, N, K);
  LAUNCH_CHECK();
  return CLIMB_OK;
}
