// HBM-bound row-wise kernels of the ViLT step: LayerNorm fwd/bwd, embeddings, im2col, column sums,
// activations and losses.  One wave (64 lanes) owns one row; each lane keeps NV float4 of it in registers,
// so every row is read from HBM exactly once per kernel.  gfx950 only.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// LayerNorm forward:  y = (x - mean) * rstd * gamma + beta      (HF modeling_vilt.py:430-451 uses it twice
// per layer with eps=1e-12; REF/modeling/vilt.py:192 head LN with eps=1e-5)
// x is the fp32 residual stream; y has the GEMM operand type TO.  Optional `add` [C] is added after the affine
// (text rows add the modality-type embedding, HF:208-210).
// RPW rows per wave (r04 measurement, climb_set_option 21; tools/ln_bench.py): with all RPW rows' loads issued before the first reduction a launch of
// M = 12288 rows is 1536 (RPW = 2) or 1024 (3) instead of 3072 workgroups -- 0.75 / 0.5 instead of 1.5 rounds of the chip's workgroup slots.  Measured
// at the step's shape: 11.8 us (4.8 TB/s) with one row per wave, 12.3 with two, 13.1 with three: the default stays 1 (more waves in flight beat fewer
// rounds; the in-step 13.5 us are cold caches, not the launch shape)
template <typename TO, int NV, int RPW>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, TO* __restrict__ y, long ldy,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out, int M, int C, long ylo = 0) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= M) return;
  float4 v[RPW][NV];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int row = row0 + r < M ? row0 + r : M - 1;
    const float* xr = x + (long)row * ldx;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = (i * 64 + lane) * 4;
      v[r][i] = (c < C) ? ld4(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int row = row0 + r;
    if (row >= M) break;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[r][i].x + v[r][i].y + v[r][i].z + v[r][i].w;
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = (i * 64 + lane) * 4;
      if (c < C) {
        float a = v[r][i].x - mean, b = v[r][i].y - mean, cc = v[r][i].z - mean, d = v[r][i].w - mean;
        q += a * a + b * b + cc * cc + d * d;
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
    TO* yr = y + (long)row * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = (i * 64 + lane) * 4;
      if (c < C) {
        float4 g = ld4(gamma + c), b = ld4(beta + c);
        float4 o = make_float4((v[r][i].x - mean) * rstd * g.x + b.x, (v[r][i].y - mean) * rstd * g.y + b.y,
                               (v[r][i].z - mean) * rstd * g.z + b.z, (v[r][i].w - mean) * rstd * g.w + b.w);
        st4x(yr + c, ylo, o);          // (ylo: the lo plane of a split output, r06)
      }
    }
  }
}

static int g_ln_rpw = 1;
void climb_ln_set_rpw(int v) { g_ln_rpw = v; }
template <typename TO>
static int layernorm_fwd_launch(const float* x, long ldx, const float* g, const float* b, float eps, TO* y, long ldy, float* mean,
                                float* rstd, int M, int C, hipStream_t st, long ylo = 0) {
  if (C % 4 || M <= 0) return CLIMB_EINVAL;
  const int rpw = (g_ln_rpw >= 2 && M >= 4096) ? g_ln_rpw : 1;          // the small launches (B rows of the pooler / heads) keep one row per wave
  dim3 grid((M + 4 * rpw - 1) / (4 * rpw)), blk(256);
#define LNF(NV_, R_) hipLaunchKernelGGL((layernorm_fwd_kernel<TO, NV_, R_>), grid, blk, 0, st, x, ldx, g, b, eps, y, ldy, mean, rstd, M, C, ylo)
  if (C <= 768) { if (rpw == 2) LNF(3, 2); else if (rpw >= 3) LNF(3, 3); else LNF(3, 1); }
  else if (C <= 1536) { if (rpw == 2) LNF(6, 2); else LNF(6, 1); }
  else return CLIMB_EUNSUPPORTED;
#undef LNF
  LAUNCH_CHECK();
  return CLIMB_OK;
}

extern "C" int climb_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, float eps, void* y, long ldy,
                                   int y_dtype, float* mean, float* rstd, int M, int C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (y_dtype == CLIMB_DT_F32) return layernorm_fwd_launch<float>(x, ldx, gamma, beta, eps, (float*)y, ldy, mean, rstd, M, C, st);
  if (y_dtype == CLIMB_DT_BF16) return layernorm_fwd_launch<bf16_t>(x, ldx, gamma, beta, eps, (bf16_t*)y, ldy, mean, rstd, M, C, st);
  if (y_dtype == CLIMB_DT_SPLIT) return layernorm_fwd_launch<sp16_t>(x, ldx, gamma, beta, eps, (sp16_t*)y, ldy, mean, rstd, M, C, st, (long)M * ldy);
  return CLIMB_EINVAL;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward fused with the residual-gradient stream:
//   dxo = dres_in + rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = dy * gamma, xhat = (x - mean) * rstd
// Writes dxo (fp32, may alias dres_in), an optional cast of it (next GEMM operand), and per-block partial column
// sums  part[blk][0]=dgamma, [1]=dbeta, [2]=colsum(dxo)  (reduced deterministically by climb_colreduce).
// rows per block = LNB_WAVES waves x 4 rows.  r03: 8 waves (512 threads) instead of 4 -- the same waves per CU, half the partial-sum rows:
// 3.5 MB less written per launch and half of what the 24 launches of a step leave for the batched column reduction to read (170 -> 85 MB)
#define LNB_WAVES 8
#define LNB_ROWS (4 * LNB_WAVES)
template <typename TI, typename TO, int NV>
__global__ __launch_bounds__(64 * LNB_WAVES) void layernorm_bwd_kernel(const TI* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* dres_in, long ldr,
                                                            float* dxo, long ldo, TO* __restrict__ dcast, long ldc,
                                                            float* __restrict__ part, int M, int C, long clo = 0) {
  __shared__ __attribute__((aligned(16))) float red[LNB_WAVES][NV * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float4 ag[NV], ab[NV], as[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) ag[i] = ab[i] = as[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gm[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int c = (i * 64 + lane) * 4;
    gm[i] = (c < C) ? ld4(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int rr = 0; rr < 4; ++rr) {
    const int row = blockIdx.x * LNB_ROWS + rr * LNB_WAVES + wid;
    if (row >= M) break;
    const float mu = mean[row], rs = rstd[row];
    float4 xh[NV], g[NV], d[NV], rin[NV];
    float s1 = 0.f, s2 = 0.f;
    // all three input streams of the row are requested before the first reduction (the residual gradient is only needed after
    // it, but its latency then hides under the row statistics instead of following them)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = (i * 64 + lane) * 4;
      rin[i] = (dres_in && c < C) ? ld4(dres_in + (long)row * ldr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = (i * 64 + lane) * 4;
      if (c < C) {
        float4 xv = ld4(x + (long)row * ldx + c);
        d[i] = ld4(dy + (long)row * lddy + c);
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        g[i] = make_float4(d[i].x * gm[i].x, d[i].y * gm[i].y, d[i].z * gm[i].z, d[i].w * gm[i].w);
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
      } else {
        xh[i] = g[i] = d[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = (i * 64 + lane) * 4;
      if (c < C) {
        const float4 r = rin[i];
        float4 o = make_float4(r.x + rs * (g[i].x - m1 - xh[i].x * m2), r.y + rs * (g[i].y - m1 - xh[i].y * m2),
                               r.z + rs * (g[i].z - m1 - xh[i].z * m2), r.w + rs * (g[i].w - m1 - xh[i].w * m2));
        st4(dxo + (long)row * ldo + c, o);
        if (dcast) st4x(dcast + (long)row * ldc + c, clo, o);
        ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
        ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
        as[i].x += o.x; as[i].y += o.y; as[i].z += o.z; as[i].w += o.w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (k) __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int c = (i * 64 + lane) * 4;
      *reinterpret_cast<float4*>(&red[wid][c]) = (k == 0 ? ag[i] : (k == 1 ? ab[i] : as[i]));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 64 * LNB_WAVES) {
      float t = red[0][c];
#pragma unroll
      for (int w = 1; w < LNB_WAVES; ++w) t += red[w][c];
      part[((long)blockIdx.x * 3 + k) * C + c] = t;
    }
  }
}

template <typename TI, typename TO>
static int layernorm_bwd_launch(const TI* dy, long lddy, const float* x, long ldx, const float* mean, const float* rstd, const float* gamma,
                                const float* dres_in, long ldr, float* dxo, long ldo, TO* dcast, long ldc, float* part, int M, int C,
                                hipStream_t st, long clo = 0) {
  if (C % 4 || M <= 0) return CLIMB_EINVAL;
  dim3 grid((M + LNB_ROWS - 1) / LNB_ROWS), blk(64 * LNB_WAVES);
  if (C <= 768)
    hipLaunchKernelGGL((layernorm_bwd_kernel<TI, TO, 3>), grid, blk, 0, st, dy, lddy, x, ldx, mean, rstd, gamma, dres_in, ldr, dxo, ldo, dcast, ldc, part, M, C, clo);
  else if (C <= 1536)
    hipLaunchKernelGGL((layernorm_bwd_kernel<TI, TO, 6>), grid, blk, 0, st, dy, lddy, x, ldx, mean, rstd, gamma, dres_in, ldr, dxo, ldo, dcast, ldc, part, M, C, clo);
  else return CLIMB_EUNSUPPORTED;
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// part must hold ceil(M/32)*3*C floats.  dtype applies to dy and dcast (CLIMB_DT_SPLIT: dy fp32, dcast a (hi, lo) plane pair [2][M][ldc]).
extern "C" int climb_layernorm_bwd(const void* dy, long lddy, int dtype, const float* x, long ldx, const float* mean, const float* rstd,
                                   const float* gamma, const float* dres_in, long ldr, float* dxo, long ldo, void* dcast, long ldc,
                                   float* part, int M, int C, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CLIMB_DT_F32)
    return layernorm_bwd_launch<float, float>((const float*)dy, lddy, x, ldx, mean, rstd, gamma, dres_in, ldr, dxo, ldo, (float*)dcast, ldc, part, M, C, st);
  if (dtype == CLIMB_DT_BF16)
    return layernorm_bwd_launch<bf16_t, bf16_t>((const bf16_t*)dy, lddy, x, ldx, mean, rstd, gamma, dres_in, ldr, dxo, ldo, (bf16_t*)dcast, ldc, part, M, C, st);
  if (dtype == CLIMB_DT_SPLIT)
    return layernorm_bwd_launch<float, sp16_t>((const float*)dy, lddy, x, ldx, mean, rstd, gamma, dres_in, ldr, dxo, ldo, (sp16_t*)dcast, ldc, part, M, C, st, (long)M * ldc);
  return CLIMB_EINVAL;
}
extern "C" int climb_layernorm_bwd_rows_per_block() { return LNB_ROWS; }

// out[c] = beta*out[c] + sum_b part[b*stride + c]     (deterministic second stage of every column reduction)
// block = 32 columns x 8 row-groups: each thread sums nblk/8 partial rows (independent loads), LDS tree over the groups.
// grid.y selects one of up to 3 (column offset -> output) pairs so the {dgamma, dbeta, colsum} triple is ONE launch.
struct ColOut { float* out[3]; long col_off[3]; };
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ part, long stride, int nblk, ColOut o, int ncols, float beta) {
  __shared__ float red[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  float* out = o.out[blockIdx.y];
  if (!out) return;
  const float* src = part + o.col_off[blockIdx.y];
  // 8 independent loads in flight per thread: the reduction is latency-bound (few hundred partial rows, tiny grid)
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  if (c < ncols) {
    int b = ry;
    for (; b + 56 < nblk; b += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += src[(long)(b + 8 * u) * stride + c];
    }
    for (; b < nblk; b += 8) s[0] += src[(long)b * stride + c];
  }
  red[ry][cx] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (ry == 0 && c < ncols) {
    float s = ((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx])) + ((red[4][cx] + red[5][cx]) + (red[6][cx] + red[7][cx]));
    out[c] = (beta != 0.f ? beta * out[c] : 0.f) + s;
  }
}
extern "C" int climb_colreduce(const float* part, long stride, int nblk, float* out, int ncols, float beta, void* stream) {
  if (ncols <= 0 || nblk <= 0) return CLIMB_EINVAL;
  ColOut o;
  o.out[0] = out; o.out[1] = o.out[2] = nullptr;
  o.col_off[0] = o.col_off[1] = o.col_off[2] = 0;
  hipLaunchKernelGGL(colreduce_kernel, dim3((ncols + 31) / 32, 1), dim3(256), 0, (hipStream_t)stream, part, stride, nblk, o, ncols, beta);
  LAUNCH_CHECK();
  return CLIMB_OK;
}
// three reductions over the same partial buffer: out_k[c] = beta*out_k[c] + sum_b part[b*stride + k*ncols + c]; NULL outputs are skipped
extern "C" int climb_colreduce3(const float* part, long stride, int nblk, float* out0, float* out1, float* out2, int ncols, float beta, void* stream) {
  if (ncols <= 0 || nblk <= 0) return CLIMB_EINVAL;
  if (!out0 && !out1 && !out2) return CLIMB_OK;
  ColOut o;
  o.out[0] = out0; o.out[1] = out1; o.out[2] = out2;
  o.col_off[0] = 0; o.col_off[1] = ncols; o.col_off[2] = 2L * ncols;
  hipLaunchKernelGGL(colreduce_kernel, dim3((ncols + 31) / 32, 3), dim3(256), 0, (hipStream_t)stream, part, stride, nblk, o, ncols, beta);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// Many {dgamma, dbeta, colsum} triples in ONE launch (r03): the 24 LayerNorm backwards of a step each leave a partial buffer; reducing
// them one launch at a time was 31 launches of ~6.5 us for ~2 us of work each.  `segs` is a device table (built once per shape by the host):
// 48-byte records { const float* part; long stride; float* out[3]; int nblk; int ncols }, out_k[c] += sum_b part[b*stride + k*ncols + c]
// (NULL outputs skipped).  grid = (ceil(max ncols / 32), 3, segments).
struct ColSeg { const float* part; long stride; float* out[3]; int nblk, ncols; };
__global__ __launch_bounds__(256) void colreduce_batched_kernel(const ColSeg* __restrict__ segs) {
  __shared__ float red[8][33];
  const ColSeg sg = segs[blockIdx.z];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx, ncols = sg.ncols, nblk = sg.nblk;
  float* out = sg.out[blockIdx.y];
  if (!out || blockIdx.x * 32 >= ncols) return;
  const float* src = sg.part + (long)blockIdx.y * ncols;
  const long stride = sg.stride;
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  if (c < ncols) {
    int b = ry;
    for (; b + 56 < nblk; b += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += src[(long)(b + 8 * u) * stride + c];
    }
    for (; b < nblk; b += 8) s[0] += src[(long)b * stride + c];
  }
  red[ry][cx] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (ry == 0 && c < ncols) out[c] += ((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx])) + ((red[4][cx] + red[5][cx]) + (red[6][cx] + red[7][cx]));
}
extern "C" int climb_colreduce_batched(const void* segs, int nseg, int max_cols, void* stream) {
  if (!segs || nseg <= 0 || max_cols <= 0) return CLIMB_EINVAL;
  hipLaunchKernelGGL(colreduce_batched_kernel, dim3((max_cols + 31) / 32, 3, nseg), dim3(256), 0, (hipStream_t)stream, (const ColSeg*)segs);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// ------------------------------------------------------------------------------------------------
// Column sums of a [M,C] activation (bias gradients) -> part[rowblk][C]; optional fp32->TO cast of the input.
#define CS_ROWS 64
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void colsum_kernel(const TI* __restrict__ x, long ldx, TO* __restrict__ cast, long ldc, float* __restrict__ part,
                                                     int M, int C) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  const int r0 = blockIdx.y * CS_ROWS;
  const int r1 = min(M, r0 + CS_ROWS);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = r0; r < r1; ++r) {
    float4 v = ld4(x + (long)r * ldx + c);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    if (cast) st4(cast + (long)r * ldc + c, v);
  }
  *reinterpret_cast<float4*>(part + (long)blockIdx.y * C + c) = s;
}
extern "C" int climb_colsum_rows_per_block() { return CS_ROWS; }
// x dtype in_dtype; optional cast output (bf16) only meaningful for f32 input.  part: ceil(M/64)*C floats.
extern "C" int climb_colsum(const void* x, long ldx, int in_dtype, void* cast_bf16, long ldc, float* part, int M, int C, void* stream) {
  if (C % 4 || M <= 0) return CLIMB_EINVAL;
  dim3 grid((C / 4 + 255) / 256, (M + CS_ROWS - 1) / CS_ROWS), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == CLIMB_DT_F32)
    hipLaunchKernelGGL((colsum_kernel<float, bf16_t>), grid, blk, 0, st, (const float*)x, ldx, (bf16_t*)cast_bf16, ldc, part, M, C);
  else if (in_dtype == CLIMB_DT_BF16)
    hipLaunchKernelGGL((colsum_kernel<bf16_t, bf16_t>), grid, blk, 0, st, (const bf16_t*)x, ldx, (bf16_t*)nullptr, 0L, part, M, C);
  else return CLIMB_EINVAL;
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// ------------------------------------------------------------------------------------------------
// Text embeddings (HF:237-269 TextEmbeddings + HF:208-210 modality add):
//   x[b, t, :] = LN(word[ids[b,t]] + type[tt[b,t]] + pos[t]) * gamma + beta + modality[0]
// One wave per token row; H <= 768.
__global__ __launch_bounds__(256) void embed_text_fwd_kernel(const long* __restrict__ ids, const long* __restrict__ tts,
                                                             const float* __restrict__ word, const float* __restrict__ type,
                                                             const float* __restrict__ pos, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ mod0, float eps,
                                                             float* __restrict__ x, int B, int T, int S_pad, int H,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out, int emb_ld) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * T) return;
  const int b = row / T, t = row - b * T;
  const long tt = tts[row];
  // ids == NULL: `word` is an inputs_embeds matrix [B, emb_ld, H] (HF ViltEmbeddings / TextEmbeddings with inputs_embeds: ViLT-BERT)
  const float* wrow = ids ? word + ids[row] * H : word + ((long)b * emb_ld + t) * H;
  float4 v[3];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int c = (i * 64 + lane) * 4;
    if (c < H) {
      float4 w = ld4(wrow + c), ty = ld4(type + tt * H + c), p = ld4(pos + (long)t * H + c);
      v[i] = make_float4(w.x + ty.x + p.x, w.y + ty.y + p.y, w.z + ty.z + p.z, w.w + ty.w + p.w);
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    } else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int c = (i * 64 + lane) * 4;
    if (c < H) {
      float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += a * a + bb * bb + cc * cc + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  float* xr = x + ((long)b * S_pad + t) * H;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int c = (i * 64 + lane) * 4;
    if (c < H) {
      float4 g = ld4(gamma + c), be = ld4(beta + c), m = ld4(mod0 + c);
      st4(xr + c, make_float4((v[i].x - mean) * rstd * g.x + be.x + m.x, (v[i].y - mean) * rstd * g.y + be.y + m.y,
                              (v[i].z - mean) * rstd * g.z + be.z + m.z, (v[i].w - mean) * rstd * g.w + be.w + m.w));
    }
  }
}
extern "C" int climb_embed_text_fwd(const long* ids, const long* tts, const float* word, const float* type, const float* pos,
                                    const float* gamma, const float* beta, const float* mod0, float eps, float* x, int B, int T, int S_pad,
                                    int H, float* mean, float* rstd, int emb_ld, void* stream) {
  if (H % 4 || H > 768) return CLIMB_EUNSUPPORTED;
  if (!ids && emb_ld < T) return CLIMB_EINVAL;
  hipLaunchKernelGGL(embed_text_fwd_kernel, dim3((B * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, tts, word, type, pos, gamma, beta,
                     mod0, eps, x, B, T, S_pad, H, mean, rstd, emb_ld);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// Backward of the above.  dres rows [b, t, :] hold d(x).  Writes the pre-LayerNorm gradient dpre [B*T, H] (summed into the
// position / token-type tables by embed_text_bwd_tables_kernel: every row hits the same few table rows, so atomics there
// would serialise), scatter-adds into the word table, and per-block partials part[blk][0]=dgamma, [1]=dbeta, [2]=dmodality0.
// rows per block: 2560 text rows are 80 blocks of 32 (8 dependent row iterations per wave on a third of the chip: 44 us) or 320 blocks of 8 (r03: 36 us)
#define ETB_ROWS 8
__global__ __launch_bounds__(256) void embed_text_bwd_kernel(const long* __restrict__ ids, const long* __restrict__ tts,
                                                             const float* __restrict__ word, const float* __restrict__ type,
                                                             const float* __restrict__ pos, const float* __restrict__ gamma,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ dres, int B, int T, int S_pad, int H,
                                                             float* dword, float* __restrict__ dpre, float* __restrict__ part, int emb_ld) {
  __shared__ __attribute__((aligned(16))) float red[3][4][768];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float4 ag[3], ab[3], am[3], gm[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ag[i] = ab[i] = am[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    int c = (i * 64 + lane) * 4;
    gm[i] = (c < H) ? ld4(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int rr = 0; rr < ETB_ROWS / 4; ++rr) {
    const int row = blockIdx.x * ETB_ROWS + rr * 4 + wid;
    if (row >= B * T) break;
    const int b = row / T, t = row - b * T;
    const long id = ids ? ids[row] : 0, tt = tts[row];
    const float* wrow = ids ? word + id * H : word + ((long)b * emb_ld + t) * H;
    const float mu = mean[row], rs = rstd[row];
    float4 xh[3], g[3], d[3];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int c = (i * 64 + lane) * 4;
      if (c < H) {
        float4 w = ld4(wrow + c), ty = ld4(type + tt * H + c), p = ld4(pos + (long)t * H + c);
        d[i] = ld4(dres + ((long)b * S_pad + t) * H + c);
        xh[i] = make_float4((w.x + ty.x + p.x - mu) * rs, (w.y + ty.y + p.y - mu) * rs, (w.z + ty.z + p.z - mu) * rs,
                            (w.w + ty.w + p.w - mu) * rs);
        g[i] = make_float4(d[i].x * gm[i].x, d[i].y * gm[i].y, d[i].z * gm[i].z, d[i].w * gm[i].w);
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
      } else xh[i] = g[i] = d[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int c = (i * 64 + lane) * 4;
      if (c < H) {
        float o[4] = {rs * (g[i].x - m1 - xh[i].x * m2), rs * (g[i].y - m1 - xh[i].y * m2), rs * (g[i].z - m1 - xh[i].z * m2),
                      rs * (g[i].w - m1 - xh[i].w * m2)};
        if (dpre) st4(dpre + (long)row * H + c, make_float4(o[0], o[1], o[2], o[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (dword && ids) atomicAdd(dword + id * H + c + j, o[j]);   // random vocabulary rows: low contention (no table behind inputs_embeds)
        ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
        ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
        am[i].x += d[i].x; am[i].y += d[i].y; am[i].z += d[i].z; am[i].w += d[i].w;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int c = (i * 64 + lane) * 4;
    *reinterpret_cast<float4*>(&red[0][wid][c]) = ag[i];
    *reinterpret_cast<float4*>(&red[1][wid][c]) = ab[i];
    *reinterpret_cast<float4*>(&red[2][wid][c]) = am[i];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 3 * H; idx += 256) {
    int k = idx / H, c = idx - k * H;
    part[((long)blockIdx.x * 3 + k) * H + c] = red[k][0][c] + red[k][1][c] + red[k][2][c] + red[k][3][c];
  }
}
// dpos[t,:] += sum_b dpre[b,t,:];  dtype[k,:] += sum_{(b,t): tt==k} dpre[b,t,:]  (k < 2).  One thread per (t, 4 columns).
__global__ void embed_text_bwd_tables_kernel(const float* __restrict__ dpre, const long* __restrict__ tts, float* dpos, float* dtype_, int B, int T,
                                             int H, float* __restrict__ part) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = H / 4;
  if (i >= T * per_row) return;
  const int t = i / per_row, c = (i - t * per_row) * 4;
  float4 tot = make_float4(0.f, 0.f, 0.f, 0.f), t1 = tot;
  for (int b = 0; b < B; ++b) {
    float4 v = ld4(dpre + ((long)b * T + t) * H + c);
    tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
    if (tts[b * T + t] != 0) { t1.x += v.x; t1.y += v.y; t1.z += v.z; t1.w += v.w; }
  }
  if (dpos) {
    float4 o = ld4(dpos + (long)t * H + c);
    st4(dpos + (long)t * H + c, make_float4(o.x + tot.x, o.y + tot.y, o.z + tot.z, o.w + tot.w));
  }
  // per-t partials of the two token-type rows: part[t][0][:] (type 0), part[t][1][:] (type 1)
  st4(part + ((long)t * 2 + 0) * H + c, make_float4(tot.x - t1.x, tot.y - t1.y, tot.z - t1.z, tot.w - t1.w));
  st4(part + ((long)t * 2 + 1) * H + c, t1);
}
extern "C" int climb_embed_text_bwd_rows_per_block() { return ETB_ROWS; }
// part: ceil(B*T/ETB_ROWS)*3*H floats; dpre: B*T*H floats; part2: T*2*H floats (token-type partials, reduce with stride 2H over T)
extern "C" int climb_embed_text_bwd(const long* ids, const long* tts, const float* word, const float* type, const float* pos,
                                    const float* gamma, const float* mean, const float* rstd, const float* dres, int B, int T, int S_pad, int H,
                                    float* dword, float* dpos, float* dpre, float* part, float* part2, int emb_ld, void* stream) {
  if (H != 768) return CLIMB_EUNSUPPORTED;
  if (!ids && emb_ld < T) return CLIMB_EINVAL;
  hipLaunchKernelGGL(embed_text_bwd_kernel, dim3((B * T + ETB_ROWS - 1) / ETB_ROWS), dim3(256), 0, (hipStream_t)stream, ids, tts, word, type, pos, gamma, mean,
                     rstd, dres, B, T, S_pad, H, dword, dpre, part, emb_ld);
  LAUNCH_CHECK();
  if (dpre) {
    int n = T * (H / 4);
    hipLaunchKernelGGL(embed_text_bwd_tables_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, dpre, tts, dpos, (float*)nullptr, B, T, H, part2);
    LAUNCH_CHECK();
  }
  return CLIMB_OK;
}

// ------------------------------------------------------------------------------------------------
// Patch projection operand (HF:292-300 Conv2d(3,768,k=32,s=32) == GEMM after im2col):
//   out[(b*NP + py*gw + px), c*P*P + ky*P + kx] = pixel[b, c, py*P+ky, px*P+kx]
template <typename TO>
__global__ void im2col_kernel(const float* __restrict__ px, TO* __restrict__ out, int B, int Cc, int Hh, int Ww, int P) {
  const long n4 = (long)B * Cc * Hh * Ww / 4;
  const int gw = Ww / P, gh = Hh / P;
  const int K = Cc * P * P;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    long e = i * 4;
    int xw = (int)(e % Ww); long r = e / Ww;
    int yh = (int)(r % Hh); r /= Hh;
    int c = (int)(r % Cc); int b = (int)(r / Cc);
    int pxi = xw / P, kx = xw - pxi * P, pyi = yh / P, ky = yh - pyi * P;
    float4 v = ld4(px + e);
    st4(out + ((long)(b * gh * gw + pyi * gw + pxi)) * K + c * P * P + ky * P + kx, v);
  }
}
extern "C" int climb_im2col(const float* pixels, void* out, int out_dtype, int B, int C, int H, int W, int P, void* stream) {
  if (P % 4 || H % P || W % P) return CLIMB_EINVAL;
  long n4 = (long)B * C * H * W / 4;
  int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == CLIMB_DT_F32) hipLaunchKernelGGL((im2col_kernel<float>), dim3(grid), dim3(256), 0, st, pixels, (float*)out, B, C, H, W, P);
  else if (out_dtype == CLIMB_DT_BF16) hipLaunchKernelGGL((im2col_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, pixels, (bf16_t*)out, B, C, H, W, P);
  else return CLIMB_EINVAL;
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// Per-sample valid patch extent of a padded, variable-resolution batch (HF:96-99): the processor puts every image top-left
// on the batch canvas, so h = #valid rows in patch column 0, w = #valid columns in patch row 0 (mask read at each patch's
// top-left pixel = nearest down-sampling).  dims[b] = {h, w}.
__global__ void patch_grid_dims_kernel(const long* __restrict__ mask, int Hh, int Ww, int P, int* __restrict__ dims) {
  const int b = blockIdx.x;
  const long* m = mask + (long)b * Hh * Ww;
  if (threadIdx.x == 0) {
    int h = 0, w = 0;
    for (int py = 0; py < Hh / P; ++py) h += m[(long)py * P * Ww] != 0;
    for (int px = 0; px < Ww / P; ++px) w += m[px * P] != 0;
    dims[2 * b] = h;
    dims[2 * b + 1] = w;
  }
}
extern "C" int climb_patch_grid_dims(const long* pixel_mask, int B, int H, int W, int P, int* dims, void* stream) {
  if (B <= 0 || H % P || W % P) return CLIMB_EINVAL;
  hipLaunchKernelGGL(patch_grid_dims_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, pixel_mask, H, W, P, dims);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// bilinear (align_corners=True) source taps of output index `o` when a length-`in` axis is resized to `out` (HF:104-111 via
// torch.nn.functional.interpolate): src = o * (in-1)/(out-1)
__device__ __forceinline__ void bilinear_tap(int o, int out, int in, int& i0, int& i1, float& l1) {
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float src = scale * (float)o;
  i0 = (int)src;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

// Image rows of the embedding (HF:168-173 cls/pos add, HF:211-213 modality add, HF:216 concat):
//   x[b, T, :]       = cls + pos[0] + modality[type_b]
//   x[b, T+1+p, :]   = proj[b*NP + p, :] + pos(p) + modality[type_b]          p in raster order over the gh x gw canvas
//   x[b, S.., :]     = 0  (rows S..S_pad-1 are padding; masked as keys)
// dims == NULL: every image fills the canvas and the canvas is the table's own g0 x g0 grid (pos(p) = pos[1+p]).
// dims != NULL (variable resolution, HF:92-178): patch (py, px) of sample b is valid iff py < h_b and px < w_b; valid patches get
// the position table resized to (h_b, w_b) on the fly; invalid canvas patches become zero rows and are masked in key_bias
// (the reference keeps max_b(h*w) randomly ordered rows instead -- same pooled output, see oracle.visual_embed_general).
// compact != 0 (needs dims): the valid patches of sample b are packed, raster order over ITS OWN h_b x w_b grid, into rows
// T+1 .. T+h_b*w_b; the canvas only addresses `proj`.  A batch that mixes portrait and landscape images has a 20 x 20 canvas (400
// patches) but never more than 240 valid patches per image (processor rule: shorter edge 384, longer <= 640), so its sequences
// stay within the 288 rows the attention tiles are sized for -- like the reference, which keeps max_b(h_b*w_b) patch rows.
__global__ void assemble_image_kernel(const float* __restrict__ proj, const float* __restrict__ cls, const float* __restrict__ pos,
                                      const float* __restrict__ mod, const int* __restrict__ img_type, const int* __restrict__ dims,
                                      float* __restrict__ x, float* __restrict__ key_bias, int B, int T, int NP, int gw, int g0, int S_pad, int H,
                                      int compact) {
  const int rows = S_pad - T;
  const long n4 = (long)B * rows * H / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    long e = i * 4;
    int c = (int)(e % H); long r = e / H;
    int rr = (int)(r % rows); int b = (int)(r / rows);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    bool valid = compact ? true : rr <= NP;
    int prow = rr - 1;                                   // canvas index of this row's patch (row of `proj`)
    if (rr == 0) {
      float4 p = ld4(pos + c), m = ld4(mod + (long)img_type[b] * H + c), v = ld4(cls + c);
      o = make_float4(v.x + p.x + m.x, v.y + p.y + m.y, v.z + p.z + m.z, v.w + p.w + m.w);
    } else if (compact || rr <= NP) {
      float4 p;
      if (dims) {
        const int hb = dims[2 * b], wb = dims[2 * b + 1];
        const int sw = compact ? wb : gw;                // row stride of the patch raster inside the sequence
        const int py = (rr - 1) / sw, px = (rr - 1) - py * sw;
        valid = py < hb && px < wb;
        prow = py * gw + px;
        if (valid) {
          int y0, y1, x0, x1; float ly, lx;
          bilinear_tap(py, hb, g0, y0, y1, ly);
          bilinear_tap(px, wb, g0, x0, x1, lx);
          const float4 a00 = ld4(pos + (long)(1 + y0 * g0 + x0) * H + c), a01 = ld4(pos + (long)(1 + y0 * g0 + x1) * H + c);
          const float4 a10 = ld4(pos + (long)(1 + y1 * g0 + x0) * H + c), a11 = ld4(pos + (long)(1 + y1 * g0 + x1) * H + c);
          const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
          p = make_float4(w00 * a00.x + w01 * a01.x + w10 * a10.x + w11 * a11.x, w00 * a00.y + w01 * a01.y + w10 * a10.y + w11 * a11.y,
                          w00 * a00.z + w01 * a01.z + w10 * a10.z + w11 * a11.z, w00 * a00.w + w01 * a01.w + w10 * a10.w + w11 * a11.w);
        }
      } else {
        p = ld4(pos + (long)rr * H + c);
      }
      if (valid) {
        float4 m = ld4(mod + (long)img_type[b] * H + c), v = ld4(proj + ((long)b * NP + prow) * H + c);
        o = make_float4(v.x + p.x + m.x, v.y + p.y + m.y, v.z + p.z + m.z, v.w + p.w + m.w);
      }
    }
    st4(x + ((long)b * S_pad + T + rr) * H + c, o);
    if (dims && c == 0) key_bias[(long)b * S_pad + T + rr] = valid ? 0.f : -3.0e38f;
  }
}
extern "C" int climb_assemble_image(const float* proj, const float* cls, const float* pos, const float* mod, const int* img_type, const int* dims,
                                    float* x, float* key_bias, int B, int T, int NP, int gw, int g0, int S_pad, int H, int compact, void* stream) {
  if (H % 4 || (!compact && T + 1 + NP > S_pad) || (dims == nullptr && (NP != g0 * g0 || compact)) || (dims && !key_bias) || NP % gw || T + 1 > S_pad)
    return CLIMB_EINVAL;
  long n4 = (long)B * (S_pad - T) * H / 4;
  int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(assemble_image_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, proj, cls, pos, mod, img_type, dims, x, key_bias, B, T, NP,
                     gw, g0, S_pad, H, compact);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// Backward of the image rows.  One thread per (image row rr in [0,NP], 4 columns): loops over the batch.
//   dproj[b*NP+p, :] = cast(dres[b, T+1+p, :]) (0 for invalid canvas patches);  dcls += sum_b dres[b, T, :]
//   part[rr][type][:] = sum_{b: type_b == type, row valid} dres[b, T+rr, :]   (colreduce'd over rr into dmodality by the caller)
//   dims == NULL: dpos[rr, :] += sum_b dres[b, T+rr, :].   dims != NULL: only dpos[0] here; the patch rows' share of dpos is the
//   transpose of the bilinear resize, done by pos_interp_bwd_kernel.
template <typename TO>
__global__ void image_embed_bwd_kernel(const float* __restrict__ dres, const int* __restrict__ img_type, const int* __restrict__ dims,
                                       TO* __restrict__ dproj, float* dpos, float* dcls, float* __restrict__ part, int B, int T, int NP, int gw,
                                       int S_pad, int H, int ntypes, int compact) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = H / 4;
  if (i >= (NP + 1) * per_row) return;
  const int rr = i / per_row, c = (i - rr * per_row) * 4;
  const int py = rr > 0 ? (rr - 1) / gw : 0, px = rr > 0 ? (rr - 1) - py * gw : 0;
  float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 pt[3];
  pt[0] = pt[1] = pt[2] = make_float4(0.f, 0.f, 0.f, 0.f);
  // UB batch elements per round, every load requested before the first is used: the thread's sums stay in batch order (bit-identical), the
  // 64 dependent round trips of the plain loop become 8 (r03: 55 -> 45 us)
  constexpr int UB = 8;
  for (int b0 = 0; b0 < B; b0 += UB) {
    float4 v[UB];
    int ty[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int b = b0 + u;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      ty[u] = -1;
      if (b < B) {
        const bool valid = rr == 0 || dims == nullptr || (py < dims[2 * b] && px < dims[2 * b + 1]);
        const int srow = (rr > 0 && compact) ? 1 + py * dims[2 * b + 1] + px : rr;       // row of this canvas patch inside the sequence
        if (valid) v[u] = ld4(dres + ((long)b * S_pad + T + srow) * H + c);
        ty[u] = img_type[b];
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int b = b0 + u;
      if (b >= B) break;
      tot.x += v[u].x; tot.y += v[u].y; tot.z += v[u].z; tot.w += v[u].w;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (ty[u] == k) { pt[k].x += v[u].x; pt[k].y += v[u].y; pt[k].z += v[u].z; pt[k].w += v[u].w; }
      if (rr > 0 && dproj) st4(dproj + ((long)b * NP + rr - 1) * H + c, v[u]);
    }
  }
  if (dpos && (rr == 0 || dims == nullptr)) {
    float4 o = ld4(dpos + (long)rr * H + c);
    st4(dpos + (long)rr * H + c, make_float4(o.x + tot.x, o.y + tot.y, o.z + tot.z, o.w + tot.w));
  }
  if (rr == 0 && dcls) {
    float4 o = ld4(dcls + c);
    st4(dcls + c, make_float4(o.x + tot.x, o.y + tot.y, o.z + tot.z, o.w + tot.w));
  }
  for (int k = 0; k < ntypes && k < 3; ++k) st4(part + ((long)rr * ntypes + k) * H + c, pt[k]);
}
// Transpose of the per-sample bilinear resize: dpos[1 + ty*g0 + tx, :] += sum_b sum_{py<h_b} sum_{px<w_b} wy(py,ty) wx(px,tx) dres[b, T+1+py*gw+px, :].
// Gather form (no atomics on the table within a block): one thread per (table row ty, 4 columns) keeps the g0 entries of that row
// in registers and walks only the output rows whose taps touch ty; batch chunks (blockIdx.y) are combined with atomics.
#define PIB_G0 12
__global__ __launch_bounds__(192) void pos_interp_bwd_kernel(const float* __restrict__ dres, const int* __restrict__ dims, float* dpos, int B, int T,
                                                             int gw, int S_pad, int H, int bchunk, int compact) {
  const int ty = blockIdx.x, c = threadIdx.x * 4;
  if (c >= H) return;
  float4 acc[PIB_G0];
#pragma unroll
  for (int t = 0; t < PIB_G0; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
  for (int b = b0; b < b1; ++b) {
    const int hb = dims[2 * b], wb = dims[2 * b + 1];
    for (int py = 0; py < hb; ++py) {
      int y0, y1; float ly;
      bilinear_tap(py, hb, PIB_G0, y0, y1, ly);
      const float wy = (y0 == ty ? 1.f - ly : 0.f) + (y1 == ty ? ly : 0.f);
      if (wy == 0.f) continue;
      const float* row = dres + ((long)b * S_pad + T + 1 + (long)py * (compact ? wb : gw)) * H + c;
      for (int px = 0; px < wb; ++px) {
        int x0, x1; float lx;
        bilinear_tap(px, wb, PIB_G0, x0, x1, lx);
        const float4 v = ld4(row + (long)px * H);
        const float w0 = wy * (1.f - lx), w1 = wy * lx;
#pragma unroll
        for (int t = 0; t < PIB_G0; ++t) {
          const float w = (t == x0 ? w0 : 0.f) + (t == x1 ? w1 : 0.f);
          acc[t].x += w * v.x; acc[t].y += w * v.y; acc[t].z += w * v.z; acc[t].w += w * v.w;
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < PIB_G0; ++t) {
    float* d = dpos + (long)(1 + ty * PIB_G0 + t) * H + c;
    atomicAdd(d, acc[t].x); atomicAdd(d + 1, acc[t].y); atomicAdd(d + 2, acc[t].z); atomicAdd(d + 3, acc[t].w);
  }
}
// part: (NP+1)*ntypes*H floats
extern "C" int climb_image_embed_bwd(const float* dres, const int* img_type, const int* dims, void* dproj, int dproj_dtype, float* dpos, float* dcls,
                                     float* part, int B, int T, int NP, int gw, int g0, int S_pad, int H, int ntypes, int compact, void* stream) {
  if (H % 4 || ntypes > 3 || NP % gw || (dims && g0 != PIB_G0) || H > 768 || (compact && !dims)) return CLIMB_EINVAL;
  int n = (NP + 1) * (H / 4);
  hipStream_t st = (hipStream_t)stream;
  if (dproj_dtype == CLIMB_DT_F32)
    hipLaunchKernelGGL((image_embed_bwd_kernel<float>), dim3((n + 127) / 128), dim3(128), 0, st, dres, img_type, dims, (float*)dproj, dpos, dcls, part, B, T, NP, gw, S_pad, H, ntypes, compact);
  else if (dproj_dtype == CLIMB_DT_BF16)
    hipLaunchKernelGGL((image_embed_bwd_kernel<bf16_t>), dim3((n + 127) / 128), dim3(128), 0, st, dres, img_type, dims, (bf16_t*)dproj, dpos, dcls, part, B, T, NP, gw, S_pad, H, ntypes, compact);
  else return CLIMB_EINVAL;
  LAUNCH_CHECK();
  if (dims && dpos) {
    const int bchunk = 8;
    hipLaunchKernelGGL(pos_interp_bwd_kernel, dim3(PIB_G0, (B + bchunk - 1) / bchunk), dim3(192), 0, st, dres, dims, dpos, B, T, gw, S_pad, H, bchunk, compact);
    LAUNCH_CHECK();
  }
  return CLIMB_OK;
}

// ------------------------------------------------------------------------------------------------
// small elementwise kernels for the pooler / task heads (fp32, [B, <=1536])
__global__ void ew_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n, float s) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float x = a[i], r;
    switch (op) {
      case 0: r = gelu_f(x); break;                  // gelu fwd
      case 1: r = x * dgelu_f(b[i]); break;          // gelu bwd: a=dy, b=pre-activation
      case 2: r = x * (1.f - b[i] * b[i]); break;    // tanh bwd: a=dy, b=tanh output
      case 3: r = x * b[i] * s; break;               // dropout: a=x, b=keep mask, s=1/(1-p)
      case 4: r = x * s; break;                      // scale
      case 5: r = x + b[i]; break;                   // add
      case 6: r = tanhf(x); break;                   // tanh fwd (pooler, after a split-K GEMM)
      default: r = x;
    }
    out[i] = r;
  }
}
extern "C" int climb_elementwise(int op, const float* a, const float* b, float* out, long n, float s, void* stream) {
  if (n <= 0) return CLIMB_EINVAL;
  long nb = (n + 255) / 256;
  hipLaunchKernelGGL(ew_kernel, dim3((int)(nb < 8192 ? nb : 8192)), dim3(256), 0, (hipStream_t)stream, op, a, b, out, n, s);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// ------------------------------------------------------------------------------------------------
// Losses, fused with their gradient.
// VQA (REF/train/visionlanguage_tasks/train_vqa.py:95,:157): BCEWithLogits(mean) * N  ==  sum_{b,n} bce / B.
//   dlogits = gscale * (sigmoid(x) - t) / B.
#define BCE_BLOCKS 128
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ logits, long ldl, const float* __restrict__ target, long ldt,
                                                         float* __restrict__ dlogits, long ldd, float* __restrict__ partials, int B, int N, float gscale) {
  __shared__ float red[4];
  float acc = 0.f;
  const float invB = 1.f / (float)B;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)B * N; i += (long)gridDim.x * 256) {
    int b = (int)(i / N), n = (int)(i - (long)b * N);
    float x = logits[b * ldl + n], t = target[b * ldt + n];
    acc += fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
    if (dlogits) dlogits[b * ldd + n] = gscale * (1.f / (1.f + __expf(-x)) - t) * invB;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(128) void bce_finish_kernel(const float* __restrict__ partials, int n, float scale, float* __restrict__ loss) {
  __shared__ float red[2];
  float acc = threadIdx.x < n ? partials[threadIdx.x] : 0.f;
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *loss = (red[0] + red[1]) * scale;
}
extern "C" int climb_bce_workspace_floats() { return BCE_BLOCKS; }
// partials: BCE_BLOCKS floats of scratch (fixed block count + fixed summation order => deterministic)
extern "C" int climb_bce_logits(const float* logits, long ldl, const float* target, long ldt, float* dlogits, long ldd, float* loss, float* partials,
                                int B, int N, float gscale, void* stream) {
  if (B <= 0 || N <= 0 || !partials) return CLIMB_EINVAL;
  hipLaunchKernelGGL(bce_logits_kernel, dim3(BCE_BLOCKS), dim3(256), 0, (hipStream_t)stream, logits, ldl, target, ldt, dlogits, ldd, partials, B, N, gscale);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, partials, BCE_BLOCKS, 1.f / (float)B, loss);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// nn.CrossEntropyLoss() (REF/train/visionlanguage_tasks/train_nlvr2.py:80): mean over rows of -log softmax[label].
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, long ldl, const long* __restrict__ labels,
                                                            float* __restrict__ dlogits, long ldd, float* __restrict__ loss, int B, int N,
                                                            float gscale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* x = logits + b * ldl;
    float mx = -3.0e38f;
    for (int n = 0; n < N; ++n) mx = fmaxf(mx, x[n]);
    float se = 0.f;
    for (int n = 0; n < N; ++n) se += __expf(x[n] - mx);
    float lse = mx + __logf(se);
    int lab = (int)labels[b];
    acc += lse - x[lab];
    if (dlogits)
      for (int n = 0; n < N; ++n) dlogits[b * ldd + n] = gscale * (__expf(x[n] - lse) - (n == lab ? 1.f : 0.f)) / (float)B;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *loss = (red[0] + red[1] + red[2] + red[3]) / (float)B;
}
extern "C" int climb_cross_entropy(const float* logits, long ldl, const long* labels, float* dlogits, long ldd, float* loss, int B, int N,
                                   float gscale, void* stream) {
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, ldl, labels, dlogits, ldd, loss, B, N, gscale);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// additive key bias from the concatenated [text | image] keep-mask (HF:623-627): 0 keep, -3e38 masked; padding rows masked.
__global__ void key_bias_kernel(const long* __restrict__ attn_mask, float* __restrict__ bias, int B, int T, int S, int S_pad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * S_pad) return;
  int b = i / S_pad, s = i - b * S_pad;
  float v = 0.f;
  if (s < T) v = attn_mask[b * T + s] != 0 ? 0.f : -3.0e38f;
  else if (s >= S) v = -3.0e38f;
  bias[i] = v;
}
extern "C" int climb_key_bias(const long* attn_mask, float* bias, int B, int T, int S, int S_pad, void* stream) {
  hipLaunchKernelGGL(key_bias_kernel, dim3((B * S_pad + 255) / 256), dim3(256), 0, (hipStream_t)stream, attn_mask, bias, B, T, S, S_pad);
  LAUNCH_CHECK();
  return CLIMB_OK;
}
