// HBM-bound flat-buffer kernels: fused multi-tensor AdamW, the EWC Fisher-weighted penalty (value + gradient in one
// pass), Fisher square-accumulate, and the fp32 -> bf16 weight shadow.  Every parameter of the learner lives in ONE
// flat fp32 buffer (each tensor 64-element aligned), so each of these is a single launch streaming at HBM rate.
#include "common.h"

struct AdamGroups { AdamGroup g[8]; };

// seg_start[nseg+1]: element offsets of the tensors (ascending, multiples of 4); seg_group[nseg]: group id or -1 (skip).
__device__ __forceinline__ int find_seg(const long* __restrict__ seg_start, int nseg, long e) {
  int lo = 0, hi = nseg;  // largest i with seg_start[i] <= e
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (seg_start[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// torch.optim.AdamW semantics (decoupled decay), REF/modeling/vilt.py:205-215:
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
template <bool SHADOW>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow, long n, const long* __restrict__ seg_start,
                                                    const signed char* __restrict__ seg_group, int nseg, AdamGroups groups, float gscale) {
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (long)gridDim.x * 1024) {
    const int sg = seg_group[find_seg(seg_start, nseg, e)];
    if (sg < 0) continue;
    const AdamGroup G = groups.g[sg];
    float4 pp = ld4(p + e), gg = ld4(g + e), mm = ld4(m + e), vv = ld4(v + e);
    float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ga[4] = {gg.x, gg.y, gg.z, gg.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
    const float isb2 = rsqrtf(G.bc2), step = G.lr / G.bc1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      adamw_update(pa[j], ma[j], va[j], ga[j] * gscale, G, isb2, step);
    }
    st4(p + e, make_float4(pa[0], pa[1], pa[2], pa[3]));
    st4(m + e, make_float4(ma[0], ma[1], ma[2], ma[3]));
    st4(v + e, make_float4(va[0], va[1], va[2], va[3]));
    if (SHADOW) st4(shadow + e, make_float4(pa[0], pa[1], pa[2], pa[3]));
  }
}

extern "C" int climb_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, long n, const long* seg_start,
                           const signed char* seg_group, int nseg, const float* groups, int ngroups, float gscale, void* stream) {
  if (n <= 0 || n % 4 || nseg <= 0 || ngroups <= 0 || ngroups > 8) return CLIMB_EINVAL;
  AdamGroups G;
  for (int i = 0; i < 8; ++i) {
    const float* s = groups + (i < ngroups ? i : 0) * 8;
    G.g[i] = AdamGroup{s[0], s[1], s[2], s[3], s[4], s[5], s[6], 0.f};
  }
  long nb = (n / 4 + 255) / 256;
  dim3 grid((unsigned)(nb < 16384 ? nb : 16384)), blk(256);
  if (shadow_bf16)
    hipLaunchKernelGGL((adamw_kernel<true>), grid, blk, 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)shadow_bf16, n, seg_start, seg_group, nseg, G, gscale);
  else
    hipLaunchKernelGGL((adamw_kernel<false>), grid, blk, 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)nullptr, n, seg_start, seg_group, nseg, G, gscale);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// r04: the same update over the ACTIVE spans of the flat buffer only.  Since the weight matrices are updated in the epilogue of the grouped
// weight-gradient launch (gemm_bf16_tnp.hip), three quarters of the buffer are skipped tensors; adamw_kernel still walked them (a binary search
// per 4 elements: 290 us for 0.9 GB of real work).  spans[4 i .. 4 i + 3] = { first element, elements, first block, gradient source } of span i (maximal runs of
// tensors with a group, in 1024-element blocks); a workgroup looks its span up once per block.  ZERO: the gradient is cleared as it is consumed, so
// the step leaves the gradient buffer all zeros and the next zero_grad() has nothing to do (engine._grad_clean).
// Data parallel (r04): a span flagged in spans[4 i + 3] takes its gradient from `g16` -- the reducer's 16-bit staging buffer, laid out like the
// gradient buffer, holding the all-reduced payload -- times g16_scale (the averaging factor / loss scale the un-cast pass used to apply): that pass
// (2 B read + 4 B written per parameter) and the 4 B read of its result disappear.  Same product, same rounding as un-cast followed by this kernel.
// r05: with EWC the term 2 lam F (theta - theta*) of REF/cl_algorithms/ewc.py:75-87 is added to the gradient of every element below `enc_n` (theta* / F
// are laid out like that range) right here, and lam F (theta - theta*)^2 of those elements to *ewc_loss: the separate penalty pass (16 B per encoder
// parameter) and the re-read of what it parked in the gradient buffer (4 B) become 8 B read in this pass.
struct SpansEwc { const float* star; const float* fisher; float* loss; long enc_n; float lam; int pad; };
template <bool SHADOW, bool ZERO, bool EWC = false>
__global__ __launch_bounds__(256) void adamw_spans_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                          bf16_t* __restrict__ shadow, const long* __restrict__ spans, int nspans, long nblocks,
                                                          const long* __restrict__ seg_start, const signed char* __restrict__ seg_group, int nseg,
                                                          AdamGroups groups, float gscale, const bf16_t* __restrict__ g16, float g16_scale, SpansEwc ew = SpansEwc()) {
  float ewc_sum = 0.f;
  for (long b = blockIdx.x; b < nblocks; b += gridDim.x) {
    int lo = 0, hi = nspans - 1;                     // last span whose first block is <= b
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (spans[4 * mid + 2] <= b) lo = mid; else hi = mid - 1;
    }
    const long start = spans[4 * lo], len = spans[4 * lo + 1], off = (b - spans[4 * lo + 2]) * 1024 + threadIdx.x * 4;
    const bool from16 = spans[4 * lo + 3] != 0;
    if (off >= len) continue;
    const long e = start + off;
    const int sg = seg_group[find_seg(seg_start, nseg, e)];
    if (sg < 0) continue;
    const AdamGroup G = groups.g[sg];
    float4 pp = ld4(p + e), gg, mm = ld4(m + e), vv = ld4(v + e);
    if (from16) {
      gg = ld4(g16 + e);
      gg.x *= g16_scale; gg.y *= g16_scale; gg.z *= g16_scale; gg.w *= g16_scale;
    } else gg = ld4(g + e);
    float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ga[4] = {gg.x, gg.y, gg.z, gg.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
    const float isb2 = rsqrtf(G.bc2), step = G.lr / G.bc1;
#pragma unroll
    for (int j = 0; j < 4; ++j) ga[j] *= gscale;
    if constexpr (EWC) {
      if (e < ew.enc_n) {          // (tensors are 64-element aligned: a float4 never straddles the end of the encoder range)
        const float4 s4 = ld4(ew.star + e), f4 = ld4(ew.fisher + e);
        const float sa[4] = {s4.x, s4.y, s4.z, s4.w}, fa[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = pa[j] - sa[j], fd = fa[j] * d;
          ewc_sum = fmaf(fd, d, ewc_sum);
          ga[j] += 2.f * ew.lam * fd;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) adamw_update(pa[j], ma[j], va[j], ga[j], G, isb2, step);
    st4(p + e, make_float4(pa[0], pa[1], pa[2], pa[3]));
    st4(m + e, make_float4(ma[0], ma[1], ma[2], ma[3]));
    st4(v + e, make_float4(va[0], va[1], va[2], va[3]));
    if (SHADOW) st4(shadow + e, make_float4(pa[0], pa[1], pa[2], pa[3]));
    if (ZERO) st4(g + e, make_float4(0.f, 0.f, 0.f, 0.f));
  }
  if constexpr (EWC) {
    ewc_sum = wave_sum(ewc_sum);
    if ((threadIdx.x & 63) == 0 && ewc_sum != 0.f) atomicAdd(ew.loss, ew.lam * ewc_sum);
  }
}

static int adamw_spans_launch(float* p, float* g, float* m, float* v, void* shadow_bf16, const long* spans, int nspans, long nblocks, const long* seg_start,
                              const signed char* seg_group, int nseg, const float* groups, int ngroups, float gscale, int zero_grad, const void* g16,
                              float g16_scale, const SpansEwc* ew, void* stream);
extern "C" int climb_adamw_spans(float* p, float* g, float* m, float* v, void* shadow_bf16, const long* spans, int nspans, long nblocks,
                                 const long* seg_start, const signed char* seg_group, int nseg, const float* groups, int ngroups, float gscale,
                                 int zero_grad, const void* g16, float g16_scale, void* stream) {
  return adamw_spans_launch(p, g, m, v, shadow_bf16, spans, nspans, nblocks, seg_start, seg_group, nseg, groups, ngroups, gscale, zero_grad, g16, g16_scale, nullptr, stream);
}
// r05: the same pass with the EWC term for the elements below enc_n (see SpansEwc); *ewc_loss is ADDED to (the caller zeroes it once per step)
extern "C" int climb_adamw_spans_ewc(float* p, float* g, float* m, float* v, void* shadow_bf16, const long* spans, int nspans, long nblocks,
                                     const long* seg_start, const signed char* seg_group, int nseg, const float* groups, int ngroups, float gscale,
                                     int zero_grad, const void* g16, float g16_scale, const float* star, const float* fisher, long enc_n, float lam,
                                     float* ewc_loss, void* stream) {
  if (!star || !fisher || !ewc_loss || enc_n <= 0 || (enc_n % 4)) return CLIMB_EINVAL;
  const SpansEwc e{star, fisher, ewc_loss, enc_n, lam, 0};
  return adamw_spans_launch(p, g, m, v, shadow_bf16, spans, nspans, nblocks, seg_start, seg_group, nseg, groups, ngroups, gscale, zero_grad, g16, g16_scale, &e, stream);
}
static int adamw_spans_launch(float* p, float* g, float* m, float* v, void* shadow_bf16, const long* spans, int nspans, long nblocks, const long* seg_start,
                              const signed char* seg_group, int nseg, const float* groups, int ngroups, float gscale, int zero_grad, const void* g16,
                              float g16_scale, const SpansEwc* ew, void* stream) {
  if (nspans <= 0 || nblocks <= 0 || nseg <= 0 || ngroups <= 0 || ngroups > 8 || !spans) return CLIMB_EINVAL;
  AdamGroups G;
  for (int i = 0; i < 8; ++i) {
    const float* s = groups + (i < ngroups ? i : 0) * 8;
    G.g[i] = AdamGroup{s[0], s[1], s[2], s[3], s[4], s[5], s[6], 0.f};
  }
  dim3 grid((unsigned)(nblocks < 16384 ? nblocks : 16384)), blk(256);
  if (ew) {
    if (!shadow_bf16) return CLIMB_EUNSUPPORTED;          // (the fold is the 16-bit training step's)
    if (zero_grad)
      hipLaunchKernelGGL((adamw_spans_kernel<true, true, true>), grid, blk, 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)shadow_bf16, spans, nspans, nblocks, seg_start,
                         seg_group, nseg, G, gscale, (const bf16_t*)g16, g16_scale, *ew);
    else
      hipLaunchKernelGGL((adamw_spans_kernel<true, false, true>), grid, blk, 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)shadow_bf16, spans, nspans, nblocks, seg_start,
                         seg_group, nseg, G, gscale, (const bf16_t*)g16, g16_scale, *ew);
    LAUNCH_CHECK();
    return CLIMB_OK;
  }
#define ADAMW_SPANS(S_, Z_)                                                                                                                          \
  hipLaunchKernelGGL((adamw_spans_kernel<S_, Z_>), grid, blk, 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)shadow_bf16, spans, nspans, nblocks, seg_start, \
                     seg_group, nseg, G, gscale, (const bf16_t*)g16, g16_scale)
  if (shadow_bf16) { if (zero_grad) ADAMW_SPANS(true, true); else ADAMW_SPANS(true, false); }
  else             { if (zero_grad) ADAMW_SPANS(false, true); else ADAMW_SPANS(false, false); }
#undef ADAMW_SPANS
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// EWC (REF/cl_algorithms/ewc.py:75-87): loss = lam * sum F (theta - theta*)^2 ;  dloss/dtheta = 2 lam F (theta - theta*).
// One pass over the encoder range: block partial sums -> partials[grid]; grad (optional) accumulated in place.
__global__ __launch_bounds__(256) void ewc_kernel(const float* __restrict__ theta, const float* __restrict__ star, const float* __restrict__ fisher,
                                                  float* __restrict__ grad, long n, float two_lam_gs, float* __restrict__ partials) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (long)gridDim.x * 1024) {
    float4 t = ld4(theta + e), s = ld4(star + e), f = ld4(fisher + e);
    float dx = t.x - s.x, dy = t.y - s.y, dz = t.z - s.z, dw = t.w - s.w;
    acc += f.x * dx * dx + f.y * dy * dy + f.z * dz * dz + f.w * dw * dw;
    if (grad) {
      float4 g = ld4(grad + e);
      st4(grad + e, make_float4(g.x + two_lam_gs * f.x * dx, g.y + two_lam_gs * f.y * dy, g.z + two_lam_gs * f.z * dz, g.w + two_lam_gs * f.w * dw));
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partials, int n, float scale, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *out = scale * ((red[0] + red[1]) + (red[2] + red[3]));
}
#define EWC_BLOCKS 2048
extern "C" int climb_ewc_workspace_floats() { return EWC_BLOCKS; }
// loss_out = lam * sum F (theta-theta*)^2 ; if grad != NULL: grad += gscale * 2 lam F (theta - theta*)
extern "C" int climb_ewc_penalty(const float* theta, const float* star, const float* fisher, float* grad, long n, float lam, float gscale,
                                 float* partials, float* loss_out, void* stream) {
  if (n <= 0 || n % 4) return CLIMB_EINVAL;
  long nb = (n / 4 + 255) / 256;
  int grid = (int)(nb < EWC_BLOCKS ? nb : EWC_BLOCKS);
  hipLaunchKernelGGL(ewc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, theta, star, fisher, grad, n, 2.f * lam * gscale, partials);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, grid, lam, loss_out);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// Fisher estimate (REF/cl_algorithms/ewc.py:62-64): F += g^2
__global__ void fisher_accum_kernel(float* __restrict__ fisher, const float* __restrict__ grad, long n) {
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (long)gridDim.x * 1024) {
    float4 f = ld4(fisher + e), g = ld4(grad + e);
    st4(fisher + e, make_float4(f.x + g.x * g.x, f.y + g.y * g.y, f.z + g.z * g.z, f.w + g.w * g.w));
  }
}
extern "C" int climb_fisher_accum(float* fisher, const float* grad, long n, void* stream) {
  if (n <= 0 || n % 4) return CLIMB_EINVAL;
  long nb = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(fisher_accum_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, (hipStream_t)stream, fisher, grad, n);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// x *= s   (Fisher / num_samples; gradient averaging under data parallelism)
__global__ void scale_kernel(float* __restrict__ x, long n, float s) {
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (long)gridDim.x * 1024) {
    float4 v = ld4(x + e);
    st4(x + e, make_float4(v.x * s, v.y * s, v.z * s, v.w * s));
  }
}
extern "C" int climb_scale(float* x, long n, float s, void* stream) {
  if (n <= 0 || n % 4) return CLIMB_EINVAL;
  long nb = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, (hipStream_t)stream, x, n, s);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// fp32 master -> bf16 shadow (round-to-nearest-even), flat
__global__ void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (long)gridDim.x * 1024) st4(y + e, ld4(x + e));
}
extern "C" int climb_cast_bf16(const float* x, void* y, long n, void* stream) {
  if (n <= 0 || n % 4) return CLIMB_EINVAL;
  long nb = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// gradient-payload helpers of the data-parallel path: y = scale * float(x) for a bf16 payload coming back from the wire, and an
// in-place scale for backends without an averaging reduction
__global__ void uncast_bf16_scale_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n, float s) {
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (long)gridDim.x * 1024) {
    float4 v = ld4(x + e);
    st4(y + e, make_float4(v.x * s, v.y * s, v.z * s, v.w * s));
  }
}
extern "C" int climb_uncast_bf16_scale(const void* x, float* y, long n, float scale, void* stream) {
  if (n <= 0 || n % 4) return CLIMB_EINVAL;
  long nb = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(uncast_bf16_scale_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, n, scale);
  LAUNCH_CHECK();
  return CLIMB_OK;
}
__global__ void scale_f32_kernel(float* __restrict__ x, long n, float s) {
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (long)gridDim.x * 1024) {
    float4 v = ld4(x + e);
    st4(x + e, make_float4(v.x * s, v.y * s, v.z * s, v.w * s));
  }
}
extern "C" int climb_scale_f32(float* x, long n, float scale, void* stream) {
  if (n <= 0 || n % 4) return CLIMB_EINVAL;
  long nb = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(scale_f32_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, (hipStream_t)stream, x, n, scale);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// bf16 [R, C] -> [C, R] (weight shadows for the dX GEMMs), 64x64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R, int C) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    int r = e >> 6, c = e & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? in[(long)(r0 + r) * C + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    int c = e >> 6, r = e & 63;
    if (r0 + r < R && c0 + c < C) out[(long)(c0 + c) * R + r0 + r] = tile[r][c];
  }
}
extern "C" int climb_transpose_bf16(const void* in, void* out, int R, int C, void* stream) {
  if (R <= 0 || C <= 0) return CLIMB_EINVAL;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, R, C);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// many [R,C] -> [C,R] transposes in one launch: table[i] = {src_off, dst_off, R, C} (elements), grid.y = matrix.
// 64x64 tiles; when R and C are multiples of 8 (every weight matrix of the model) both global sides move 16 bytes per thread:
// a wave reads 8 rows x 128 B and writes 8 transposed rows x 128 B, the 2-byte element shuffle happens in LDS only.
__global__ __launch_bounds__(256) void transpose_bf16_batched_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                                     const long* __restrict__ table) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];
  const long* t = table + 4 * blockIdx.y;
  const int R = (int)t[2], C = (int)t[3];
  const int tc = (C + 63) / 64, tr = (R + 63) / 64;
  const bf16_t* in = src + t[0];
  bf16_t* out = dst + t[1];
  const bool vec = ((R | C) & 7) == 0 && ((t[0] | t[1]) & 7) == 0;
  for (int tileid = blockIdx.x; tileid < tc * tr; tileid += gridDim.x) {
    const int r0 = (tileid / tc) * 64, c0 = (tileid % tc) * 64;
    __syncthreads();
    if (vec) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = threadIdx.x + 256 * i, r = q >> 3, c = (q & 7) * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r0 + r < R && c0 + c < C) v = *reinterpret_cast<const uint4*>(in + (long)(r0 + r) * C + c0 + c);
        *reinterpret_cast<uint4*>(&tile[r][c]) = v;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = threadIdx.x + 256 * i, c = q >> 3, r = (q & 7) * 8;
        if (r0 + r < R && c0 + c < C) {
          uint4 v;
          v.x = tile[r][c] | ((unsigned)tile[r + 1][c] << 16);
          v.y = tile[r + 2][c] | ((unsigned)tile[r + 3][c] << 16);
          v.z = tile[r + 4][c] | ((unsigned)tile[r + 5][c] << 16);
          v.w = tile[r + 6][c] | ((unsigned)tile[r + 7][c] << 16);
          *reinterpret_cast<uint4*>(out + (long)(c0 + c) * R + r0 + r) = v;
        }
      }
    } else {
      for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        int r = e >> 6, c = e & 63;
        tile[r][c] = (r0 + r < R && c0 + c < C) ? in[(long)(r0 + r) * C + c0 + c] : (bf16_t)0;
      }
      __syncthreads();
      for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        int c = e >> 6, r = e & 63;
        if (r0 + r < R && c0 + c < C) out[(long)(c0 + c) * R + r0 + r] = tile[r][c];
      }
    }
  }
}
extern "C" int climb_transpose_bf16_batched(const void* src, void* dst, const long* table, int n, int tiles_per_matrix, void* stream) {
  if (n <= 0 || tiles_per_matrix <= 0) return CLIMB_EINVAL;
  hipLaunchKernelGGL(transpose_bf16_batched_kernel, dim3(tiles_per_matrix, n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, table);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// Diagnostic (bench.py's roofline): the matrix pipe alone -- 16 independent 32 x 32 x 16 accumulations per wave, one wave per SIMD, operands
// loaded once into registers, `iters` rounds, no memory traffic in the loop.  Timed by the caller, it gives what the chip SUSTAINS on the given
// operand data: the clock follows the power budget, which follows the bits that toggle (1.75 PF on N(0,1) bf16 operands
// against 2.48 on zeros, all 256 CUs).  src: blocks * 256 * 64 16-bit values; out: blocks * 256 floats (keeps the loop alive).
__global__ __launch_bounds__(256) void mfma_sustained_kernel(const bf16_t* __restrict__ src, float* __restrict__ out, int iters) {
  bf16x8 a[4], b[4];
  const bf16x8* p = reinterpret_cast<const bf16x8*>(src) + ((long)blockIdx.x * 256 + threadIdx.x) * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = p[i]; b[i] = p[4 + i]; }
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = CLIMB_MFMA_H16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(long)blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" int climb_mfma_sustained_probe(const void* src, float* out, int blocks, int iters, void* stream) {
  if (!src || !out || blocks <= 0 || iters <= 0) return CLIMB_EINVAL;
  hipLaunchKernelGGL(mfma_sustained_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, out, iters);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

extern "C" int climb_version() { return 100; }
extern "C" const char* climb_arch() { return "gfx950"; }
extern "C" const char* climb_h16() { return CLIMB_H16_NAME; }
extern "C" const char* climb_error_string(int code) {
  if (code == CLIMB_OK) return "ok";
  if (code == CLIMB_EINVAL) return "climb: invalid argument";
  if (code == CLIMB_EUNSUPPORTED) return "climb: unsupported shape";
  return hipGetErrorString((hipError_t)code);
}
extern "C" int climb_device_sync() { return (int)hipDeviceSynchronize(); }
