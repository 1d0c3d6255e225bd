// Houlsby bottleneck adapter, forward, as ONE kernel of the 16-bit throughput mode (BASELINE configs[2]):
//     z = y Wd^T + bd        [M, r]   (r = H / reduction factor: 48)
//     s = silu(z)            [M, r]
//     out = resid + y + s Wu^T + bu   [M, H], fp32 (the residual stream)
// As two launches of the general NT GEMM these are 18 + 27 us per adapter for 0.04 % of a layer's FLOPs: both are skinny (N = 48, then
// K = 48), each reads or writes the whole [M, 768] activation, and the second needs two residual operands.  Here a workgroup owns 32 rows:
// the y tile is staged in LDS once, the four waves split K = 768 of the down-projection and meet in LDS, s never leaves the CU before
// the up-projection consumes it, and the epilogue adds both residuals with row-contiguous 128-byte accesses.  HBM traffic ~ 2 x y + resid +
// out = 115 MB at M = 12288, against 2 x 100 MB and two launch / fill / drain phases.  Measured: 31.7 us against 38.6 for the two GEMMs (the
// adapter step 10.83 -> 10.72 ms); what kept it from the ~20 us its bytes allow was the epilogue's per-lane 4- and 2-byte accesses (a lane owned a
// column: 48 memory instructions of 256 B per 32 x 32 block).  r06: the up-projection is issued with its operands swapped (a lane owns ONE row and
// groups of 4 consecutive columns, as in the NT GEMMs) and every 32 x 32 block is turned through a per-wave 4 KB LDS region (gemm_bf16_nt.h: same
// swizzle), so that residual, y and the result move as 16- / 8-byte accesses, 8 lanes a 128-byte row segment: 12 memory instructions per block.
// MFMA orientation of the down-projection: D = A B^T with A = activation rows (m) and B = weight rows (n): a lane owns column n = lane & 31, its 16
// registers the rows m = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
#include "common.h"

#define AD_ROWS 32
#define AD_MAXR 64          // bottleneck width: a multiple of 16, at most 64

typedef __attribute__((ext_vector_type(4))) unsigned int ad_u32x4;
__device__ __forceinline__ bf16x8 ad_frag(ad_u32x4 v) {
  union { ad_u32x4 u; bf16x8 b; } c;
  c.u = v;
  return c.b;
}

// BWD (r03): the input-gradient half of the adapter's backward has the same shape,
//     t  = dout Wu           [M, r]   (Wu [H, r]: the kernel reads its TRANSPOSED shadow [r, H] where the forward reads Wd)
//     dz = t * silu'(z)      [M, r]   (z saved by the forward; dz is kept for the down-projection's weight gradient)
//     dy = dres + dz Wd      [M, H]   (Wd [r, H]: transposed shadow [H, r]; dres = the fp32 residual gradient; dy in the 16-bit operand type)
// so it is the same kernel with another middle and another epilogue: one launch instead of the two skinny GEMMs (2 x 15.5 us).
// dynamic LDS: y tile [32][H] (row stride H * 2 + 16 bytes) -- later reused as zred[4][32][64] floats -- then sbuf[32][64] 16-bit, then the four
// per-wave 4 KB turn regions of the epilogue
template <bool BWD>
__global__ __launch_bounds__(256) void adapter_kernel(const bf16_t* __restrict__ y, long ldy, const float* __restrict__ resid, long ldr,
                                                      const bf16_t* __restrict__ wd, const float* __restrict__ bd, const bf16_t* __restrict__ wu,
                                                      const float* __restrict__ bu, bf16_t* __restrict__ z, bf16_t* __restrict__ s, long ldz,
                                                      void* __restrict__ out_, long ldo, int M, int H, int r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * AD_ROWS;
  const int ystride = H * 2 + 16;
  const int big = AD_ROWS * ystride > 4 * AD_ROWS * 64 * 4 ? AD_ROWS * ystride : 4 * AD_ROWS * 64 * 4;
  float* zred = reinterpret_cast<float*>(smem);
  bf16_t* sbuf = reinterpret_cast<bf16_t*>(smem + big);
  unsigned char* turn = smem + big + AD_ROWS * 64 * 2 + wid * 4096;
  // ---- stage the y tile (rows past M re-read the last row: computed, never stored)
  const int chunks = H / 8;                                        // 16-byte chunks per row
  // (r06) six loads in flight per thread before the first LDS write: one load per loop iteration made the 12 chunks a thread stages 12 serial round trips
  for (int i0 = tid; i0 < AD_ROWS * chunks; i0 += 256 * 6) {
    ad_u32x4 v[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int idx = i0 + q * 256;
      if (idx < AD_ROWS * chunks) {
        const int row = idx / chunks, c = idx - row * chunks;
        const int gr = m0 + row < M ? m0 + row : M - 1;
        v[q] = *reinterpret_cast<const ad_u32x4*>(y + (long)gr * ldy + c * 8);
      }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int idx = i0 + q * 256;
      if (idx < AD_ROWS * chunks) {
        const int row = idx / chunks, c = idx - row * chunks;
        *reinterpret_cast<ad_u32x4*>(smem + row * ystride + c * 16) = v[q];
      }
    }
  }
  __syncthreads();
  // ---- down-projection: wave w owns k in [w H/4, (w+1) H/4)
  f32x16 acc[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[nb][q] = 0.f;
  const int kq = H / 4;
#pragma unroll 4
  for (int k0 = wid * kq; k0 < (wid + 1) * kq; k0 += 16) {
    const bf16x8 a = ad_frag(*reinterpret_cast<const ad_u32x4*>(smem + l31 * ystride + (k0 + 8 * half) * 2));
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = nb * 32 + l31;
      ad_u32x4 w = {0u, 0u, 0u, 0u};
      if (n < r) w = *reinterpret_cast<const ad_u32x4*>(wd + (long)n * H + k0 + 8 * half);
      acc[nb] = CLIMB_MFMA_H16(a, ad_frag(w), acc[nb], 0, 0, 0);
    }
  }
  __syncthreads();                                                 // every wave is done with the y tile: its space becomes zred
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = (q & 3) + 8 * (q >> 2) + 4 * half;
      zred[(wid * AD_ROWS + m) * 64 + nb * 32 + l31] = acc[nb][q];
    }
  __syncthreads();
  for (int e = tid; e < AD_ROWS * 64; e += 256) {
    const int m = e >> 6, n = e & 63;
    float v = 0.f, sv = 0.f;
    if (n < r) {
      v = zred[(0 * AD_ROWS + m) * 64 + n] + zred[(1 * AD_ROWS + m) * 64 + n] + zred[(2 * AD_ROWS + m) * 64 + n] + zred[(3 * AD_ROWS + m) * 64 + n];
      if (!BWD) {
        v += bd[n];
        sv = silu_f(v);                                            // of the fp32 value, like the GEMM's EPI_SILU epilogue
        if (m0 + m < M && blockIdx.y == 0) {
          z[(long)(m0 + m) * ldz + n] = f32_to_bf16(v);
          s[(long)(m0 + m) * ldz + n] = f32_to_bf16(sv);
        }
      } else {                                                     // z: the forward's pre-activation (input); s: where dz goes (output)
        const int gm = m0 + m < M ? m0 + m : M - 1;
        sv = v * dsilu_f(bf16_to_f32(z[(long)gm * ldz + n]));       // like the GEMM's EPI_DSILU epilogue
        if (m0 + m < M && blockIdx.y == 0) s[(long)(m0 + m) * ldz + n] = f32_to_bf16(sv);
      }
    }
    sbuf[m * 64 + n] = f32_to_bf16(sv);
  }
  __syncthreads();
  // ---- up-projection + both residuals: workgroup (., c) of gridDim.y column slices, wave w of it: columns [(4 c + w) H / (4 gridDim.y), ...), 32 at a time.
  // (The down-projection above is repeated by every column slice; see the launcher for why there is one.)
  const int ksteps = r / 16;
  bf16x8 sf[AD_MAXR / 16];
#pragma unroll
  for (int ks = 0; ks < AD_MAXR / 16; ++ks)
    if (ks < ksteps) sf[ks] = ad_frag(*reinterpret_cast<const ad_u32x4*>(reinterpret_cast<const unsigned char*>(sbuf) + l31 * 128 + (16 * ks + 8 * half) * 2));
  const int cw = H / (4 * gridDim.y);                              // columns per wave
#pragma unroll 1
  for (int nb = 0; nb < cw / 32; ++nb) {
    const int n0 = (blockIdx.y * 4 + wid) * cw + nb * 32;
    const int n = n0 + l31;
    f32x16 o;
#pragma unroll
    for (int q = 0; q < 16; ++q) o[q] = 0.f;
    // operands swapped: D[i = column n0 + ..][j = row l31] -- the lane owns row m0 + l31 and columns n0 + 8 g + 4 half .. + 3 per register group g
#pragma unroll
    for (int ks = 0; ks < AD_MAXR / 16; ++ks)
      if (ks < ksteps) o = CLIMB_MFMA_H16(ad_frag(*reinterpret_cast<const ad_u32x4*>(wu + (long)n * r + 16 * ks + 8 * half)), sf[ks], o, 0, 0, 0);
    // turn: row l31, 16-byte chunk 2 g + half at chunk position (2 g + half) ^ (l31 & 7) (conflict-free writes and row-major reads, gemm_bf16_nt.h)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(turn + l31 * 128 + (((2 * g + half) ^ (l31 & 7)) << 4)) = make_float4(o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]);
    // row-major walk: pass p covers rows 8 p .. 8 p + 7, a lane one float4 (columns 4 c .. 4 c + 3) of row 8 p + lane / 8.  All loads before the first store.
    const int rr = lane >> 3, c = lane & 7;
    float4 ov[4], rv[4];
    uint2 yv[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = 8 * p + rr;
      ov[p] = *reinterpret_cast<const float4*>(turn + row * 128 + (((c ^ row) & 7) << 4));
      const int gm = m0 + row < M ? m0 + row : M - 1;
      rv[p] = ld4(resid + (long)gm * ldr + n0 + 4 * c);
      if (!BWD) yv[p] = *reinterpret_cast<const uint2*>(y + (long)gm * ldy + n0 + 4 * c);
    }
    const float4 b4 = BWD ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4(bu + n0 + 4 * c);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = m0 + 8 * p + rr;
      if (m < M) {
        if (BWD) {
          st4(reinterpret_cast<bf16_t*>(out_) + (long)m * ldo + n0 + 4 * c, make_float4(ov[p].x + rv[p].x, ov[p].y + rv[p].y, ov[p].z + rv[p].z, ov[p].w + rv[p].w));
        } else {          // o + b + resid + y, summed in that order (as before the turn: same bits)
          st4(reinterpret_cast<float*>(out_) + (long)m * ldo + n0 + 4 * c,
              make_float4(ov[p].x + b4.x + rv[p].x + h16lo_to_f32(yv[p].x), ov[p].y + b4.y + rv[p].y + h16hi_to_f32(yv[p].x),
                          ov[p].z + b4.z + rv[p].z + h16lo_to_f32(yv[p].y), ov[p].w + b4.w + rv[p].w + h16hi_to_f32(yv[p].y)));
        }
      }
    }
  }
}

// y [M, H] 16-bit (the sub-layer's output incl. its bias), resid [M, H] fp32, wd [r, H] / wu [H, r] 16-bit weight shadows, bd [r] / bu [H] fp32;
// z, s [M, r] 16-bit (saved for the backward), out [M, H] fp32 (may not alias y; may alias nothing else it reads except resid element-wise).
template <bool BWD>
static int adapter_launch(const void* y, long ldy, const float* resid, long ldr, const void* wd, const float* bd, const void* wu, const float* bu, void* z,
                          void* s, long ldz, void* out, long ldo, int M, int H, int r, void* stream) {
  if (M <= 0 || H <= 0 || (H % 128) || r <= 0 || (r % 16) || r > AD_MAXR || (ldy % 8)) return CLIMB_EUNSUPPORTED;
  const int ystride = H * 2 + 16;
  size_t big = (size_t)AD_ROWS * ystride;
  if (big < (size_t)4 * AD_ROWS * 64 * 4) big = (size_t)4 * AD_ROWS * 64 * 4;
  const size_t lds = big + (size_t)AD_ROWS * 64 * 2 + 4 * 4096;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)adapter_kernel<BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    lds_set = lds;
  }
  const int slices = 1;      // column slices per row block (2 measured 42 us against 32: the repeated y tile costs more than the extra workgroups bring)
  hipLaunchKernelGGL(adapter_kernel<BWD>, dim3((M + AD_ROWS - 1) / AD_ROWS, slices), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)y, ldy, resid, ldr,
                     (const bf16_t*)wd, bd, (const bf16_t*)wu, bu, (bf16_t*)z, (bf16_t*)s, ldz, out, ldo, M, H, r);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// y [M, H] 16-bit (the sub-layer's output incl. its bias), resid [M, H] fp32, wd [r, H] / wu [H, r] 16-bit weight shadows, bd [r] / bu [H] fp32;
// z, s [M, r] 16-bit (saved for the backward), out [M, H] fp32 (may not alias y; may alias nothing else it reads except resid element-wise).
extern "C" int climb_adapter_fwd_bf16(const void* y, long ldy, const float* resid, long ldr, const void* wd, const float* bd, const void* wu,
                                      const float* bu, void* z, void* s, long ldz, float* out, long ldo, int M, int H, int r, void* stream) {
  return adapter_launch<false>(y, ldy, resid, ldr, wd, bd, wu, bu, z, s, ldz, out, ldo, M, H, r, stream);
}

// The input-gradient half of the adapter's backward in one launch: dz = (dout Wu) * silu'(z), dy = dres + dz Wd.
// dout [M, H] 16-bit; dres [M, H] fp32 (the residual gradient d(out)); wu_t [r, H] = Wu^T and wd_t [H, r] = Wd^T (the transposed 16-bit
// shadows); z [M, r] the forward's saved pre-activation; dz [M, r] and dy [M, H] 16-bit outputs (dy may not alias dout).
extern "C" int climb_adapter_bwd_bf16(const void* dout, long ldd, const float* dres, long ldr, const void* wu_t, const void* wd_t, const void* z, void* dz,
                                      long ldz, void* dy, long ldo, int M, int H, int r, void* stream) {
  return adapter_launch<true>(dout, ldd, dres, ldr, wu_t, nullptr, wd_t, nullptr, const_cast<void*>(z), dz, ldz, dy, ldo, M, H, r, stream);
}
