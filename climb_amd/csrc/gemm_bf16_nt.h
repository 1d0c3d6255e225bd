// Shared pieces of the bf16 NT GEMM kernels (gfx950): LDS swizzle, LDS-DMA staging, the row-contiguous LDS-turned epilogue,
// the XCD-aware tile order.  Included by gemm_bf16.hip (128x128 / 64x128 / 96x192 / 192x192 tiles) and gemm_bf16_nt256.hip
// (256x256 tiles, 8-phase schedule).
#pragma once
#include "common.h"

#define GB_BM 128
#define GB_BN 128
#define GB_BK 64

__device__ __forceinline__ int swz(int row) {  // chunk XOR mask: rows (r, r+2) differ in bit 2, 16 rows of a group all distinct with parity
  int y = (row >> 1) & 7;
  return ((y & 1) << 2) | (y >> 1);
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) {
  union { u32x4 u; bf16x8 b; } c;
  c.u = v;
  return c.b;
}

// LDS-DMA staging (global_load_lds_dwordx4): no VGPR round trip and no ds_write pass -- the register-staged path spends
// more LDS cycles on its 13-cycle ds_write_b128s than on the fragment reads.  The DMA writes LDS linearly
// (wave-uniform base + lane*16), so the swizzle is applied on the SOURCE side: the lane that owns LDS slot (row, cpos)
// fetches logical chunk cpos ^ swz(row) of that row.  A [128][64] tile is 16 wave-instructions, dealt evenly to NW waves.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
template <int NW, int ROWS = 128>
__device__ __forceinline__ void nt_glds(const bf16_t* __restrict__ P, long ld, int row0, int k0, int R, unsigned char* __restrict__ S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < ROWS / 8 / NW; ++i) {
    const int j = wave * (ROWS / 8 / NW) + i;
    const int slot = j * 64 + lane, row = slot >> 3, c = (slot & 7) ^ swz(row);
    int grow = row0 + row;
    grow = grow < R ? grow : R - 1;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(P + (long)grow * ld + k0 + c * 8), (lds_void_t*)(S + j * 1024), 16, 0, 0);
  }
}

// Epilogue.  In the 32x32 accumulator layout a lane owns ONE output row and 4-column groups of it, so a direct store touches 32
// different rows with 8-16 bytes each: 8x the memory transactions of a row-contiguous store, and measured (r01 ablation: 633 -> 448
// us per layer with the epilogue removed) the dominant cost of the K = 768 GEMMs.  Instead every wave turns its 32 x (NI*32)
// block through a PRIVATE fp32 LDS region (the k-loop buffers are free by then) and then walks it row-major: 16 B of LDS per lane,
// NI*8 lanes per row, so bias / residual / pre-activation are read and C / aux_out written as whole 128-byte lines.
// Staged row r holds 16-byte chunk c at chunk position (c & ~7) | ((c ^ r) & 7): the transposing ds_write_b128s (8 consecutive
// rows, same chunk) and the row-major ds_read_b128s are both bank-conflict free.  No workgroup barrier: the region is per wave
// and LDS operations of one wave execute in order.
// The residual / pre-activation operand of the epilogue, fetched BEFORE the k-loop in the same row-major lane assignment the
// epilogue uses: its latency (measured: 36 us per layer when loaded in the epilogue) disappears under the MFMAs.
template <int EPI, int CNT>
struct AuxRegs {
  float4 r[CNT];     // fp32 residual            (EPI_RESID, EPI_RESID2)
  uint2 h[CNT];      // 4 bf16: pre-activation   (EPI_DGELU, EPI_DSILU) or second residual (EPI_RESID2)
};
template <int EPI, int NI, int MJ>
__device__ __forceinline__ void nt_aux_prefetch_l(AuxRegs<EPI, NI * 4 * MJ>& ax, int lane, int mw, int nw, int M, int N, const void* __restrict__ aux,
                                                  long ldaux, const bf16_t* __restrict__ aux2, long ldaux2) {
  constexpr int CPR = NI * 8;
#pragma unroll
  for (int j = 0; j < MJ; ++j)
#pragma unroll
    for (int p = 0; p < NI * 4; ++p) {
      const int q = p * 64 + lane, r = q / CPR, c = q % CPR;
      const int m = min(mw + j * 32 + r, M - 1), n = min(nw + c * 4, N - 4);       // clamped: out-of-range results are never used
      if (EPI == EPI_RESID || EPI == EPI_RESID2) ax.r[j * NI * 4 + p] = ld4(reinterpret_cast<const float*>(aux) + (long)m * ldaux + n);
      if (EPI == EPI_DGELU || EPI == EPI_DSILU || EPI == EPI_MUL) ax.h[j * NI * 4 + p] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(aux) + (long)m * ldaux + n);
      if (EPI == EPI_RESID2) ax.h[j * NI * 4 + p] = *reinterpret_cast<const uint2*>(aux2 + (long)m * ldaux2 + n);
    }
}
template <int EPI, int NI, int MJ>
__device__ __forceinline__ void nt_aux_prefetch(AuxRegs<EPI, NI * 4 * MJ>& ax, int mw, int nw, int M, int N, const void* __restrict__ aux, long ldaux,
                                                const bf16_t* __restrict__ aux2, long ldaux2) {
  nt_aux_prefetch_l<EPI, NI, MJ>(ax, threadIdx.x & 63, mw, nw, M, N, aux, ldaux, aux2, ldaux2);
}
// Makes the compiler wait for the operand loads HERE (an empty asm that "uses" every register): placed before a block's first store,
// so that the wait is a counted load wait and not the vmcnt(0) store drain a later first use would cost (see NtBias).
template <int EPI, int CNT>
__device__ __forceinline__ void nt_aux_touch(AuxRegs<EPI, CNT>& ax) {
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    if (EPI == EPI_RESID || EPI == EPI_RESID2) asm volatile("" : "+v"(ax.r[i].x), "+v"(ax.r[i].y), "+v"(ax.r[i].z), "+v"(ax.r[i].w));
    if (EPI == EPI_DGELU || EPI == EPI_DSILU || EPI == EPI_RESID2 || EPI == EPI_MUL) asm volatile("" : "+v"(ax.h[i].x), "+v"(ax.h[i].y));
  }
}
__device__ __forceinline__ float4 bf16x4_to_f32(uint2 r) {
  return make_float4(h16lo_to_f32(r.x), h16hi_to_f32(r.x), h16lo_to_f32(r.y), h16hi_to_f32(r.y));
}

// One 32-row block of a wave's output, in two steps: nt_epi_stage writes the accumulators acc[0..NI) (32 x NI*32, lane = row) into the
// wave's staging region; nt_epi_drain walks the region row-major, applies the epilogue and stores.  `ax` holds this lane's
// residual / pre-activation operands for the block at indices AXJ .. AXJ + NI*4 (fetched in the drain's lane assignment).
template <int NI>
__device__ __forceinline__ void nt_epi_stage(const f32x16 (&acc)[NI], unsigned char* __restrict__ stage, int lane) {
  constexpr int PITCH = NI * 128;          // bytes per staged row
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(stage + l31 * PITCH + ((i * 8 + ((2 * g + half) ^ (l31 & 7))) << 4)) =
          make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
}
// The bias a lane needs while it drains a block: its column chunk c = (p*64 + lane) % CPR repeats with period CPR / gcd(64, CPR)
// (1 for 16 / 32 chunks per row, 3 for 24), so 1 - 3 float4 loaded ONCE, before the wave's first store.  Why it matters: on gfx9
// loads and stores share the vmcnt counter and hipcc must treat a counter with both kinds pending as out of order -- a bias load
// issued after a store is waited for with vmcnt(0), i.e. every iteration of the drain loop waited for ALL earlier stores to
// complete (visible in the ISA as `s_waitcnt vmcnt(0)` per iteration; measured r02: the dominant cost of the epilogues).
template <int NI> struct NtBias {
  static constexpr int CPR = NI * 8;
  static constexpr int gcd_(int a, int b) { return b == 0 ? a : gcd_(b, a % b); }
  static constexpr int PERIOD = CPR / gcd_(64, CPR);
  float4 v[PERIOD];
};
template <int NI>
__device__ __forceinline__ void nt_bias_preload(NtBias<NI>& b, const float* __restrict__ bias, int lane, int nw, int N) {
#pragma unroll
  for (int pp = 0; pp < NtBias<NI>::PERIOD; ++pp) {
    const int c = (pp * 64 + lane) % NtBias<NI>::CPR, n = min(nw + c * 4, N - 4);
    b.v[pp] = bias ? ld4(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int pp = 0; pp < NtBias<NI>::PERIOD; ++pp) asm volatile("" : "+v"(b.v[pp].x), "+v"(b.v[pp].y), "+v"(b.v[pp].z), "+v"(b.v[pp].w));
}
template <typename TO, int EPI, int NI, int AXJ, typename AX>
__device__ __forceinline__ void nt_epi_drain(const AX& ax, const NtBias<NI>& bb, const unsigned char* __restrict__ stage, int lane, int mb, int nw, int M,
                                             int N, TO* __restrict__ C, long ldc, bf16_t* __restrict__ aux_out, long ldauxo) {
  constexpr int CPR = NI * 8, PITCH = NI * 128;          // chunks / bytes per staged row
#pragma unroll
  for (int p = 0; p < NI * 4; ++p) {
    const int q = p * 64 + lane, r = q / CPR, c = q % CPR;
    float4 v = *reinterpret_cast<const float4*>(stage + r * PITCH + (((c & ~7) | ((c ^ r) & 7)) << 4));
    const int m = mb + r, n = nw + c * 4;
    if (m >= M || n >= N) continue;
    {
      const float4 bv = bb.v[p % NtBias<NI>::PERIOD];
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    }
    if (EPI == EPI_GELU) {
      st4(aux_out + (long)m * ldauxo + n, v);
      v = make_float4(gelu_fast(v.x), gelu_fast(v.y), gelu_fast(v.z), gelu_fast(v.w));
    } else if (EPI == EPI_RESID) {
      const float4 rv = ax.r[AXJ + p];
      v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
    } else if (EPI == EPI_DGELU) {
      const float4 uv = bf16x4_to_f32(ax.h[AXJ + p]);
      v.x *= dgelu_fast(uv.x); v.y *= dgelu_fast(uv.y); v.z *= dgelu_fast(uv.z); v.w *= dgelu_fast(uv.w);
    } else if (EPI == EPI_GELUD) {
      float4 d;
      gelu_fast_pair(v.x, v.x, d.x); gelu_fast_pair(v.y, v.y, d.y); gelu_fast_pair(v.z, v.z, d.z); gelu_fast_pair(v.w, v.w, d.w);
      st4(aux_out + (long)m * ldauxo + n, d);
    } else if (EPI == EPI_MUL) {
      const float4 uv = bf16x4_to_f32(ax.h[AXJ + p]);
      v.x *= uv.x; v.y *= uv.y; v.z *= uv.z; v.w *= uv.w;
    } else if (EPI == EPI_SILU) {
      st4(aux_out + (long)m * ldauxo + n, v);
      v = make_float4(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w));
    } else if (EPI == EPI_DSILU) {
      const float4 uv = bf16x4_to_f32(ax.h[AXJ + p]);
      v.x *= dsilu_f(uv.x); v.y *= dsilu_f(uv.y); v.z *= dsilu_f(uv.z); v.w *= dsilu_f(uv.w);
    } else if (EPI == EPI_RESID2) {
      const float4 rv = ax.r[AXJ + p];
      const float4 yv = bf16x4_to_f32(ax.h[AXJ + p]);
      v.x += rv.x + yv.x; v.y += rv.y + yv.y; v.z += rv.z + yv.z; v.w += rv.w + yv.w;
    }
    st4(C + (long)m * ldc + n, v);
  }
}
template <typename TO, int EPI, int NI, int AXJ, typename AX>
__device__ __forceinline__ void nt_epi_block(const f32x16 (&acc)[NI], const AX& ax, const NtBias<NI>& bb, unsigned char* __restrict__ stage, int mb, int nw,
                                             int M, int N, TO* __restrict__ C, long ldc, bf16_t* __restrict__ aux_out, long ldauxo) {
  const int lane = threadIdx.x & 63;
  nt_epi_stage<NI>(acc, stage, lane);
  nt_epi_drain<TO, EPI, NI, AXJ>(ax, bb, stage, lane, mb, nw, M, N, C, ldc, aux_out, ldauxo);
}

template <int J, typename TO, int EPI, int NI, int MJ>
__device__ __forceinline__ void nt_epilogue_j(const f32x16 (&acc)[NI][MJ], const AuxRegs<EPI, NI * 4 * MJ>& ax, const NtBias<NI>& bb,
                                              unsigned char* __restrict__ stage, int mw, int nw, int M, int N, TO* __restrict__ C, long ldc,
                                              bf16_t* __restrict__ aux_out, long ldauxo) {
  if constexpr (J < MJ) {
    f32x16 a[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) a[i] = acc[i][J];
    nt_epi_block<TO, EPI, NI, J * NI * 4>(a, ax, bb, stage, mw + J * 32, nw, M, N, C, ldc, aux_out, ldauxo);
    nt_epilogue_j<J + 1, TO, EPI, NI, MJ>(acc, ax, bb, stage, mw, nw, M, N, C, ldc, aux_out, ldauxo);
  }
}

template <typename TO, int EPI, int NI, int MJ>
__device__ __forceinline__ void nt_epilogue(const f32x16 (&acc)[NI][MJ], const AuxRegs<EPI, NI * 4 * MJ>& ax, unsigned char* __restrict__ stage, int mw,
                                            int nw, int M, int N, TO* __restrict__ C, long ldc, const float* __restrict__ bias,
                                            bf16_t* __restrict__ aux_out, long ldauxo) {
  NtBias<NI> bb;                     // every load of the epilogue is issued before its first store (see NtBias)
  nt_bias_preload<NI>(bb, bias, threadIdx.x & 63, nw, N);
  nt_epilogue_j<0, TO, EPI, NI, MJ>(acc, ax, bb, stage, mw, nw, M, N, C, ldc, aux_out, ldauxo);
}

// XCD-aware tile order.  Blocks are dealt round-robin to the 8 XCDs (private 4 MB L2 each); remap so that every XCD owns a
// CONTIGUOUS chunk of a supertile order: groups of GM M-tiles, inside a group N-tile-major.  The ~64 workgroups an XCD runs
// concurrently then cover ~8 A panels x ~8 B panels (~3 MB at K = 768) instead of 64 A panels x 1 B panel, so both operands are
// re-read from that L2, not from HBM / Infinity Cache.
// gm = M-tiles per supertile.  When the M-tiles divide evenly over the 8 XCDs (12288 rows: 48 tiles of 256 = 6 per XCD) gm = nbm / 8 makes a
// supertile exactly one XCD's share, so every A panel is fetched into ONE L2; with gm = 8 the boundary between two XCDs' shares cuts through
// the supertiles and each A panel is fetched by two of them (measured: 153 MB fetched for 80 MB of operands on the N = 768 GEMMs).
__device__ __forceinline__ void nt_tile_id_gm(int gm, int id, int nbm, int nbn, int& tm, int& tn) {      // id: position in launch order (id % 8 = XCD)
  const int nwg = nbm * nbn, q = nwg / 8, rr = nwg % 8, xcd = id % 8, idx = id / 8;
  const int bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;      // bijective for any nwg
  const int per_group = gm * nbn, grp = bid / per_group, in = bid - grp * per_group;
  const int rows = min(gm, nbm - grp * gm);          // last group may hold fewer than gm M-tiles
  tn = in / rows;
  tm = grp * gm + (in - tn * rows);
}
template <int GM>
__device__ __forceinline__ void nt_tile_id(int id, int nbm, int nbn, int& tm, int& tn) { nt_tile_id_gm(GM, id, nbm, nbn, tm, tn); }
template <int GM>
__device__ __forceinline__ void nt_tile(int nbm, int nbn, int& tm, int& tn) { nt_tile_id<GM>(blockIdx.x, nbm, nbn, tm, tn); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// gemm_bf16_nt256.hip: 256 x {256, 192} tiles, one persistent workgroup per CU, counted-vmcnt phase schedule (K % 64 == 0, K >= 128;
// epilogues NONE / GELU / RESID / DGELU).  bn = 256 or 192.
int climb_nt256_launch(int bn, const bf16_t* A, long lda, const bf16_t* B, long ldb, void* C, long ldc, int c_dtype, int M, int N, int K, const float* bias,
                       int epi, const void* aux, long ldaux, bf16_t* aux_out, long ldauxo, const bf16_t* aux2, long ldaux2, hipStream_t st);
void climb_nt256_set_probe(int v);      // measurement aid: 1 = run the k-loop only (no epilogue, nothing stored)
void climb_nt256_set_grid(int v);       // workgroups launched at most (0 = one per tile; default 256 = persistent, one per CU)
// gemm_bf16_tnp.hip: the weight-gradient GEMM C[N,K] += A[M,N]^T B[M,K] on the same persistent phase structure; CLIMB_EUNSUPPORTED for
// shapes it does not take (the caller then uses the 128 x 128 kernel)
int climb_tnp_launch(const bf16_t* A, long lda, const bf16_t* B, long ldb, float* C, long ldc, int M, int N, int K, float* dbias, hipStream_t st);
void climb_tnp_set_workspace(void* ptr, long bytes);
void climb_tn_set_stagger(int groups);      // climb_set_option 22: phase groups of the grouped launch's plan (staggered epilogues; 0 / 1 = off)
// gemm_bf16_nt2p.hip: 128 x 192 tiles, TWO persistent 4-wave workgroups per CU (the epilogue of one runs under the k-loop of the other)
int climb_nt2_launch(const bf16_t* A, long lda, const bf16_t* B, long ldb, void* C, long ldc, int c_dtype, int M, int N, int K, const float* bias, int epi,
                     const void* aux, long ldaux, bf16_t* aux_out, long ldauxo, hipStream_t st);
void climb_nt2_set_dephase(int units_of_64_clocks);
// gemm_bf16_nt4.hip (r04): 192 x 192 tiles, four waves (one per SIMD), two accumulator sets -- the epilogue of a tile runs inside the k-loop of the
// next one.  CLIMB_EUNSUPPORTED for the shapes / epilogues it does not take.
int climb_nt4_launch(const bf16_t* A, long lda, const bf16_t* B, long ldb, void* C, long ldc, int c_dtype, int M, int N, int K, const float* bias, int epi,
                     const void* aux, long ldaux, bf16_t* aux_out, long ldauxo, hipStream_t st);
void climb_nt4_set(int v);        // climb_set_option 17
void climb_nt4_set_grid(int v);   // follows climb_set_option 9
void climb_nt4_set_probe(int v);  // climb_set_option 18 (measurement only)
