// bf16 multi-head self-attention for the concatenated [text | image] tokens (HF modeling_vilt.py:322-351), forward and
// backward, on v_mfma_f32_32x32x16_bf16.  Throughput-mode twin of attention_f32.hip; same data layout and the same
// "swapped" formulation: every product has the reduction index in MFMA rows of the previous result, so softmax
// statistics are per-lane scalars (+ one xor-32 shuffle) and P / dS feed the next MFMA straight from registers.
//
// One workgroup per (batch, head); S <= 512 keys means a whole head's K and V (S_pad x 64 bf16 = 24.6 KB each at
// S_pad = 192) are LDS-resident: single pass over HBM, the S x S scores never leave the CU.  The images are STREAMED in (see "streamed
// staging" below): the inner loop starts on the first 32-row block while the later ones are still in flight.
//
// LDS image of every [rows][64] operand: 128-B rows, 16-B chunks XOR-swizzled by swz(row); the two operands of a kernel (K and V, or Q
// and dO) are interleaved by 32-row block -- block ib = 8 KB: X rows, then Y rows -- so one set of per-lane addresses serves both
// (Y = X + 4096, an immediate offset).  It is read two ways:
//   * "row" fragments (8 consecutive d of one row)      -> ds_read_b128, conflict-free
//   * "column" fragments (4 consecutive rows at one d)  -> ds_read_b64_tr_b16 (LDS transpose read), conflict-free
// so K serves both S^T = K Q^T and dQ^T = K^T dS^T from one image, likewise V, Q, dO.
#include "common.h"

#define AB_D 64
// measurement builds only (tools/attn_probe.sh): 1 = skip the inner-loop arithmetic, 2 = skip the staging stream, 4 = skip the output stores
#ifndef AB_PROBE
#define AB_PROBE 0
#endif
#ifdef AB_TRACE          // measurement builds only: wave 0 of every forward workgroup logs wall-clock stamps (100 MHz) into a host-provided buffer
__device__ unsigned long long* ab_trace_buf;
extern "C" int climb_attn_set_trace(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(ab_trace_buf), &p, sizeof(p)); }
#define AB_STAMP(i) do { if (tid == 0) ab_trace_buf[(long)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define AB_STAMP(i) do {} while (0)
#endif

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ int aswz(int row) {
  int y = (row >> 1) & 7;
  return ((y & 1) << 2) | (y >> 1);
}
__device__ __forceinline__ bf16x8 to_bf16x8(u32x4 v) {
  union { u32x4 u; bf16x8 b; } c;
  c.u = v;
  return c.b;
}
// accumulator register r of a 32 x 32 block holds matrix row (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)

typedef __attribute__((address_space(3))) void lds_void_t;
#define AB_BLK 8192          // LDS bytes per 32-row block: X half, Y half
#define AB_Y 4096
// row fragment: row (lane&31) of the 32-row half-block at S, d = 16*ks + 8*(lane>>5) .. +7
__device__ __forceinline__ bf16x8 row_frag(const unsigned char* S, int ks, int lane) {
  const int row = lane & 31, c = 2 * ks + (lane >> 5);
  return to_bf16x8(*reinterpret_cast<const u32x4*>(S + row * 128 + ((c ^ aswz(row)) << 4)));
}
// The four column fragments of one 32-row block (st = 0, 1: rows r0 .. +15 / r0 + 16 .. +31; d0 = 0, 32) in ONE asm statement with its own
// lgkmcnt(0).  Used while the staging stream is in flight: hipcc orders every LDS access it cannot prove independent behind pending LDS
// DMA with a vmcnt(0), the transpose-read builtin carries no alias information to prove anything with, and an asm read is invisible to
// that bookkeeping.  (row + 16 keeps the swizzle, so st = 1 is the immediate offset 2048.)
struct ColAddr { unsigned a0, b0, a32, b32; };      // LDS byte addresses of the (rowa, rowb) x (d0 = 0, 32) reads for the X half of block 0
__device__ __forceinline__ ColAddr col_addr(const unsigned char* S, int lane) {
  const int g16 = lane >> 4, i = lane & 15;
  const int rowa = 4 * (g16 >> 1) + (i >> 2), rowb = rowa + 8;
  const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)S;
  ColAddr r;
#pragma unroll
  for (int dd = 0; dd < 2; ++dd) {
    const int col = dd * 32 + (g16 & 1) * 16 + 4 * (i & 3), c = col >> 3, within = (col & 7) * 2;
    const unsigned a = base + rowa * 128 + ((c ^ aswz(rowa)) << 4) + within, bb = base + rowb * 128 + ((c ^ aswz(rowb)) << 4) + within;
    if (dd == 0) { r.a0 = a; r.b0 = bb; } else { r.a32 = a; r.b32 = bb; }
  }
  return r;
}
// f[st][dd]; Y = 0 / 1: the block's X / Y half
template <int Y>
__device__ __forceinline__ void col_frags4(bf16x8 (&f)[2][2], const ColAddr& ca, int blk) {
  const unsigned o = blk * AB_BLK;
  const unsigned a0 = ca.a0 + o, b0 = ca.b0 + o, a32 = ca.a32 + o, b32 = ca.b32 + o;
  s16x4 l00, h00, l01, h01, l10, h10, l11, h11;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %1, %9 offset:%12\n\t"
      "ds_read_b64_tr_b16 %2, %10 offset:%12\n\t"
      "ds_read_b64_tr_b16 %3, %11 offset:%12\n\t"
      "ds_read_b64_tr_b16 %4, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %5, %9 offset:%13\n\t"
      "ds_read_b64_tr_b16 %6, %10 offset:%13\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:%13\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(l00), "=&v"(h00), "=&v"(l01), "=&v"(h01), "=&v"(l10), "=&v"(h10), "=&v"(l11), "=&v"(h11)
      : "v"(a0), "v"(b0), "v"(a32), "v"(b32), "n"(Y * AB_Y), "n"(Y * AB_Y + 2048)
      : "memory");
  union { s16x4 h[2]; bf16x8 b; } u;
  u.h[0] = l00; u.h[1] = h00; f[0][0] = u.b;
  u.h[0] = l01; u.h[1] = h01; f[0][1] = u.b;
  u.h[0] = l10; u.h[1] = h10; f[1][0] = u.b;
  u.h[0] = l11; u.h[1] = h11; f[1][1] = u.b;
}
// bf16 pairs 4 st .. 4 st + 3 (accumulator registers 8 st .. 8 st + 7 of a block) as the MFMA operand of k-step st
__device__ __forceinline__ bf16x8 packed4(const unsigned int (&v)[8], int st) {
  union { unsigned int u[4]; bf16x8 b; } c;
#pragma unroll
  for (int j = 0; j < 4; ++j) c.u[j] = v[4 * st + j];
  return c.b;
}
// the two fragments (d0 = 0, 32) of k-step st only: half the registers, for the kernel that has none to spare
template <int Y, int ST>
__device__ __forceinline__ void col_frags2(bf16x8 (&f)[2], const ColAddr& ca, int blk) {
  const unsigned o = blk * AB_BLK;
  const unsigned a0 = ca.a0 + o, b0 = ca.b0 + o, a32 = ca.a32 + o, b32 = ca.b32 + o;
  s16x4 l0, h0, l1, h1;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %4 offset:%8\n\t"
      "ds_read_b64_tr_b16 %1, %5 offset:%8\n\t"
      "ds_read_b64_tr_b16 %2, %6 offset:%8\n\t"
      "ds_read_b64_tr_b16 %3, %7 offset:%8\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1)
      : "v"(a0), "v"(b0), "v"(a32), "v"(b32), "n"(Y * AB_Y + ST * 2048)
      : "memory");
  union { s16x4 h[2]; bf16x8 b; } u;
  u.h[0] = l0; u.h[1] = h0; f[0] = u.b;
  u.h[0] = l1; u.h[1] = h1; f[1] = u.b;
}
// this lane's B operand for a 32-row register block: row (row0 + lane&31) of a global [.,64] bf16 matrix, 4 k-steps
__device__ __forceinline__ void load_rows(bf16x8 (&f)[4], const bf16_t* __restrict__ src, long ld, int row0, int lane) {
  const bf16_t* p = src + (long)(row0 + (lane & 31)) * ld + (lane >> 5) * 8;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f[ks] = to_bf16x8(*reinterpret_cast<const u32x4*>(p + 16 * ks));
}
// O^T-style accumulator (col = row index of the output matrix, rows = d): a lane owns 4 consecutive d per register group, and its
// partner lane (xor 32) owns the 4 next to them.  The pair trades groups (v_permlane32_swap) so every lane stores 16 contiguous
// bytes and a store instruction covers each row with 32-byte pieces instead of 16-byte ones.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__device__ __forceinline__ void store_acc(bf16_t* __restrict__ dst, const f32x16 (&o)[2], int half, float scale) {
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int g0 = 8 * gp, g1 = 8 * gp + 4;
      const u32x2 r0 = __builtin_amdgcn_permlane32_swap(pack_bf16x2(o[d][g0] * scale, o[d][g0 + 1] * scale),
                                                        pack_bf16x2(o[d][g1] * scale, o[d][g1 + 1] * scale), false, false);
      const u32x2 r1 = __builtin_amdgcn_permlane32_swap(pack_bf16x2(o[d][g0 + 2] * scale, o[d][g0 + 3] * scale),
                                                        pack_bf16x2(o[d][g1 + 2] * scale, o[d][g1 + 3] * scale), false, false);
      u32x4 v;
      v.x = r0.x; v.y = r1.x; v.z = r0.y; v.w = r1.y;
      *reinterpret_cast<u32x4*>(dst + d * 32 + 8 * (2 * gp + half)) = v;
    }
}
// sum / max over the two lanes (xor 32) that share a matrix column
__device__ __forceinline__ float pair_max(float x) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
__device__ __forceinline__ float pair_sum(float x) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// ------------------------------------------------------------------------------------------------------ streamed staging
// The two LDS images of a workgroup are consumed in 32-row blocks by the inner loop, so they are STREAMED: every wave issues its
// share of both images in block order (unit = 8 rows = one 1-KB wave-instruction; wave w owns units w, w + NW, ...: a "round"; the X and
// Y unit of a round go out back to back), and before block ib a wave waits only until the loads of LATER blocks are the ones still
// outstanding (counted vmcnt) -- the raw s_barrier that follows makes every wave's share of the block visible.  The first MFMA starts
// when 1/NB of the data has arrived instead of all of it; with every workgroup of the launch in lock step that is the difference between
// "load, then compute" and the two overlapped.
#define AB_LOG2E 1.44269504088896340736f
#define AB_NEG (-3.0e38f)
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int N> __device__ __forceinline__ void ab_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void ab_wait_vm_pairs(int pairs) {      // at most 2 * pairs loads outstanding; `pairs` is wave-uniform
  switch (pairs) {
    case 0: ab_wait_vm<0>(); break;
    case 1: ab_wait_vm<2>(); break;
    case 2: ab_wait_vm<4>(); break;
    case 3: ab_wait_vm<6>(); break;
    case 4: ab_wait_vm<8>(); break;
    case 5: ab_wait_vm<10>(); break;
    case 6: ab_wait_vm<12>(); break;
    default: ab_wait_vm<14>(); break;
  }
}
__device__ __forceinline__ void negate_rows(bf16x8 (&f)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    union { bf16x8 b; u32x4 u; } c;
    c.b = f[k];
    c.u ^= 0x80008000u;
    f[k] = c.b;
  }
}
// pins hipcc's s_waitcnt for register loads to this point
__device__ __forceinline__ void use_rows(bf16x8 (&f)[4]) { asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); }
// The DMA is the MUBUF form (buffer_load_dwordx4 ... lds), not global_load_lds: hipcc books a FLAT-encoded LDS load as "pending flat" and
// then drains vmcnt to 0 for ANY register load that is waited on while one is in flight; buffer loads are counted exactly.
//
// Issue order.  Asking for everything at once (all rounds up front, every workgroup of the launch at the same moment) puts ~60 MB in
// flight and the FIRST data a wave needs -- its Q rows, block 0 -- arrives after 5-8 us of queueing (tools/attn_trace.py).  So only
// AB_PRO rounds go out before the loop (compile-time count: hipcc's own vmcnt arithmetic for the register loads stays exact) and the
// rest are topped up inside it, two blocks ahead of the one being computed.
#define AB_PRO 3
struct Streamer {
  __amdgpu_buffer_rsrc_t rx, ry;
  unsigned char* Ls;
  int ldx2, ldy2, nunits, wave, nwaves, lane;      // row strides in bytes
  int issued, need, cover, total;                  // rounds issued / rounds block `ib` needs / units those cover / rounds there are

  __device__ __forceinline__ void init(unsigned char* ls, const bf16_t* Xg, long ldx, const bf16_t* Yg, long ldy, int nunits_, int wave_,
                                       int nwaves_, int lane_) {
    rx = __builtin_amdgcn_make_buffer_rsrc((void*)Xg, 0, 0x7fffffff, 0x00020000);
    ry = __builtin_amdgcn_make_buffer_rsrc((void*)Yg, 0, 0x7fffffff, 0x00020000);
    Ls = ls; ldx2 = (int)ldx * 2; ldy2 = (int)ldy * 2; nunits = nunits_; wave = wave_; nwaves = nwaves_; lane = lane_;
    issued = need = cover = 0;
    total = (nunits + nwaves - 1) / nwaves;        // >= 4 >= AB_PRO for every launch shape of this file
  }
  // round r: unit wave + r * nwaves of both images (past the end: the last unit again -- same bytes to the same place)
  __device__ __forceinline__ void round(int r) {
    int j = wave + r * nwaves;
    j = j < nunits ? j : nunits - 1;
    const int slot = j * 64 + lane, row = slot >> 3, c = (slot & 7) ^ aswz(row);
    unsigned char* dst = Ls + (j >> 2) * AB_BLK + (j & 3) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)dst, 16, row * ldx2 + c * 16, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_void_t*)(dst + AB_Y), 16, row * ldy2 + c * 16, 0, 0, 0);
  }
  __device__ __forceinline__ void prologue() {
#pragma unroll
    for (int r = 0; r < AB_PRO; ++r) round(r);
    issued = AB_PRO;
  }
  // before block ib is read: wait until only rounds block ib does not need are outstanding (block ib = units 4 ib .. 4 ib + 3, whole once
  // every wave has landed ceil(4 (ib + 1) / NW) rounds), make that visible, then top the stream up to what block ib + 2 needs
  __device__ __forceinline__ void block(int ib) {
    while (cover < 4 * (ib + 1) && need < total) { cover += nwaves; ++need; }
    while (issued < need) round(issued++);         // (only if AB_PRO were too small for block 0)
    ab_wait_vm_pairs(issued - need);
    __builtin_amdgcn_s_barrier();
    int need2 = need, cover2 = cover;
    while (cover2 < 4 * (ib + 3) && need2 < total) { cover2 += nwaves; ++need2; }
    while (issued < need2) round(issued++);
  }
};
// Two things keep hipcc from draining vmcnt to 0 in front of a ds_read while DMAs are in flight (SIInsertWaitcnts orders LDS reads
// behind LDS DMA unless alias information says otherwise): no __restrict__ on anything that touches LDS here (its alias scopes make
// every DMA a tracked store the read "may alias"), and every LDS read is a typed vector load (carries TBAA; a HIP float4 struct copy
// carries none and gets the conservative wait).
// a float vector (key bias, lse, delta) of n elements into LDS, raw, 64 elements per wave-instruction; lanes past n read 0 (buffer range
// check) into the padding of the LDS vector.  Nothing in these kernels WRITES LDS with ds_write while the stream is in flight: hipcc
// orders a DS store behind every pending LDS DMA (vmcnt(0)), which would serialise the very thing being overlapped.
#define AB_VEC(S_pad) (((S_pad) + 63) & ~63)
__device__ __forceinline__ void vec_issue(float* dst_s, const float* src, int n, int wave, int nwaves, int lane) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n * 4, 0x00020000);
  for (int u = wave; u < (n + 63) / 64; u += nwaves)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(dst_s + u * 64), 4, (u * 64 + lane) * 4, 0, 0, 0);
}
// ------------------------------------------------------------------------------------------------------ forward
// dynamic LDS: NB blocks of (K rows | V rows) | bias_s[AB_VEC(S_pad)] f32
// One wave owns 64 queries (two 32-query blocks, NW = ceil(NB / 2) waves): every K row fragment and every V column fragment it reads
// from LDS feeds two MFMA chains, and -- the reason for the shape -- at ~160 VGPRs and 3 waves per workgroup, three workgroups are
// resident per CU (one wave per 32 queries needs <= 96 VGPRs for that and does not get there; its 768 workgroups then run as 1.5 rounds).
// Softmax: v_exp_f32 is base 2.  Scores stay in MFMA units (S / scale): s starts at bias[key] / scale (exact, scale = 1/8; a masked key is
// -inf from there on) and the running maximum m_run is in the same units, so the exponent is ONE fma per pair,
// (s - m_run) * (scale log2 e).  The maximum is LAZY: the
// reference point m_run only moves when some row's block maximum exceeds it by more than 2^8 (a wave-uniform branch), so the 32-register
// rescale of O and its exp are skipped in almost every block; P <= 256 then, which costs bf16 nothing (relative precision) and the fp32
// sums nothing either.  The row sum stays per lane (each lane sums its own 16 keys of a block) and the two halves meet once, at the end.
#define AB_TAU 5.545f            // 8 ln 2
#define AB_M0 (-1.0e30f)         // initial reference point: far below any score, and -AB_M0 * scale log2 e still finite

struct SoftmaxRow { float m_run, nm2, l_run; };
// one 32x32 score block: lazy maximum, P as bf16 pairs (MFMA k-slot pair i of step st = i >> 2), row sums
__device__ __forceinline__ void softmax_block(const f32x16& s, f32x16 (&o)[2], SoftmaxRow& st, unsigned int (&pk)[8], float c1, float tau) {
  float mx = -3.0e38f;
#pragma unroll
  for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
  mx = pair_max(mx);
  if (__builtin_amdgcn_ballot_w64(mx > st.m_run + tau) != 0) {
    const float m_new = fmaxf(st.m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((st.m_run - m_new) * c1);
    st.l_run *= alpha;
    st.m_run = m_new;
    st.nm2 = -m_new * c1;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
  }
  f32x2 sum2 = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x2 e = f32x2{s[2 * i], s[2 * i + 1]} * c1 + st.nm2;
    const f32x2 pe = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    sum2 += pe;
    pk[i] = pack_bf16x2(pe.x, pe.y);
  }
  st.l_run += sum2.x + sum2.y;
}
__device__ __forceinline__ void bias_init(f32x16& s, const float* bias_blk, float inv_scale) {
  const f32x4* bp = reinterpret_cast<const f32x4*>(bias_blk);    // registers 4g .. 4g+3 <-> keys 8g + 4 half + {0..3} of the block
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 v = bp[2 * g];
    const f32x2 a0 = f32x2{v.x, v.y} * inv_scale, a1 = f32x2{v.z, v.w} * inv_scale;
    s[4 * g] = a0.x; s[4 * g + 1] = a0.y; s[4 * g + 2] = a1.x; s[4 * g + 3] = a1.y;
  }
}

template <int QB>
__global__ __launch_bounds__(576) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ key_bias,
                                                            bf16_t* __restrict__ ctx, float* __restrict__ lse_out, int S_pad, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* bias_s = reinterpret_cast<float*>(smem + S_pad * 256);
  const int H = heads * AB_D, ld = 3 * H;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l31 = lane & 31;
  const int nwaves = blockDim.x >> 6;
  const bf16_t* Qg = qkv + (long)b * S_pad * ld + h * AB_D;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Qg + 2 * H;
  const int NB = S_pad / 32;
  // oldest in the vmcnt queue: what the first MFMA needs from registers, then the key bias, then the stream
  int qbs[QB];                   // QB = 2, odd NB: the last wave computes its one block twice and stores it once
#pragma unroll
  for (int q = 0; q < QB; ++q) qbs[q] = QB * wid + q < NB ? QB * wid + q : QB * wid;
  AB_STAMP(0);
  bf16x8 qf[QB][4];
#pragma unroll
  for (int q = 0; q < QB; ++q) load_rows(qf[q], Qg, ld, qbs[q] * 32, lane);
  __builtin_amdgcn_sched_barrier(0);
  vec_issue(bias_s, key_bias + (long)b * S_pad, S_pad, wid, nwaves, lane);
  Streamer sw;
  sw.init(smem, Kg, ld, Vg, ld, S_pad / 8, wid, nwaves, lane);
  if (!(AB_PROBE & 2)) sw.prologue();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < QB; ++q) use_rows(qf[q]);          // hipcc's wait for the Q rows lands HERE (straight-line code, exact count), not inside the loop
  AB_STAMP(1);
  const ColAddr va = col_addr(smem, lane);
  const float c1 = scale * AB_LOG2E, inv_scale = 1.0f / scale, tau = AB_TAU * inv_scale;
  f32x16 o[QB][2];
  SoftmaxRow sr[QB];
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    sr[q].m_run = AB_M0; sr[q].nm2 = -AB_M0 * c1; sr[q].l_run = 0.f;
  }
#pragma unroll
  for (int q = 0; q < QB; ++q)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[q][d][r] = 0.f;
  for (int kb = 0; kb < NB; ++kb) {
    if (!(AB_PROBE & 2)) sw.block(kb);
    if (kb < 10) AB_STAMP(2 + kb);
    if (AB_PROBE & 1) continue;
    f32x16 s[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) bias_init(s[q], bias_s + kb * 32 + 4 * half, inv_scale);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kf = row_frag(smem + kb * AB_BLK, ks, lane);
#pragma unroll
      for (int q = 0; q < QB; ++q) s[q] = CLIMB_MFMA_H16(kf, qf[q][ks], s[q], 0, 0, 0);
    }
    unsigned int pk[QB][8];
#pragma unroll
    for (int q = 0; q < QB; ++q) softmax_block(s[q], o[q], sr[q], pk[q], c1, tau);
    __builtin_amdgcn_sched_barrier(0);
    // O^T[d][q] += V^T[d][key] P^T[key][q]; MFMA k-slot (half, j) <-> key kb*32 + 16*st + 8*(j>>2) + 4*half + (j&3)
    bf16x8 vf[2];
    col_frags2<1, 0>(vf, va, kb);
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      o[q][0] = CLIMB_MFMA_H16(vf[0], packed4(pk[q], 0), o[q][0], 0, 0, 0);
      o[q][1] = CLIMB_MFMA_H16(vf[1], packed4(pk[q], 0), o[q][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    col_frags2<1, 1>(vf, va, kb);
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      o[q][0] = CLIMB_MFMA_H16(vf[0], packed4(pk[q], 1), o[q][0], 0, 0, 0);
      o[q][1] = CLIMB_MFMA_H16(vf[1], packed4(pk[q], 1), o[q][1], 0, 0, 0);
    }
  }
  AB_STAMP(12);
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    if (q > 0 && qbs[q] == qbs[0]) break;
    const int qb = qbs[q];
    const float l_tot = pair_sum(sr[q].l_run);
    // a row with every key masked (does not occur: the first text token is always valid) gives 0, not NaN, here and in the backward
    if (!(AB_PROBE & 4) || l_tot == 12345.f)
    store_acc(ctx + ((long)b * S_pad + qb * 32 + l31) * H + h * AB_D, o[q], half, l_tot > 0.f ? 1.0f / l_tot : 0.f);
    if (half == 0) lse_out[((long)b * heads + h) * S_pad + qb * 32 + l31] = l_tot > 0.f ? sr[q].m_run * scale + __logf(l_tot) : 0.f;
  }
  AB_STAMP(13);
#ifdef AB_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  AB_STAMP(14);
#endif
}

static int g_attn_qb = 0;      // measurement knob (climb_set_option 12): force 1 or 2 query blocks per wave
void climb_attn_set_qb(int v) { g_attn_qb = v; }     // C++ linkage: library-internal (climb_set_option key 12)
extern "C" int climb_attn_fwd_bf16(const void* qkv, const float* key_bias, void* ctx, float* lse, int B, int S_pad, int heads, int head_dim,
                                   void* stream) {
  if (head_dim != AB_D || S_pad % 32 || S_pad <= 0 || S_pad > 512) return CLIMB_EUNSUPPORTED;
  size_t lds = (size_t)S_pad * 256 + (size_t)AB_VEC(S_pad) * 4;
  static size_t lds_set = 0;          // raise the dynamic-LDS cap once per size (not a stream operation: keep it out of graph capture)
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_fwd_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    lds_set = lds;
  }
  const int NB = S_pad / 32;
  const float scale = 1.0f / sqrtf((float)head_dim);
  // query blocks per wave: 2 where the workgroup's LDS lets three (NB <= 6) sit on a CU -- or where one wave per block would need more than
  // 9 waves; 1 in between (measured at NB = 9: 50 vs 57 us per layer)
  const int qb = g_attn_qb ? g_attn_qb : ((NB <= 6 || NB > 9) ? 2 : 1);
  if (qb == 1 && NB <= 9)
    hipLaunchKernelGGL(attn_fwd_bf16_kernel<1>, dim3(B * heads), dim3(64 * NB), lds, (hipStream_t)stream, (const bf16_t*)qkv, key_bias,
                       (bf16_t*)ctx, lse, S_pad, heads, scale);
  else
    hipLaunchKernelGGL(attn_fwd_bf16_kernel<2>, dim3(B * heads), dim3(64 * ((NB + 1) / 2)), lds, (hipStream_t)stream, (const bf16_t*)qkv, key_bias,
                       (bf16_t*)ctx, lse, S_pad, heads, scale);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// ------------------------------------------------------------------------------------------------------ backward
// PHASE 0 (dQ):      outer = query block (Q, dO rows in registers), inner = keys (K, V images in LDS)
//     S^T = K Q^T, dP^T = V dO^T, P^T = exp(S^T*scale + bias - lse), dS^T = P^T (dP^T - delta) scale, dQ^T += K^T dS^T
// PHASE 1 (dK, dV):  outer = key block (K, V rows in registers), inner = queries (Q, dO images in LDS)
//     S = Q K^T, dP = dO V^T, P, dS;  dV^T += dO^T P,  dK^T += Q^T dS
// delta[q] = sum_d dO[q][d] O[q][d] (the softmax-backward row term) is computed by PHASE 0 itself, whose waves already hold
// their dO rows in registers, and written out for PHASE 1, which needs it for every query.
// Base-2 exponent like the forward; the factor `scale` of dS is a power of two here (1/8), applied to the fp32 accumulators at the store
// instead of to every dS element: bit-identical, 16 multiplies per block cheaper.  Images are streamed (see above), 8 rounds:
// NW = ceil(NB / 2) waves, two outer blocks per wave.
// dynamic LDS: NB blocks of (X rows | Y rows) | bias_s | lse_s | delta_s (AB_VEC(S_pad) floats each)
// launch bound 576 (= 9 waves, 3 per SIMD) caps the kernel at 168 VGPRs: three workgroups stay resident per CU at S_pad = 192
// FUSED (r03): both phases in ONE launch, one after the other in the same workgroup.  The second phase then finds K, V, Q, dO of its
// head where the first one just read them (the XCD's L2 / the MALL instead of HBM), delta goes from phase 0 to phase 1 through LDS
// instead of through memory, and a launch per layer disappears.
template <int PHASE, bool FUSED>
__device__ __forceinline__ void attn_bwd_body(unsigned char* smem, const bf16_t* __restrict__ qkv, const float* __restrict__ key_bias,
                                              const bf16_t* __restrict__ dctx, const bf16_t* __restrict__ ctx,
                                              const float* __restrict__ lse, float* __restrict__ delta,
                                              bf16_t* __restrict__ dqkv, int S_pad, int heads, float scale) {
  float* bias_s = reinterpret_cast<float*>(smem + S_pad * 256);
  float* lse_s = bias_s + AB_VEC(S_pad);
  float* delta_s = lse_s + AB_VEC(S_pad);
  const int H = heads * AB_D, ld = 3 * H;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l31 = lane & 31;
  const int nwaves = blockDim.x >> 6;
  const bf16_t* Qg = qkv + (long)b * S_pad * ld + h * AB_D;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Qg + 2 * H;
  const bf16_t* dOg = dctx + (long)b * S_pad * H + h * AB_D;
  const bf16_t* Og = ctx + (long)b * S_pad * H + h * AB_D;
  const long rowv = ((long)b * heads + h) * S_pad;     // this head's row of lse / delta
  const int NB = S_pad / 32;
  int ob = wid;                                   // nwaves <= NB
  bf16x8 f1[4], f2[4], fo[4];
  if (PHASE == 0) { load_rows(f1, Qg, ld, ob * 32, lane); load_rows(f2, dOg, H, ob * 32, lane); load_rows(fo, Og, H, ob * 32, lane); }
  else            { load_rows(f1, Kg, ld, ob * 32, lane); load_rows(f2, Vg, ld, ob * 32, lane); }
  // per-row scalar of this lane's outer row: lse (phase 0: the row is a query) or key bias (phase 1: a key)
  float my_c = PHASE == 0 ? lse[rowv + ob * 32 + l31] : key_bias[(long)b * S_pad + ob * 32 + l31];
  __builtin_amdgcn_sched_barrier(0);
  // the vectors every lane needs for the INNER index: LDS, raw
  if (PHASE == 0) vec_issue(bias_s, key_bias + (long)b * S_pad, S_pad, wid, nwaves, lane);
  else { vec_issue(lse_s, lse + rowv, S_pad, wid, nwaves, lane); if (!FUSED) vec_issue(delta_s, delta + rowv, S_pad, wid, nwaves, lane); }
  Streamer sw;
  if (PHASE == 0) sw.init(smem, Kg, ld, Vg, ld, S_pad / 8, wid, nwaves, lane);
  else            sw.init(smem, Qg, ld, dOg, H, S_pad / 8, wid, nwaves, lane);
  if (!(AB_PROBE & 2)) sw.prologue();
  __builtin_amdgcn_sched_barrier(0);
  use_rows(f1);                  // hipcc's waits for the register operands land here (exact counts), not inside the loops
  use_rows(f2);
  if (PHASE == 0) use_rows(fo);
  else negate_rows(f2);
  asm volatile("" : "+v"(my_c));
  const ColAddr xa = col_addr(smem, lane);
  const float c1 = scale * AB_LOG2E;
  for (bool first = true; ob < NB; first = false) {
    f32x16 acc1[2], acc2[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[d][r] = acc2[d][r] = 0.f;
    const int my = ob * 32 + l31;
    float my_delta = 0.f;
    if (PHASE == 0) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) my_delta += (float)fo[ks][e] * (float)f2[ks][e];
      my_delta = pair_sum(my_delta);
    }
    // The per-row vectors ride in the accumulators: s starts at the INNER index's term, bias[key] / scale (phase 0) or -lse[query] / scale
    // (phase 1) -- exact, scale = 1/8 -- and the lane's own term (-lse[query] / bias[key]) joins in the exponent's fma:
    // P = 2^(s * scale log2 e + own log2 e).  Phase 1 also starts dP at
    // delta[query] and multiplies by the NEGATED V rows (a sign-bit flip, exact): dp = delta - dO V^T, dS comes out negated, and the
    // sign goes into the factor dK is stored with.  No vector is live next to s / dp, which is what keeps phase 1 inside 168 VGPRs.
    const float inv_scale = 1.0f / scale;
    const float is = PHASE == 0 ? inv_scale : -inv_scale;
    const float k2 = (PHASE == 0 ? -my_c : fmaxf(my_c, AB_NEG)) * AB_LOG2E;      // this lane's own term, added in the exponent's fma
    for (int ib = 0; ib < NB; ++ib) {
      if (first && !(AB_PROBE & 2)) sw.block(ib);
      if (AB_PROBE & 1) continue;
      f32x16 s, dp;
      {
        const f32x4* ap = reinterpret_cast<const f32x4*>((PHASE == 0 ? bias_s : lse_s) + ib * 32 + 4 * half);
        const f32x4* dl = reinterpret_cast<const f32x4*>(delta_s + ib * 32 + 4 * half);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = ap[2 * g];
          const f32x2 a0 = f32x2{v.x, v.y} * is, a1 = f32x2{v.z, v.w} * is;
          s[4 * g] = a0.x; s[4 * g + 1] = a0.y; s[4 * g + 2] = a1.x; s[4 * g + 3] = a1.y;
          if (PHASE == 1) {
            const f32x4 dd = dl[2 * g];
            dp[4 * g] = dd.x; dp[4 * g + 1] = dd.y; dp[4 * g + 2] = dd.z; dp[4 * g + 3] = dd.w;
          } else {
            dp[4 * g] = dp[4 * g + 1] = dp[4 * g + 2] = dp[4 * g + 3] = 0.f;
          }
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = CLIMB_MFMA_H16(row_frag(smem + ib * AB_BLK, ks, lane), f1[ks], s, 0, 0, 0);
        dp = CLIMB_MFMA_H16(row_frag(smem + ib * AB_BLK + AB_Y, ks, lane), f2[ks], dp, 0, 0, 0);
      }
      // P and dS go to bf16 pairs as they are made (MFMA k-slot pair i of step st = i >> 2): 16 live registers, not 32
      unsigned int pk[8], dsk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x2 e = f32x2{s[2 * i], s[2 * i + 1]} * c1 + k2;
        const f32x2 pe = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
        f32x2 d = {dp[2 * i], dp[2 * i + 1]};
        if (PHASE == 0) d -= my_delta;
        d *= pe;
        if (PHASE == 1) pk[i] = pack_bf16x2(pe.x, pe.y);
        dsk[i] = pack_bf16x2(d.x, d.y);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (PHASE == 0) {
        bf16x8 xf[2][2];
        col_frags4<0>(xf, xa, ib);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const bf16x8 dsf = packed4(dsk, st);
          acc1[0] = CLIMB_MFMA_H16(xf[st][0], dsf, acc1[0], 0, 0, 0);
          acc1[1] = CLIMB_MFMA_H16(xf[st][1], dsf, acc1[1], 0, 0, 0);
        }
      } else {
        // phase 1 has no registers to spare (64 accumulators + 32 K/V rows): one k-step's fragments at a time
        bf16x8 f[2];
        col_frags2<0, 0>(f, xa, ib);
        acc1[0] = CLIMB_MFMA_H16(f[0], packed4(dsk, 0), acc1[0], 0, 0, 0);
        acc1[1] = CLIMB_MFMA_H16(f[1], packed4(dsk, 0), acc1[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        col_frags2<1, 0>(f, xa, ib);
        acc2[0] = CLIMB_MFMA_H16(f[0], packed4(pk, 0), acc2[0], 0, 0, 0);
        acc2[1] = CLIMB_MFMA_H16(f[1], packed4(pk, 0), acc2[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        col_frags2<0, 1>(f, xa, ib);
        acc1[0] = CLIMB_MFMA_H16(f[0], packed4(dsk, 1), acc1[0], 0, 0, 0);
        acc1[1] = CLIMB_MFMA_H16(f[1], packed4(dsk, 1), acc1[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        col_frags2<1, 1>(f, xa, ib);
        acc2[0] = CLIMB_MFMA_H16(f[0], packed4(pk, 1), acc2[0], 0, 0, 0);
        acc2[1] = CLIMB_MFMA_H16(f[1], packed4(pk, 1), acc2[1], 0, 0, 0);
      }
    }
    bf16_t* orow = dqkv + ((long)b * S_pad + my) * ld + h * AB_D;
    // FUSED: delta goes to phase 1 through LDS.  Written here -- the staging stream ended inside the first outer block and the next block's
    // register loads are not yet in flight, so the vmcnt(0) hipcc puts in front of a DS store that may follow LDS DMA waits for nothing
    if (FUSED && PHASE == 0 && half == 0) delta_s[my] = my_delta;
    // the next outer block's register operands are requested BEFORE this block's results are stored: they are not queued behind 19-38 MB
    // of stores, and their latency overlaps the store issue
    const float was_c = my_c;
    ob += nwaves;
    if (ob < NB) {
      if (PHASE == 0) { load_rows(f1, Qg, ld, ob * 32, lane); load_rows(f2, dOg, H, ob * 32, lane); load_rows(fo, Og, H, ob * 32, lane); }
      else            { load_rows(f1, Kg, ld, ob * 32, lane); load_rows(f2, Vg, ld, ob * 32, lane); }
      my_c = PHASE == 0 ? lse[rowv + ob * 32 + l31] : key_bias[(long)b * S_pad + ob * 32 + l31];
    }
    __builtin_amdgcn_sched_barrier(0);
    if ((AB_PROBE & 4) && was_c != 12345.f) {}
    else if (PHASE == 0) {
      store_acc(orow, acc1, half, scale);
      if (half == 0 && !FUSED) delta[rowv + my] = my_delta;
    } else {
      store_acc(orow + H, acc1, half, -scale);
      store_acc(orow + 2 * H, acc2, half, 1.0f);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (ob < NB) {
      use_rows(f1);
      use_rows(f2);
      if (PHASE == 0) use_rows(fo);
      else negate_rows(f2);
      asm volatile("" : "+v"(my_c));
    }
  }
}

template <int PHASE>
__global__ __launch_bounds__(576) void attn_bwd_bf16_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ key_bias,
                                                            const bf16_t* __restrict__ dctx, const bf16_t* __restrict__ ctx,
                                                            const float* __restrict__ lse, float* __restrict__ delta,
                                                            bf16_t* __restrict__ dqkv, int S_pad, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  attn_bwd_body<PHASE, false>(smem, qkv, key_bias, dctx, ctx, lse, delta, dqkv, S_pad, heads, scale);
}
__global__ __launch_bounds__(576) void attn_bwd_bf16_fused_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ key_bias,
                                                                  const bf16_t* __restrict__ dctx, const bf16_t* __restrict__ ctx,
                                                                  const float* __restrict__ lse, float* __restrict__ delta,
                                                                  bf16_t* __restrict__ dqkv, int S_pad, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  attn_bwd_body<0, true>(smem, qkv, key_bias, dctx, ctx, lse, delta, dqkv, S_pad, heads, scale);
  __syncthreads();          // every wave is done with the K / V images and has left its rows' delta in LDS
  attn_bwd_body<1, true>(smem, qkv, key_bias, dctx, ctx, lse, delta, dqkv, S_pad, heads, scale);
}
// ------------------------------------------------------------------------------------------------------ backward, SINGLE PASS (r04)
// S, dP, P and dS of a (key block, query block) pair are computed ONCE and feed all three products: 20 MFMAs and 16 exponentials per lane
// and pair against 28 / 32 in the two phases above, and every operand is read from HBM once (151 MB per layer at B = 64, S_pad = 192: the
// algorithmic figure).  One workgroup per (batch, head), one wave per KEY block j (NB <= 6 waves): the wave keeps K_j / V_j rows and the
// K_j^T fragments in registers and owns dK_j^T, dV_j^T (64 accumulators); Q and dO are LDS images (X | Y halves of a block, as above).
//   * dQ needs dS with the KEY index inside a lane (dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]), but the wave holds dS[q][key] with the key
//     on its lanes: the 32 x 32 block is turned through a 2 KB per-wave LDS tile (4 ds_write_b64, 4 ds_read_b64_tr_b16) -- not recomputed.
//   * dQ_i sums over ALL key blocks, i.e. over all waves: the accumulators of the NB query blocks live in LDS (fp32, raw register layout, 8 KB
//     each) and the waves walk the query blocks ROTATED -- in step t wave j works on query block (j + t) mod NB -- so no two waves touch the same
//     block in a step: read 32 registers, 4 MFMAs on top, write them back; one s_barrier per step.  (Step 0 writes without reading.)
//   * step 0 needs nothing from another wave (wave j brings in Q_j / dO_j itself and computes delta_j from the O_j / dO_j rows it loads), so
//     there is no barrier and no cross-wave wait before the first MFMA.
// LDS: NB x (8 KB images + 8 KB dQ + 2 KB turn) + lse / delta vectors = 110 KB at S_pad = 192: one workgroup per CU, which is why this
// kernel takes S_pad <= 192 only; longer sequences stay on the two-phase kernel above.  ~210 VGPRs at 1.5 waves per SIMD.
// The arithmetic is phase 1's (lanes = keys: the key bias is the lane's own term, -lse[query] / scale and delta[query] are the accumulators'
// initial values, V rows sign-flipped so that dS comes out negated); dQ takes the same negated dS, hence its -scale at the store.
#define AB1_TURN 2048
__global__ __launch_bounds__(384) void attn_bwd_bf16_1p_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ key_bias,
                                                               const bf16_t* __restrict__ dctx, const bf16_t* __restrict__ ctx,
                                                               const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int S_pad, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int NB = S_pad / 32;
  unsigned char* dqs = smem + NB * AB_BLK;                        // fp32 dQ^T accumulators, raw: float4 (dd * 4 + g) of lane l at ((dd * 4 + g) * 64 + l) * 16
  unsigned char* tsc = dqs + NB * AB_BLK;                         // per-wave turn tiles
  float* lse_s = reinterpret_cast<float*>(tsc + NB * AB1_TURN);
  float* delta_s = lse_s + AB_VEC(S_pad);
  const int H = heads * AB_D, ld = 3 * H;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l31 = lane & 31;
  const bf16_t* Qg = qkv + (long)b * S_pad * ld + h * AB_D;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Qg + 2 * H;
  const bf16_t* dOg = dctx + (long)b * S_pad * H + h * AB_D;
  const bf16_t* Og = ctx + (long)b * S_pad * H + h * AB_D;
  const long rowv = ((long)b * heads + h) * S_pad;
  const int my = w * 32 + l31;                                    // this lane's key; also the query row whose delta it computes
  bf16x8 f1[4], f2[4];
  float my_c;                                                     // this lane's key bias
  {
    bf16x8 fo[4], fd[4];
    load_rows(f1, Kg, ld, w * 32, lane);
    load_rows(f2, Vg, ld, w * 32, lane);
    load_rows(fo, Og, H, w * 32, lane);
    load_rows(fd, dOg, H, w * 32, lane);
    my_c = key_bias[(long)b * S_pad + my];
    float my_lse = lse[rowv + my];
    __builtin_amdgcn_sched_barrier(0);
    // this wave's own Q / dO block: 4 units of 8 rows each, X (Q) and Y (dO) back to back
    {
      const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)Qg, 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dOg, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int slot = (4 * w + u) * 64 + lane, row = slot >> 3, c = (slot & 7) ^ aswz(row);
        unsigned char* dst = smem + w * AB_BLK + u * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void_t*)dst, 16, row * (ld * 2) + c * 16, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_void_t*)(dst + AB_Y), 16, row * (H * 2) + c * 16, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    ab_wait_vm<0>();
    use_rows(f1); use_rows(f2); use_rows(fo); use_rows(fd);
    asm volatile("" : "+v"(my_c), "+v"(my_lse));
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) dsum += (float)fo[ks][e] * (float)fd[ks][e];
    const float my_delta = pair_sum(dsum);
    if (half == 0) { delta_s[my] = my_delta; lse_s[my] = my_lse; }
  }
  // K_j^T fragments: the K rows go through LDS once (a 4 KB row image in this wave's own dQ block, which step 0 overwrites) and come back transposed
  bf16x8 kT[2][2];
  {
    unsigned char* ksc = dqs + w * AB_BLK;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      union { bf16x8 b; u32x4 u; } c;
      c.b = f1[ks];
      *reinterpret_cast<u32x4*>(ksc + l31 * 128 + (((2 * ks + half) ^ aswz(l31)) << 4)) = c.u;
    }
    const ColAddr ka = col_addr(ksc, lane);
    col_frags4<0>(kT, ka, 0);
  }
  negate_rows(f2);
  const ColAddr xa = col_addr(smem, lane);
  // turn tile [32 keys][32 queries] bf16, 64-byte rows, 16-byte chunks XOR (row >> 2) & 3
  const unsigned tbase = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)(tsc + w * AB1_TURN);
  unsigned tw[4], ta, tb;
#pragma unroll
  for (int g = 0; g < 4; ++g) tw[g] = tbase + l31 * 64 + ((g ^ ((l31 >> 2) & 3)) << 4) + 8 * half;
  {
    const int g16 = lane >> 4, i = lane & 15;
    const int rowa = 4 * (g16 >> 1) + (i >> 2), rowb = rowa + 8, col = (g16 & 1) * 16 + 4 * (i & 3);
    ta = tbase + rowa * 64 + (((col >> 3) ^ ((rowa >> 2) & 3)) << 4) + (col & 7) * 2;
    tb = tbase + rowb * 64 + (((col >> 3) ^ ((rowb >> 2) & 3)) << 4) + (col & 7) * 2;
  }
  const float c1 = scale * AB_LOG2E, is = -1.0f / scale, k2 = fmaxf(my_c, AB_NEG) * AB_LOG2E;
  f32x16 accK[2], accV[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) accK[d][r] = accV[d][r] = 0.f;
  // transposed 8-byte reads the compiler schedules and counts itself (nothing is in flight on the DMA path after the prologue)
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
#define AB1_TR(addr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(size_t)(addr))
  // S and dP of step t + 1 are computed INSIDE step t (the images never change): with 1.5 waves per SIMD there is no other wave to cover the
  // LDS round trips and the exponentials of a pair, so the pair's own next products are what the matrix pipe runs meanwhile
  f32x16 s, dp;
#define AB1_SDP(IB)                                                                                                                    \
  {                                                                                                                                    \
    const f32x4* ap = reinterpret_cast<const f32x4*>(lse_s + (IB) * 32 + 4 * half);                                                    \
    const f32x4* dl = reinterpret_cast<const f32x4*>(delta_s + (IB) * 32 + 4 * half);                                                  \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                                    \
      const f32x4 v = ap[2 * g], dd = dl[2 * g];                                                                                       \
      s[4 * g] = v.x * is; s[4 * g + 1] = v.y * is; s[4 * g + 2] = v.z * is; s[4 * g + 3] = v.w * is;                                  \
      dp[4 * g] = dd.x; dp[4 * g + 1] = dd.y; dp[4 * g + 2] = dd.z; dp[4 * g + 3] = dd.w;                                              \
    }                                                                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                                 \
      s = CLIMB_MFMA_H16(row_frag(smem + (IB) * AB_BLK, ks, lane), f1[ks], s, 0, 0, 0);                                                \
      dp = CLIMB_MFMA_H16(row_frag(smem + (IB) * AB_BLK + AB_Y, ks, lane), f2[ks], dp, 0, 0, 0);                                       \
    }                                                                                                                                  \
  }
  AB1_SDP(w)                      // step 0 works on this wave's own block: nothing of another wave is needed yet
  __syncthreads();                // every wave's Q / dO block, lse and delta are in LDS
  for (int t = 0; t < NB; ++t) {
    int ib = w + t;
    ib = ib >= NB ? ib - NB : ib;
    int ibn = ib + 1;
    ibn = ibn >= NB ? 0 : ibn;
    f32x4* dqp = reinterpret_cast<f32x4*>(dqs + ib * AB_BLK) + lane;
    unsigned int pk[8], dsk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x2 e = f32x2{s[2 * i], s[2 * i + 1]} * c1 + k2;
      const f32x2 pe = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
      const f32x2 d = f32x2{dp[2 * i], dp[2 * i + 1]} * pe;
      pk[i] = pack_bf16x2(pe.x, pe.y);
      dsk[i] = pack_bf16x2(d.x, d.y);
    }
    // dS[q][key = lane] into the turn tile: registers 4 g .. 4 g + 3 are queries 8 g + 4 half .. + 3 of row `key`
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<u32x2*>(tsc + w * AB1_TURN + (tw[g] - tbase)) = u32x2{dsk[2 * g], dsk[2 * g + 1]};
    // the column fragments of Q_i (X half) and dO_i (Y half): f[st][dd]
    bf16x8 xq[2][2], xo[2][2];
    {
      const unsigned o = ib * AB_BLK;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        union { s16x4 hh[2]; bf16x8 bb; } u;
        u.hh[0] = AB1_TR(xa.a0 + o + st * 2048); u.hh[1] = AB1_TR(xa.b0 + o + st * 2048); xq[st][0] = u.bb;
        u.hh[0] = AB1_TR(xa.a32 + o + st * 2048); u.hh[1] = AB1_TR(xa.b32 + o + st * 2048); xq[st][1] = u.bb;
        u.hh[0] = AB1_TR(xa.a0 + o + AB_Y + st * 2048); u.hh[1] = AB1_TR(xa.b0 + o + AB_Y + st * 2048); xo[st][0] = u.bb;
        u.hh[0] = AB1_TR(xa.a32 + o + AB_Y + st * 2048); u.hh[1] = AB1_TR(xa.b32 + o + AB_Y + st * 2048); xo[st][1] = u.bb;
      }
    }
    // dS^T fragments (lane = query, 8 keys per k-step) back from the turn tile
    bf16x8 dst[2];
    {
      union { s16x4 hh[2]; bf16x8 bb; } u;
      u.hh[0] = AB1_TR(ta); u.hh[1] = AB1_TR(tb); dst[0] = u.bb;
      u.hh[0] = AB1_TR(ta + 1024); u.hh[1] = AB1_TR(tb + 1024); dst[1] = u.bb;
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      accK[0] = CLIMB_MFMA_H16(xq[st][0], packed4(dsk, st), accK[0], 0, 0, 0);
      accK[1] = CLIMB_MFMA_H16(xq[st][1], packed4(dsk, st), accK[1], 0, 0, 0);
      accV[0] = CLIMB_MFMA_H16(xo[st][0], packed4(pk, st), accV[0], 0, 0, 0);
      accV[1] = CLIMB_MFMA_H16(xo[st][1], packed4(pk, st), accV[1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);          // (register budget: the dQ accumulators come in only now, when xq / xo / pk / dsk are dead)
    f32x16 dq[2];
    if (t > 0) {
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = dqp[(d * 4 + g) * 64];
          dq[d][4 * g] = v.x; dq[d][4 * g + 1] = v.y; dq[d][4 * g + 2] = v.z; dq[d][4 * g + 3] = v.w;
        }
    } else {
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
    }
    if (t + 1 < NB) AB1_SDP(ibn)                // next step's S / dP: covers the round trip of the dQ accumulators
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      dq[0] = CLIMB_MFMA_H16(kT[st][0], dst[st], dq[0], 0, 0, 0);
      dq[1] = CLIMB_MFMA_H16(kT[st][1], dst[st], dq[1], 0, 0, 0);
    }
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) dqp[(d * 4 + g) * 64] = f32x4{dq[d][4 * g], dq[d][4 * g + 1], dq[d][4 * g + 2], dq[d][4 * g + 3]};
    __syncthreads();
  }
#undef AB1_SDP
#undef AB1_TR
  // every key block has added its share to every dQ block: this wave stores query block w, and its own dK / dV
  f32x16 dq[2];
  {
    const f32x4* dqp = reinterpret_cast<const f32x4*>(dqs + w * AB_BLK) + lane;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = dqp[(d * 4 + g) * 64];
        dq[d][4 * g] = v.x; dq[d][4 * g + 1] = v.y; dq[d][4 * g + 2] = v.z; dq[d][4 * g + 3] = v.w;
      }
  }
  bf16_t* orow = dqkv + ((long)b * S_pad + my) * ld + h * AB_D;
  store_acc(orow, dq, half, -scale);
  store_acc(orow + H, accK, half, -scale);
  store_acc(orow + 2 * H, accV, half, 1.0f);
}

// ------------------------------------------------------------------------------------------------------ single pass, PERSISTENT (r04)
// The kernel above at S_pad = 192 needs 110 KB of LDS: one workgroup per CU, so a CU loads (144 KB, 5 us when 256 CUs ask at once), then
// computes (9 us), then stores, and the chip alternates between an HBM burst and an idle fabric -- 60 us per layer against the two-phase
// kernel's 54.  Here a workgroup WALKS (batch, head) items (grid = CUs) and the Q / dO images are double-buffered: the eight LDS-DMA
// instructions that bring in the NEXT item's block go out right after the barrier that publishes the current one and land while its six steps
// run.  What stays exposed per item is the round trip of the wave's own K / V / O rows (12 KB per wave, registers).
// With DMA in flight inside the step loop every LDS access in it follows the rules of the streamed kernels at the top of this file: no
// __restrict__ on the kernel's pointers, reads are typed vector loads or asm, every WRITE is asm (hipcc puts a vmcnt(0) in front of a ds_write
// it can see while LDS DMA is pending -- that would be the whole prefetch), the barrier is a raw s_barrier behind an explicit lgkmcnt(0)
// (__syncthreads() carries a workgroup-scope release fence = vmcnt(0)).
// LDS: 2 x NB x 8 KB images | NB x 8 KB dQ accumulators | NB x 2 KB turn tiles | lse, delta = 157.5 KB at S_pad = 192.
__device__ __forceinline__ void ab1_ds_write_b128(unsigned addr, const f32x4& v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void ab1_lds_fence_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
__global__ __launch_bounds__(384) void attn_bwd_bf16_1pp_kernel(const bf16_t* qkv, const float* key_bias, const bf16_t* dctx, const bf16_t* ctx, const float* lse,
                                                                bf16_t* dqkv, int S_pad, int heads, float scale, int items) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int NB = S_pad / 32;
  unsigned char* dqs = smem + 2 * NB * AB_BLK;
  unsigned char* tsc = dqs + NB * AB_BLK;
  float* lse_s = reinterpret_cast<float*>(tsc + NB * AB1_TURN);
  float* delta_s = lse_s + AB_VEC(S_pad);
  const int H = heads * AB_D, ld = 3 * H;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l31 = lane & 31;
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
  // lane constants: turn tile, dQ accumulators (raw float4 (dd * 4 + g) of lane l at ((dd * 4 + g) * 64 + l) * 16), K scratch
  const unsigned tbase = lds0 + (unsigned)(tsc - smem) + w * AB1_TURN;
  unsigned tw[4], ta, tb;
#pragma unroll
  for (int g = 0; g < 4; ++g) tw[g] = tbase + l31 * 64 + ((g ^ ((l31 >> 2) & 3)) << 4) + 8 * half;
  {
    const int g16 = lane >> 4, i = lane & 15;
    const int rowa = 4 * (g16 >> 1) + (i >> 2), rowb = rowa + 8, col = (g16 & 1) * 16 + 4 * (i & 3);
    ta = tbase + rowa * 64 + (((col >> 3) ^ ((rowa >> 2) & 3)) << 4) + (col & 7) * 2;
    tb = tbase + rowb * 64 + (((col >> 3) ^ ((rowb >> 2) & 3)) << 4) + (col & 7) * 2;
  }
  const unsigned dq0 = lds0 + (unsigned)(dqs - smem) + lane * 16;
  const float c1 = scale * AB_LOG2E, is = -1.0f / scale;
  // this wave's block of an item's Q / dO into image buffer `buf`
  auto prefetch = [&](int item, int buf) {
    const int b = item / heads, h = item % heads;
    const bf16_t* Qg = qkv + (long)b * S_pad * ld + h * AB_D;
    const bf16_t* dOg = dctx + (long)b * S_pad * H + h * AB_D;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)Qg, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dOg, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int slot = (4 * w + u) * 64 + lane, row = slot >> 3, c = (slot & 7) ^ aswz(row);
      unsigned char* dst = smem + buf * NB * AB_BLK + w * AB_BLK + u * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void_t*)dst, 16, row * (ld * 2) + c * 16, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_void_t*)(dst + AB_Y), 16, row * (H * 2) + c * 16, 0, 0, 0);
    }
  };
  int item = blockIdx.x;
  if (item < items) prefetch(item, 0);
  for (int n = 0; item < items; ++n, item += gridDim.x) {
    const int buf = n & 1;
    unsigned char* img = smem + buf * NB * AB_BLK;
    const int b = item / heads, h = item % heads;
    const bf16_t* Kg = qkv + (long)b * S_pad * ld + h * AB_D + H;
    const bf16_t* Vg = Kg + H;
    const bf16_t* Og = ctx + (long)b * S_pad * H + h * AB_D;
    const long rowv = ((long)b * heads + h) * S_pad;
    const int my = w * 32 + l31;
    bf16x8 f1[4], f2[4];
    float my_c;
    {
      bf16x8 fo[4];
      load_rows(f1, Kg, ld, w * 32, lane);
      load_rows(f2, Vg, ld, w * 32, lane);
      load_rows(fo, Og, H, w * 32, lane);
      my_c = key_bias[(long)b * S_pad + my];
      float my_lse = lse[rowv + my];
      __builtin_amdgcn_sched_barrier(0);
      ab_wait_vm<0>();                 // this wave's rows, its block of the item's images (asked for an item ago), and the previous item's stores
      use_rows(f1); use_rows(f2); use_rows(fo);
      asm volatile("" : "+v"(my_c), "+v"(my_lse));
      float dsum = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 fd = row_frag(img + w * AB_BLK + AB_Y, ks, lane);          // dO rows of this wave's own block: same elements as fo's
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += (float)fo[ks][e] * (float)fd[e];
      }
      const float my_delta = pair_sum(dsum);
      if (half == 0) { delta_s[my] = my_delta; lse_s[my] = my_lse; }      // (no LDS DMA is in flight here: plain stores)
    }
    bf16x8 kT[2][2];
    {
      unsigned char* ksc = dqs + w * AB_BLK;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        union { bf16x8 b; u32x4 u; } c;
        c.b = f1[ks];
        *reinterpret_cast<u32x4*>(ksc + l31 * 128 + (((2 * ks + half) ^ aswz(l31)) << 4)) = c.u;
      }
      const ColAddr ka = col_addr(ksc, lane);
      col_frags4<0>(kT, ka, 0);
    }
    negate_rows(f2);
    const ColAddr xa = col_addr(img, lane);
    const float k2 = fmaxf(my_c, AB_NEG) * AB_LOG2E;
    ab1_lds_fence_barrier();           // every wave's block of this item, lse and delta are in LDS; nobody reads the other image buffer any more
    if (item + (int)gridDim.x < items) prefetch(item + gridDim.x, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 accK[2], accV[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) accK[d][r] = accV[d][r] = 0.f;
    for (int t = 0; t < NB; ++t) {
      int ib = w + t;
      ib = ib >= NB ? ib - NB : ib;
      f32x16 s, dp;
      {
        const f32x4* ap = reinterpret_cast<const f32x4*>(lse_s + ib * 32 + 4 * half);
        const f32x4* dl = reinterpret_cast<const f32x4*>(delta_s + ib * 32 + 4 * half);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = ap[2 * g], dd = dl[2 * g];
          s[4 * g] = v.x * is; s[4 * g + 1] = v.y * is; s[4 * g + 2] = v.z * is; s[4 * g + 3] = v.w * is;
          dp[4 * g] = dd.x; dp[4 * g + 1] = dd.y; dp[4 * g + 2] = dd.z; dp[4 * g + 3] = dd.w;
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = CLIMB_MFMA_H16(row_frag(img + ib * AB_BLK, ks, lane), f1[ks], s, 0, 0, 0);
        dp = CLIMB_MFMA_H16(row_frag(img + ib * AB_BLK + AB_Y, ks, lane), f2[ks], dp, 0, 0, 0);
      }
      unsigned int pk[8], dsk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x2 e = f32x2{s[2 * i], s[2 * i + 1]} * c1 + k2;
        const f32x2 pe = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
        const f32x2 d = f32x2{dp[2 * i], dp[2 * i + 1]} * pe;
        pk[i] = pack_bf16x2(pe.x, pe.y);
        dsk[i] = pack_bf16x2(d.x, d.y);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u32x2 v = {dsk[2 * g], dsk[2 * g + 1]};
        asm volatile("ds_write_b64 %0, %1" ::"v"(tw[g]), "v"(v) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 f[2];
      col_frags2<0, 0>(f, xa, ib);
      accK[0] = CLIMB_MFMA_H16(f[0], packed4(dsk, 0), accK[0], 0, 0, 0);
      accK[1] = CLIMB_MFMA_H16(f[1], packed4(dsk, 0), accK[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      col_frags2<1, 0>(f, xa, ib);
      accV[0] = CLIMB_MFMA_H16(f[0], packed4(pk, 0), accV[0], 0, 0, 0);
      accV[1] = CLIMB_MFMA_H16(f[1], packed4(pk, 0), accV[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      col_frags2<0, 1>(f, xa, ib);
      accK[0] = CLIMB_MFMA_H16(f[0], packed4(dsk, 1), accK[0], 0, 0, 0);
      accK[1] = CLIMB_MFMA_H16(f[1], packed4(dsk, 1), accK[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      col_frags2<1, 1>(f, xa, ib);
      accV[0] = CLIMB_MFMA_H16(f[0], packed4(pk, 1), accV[0], 0, 0, 0);
      accV[1] = CLIMB_MFMA_H16(f[1], packed4(pk, 1), accV[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the dQ accumulators of block ib (nobody else touches them in this step) and the turned dS
      f32x16 dq[2];
      const unsigned dqa = dq0 + ib * AB_BLK;
      if (t > 0) {
        const f32x4* dqp = reinterpret_cast<const f32x4*>(dqs + ib * AB_BLK) + lane;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = dqp[(d * 4 + g) * 64];
            dq[d][4 * g] = v.x; dq[d][4 * g + 1] = v.y; dq[d][4 * g + 2] = v.z; dq[d][4 * g + 3] = v.w;
          }
      } else {
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
      }
      bf16x8 dst[2];
      {
        s16x4 l0, h0, l1, h1;
        asm volatile(
            "ds_read_b64_tr_b16 %0, %4\n\t"
            "ds_read_b64_tr_b16 %1, %5\n\t"
            "ds_read_b64_tr_b16 %2, %4 offset:1024\n\t"
            "ds_read_b64_tr_b16 %3, %5 offset:1024\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1)
            : "v"(ta), "v"(tb)
            : "memory");
        union { s16x4 hh[2]; bf16x8 bb; } u;
        u.hh[0] = l0; u.hh[1] = h0; dst[0] = u.bb;
        u.hh[0] = l1; u.hh[1] = h1; dst[1] = u.bb;
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        dq[0] = CLIMB_MFMA_H16(kT[st][0], dst[st], dq[0], 0, 0, 0);
        dq[1] = CLIMB_MFMA_H16(kT[st][1], dst[st], dq[1], 0, 0, 0);
      }
      // hipcc's hazard recognizer does not look inside inline asm: an asm ds_write that reads the registers a just-issued MFMA is still writing
      // gets no wait states (the compiler's own ds_write gets `s_nop 11` here) and stores stale accumulators -- found as a wrong dQ only
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        asm volatile("s_nop 15" : "+v"(dq[d]));
#pragma unroll
        for (int g = 0; g < 4; ++g)
          ab1_ds_write_b128(dqa + (d * 4 + g) * 1024, f32x4{dq[d][4 * g], dq[d][4 * g + 1], dq[d][4 * g + 2], dq[d][4 * g + 3]});
      }
      ab1_lds_fence_barrier();
    }
    f32x16 dq[2];
    {
      const f32x4* dqp = reinterpret_cast<const f32x4*>(dqs + w * AB_BLK) + lane;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = dqp[(d * 4 + g) * 64];
          dq[d][4 * g] = v.x; dq[d][4 * g + 1] = v.y; dq[d][4 * g + 2] = v.z; dq[d][4 * g + 3] = v.w;
        }
    }
    bf16_t* orow = dqkv + ((long)b * S_pad + my) * ld + h * AB_D;
    store_acc(orow, dq, half, -scale);
    store_acc(orow + H, accK, half, -scale);
    store_acc(orow + 2 * H, accV, half, 1.0f);
    // the next item's prologue overwrites lse_s / delta_s and this wave's own dQ block: its first LDS write must not pass a slower wave's last
    // read of this item -- the step loop's final barrier already orders that (everything after it reads only the wave's own dQ block)
  }
  ab_wait_vm<0>();                     // no LDS DMA may outlive the workgroup's allocation
}

// climb_set_option 13: 0 = the two launches, 1 = both phases in one launch, 2 (default) = the single pass where two workgroups fit a CU (S_pad <= 128:
// 28.2 against 33.9 us per layer at S_pad = 128, B = 64) and both phases in one launch above that, 3 / 4 = the plain / the persistent single pass
// wherever it runs (S_pad <= 192).  Measured at the benchmark's S_pad = 192 (tools/attn_bench.py): two phases 53.9 us, single pass 60.4, persistent
// single pass 55.9 -- 30 % fewer MFMAs, half the exponentials and 151 instead of 231 MB do NOT win there: at 208 - 241 registers and 110 - 158 KB of
// LDS one workgroup of six waves owns the CU (1.5 waves per SIMD, two SIMDs with one wave), and a block pair is a dependent chain (fragments ->
// S / dP -> exponentials -> turn -> four fragment round trips -> dQ read-modify-write -> barrier) that nothing else on the SIMD covers: 18 us per
// item with its loads prefetched.  The two-phase kernel keeps nine waves per CU.  Kept as tested options and as the priced answer.
static int g_attn_bwd_fused = 2;
void climb_attn_set_bwd_fused(int v) { g_attn_bwd_fused = v; }
static int g_attn_1pp_grid = 0;
void climb_attn_set_1pp_grid(int v) { g_attn_1pp_grid = v; }

extern "C" int climb_attn_bwd_bf16(const void* qkv, const float* key_bias, const void* dctx, const void* ctx, const float* lse, float* delta,
                                   void* dqkv, int B, int S_pad, int heads, int head_dim, void* stream) {
  if (head_dim != AB_D || S_pad % 32 || S_pad <= 0 || S_pad > 512) return CLIMB_EUNSUPPORTED;
  size_t lds = (size_t)S_pad * 256 + (size_t)AB_VEC(S_pad) * 12;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_bf16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd_bf16_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    lds_set = lds;
  }
  const float scale = 1.0f / sqrtf((float)head_dim);
  // two outer blocks per wave: 64 nw >= S_pad threads (one element of the LDS vectors each) and 4 NB units over nw waves is at most 8 rounds;
  // measured at S_pad = 192: 3 waves 63 us, 6 waves 78 us per layer (register pressure: 3 workgroups x 3 waves fill the 168-VGPR budget)
  const int nthreads = 64 * ((S_pad / 32 + 1) / 2);
  if (g_attn_bwd_fused == 4 && S_pad <= 192) {
    const int NB = S_pad / 32, items = B * heads;
    const size_t ldsp = (size_t)NB * (3 * AB_BLK + AB1_TURN) + (size_t)AB_VEC(S_pad) * 8;
    static size_t ldsp_set = 0;
    static int ncu = 0;
    if (ldsp > ldsp_set) {
      hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_bf16_1pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
      if (e != hipSuccess) return (int)e;
      ldsp_set = ldsp;
    }
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return CLIMB_EINVAL;
      ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int per_cu = (int)(163840 / ldsp) > 0 ? (int)(163840 / ldsp) : 1;          // resident workgroups per CU by LDS
    int grid = g_attn_1pp_grid > 0 ? g_attn_1pp_grid : ncu * per_cu;          // (climb_set_option 20: tests walk several items per workgroup on small problems)
    if (grid > items) grid = items;
    hipLaunchKernelGGL(attn_bwd_bf16_1pp_kernel, dim3(grid), dim3(64 * NB), ldsp, (hipStream_t)stream, (const bf16_t*)qkv, key_bias, (const bf16_t*)dctx,
                       (const bf16_t*)ctx, lse, (bf16_t*)dqkv, S_pad, heads, scale, items);
    LAUNCH_CHECK();
    return CLIMB_OK;
  }
  if ((g_attn_bwd_fused == 2 && S_pad <= 128) || (g_attn_bwd_fused == 3 && S_pad <= 192)) {
    const int NB = S_pad / 32;
    const size_t lds1 = (size_t)NB * (2 * AB_BLK + AB1_TURN) + (size_t)AB_VEC(S_pad) * 8;
    static size_t lds1_set = 0;
    if (lds1 > lds1_set) {
      hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_bf16_1p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
      if (e != hipSuccess) return (int)e;
      lds1_set = lds1;
    }
    hipLaunchKernelGGL(attn_bwd_bf16_1p_kernel, dim3(B * heads), dim3(64 * NB), lds1, (hipStream_t)stream, (const bf16_t*)qkv, key_bias,
                       (const bf16_t*)dctx, (const bf16_t*)ctx, lse, (bf16_t*)dqkv, S_pad, heads, scale);
    LAUNCH_CHECK();
    return CLIMB_OK;
  }
  if (g_attn_bwd_fused) {
    hipLaunchKernelGGL(attn_bwd_bf16_fused_kernel, dim3(B * heads), dim3(nthreads), lds, (hipStream_t)stream, (const bf16_t*)qkv, key_bias,
                       (const bf16_t*)dctx, (const bf16_t*)ctx, lse, delta, (bf16_t*)dqkv, S_pad, heads, scale);
    LAUNCH_CHECK();
    return CLIMB_OK;
  }
  hipLaunchKernelGGL((attn_bwd_bf16_kernel<0>), dim3(B * heads), dim3(nthreads), lds, (hipStream_t)stream, (const bf16_t*)qkv, key_bias,
                     (const bf16_t*)dctx, (const bf16_t*)ctx, lse, delta, (bf16_t*)dqkv, S_pad, heads, scale);
  LAUNCH_CHECK();
  hipLaunchKernelGGL((attn_bwd_bf16_kernel<1>), dim3(B * heads), dim3(nthreads), lds, (hipStream_t)stream, (const bf16_t*)qkv, key_bias,
                     (const bf16_t*)dctx, (const bf16_t*)ctx, lse, delta, (bf16_t*)dqkv, S_pad, heads, scale);
  LAUNCH_CHECK();
  return CLIMB_OK;
}
