// Device side of the ViLT image pre-processing: what the reference does on its training thread with
// ViltProcessor / Pillow (transformers image_processing_pil_vilt.py:127-242; Pillow libImaging/Resample.c) done on the GPU from the
// raw uint8 pixels: 8-bit two-pass bicubic resample with the host-computed 22-bit fixed-point coefficients (bit-exact with
// Pillow: integer arithmetic end to end), then rescale + normalise through a 256-entry table (the whole float path of the
// reference is a function of one byte) fused with the zero padding and the pixel mask.
// HBM-bound byte work, tiny next to the step (a 64-image batch is ~60 MB in, ~190 MB out); one launch per stage per batch.
#include "common.h"

#define IMG_PRECISION_BITS 22
#define IMG_DESC 16          // longs per image: src_off tmp_off dst_off sh sw dh dw kh_off bh_off ksh kv_off bv_off ksv - - -

__device__ __forceinline__ unsigned char clip8(int ss) {
  const int v = ss >> IMG_PRECISION_BITS;          // arithmetic shift, as Resample.c's clip8 lookup
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// PASS 0: rows of src [sh][sw][3] -> tmp [sh][dw][3]   (reduction along x)
// PASS 1: columns of tmp [sh][dw][3] -> dst [dh][dw][3] (reduction along y)
template <int PASS>
__global__ __launch_bounds__(256) void image_resample_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                             const int* __restrict__ coef, const long* __restrict__ table) {
  const long* t = table + (long)blockIdx.y * IMG_DESC;
  const int sh = (int)t[3], sw = (int)t[4], dh = (int)t[5], dw = (int)t[6];
  const unsigned char* in = src + (PASS == 0 ? t[0] : t[1]);
  unsigned char* out = dst + (PASS == 0 ? t[1] : t[2]);
  const int* kk = coef + (PASS == 0 ? t[7] : t[10]);
  const int* bounds = coef + (PASS == 0 ? t[8] : t[11]);
  const int ks = (int)(PASS == 0 ? t[9] : t[12]);
  const int rows = PASS == 0 ? sh : dh, rowlen = dw * 3;
  const long total = (long)rows * rowlen;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += gridDim.x * 256L) {
    const int y = (int)(e / rowlen), r = (int)(e - (long)y * rowlen), x = r / 3, c = r - x * 3;
    const int o = PASS == 0 ? x : y;                       // output index along the resampled axis
    const int first = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = kk + (long)o * ks;
    int ss = 1 << (IMG_PRECISION_BITS - 1);
    if (PASS == 0) {
      const unsigned char* p = in + ((long)y * sw + first) * 3 + c;
      for (int i = 0; i < n; ++i) ss += (int)p[i * 3] * k[i];
    } else {
      const unsigned char* p = in + ((long)first * dw + x) * 3 + c;
      for (int i = 0; i < n; ++i) ss += (int)p[(long)i * rowlen] * k[i];
    }
    out[e] = clip8(ss);
  }
}

extern "C" int climb_image_resample(const void* src, void* tmp, void* dst, const int* coef, const long* table, int n_images, long max_elems,
                                    void* stream) {
  if (n_images <= 0 || max_elems <= 0) return CLIMB_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  long nb = (max_elems + 255) / 256;
  if (nb > 4096) nb = 4096;
  // src / tmp / dst may be one arena: the table holds byte offsets into the pointer of the stage being read or written
  hipLaunchKernelGGL((image_resample_kernel<0>), dim3((unsigned)nb, n_images), dim3(256), 0, st, (const unsigned char*)src, (unsigned char*)tmp, coef, table);
  LAUNCH_CHECK();
  hipLaunchKernelGGL((image_resample_kernel<1>), dim3((unsigned)nb, n_images), dim3(256), 0, st, (const unsigned char*)tmp, (unsigned char*)dst, coef, table);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// dst [dh][dw][3] uint8 -> pixel_values[b] [3][Hc][Wc] float (lut[byte], zeros right of / below the image), pixel_mask[b] [Hc][Wc]
__global__ __launch_bounds__(256) void image_normalize_pad_kernel(const unsigned char* __restrict__ img, const long* __restrict__ table,
                                                                  const float* __restrict__ lut, float* __restrict__ pixel_values,
                                                                  long* __restrict__ pixel_mask, int Hc, int Wc) {
  __shared__ float slut[256];
  slut[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const long* t = table + (long)blockIdx.y * IMG_DESC;
  const int dh = (int)t[5], dw = (int)t[6];
  const unsigned char* in = img + t[2];
  const long plane = (long)Hc * Wc;
  float* pv = pixel_values + (long)blockIdx.y * 3 * plane;
  long* pm = pixel_mask + (long)blockIdx.y * plane;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < plane; e += gridDim.x * 256L) {
    const int y = (int)(e / Wc), x = (int)(e - (long)y * Wc);
    const bool inside = y < dh && x < dw;
    float r = 0.f, g = 0.f, b = 0.f;
    if (inside) {
      const unsigned char* p = in + ((long)y * dw + x) * 3;
      r = slut[p[0]]; g = slut[p[1]]; b = slut[p[2]];
    }
    pv[e] = r; pv[plane + e] = g; pv[2 * plane + e] = b;
    pm[e] = inside ? 1 : 0;
  }
}

extern "C" int climb_image_normalize_pad(const void* img, const long* table, const float* lut, float* pixel_values, long* pixel_mask, int n_images,
                                         int Hc, int Wc, void* stream) {
  if (n_images <= 0 || Hc <= 0 || Wc <= 0) return CLIMB_EINVAL;
  long nb = ((long)Hc * Wc + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(image_normalize_pad_kernel, dim3((unsigned)nb, n_images), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)img, table, lut,
                     pixel_values, pixel_mask, Hc, Wc);
  LAUNCH_CHECK();
  return CLIMB_OK;
}
