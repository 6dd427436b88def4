// Small data-movement kernels on the VACNIC path (gfx950): token-axis concat copies, CLIP patch
// im2col, id preprocessing, bias gradients, arg-max.  All HBM/latency-bound, 16-byte accesses
// where the layout allows.
#include "common.h"

namespace {

inline unsigned grid_for(long work) {
  long b = (work + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// dst[b][r][c] (+)= src[b][r][c], cols multiple of 8
__global__ __launch_bounds__(256) void copy3d_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, long B,
                                                     long rows, long cols8, long lds, long ldd, long bss, long bsd,
                                                     int accumulate) {
  const long total = B * rows * cols8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long c = i % cols8; long r = i / cols8;
    const long b = r / rows; r = r % rows;
    const u32x4 s = *(const u32x4*)(src + b * bss + r * lds + c * 8);
    bf16_t* d = dst + b * bsd + r * ldd + c * 8;
    if (!accumulate) {
      *(u32x4*)d = s;
    } else {
      u32x4 o = *(u32x4*)d, w;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        w[k] = pack2bf(__uint_as_float(o[k] << 16) + __uint_as_float(s[k] << 16),
                       __uint_as_float(o[k] & 0xffff0000u) + __uint_as_float(s[k] & 0xffff0000u));
      *(u32x4*)d = w;
    }
  }
}

__global__ __launch_bounds__(256) void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                  bf16_t* __restrict__ o, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const u32x4 x = ((const u32x4*)a)[i], y = ((const u32x4*)b)[i];
    u32x4 w;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      w[k] = pack2bf(__uint_as_float(x[k] << 16) + __uint_as_float(y[k] << 16),
                     __uint_as_float(x[k] & 0xffff0000u) + __uint_as_float(y[k] & 0xffff0000u));
    ((u32x4*)o)[i] = w;
  }
}

// dst[r][0..Cp) = src[r][0..C) then zeros: gives an odd-width activation 16-byte rows for the GEMM
__global__ __launch_bounds__(256) void pad_cols_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, long R, int C,
                                                       int Cp, long lds) {
  const long total = R * Cp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp); const long r = i / Cp;
    dst[i] = c < C ? src[r * lds + c] : (bf16_t)0;
  }
}

// patches[(b*g + gy)*g + gx][k], k = c*p*p + py*p + px  (conv1 weight [w][3][p][p] flattened)
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, long B, int HW,
                                                     int p, int Kp) {
  const int g = HW / p, K = 3 * p * p;
  const long total = B * g * g * Kp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kp); long r = i / Kp;
    float v = 0.f;
    if (k < K) {
      const int gx = (int)(r % g); r /= g;
      const int gy = (int)(r % g); const long b = r / g;
      const int c = k / (p * p), rem = k % (p * p), py = rem / p, px = rem % p;
      v = img[((b * 3 + c) * HW + gy * p + py) * (long)HW + gx * p + px];
    }
    out[i] = f2bf(v);
  }
}

// x[b][0] = cls + pos[0]; x[b][1+i] = patch_emb[b*G2+i] + pos[1+i]
__global__ __launch_bounds__(256) void vit_assemble_kernel(const bf16_t* __restrict__ pe, const bf16_t* __restrict__ cls,
                                                           const bf16_t* __restrict__ pos, bf16_t* __restrict__ out, long B,
                                                           int G2, int W) {
  const long total = B * (G2 + 1) * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W); long r = i / W;
    const int t = (int)(r % (G2 + 1)); const long b = r / (G2 + 1);
    const float base = t == 0 ? bf2f(cls[w]) : bf2f(pe[(b * G2 + (t - 1)) * W + w]);
    out[i] = f2bf(base + bf2f(pos[(long)t * W + w]));
  }
}

__global__ void prep_ids_kernel(const int64_t* __restrict__ ids, uint8_t* __restrict__ mask, int64_t* __restrict__ shifted,
                                long B, long T, int64_t pad, int64_t start) {
  const long total = B * T;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int64_t v = ids[i];
    if (mask) mask[i] = v != pad;
    if (shifted) {
      int64_t s = (i % T) == 0 ? start : ids[i - 1];
      if (s == -100) s = pad;
      shifted[i] = s;
    }
  }
}
// faces: mask = (face_emb[b][f][D-1] != 1)   (TRAIN:269)
__global__ void face_mask_kernel(const float* __restrict__ faces, uint8_t* __restrict__ mask, long BF, long D) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < BF) mask[i] = faces[i * D + D - 1] != 1.0f;
}

template <bool F32>
__global__ __launch_bounds__(256) void argmax_kernel(const void* __restrict__ logits, int64_t* __restrict__ out, int V, long ldl) {
  __shared__ float bv[256];
  __shared__ int bi[256];
  const long r = blockIdx.x;
  float best = -INFINITY; int idx = 0x7fffffff;
  for (int j = threadIdx.x; j < V; j += 256) {
    const float v = F32 ? ((const float*)logits)[r * ldl + j] : bf2f(((const bf16_t*)logits)[r * ldl + j]);
    if (idx == 0x7fffffff || v > best) { best = v; idx = j; }   // j ascends: lowest index wins ties
  }
  bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const float ov = bv[threadIdx.x + s]; const int oi = bi[threadIdx.x + s];
      if (oi != 0x7fffffff && (bi[threadIdx.x] == 0x7fffffff || ov > bv[threadIdx.x] || (ov == bv[threadIdx.x] && oi < bi[threadIdx.x]))) {
        bv[threadIdx.x] = ov; bi[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[r] = bi[0];
}

// dbias[n] += sum_m dy[m][n]; block = 64 column-chunks (512 cols) x 4 row groups over a 32..64-row slab, so a
// [16384 x 1024] gradient spreads over ~1000 workgroups with 8 independent 16-byte loads in flight per thread
__global__ __launch_bounds__(256) void bias_grad_kernel(const bf16_t* __restrict__ dy, float* __restrict__ dbias, long M, int N,
                                                        long ldy, int rows_per_block) {
  __shared__ float red[4][64 * 8 + 1];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;           // 8-col chunk
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(M, r0 + rows_per_block);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (c * 8 + 8 <= N) {
    long r = r0 + rg;
    for (; r + 28 < r1; r += 32) {                // 8 rows per trip, all loads issued before the adds
      u32x4 d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d[u] = *(const u32x4*)(dy + (r + 4 * u) * ldy + c * 8);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[2 * k] += __uint_as_float(d[u][k] << 16); acc[2 * k + 1] += __uint_as_float(d[u][k] & 0xffff0000u); }
    }
    for (; r < r1; r += 4) {
      const u32x4 d = *(const u32x4*)(dy + r * ldy + c * 8);
#pragma unroll
      for (int k = 0; k < 4; ++k) { acc[2 * k] += __uint_as_float(d[k] << 16); acc[2 * k + 1] += __uint_as_float(d[k] & 0xffff0000u); }
    }
  } else if (c * 8 < N) {
    for (long r = r0 + rg; r < r1; r += 4)
      for (int j = 0; j < 8; ++j) if (c * 8 + j < N) acc[j] += bf2f(dy[r * ldy + c * 8 + j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rg][lane * 8 + j] = acc[j];
  __syncthreads();
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int col = blockIdx.x * 512 + e;
    if (col < N) atomicAdd(dbias + col, red[0][e] + red[1][e] + red[2][e] + red[3][e]);
  }
}

}  // namespace

extern "C" int vacnic_copy3d_bf16(const void* src, void* dst, int64_t B, int64_t rows, int64_t cols, int64_t lds,
                                  int64_t ldd, int64_t bss, int64_t bsd, int32_t accumulate, void* stream) {
  VPLAN_REC(vacnic_copy3d_bf16, src, dst, B, rows, cols, lds, ldd, bss, bsd, accumulate, stream);
  VCHECK(src && dst, VACNIC_BAD_SHAPE, "copy3d: null operand");
  VCHECK((cols & 7) == 0 && (lds & 7) == 0 && (ldd & 7) == 0 && (bss & 7) == 0 && (bsd & 7) == 0 && aligned16(src) && aligned16(dst),
         VACNIC_MISALIGNED, "copy3d: cols/strides must be multiples of 8 and pointers 16-byte aligned");
  if (B * rows * cols == 0) return VACNIC_OK;
  hipLaunchKernelGGL(copy3d_kernel, dim3(grid_for(B * rows * (cols >> 3))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, (bf16_t*)dst, (long)B, (long)rows, (long)(cols >> 3), (long)lds, (long)ldd,
                     (long)bss, (long)bsd, accumulate);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
extern "C" int vacnic_copy2d_bf16(const void* src, void* dst, int64_t rows, int64_t cols, int64_t lds, int64_t ldd,
                                  int32_t accumulate, void* stream) {
  VPLAN_REC(vacnic_copy2d_bf16, src, dst, rows, cols, lds, ldd, accumulate, stream);
  return vacnic_copy3d_bf16(src, dst, 1, rows, cols, lds, ldd, 0, 0, accumulate, stream);
}
extern "C" int vacnic_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
  VPLAN_REC(vacnic_add_bf16, a, b, out, n, stream);
  VCHECK(a && b && out, VACNIC_BAD_SHAPE, "add: null operand");
  VCHECK((n & 7) == 0 && aligned16(a) && aligned16(b) && aligned16(out), VACNIC_MISALIGNED, "add: n must be a multiple of 8, pointers 16-byte aligned");
  if (n == 0) return VACNIC_OK;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                     (const bf16_t*)b, (bf16_t*)out, (long)(n >> 3));
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
extern "C" int vacnic_pad_cols_bf16(const void* src, void* dst, int64_t R, int64_t C, int64_t Cp, int64_t lds, void* stream) {
  VPLAN_REC(vacnic_pad_cols_bf16, src, dst, R, C, Cp, lds, stream);
  VCHECK(src && dst && Cp >= C && lds >= C, VACNIC_BAD_SHAPE, "pad_cols: bad operand");
  if (R * Cp == 0) return VACNIC_OK;
  hipLaunchKernelGGL(pad_cols_kernel, dim3(grid_for(R * Cp)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                     (bf16_t*)dst, (long)R, (int)C, (int)Cp, (long)lds);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
extern "C" int vacnic_im2col_patches(const float* img, void* patches, int64_t B, int64_t HW, int64_t patch, int64_t Kp,
                                     void* stream) {
  VPLAN_REC(vacnic_im2col_patches, img, patches, B, HW, patch, Kp, stream);
  VCHECK(img && patches, VACNIC_BAD_SHAPE, "im2col: null operand");
  VCHECK(patch > 0 && HW % patch == 0 && Kp >= 3 * patch * patch, VACNIC_BAD_SHAPE, "im2col: bad geometry");
  const long g = HW / patch;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(B * g * g * Kp)), dim3(256), 0, (hipStream_t)stream, img,
                     (bf16_t*)patches, (long)B, (int)HW, (int)patch, (int)Kp);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
extern "C" int vacnic_vit_assemble(const void* patch_emb, const void* cls, const void* pos, void* out, int64_t B,
                                   int64_t G2, int64_t W, void* stream) {
  VPLAN_REC(vacnic_vit_assemble, patch_emb, cls, pos, out, B, G2, W, stream);
  VCHECK(patch_emb && cls && pos && out, VACNIC_BAD_SHAPE, "vit_assemble: null operand");
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(grid_for(B * (G2 + 1) * W)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)patch_emb, (const bf16_t*)cls, (const bf16_t*)pos, (bf16_t*)out, (long)B, (int)G2, (int)W);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
extern "C" int vacnic_prep_ids(const int64_t* ids, uint8_t* mask, int64_t* shifted, int64_t B, int64_t T, int64_t pad_id,
                               int64_t start_id, void* stream) {
  VPLAN_REC(vacnic_prep_ids, ids, mask, shifted, B, T, pad_id, start_id, stream);
  VCHECK(ids, VACNIC_BAD_SHAPE, "prep_ids: null ids");
  if (B * T == 0) return VACNIC_OK;
  hipLaunchKernelGGL(prep_ids_kernel, dim3(grid_for(B * T)), dim3(256), 0, (hipStream_t)stream, ids, mask, shifted,
                     (long)B, (long)T, pad_id, start_id);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
namespace {
__global__ void cat2_u8_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint8_t* __restrict__ out, int64_t B, int64_t na,
                               int64_t nb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = na + nb;
  if (i >= B * n) return;
  const int64_t r = i / n, c = i - r * n;
  out[i] = c < na ? a[r * na + c] : b[r * nb + (c - na)];
}
}  // namespace

// out[B, na + nb] = cat(a[B, na], b[B, nb], dim 1) on bytes: the [faces ; names] key mask of MFULL:1262 (torch.cat of two masks)
extern "C" int vacnic_cat2_u8(const uint8_t* a, const uint8_t* b, uint8_t* out, int64_t B, int64_t na, int64_t nb, void* stream) {
  VPLAN_REC(vacnic_cat2_u8, a, b, out, B, na, nb, stream);
  VCHECK(a && b && out && B > 0 && na >= 0 && nb >= 0 && na + nb > 0, VACNIC_BAD_SHAPE, "cat2_u8: bad operand");
  const int64_t tot = B * (na + nb);
  hipLaunchKernelGGL(cat2_u8_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, B, na, nb);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_face_mask(const float* faces, uint8_t* mask, int64_t BF, int64_t D, void* stream) {
  VPLAN_REC(vacnic_face_mask, faces, mask, BF, D, stream);
  VCHECK(faces && mask, VACNIC_BAD_SHAPE, "face_mask: null operand");
  if (BF == 0) return VACNIC_OK;
  hipLaunchKernelGGL(face_mask_kernel, dim3((unsigned)((BF + 255) / 256)), dim3(256), 0, (hipStream_t)stream, faces, mask,
                     (long)BF, (long)D);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
extern "C" int vacnic_argmax_rows(const void* logits, int64_t* out, int64_t R, int64_t V, int64_t ldl, int32_t logits_f32,
                                  void* stream) {
  VPLAN_REC(vacnic_argmax_rows, logits, out, R, V, ldl, logits_f32, stream);
  VCHECK(logits && out && V > 0, VACNIC_BAD_SHAPE, "argmax: bad operand");
  if (R == 0) return VACNIC_OK;
  if (logits_f32) hipLaunchKernelGGL(argmax_kernel<true>, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, out, (int)V, (long)ldl);
  else hipLaunchKernelGGL(argmax_kernel<false>, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, out, (int)V, (long)ldl);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
extern "C" int vacnic_bias_grad(const void* dy, float* dbias, int64_t M, int64_t N, int64_t ldy, void* stream) {
  VPLAN_REC(vacnic_bias_grad, dy, dbias, M, N, ldy, stream);
  VCHECK(dy && dbias, VACNIC_BAD_SHAPE, "bias_grad: null operand");
  VCHECK((ldy & 7) == 0 && aligned16(dy), VACNIC_MISALIGNED, "bias_grad: dy rows must be 16-byte aligned");
  if (M == 0 || N == 0) return VACNIC_OK;
  const int cb = (int)((N + 511) / 512);
  // ~64-row slabs, but keep the grid around 2k workgroups for very tall inputs
  long rb = (M + 63) / 64;
  const long cap = 2048 / cb > 1 ? 2048 / cb : 1;
  if (rb > cap) rb = cap;
  const int rows_per_block = (int)((M + rb - 1) / rb);
  hipLaunchKernelGGL(bias_grad_kernel, dim3(cb, (unsigned)rb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, dbias, (long)M,
                     (int)N, (long)ldy, rows_per_block);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}


// ---- uint8 image -> normalised fp32 (ToTensor + Normalize + optional horizontal flip), 4 pixels per thread
namespace {
__global__ __launch_bounds__(256) void image_u8_normalize_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ flip,
                                                                 float* __restrict__ dst, long B, int H, int W, float m0, float m1,
                                                                 float m2, float s0, float s1, float s2) {
  const long plane = (long)H * W, total = B * 3 * plane;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long)gridDim.x * 256 * 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long j = i + e;
      if (j >= total) break;
      const long b = j / (3 * plane), r = j - b * 3 * plane;
      const int c = (int)(r / plane);
      const long yx = r - (long)c * plane;
      const int y = (int)(yx / W), x = (int)(yx - (long)y * W);
      const int sx = (flip && flip[b]) ? W - 1 - x : x;
      const float t = (float)src[(b * 3 + c) * plane + (long)y * W + sx] / 255.0f;
      const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
      dst[j] = (t - mean) / sd;
    }
  }
}
}  // namespace

extern "C" int vacnic_image_u8_normalize(const uint8_t* src, const uint8_t* flip, float* dst, int64_t B, int64_t H, int64_t W,
                                         float mean0, float mean1, float mean2, float std0, float std1, float std2, void* stream) {
  VPLAN_REC(vacnic_image_u8_normalize, src, flip, dst, B, H, W, mean0, mean1, mean2, std0, std1, std2, stream);
  VCHECK(src && dst, VACNIC_BAD_SHAPE, "image_u8_normalize: null operand");
  VCHECK(B >= 0 && H > 0 && W > 0 && std0 != 0.f && std1 != 0.f && std2 != 0.f, VACNIC_BAD_SHAPE, "image_u8_normalize: bad shape / zero std");
  const long total = B * 3 * H * W;
  if (total == 0) return VACNIC_OK;
  long nb = (total / 4 + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(image_u8_normalize_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, flip, dst, (long)B, (int)H,
                     (int)W, mean0, mean1, mean2, std0, std1, std2);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

namespace {
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ p, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0u;
}
}  // namespace

// zero-fill on the caller's stream — the fp32 accumulators of split-K GEMMs, the arrival counters of the split-K fix-up.  A memset
// (no kernel) on an ordinary stream; while the stream is being CAPTURED into a hipGraph a fill kernel instead (VACNIC_ZERO_MEMSET_IN_GRAPHS=1
// keeps the memset node): see DESIGN section 0, item 5 (c).
extern "C" int vacnic_zero_bytes(void* ptr, int64_t bytes, void* stream) {
  VPLAN_REC(vacnic_zero_bytes, ptr, bytes, stream);
  VCHECK(ptr && bytes >= 0, VACNIC_BAD_SHAPE, "zero_bytes: bad operand");
  if (bytes == 0) return VACNIC_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  static const bool memset_in_graphs = getenv("VACNIC_ZERO_MEMSET_IN_GRAPHS") && atoi(getenv("VACNIC_ZERO_MEMSET_IN_GRAPHS"));
  if (!memset_in_graphs && (((uintptr_t)ptr | (uintptr_t)bytes) & 3) == 0 && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess &&
      cs == hipStreamCaptureStatusActive) {
    const long n = (long)(bytes >> 2);
    long nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (unsigned*)ptr, n);
    VLAUNCH_CHECK();
    return VACNIC_OK;
  }
  if (hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream) != hipSuccess) {
    vacnic_set_error("zero_bytes: hipMemsetAsync failed");
    return VACNIC_HIP_ERROR;
  }
  return VACNIC_OK;
}
