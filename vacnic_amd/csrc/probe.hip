// Hardware-layout probes (tests only).  They compute small products under the lane maps the
// production kernels assume (guide §3 / T10) and dump raw results so a test can compare against
// numpy on exact integer data.  out layout (floats):
//   [0,256)      C16 = A16 . B16   (16x16, A 16x32, B 32x16)   A[i][k]=(i*3+k)%7-3, B[k][j]=(k*5+j*2)%9-4
//   [256,1280)   C32 = A32 . B32   (32x32, A 32x16, B 16x32)   same formulas
//   [1280,1536)  tr-read dump: lane l, element e -> value read with the [4 rows][16 cols] block model
//                from a [16][16] LDS tile holding row*16+col
//   [1536,1600)  buffer_load..lds OOB probe: 64 floats, lanes >= 32 are out of range -> must be 0
//   [1600,2624)  acc-as-operand check: Y = A2 . X where X = C32 reused as B operand (k order permuted)
#include "common.h"

namespace {
__device__ __forceinline__ float fa(int i, int k) { return (float)((i * 3 + k) % 7 - 3); }
__device__ __forceinline__ float fb(int k, int j) { return (float)((k * 5 + j * 2) % 9 - 4); }

__global__ __launch_bounds__(64) void probe_kernel(float* out, const float* oob_src) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[16 * 16];
  __shared__ __attribute__((aligned(16))) float stage[64 * 4];
  const int l = threadIdx.x;
  {  // 16x16x32
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * (l >> 4) + j;
      a[j] = (short)f2bf(fa(l & 15, k));
      b[j] = (short)f2bf(fb(k, l & 15));
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
  }
  f32x16 c32 = (f32x16)(0.f);
  {  // 32x32x16
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * (l >> 5) + j;
      a[j] = (short)f2bf(fa(l & 31, k));
      b[j] = (short)f2bf(fb(k, l & 31));
    }
    c32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c32, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[256 + ((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c32[r];
  }
  {  // tr read
    for (int i = l; i < 256; i += 64) tile[i] = f2bf((float)i);
    __syncthreads();
    const int i = l & 15;
    const bf16_t* addr = tile + (i >> 2) * 16 + 4 * (i & 3);   // block rows 0..3 for every group
    bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)LDS_PTR(addr + (l >> 4) * 64));
    for (int e = 0; e < 4; ++e) out[1280 + l * 4 + e] = bf2f((bf16_t)t[e]);
  }
  {  // LDS-DMA with out-of-range lanes
    for (int i = l; i < 256; i += 64) stage[i] = -1.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)oob_src, 0, 32 * 16, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(stage), 16, l < 32 ? l * 16 : 0x7ffffff0, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[1536 + l] = stage[l * 4];
  }
  {  // accumulator tile as B operand: Y[i][j] = sum_k A2[i][k] X[k][j], X = c32 (32x32), two k-steps of 16
    f32x16 y = (f32x16)(0.f);
    for (int s = 0; s < 2; ++s) {
      bf16x8 xb, a2;
      for (int j = 0; j < 8; ++j) {
        xb[j] = (short)f2bf(c32[8 * s + j]);
        const int k = 16 * s + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
        a2[j] = (short)f2bf((float)(((l & 31) + 2 * k) % 5 - 2));
      }
      y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) out[1600 + ((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = y[r];
  }
}
}  // namespace

extern "C" int vacnic_probe_layouts(float* out, const float* src128, int64_t n_out, void* stream) {
  VPLAN_REC(vacnic_probe_layouts, out, src128, n_out, stream);
  VCHECK(out && src128 && n_out >= 2624, VACNIC_BAD_SHAPE, "probe: need >= 2624 output floats and a 128-float source");
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, src128);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
