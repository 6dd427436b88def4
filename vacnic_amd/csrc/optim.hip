// Fused AdamW over one flat fp32 arena + LR schedule on device (gfx950, HBM-bound: 16 B/param read,
// 14 B/param written).  Replaces torch.optim.AdamW's per-tensor loop (TRAIN:91,372-374) and
// get_linear_schedule_with_warmup (TRAIN:99-107).  lr/step live in device memory so the whole
// step is hipGraph-capturable.
#include "common.h"
#include <cstdlib>

namespace {

// hyper[0] = lr for this step, hyper[1] = step count t (float, 1-based after the increment)
__global__ void lr_step_kernel(float* hyper, float base_lr, float warmup, float total, unsigned long long* rng_counter) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (rng_counter) *rng_counter += 1ull;
    const float k = hyper[1];                 // optimizer steps taken so far == LambdaLR's current_step
    float lam;
    if (k < warmup) lam = k / fmaxf(1.f, warmup);
    else lam = fmaxf(0.f, (total - k) / fmaxf(1.f, total - warmup));
    hyper[0] = base_lr * lam;
    hyper[1] = k + 1.f;
  }
}

template <int UNR>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ pb,
                                                    const float* __restrict__ hyper, long n4, float b1, float b2,
                                                    float eps, float wd, float gscale, int zero_grad,
                                                    const float* __restrict__ clip) {
  const float lr = hyper[0], t = hyper[1];
  if (clip) gscale *= clip[0];                  // clip_grad_norm_'s coefficient, computed on device by grad_clip_coef
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
  const long stride = (long)gridDim.x * blockDim.x;
  // Launch shape (profiles/r1_adamw_microbench.txt): ALONE on the GPU this pass is fastest with one 4-wave workgroup per CU
  // (256 workgroups: 6.2 TB/s; 2048: 4.6; 65536: 5.0 on a 440M-parameter arena) — but in the training step it shares the GPU
  // with the next step's frozen-tower graphs, where a 256-workgroup launch loses its share of CUs and bandwidth (step 74.0 ms vs
  // 71.1 ms).  It sits on the critical path, so the launch is wide (65536 workgroups: 70.9 ms).  UNR (loads in flight per
  // stream per lane) made no difference at 1, 2, 4.
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * UNR) {
    f32x4 pv[UNR], gv[UNR], mv[UNR], vv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long i = i0 + u * stride;
      if (i < n4) { pv[u] = ((f32x4*)p)[i]; gv[u] = ((f32x4*)g)[i]; mv[u] = ((f32x4*)m)[i]; vv[u] = ((f32x4*)v)[i]; }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long i = i0 + u * stride;
      if (i >= n4) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gr = gv[u][e] * gscale;
        float pe = pv[u][e] * decay;
        const float me = b1 * mv[u][e] + (1.f - b1) * gr;
        const float ve = b2 * vv[u][e] + (1.f - b2) * gr * gr;
        pe -= step_size * me / (sqrtf(ve) * inv_sqrt_bc2 + eps);
        pv[u][e] = pe; mv[u][e] = me; vv[u][e] = ve;
      }
      ((f32x4*)p)[i] = pv[u]; ((f32x4*)m)[i] = mv[u]; ((f32x4*)v)[i] = vv[u];
      if (zero_grad) ((f32x4*)g)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (pb) ((u32x2*)pb)[i] = (u32x2){pack2bf(pv[u][0], pv[u][1]), pack2bf(pv[u][2], pv[u][3])};
    }
  }
}

// clip_grad_norm_ (TRAIN:365-366), pass 1: fixed-grid partial sums of (g*gscale)^2 — fixed grid + fixed
// per-thread order + a fixed-shape tree, so the norm is bit-reproducible run to run (no float atomics).
constexpr int kNormBlocks = 1024;
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, long n4, float gscale,
                                                         float* __restrict__ partials) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 gv = ((const f32x4*)g)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float x = gv[e] * gscale; acc = fmaf(x, x, acc); }
  }
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// pass 2 (one workgroup): out[0] = min(1, max_norm / (||g|| + 1e-6)), out[1] = ||g||   (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void grad_clip_coef_kernel(const float* __restrict__ partials, int nparts, float max_norm,
                                                             float* __restrict__ out) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += partials[i];
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    out[0] = fminf(1.f, max_norm / (norm + 1e-6f));
    out[1] = norm;
  }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = ((const f32x4*)s)[i];
    ((u32x2*)d)[i] = (u32x2){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) d[i] = f2bf(s[i]);
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = bf2f(s[i]);
}

inline unsigned grid_for(long work) {
  long b = (work + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int vacnic_lr_step(float* hyper, float base_lr, float warmup_steps, float total_steps, uint64_t* rng_counter, void* stream) {
  VPLAN_REC(vacnic_lr_step, hyper, base_lr, warmup_steps, total_steps, rng_counter, stream);
  VCHECK(hyper, VACNIC_BAD_SHAPE, "lr_step: null hyper");
  hipLaunchKernelGGL(lr_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper, base_lr, warmup_steps, total_steps, (unsigned long long*)rng_counter);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_adamw(const vacnic_adamw_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_adamw, a, stream);
  VCHECK(a && a->p && a->g && a->m && a->v && a->hyper, VACNIC_BAD_SHAPE, "adamw: null operand");
  VCHECK((a->n & 3) == 0, VACNIC_BAD_SHAPE, "adamw: n=%ld must be a multiple of 4 (pad the arena)", (long)a->n);
  VCHECK(aligned16(a->p) && aligned16(a->g) && aligned16(a->m) && aligned16(a->v) && (!a->p_bf16 || (((uintptr_t)a->p_bf16) & 7) == 0),
         VACNIC_MISALIGNED, "adamw: arenas must be 16-byte aligned");
  if (a->n == 0) return VACNIC_OK;
  const long n4 = a->n >> 2;
  unsigned blocks = 65536;                     // wide launch (see the kernel comment); small arenas: one element per thread
  const long need = (n4 + 255) / 256;
  if (const char* e = getenv("VACNIC_ADAMW_BLOCKS")) { const long bb = atol(e); if (bb > 0) blocks = (unsigned)bb; }   // A/B knob
  if (need < blocks) blocks = (unsigned)(need < 1 ? 1 : need);
  hipLaunchKernelGGL(adamw_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a->p, a->g, a->m, a->v,
                     (bf16_t*)a->p_bf16, a->hyper, n4, a->beta1, a->beta2, a->eps, a->weight_decay, a->grad_scale, a->zero_grad,
                     a->clip_coef);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_grad_clip_coef(const float* g, int64_t n, float grad_scale, float max_norm, float* partials,
                                     float* out, void* stream) {
  VPLAN_REC(vacnic_grad_clip_coef, g, n, grad_scale, max_norm, partials, out, stream);
  VCHECK(g && partials && out, VACNIC_BAD_SHAPE, "grad_clip_coef: null operand");
  VCHECK((n & 3) == 0 && aligned16(g), VACNIC_BAD_SHAPE, "grad_clip_coef: arena must be 16-byte aligned, n=%ld a multiple of 4", (long)n);
  VCHECK(max_norm > 0.f, VACNIC_BAD_SHAPE, "grad_clip_coef: max_norm must be > 0");
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(kNormBlocks), dim3(256), 0, (hipStream_t)stream, g, (long)(n >> 2), grad_scale, partials);
  VLAUNCH_CHECK();
  hipLaunchKernelGGL(grad_clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, kNormBlocks, max_norm, out);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
  VPLAN_REC(vacnic_cast_f32_bf16, src, dst, n, stream);
  VCHECK(src && dst, VACNIC_BAD_SHAPE, "cast: null operand");
  VCHECK(aligned16(src) && (((uintptr_t)dst) & 7) == 0, VACNIC_MISALIGNED, "cast_f32_bf16: misaligned");
  if (n == 0) return VACNIC_OK;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
extern "C" int vacnic_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream) {
  VPLAN_REC(vacnic_cast_bf16_f32, src, dst, n, stream);
  VCHECK(src && dst, VACNIC_BAD_SHAPE, "cast: null operand");
  if (n == 0) return VACNIC_OK;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, (long)n);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
