// Decode-side kernels (SURVEY §8 row a15 / BASELINE config 5: beam search, max_length 50, length_penalty 2.0).
//   beam_topk   : log_softmax over the vocabulary + HF logits processors (NoRepeatNGram bans, MinLength,
//                 ForcedEOS) + running beam score, then the top-K candidates per beam row
//                 (transformers 4.18 GenerationMixin.beam_search: log_softmax -> processors -> + beam_scores -> topk)
//   gather_rows : KV-cache beam reorder (_reorder_cache, MFULL:2066-2074) for all layers in one launch
#include "common.h"

namespace {

template <bool F32>
__device__ __forceinline__ float ldl(const void* row, int j) {
  return F32 ? ((const float*)row)[j] : bf2f(((const bf16_t*)row)[j]);
}

// one block per row; K rounds of block-wide arg-max under the strict order (score desc, index asc)
template <bool F32>
__global__ __launch_bounds__(256) void beam_topk_kernel(const void* __restrict__ logits, const float* __restrict__ beam_scores,
                                                        const int* __restrict__ bans, int n_ban, int eos, int suppress_eos,
                                                        int forced_token, float* __restrict__ top_val, int* __restrict__ top_idx,
                                                        int V, long ldl_, int K) {
  __shared__ float rv[256];
  __shared__ int ri[256];
  __shared__ float s_lse;
  const long r = blockIdx.x;
  const char* row = (const char*)logits + r * ldl_ * (F32 ? 4 : 2);
  const int* ban = bans ? bans + r * n_ban : nullptr;
  // ---- log-sum-exp of the raw logits
  float m = -INFINITY, s = 0.f;
  for (int j = threadIdx.x; j < V; j += 256) {
    const float v = ldl<F32>(row, j);
    const float nm = fmaxf(m, v);
    s = s * __expf(m - nm) + __expf(v - nm);
    m = nm;
  }
  rv[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) rv[threadIdx.x] = fmaxf(rv[threadIdx.x], rv[threadIdx.x + o]); __syncthreads(); }
  const float gm = rv[0];
  __syncthreads();
  rv[threadIdx.x] = m == -INFINITY ? 0.f : s * __expf(m - gm);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) rv[threadIdx.x] += rv[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) s_lse = gm + logf(rv[0]);
  __syncthreads();
  const float lse = s_lse, base = beam_scores ? beam_scores[r] : 0.f;

  auto score = [&](int j) -> float {
    if (forced_token >= 0) return j == forced_token ? base : -INFINITY;       // ForcedEOSTokenLogitsProcessor
    if (suppress_eos && j == eos) return -INFINITY;                            // MinLengthLogitsProcessor
    for (int b = 0; b < n_ban; ++b) if (ban && ban[b] == j) return -INFINITY;  // NoRepeatNGramLogitsProcessor
    return ldl<F32>(row, j) - lse + base;
  };
  float last_v = INFINITY; int last_i = -1;
  for (int k = 0; k < K; ++k) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < V; j += 256) {
      const float v = score(j);
      const bool after = (v < last_v) || (v == last_v && j > last_i);        // not selected yet
      if (after && (v > bv || (v == bv && j < bi))) { bv = v; bi = j; }
    }
    rv[threadIdx.x] = bv; ri[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
        const float ov = rv[threadIdx.x + o]; const int oi = ri[threadIdx.x + o];
        if (ov > rv[threadIdx.x] || (ov == rv[threadIdx.x] && oi < ri[threadIdx.x])) { rv[threadIdx.x] = ov; ri[threadIdx.x] = oi; }
      }
      __syncthreads();
    }
    last_v = rv[0]; last_i = ri[0];
    if (threadIdx.x == 0) { top_val[r * K + k] = last_v; top_idx[r * K + k] = last_i == 0x7fffffff ? -1 : last_i; }
    __syncthreads();
  }
}

// dst[r][:] = src[idx[r]][:], rows of row_bytes (multiple of 16)
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                          const int64_t* __restrict__ idx, long rows, long chunks) {
  const long total = rows * chunks;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / chunks, c = i % chunks;
    ((u32x4*)dst)[r * chunks + c] = ((const u32x4*)src)[idx[r] * chunks + c];
  }
}

}  // namespace

extern "C" int vacnic_beam_topk(const void* logits, const float* beam_scores, const int32_t* bans, int32_t n_ban, int32_t eos,
                                int32_t suppress_eos, int32_t forced_token, float* top_val, int32_t* top_idx, int64_t R, int64_t V,
                                int64_t ldl, int32_t K, int32_t logits_f32, void* stream) {
  VCHECK(logits && top_val && top_idx, VACNIC_BAD_SHAPE, "beam_topk: null operand");
  VCHECK(V > 0 && ldl >= V && K > 0 && K <= V, VACNIC_BAD_SHAPE, "beam_topk: bad V/ldl/K");
  if (R == 0) return VACNIC_OK;
  hipStream_t s = (hipStream_t)stream;
  if (logits_f32)
    hipLaunchKernelGGL(beam_topk_kernel<true>, dim3((unsigned)R), dim3(256), 0, s, logits, beam_scores, bans, n_ban, eos,
                       suppress_eos, forced_token, top_val, top_idx, (int)V, (long)ldl, K);
  else
    hipLaunchKernelGGL(beam_topk_kernel<false>, dim3((unsigned)R), dim3(256), 0, s, logits, beam_scores, bans, n_ban, eos,
                       suppress_eos, forced_token, top_val, top_idx, (int)V, (long)ldl, K);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_gather_rows(const void* src, void* dst, const int64_t* idx, int64_t rows, int64_t row_bytes, void* stream) {
  VCHECK(src && dst && idx, VACNIC_BAD_SHAPE, "gather_rows: null operand");
  VCHECK((row_bytes & 15) == 0 && aligned16(src) && aligned16(dst), VACNIC_MISALIGNED, "gather_rows: rows must be 16-byte multiples");
  if (rows * row_bytes == 0) return VACNIC_OK;
  const long chunks = row_bytes / 16;
  long nb = (rows * chunks + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, idx,
                     (long)rows, chunks);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
