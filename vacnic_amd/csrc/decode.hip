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

// ---- threshold variant (no forced token): one 1024-thread block per row, two streaming passes over the row.
//   pass 1: online log-sum-exp + every thread's maximum (banned / suppressed tokens looked up in an LDS bitmap of the vocabulary)
//   threshold T = K-th largest thread maximum — K distinct elements are >= T, so the K-th best score is >= T
//   pass 2: the few elements >= T are appended to an LDS candidate list; K rounds of a block-wide arg-max under the strict
//           order (score desc, index asc) emit the result.  If more than TOPK_CAND elements tie at the threshold (flat
//           logits) the block falls back to K arg-max passes over the row.
// Same result as beam_topk_kernel above; it replaces 10 passes x 50 ban compares per logit (578 us per position at V = 50267,
// beam 5) — and a register-resident insertion sort whose divergent inserts cost 85 us — by ~10 us.
constexpr int TOPK_THREADS = 1024;
constexpr int TOPK_CAND = 1024;

__device__ __forceinline__ void block_argmax(float& bv, int& bi, float* rv, int* ri, int lane, int wave) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  __syncthreads();
  if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
  __syncthreads();
  bv = rv[0]; bi = ri[0];
#pragma unroll
  for (int w = 1; w < TOPK_THREADS / 64; ++w) {
    const float ov = rv[w]; const int oi = ri[w];
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
}

template <bool F32>
__global__ __launch_bounds__(TOPK_THREADS) void beam_topk_fast_kernel(const void* __restrict__ logits, const float* __restrict__ beam_scores,
                                                                      const int* __restrict__ bans, int n_ban, int eos, int suppress_eos,
                                                                      float* __restrict__ top_val, int* __restrict__ top_idx,
                                                                      int V, long ldl_, int K) {
  extern __shared__ unsigned banbits[];               // ceil(V/32) words, then scratch
  const int nwords = (V + 31) >> 5;
  float* rv = (float*)(banbits + nwords);             // [16]
  int* ri = (int*)(rv + 16);                          // [16]
  float* tm = (float*)(ri + 16);                      // [1024] thread maxima
  float* cand_v = tm + TOPK_THREADS;                  // [TOPK_CAND]
  int* cand_i = (int*)(cand_v + TOPK_CAND);           // [TOPK_CAND]
  int* cnt = cand_i + TOPK_CAND;                      // [1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long r = blockIdx.x;
  const char* row = (const char*)logits + r * ldl_ * (F32 ? 4 : 2);
  for (int w = tid; w < nwords; w += TOPK_THREADS) banbits[w] = 0u;
  if (tid == 0) *cnt = 0;
  __syncthreads();
  if (bans) {
    for (int b = tid; b < n_ban; b += TOPK_THREADS) {
      const int t = bans[r * n_ban + b];
      if (t >= 0 && t < V) atomicOr(&banbits[t >> 5], 1u << (t & 31));
    }
  }
  if (suppress_eos && tid == 0 && eos >= 0 && eos < V) atomicOr(&banbits[eos >> 5], 1u << (eos & 31));
  __syncthreads();
  const bool vec = F32 && ((uintptr_t)row & 15) == 0;
  const int nvec = vec ? (V >> 2) : 0;
  auto masked = [&](int j, float v) -> float { return (banbits[j >> 5] >> (j & 31)) & 1u ? -INFINITY : v; };
  // generic row walk: f(j, raw logit) over this thread's share, 16-byte loads four deep where the layout allows
  auto walk = [&](auto&& f) {
    for (int v0 = tid; v0 < nvec; v0 += 4 * TOPK_THREADS) {
      f32x4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int vi = v0 + u * TOPK_THREADS;
        x[u] = vi < nvec ? ((const f32x4*)row)[vi] : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int vi = v0 + u * TOPK_THREADS;
        if (vi < nvec) {
#pragma unroll
          for (int e = 0; e < 4; ++e) f(vi * 4 + e, x[u][e]);
        }
      }
    }
    for (int j = (nvec << 2) + tid; j < V; j += TOPK_THREADS) f(j, ldl<F32>(row, j));
  };
  // ---- pass 1
  float m = -INFINITY, ssum = 0.f, tmax = -INFINITY;
  walk([&](int j, float v) {
    const float nm = fmaxf(m, v);
    ssum = ssum * __expf(m - nm) + __expf(v - nm);
    m = nm;
    tmax = fmaxf(tmax, masked(j, v));
  });
  float gm = wave_max(m);
  if (lane == 0) rv[wave] = gm;
  __syncthreads();
  gm = rv[0];
#pragma unroll
  for (int w = 1; w < TOPK_THREADS / 64; ++w) gm = fmaxf(gm, rv[w]);
  const float part = wave_sum(m == -INFINITY ? 0.f : ssum * __expf(m - gm));
  __syncthreads();
  if (lane == 0) rv[wave] = part;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < TOPK_THREADS / 64; ++w) tot += rv[w];
  const float lse = gm + logf(tot), base = beam_scores ? beam_scores[r] : 0.f;
  // ---- threshold: K-th largest thread maximum
  float mine = tmax, T = -INFINITY;
  for (int k = 0; k < K; ++k) {
    float bv = mine; int bi = tid;
    block_argmax(bv, bi, rv, ri, lane, wave);
    if (bi == tid) mine = -INFINITY;
    T = bv;
  }
  __syncthreads();
  // ---- pass 2: candidates
  bool overflow = T == -INFINITY;
  if (!overflow) {
    walk([&](int j, float v) {
      const float sv = masked(j, v);
      if (sv >= T) {
        const int pos = atomicAdd(cnt, 1);
        if (pos < TOPK_CAND) { cand_v[pos] = sv; cand_i[pos] = j; }
      }
    });
    __syncthreads();
    overflow = *cnt > TOPK_CAND;
  }
  if (!overflow) {
    const int n = *cnt;
    float cv = tid < n ? cand_v[tid] : -INFINITY;
    int ci = tid < n ? cand_i[tid] : 0x7fffffff;
    for (int k = 0; k < K; ++k) {
      float bv = cv; int bi = ci;
      block_argmax(bv, bi, rv, ri, lane, wave);
      if (bi == ci && ci != 0x7fffffff) { cv = -INFINITY; ci = 0x7fffffff; }
      if (tid == 0) {
        top_val[r * K + k] = bv == -INFINITY ? -INFINITY : bv - lse + base;
        top_idx[r * K + k] = bi == 0x7fffffff ? -1 : bi;
      }
    }
    return;
  }
  // ---- fallback: K arg-max passes over the row (massive ties / fewer than K finite scores)
  float last_v = INFINITY; int last_i = -1;
  for (int k = 0; k < K; ++k) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    walk([&](int j, float v) {
      const float sv = masked(j, v);
      const bool after = (sv < last_v) || (sv == last_v && j > last_i);
      if (after && (sv > bv || (sv == bv && j < bi))) { bv = sv; bi = j; }
    });
    block_argmax(bv, bi, rv, ri, lane, wave);
    last_v = bv; last_i = bi;
    if (tid == 0) {
      top_val[r * K + k] = bv == -INFINITY ? -INFINITY : bv - lse + base;
      top_idx[r * K + k] = bi == 0x7fffffff ? -1 : bi;
    }
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
  if (forced_token < 0 && K <= TOPK_THREADS && V <= 262144) {
    const size_t lds = ((size_t)((V + 31) >> 5) + 32 + TOPK_THREADS + 2 * TOPK_CAND + 4) * 4;
    if (logits_f32)
      hipLaunchKernelGGL(beam_topk_fast_kernel<true>, dim3((unsigned)R), dim3(TOPK_THREADS), lds, s, logits, beam_scores, bans, n_ban,
                         eos, suppress_eos, top_val, top_idx, (int)V, (long)ldl, K);
    else
      hipLaunchKernelGGL(beam_topk_fast_kernel<false>, dim3((unsigned)R), dim3(TOPK_THREADS), lds, s, logits, beam_scores, bans, n_ban,
                         eos, suppress_eos, top_val, top_idx, (int)V, (long)ldl, K);
    VLAUNCH_CHECK();
    return VACNIC_OK;
  }
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
