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

// wave-wide best under the strict order (value desc, index asc), result in every lane: four DPP compare-exchange steps (quad
// xor 1, xor 2, mirror within 8, mirror within 16: ~8 cycles each, no ds_bpermute) and a scalar pass over the four rows
__device__ __forceinline__ void wave_argbest(float& bv, int& bi) {
#define VAC_STEP(CTRL_)                                                                                                        \
  {                                                                                                                            \
    const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, bv), CTRL_, 0xf, 0xf, true)); \
    const int oi = __builtin_amdgcn_update_dpp(0, bi, CTRL_, 0xf, 0xf, true);                                                  \
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }                                                                \
  }
  VAC_STEP(0xB1) VAC_STEP(0x4E) VAC_STEP(0x141) VAC_STEP(0x140)
#undef VAC_STEP
  float rv_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), 0));
  int ri_ = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
  for (int row = 1; row < 4; ++row) {
    const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), 16 * row));
    const int oi = __builtin_amdgcn_readlane(bi, 16 * row);
    if (ov > rv_ || (ov == rv_ && oi < ri_)) { rv_ = ov; ri_ = oi; }
  }
  bv = rv_; bi = ri_;
}

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
  // ---- threshold: a lower bound of the K-th best score from K distinct elements
  float T = -INFINITY;
  if (K <= TOPK_THREADS / 64) {
    // K-th largest WAVE maximum (16 waves; the expected number of elements above it is ~15 at V = 50 K): one workgroup barrier and
    // K wave-level rounds instead of K block-wide arg-max rounds with two barriers each (~1.2 us apiece)
    float wv = tmax; int wi = tid;
    wave_argbest(wv, wi);
    __syncthreads();
    if (lane == 0) rv[wave] = wv;
    __syncthreads();
    float x = lane < TOPK_THREADS / 64 ? rv[lane] : -INFINITY;
    for (int k = 0; k < K; ++k) {
      float bv = x; int bi = lane;
      wave_argbest(bv, bi);
      if (bi == lane) x = -INFINITY;
      T = bv;
    }
  } else {
    float mine = tmax;                                  // K-th largest thread maximum
    for (int k = 0; k < K; ++k) {
      float bv = mine; int bi = tid;
      block_argmax(bv, bi, rv, ri, lane, wave);
      if (bi == tid) mine = -INFINITY;
      T = bv;
    }
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
  if (!overflow && *cnt <= 256) {
    // the usual case (a few dozen candidates): wave 0 alone picks the K best, four candidates per lane, no workgroup barriers
    if (wave == 0) {
      const int n = *cnt;
      float cv[4]; int ci[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = lane + 64 * u;
        cv[u] = c < n ? cand_v[c] : -INFINITY;
        ci[u] = c < n ? cand_i[c] : 0x7fffffff;
      }
      for (int k = 0; k < K; ++k) {
        float bv = cv[0]; int bi = ci[0];
#pragma unroll
        for (int u = 1; u < 4; ++u)
          if (cv[u] > bv || (cv[u] == bv && ci[u] < bi)) { bv = cv[u]; bi = ci[u]; }
        wave_argbest(bv, bi);
        if (bi != 0x7fffffff) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (ci[u] == bi) { cv[u] = -INFINITY; ci[u] = 0x7fffffff; }
        }
        if (lane == 0) {
          top_val[r * K + k] = bv == -INFINITY ? -INFINITY : bv - lse + base;
          top_idx[r * K + k] = bi == 0x7fffffff ? -1 : bi;
        }
      }
    }
    return;
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

// ---------------------------------------------------------------------------------------------------------------------
// LM head + log_softmax + logits processors + top-K of the single-token decoder in TWO launches, without [R][V] logits in HBM
// (MFULL:1997 lm_head + final_logits_bias; transformers 4.18 beam_search: log_softmax -> processors -> + beam_scores -> topk).
// What it replaces per position: gemm_skinny (12.5 K one-wave workgroups, 2 loads in flight per lane: 35 us for 103 MB) + a
// 5-workgroup top-k that reads the logits back twice (26 us).
//   part kernel : G workgroups x 4 waves; a workgroup owns a contiguous range of <= 128 vocabulary columns and walks it in tiles of
//                 16 columns.  The product is on the matrix cores (v_mfma_f32_16x16x32_bf16: A = 16 embedding rows, B = the hidden
//                 rows, zero beyond R) — the VALU form (4 columns per wave like gemm_skinny, 8 rows) spends ~11 us per SIMD on
//                 bf16 -> fp32 conversions and FMAs and was SLOWER than the chain it replaces (44 us).  The 4 waves split K (wave w
//                 takes the 32-wide K steps w, w + 4, ...: 8 x 16-byte loads per lane and tile, the next TWO tiles in flight), their
//                 partial tiles meet in LDS (one barrier per tile) and wave 0 finishes: + bias, an online (max, sum-exp) of the RAW
//                 logits per lane, the masked logits (n-gram bans, suppressed EOS) into an LDS row buffer.  At the end one wave per
//                 row picks the workgroup's K2 best under the order (score desc, token asc).
//                 Output (field-major so that the merge reads contiguously): part[(r * (2 + 2 K2) + f) * G + g], f = max | sum-exp |
//                 K2 values | K2 ids.
//   merge kernel: one workgroup per row: log-sum-exp over the G partials, then K2 rounds of a workgroup-wide arg-best over the
//                 G x K2 candidates (token ids are distinct, so "after the last pick" under the strict order excludes picks).
// A forced token (ForcedBOS / ForcedEOS) needs neither logits nor lse: the merge kernel alone writes {forced: beam score, rest -inf}.
constexpr int LT_MAXCOLS = 128;      // vocabulary columns per workgroup of the part kernel

struct LmTopkP {
  const bf16_t* h; const bf16_t* emb; const float* bias; float* logits; float* part;
  const int* bans; const float* beam_scores; float* top_val; int* top_idx;
  int R, V, K, ldw, ldl, n_ban, eos, suppress_eos, forced, K2, G, tpw;      // tpw: 16-column tiles per workgroup
};

template <int KS>                     // K steps (32 wide) per wave: K <= 128 * KS
__global__ __launch_bounds__(256) void lmhead_part_kernel(LmTopkP p) {
  __shared__ float sc[8][LT_MAXCOLS];                               // masked logits of this workgroup's columns
  __shared__ unsigned banw[8][LT_MAXCOLS / 32];
  __shared__ __attribute__((aligned(16))) float red[2][4][256];     // the 4 waves' partial tiles, double-buffered by tile parity
  __shared__ float rm[64], rs[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, q = lane >> 4;
  const int nst = (p.K + 31) >> 5;
  const int t_lo = blockIdx.x * p.tpw, c_lo = t_lo * 16;
  int ncol = p.V - c_lo;
  if (ncol > p.tpw * 16) ncol = p.tpw * 16;
  if (ncol < 0) ncol = 0;
  const int t_hi = t_lo + ((ncol + 15) >> 4);
  auto wload = [&](int t, u32x4 (&w)[KS]) {
    const int n = t * 16 + col;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int k = (wave + 4 * i) * 32 + q * 8;
      w[i] = (k < p.K && n < p.V) ? *(const u32x4*)(p.emb + (size_t)n * p.ldw + k) : (u32x4){0u, 0u, 0u, 0u};
    }
  };
  // the first two tiles' weights are requested before anything else: they do not depend on the hidden rows
  u32x4 wA[KS], wB[KS];
  if (t_lo < t_hi) wload(t_lo, wA);
  if (t_lo + 1 < t_hi) wload(t_lo + 1, wB);
  bf16x8 xa[KS];                                                    // B operand: hidden row `col`, this wave's K steps
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int k = (wave + 4 * i) * 32 + q * 8;
    const u32x4 v = (k < p.K && col < p.R) ? *(const u32x4*)(p.h + (size_t)col * p.K + k) : (u32x4){0u, 0u, 0u, 0u};
    xa[i] = __builtin_bit_cast(bf16x8, v);
  }
  for (int i = tid; i < 8 * (LT_MAXCOLS / 32); i += 256) (&banw[0][0])[i] = 0u;
  for (int i = tid; i < 8 * LT_MAXCOLS; i += 256) (&sc[0][0])[i] = -INFINITY;
  __syncthreads();
  if (p.bans) {
    for (int i = tid; i < p.R * p.n_ban; i += 256) {
      const int t = p.bans[i] - c_lo;
      if (t >= 0 && t < ncol) atomicOr(&banw[i / p.n_ban][t >> 5], 1u << (t & 31));
    }
  }
  if (p.suppress_eos && tid < p.R && p.eos >= c_lo && p.eos < c_lo + ncol) atomicOr(&banw[tid][(p.eos - c_lo) >> 5], 1u << ((p.eos - c_lo) & 31));
  __syncthreads();
  float mrun = -INFINITY, srun = 0.f;                               // wave 0, lanes with col < R: row `col`, columns q * 4 .. + 3 of every tile
  const bool fin = wave == 0 && col < p.R;
  auto tile = [&](u32x4 (&cur)[KS], int t, int par, int t_next) {
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (fin && p.bias) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int n = t * 16 + q * 4 + v;
        if (n < p.V) bv[v] = p.bias[n];
      }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KS; ++i)
      if ((wave + 4 * i) < nst) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur[i]), xa[i], acc, 0, 0, 0);
    if (t_next < t_hi) wload(t_next, cur);
    *(f32x4*)&red[par][wave][lane * 4] = acc;
    __syncthreads();
    if (fin) {
      const f32x4 s0 = *(const f32x4*)&red[par][0][lane * 4], s1 = *(const f32x4*)&red[par][1][lane * 4];
      const f32x4 s2 = *(const f32x4*)&red[par][2][lane * 4], s3 = *(const f32x4*)&red[par][3][lane * 4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int n = t * 16 + q * 4 + v;
        if (n < p.V) {
          const float x = ((s0[v] + s1[v]) + s2[v]) + s3[v] + bv[v];
          if (p.logits) p.logits[(size_t)col * p.ldl + n] = x;
          const float nm = fmaxf(mrun, x);
          srun = srun * __expf(mrun - nm) + __expf(x - nm);
          mrun = nm;
          const int j = n - c_lo;
          sc[col][j] = (banw[col][j >> 5] >> (j & 31)) & 1u ? -INFINITY : x;
        }
      }
    }
  };
  for (int t = t_lo; t < t_hi; t += 2) {
    tile(wA, t, 0, t + 2);
    if (t + 1 < t_hi) tile(wB, t + 1, 1, t + 3);
  }
  if (wave == 0) { rm[lane] = mrun; rs[lane] = srun; }
  __syncthreads();
  const int W = 2 + 2 * p.K2;
  for (int r = wave; r < p.R; r += 4) {
    float* out = p.part + (size_t)r * W * p.G + blockIdx.x;         // field f at out[f * G]
    // log-sum-exp pieces of row r: wave 0's lanes r, r + 16, r + 32, r + 48
    const float me = lane < 4 ? rm[r + 16 * lane] : -INFINITY;
    const float se = lane < 4 ? rs[r + 16 * lane] : 0.f;
    const float gm = wave_max(me);
    const float tot = wave_sum(me == -INFINITY ? 0.f : se * __expf(me - gm));
    if (lane == 0) { out[0] = gm; out[p.G] = tot; }
    float cv[2]; int ci[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = lane + 64 * u;
      cv[u] = j < ncol ? sc[r][j] : -INFINITY;
      ci[u] = j < ncol ? c_lo + j : 0x7fffffff;
    }
    for (int k = 0; k < p.K2; ++k) {
      float bv = cv[0]; int bi = ci[0];
      if (cv[1] > bv || (cv[1] == bv && ci[1] < bi)) { bv = cv[1]; bi = ci[1]; }
      wave_argbest(bv, bi);
      if (bi != 0x7fffffff) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (ci[u] == bi) { cv[u] = -INFINITY; ci[u] = 0x7fffffff; }
      }
      if (lane == 0) {
        const bool none = bi == 0x7fffffff || bv == -INFINITY;       // banned / missing columns never become candidates
        out[(size_t)(2 + k) * p.G] = none ? -INFINITY : bv;
        ((int*)out)[(size_t)(2 + p.K2 + k) * p.G] = none ? 0x7fffffff : bi;
      }
    }
  }
}

__global__ __launch_bounds__(256) void lmhead_merge_kernel(LmTopkP p) {
  extern __shared__ __attribute__((aligned(16))) char lt_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = blockIdx.x, K2 = p.K2, W = 2 + 2 * K2, G = p.G;
  const float base = p.beam_scores ? p.beam_scores[r] : 0.f;
  if (p.forced >= 0) {
    // ForcedBOS / ForcedEOS: the forced token scores `base`, everything else -inf in token order (beam_topk_kernel's result)
    if (tid < K2) {
      const int rest = tid - 1 + (tid - 1 >= p.forced ? 1 : 0);
      p.top_val[(size_t)r * K2 + tid] = tid == 0 ? base : -INFINITY;
      p.top_idx[(size_t)r * K2 + tid] = tid == 0 ? p.forced : (rest < p.V ? rest : -1);
    }
    return;
  }
  float* cand_v = (float*)lt_smem;                 // [K2][G]
  int* cand_i = (int*)(cand_v + (size_t)G * K2);
  __shared__ float rv[4]; __shared__ int ri[4]; __shared__ float red[4];
  const float* prow = p.part + (size_t)r * W * G;  // this row's fields, each G contiguous floats
  // every load of the kernel is issued before the first is consumed (they come from another kernel's stores: ~2 us a round trip;
  // batches of 8 made this kernel five round trips = 15 us long): the log-sum-exp pieces, then the candidates 40 per thread
  float mloc[4], sloc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int g = tid + 256 * u;
    mloc[u] = g < G ? prow[g] : -INFINITY;
    sloc[u] = g < G ? prow[G + g] : 0.f;
  }
  const int n = G * K2;
  for (int e0 = tid; e0 < 2 * n; e0 += 256 * 40) {
    unsigned x[40];
#pragma unroll
    for (int u = 0; u < 40; ++u) {
      const int e = e0 + 256 * u;
      x[u] = e < 2 * n ? ((const unsigned*)prow)[2 * G + e] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 40; ++u) {
      const int e = e0 + 256 * u;
      if (e < 2 * n) ((unsigned*)lt_smem)[e] = x[u];          // values then ids: the same [field][g] order as in HBM
    }
  }
  // ---- log-sum-exp over the workgroups' pieces (fixed order: thread-strided, then wave butterflies, then 4 waves in order)
  float mt = fmaxf(fmaxf(mloc[0], mloc[1]), fmaxf(mloc[2], mloc[3]));
  for (int g = tid + 1024; g < G; g += 256) mt = fmaxf(mt, prow[g]);
  mt = wave_max(mt);
  if (lane == 0) red[wave] = mt;
  __syncthreads();
  const float gm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float st = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (mloc[u] != -INFINITY) st += sloc[u] * __expf(mloc[u] - gm);
  for (int g = tid + 1024; g < G; g += 256) {
    const float me = prow[g];
    if (me != -INFINITY) st += prow[G + g] * __expf(me - gm);
  }
  st = wave_sum(st);
  if (lane == 0) red[wave] = st;
  __syncthreads();
  const float lse = gm + logf(((red[0] + red[1]) + red[2]) + red[3]);
  // ---- K2 picks.  Every workgroup's list is sorted (score desc, token asc), so only list HEADS compete: a thread keeps the heads of
  // its (<= 2) lists in registers, a round is one workgroup-wide arg-best over them and the winner steps to its list's next entry.
  // (Scanning all G x K2 candidates per round — 18 dependent LDS round trips per thread — cost 40 us.)
  float hv[2]; int hi[2], hp[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int g = tid + 256 * u;
    hp[u] = 0;
    hv[u] = g < G ? cand_v[g] : -INFINITY;
    hi[u] = g < G ? cand_i[g] : 0x7fffffff;
  }
  for (int k = 0; k < K2; ++k) {
    float bv = hv[0]; int bi = hi[0];
    if (hi[1] != 0x7fffffff && (bi == 0x7fffffff || hv[1] > bv || (hv[1] == bv && hi[1] < bi))) { bv = hv[1]; bi = hi[1]; }
    if (bi == 0x7fffffff) bv = -INFINITY;
    wave_argbest(bv, bi);
    __syncthreads();
    if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
    __syncthreads();
    bv = rv[0]; bi = ri[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
    if (tid == 0) {
      const bool none = bi == 0x7fffffff;
      p.top_val[(size_t)r * K2 + k] = none ? -INFINITY : bv - lse + base;
      p.top_idx[(size_t)r * K2 + k] = none ? -1 : bi;
    }
    if (bi != 0x7fffffff) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (hi[u] == bi) {                             // this thread's list won: its next entry becomes the head
          const int g = tid + 256 * u;
          ++hp[u];
          hv[u] = hp[u] < K2 ? cand_v[hp[u] * G + g] : -INFINITY;
          hi[u] = hp[u] < K2 ? cand_i[hp[u] * G + g] : 0x7fffffff;
        }
    }
  }
}

// dst[r][:row_bytes] = src[g][:row_bytes], g = idx[r] (period == 0) or (r / period) * period + idx[r % period] (the same beam
// permutation applied to every layer's block of `period` rows); rows are `stride` 16-byte chunks apart, `chunks` of them copied
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                          const int64_t* __restrict__ idx, long rows, long chunks, long stride, long period) {
  const long total = rows * chunks;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / chunks, c = i % chunks;
    const long g = period > 0 ? (r / period) * period + idx[r % period] : idx[r];
    ((u32x4*)dst)[r * stride + c] = ((const u32x4*)src)[g * stride + c];
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// On-device beam-search bookkeeping: BeamSearchScorer.process of transformers 4.18 (the n-best lists,
// EOS handling, done flags) + the NoRepeatNGram ban lists for the NEXT position, one wave per batch item.  With this the
// decode loop has no device->host copy per token; the host reads the state once at the end (finalize).
// State (all per session, device memory):
//   seq_in / seq_out  int32 [R][Lmax]   token histories, ping-pong per position
//   beam_scores       f32   [R]         running sum of log-probs (also the input of the next beam_topk)
//   done              int32 [B];  hyp_cnt int32 [B];  hyp_worst f64 [B]
//   hyp_score f64 [B][nb], hyp_len int32 [B][nb], hyp_seq int32 [B][nb][Lmax]      finished hypotheses, in list order
// Outputs for the next position: next_ids int64 [R] (last tokens), src_idx int64 [R] (beam each new row continues: the
// KV-cache reorder index), bans int32 [R][Lmax] (-1 padded).
struct BeamP {
  const float* top_val; const int32_t* top_idx;      // [R][K2] from beam_topk (scores include the beam score)
  const int32_t* seq_in; int32_t* seq_out;
  float* beam_scores; int32_t* done; int32_t* hyp_cnt; double* hyp_worst; double* hyp_score; int32_t* hyp_len; int32_t* hyp_seq;
  int64_t* next_ids; int64_t* src_idx; int32_t* bans;
  int nb, K2, Lmax, cur_len, V, eos, pad, ngram, early;
  float length_penalty;
};

// One workgroup per batch item, one wave per beam: wave 0 merges the candidates and does BeamSearchScorer.process, then wave j
// writes beam j's history / score / source row and its ban list (one global round trip per beam in parallel instead of 2 nb in a row).
__global__ __launch_bounds__(1024) void beam_step_kernel(BeamP p) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nthr = blockDim.x;
  const int nb = p.nb, K2 = p.K2, L = p.Lmax, cur = p.cur_len;
  __shared__ float cs[64]; __shared__ int cj[64], ct[64];          // the group's top-2nb candidates, in rank order
  __shared__ float nsc[32]; __shared__ int ntok[32], nsrc[32];      // the nb continuing beams
  __shared__ int hyp_from[288]; __shared__ int hyp_slot[288]; __shared__ int n_hyp_copy;   // <= nb adds x (nb - 1 shifts + 1) per position
  __shared__ int is_done, nsel_s;
  if (tid == 0) { n_hyp_copy = 0; is_done = p.done[b]; nsel_s = 0; }
  __syncthreads();
  if (!is_done && wv == 0) {
    // ---- merge the per-beam top-2nb lists: order (score desc, beam * V + token asc), keep 2nb.  Every lane owns up to two
    // candidates; each round is a wave-wide arg-best by shuffles (a serial scan by one lane over global memory cost ~0.8 ms).
    const int ncand = nb * K2;
    float v0 = 0.f, v1 = 0.f; long long k0 = 0, k1 = 0; int t0 = -1, t1 = -1;
    if (lane < ncand) { t0 = p.top_idx[(size_t)b * ncand + lane]; v0 = p.top_val[(size_t)b * ncand + lane]; k0 = (long long)(lane / K2) * p.V + t0; }
    if (lane + 64 < ncand) { t1 = p.top_idx[(size_t)b * ncand + lane + 64]; v1 = p.top_val[(size_t)b * ncand + lane + 64]; k1 = (long long)((lane + 64) / K2) * p.V + t1; }
    bool u0 = t0 < 0, u1 = t1 < 0;
    for (int r = 0; r < K2; ++r) {
      // this lane's best unused candidate
      int bi = -1; float bv = 0.f; long long bk = 0;
      if (!u0) { bi = lane; bv = v0; bk = k0; }
      if (!u1 && (bi < 0 || v1 > bv || (v1 == bv && k1 < bk))) { bi = lane + 64; bv = v1; bk = k1; }
      // wave-wide arg-best through DPP (quad xor 1, xor 2, mirror within 8, mirror within 16) + a scalar pass over the four rows:
      // the 24 ds_bpermute of the xor butterfly per round were most of this kernel's 16 us
      {
        int bh = (int)(bk >> 32), bl = (int)(bk & 0xffffffffLL);
#define VAC_STEP(CTRL_)                                                                                                          \
        {                                                                                                                        \
          const int oi = __builtin_amdgcn_update_dpp(0, bi, CTRL_, 0xf, 0xf, true);                                             \
          const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, bv), CTRL_, 0xf, 0xf, true)); \
          const int oh = __builtin_amdgcn_update_dpp(0, bh, CTRL_, 0xf, 0xf, true), ol = __builtin_amdgcn_update_dpp(0, bl, CTRL_, 0xf, 0xf, true); \
          const long long ok = ((long long)oh << 32) | (unsigned)ol, mk = ((long long)bh << 32) | (unsigned)bl;                  \
          if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && ok < mk))) { bi = oi; bv = ov; bh = oh; bl = ol; }                   \
        }
        VAC_STEP(0xB1) VAC_STEP(0x4E) VAC_STEP(0x141) VAC_STEP(0x140)
#undef VAC_STEP
        int ri_ = __builtin_amdgcn_readlane(bi, 0), rh_ = __builtin_amdgcn_readlane(bh, 0), rl_ = __builtin_amdgcn_readlane(bl, 0);
        float rv_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), 0));
#pragma unroll
        for (int row = 1; row < 4; ++row) {
          const int oi = __builtin_amdgcn_readlane(bi, 16 * row), oh = __builtin_amdgcn_readlane(bh, 16 * row), ol = __builtin_amdgcn_readlane(bl, 16 * row);
          const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv), 16 * row));
          const long long ok = ((long long)oh << 32) | (unsigned)ol, mk = ((long long)rh_ << 32) | (unsigned)rl_;
          if (oi >= 0 && (ri_ < 0 || ov > rv_ || (ov == rv_ && ok < mk))) { ri_ = oi; rv_ = ov; rh_ = oh; rl_ = ol; }
        }
        bi = ri_; bv = rv_; bk = ((long long)rh_ << 32) | (unsigned)rl_;
      }
      if (bi < 0) break;                                    // wave-uniform
      if (bi == lane) u0 = true;
      if (bi == lane + 64) u1 = true;
      if (lane == 0) { cs[r] = bv; cj[r] = bi / K2; ct[r] = (int)(bk - (long long)(bi / K2) * p.V); nsel_s = r + 1; }
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (!is_done) {
      const int nsel = nsel_s;
      // ---- BeamSearchScorer.process: EOS candidates within the first nb ranks finish a hypothesis, the rest continue
      const double lenpow = pow((double)cur, (double)p.length_penalty);
      int cnt = p.hyp_cnt[b]; double worst = p.hyp_worst[b];
      int nch = 0;
      for (int r = 0; r < nsel && nch < nb; ++r) {
        if (ct[r] == p.eos) {
          if (r >= nb) continue;
          const double score = (double)cs[r] / lenpow;
          if (cnt < nb || score > worst) {                                  // _BeamHyps.add
            int slot = cnt;
            if (cnt == nb) {
              // list full: append, then drop the lowest score (lowest index among ties) -> compact in list order
              int lo = 0;
              for (int i = 1; i < nb; ++i) if (p.hyp_score[(size_t)b * nb + i] < p.hyp_score[(size_t)b * nb + lo]) lo = i;
              const bool drop_new = score < p.hyp_score[(size_t)b * nb + lo];     // cannot happen (score > worst), kept for symmetry
              if (!drop_new) {
                for (int i = lo; i + 1 < nb; ++i) {
                  p.hyp_score[(size_t)b * nb + i] = p.hyp_score[(size_t)b * nb + i + 1];
                  p.hyp_len[(size_t)b * nb + i] = p.hyp_len[(size_t)b * nb + i + 1];
                  // sequences of the shifted slots move too: queued as copies slot i+1 -> i (done by all lanes below, in order)
                  hyp_from[n_hyp_copy] = -(i + 1) - 1; hyp_slot[n_hyp_copy] = i; ++n_hyp_copy;
                }
                slot = nb - 1;
              } else {
                slot = -1;
              }
            }
            if (slot >= 0) {
              p.hyp_score[(size_t)b * nb + slot] = score;
              p.hyp_len[(size_t)b * nb + slot] = cur;
              hyp_from[n_hyp_copy] = b * nb + cj[r]; hyp_slot[n_hyp_copy] = slot; ++n_hyp_copy;      // copy seq_in[row] -> hyp_seq[slot]
              if (cnt < nb) ++cnt;
              // worst = lowest score on the list (HF keeps it incrementally; identical value)
              worst = p.hyp_score[(size_t)b * nb];
              for (int i = 1; i < cnt; ++i) worst = fmin(worst, p.hyp_score[(size_t)b * nb + i]);
            }
          }
        } else {
          nsc[nch] = cs[r]; ntok[nch] = ct[r]; nsrc[nch] = b * nb + cj[r]; ++nch;
        }
      }
      p.hyp_cnt[b] = cnt; p.hyp_worst[b] = worst;
      // is_done(best_sum_logprobs = best candidate of the group, cur_len)
      bool dn = false;
      if (cnt >= nb) dn = p.early ? true : (worst >= (double)cs[0] / lenpow);
      if (dn) p.done[b] = 1;
      for (; nch < nb; ++nch) { nsc[nch] = -1e9f; ntok[nch] = p.pad; nsrc[nch] = b * nb; }       // cannot happen with 2nb candidates
    } else {
      for (int j = 0; j < nb; ++j) { nsc[j] = 0.f; ntok[j] = p.pad; nsrc[j] = b * nb; }           // finished batch: padded along
    }
  }
  __syncthreads();
  // ---- finished-hypothesis sequence moves, in the order they were queued (a shift reads a slot a later entry overwrites)
  for (int c = 0; c < n_hyp_copy; ++c) {
    const int from = hyp_from[c], slot = hyp_slot[c];
    const int32_t* src = from < 0 ? p.hyp_seq + ((size_t)b * nb + (-from - 1)) * L : p.seq_in + (size_t)from * L;
    int32_t* dst = p.hyp_seq + ((size_t)b * nb + slot) * L;
    int32_t tmp[8];                                     // L <= 512, >= 64 threads
    int n = 0;
    for (int i = tid; i < L; i += nthr) tmp[n++] = src[i];
    __syncthreads();
    n = 0;
    for (int i = tid; i < L; i += nthr) dst[i] = tmp[n++];
    __syncthreads();
  }
  // ---- the nb continuing beams, one wave each: history + new token (kept in LDS for the ban search), score, source row, last
  // token; then the NoRepeatNGram bans of the next position
  __shared__ int sseq[16][512];
  __shared__ int nban[16];
  if (wv < nb) {
    const int j = wv, row = b * nb + j;
    const int32_t* src = p.seq_in + (size_t)nsrc[j] * L;
    int32_t* dst = p.seq_out + (size_t)row * L;
    if (lane == 0) nban[j] = 0;
    for (int i = lane; i < L; i += 64) {
      const int v = i < cur ? src[i] : (i == cur ? ntok[j] : p.pad);
      dst[i] = v;
      sseq[j][i] = v;
    }
    if (lane == 0) {
      p.beam_scores[row] = nsc[j];
      p.next_ids[row] = ntok[j];
      p.src_idx[row] = nsrc[j];
    }
    if (p.bans) {
      // (a wave's LDS operations execute in order: the history written above is visible to all its lanes here)
      const int n = p.ngram, len = cur + 1;               // histories now hold cur + 1 tokens
      const int* sq = sseq[j];
      int32_t* bn = p.bans + (size_t)row * L;
      if (n > 0 && len + 1 >= n && !is_done) {
        for (int i = lane; i <= len - n; i += 64) {
          bool match = true;
          for (int k = 0; k < n - 1; ++k) match = match && sq[i + k] == sq[len - (n - 1) + k];
          if (match) bn[atomicAdd(&nban[j], 1)] = sq[i + n - 1];
        }
      }
      const int nb_ = nban[j];
      for (int i = nb_ + lane; i < L; i += 64) bn[i] = -1;
    }
  }
}

__global__ void beam_init_kernel(int32_t* seq, float* beam_scores, int32_t* done, int32_t* hyp_cnt, double* hyp_worst, int64_t* next_ids,
                                 int32_t* bans, int R, int nb, int L, int start, int pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < R * L) {
    seq[i] = (i % L) == 0 ? start : pad;
    if (bans) bans[i] = -1;
  }
  if (i < R) { beam_scores[i] = (i % nb) == 0 ? 0.f : -1e9f; next_ids[i] = start; }
  if (i < R / nb) { done[i] = 0; hyp_cnt[i] = 0; hyp_worst[i] = 1e9; }
}
}  // namespace

extern "C" int vacnic_beam_topk(const void* logits, const float* beam_scores, const int32_t* bans, int32_t n_ban, int32_t eos,
                                int32_t suppress_eos, int32_t forced_token, float* top_val, int32_t* top_idx, int64_t R, int64_t V,
                                int64_t ldl, int32_t K, int32_t logits_f32, void* stream) {
  VPLAN_REC(vacnic_beam_topk, logits, beam_scores, bans, n_ban, eos, suppress_eos, forced_token, top_val, top_idx, R, V, ldl, K, logits_f32, stream);
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

static int lt_tiles_per_wg(int64_t V) {
  const int64_t nt = (V + 15) / 16;
  int64_t tpw = (nt + 511) / 512;                  // at most 512 workgroups (two per CU), all of them with the same number of tiles
  if (tpw < 1) tpw = 1;
  static const char* dbg = getenv("VACNIC_LT_TPW");    // measurement aid (tools/bench_lmhead_topk.py): tiles per workgroup, >= the default
  if (dbg && atoi(dbg) >= tpw && atoi(dbg) * 16 <= LT_MAXCOLS) tpw = atoi(dbg);
  return (int)tpw;
}

extern "C" int64_t vacnic_lmhead_topk_workspace(int64_t R, int64_t V, int32_t K2) {
  if (R <= 0 || V <= 0 || K2 <= 0) return 0;
  const int tpw = lt_tiles_per_wg(V);
  const int64_t G = ((V + 15) / 16 + tpw - 1) / tpw;
  return G * R * (2 + 2 * (int64_t)K2);
}

extern "C" int vacnic_lmhead_topk(const vacnic_lmhead_topk_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_lmhead_topk, a, stream);
  VCHECK(a && a->top_val && a->top_idx, VACNIC_BAD_SHAPE, "lmhead_topk: null operand");
  VCHECK(a->R >= 1 && a->R <= 8 && a->V > 0 && a->K2 >= 1 && a->K2 <= 64 && a->K2 <= a->V, VACNIC_UNSUPPORTED, "lmhead_topk: 1 <= R <= 8, 1 <= K2 <= min(64, V)");
  LmTopkP p;
  p.h = (const bf16_t*)a->h; p.emb = (const bf16_t*)a->emb; p.bias = a->bias; p.logits = a->logits; p.part = a->workspace;
  p.bans = a->bans; p.beam_scores = a->beam_scores; p.top_val = a->top_val; p.top_idx = a->top_idx;
  p.R = (int)a->R; p.V = (int)a->V; p.K = (int)a->d; p.ldw = (int)a->ldw; p.ldl = (int)a->ldl; p.n_ban = a->bans ? a->n_ban : 0;
  p.eos = a->eos; p.suppress_eos = a->suppress_eos; p.forced = a->forced_token; p.K2 = a->K2;
  p.tpw = lt_tiles_per_wg(a->V);
  p.G = (int)(((a->V + 15) / 16 + p.tpw - 1) / p.tpw);
  hipStream_t s = (hipStream_t)stream;
  if (a->forced_token >= 0) {
    VCHECK(a->forced_token < a->V, VACNIC_BAD_SHAPE, "lmhead_topk: forced token outside the vocabulary");
    VCHECK(!a->logits, VACNIC_UNSUPPORTED, "lmhead_topk: a forced position computes no logits (pass logits = NULL)");
    hipLaunchKernelGGL(lmhead_merge_kernel, dim3((unsigned)a->R), dim3(256), 0, s, p);
    VLAUNCH_CHECK();
    return VACNIC_OK;
  }
  VCHECK(a->h && a->emb && a->workspace, VACNIC_BAD_SHAPE, "lmhead_topk: null operand");
  VCHECK(a->d >= 8 && a->d <= 1024 && (a->d & 7) == 0 && a->ldw >= a->d && (a->ldw & 7) == 0 && aligned16(a->h) && aligned16(a->emb), VACNIC_UNSUPPORTED,
         "lmhead_topk: d_model <= 1024, a multiple of 8, 16-byte aligned rows");
  VCHECK(p.tpw * 16 <= LT_MAXCOLS, VACNIC_UNSUPPORTED, "lmhead_topk: vocabulary above %d columns", 512 * LT_MAXCOLS);
  VCHECK(!a->logits || a->ldl >= a->V, VACNIC_BAD_SHAPE, "lmhead_topk: ldl < V");
  VCHECK(a->workspace_floats >= vacnic_lmhead_topk_workspace(a->R, a->V, a->K2), VACNIC_BAD_SHAPE, "lmhead_topk: workspace of %ld floats, need %ld",
         (long)a->workspace_floats, (long)vacnic_lmhead_topk_workspace(a->R, a->V, a->K2));
  if (a->d > 256) hipLaunchKernelGGL(lmhead_part_kernel<8>, dim3((unsigned)p.G), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(lmhead_part_kernel<2>, dim3((unsigned)p.G), dim3(256), 0, s, p);
  VLAUNCH_CHECK();
  const size_t lds_merge = (size_t)p.G * a->K2 * 8;
  hipLaunchKernelGGL(lmhead_merge_kernel, dim3((unsigned)a->R), dim3(256), lds_merge, s, p);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_gather_rows(const void* src, void* dst, const int64_t* idx, int64_t rows, int64_t row_bytes, int64_t row_stride_bytes,
                                  int64_t period, void* stream) {
  VPLAN_REC(vacnic_gather_rows, src, dst, idx, rows, row_bytes, row_stride_bytes, period, stream);
  VCHECK(src && dst && idx, VACNIC_BAD_SHAPE, "gather_rows: null operand");
  if (row_stride_bytes == 0) row_stride_bytes = row_bytes;
  VCHECK((row_bytes & 15) == 0 && (row_stride_bytes & 15) == 0 && aligned16(src) && aligned16(dst), VACNIC_MISALIGNED,
         "gather_rows: rows must be 16-byte multiples");
  VCHECK(row_stride_bytes >= row_bytes && period >= 0 && (period == 0 || rows % period == 0), VACNIC_BAD_SHAPE,
         "gather_rows: stride < row bytes, or rows not a multiple of the period");
  if (rows * row_bytes == 0) return VACNIC_OK;
  const long chunks = row_bytes / 16;
  long nb = (rows * chunks + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, idx,
                     (long)rows, chunks, (long)(row_stride_bytes / 16), (long)period);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_beam_init(const vacnic_beam_state* st, int32_t start_token, void* stream) {
  VPLAN_REC_STRUCT(vacnic_beam_init, st, start_token, stream);
  VCHECK(st && st->seq[0] && st->beam_scores && st->done && st->hyp_cnt && st->hyp_worst && st->next_ids, VACNIC_BAD_SHAPE, "beam_init: null state");
  VCHECK(st->B > 0 && st->nb > 0 && st->Lmax > 0, VACNIC_BAD_SHAPE, "beam_init: bad sizes");
  const int R = (int)(st->B * st->nb);
  const int n = R * (int)st->Lmax;
  hipLaunchKernelGGL(beam_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, st->seq[0], st->beam_scores,
                     st->done, st->hyp_cnt, st->hyp_worst, st->next_ids, st->bans, R, (int)st->nb, (int)st->Lmax, start_token, (int)st->pad);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_beam_step(const vacnic_beam_state* st, const float* top_val, const int32_t* top_idx, int32_t K2, int32_t cur_len,
                                void* stream) {
  VPLAN_REC_STRUCT(vacnic_beam_step, st, top_val, top_idx, K2, cur_len, stream);
  VCHECK(st && top_val && top_idx && st->seq[0] && st->seq[1] && st->hyp_score && st->hyp_len && st->hyp_seq && st->src_idx,
         VACNIC_BAD_SHAPE, "beam_step: null operand");
  VCHECK(st->nb >= 1 && st->nb <= 16 && K2 >= 1 && K2 <= 64 && st->nb * K2 <= 128, VACNIC_UNSUPPORTED, "beam_step: nb <= 16, K2 <= 64, nb*K2 <= 128");
  VCHECK(cur_len >= 1 && cur_len < st->Lmax && st->Lmax <= 512, VACNIC_BAD_SHAPE, "beam_step: cur_len %d outside [1, Lmax=%ld) or Lmax > 512",
         (int)cur_len, (long)st->Lmax);
  BeamP p;
  p.top_val = top_val; p.top_idx = top_idx;
  p.seq_in = st->seq[(cur_len - 1) & 1]; p.seq_out = st->seq[cur_len & 1];
  p.beam_scores = st->beam_scores; p.done = st->done; p.hyp_cnt = st->hyp_cnt; p.hyp_worst = st->hyp_worst; p.hyp_score = st->hyp_score;
  p.hyp_len = st->hyp_len; p.hyp_seq = st->hyp_seq; p.next_ids = st->next_ids; p.src_idx = st->src_idx; p.bans = st->bans;
  p.nb = (int)st->nb; p.K2 = K2; p.Lmax = (int)st->Lmax; p.cur_len = cur_len; p.V = (int)st->V; p.eos = (int)st->eos; p.pad = (int)st->pad;
  p.ngram = (int)st->no_repeat_ngram_size; p.early = (int)st->early_stopping; p.length_penalty = st->length_penalty;
  hipLaunchKernelGGL(beam_step_kernel, dim3((unsigned)st->B), dim3((unsigned)(64 * st->nb)), 0, (hipStream_t)stream, p);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
