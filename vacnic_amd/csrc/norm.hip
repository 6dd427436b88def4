// Row-wise LayerNorm family for gfx950 — HBM-bound kernels: one wave64 per row, 16-byte coalesced
// loads, wavefront shuffle reductions, fp32 statistics, Philox dropout regenerated (never stored).
//
//   add_ln      : LayerNorm(residual + dropout(x))           MFULL:651-653,662-664,678-679,705-707,
//                                                            721-723,742-744 and decoder 839-886
//   embed_ln    : dropout(LayerNorm(embed[id]*s + pos[t+2])) MFULL:1243-1249,1254-1260,1553-1562
//   name_embed  : mean_t LayerNorm(embed_ner[id]*s + pos)    TRAIN:112-133 (get_embedding_ner)
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;   // 4 waves

// per-site host seed combined with a per-step counter that lives in device memory, so a captured hipGraph
// draws fresh dropout masks on every replay (the host seed is frozen into the graph)
__device__ __forceinline__ uint64_t mix_seed(uint64_t seed, const uint64_t* seed_dev) {
  return seed_dev ? seed ^ (*seed_dev * 0x9E3779B97F4A7C15ull) : seed;
}

__device__ __forceinline__ void load8(const bf16_t* p, float v[8]) {
  u32x4 r = *(const u32x4*)p;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(r[i] << 16);
    v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void store8(bf16_t* p, const float v[8]) {
  u32x4 r = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
  *(u32x4*)p = r;
}
// keep-scale factors for the 8 elements starting at flat index idx (multiple of 8): ONE Philox4x32-10 block per 8 elements,
// 16 random bits per element (drop probability quantised to 2^-16) — the LayerNorm backward is VALU-bound on the mask
// recompute, and 32-bit draws (two blocks per 8 elements) cost it 10 us per 16384x1024 launch.
// Hidden-state dropout of the LayerNorm family (add_ln / embed_ln, forward and backward): p quantised to 1/256 like the other two
// dropouts; ONE Philox block per lane and PAIR of chunks — chunk i of a lane (columns 8 (lane + 64 i) ..) takes bytes
// 8 (i & 1) .. of the block (row, lane, i >> 1) — so a 1024-wide row costs every lane one Philox call instead of two (the
// generator is ~40 quarter-rate integer multiplies a call: it was a third of these kernels' issue time).  Loops over i are
// unrolled in ascending order, so the block generated for an even i is still in `cache` for i + 1.
struct HDrop { uint32_t w[4]; };
__device__ __forceinline__ uint32_t hdrop_threshold(float p) { const uint32_t t = (uint32_t)(p * 256.f + 0.5f); return t > 255u ? 255u : t; }
__device__ __forceinline__ float hdrop_inv_keep(float p) { return p > 0.f ? 256.f / (256.f - (float)hdrop_threshold(p)) : 1.f; }
__device__ __forceinline__ void drop8(uint64_t seed, int64_t row, int lane, int i, uint32_t thr8, float inv_keep, float m[8], HDrop& cache) {
  if ((i & 1) == 0) {
    const uint64_t blk = (uint64_t)row * 128u + (unsigned)lane + 64u * (unsigned)(i >> 1);
    philox4x32((uint32_t)blk, (uint32_t)(blk >> 32), 0x4C4E4452u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), cache.w);
  }
  actdrop_factors8(cache.w[2 * (i & 1)], cache.w[2 * (i & 1) + 1], thr8, inv_keep, m);
}

// ------------------------------------------------------------------------------------------------
template <int NCH>   // chunks of 8 per lane: D <= NCH*512
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         bf16_t* __restrict__ out, float* __restrict__ mean_o,
                                                         float* __restrict__ rstd_o, int64_t R, int D, float eps,
                                                         float p_drop, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= R) return;
  seed = mix_seed(seed, seed_dev);
  const int nchunk = D >> 3;
  const uint32_t thr = hdrop_threshold(p_drop);
  HDrop hd;
  const float inv_keep = hdrop_inv_keep(p_drop);
  float h[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float xv[8];
      load8(x + row * D + c * 8, xv);
      if (p_drop > 0.f) {
        float m[8];
        drop8(seed, row, lane, i, thr, inv_keep, m, hd);
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] *= m[j];
      }
      if (res) {
        float rv[8];
        load8(res + row * D + c * 8, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] += rv[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { h[i][j] = xv[j]; s += xv[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) h[i][j] = 0.f;
    }
  }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = h[i][j] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float y[8];
      f32x4 g0 = *(const f32x4*)(gamma + c * 8), g1 = *(const f32x4*)(gamma + c * 8 + 4);
      f32x4 b0 = *(const f32x4*)(beta + c * 8), b1 = *(const f32x4*)(beta + c * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y[j] = (h[i][j] - mean) * rstd * g0[j] + b0[j];
        y[4 + j] = (h[i][4 + j] - mean) * rstd * g1[j] + b1[j];
      }
      store8(out + row * D + c * 8, y);
    }
  }
  if (lane == 0) {
    if (mean_o) mean_o[row] = mean;
    if (rstd_o) rstd_o[row] = rstd;
  }
}

// backward: recompute h = residual + dropout(x) from the saved inputs, xhat = (h-mean)*rstd.
//   dxhat = dout*gamma; dh = rstd*(dxhat - mean(dxhat) - xhat*mean(dxhat*xhat))
// Each block walks a strided set of rows and keeps per-column dgamma/dbeta partials in registers,
// then one f32 atomic per column per block.
template <int NCH, int NW>
__global__ __launch_bounds__(64 * NW) void add_ln_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x,
                                                         const bf16_t* __restrict__ res, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                         bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         float* __restrict__ partials,
                                                         int64_t R, int D, float p_drop, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  seed = mix_seed(seed, seed_dev);
  const int nchunk = D >> 3;
  const uint32_t thr = hdrop_threshold(p_drop);
  HDrop hd;
  const float inv_keep = hdrop_inv_keep(p_drop);
  float dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; }

  // A wave walks its rows one after the other, and a row is two dependent wave reductions behind its loads: the NEXT row's
  // x / residual / dout chunks are requested before the current row is reduced, so that a memory round trip is always in
  // flight (the kernel ran at 3.5-4 TB/s with one row per wave in flight).
  const int64_t rstep = (int64_t)gridDim.x * NW;
  u32x4 nx[NCH], nr[NCH], nd[NCH];
  auto fetch = [&](int64_t row) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk && row < R) {
        nx[i] = *(const u32x4*)(x + row * D + c * 8);
        if (res) nr[i] = *(const u32x4*)(res + row * D + c * 8);
        nd[i] = *(const u32x4*)(dout + row * D + c * 8);
      }
    }
  };
  auto unpack = [](u32x4 r, float v[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
  };
  fetch((int64_t)blockIdx.x * NW + wave);
  for (int64_t row = (int64_t)blockIdx.x * NW + wave; row < R; row += rstep) {
    const float mean = mean_i[row], rstd = rstd_i[row];
    float xh[NCH][8], dxh[NCH][8];
    uint32_t keep[NCH];                      // dropout keep bits (one register per chunk instead of 8 scale factors)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float xv[8], dv[8];
        unpack(nx[i], xv);
        if (p_drop > 0.f) {
          float m[8];
          drop8(seed, row, lane, i, thr, inv_keep, m, hd);
          uint32_t kb = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) { xv[j] *= m[j]; kb |= (m[j] != 0.f ? 1u : 0u) << j; }
          keep[i] = kb;
        }
        if (res) {
          float rv[8];
          unpack(nr[i], rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) xv[j] += rv[j];
        }
        unpack(nd[i], dv);
        f32x4 g0 = *(const f32x4*)(gamma + c * 8), g1 = *(const f32x4*)(gamma + c * 8 + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xhat = (xv[j] - mean) * rstd;
          const float g = j < 4 ? g0[j & 3] : g1[j & 3];
          const float d = dv[j] * g;
          xh[i][j] = xhat; dxh[i][j] = d;
          s1 += d; s2 += d * xhat;
          dg[i][j] += dv[j] * xhat; db[i][j] += dv[j];
        }
      }
    }
    fetch(row + rstep);                      // (this row's raw chunks are consumed: the next row travels during the reductions)
    s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float dh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dh[j] = rstd * (dxh[i][j] - s1 - xh[i][j] * s2);
        if (dres) store8(dres + row * D + c * 8, dh);
        if (dx) {
          if (p_drop > 0.f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dh[j] = (keep[i] >> j) & 1u ? dh[j] * inv_keep : 0.f;
          }
          store8(dx + row * D + c * 8, dh);
        }
      }
    }
  }
  // cross-wave reduce of dgamma/dbeta through LDS, then one atomic per column per block
  __shared__ float red[NW][64 * 8 + 1];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = pass == 0 ? dg[i][j] : db[i][j];
      __syncthreads();
      float* dst = pass == 0 ? dgamma : dbeta;
      if (dst) {
        for (int e = threadIdx.x; e < 512; e += 64 * NW) {
          const int c = (e >> 3) + 64 * i;
          if (c < nchunk) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[w][e];
            // two-stage reduction: this block's column sums go to its own row of the scratch [gridDim.x][2][D] (plain
            // 256-byte stores); ln_partial_reduce_kernel folds the rows into dgamma / dbeta.  With ~1000 blocks adding
            // into the same 8 KiB the memory-side atomic units serialise (the kernel ran at 2.6 TB/s, its forward at 4.7).
            if (partials) partials[((size_t)blockIdx.x * 2 + pass) * D + c * 8 + (e & 7)] = t;
            else atomicAdd(dst + c * 8 + (e & 7), t);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ emb,
                                                           const bf16_t* __restrict__ pos, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                           float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                           int64_t R, int T, int D, int64_t V, int pos_offset, float scale,
                                                           float eps, float p_drop, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= R) return;
  seed = mix_seed(seed, seed_dev);
  const int nchunk = D >> 3;
  int64_t id = ids[row];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);   // clamp: a bad id must not fault the GPU
  const int t = (int)(row % T) + pos_offset;
  const uint32_t thr = hdrop_threshold(p_drop);
  HDrop hd;
  const float inv_keep = hdrop_inv_keep(p_drop);
  float h[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float ev[8], pv[8];
      load8(emb + id * D + c * 8, ev);
      load8(pos + (int64_t)t * D + c * 8, pv);
#pragma unroll
      for (int j = 0; j < 8; ++j) { h[i][j] = ev[j] * scale + pv[j]; s += h[i][j]; }
    }
  }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = h[i][j] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float y[8], m[8];
      if (p_drop > 0.f) drop8(seed, row, lane, i, thr, inv_keep, m, hd);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        y[j] = (h[i][j] - mean) * rstd * gamma[c * 8 + j] + beta[c * 8 + j];
        if (p_drop > 0.f) y[j] *= m[j];
      }
      store8(out + row * D + c * 8, y);
    }
  }
  if (lane == 0) {
    if (mean_o) mean_o[row] = mean;
    if (rstd_o) rstd_o[row] = rstd;
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void embed_ln_bwd_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ emb,
                                                           const bf16_t* __restrict__ pos, const bf16_t* __restrict__ dout,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                           const float* __restrict__ rstd_i, float* __restrict__ demb,
                                                           float* __restrict__ dpos, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int64_t R, int T, int D, int64_t V,
                                                           int pos_offset, float scale, int64_t padding_idx, float p_drop,
                                                           uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  __shared__ float stage[ROWS_PER_BLOCK][512];        // per-wave transpose buffer for the scatter-add
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  seed = mix_seed(seed, seed_dev);
  const int nchunk = D >> 3;
  const uint32_t thr = hdrop_threshold(p_drop);
  HDrop hd;
  const float inv_keep = hdrop_inv_keep(p_drop);
  float dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; }

  for (int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + wave; row < R; row += (int64_t)gridDim.x * ROWS_PER_BLOCK) {
    int64_t id = ids[row];
    const bool is_pad = (id == padding_idx);
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const int t = (int)(row % T) + pos_offset;
    const float mean = mean_i[row], rstd = rstd_i[row];
    float xh[NCH][8], dxh[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float ev[8], pv[8], dv[8], m[8];
        load8(emb + id * D + c * 8, ev);
        load8(pos + (int64_t)t * D + c * 8, pv);
        load8(dout + row * D + c * 8, dv);
        if (p_drop > 0.f) {
          drop8(seed, row, lane, i, thr, inv_keep, m, hd);
#pragma unroll
          for (int j = 0; j < 8; ++j) dv[j] *= m[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xhat = (ev[j] * scale + pv[j] - mean) * rstd;
          const float d = dv[j] * gamma[c * 8 + j];
          xh[i][j] = xhat; dxh[i][j] = d;
          s1 += d; s2 += d * xhat;
          dg[i][j] += dv[j] * xhat; db[i][j] += dv[j];
        }
      }
    }
    s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
    // scatter-add into the embedding / position rows.  The lane owns 8 consecutive columns, so a direct atomicAdd per j would
    // touch 64 scattered 4-byte words per wave-instruction; the values take a turn through this wave's 2 KiB of LDS and leave as
    // 256 contiguous bytes per wave-instruction — the shape the memory-side atomic units run at full rate on (881 -> ~250 us
    // for the 16384-row encoder embedding).
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      float* st = stage[wave];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        st[lane * 8 + j] = c < nchunk ? rstd * (dxh[i][j] - s1 - xh[i][j] * s2) : 0.f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // wave-private buffer: no barrier needed
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = 512 * i + e * 64 + lane;
        if (col < D) {
          const float dh = st[e * 64 + lane];
          if (demb && !is_pad) atomicAdd(demb + id * D + col, dh * scale);
          if (dpos) atomicAdd(dpos + (int64_t)t * D + col, dh);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __shared__ float red[ROWS_PER_BLOCK][64 * 8 + 1];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = pass == 0 ? dg[i][j] : db[i][j];
      __syncthreads();
      float* dst = pass == 0 ? dgamma : dbeta;
      if (dst) {
        for (int e = threadIdx.x; e < 512; e += 256) {
          const int c = (e >> 3) + 64 * i;
          if (c < nchunk) atomicAdd(dst + c * 8 + (e & 7), red[0][e] + red[1][e] + red[2][e] + red[3][e]);
        }
      }
    }
  }
}

// mean over Ln tokens of LN(embed*scale + pos): one wave per (b, name)
template <int NCH>
__global__ __launch_bounds__(256) void name_embed_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ emb,
                                                         const bf16_t* __restrict__ pos, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ out,
                                                         int64_t R, int Ln, int D, int64_t V, int pos_offset, float scale,
                                                         float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= R) return;
  const int nchunk = D >> 3;
  float acc[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  for (int t = 0; t < Ln; ++t) {
    int64_t id = ids[row * Ln + t];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    float h[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float ev[8], pv[8];
        load8(emb + id * D + c * 8, ev);
        load8(pos + (int64_t)(t + pos_offset) * D + c * 8, pv);
#pragma unroll
        for (int j = 0; j < 8; ++j) { h[i][j] = ev[j] * scale + pv[j]; s += h[i][j]; }
      }
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { float d = h[i][j] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] += (h[i][j] - mean) * rstd * gamma[c * 8 + j] + beta[c * 8 + j];
      }
    }
  }
  const float inv = 1.f / Ln;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) out[row * D + c * 8 + j] = acc[i][j] * inv;
    }
  }
}

inline int nch_for(int64_t D) { return (int)((D / 8 + 63) / 64); }

}  // namespace

#define DISPATCH_NCH(nch, KERNEL, grid, stream, ...)                                                    \
  switch (nch) {                                                                                        \
    case 1: hipLaunchKernelGGL((KERNEL<1>), grid, dim3(256), 0, stream, __VA_ARGS__); break;            \
    case 2: hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), 0, stream, __VA_ARGS__); break;            \
    case 3: hipLaunchKernelGGL((KERNEL<3>), grid, dim3(256), 0, stream, __VA_ARGS__); break;            \
    default: hipLaunchKernelGGL((KERNEL<4>), grid, dim3(256), 0, stream, __VA_ARGS__); break;           \
  }

static int check_d(int64_t D, const char* who) {
  if (D <= 0 || (D & 7) != 0 || D > 2048) {
    vacnic_set_error("%s: D=%ld must be a multiple of 8 and <= 2048", who, (long)D);
    return VACNIC_BAD_SHAPE;
  }
  return VACNIC_OK;
}

// out[i] = x[i] * keep_i / (1 - p): inverted dropout on a flat bf16 array (8 elements per thread, in place allowed).  The
// activation dropout of the FFN blocks (MFULL:649,660,684,740,874): forward on act(fc1 x), backward on the gradient of the same
// elements with the same (seed, element index) -> the same mask, nothing stored.
namespace {
__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int64_t nchunk, unsigned thr,
                                                      float inv_keep, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  // one thread per 16-element Philox block (two 8-element chunks; the last block of an array with n % 16 == 8 has one)
  const int64_t blk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * blk >= nchunk) return;
  seed = mix_seed(seed, seed_dev);
  uint32_t w[4];
  actdrop_block(seed, (uint64_t)blk, w);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t c = 2 * blk + h;
    if (c < nchunk) {
      float v[8], m[8];
      load8(x + c * 8, v);
      actdrop_factors8(w[2 * h], w[2 * h + 1], thr, inv_keep, m);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= m[j];
      store8(out + c * 8, v);
    }
  }
}
}  // namespace

extern "C" int vacnic_dropout_bf16(const void* x, void* out, int64_t n, float p_drop, uint64_t seed, const uint64_t* seed_dev, void* stream) {
  VPLAN_REC(vacnic_dropout_bf16, x, out, n, p_drop, seed, seed_dev, stream);
  VCHECK(x && out && n >= 0, VACNIC_BAD_SHAPE, "dropout: null operand");
  VCHECK((n & 7) == 0 && aligned16(x) && aligned16(out), VACNIC_MISALIGNED, "dropout: n %% 8 == 0 and 16-byte aligned arrays");
  VCHECK(p_drop > 0.f && p_drop < 1.f, VACNIC_BAD_SHAPE, "dropout: 0 < p < 1");
  if (n == 0) return VACNIC_OK;
  const int64_t nchunk = n >> 3;
  unsigned thr = (unsigned)(p_drop * 256.f + 0.5f);        // p quantised to 1/256, as in the GEMM epilogues that apply the same mask
  if (thr > 255) thr = 255;
  const int64_t nblk = (nchunk + 1) / 2;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out,
                     nchunk, thr, 256.f / (256.f - (float)thr), seed, seed_dev);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_add_ln_fwd(const vacnic_add_ln_fwd_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_add_ln_fwd, a, stream);
  VCHECK(a && a->x && a->gamma && a->beta && a->out, VACNIC_BAD_SHAPE, "add_ln_fwd: null operand");
  if (int e = check_d(a->D, "add_ln_fwd")) return e;
  if (a->R == 0) return VACNIC_OK;
  VCHECK(a->p_drop >= 0.f && a->p_drop < 1.f, VACNIC_BAD_SHAPE, "add_ln_fwd: p_drop out of range");
  VCHECK(aligned16(a->x) && aligned16(a->out) && aligned16(a->gamma) && aligned16(a->beta) &&
         (!a->residual || aligned16(a->residual)), VACNIC_MISALIGNED, "add_ln_fwd: pointers must be 16-byte aligned");
  dim3 grid((unsigned)((a->R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
  DISPATCH_NCH(nch_for(a->D), add_ln_fwd_kernel, grid, (hipStream_t)stream, (const bf16_t*)a->x,
               (const bf16_t*)a->residual, a->gamma, a->beta, (bf16_t*)a->out, a->mean, a->rstd, a->R, (int)a->D,
               a->eps, a->p_drop, a->seed, a->seed_dev);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

// partials [NB][2][D] -> dgamma[D] += sum_b partials[b][0][:], dbeta[D] += sum_b partials[b][1][:].
// grid (2D/64, RS): block (cb, rs) owns 64 columns of the [.., 2D] view and every RS-th group of rows; its 4 waves take
// interleaved rows (256 contiguous bytes per wave-load, 8 loads in flight), meet in LDS, one atomic per column per block.
__global__ __launch_bounds__(256) void ln_partial_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int NB, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;                 // column of the [NB][2D] view
  const int RS = gridDim.y;
  float t = 0.f;
  if (col < 2 * D) {
    int b = blockIdx.y * 4 + wave;
    const int step = RS * 4;
    for (; b + 7 * step < NB; b += 8 * step) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(b + u * step) * 2 * D + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; b < NB; b += step) t += partials[(size_t)b * 2 * D + col];
  }
  __shared__ float red[4][64];
  red[wave][lane] = t;
  __syncthreads();
  if (wave == 0 && col < 2 * D) {
    t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    float* dst = col < D ? dgamma + col : dbeta + (col - D);
    if (col < D ? dgamma != nullptr : dbeta != nullptr) atomicAdd(dst, t);
  }
}

extern "C" int vacnic_add_ln_bwd(const vacnic_add_ln_bwd_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_add_ln_bwd, a, stream);
  VCHECK(a && a->dout && a->x && a->gamma && a->mean && a->rstd, VACNIC_BAD_SHAPE, "add_ln_bwd: null operand");
  if (int e = check_d(a->D, "add_ln_bwd")) return e;
  if (a->R == 0) return VACNIC_OK;
  const hipStream_t st = (hipStream_t)stream;
#define LAUNCH_LN_BWD(NCH_, NW_, NB_)                                                                                  \
  hipLaunchKernelGGL((add_ln_bwd_kernel<NCH_, NW_>), dim3((unsigned)(NB_)), dim3(64 * NW_), 0, st, (const bf16_t*)a->dout, \
                     (const bf16_t*)a->x, (const bf16_t*)a->residual, a->gamma, a->mean, a->rstd, (bf16_t*)a->dresidual, \
                     (bf16_t*)a->dx, a->dgamma, a->dbeta, part, a->R, (int)a->D, a->p_drop, a->seed, a->seed_dev)
  const int nch = nch_for(a->D);
  // 4-wave blocks (<= 1024 of them).  16-wave blocks cut the same-address dgamma/dbeta atomics 4x and are 5 us faster when the
  // kernel runs alone, but a 1024-thread block needs a completely free CU: beside the weight-gradient stream's GEMMs the
  // launch then waits for whole CUs and takes 125 us instead of 48 (measured in the step: 431 vs 428 samples/s).
  {
    int64_t nb = (a->R + 3) / 4;
    if (nb > 1024) nb = 1024;
    // two-stage dgamma/dbeta when the caller lends a scratch of >= nb rows and the atomics would collide (many blocks)
    float* part = (a->partials && a->partial_rows >= nb && nb >= 64 && (a->dgamma || a->dbeta)) ? a->partials : nullptr;
    switch (nch) {
      case 1: LAUNCH_LN_BWD(1, 4, nb); break;
      case 2: LAUNCH_LN_BWD(2, 4, nb); break;
      case 3: LAUNCH_LN_BWD(3, 4, nb); break;
      default: LAUNCH_LN_BWD(4, 4, nb); break;
    }
    if (a->defer_fold && !part) { vacnic_set_error("add_ln_bwd: defer_fold needs a partials scratch of >= %ld rows and R >= 256", (long)nb); return VACNIC_BAD_SHAPE; }
    if (part && !a->defer_fold) {
      VLAUNCH_CHECK();
      hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3((unsigned)((2 * a->D + 63) / 64), 8), dim3(256), 0, st, part, a->dgamma,
                         a->dbeta, (int)nb, (int)a->D);
    }
  }
#undef LAUNCH_LN_BWD
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_ln_partial_fold(const float* partials, float* dgamma, float* dbeta, int64_t rows, int64_t D, void* stream) {
  VPLAN_REC(vacnic_ln_partial_fold, partials, dgamma, dbeta, rows, D, stream);
  VCHECK(partials && rows > 0 && (dgamma || dbeta), VACNIC_BAD_SHAPE, "ln_partial_fold: null operand");
  if (int e = check_d(D, "ln_partial_fold")) return e;
  hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3((unsigned)((2 * D + 63) / 64), 8), dim3(256), 0, (hipStream_t)stream, partials, dgamma,
                     dbeta, (int)rows, (int)D);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_embed_ln_fwd(const vacnic_embed_ln_fwd_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_embed_ln_fwd, a, stream);
  VCHECK(a && a->ids && a->embed && a->pos && a->gamma && a->beta && a->out, VACNIC_BAD_SHAPE, "embed_ln_fwd: null operand");
  if (int e = check_d(a->D, "embed_ln_fwd")) return e;
  const int64_t R = a->B * a->T;
  if (R == 0) return VACNIC_OK;
  dim3 grid((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
  DISPATCH_NCH(nch_for(a->D), embed_ln_fwd_kernel, grid, (hipStream_t)stream, a->ids, (const bf16_t*)a->embed,
               (const bf16_t*)a->pos, a->gamma, a->beta, (bf16_t*)a->out, a->mean, a->rstd, R, (int)a->T, (int)a->D,
               a->V, (int)a->pos_offset, a->embed_scale, a->eps, a->p_drop, a->seed, a->seed_dev);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_embed_ln_bwd(const vacnic_embed_ln_bwd_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_embed_ln_bwd, a, stream);
  VCHECK(a && a->ids && a->embed && a->pos && a->dout && a->gamma && a->mean && a->rstd, VACNIC_BAD_SHAPE,
         "embed_ln_bwd: null operand");
  if (int e = check_d(a->D, "embed_ln_bwd")) return e;
  const int64_t R = a->B * a->T;
  if (R == 0) return VACNIC_OK;
  int64_t nb = (R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  if (nb > 1024) nb = 1024;
  dim3 grid((unsigned)nb);
  DISPATCH_NCH(nch_for(a->D), embed_ln_bwd_kernel, grid, (hipStream_t)stream, a->ids, (const bf16_t*)a->embed,
               (const bf16_t*)a->pos, (const bf16_t*)a->dout, a->gamma, a->mean, a->rstd, a->dembed, a->dpos,
               a->dgamma, a->dbeta, R, (int)a->T, (int)a->D, a->V, (int)a->pos_offset, a->embed_scale,
               a->padding_idx, a->p_drop, a->seed, a->seed_dev);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_name_embed_mean(const vacnic_name_embed_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_name_embed_mean, a, stream);
  VCHECK(a && a->ids && a->embed && a->pos && a->gamma && a->beta && a->out, VACNIC_BAD_SHAPE, "name_embed: null operand");
  if (int e = check_d(a->D, "name_embed")) return e;
  const int64_t R = a->B * a->Nn;
  if (R == 0) return VACNIC_OK;
  dim3 grid((unsigned)((R + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
  DISPATCH_NCH(nch_for(a->D), name_embed_kernel, grid, (hipStream_t)stream, a->ids, (const bf16_t*)a->embed,
               (const bf16_t*)a->pos, a->gamma, a->beta, a->out, R, (int)a->Ln, (int)a->D, a->V, (int)a->pos_offset,
               a->embed_scale, a->eps);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
