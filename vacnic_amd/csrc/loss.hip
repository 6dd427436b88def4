// Loss kernels of the VACNIC step (gfx950): token cross-entropy, CoLaM margin loss, SECLA
// face-name loss.  All are HBM/latency-bound reductions: wavefront shuffles + LDS block reduces.
//   CE     CrossEntropyLoss(ignore_index=pad)                TRAIN:287,816
//   CoLaM  pool -> normalise -> diag cos -> HingeEmbedding   TRAIN:296-307,178-182,820
//   SECLA  BatchSoftmax(face, names)                         TRAIN:631-660
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, red[i]);
  return t;
}

template <bool F32>
__device__ __forceinline__ float ld_logit(const void* row, int j) {
  if (F32) return ((const float*)row)[j];
  return bf2f(((const bf16_t*)row)[j]);
}

// ---- cross entropy --------------------------------------------------------------------------
template <bool F32>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const void* __restrict__ logits, const int64_t* __restrict__ targets,
                                                     float* __restrict__ row_lse, float* __restrict__ row_loss,
                                                     float* __restrict__ loss_sum, float* __restrict__ count,
                                                     int V, long ldl, int64_t ignore_index) {
  __shared__ float red[8];
  const long r = blockIdx.x;
  const char* row = (const char*)logits + r * ldl * (F32 ? 4 : 2);
  // online max/sum: each thread keeps (m, s)
  float m = -INFINITY, s = 0.f;
  if (!F32) {
    const int nchunk = V >> 3;
    for (int c = threadIdx.x; c < nchunk; c += 256) {
      u32x4 d = *(const u32x4*)(row + (long)c * 16);
      float v[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(d[i] << 16); v[2 * i + 1] = __uint_as_float(d[i] & 0xffff0000u); }
      float cm = v[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) cm = fmaxf(cm, v[i]);
      const float nm = fmaxf(m, cm);
      float cs = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) cs += __expf(v[i] - nm);
      s = s * __expf(m - nm) + cs;
      m = nm;
    }
    for (int j = (nchunk << 3) + threadIdx.x; j < V; j += 256) {
      const float v = ld_logit<F32>(row, j);
      const float nm = fmaxf(m, v);
      s = s * __expf(m - nm) + __expf(v - nm);
      m = nm;
    }
  } else {
    // fp32 logits: 16-byte loads, two vectors in flight, one rescale per 8 values (the element-at-a-time loop was a serial
    // chain of two exps per logit: 182 us for the 2048 x 50272 LM-head output)
    const int nvec = (((uintptr_t)row & 15) == 0) ? (V >> 2) : 0;
    for (int c = threadIdx.x; c < nvec; c += 512) {
      const f32x4 a = ((const f32x4*)row)[c];
      const bool two = c + 256 < nvec;
      const f32x4 b = two ? ((const f32x4*)row)[c + 256] : (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      float cm = fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])));
      const float nm = fmaxf(m, cm);
      float cs = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) cs += __expf(a[i] - nm) + __expf(b[i] - nm);      // exp(-inf) = 0 for the absent vector
      s = s * __expf(m - nm) + cs;
      m = nm;
    }
    for (int j = (nvec << 2) + threadIdx.x; j < V; j += 256) {
      const float v = ld_logit<F32>(row, j);
      const float nm = fmaxf(m, v);
      s = s * __expf(m - nm) + __expf(v - nm);
      m = nm;
    }
  }
  const float gm = block_max(m, red);
  const float gs = block_sum(m == -INFINITY ? 0.f : s * __expf(m - gm), red);
  if (threadIdx.x == 0) {
    const float lse = gm + __logf(gs);
    row_lse[r] = lse;
    const int64_t t = targets[r];
    float l = 0.f;
    if (t != ignore_index && t >= 0 && t < V) {
      l = lse - ld_logit<F32>(row, (int)t);
      atomicAdd(loss_sum, l);
      atomicAdd(count, 1.f);
    }
    if (row_loss) row_loss[r] = l;
  }
}

// dlogits = (softmax - onehot) * valid * gscale, gscale = grad_scale * (*grad_out) / (*count)
template <bool F32>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const void* __restrict__ logits, const int64_t* __restrict__ targets,
                                                     const float* __restrict__ row_lse, const float* __restrict__ count,
                                                     const float* __restrict__ grad_out, float grad_scale,
                                                     bf16_t* __restrict__ dlogits, int V, long ldl, long ldd,
                                                     int64_t ignore_index) {
  const long r = blockIdx.x;
  const char* row = (const char*)logits + r * ldl * (F32 ? 4 : 2);
  bf16_t* drow = dlogits + r * ldd;
  const int64_t t = targets[r];
  const bool valid = (t != ignore_index && t >= 0 && t < V);
  const float g = valid ? grad_scale * (grad_out ? *grad_out : 1.f) / *count : 0.f;
  const float lse = row_lse[r];
  for (int j = threadIdx.x; j < (int)ldd; j += 256) {
    float d = 0.f;
    if (j < V && valid) {
      d = __expf(ld_logit<F32>(row, j) - lse);
      if (j == (int)t) d -= 1.f;
      d *= g;
    }
    drow[j] = f2bf(d);
  }
}

// total = ce_sum/count + w_secla*secla + w_colam*colam ; out = {total, txt, secla, colam}
__global__ void combine_losses_kernel(const float* ce_sum, const float* count, const float* secla, const float* colam,
                                      float w_secla, float w_colam, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float txt = *ce_sum / *count;       // 0/0 -> NaN like torch's mean over zero valid targets
    const float s = secla ? *secla : 0.f, c = colam ? *colam : 0.f;
    out[0] = txt + w_secla * s + w_colam * c;
    out[1] = txt; out[2] = s; out[3] = c;
  }
}

// ---- CoLaM ---------------------------------------------------------------------------------
// block b: pooled_x[b][d] = nan_to_num(sum_t mask*h / sum_t mask, nan=1); cos_b = <a,b>/(|a||b|)
__global__ __launch_bounds__(256) void colam_fwd_kernel(const bf16_t* __restrict__ hs, const bf16_t* __restrict__ hg,
                                                        const uint8_t* __restrict__ mask, float* __restrict__ cosv,
                                                        float* __restrict__ ps, float* __restrict__ pg, int T, int D) {
  // one block per sample; a thread owns 8 consecutive columns (16-byte loads) and walks the T tokens 8 at a time with all 16
  // loads of a batch in flight (the kernel is pure latency: 32 blocks, 8 MB; the token-at-a-time 2-byte version took 205 us)
  __shared__ float red[8];
  __shared__ float wt[1024];                       // mask as 0/1 weights (T <= 1024 checked by the host)
  const int b = blockIdx.x;
  float cnt = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) { const float w = mask[(long)b * T + t] ? 1.f : 0.f; wt[t] = w; cnt += w; }
  cnt = block_sum(cnt, red);
  __syncthreads();
  float dot = 0.f, na = 0.f, nb = 0.f;
  const bf16_t* sb = hs + (long)b * T * D;
  const bf16_t* gb = hg + (long)b * T * D;
  for (int c = threadIdx.x; c < (D >> 3); c += 256) {
    float a[8], g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = 0.f; g[j] = 0.f; }
    for (int t0 = 0; t0 < T; t0 += 8) {
      u32x4 ra[8], rg[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = min(t0 + u, T - 1);
        ra[u] = *(const u32x4*)(sb + (long)t * D + c * 8);
        rg[u] = *(const u32x4*)(gb + (long)t * D + c * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float w = t0 + u < T ? wt[t0 + u] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[2 * i] += w * __uint_as_float(ra[u][i] << 16); a[2 * i + 1] += w * __uint_as_float(ra[u][i] & 0xffff0000u);
          g[2 * i] += w * __uint_as_float(rg[u][i] << 16); g[2 * i + 1] += w * __uint_as_float(rg[u][i] & 0xffff0000u);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float av = a[j] / cnt, gv = g[j] / cnt;      // cnt == 0 -> 0/0 = NaN
      if (av != av) av = 1.f;                      // torch.nan_to_num(nan=1.0), TRAIN:181
      if (gv != gv) gv = 1.f;
      ps[(long)b * D + c * 8 + j] = av; pg[(long)b * D + c * 8 + j] = gv;
      dot += av * gv; na += av * av; nb += gv * gv;
    }
  }
  dot = block_sum(dot, red); na = block_sum(na, red); nb = block_sum(nb, red);
  if (threadIdx.x == 0) cosv[b] = dot / (sqrtf(na) * sqrtf(nb));
}
__global__ void colam_loss_kernel(const float* cosv, float* loss, int B, float margin) {
  __shared__ float red[8];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) s += fmaxf(0.f, margin - cosv[i]);
  s = block_sum(s, red);
  if (threadIdx.x == 0) *loss = s / B;
}
// d loss / d hs[b][t][d] = mask ? (g/B) * (-1)[margin-cos>0] * (bn/|a| - cos * a/|a|^2) / cnt : 0
__global__ __launch_bounds__(256) void colam_bwd_kernel(const float* __restrict__ cosv, const float* __restrict__ ps,
                                                        const float* __restrict__ pg, const uint8_t* __restrict__ mask,
                                                        bf16_t* __restrict__ dhs, int B, int T, int D, float margin,
                                                        const float* grad_out, float grad_scale) {
  __shared__ float red[8];
  const int b = blockIdx.x;
  float cnt = 0.f;
  for (int t = 0; t < T; ++t) cnt += mask[(long)b * T + t] ? 1.f : 0.f;
  float na = 0.f, nb = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float a = ps[(long)b * D + d], g = pg[(long)b * D + d];
    na += a * a; nb += g * g;
  }
  na = sqrtf(block_sum(na, red)); nb = sqrtf(block_sum(nb, red));
  const float c = cosv[b];
  float gl = (margin - c > 0.f) ? -grad_scale * (grad_out ? *grad_out : 1.f) / B : 0.f;
  if (cnt == 0.f) gl = 0.f;                   // nan_to_num path carries no gradient
  for (int d = threadIdx.x; d < D; d += 256) {
    const float a = ps[(long)b * D + d], g = pg[(long)b * D + d];
    const float da = gl * (g / (na * nb) - c * a / (na * na)) / cnt;
    const bf16_t v = f2bf(da);
    for (int t = 0; t < T; ++t) dhs[((long)b * T + t) * D + d] = mask[(long)b * T + t] ? v : (bf16_t)0;
  }
}

// ---- SECLA ---------------------------------------------------------------------------------
// sim[i][n][j][f] = <names[i][n], faces[j][f]>; block per (i,n), waves stride over the B*F faces
__global__ __launch_bounds__(256) void secla_sim_kernel(const bf16_t* __restrict__ faces, const float* __restrict__ names,
                                                        float* __restrict__ sim, int BF, int D) {
  // block (name row, group of 16 faces): the name row sits in registers (16 floats per lane per 512 columns), every wave takes
  // 4 faces with their 16-byte loads issued together.  (One block per name row walking all faces with 2-byte loads: 176 us.)
  const int row = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* nm = names + (long)row * D;
  const int nch = D >> 3;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const int jf0 = blockIdx.y * 16 + wave * 4;
  for (int c = lane; c < nch; c += 64) {
    const f32x4 n0 = *(const f32x4*)(nm + c * 8), n1 = *(const f32x4*)(nm + c * 8 + 4);
    u32x4 fr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int jf = min(jf0 + u, BF - 1);
      fr[u] = *(const u32x4*)(faces + (long)jf * D + c * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float nlo = i < 2 ? n0[2 * i] : n1[2 * i - 4], nhi = i < 2 ? n0[2 * i + 1] : n1[2 * i - 3];
        s[u] += nlo * __uint_as_float(fr[u][i] << 16) + nhi * __uint_as_float(fr[u][i] & 0xffff0000u);
      }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float v = wave_sum(s[u]);
    if (lane == 0 && jf0 + u < BF) sim[(long)row * BF + jf0 + u] = v;
  }
}
// logits1[i][j] = sum_n max_f sim[i][n][j][f] / N ; logits2[i][j] = sum_f max_n sim[j][n][i][f] / F
__global__ void secla_logits_kernel(const float* __restrict__ sim, float* __restrict__ l1, float* __restrict__ l2,
                                    int B, int F, int N) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * B) return;
  const int i = idx / B, j = idx % B;
  float a = 0.f;
  for (int n = 0; n < N; ++n) {
    float m = -INFINITY;
    for (int f = 0; f < F; ++f) m = fmaxf(m, sim[(((long)i * N + n) * B + j) * F + f]);
    a += m;
  }
  l1[idx] = a / N;
  float c = 0.f;
  for (int f = 0; f < F; ++f) {
    float m = -INFINITY;
    for (int n = 0; n < N; ++n) m = fmaxf(m, sim[(((long)j * N + n) * B + i) * F + f]);
    c += m;
  }
  l2[idx] = c / F;
}
// loss = CE(l1, arange) + CE(l2, arange), one block
__global__ void secla_loss_kernel(const float* __restrict__ l1, const float* __restrict__ l2, float* loss, int B) {
  __shared__ float red[8];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    for (int w = 0; w < 2; ++w) {
      const float* row = (w ? l2 : l1) + (long)i * B;
      float m = -INFINITY;
      for (int j = 0; j < B; ++j) m = fmaxf(m, row[j]);
      float z = 0.f;
      for (int j = 0; j < B; ++j) z += expf(row[j] - m);
      s += (m + logf(z)) - row[i];
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) *loss = s / B;
}
// wsim[i][n][j][f] = dL/dsim: routed through the arg-max entries (lowest index wins ties)
__global__ void secla_wsim_kernel(const float* __restrict__ sim, const float* __restrict__ l1, const float* __restrict__ l2,
                                  float* __restrict__ wsim, int B, int F, int N, const float* grad_out, float grad_scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * N * B * F;
  if (idx >= total) return;
  const int f = (int)(idx % F); long r = idx / F;
  const int j = (int)(r % B); r /= B;
  const int n = (int)(r % N); const int i = (int)(r / N);
  const float g = grad_scale * (grad_out ? *grad_out : 1.f) / B;
  float w = 0.f;
  {  // term 1: logits1[i][j], max over f for this (i,n,j)
    const float* s = sim + (((long)i * N + n) * B + j) * F;
    int am = 0; float m = s[0];
    for (int ff = 1; ff < F; ++ff) if (s[ff] > m) { m = s[ff]; am = ff; }
    if (am == f) {
      const float* row = l1 + (long)i * B;
      float mx = -INFINITY; for (int jj = 0; jj < B; ++jj) mx = fmaxf(mx, row[jj]);
      float z = 0.f; for (int jj = 0; jj < B; ++jj) z += expf(row[jj] - mx);
      const float sm = expf(row[j] - mx) / z;
      w += g * (sm - (i == j ? 1.f : 0.f)) / N;
    }
  }
  {  // term 2: logits2[j][i] (faces of j, names of i): max over n for this (j, f)
    int am = 0; float m = sim[(((long)i * N + 0) * B + j) * F + f];
    for (int nn = 1; nn < N; ++nn) {
      const float v = sim[(((long)i * N + nn) * B + j) * F + f];
      if (v > m) { m = v; am = nn; }
    }
    if (am == n) {
      const float* row = l2 + (long)j * B;
      float mx = -INFINITY; for (int jj = 0; jj < B; ++jj) mx = fmaxf(mx, row[jj]);
      float z = 0.f; for (int jj = 0; jj < B; ++jj) z += expf(row[jj] - mx);
      const float sm = expf(row[i] - mx) / z;
      w += g * (sm - (i == j ? 1.f : 0.f)) / F;
    }
  }
  wsim[idx] = w;
}
// dfaces[j][f][:] = sum_{i,n} wsim[i][n][j][f] * names[i][n][:]
__global__ __launch_bounds__(256) void secla_dfaces_kernel(const float* __restrict__ wsim, const float* __restrict__ names,
                                                           bf16_t* __restrict__ dfaces, int BN, int BF, int D) {
  // dfaces[jf] = sum_r wsim[r][jf] * names[r]; the routing weights are sparse (arg-max entries only), so the block first
  // compacts the non-zero rows of its column into LDS and then streams just those name rows with 16-byte loads.
  __shared__ int ridx[1024];
  __shared__ float rw[1024];
  __shared__ int nnz;
  const int jf = blockIdx.x;
  if (threadIdx.x < 64) {                          // ordered compaction by one wave (rows stay in increasing order: the sum
    int count = 0;                                 // below adds in the same order as a plain walk -> deterministic)
    const int lane = threadIdx.x;
    for (int base = 0; base < BN; base += 64) {
      const int r = base + lane;
      const float w = r < BN ? wsim[(long)r * BF + jf] : 0.f;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(w != 0.f);
      const int pos = count + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (w != 0.f && pos < 1024) { ridx[pos] = r; rw[pos] = w; }
      count += __builtin_popcountll(m);
    }
    if (lane == 0) nnz = count;
  }
  __syncthreads();
  const int n = min(nnz, 1024);
  for (int c = threadIdx.x; c < (D >> 2); c += 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (nnz <= 1024) {
      for (int k0 = 0; k0 < n; k0 += 4) {
        f32x4 v[4]; float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = min(k0 + u, n - 1);
          v[u] = *(const f32x4*)(names + (long)ridx[k] * D + c * 4);
          w[u] = k0 + u < n ? rw[k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += w[u] * v[u][e];
      }
    } else {                                        // dense column (cannot happen with arg-max routing): plain walk
      for (int r = 0; r < BN; ++r) {
        const float w = wsim[(long)r * BF + jf];
        if (w != 0.f) {
          const f32x4 v = *(const f32x4*)(names + (long)r * D + c * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += w * v[e];
        }
      }
    }
    *(u32x2*)(dfaces + (long)jf * D + c * 4) = (u32x2){pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3])};
  }
}

}  // namespace

extern "C" int vacnic_ce_fwd(const vacnic_ce_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_ce_fwd, a, stream);
  VCHECK(a && a->logits && a->targets && a->row_lse && a->loss_sum && a->count, VACNIC_BAD_SHAPE, "ce_fwd: null operand");
  VCHECK(a->V > 0 && a->ldl >= a->V, VACNIC_BAD_SHAPE, "ce_fwd: bad V/ldl");
  if (a->R == 0) return VACNIC_OK;
  VCHECK(a->logits_f32 || ((a->ldl & 7) == 0 && aligned16(a->logits)), VACNIC_MISALIGNED, "ce_fwd: bf16 logits rows must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (a->logits_f32)
    hipLaunchKernelGGL(ce_fwd_kernel<true>, dim3((unsigned)a->R), dim3(256), 0, s, a->logits, a->targets, a->row_lse,
                       a->row_loss, a->loss_sum, a->count, (int)a->V, (long)a->ldl, a->ignore_index);
  else
    hipLaunchKernelGGL(ce_fwd_kernel<false>, dim3((unsigned)a->R), dim3(256), 0, s, a->logits, a->targets, a->row_lse,
                       a->row_loss, a->loss_sum, a->count, (int)a->V, (long)a->ldl, a->ignore_index);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_ce_bwd(const vacnic_ce_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_ce_bwd, a, stream);
  VCHECK(a && a->logits && a->targets && a->row_lse && a->count && a->dlogits, VACNIC_BAD_SHAPE, "ce_bwd: null operand");
  VCHECK(a->V > 0 && a->ldl >= a->V && a->ldd >= a->V, VACNIC_BAD_SHAPE, "ce_bwd: bad V/ldl/ldd");
  if (a->R == 0) return VACNIC_OK;
  hipStream_t s = (hipStream_t)stream;
  if (a->logits_f32)
    hipLaunchKernelGGL(ce_bwd_kernel<true>, dim3((unsigned)a->R), dim3(256), 0, s, a->logits, a->targets, a->row_lse,
                       a->count, a->grad_out, a->grad_scale, (bf16_t*)a->dlogits, (int)a->V, (long)a->ldl, (long)a->ldd,
                       a->ignore_index);
  else
    hipLaunchKernelGGL(ce_bwd_kernel<false>, dim3((unsigned)a->R), dim3(256), 0, s, a->logits, a->targets, a->row_lse,
                       a->count, a->grad_out, a->grad_scale, (bf16_t*)a->dlogits, (int)a->V, (long)a->ldl, (long)a->ldd,
                       a->ignore_index);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_combine_losses(const float* ce_sum, const float* count, const float* secla, const float* colam,
                                     float w_secla, float w_colam, float* out4, void* stream) {
  VPLAN_REC(vacnic_combine_losses, ce_sum, count, secla, colam, w_secla, w_colam, out4, stream);
  VCHECK(ce_sum && count && out4, VACNIC_BAD_SHAPE, "combine_losses: null operand");
  hipLaunchKernelGGL(combine_losses_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ce_sum, count, secla, colam,
                     w_secla, w_colam, out4);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_colam_fwd(const vacnic_colam_fwd_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_colam_fwd, a, stream);
  VCHECK(a && a->hs && a->hg && a->mask && a->loss && a->cos && a->pooled_s && a->pooled_g, VACNIC_BAD_SHAPE, "colam_fwd: null operand");
  VCHECK(a->B > 0 && a->T > 0 && a->D > 0, VACNIC_BAD_SHAPE, "colam_fwd: empty");
  VCHECK(a->T <= 1024 && (a->D & 7) == 0 && aligned16(a->hs) && aligned16(a->hg), VACNIC_BAD_SHAPE, "colam_fwd: needs T <= 1024, D %% 8 == 0, 16-byte aligned states");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colam_fwd_kernel, dim3((unsigned)a->B), dim3(256), 0, s, (const bf16_t*)a->hs, (const bf16_t*)a->hg,
                     a->mask, a->cos, a->pooled_s, a->pooled_g, (int)a->T, (int)a->D);
  VLAUNCH_CHECK();
  hipLaunchKernelGGL(colam_loss_kernel, dim3(1), dim3(256), 0, s, a->cos, a->loss, (int)a->B, a->margin);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_colam_bwd(const vacnic_colam_bwd_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_colam_bwd, a, stream);
  VCHECK(a && a->cos && a->pooled_s && a->pooled_g && a->mask && a->dhs, VACNIC_BAD_SHAPE, "colam_bwd: null operand");
  hipLaunchKernelGGL(colam_bwd_kernel, dim3((unsigned)a->B), dim3(256), 0, (hipStream_t)stream, a->cos, a->pooled_s,
                     a->pooled_g, a->mask, (bf16_t*)a->dhs, (int)a->B, (int)a->T, (int)a->D, a->margin, a->grad_out,
                     a->grad_scale);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_secla_fwd(const vacnic_secla_fwd_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_secla_fwd, a, stream);
  VCHECK(a && a->faces && a->names && a->sim && a->logits1 && a->logits2 && a->loss, VACNIC_BAD_SHAPE, "secla_fwd: null operand");
  VCHECK(a->B > 0 && a->F > 0 && a->N > 0 && a->D > 0, VACNIC_BAD_SHAPE, "secla_fwd: empty");
  VCHECK((a->D & 7) == 0 && aligned16(a->faces) && aligned16(a->names), VACNIC_BAD_SHAPE, "secla_fwd: needs D %% 8 == 0 and 16-byte aligned faces / names");
  hipStream_t s = (hipStream_t)stream;
  const int B = (int)a->B, F = (int)a->F, N = (int)a->N, D = (int)a->D;
  hipLaunchKernelGGL(secla_sim_kernel, dim3(B * N, (B * F + 15) / 16), dim3(256), 0, s, (const bf16_t*)a->faces, a->names, a->sim, B * F, D);
  VLAUNCH_CHECK();
  hipLaunchKernelGGL(secla_logits_kernel, dim3((B * B + 255) / 256), dim3(256), 0, s, a->sim, a->logits1, a->logits2, B, F, N);
  VLAUNCH_CHECK();
  hipLaunchKernelGGL(secla_loss_kernel, dim3(1), dim3(256), 0, s, a->logits1, a->logits2, a->loss, B);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_secla_bwd(const vacnic_secla_bwd_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_secla_bwd, a, stream);
  VCHECK(a && a->names && a->sim && a->logits1 && a->logits2 && a->dfaces && a->wsim, VACNIC_BAD_SHAPE, "secla_bwd: null operand");
  hipStream_t s = (hipStream_t)stream;
  const int B = (int)a->B, F = (int)a->F, N = (int)a->N, D = (int)a->D;
  const long total = (long)B * N * B * F;
  hipLaunchKernelGGL(secla_wsim_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a->sim, a->logits1,
                     a->logits2, a->wsim, B, F, N, a->grad_out, a->grad_scale);
  VLAUNCH_CHECK();
  hipLaunchKernelGGL(secla_dfaces_kernel, dim3(B * F), dim3(256), 0, s, a->wsim, a->names, (bf16_t*)a->dfaces, B * N, B * F, D);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
