// decstep.hip — the single-token decoder step of caption generation as ONE persistent kernel.
//
// What it replaces: BartDecoderLayer.forward with past_key_value (MFULL:793-890: self-attention over the KV cache, cross-attention
// over the encoder K/V projected once, FFN, three post-LayerNorms; eval mode) for all decoder layers of one position — 8 dependent
// launches per layer in the kernel-per-op path (gemv_ln / gemm_skinny / attn_decode), each 6-12 us for 2-8 MB of weights.
//
// Design.  G = one workgroup per CU (256 threads) walks the 8 phases of every layer; phases are separated by a grid barrier
// (two-level arrive over 16 counters + 16 release flags on their own cache lines: 1.6 us at G = 256, tools/gridbar_probe.hip)
// instead of a kernel boundary (~5 us + cold caches), and the weights of the NEXT projection are pulled into registers between
// the arrive and the wait of a barrier — they do not depend on the activations, so the HBM round trip of a phase is hidden
// behind the barrier it follows.
//   P1  x = LN(o + h)      -> k|v|q = x Wkvq^T + b        (written in place into the KV cache row of position t)
//   P2  self-attention over cache[0..t]                     one (row, head) pair per workgroup
//   P3  o = ctx Wo^T + b
//   P4  x = LN(o + h) -> h -> q = x Wq^T + b
//   P5  cross-attention over the encoder K/V (key mask)     one (row, head) pair per workgroup, 4 waves split the keys
//   P6  o = ctx Wo^T + b
//   P7  x = LN(o + h) -> h -> f = gelu(x W1^T + b)
//   P8  o = f W2^T + b                                      (the last layer's (o, h) pair is normalised by the LM-head kernel)
// Activations that cross a phase (k|v|q, ctx, o, h, q, f: <= 8 rows) live in HBM and are exchanged with sc1 buffer loads /
// stores (agent-coherent: write-through, miss-always in the XCD's L2; a store is complete when vmcnt says so) — release /
// acquire fences (buffer_wbl2 / buffer_inv) cost 20-35 us per phase on this part, sc1 traffic costs nothing measurable.
// Weights, the cross-attention K/V and LayerNorm parameters are read-only for the launch and use ordinary cached loads.
//
// Arithmetic: the same per-lane accumulation order, wave-reduction tree and bf16 rounding points as vacnic_gemv_ln_bf16,
// gemm_skinny_kernel and attn_decode_kernel: the logits of a position match the kernel-per-op path bit for bit except where the
// two compilations contract an fma differently (one bf16 ulp on a rare activation; -ffp-contract=fast leaves that to the
// optimiser) — tests/test_model_gpu.py::test_decoder_step_kernel_matches_per_op_path.
//
// Safety: a barrier wait that exceeds ~2 s (100 MHz wall clock) sets the error word and every workgroup leaves; the last
// workgroup out clears the barrier state, so a launch always drains and the next one starts clean.
#include "common.h"

namespace {

constexpr int NTHR = 256, NWAVE = 4, MR = 8;
constexpr int NGRP = 16, NFLAG = 16;
constexpr int COH = 16;                       // aux bits of the buffer intrinsics: sc1
// barrier state (uint32 words, one 128-byte line each): grp[16] | top | flag[16] | exit | err
constexpr int BAR_TOP = 32 * NGRP, BAR_FLAG = 32 * (NGRP + 1), BAR_EXIT = 32 * (NGRP + 1 + NFLAG), BAR_ERR = BAR_EXIT + 32;
#define AGENT __HIP_MEMORY_SCOPE_AGENT

struct DecP {
  const vacnic_decoder_layer* layers;
  bf16_t* cache; const bf16_t* h0;
  bf16_t *hb0, *hb1, *o, *ctx, *q, *f;
  const uint8_t* enc_mask;
  unsigned* bar;
  int L, R, d, H, F, S, t, Tstride;           // Tstride = (Tmax + 1) * 2d elements between two rows of one layer's cache
  unsigned cache_bytes, xs_bytes;
  float eps, scale;
  unsigned long long* trace; int trace_wg;    // profiling aid: 100 MHz time stamps of one workgroup, [phase][8]
  char* slots;                                // tagged-slot exchange buffers (slot variant)
};

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t mkrs(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000); }
__device__ __forceinline__ u32x4 cld(rsrc_t rs, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, COH); }
__device__ __forceinline__ void cst(u32x4 v, rsrc_t rs, unsigned off) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, COH); }
__device__ __forceinline__ void cst16(float v, rsrc_t rs, unsigned off) { __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(v), rs, off, 0, COH); }

__device__ __forceinline__ void unpack8(u32x4 r, float v[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
}

// ---- grid barrier ------------------------------------------------------------------------------------------------
struct Bar { unsigned* base; int G; unsigned ph; bool last; };

// every thread has waited for its own stores (s_waitcnt vmcnt(0)) before this
__device__ __forceinline__ void bar_arrive(Bar& b) {
  __syncthreads();
  b.ph += 1;
  if (threadIdx.x == 0) {
    const int g = blockIdx.x % NGRP;
    const unsigned gsize = (unsigned)((b.G - g + NGRP - 1) / NGRP);
    const unsigned ngrp = (unsigned)(b.G < NGRP ? b.G : NGRP);
    bool last = __hip_atomic_fetch_add(b.base + 32 * g, 1u, __ATOMIC_RELAXED, AGENT) + 1 == b.ph * gsize;
    if (last) last = __hip_atomic_fetch_add(b.base + BAR_TOP, 1u, __ATOMIC_RELAXED, AGENT) + 1 == b.ph * ngrp;
    if (last) {
#pragma unroll
      for (int f = 0; f < NFLAG; ++f) __hip_atomic_store(b.base + BAR_FLAG + 32 * f, b.ph, __ATOMIC_RELAXED, AGENT);
    }
    b.last = last;
  }
}
__device__ __forceinline__ bool bar_wait(Bar& b, int* s_bad) {
  if (threadIdx.x == 0) {
    if (!b.last) {
      unsigned* fl = b.base + BAR_FLAG + 32 * (blockIdx.x % NFLAG);
      const long long t0 = wall_clock64();
      unsigned polls = 0;
      while (__hip_atomic_load(fl, __ATOMIC_RELAXED, AGENT) < b.ph) {
        __builtin_amdgcn_s_sleep(1);
        if ((++polls & 1023u) == 0 &&
            (wall_clock64() - t0 > 200000000ll || __hip_atomic_load(b.base + BAR_ERR, __ATOMIC_RELAXED, AGENT) != 0)) {
          __hip_atomic_store(b.base + BAR_ERR, 1u, __ATOMIC_RELAXED, AGENT);
          *s_bad = 1;
          break;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's prefetch DMA has landed -> visible to the whole workgroup below
  __syncthreads();
  return *s_bad == 0;
}

// ---- weight prefetch --------------------------------------------------------------------------------------------------
// The next projection's weights are ISSUED between the arrive and the wait of a barrier and must stay in flight across it.
// They go HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: no destination VGPRs, so nothing the register allocator could
// move or spill while the data is still on its way, and an intrinsic with side effects is not sunk behind the wait the way
// plain C++ loads were).  Every wave owns private 1-KiB slots (one per 64-lane x 16-byte piece: lane-linear, conflict-free
// to read back) and only ever reads its own, so `s_waitcnt vmcnt(0)` by that wave is all the synchronisation needed.
// Columns >= N and chunks >= K / 8 point outside the descriptor and are zero-filled.
constexpr unsigned OOB = 0x7ffffff0u;
constexpr int WBIG = 8 * 1024, WSMALL = 2 * 1024;       // bytes per wave: 4 columns x 2 chunks / 1 column x 2 chunks
__device__ __forceinline__ void wdma(rsrc_t wrs, char* slot, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, LDS_PTR(slot), 16, (int)voff, 0, 0, 0);
}
__device__ __forceinline__ void w_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ u32x4 w_get(const char* wl, int slot, int lane) { return *(const u32x4*)(wl + slot * 1024 + lane * 16); }

// C columns n0 .. n0+C-1 of W [N][K] for one wave, K <= 1024: chunks lane, lane + 64 -> slots c * 2 + i
template <int C>
__device__ __forceinline__ void w_issue(char* wl, rsrc_t wrs, int N, int K, int n0, int lane) {
  const int nchunk = K >> 3;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = lane + 64 * i, n = n0 + c;
      wdma(wrs, wl + (c * 2 + i) * 1024, (ch < nchunk && n < N) ? (unsigned)(n * K + ch * 8) * 2u : OOB);
    }
}
// 4 columns per workgroup, K <= 4096 split over the 4 waves: chunks lane + 64 * wave + 256 * i
__device__ __forceinline__ void w2_issue(char* wl, rsrc_t wrs, int N, int K, int n0, int wave, int lane) {
  const int nchunk = K >> 3;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = lane + 64 * wave + 256 * i, n = n0 + c;
      wdma(wrs, wl + (c * 2 + i) * 1024, (ch < nchunk && n < N) ? (unsigned)(n * K + ch * 8) * 2u : OOB);
    }
}

// LayerNorm parameters (fp32 gamma | beta, K <= 1024 each) into the workgroup's shared slot: wave w moves piece w of both
__device__ __forceinline__ void ln_issue(float* lnp, const float* gamma, const float* beta, int K, int wave, int lane) {
  const rsrc_t rg = mkrs(gamma, (unsigned)K * 4u), rb = mkrs(beta, (unsigned)K * 4u);
  const int f0 = wave * 256 + lane * 4;
  const unsigned voff = f0 < K ? (unsigned)f0 * 4u : OOB;
  wdma(rg, (char*)lnp + wave * 1024, voff);
  wdma(rb, (char*)(lnp + 1024) + wave * 1024, voff);
}
// the bias values of this workgroup's first column tile (<= 16 columns) into the shared slot, by wave 0 (one 4-byte DMA piece)
__device__ __forceinline__ void bias_issue(float* biasl, const float* bias, int N, int n_first, int ncols, int wave, int lane) {
  if (wave == 0) {
    const rsrc_t rb = mkrs(bias, bias ? (unsigned)N * 4u : 0u);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(biasl), 4, (int)((lane < ncols && n_first + lane < N) ? (unsigned)(n_first + lane) * 4u : OOB), 0, 0, 0);
  }
}

struct OutD { rsrc_t rs; unsigned base, rstride; };     // byte offset of (row 0, column 0) and row stride in bytes

// reduce-scatter butterflies over the 64 lanes (xor offsets 32, 16, 8, 4, 2, 1 — the tree of gemm_skinny_kernel)
#define VAC_BFLY(OFF, HALF)                                                      \
  {                                                                              \
    const bool up = (lane & OFF) != 0;                                           \
    _Pragma("unroll") for (int i = 0; i < HALF; ++i) {                           \
      const float send = up ? acc[i] : acc[HALF + i];                            \
      const float keep = up ? acc[HALF + i] : acc[i];                            \
      acc[i] = keep + __shfl_xor(send, OFF, 64);                                 \
    }                                                                            \
  }
__device__ __forceinline__ float bfly32(float (&acc)[32], int lane, int& idx, bool& owner) {
  VAC_BFLY(32, 16) VAC_BFLY(16, 8) VAC_BFLY(8, 4) VAC_BFLY(4, 2) VAC_BFLY(2, 1)
  const float v = acc[0] + __shfl_xor(acc[0], 1, 64);
  idx = (((lane >> 5) & 1) << 4) | (((lane >> 4) & 1) << 3) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 1) | ((lane >> 1) & 1);
  owner = (lane & 1) == 0;
  return v;
}
__device__ __forceinline__ float bfly8(float (&acc)[8], int lane, int& idx, bool& owner) {
  VAC_BFLY(32, 4) VAC_BFLY(16, 2) VAC_BFLY(8, 1)
  float v = acc[0];
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  idx = (((lane >> 5) & 1) << 2) | (((lane >> 4) & 1) << 1) | ((lane >> 3) & 1);
  owner = (lane & 7) == 0;
  return v;
}
#undef VAC_BFLY

// one wave: the dot products of C weight rows (this wave's prefetch slots) with all R rows of x (bf16, LDS, [R][K]); after the
// reduce-scatter butterfly `owner` lanes hold the sum of (row idx / C, column idx % C)
template <int C>
__device__ __forceinline__ float gemv_reduce(const char* wl, const bf16_t* xs, int R, int K, int lane, int& idx, bool& owner) {
  constexpr int NV = MR * C;
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.f;
  const int nchunk = K >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunk) {
      float wv[C][8];
#pragma unroll
      for (int c = 0; c < C; ++c) unpack8(w_get(wl, c * 2 + i, lane), wv[c]);
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        if (m < R) {
          float xv[8];
          unpack8(*(const u32x4*)(xs + (size_t)m * K + ch * 8), xv);
#pragma unroll
          for (int c = 0; c < C; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[m * C + c] += xv[j] * wv[c][j];
        }
      }
    }
  }
  if constexpr (C == 4) return bfly32(acc, lane, idx, owner);
  else return bfly8(acc, lane, idx, owner);
}

// one wave: C output columns n0 .. n0+C-1 for all R rows
template <int C>
__device__ __forceinline__ void gemv_tile(const char* wl, const bf16_t* xs, int R, int N, int K, int n0, const float* bias, const float* biasl,
                                          int act, const OutD& o, int lane) {
  int idx; bool owner;
  float v = gemv_reduce<C>(wl, xs, R, K, lane, idx, owner);
  if (owner) {
    const int m = idx / C, c = idx % C, n = n0 + c;
    if (m < R && n < N) {
      if (biasl) v += biasl[c];
      else if (bias) v += bias[n];
      v = act_fwd(act, v);
      cst16(v, o.rs, o.base + (unsigned)m * o.rstride + (unsigned)n * 2u);
    }
  }
}

template <int C>
__device__ __forceinline__ void gemv_phase(char* wl, rsrc_t wrs, const float* bias, const float* biasl, int N, int K, int act, const OutD& o,
                                           const bf16_t* xs, int R, int G, int wg, int wave, int lane) {
  const int ntile = (N + NWAVE * C - 1) / (NWAVE * C);
  for (int T = wg; T < ntile; T += G) {
    const int n0 = (T * NWAVE + wave) * C;
    if (T != wg) { w_issue<C>(wl, wrs, N, K, n0, lane); w_wait(); }         // tiles beyond the prefetched one (N > 4 C G)
    gemv_tile<C>(wl, xs, R, N, K, n0, bias, T == wg ? biasl + wave * C : nullptr, act, o, lane);
  }
}

// long reduction (fc2): a workgroup owns 4 columns, its 4 waves split K and meet in LDS (gemm_skinny_kernel<8, 4, 4>)
__device__ __forceinline__ void gemv2_phase(char* wl, rsrc_t wrs, const float* bias, const float* biasl, int N, int K, const OutD& o, const bf16_t* xs,
                                            int R, int G, int wg, int wave, int lane, float (*part)[32]) {
  const int ntile = (N + 3) / 4;
  const int nchunk = K >> 3;
  for (int T = wg; T < ntile; T += G) {
    const int n0 = T * 4;
    if (T != wg) { w2_issue(wl, wrs, N, K, n0, wave, lane); w_wait(); }
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = lane + 64 * wave + 256 * i;
      if (ch < nchunk) {
        float wv[4][8];
#pragma unroll
        for (int c = 0; c < 4; ++c) unpack8(w_get(wl, c * 2 + i, lane), wv[c]);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          if (m < R) {
            float xv[8];
            unpack8(*(const u32x4*)(xs + (size_t)m * K + ch * 8), xv);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[m * 4 + c] += xv[j] * wv[c][j];
          }
        }
      }
    }
    int idx; bool owner;
    float v = bfly32(acc, lane, idx, owner);
    if (owner) part[wave][idx] = v;
    __syncthreads();
    if (wave == 0 && owner) {
      v = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) v += part[w][idx];
      const int m = idx / 4, c = idx % 4, n = n0 + c;
      if (m < R && n < N) {
        if (T == wg) v += biasl[c];
        else if (bias) v += bias[n];
        cst16(v, o.rs, o.base + (unsigned)m * o.rstride + (unsigned)n * 2u);
      }
    }
    __syncthreads();
  }
}

// ---- x -> LDS ------------------------------------------------------------------------------------------------------
// rows of a phase-crossing activation (bf16 [R][K], contiguous) into LDS
__device__ __forceinline__ void stage_plain(rsrc_t src, bf16_t* xs, int R, int K, int tid) {
  const int n = R * (K >> 3);
  for (int c0 = tid; c0 < n; c0 += 4 * NTHR) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u * NTHR;
      v[u] = c < n ? cld(src, (unsigned)c * 16u) : (u32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u * NTHR;
      if (c < n) *(u32x4*)(xs + (size_t)c * 8) = v[u];
    }
  }
}

// x = LayerNorm(o + h) (add_ln_fwd_kernel's arithmetic: one wave per row, chunks lane / lane + 64, fp32 statistics), rounded
// to bf16 into LDS; workgroup 0 also writes the rows to hnew — the next block's residual.
__device__ __forceinline__ void stage_ln(rsrc_t osrc, rsrc_t hsrc, const float* lnp, rsrc_t hnew, bool write_h,
                                         bf16_t* xs, int R, int K, float eps, int wave, int lane) {
  const int nchunk = K >> 3;
  u32x4 xraw[2][2], rraw[2][2];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = wave + NWAVE * rr, ch = lane + 64 * i;
      xraw[rr][i] = (u32x4){0, 0, 0, 0}; rraw[rr][i] = (u32x4){0, 0, 0, 0};
      if (m < R && ch < nchunk) {
        xraw[rr][i] = cld(osrc, (unsigned)(m * K + ch * 8) * 2u);
        rraw[rr][i] = cld(hsrc, (unsigned)(m * K + ch * 8) * 2u);
      }
    }
  float gam[2][8], bet[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = lane + 64 * i;
#pragma unroll
    for (int j = 0; j < 8; ++j) { gam[i][j] = 0.f; bet[i][j] = 0.f; }
    if (ch < nchunk) {
      const f32x4 g0 = *(const f32x4*)(lnp + ch * 8), g1 = *(const f32x4*)(lnp + ch * 8 + 4);
      const f32x4 b0 = *(const f32x4*)(lnp + 1024 + ch * 8), b1 = *(const f32x4*)(lnp + 1024 + ch * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { gam[i][j] = g0[j]; gam[i][4 + j] = g1[j]; bet[i][j] = b0[j]; bet[i][4 + j] = b1[j]; }
    }
  }
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int m = wave + NWAVE * rr;
    if (m >= R) continue;
    float h[2][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (lane + 64 * i < nchunk) {
        float xv[8], rv[8];
        unpack8(xraw[rr][i], xv);
        unpack8(rraw[rr][i], rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) { xv[j] += rv[j]; h[i][j] = xv[j]; s += xv[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) h[i][j] = 0.f;
      }
    }
    const float mean = wave_sum(s) / K;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (lane + 64 * i < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float dd = h[i][j] - mean; q += dd * dd; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / K + eps);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nchunk) {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = (h[i][j] - mean) * rstd * gam[i][j] + bet[i][j];
        const u32x4 packed = (u32x4){pack2bf(y[0], y[1]), pack2bf(y[2], y[3]), pack2bf(y[4], y[5]), pack2bf(y[6], y[7])};
        *(u32x4*)(xs + (size_t)m * K + ch * 8) = packed;
        if (write_h) cst(packed, hnew, (unsigned)(m * K + ch * 8) * 2u);
      }
    }
  }
}

// wave-wide sum / max through DPP (quad xor 1, xor 2, mirror within 8, mirror within 16: ~8 cycles each instead of a ~100-cycle
// ds_bpermute per butterfly level) and four v_readlane; a different association than the xor butterfly of wave_sum / wave_max
#define VAC_DPP(V_, CTRL_) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V_), CTRL_, 0xf, 0xf, true))
__device__ __forceinline__ float wave_sum_fast(float v) {
  v += VAC_DPP(v, 0xB1); v += VAC_DPP(v, 0x4E); v += VAC_DPP(v, 0x141); v += VAC_DPP(v, 0x140);
  const int iv = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
}
__device__ __forceinline__ float wave_max_fast(float v) {
  v = fmaxf(v, VAC_DPP(v, 0xB1)); v = fmaxf(v, VAC_DPP(v, 0x4E)); v = fmaxf(v, VAC_DPP(v, 0x141)); v = fmaxf(v, VAC_DPP(v, 0x140));
  const int iv = __builtin_bit_cast(int, v);
  return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))),
               fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48))));
}
#undef VAC_DPP

// ---- single-query attention of one (row, head) pair (attn_decode_kernel's arithmetic) -------------------------------------
// q: 64 bf16 at qoff of qrs (coherent).  Keys / values: rows j = 0..Tk-1 at koff + j * ldb / voff + j * ldb of kvrs (AUX = COH
// for the self-attention cache, whose newest row was written in this launch; 0 for the static cross-attention K/V).
// nw = 4 waves split the keys when Tk >= 256 (as the per-op path picks attn_decode_kernel<4>), else wave 0 alone.
//
// One wave's share [k_lo, k_hi) of the keys.  KU > 0: the share fits 64 * KU keys and EVERY load of the phase (q, the key rows,
// the value rows) is issued before the first is consumed — the phase is a chain of dependent ~2 us round trips otherwise
// (q -> keys -> values per 64 keys: 12 us for a 133-key share); the arithmetic and its order are unchanged.  KU == 0: any length.
// (slot variant) qlds: the query as 64 bf16 in LDS instead of qrs; knew / vnew: the key / value row of position t_new in LDS — that
// row of the cache is being written by other workgroups in this launch and is not read.
struct AttnNew { const bf16_t* qlds; const bf16_t* knew; const bf16_t* vnew; int t_new; bool fast; };     // fast: DPP wave reductions

template <int AUX, int KU>
__device__ __forceinline__ void attn_share(rsrc_t qrs, unsigned qoff, rsrc_t kvrs, unsigned koff, unsigned voff, unsigned ldb, int k_lo, int k_hi,
                                           const uint8_t* km, float scale, float* probs, int lane, float& m, float& l, float (&acc)[8],
                                           const AttnNew& an) {
  const int kg = lane >> 3, dc = lane & 7;
  u32x4 qr[8];
  if (an.qlds) {
#pragma unroll
    for (int c = 0; c < 8; ++c) qr[c] = *(const u32x4*)(an.qlds + c * 8);
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) qr[c] = cld(qrs, qoff + c * 16);
  }
  if constexpr (KU > 0) {
    u32x4 kr[KU][8], vr[KU][8];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int key = k_lo + lane + 64 * u;
      const unsigned ro = koff + (unsigned)(key < k_hi ? key : k_lo) * ldb;
#pragma unroll
      for (int c = 0; c < 8; ++c) kr[u][c] = __builtin_amdgcn_raw_buffer_load_b128(kvrs, ro + c * 16, 0, AUX);
    }
#pragma unroll
    for (int it = 0; it < KU; ++it)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int key = k_lo + kg + 64 * it + 8 * u;
        vr[it][u] = __builtin_amdgcn_raw_buffer_load_b128(kvrs, voff + (unsigned)(key < k_hi ? key : k_lo) * ldb + dc * 16, 0, AUX);
      }
    if (an.t_new >= 0) {
#pragma unroll
      for (int u = 0; u < KU; ++u)
        if (k_lo + lane + 64 * u == an.t_new) {
#pragma unroll
          for (int c = 0; c < 8; ++c) kr[u][c] = *(const u32x4*)(an.knew + c * 8);
        }
#pragma unroll
      for (int it = 0; it < KU; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k_lo + kg + 64 * it + 8 * u == an.t_new) vr[it][u] = *(const u32x4*)(an.vnew + dc * 8);
    }
    float qf[64];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) { qf[c * 8 + 2 * i] = __uint_as_float(qr[c][i] << 16); qf[c * 8 + 2 * i + 1] = __uint_as_float(qr[c][i] & 0xffff0000u); }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int key = k_lo + lane + 64 * u;
      if (key < k_hi) {
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            dot += qf[c * 8 + 2 * i] * __uint_as_float(kr[u][c][i] << 16);
            dot += qf[c * 8 + 2 * i + 1] * __uint_as_float(kr[u][c][i] & 0xffff0000u);
          }
        float sc = dot * scale;
        if (km && km[key] == 0) sc += -3.4028234663852886e38f;
        probs[key] = sc;
        m = fmaxf(m, sc);
      }
    }
    m = an.fast ? wave_max_fast(m) : wave_max(m);
    for (int key = k_lo + lane; key < k_hi; key += 64) {
      const float e = __expf(probs[key] - m);
      probs[key] = e;
      l += e;
    }
    l = an.fast ? wave_sum_fast(l) : wave_sum(l);
#pragma unroll
    for (int it = 0; it < KU; ++it) {
      if (k_lo + kg + 64 * it < k_hi) {
        float pk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int key = k_lo + kg + 64 * it + 8 * u;
          pk[u] = key < k_hi ? probs[key] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[2 * i] += pk[u] * __uint_as_float(vr[it][u][i] << 16);
            acc[2 * i + 1] += pk[u] * __uint_as_float(vr[it][u][i] & 0xffff0000u);
          }
      }
    }
  } else {
    float qf[64];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) { qf[c * 8 + 2 * i] = __uint_as_float(qr[c][i] << 16); qf[c * 8 + 2 * i + 1] = __uint_as_float(qr[c][i] & 0xffff0000u); }
    for (int key0 = k_lo + lane; key0 < k_hi; key0 += 256) {
      u32x4 kr[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int key = key0 + 64 * u;
        const unsigned ro = koff + (unsigned)(key < k_hi ? key : key0) * ldb;
#pragma unroll
        for (int c = 0; c < 8; ++c) kr[u][c] = __builtin_amdgcn_raw_buffer_load_b128(kvrs, ro + c * 16, 0, AUX);
        if (key == an.t_new) {
#pragma unroll
          for (int c = 0; c < 8; ++c) kr[u][c] = *(const u32x4*)(an.knew + c * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int key = key0 + 64 * u;
        if (key < k_hi) {
          float dot = 0.f;
#pragma unroll
          for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              dot += qf[c * 8 + 2 * i] * __uint_as_float(kr[u][c][i] << 16);
              dot += qf[c * 8 + 2 * i + 1] * __uint_as_float(kr[u][c][i] & 0xffff0000u);
            }
          float sc = dot * scale;
          if (km && km[key] == 0) sc += -3.4028234663852886e38f;
          probs[key] = sc;
          m = fmaxf(m, sc);
        }
      }
    }
    m = wave_max(m);
    for (int key = k_lo + lane; key < k_hi; key += 64) {
      const float e = __expf(probs[key] - m);
      probs[key] = e;
      l += e;
    }
    l = wave_sum(l);
    for (int key0 = k_lo + kg; key0 < k_hi; key0 += 64) {
      u32x4 vr[8]; float pk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int key = key0 + 8 * u;
        const bool ok = key < k_hi;
        vr[u] = __builtin_amdgcn_raw_buffer_load_b128(kvrs, voff + (unsigned)(ok ? key : key0) * ldb + dc * 16, 0, AUX);
        if (key == an.t_new) vr[u] = *(const u32x4*)(an.vnew + dc * 8);
        pk[u] = ok ? probs[key] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] += pk[u] * __uint_as_float(vr[u][i] << 16);
          acc[2 * i + 1] += pk[u] * __uint_as_float(vr[u][i] & 0xffff0000u);
        }
    }
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
}

// res8 == nullptr: the 64 outputs go to ooff of ors (coherent store); else lanes dc = 0..7 of wave 0 (kg == 0) leave their 8 values there
template <int AUX>
__device__ __forceinline__ void attn_pair(rsrc_t qrs, unsigned qoff, rsrc_t kvrs, unsigned koff, unsigned voff, unsigned ldb, int Tk,
                                          const uint8_t* km, float scale, rsrc_t ors, unsigned ooff, float* probs, int wave, int lane,
                                          const AttnNew& an = AttnNew{nullptr, nullptr, nullptr, -1, false}, float* res8 = nullptr) {
  const int nw = Tk >= 256 ? NWAVE : 1;
  const int per = (Tk + nw - 1) / nw;
  const int k_lo = wave * per, k_hi = wave < nw ? min(Tk, k_lo + per) : k_lo;
  float m = -INFINITY, l = 0.f;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int kg = lane >> 3, dc = lane & 7;
  if (wave < nw) {
    // a wave reads back only the probabilities it wrote itself (LDS operations of one wave execute in order)
    if (per <= 64) attn_share<AUX, 1>(qrs, qoff, kvrs, koff, voff, ldb, k_lo, k_hi, km, scale, probs, lane, m, l, acc, an);
    else if (per <= 128) attn_share<AUX, 2>(qrs, qoff, kvrs, koff, voff, ldb, k_lo, k_hi, km, scale, probs, lane, m, l, acc, an);
    else if (per <= 192) attn_share<AUX, 3>(qrs, qoff, kvrs, koff, voff, ldb, k_lo, k_hi, km, scale, probs, lane, m, l, acc, an);
    else attn_share<AUX, 0>(qrs, qoff, kvrs, koff, voff, ldb, k_lo, k_hi, km, scale, probs, lane, m, l, acc, an);
  }
  if (nw > 1) {
    float* pacc = probs + Tk;                          // [4][64]
    float* pm = pacc + NWAVE * 64;                     // [4]
    float* pl = pm + NWAVE;                            // [4]
    if (kg == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) pacc[wave * 64 + dc * 8 + j] = acc[j];
    }
    if (lane == 0) { pm[wave] = m; pl[wave] = l; }
    __syncthreads();
    if (wave == 0) {
      float gm = pm[0];
#pragma unroll
      for (int w = 1; w < NWAVE; ++w) gm = fmaxf(gm, pm[w]);
      float gl = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) {
        const float sc = pl[w] > 0.f ? __expf(pm[w] - gm) : 0.f;
        gl += pl[w] * sc;
        if (kg == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += pacc[w * 64 + dc * 8 + j] * sc;
        }
      }
      l = gl;
    }
  }
  if (wave == 0 && kg == 0) {
    const float inv = 1.f / l;
    if (res8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) res8[j] = acc[j] * inv;
    } else {
      cst((u32x4){pack2bf(acc[0] * inv, acc[1] * inv), pack2bf(acc[2] * inv, acc[3] * inv), pack2bf(acc[4] * inv, acc[5] * inv),
                  pack2bf(acc[6] * inv, acc[7] * inv)}, ors, ooff + dc * 16);
    }
  }
}

__device__ __forceinline__ int wave_of(int tid) { return __builtin_amdgcn_readfirstlane(tid >> 6); }

#define TR(K_) do { if (p.trace && wg == p.trace_wg && tid == 0) p.trace[bar.ph * 8 + (K_)] = wall_clock64(); } while (0)
#define END_PHASE(PREFETCH)                                     \
  TR(3);                                                        \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              \
  TR(4);                                                        \
  bar_arrive(bar);                                              \
  { PREFETCH; }                                                 \
  TR(5 - 8);                                                    \
  if (!bar_wait(bar, &s_bad)) return;                           \
  TR(0);

__global__ __launch_bounds__(NTHR) void decoder_step_kernel(DecP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = (bf16_t*)smem;                                   // [R][max(d, F)] bf16
  char* wbig = smem + p.xs_bytes + wave_of(threadIdx.x) * WBIG;                         // this wave's prefetch slots
  char* wsml = smem + p.xs_bytes + NWAVE * WBIG + wave_of(threadIdx.x) * WSMALL;
  float* lnp = (float*)(smem + p.xs_bytes + NWAVE * (WBIG + WSMALL));     // gamma[1024] | beta[1024] of the next LayerNorm
  float* biasl = lnp + 2048;                                              // bias of this workgroup's (<= 16) output columns
  float* probs = biasl + 64;                                              // [max(S, t + 1)] + 4 * 64 + 8
  __shared__ float part[NWAVE][32];
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x, wg = blockIdx.x;
  const int R = p.R, d = p.d, F = p.F, H = p.H;
  if (tid == 0) s_bad = 0;
  Bar bar = {p.bar, G, 0u, false};

  const unsigned act_b = (unsigned)(R * d * 2), f_b = (unsigned)(R * F * 2);
  const rsrc_t rs_cache = mkrs(p.cache, p.cache_bytes);
  const rsrc_t rs_h0 = mkrs(p.h0, act_b), rs_hb0 = mkrs(p.hb0, act_b), rs_hb1 = mkrs(p.hb1, act_b);
  const rsrc_t rs_o = mkrs(p.o, act_b), rs_ctx = mkrs(p.ctx, act_b), rs_q = mkrs(p.q, act_b), rs_f = mkrs(p.f, f_b);
  const OutD out_o = {rs_o, 0u, (unsigned)(d * 2)}, out_q = {rs_q, 0u, (unsigned)(d * 2)}, out_f = {rs_f, 0u, (unsigned)(F * 2)};
  rsrc_t hcur = rs_h0;
  int nln = 0;                                                  // LayerNorm k writes hb[k & 1]
  const unsigned row_b = (unsigned)p.Tstride * 2u;              // bytes between two rows of one layer's cache
  const int Tk_self = p.t + 1;

  const unsigned wb_dd = (unsigned)(d * d * 2), wb_fd = (unsigned)(F * d * 2);
  const int c1 = wg * NWAVE + wave, c4 = c1 * 4;               // this wave's first column in a 1- / 4-column-per-wave phase
  // What is in flight across a barrier (issued between its arrive and its wait, retired inside the wait): the next projection's
  // weight tile (per-wave slots), its bias values and, before a LayerNorm phase, gamma | beta (workgroup-shared slots).
  {
    const vacnic_decoder_layer l0 = p.layers[0];
    w_issue<4>(wbig, mkrs(l0.w_kvq, 3 * wb_dd), 3 * d, d, c4, lane);
    bias_issue(biasl, l0.b_kvq, 3 * d, wg * 16, 16, wave, lane);
  }

  for (int li = 0; li < p.L; ++li) {
    const vacnic_decoder_layer ly = p.layers[li];
    const unsigned lay_b = (unsigned)li * (unsigned)R * row_b;
    // ---- P1: k|v|q of position t, in place in the cache (k|v at row t, q parked in the first d columns of row t + 1)
    if (li == 0) {
      stage_plain(rs_h0, xs, R, d, tid);
      w_wait();
    } else {
      const rsrc_t hn = (nln & 1) ? rs_hb1 : rs_hb0;
      stage_ln(rs_o, hcur, lnp, hn, wg == 0, xs, R, d, p.eps, wave, lane);
      hcur = hn; ++nln;
    }
    __syncthreads();
    TR(1);
    {
      const OutD out_c = {rs_cache, lay_b + (unsigned)p.t * (unsigned)(2 * d) * 2u, row_b};
      gemv_phase<4>(wbig, mkrs(ly.w_kvq, 3 * wb_dd), ly.b_kvq, biasl, 3 * d, d, VACNIC_ACT_NONE, out_c, xs, R, G, wg, wave, lane);
    }
    END_PHASE(w_issue<1>(wsml, mkrs(ly.w_so, wb_dd), d, d, c1, lane); bias_issue(biasl, ly.b_so, d, wg * 4, 4, wave, lane))
    // ---- P2: self-attention over cache rows 0..t
    for (int pr = wg; pr < R * H; pr += G) {
      const int r = pr / H, h = pr - r * H;
      const unsigned rb = lay_b + (unsigned)r * row_b + (unsigned)h * 128u;
      attn_pair<COH>(rs_cache, rb + (unsigned)(p.t + 1) * (unsigned)(2 * d) * 2u, rs_cache, rb, rb + (unsigned)d * 2u, (unsigned)(2 * d) * 2u,
                     Tk_self, nullptr, p.scale, rs_ctx, (unsigned)(r * d + h * 64) * 2u, probs, wave, lane);
      __syncthreads();
    }
    END_PHASE()
    // ---- P3: self-attention output projection
    stage_plain(rs_ctx, xs, R, d, tid);
    __syncthreads();
    TR(1);
    gemv_phase<1>(wsml, mkrs(ly.w_so, wb_dd), ly.b_so, biasl, d, d, VACNIC_ACT_NONE, out_o, xs, R, G, wg, wave, lane);
    END_PHASE(w_issue<1>(wsml, mkrs(ly.w_cq, wb_dd), d, d, c1, lane); bias_issue(biasl, ly.b_cq, d, wg * 4, 4, wave, lane);
              ln_issue(lnp, ly.ln_self_g, ly.ln_self_b, d, wave, lane))
    // ---- P4: post-LN of the self-attention block, cross-attention query
    {
      const rsrc_t hn = (nln & 1) ? rs_hb1 : rs_hb0;
      stage_ln(rs_o, hcur, lnp, hn, wg == 0, xs, R, d, p.eps, wave, lane);
      hcur = hn; ++nln;
    }
    __syncthreads();
    TR(1);
    gemv_phase<1>(wsml, mkrs(ly.w_cq, wb_dd), ly.b_cq, biasl, d, d, VACNIC_ACT_NONE, out_q, xs, R, G, wg, wave, lane);
    END_PHASE(w_issue<1>(wsml, mkrs(ly.w_co, wb_dd), d, d, c1, lane); bias_issue(biasl, ly.b_co, d, wg * 4, 4, wave, lane))
    // ---- P5: cross-attention over the encoder K/V
    {
      const unsigned kv_rows = ly.cross_bs == 0 ? 1u : (unsigned)R;
      const rsrc_t rs_kv = mkrs(ly.cross_kv, kv_rows * (unsigned)p.S * (unsigned)(2 * d) * 2u);
      for (int pr = wg; pr < R * H; pr += G) {
        const int r = pr / H, h = pr - r * H;
        const unsigned kb = (unsigned)r * (unsigned)ly.cross_bs * 2u + (unsigned)h * 128u;
        attn_pair<0>(rs_q, (unsigned)(r * d + h * 64) * 2u, rs_kv, kb, kb + (unsigned)d * 2u, (unsigned)(2 * d) * 2u, p.S,
                     p.enc_mask ? p.enc_mask + (size_t)r * p.S : nullptr, p.scale, rs_ctx, (unsigned)(r * d + h * 64) * 2u, probs, wave, lane);
        __syncthreads();
      }
    }
    END_PHASE()
    // ---- P6: cross-attention output projection
    stage_plain(rs_ctx, xs, R, d, tid);
    __syncthreads();
    TR(1);
    gemv_phase<1>(wsml, mkrs(ly.w_co, wb_dd), ly.b_co, biasl, d, d, VACNIC_ACT_NONE, out_o, xs, R, G, wg, wave, lane);
    END_PHASE(w_issue<4>(wbig, mkrs(ly.w_fc1, wb_fd), F, d, c4, lane); bias_issue(biasl, ly.b_fc1, F, wg * 16, 16, wave, lane);
              ln_issue(lnp, ly.ln_cross_g, ly.ln_cross_b, d, wave, lane))
    // ---- P7: post-LN of the cross-attention block, fc1 + GELU
    {
      const rsrc_t hn = (nln & 1) ? rs_hb1 : rs_hb0;
      stage_ln(rs_o, hcur, lnp, hn, wg == 0, xs, R, d, p.eps, wave, lane);
      hcur = hn; ++nln;
    }
    __syncthreads();
    TR(1);
    gemv_phase<4>(wbig, mkrs(ly.w_fc1, wb_fd), ly.b_fc1, biasl, F, d, VACNIC_ACT_GELU, out_f, xs, R, G, wg, wave, lane);
    END_PHASE(w2_issue(wbig, mkrs(ly.w_fc2, wb_fd), d, F, wg * 4, wave, lane); bias_issue(biasl, ly.b_fc2, d, wg * 4, 4, wave, lane))
    // ---- P8: fc2
    stage_plain(rs_f, xs, R, F, tid);
    __syncthreads();
    TR(1);
    gemv2_phase(wbig, mkrs(ly.w_fc2, wb_fd), ly.b_fc2, biasl, d, F, out_o, xs, R, G, wg, wave, lane, part);
    if (li + 1 < p.L) {
      const vacnic_decoder_layer nx = p.layers[li + 1];
      END_PHASE(w_issue<4>(wbig, mkrs(nx.w_kvq, 3 * wb_dd), 3 * d, d, c4, lane); bias_issue(biasl, nx.b_kvq, 3 * d, wg * 16, 16, wave, lane);
                ln_issue(lnp, ly.ln_final_g, ly.ln_final_b, d, wave, lane))
    }
  }
  // exit: the last workgroup out clears the barrier state (stream order makes it visible to the next launch)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    if (__hip_atomic_fetch_add(p.bar + BAR_EXIT, 1u, __ATOMIC_RELAXED, AGENT) == (unsigned)G - 1) {
      for (int i = 0; i < NGRP; ++i) __hip_atomic_store(p.bar + 32 * i, 0u, __ATOMIC_RELAXED, AGENT);
      __hip_atomic_store(p.bar + BAR_TOP, 0u, __ATOMIC_RELAXED, AGENT);
      for (int f = 0; f < NFLAG; ++f) __hip_atomic_store(p.bar + BAR_FLAG + 32 * f, 0u, __ATOMIC_RELAXED, AGENT);
      __hip_atomic_store(p.bar + BAR_EXIT, 0u, __ATOMIC_RELAXED, AGENT);
    }
  }
}


// =====================================================================================================================
// Slot variant (the default): no barriers at all.  A phase's producers hand their results to the consumers through TAGGED
// SLOTS in HBM: a slot is a run of 16-byte units {6 bf16 values, 32-bit tag}, written with one coherent 16-byte store per unit
// and NO wait for the acknowledgement; a consumer thread polls "its" producer's slot until every unit carries the tag of
// (this launch, this phase) and scatters the values into the workgroup's LDS copy of the activation.  The data IS the
// signal, so a phase boundary costs one store-to-visible latency plus one poll round trip (~1.5 us) instead of store
// acknowledgement + two dependent atomics + flag poll + reload (~4 us with the barrier).  Tag = nonce * 1024 + phase, nonce =
// a per-buffer launch counter (read by everybody at the start, bumped by workgroup 0 at its exit — which cannot happen before
// every workgroup has produced its last phase), so a stale slot can never match.  A slot array is reused once per layer:
// whoever produces phase X of layer l+1 has consumed data that transitively required every workgroup to be past its read of
// (l, X).  The residual stream h never leaves the chip: every workgroup normalises all rows anyway and keeps h in LDS.
//   slot arrays (bytes): 6 GEMV outputs x 256 workgroups x 384  |  2 attention outputs x 128 (row, head) pairs x 192
constexpr unsigned SLOT_STRIDE = 384, CTX_STRIDE = 192, SLOT_ARR = 256 * SLOT_STRIDE, CTX_ARR0 = 6 * SLOT_ARR, CTX_ARR = 128 * CTX_STRIDE;
constexpr unsigned FC1L_ARR0 = CTX_ARR0 + 2 * CTX_ARR;      // + layer * SLOT_ARR: the fc1 outputs get one array PER LAYER (see gather_cols)
constexpr int MAX_L = 120;
constexpr int BAR_NONCE = BAR_ERR + 32;
enum { A_KVQ = 0, A_SO = 1, A_CQ = 2, A_CO = 3, A_FC1 = 4, A_FC2 = 5 };

struct PollCtx { unsigned* errw; int* s_bad; };

// fetch `units` 16-byte units of one slot until all carry `tag`
// VIA_L2 (slot arrays written ONCE per launch — the per-layer fc1 arrays): after the coherent poll of the last unit the slot is read
// with ordinary loads, so the 32 workgroups of an XCD share one fetch of its 57 KB instead of 32 trips to the memory side; the L2
// was invalidated at kernel start and nobody reads a line before it is written, except that a line fetched while its units were
// landing may be partially old — the tags catch that and the slot is re-read coherently.
// (Keeping two or three probes of a slot in flight, ~0.4 us apart, to see it sooner was measured SLOWER: 555 -> 679 us per token —
// the memory side is sensitive to the extra poll traffic of 65 K threads.)
template <int UMAX, bool VIA_L2 = false>
__device__ __forceinline__ bool slot_fetch(rsrc_t srs, unsigned off, int units, unsigned tag, u32x4 (&un)[UMAX], const PollCtx& pc) {
  const long long t0 = wall_clock64();
  unsigned spins = 0;
  bool plain = VIA_L2;
  for (;;) {
    bool ok = true;
    if (units > 11) ok = cld(srs, off + (unsigned)(units - 1) * 16u)[3] == tag;     // long slots: the last unit first (1/14 of the traffic)
    if (ok) {
      if (plain) {
#pragma unroll
        for (int u = 0; u < UMAX; ++u)
          if (u < units) un[u] = __builtin_amdgcn_raw_buffer_load_b128(srs, off + u * 16, 0, 0);
        plain = false;
      } else {
#pragma unroll
        for (int u = 0; u < UMAX; ++u)
          if (u < units) un[u] = cld(srs, off + u * 16);
      }
#pragma unroll
      for (int u = 0; u < UMAX; ++u)
        if (u < units) ok = ok && un[u][3] == tag;
      if (ok) return true;
    }
    if ((++spins & 63u) == 0 && (wall_clock64() - t0 > 200000000ll || __hip_atomic_load(pc.errw, __ATOMIC_RELAXED, AGENT) != 0)) {
      __hip_atomic_store(pc.errw, 1u, __ATOMIC_RELAXED, AGENT);
      *pc.s_bad = 1;
      return false;
    }
  }
}

// consumer of a GEMV output [R][N]: thread i polls producer i's slot (ncw = 1 << lg columns x R rows) and scatters it into xs
template <int UMAX, bool VIA_L2 = false>
__device__ __forceinline__ void gather_cols(rsrc_t srs, unsigned arr, int nslots, int R, int lg, unsigned tag, bf16_t* xs, int K, int tid,
                                            const PollCtx& pc) {
  if (tid < nslots) {
    const int ncw = 1 << lg, nval = R * ncw, units = (nval + 5) / 6;
    u32x4 un[UMAX];
    if (slot_fetch<UMAX, VIA_L2>(srs, arr + (unsigned)tid * SLOT_STRIDE, units, tag, un, pc)) {
#pragma unroll
      for (int u = 0; u < UMAX; ++u)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int v = 6 * u + 2 * j;
          if (u < units && v < nval) *(unsigned*)(xs + (size_t)(v >> lg) * K + tid * ncw + (v & (ncw - 1))) = un[u][j];
        }
    }
  }
}
// consumer of an attention output: thread i polls (row, head) pair i's slot (64 values) into xs[row][head * 64 ..]
__device__ __forceinline__ void gather_ctx(rsrc_t srs, unsigned arr, int npair, int H, unsigned tag, bf16_t* xs, int K, int tid, const PollCtx& pc) {
  if (tid < npair) {
    u32x4 un[11];
    if (slot_fetch<11>(srs, arr + (unsigned)tid * CTX_STRIDE, 11, tag, un, pc)) {
      const int r = tid / H, h = tid - r * H;
#pragma unroll
      for (int u = 0; u < 11; ++u)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int v = 6 * u + 2 * j;
          if (v < 64) *(unsigned*)(xs + (size_t)r * K + h * 64 + v) = un[u][j];
        }
    }
  }
}
// a (row r, head h) pair's 64-value pieces of a GEMV output: columns col0 .. col0+63 of row r -> dst[0..63] (npc = 64 >> lg slots)
template <int UMAX>
__device__ __forceinline__ void gather_piece(rsrc_t srs, unsigned arr, int col0, int R, int r, int lg, unsigned tag, bf16_t* dst, int j, const PollCtx& pc) {
  const int ncw = 1 << lg, nval = R * ncw, units = (nval + 5) / 6;
  u32x4 un[UMAX];
  if (slot_fetch<UMAX>(srs, arr + (unsigned)((col0 >> lg) + j) * SLOT_STRIDE, units, tag, un, pc)) {
#pragma unroll
    for (int u = 0; u < UMAX; ++u)
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const int v = 6 * u + 2 * jj;
        if (u < units && v < nval && (v >> lg) == r) *(unsigned*)(dst + j * ncw + (v & (ncw - 1))) = un[u][jj];
      }
  }
}
// producer: the workgroup's packed results (bf16 in LDS, value v = row * ncw + column) -> its slot, one unit per thread
__device__ __forceinline__ void slot_put(rsrc_t srs, unsigned off, const unsigned short* pack, int nval, unsigned tag, int tid) {
  if (tid < (nval + 5) / 6) {
    const unsigned* pw = (const unsigned*)(pack + 6 * tid);
    cst((u32x4){pw[0], pw[1], pw[2], tag}, srs, off + (unsigned)tid * 16u);
  }
}

// ---- slot variant: MFMA projections ------------------------------------------------------------------------------------
// A workgroup's output tile is NC = 16 (k|v|q, fc1) or 4 (d-wide projections, fc2) weight rows x all activation rows; one
// v_mfma_f32_16x16x32_bf16 multiplies 16 weight rows by 16 activation rows (R valid, the rest zero) over 32 k, the 4 waves split
// K and meet in LDS.  ~0.3 us instead of ~500-1000 VALU instructions per wave (1.5-3 us with one wave per SIMD).  The weight
// tile arrives by LDS-DMA already in fragment order:
//   16-row tile: piece s of wave w = MFMA step kk = w * per + s: lane l <- W[n0 + (l & 15)][8 * (4 kk + (l >> 4)) .. + 8]
//   4-row tile:  piece s of wave w covers 4 steps: lane l <- W[n0 + (l & 3)][8 * (16 (w * per4 + s) + (l >> 2)) .. + 8]
// Sums differ from the VALU path in the last fp32 bits (different association).
__device__ __forceinline__ int steps_per_wave(int K) { return ((K + 31) / 32 + NWAVE - 1) / NWAVE; }

__device__ __forceinline__ void w16_issue(char* wl, rsrc_t wrs, int N, int K, int n0, int wave, int lane) {
  const int nchunk = K >> 3, per = steps_per_wave(K);
  const int col = n0 + (lane & 15);
  for (int s_ = 0; s_ < per; ++s_) {
    const int ch = 4 * (wave * per + s_) + (lane >> 4);
    wdma(wrs, wl + s_ * 1024, (col < N && ch < nchunk) ? (unsigned)(col * K + ch * 8) * 2u : OOB);
  }
}
__device__ __forceinline__ void w4_issue(char* wl, rsrc_t wrs, int N, int K, int n0, int wave, int lane) {
  const int nchunk = K >> 3, per4 = (steps_per_wave(K) + 3) / 4;
  const int col = n0 + (lane & 3);
  for (int s_ = 0; s_ < per4; ++s_) {
    const int ch = 16 * (wave * per4 + s_) + (lane >> 2);
    wdma(wrs, wl + s_ * 1024, (col < N && ch < nchunk) ? (unsigned)(col * K + ch * 8) * 2u : OOB);
  }
}
// this wave's share of D[weight row][activation row]; xs rows have stride Kp (elements), rows >= R count as zero
template <bool NC16>
__device__ __forceinline__ f32x4 mfma_share(const char* wl, const bf16_t* xs, int Kp, int R, int K, int wave, int lane) {
  const int nchunk = K >> 3;
  const int per = NC16 ? steps_per_wave(K) : 4 * ((steps_per_wave(K) + 3) / 4);
  const int m = lane & 15, g = lane >> 4;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int s_ = 0; s_ < per; ++s_) {
    const int kk = wave * per + s_, ch = 4 * kk + g;
    u32x4 xr = (u32x4){0, 0, 0, 0}, wr = (u32x4){0, 0, 0, 0};
    if (m < R && ch < nchunk) xr = *(const u32x4*)(xs + (size_t)m * Kp + ch * 8);
    if (NC16) wr = *(const u32x4*)(wl + s_ * 1024 + lane * 16);
    else if (m < 4) wr = *(const u32x4*)(wl + (s_ >> 2) * 1024 + (m + 4 * (4 * (s_ & 3) + g)) * 16);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wr), __builtin_bit_cast(bf16x8, xr), acc, 0, 0, 0);
  }
  return acc;
}
// the 4 waves' shares meet in LDS; thread e = lane * 4 + j of the fragment layout owns (weight row n = 4 (e >> 6) + (e & 3),
// activation row m = (e >> 2) & 15): bias, activation, bf16 into pack[m * ncw + n].  Ends with a workgroup barrier.
template <bool NC16>
__device__ __forceinline__ void mfma_finish(f32x4 acc, float* red, const float* biasl, int act, unsigned short* pack, int R, int tid, int wave, int lane) {
  *(f32x4*)(red + wave * 256 + lane * 4) = acc;
  __syncthreads();
  const int n = 4 * (tid >> 6) + (tid & 3), m = (tid >> 2) & 15;
  constexpr int NC = NC16 ? 16 : 4;
  if (m < R && n < NC) {
    float v = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid] + biasl[n];
    pack[m * NC + n] = f2bf(act_fwd(act, v));
  }
  __syncthreads();
}

// LayerNorm in place: xs rows hold o, hres rows hold the residual h; xs <- LN(o + h) (bf16), hres <- the same rows
// (add_ln_fwd_kernel's arithmetic: one wave per row, chunks lane / lane + 64, fp32 statistics; two rows per wave interleaved)
__device__ __forceinline__ void ln_inplace(bf16_t* xs, int Kp, bf16_t* hres, const float* lnp, int R, int K, float eps, int wave, int lane) {
  const int nchunk = K >> 3;
  float h[2][2][8];
  float s[2] = {0.f, 0.f};
  const bool on[2] = {wave < R, wave + NWAVE < R};
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = wave + NWAVE * rr, ch = lane + 64 * i;
      if (on[rr] && ch < nchunk) {
        float xv[8], rv[8];
        unpack8(*(const u32x4*)(xs + (size_t)m * Kp + ch * 8), xv);
        unpack8(*(const u32x4*)(hres + (size_t)m * K + ch * 8), rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) { xv[j] += rv[j]; h[rr][i][j] = xv[j]; s[rr] += xv[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) h[rr][i][j] = 0.f;
      }
    }
  s[0] = wave_sum_fast(s[0]); s[1] = wave_sum_fast(s[1]);
  float mean[2], q[2] = {0.f, 0.f};
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    mean[rr] = s[rr] / K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (lane + 64 * i < nchunk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float dd = h[rr][i][j] - mean[rr]; q[rr] += dd * dd; }
      }
  }
  q[0] = wave_sum_fast(q[0]); q[1] = wave_sum_fast(q[1]);
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const float rstd = rsqrtf(q[rr] / K + eps);
    const int m = wave + NWAVE * rr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = lane + 64 * i;
      if (on[rr] && ch < nchunk) {
        const f32x4 g0 = *(const f32x4*)(lnp + ch * 8), g1 = *(const f32x4*)(lnp + ch * 8 + 4);
        const f32x4 b0 = *(const f32x4*)(lnp + 1024 + ch * 8), b1 = *(const f32x4*)(lnp + 1024 + ch * 8 + 4);
        float y[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          y[j] = (h[rr][i][j] - mean[rr]) * rstd * g0[j] + b0[j];
          y[4 + j] = (h[rr][i][4 + j] - mean[rr]) * rstd * g1[j] + b1[j];
        }
        const u32x4 packed = (u32x4){pack2bf(y[0], y[1]), pack2bf(y[2], y[3]), pack2bf(y[4], y[5]), pack2bf(y[6], y[7])};
        *(u32x4*)(xs + (size_t)m * Kp + ch * 8) = packed;
        *(u32x4*)(hres + (size_t)m * K + ch * 8) = packed;
      }
    }
  }
}


#define TR3(K_) do { if (p.trace && wg == p.trace_wg && tid == 0) p.trace[phase * 8 + (K_)] = wall_clock64(); } while (0)
#define SYNC_OR_QUIT() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); if (s_bad) return; } while (0)

__global__ __launch_bounds__(NTHR) void decoder_step_slots_kernel(DecP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  const int R = p.R, d = p.d, F = p.F, H = p.H;
  const int Kd = d + 8, Kf = F + 8;                             // xs row strides (elements): +16 bytes spreads the rows over the LDS banks
  bf16_t* xs = (bf16_t*)smem;                                   // [R][max(d, F) + 8] bf16: the current phase's input rows
  char* wbig = smem + p.xs_bytes + wave * WBIG;
  char* wsml = smem + p.xs_bytes + NWAVE * WBIG + wave * WSMALL;
  float* lnp = (float*)(smem + p.xs_bytes + NWAVE * (WBIG + WSMALL));
  float* biasl = lnp + 2048;
  bf16_t* hres = (bf16_t*)(biasl + 64);                         // [R][d] bf16: the residual stream
  unsigned short* pack = (unsigned short*)(hres + (size_t)MR * 1024);   // [<= 8 rows][<= 16 columns] (+ unit padding)
  bf16_t* qkvn = (bf16_t*)(pack + 160);                         // k | v | q of this position for one (row, head) pair: [3][64]
  float* red = (float*)(qkvn + 192);                            // [4 waves][256]: partial MFMA tiles
  float* probs = red + 1024;
  __shared__ int s_bad;
  if (tid == 0) s_bad = 0;
  const PollCtx pc = {p.bar + BAR_ERR, &s_bad};
  const unsigned tag0 = __hip_atomic_load(p.bar + BAR_NONCE, __ATOMIC_RELAXED, AGENT) << 10;
  const rsrc_t srs = mkrs(p.slots, FC1L_ARR0 + (unsigned)p.L * SLOT_ARR);
  const rsrc_t rs_cache = mkrs(p.cache, p.cache_bytes);
  const unsigned row_b = (unsigned)p.Tstride * 2u;
  const unsigned wb_dd = (unsigned)(d * d * 2), wb_fd = (unsigned)(F * d * 2);
  const int npair = R * H, t_dd = d >> 2, t_3d = (3 * d) >> 4, t_f = F >> 4;    // producers of an N = d / 3d / F phase
  int phase = 0;
  if (p.trace && wg == p.trace_wg && tid == 0) { p.trace[p.L * 64 + 4] = wall_clock64(); p.trace[p.L * 64 + 5] = clock64(); }
  {
    const vacnic_decoder_layer l0 = p.layers[0];
    w16_issue(wbig, mkrs(l0.w_kvq, 3 * wb_dd), 3 * d, d, wg * 16, wave, lane);
    bias_issue(biasl, l0.b_kvq, 3 * d, wg * 16, 16, wave, lane);
  }
  // h <- h0 (written by the embedding kernel before this launch)
  for (int c = tid; c < R * (d >> 3); c += NTHR) {
    const int m = c / (d >> 3), ch = c - m * (d >> 3);
    const u32x4 v = *(const u32x4*)(p.h0 + (size_t)c * 8);
    *(u32x4*)(hres + (size_t)c * 8) = v;
    *(u32x4*)(xs + (size_t)m * Kd + ch * 8) = v;
  }
  for (int li = 0; li < p.L; ++li) {
    const vacnic_decoder_layer ly = p.layers[li];
    const unsigned lay_b = (unsigned)li * (unsigned)R * row_b;
    const unsigned tg = tag0 + (unsigned)li * 8u;               // tag of this layer's phase X: tg + X + 1
    // ---- P1: k|v|q.  k|v also go to the cache row of position t (plain stores: read by later launches only)
    phase = li * 8;
    if (li > 0) gather_cols<6>(srs, A_FC2 * SLOT_ARR, t_dd, R, 2, tg, xs, Kd, tid, pc);     // tag of (li - 1, P8) = tg
    SYNC_OR_QUIT();
    TR3(1);
    if (li > 0) { ln_inplace(xs, Kd, hres, lnp, R, d, p.eps, wave, lane); __syncthreads(); }
    mfma_finish<true>(mfma_share<true>(wbig, xs, Kd, R, d, wave, lane), red, biasl, VACNIC_ACT_NONE, pack, R, tid, wave, lane);
    TR3(3);
    if (wg < t_3d) {
      slot_put(srs, A_KVQ * SLOT_ARR + (unsigned)wg * SLOT_STRIDE, pack, R * 16, tg + 1, tid);
      const int m = tid >> 4, n = wg * 16 + (tid & 15);
      if (m < R && n < 2 * d) p.cache[(size_t)li * R * p.Tstride + (size_t)m * p.Tstride + (size_t)p.t * 2 * d + n] = pack[m * 16 + (tid & 15)];
    }
    w4_issue(wsml, mkrs(ly.w_so, wb_dd), d, d, wg * 4, wave, lane); bias_issue(biasl, ly.b_so, d, wg * 4, 4, wave, lane);
    // ---- P2: self-attention of (row, head) pair wg over cache rows 0..t-1 and the new row
    phase = li * 8 + 1;
    if (wg < npair) {
      const int r = wg / H, h = wg - r * H;
      if (tid < 12) {
        const int which = tid >> 2;
        gather_piece<22>(srs, A_KVQ * SLOT_ARR, which * d + h * 64, R, r, 4, tg + 1, qkvn + which * 64, tid & 3, pc);
      }
      SYNC_OR_QUIT();
      TR3(1);
      const unsigned rb = lay_b + (unsigned)r * row_b + (unsigned)h * 128u;
      const AttnNew an = {qkvn + 128, qkvn, qkvn + 64, p.t, true};
      float res[8];
      attn_pair<0>(rs_cache, 0u, rs_cache, rb, rb + (unsigned)d * 2u, (unsigned)(2 * d) * 2u, p.t + 1, nullptr, p.scale, rs_cache, 0u, probs,
                   wave, lane, an, res);
      if (wave == 0 && lane < 8)
        *(u32x4*)(pack + lane * 8) = (u32x4){pack2bf(res[0], res[1]), pack2bf(res[2], res[3]), pack2bf(res[4], res[5]), pack2bf(res[6], res[7])};
      __syncthreads();
      TR3(3);
      slot_put(srs, CTX_ARR0 + (unsigned)wg * CTX_STRIDE, pack, 64, tg + 2, tid);
    }
    // ---- P3: self-attention output projection
    phase = li * 8 + 2;
    gather_ctx(srs, CTX_ARR0, npair, H, tg + 2, xs, Kd, tid, pc);
    SYNC_OR_QUIT();
    TR3(1);
    mfma_finish<false>(mfma_share<false>(wsml, xs, Kd, R, d, wave, lane), red, biasl, VACNIC_ACT_NONE, pack, R, tid, wave, lane);
    TR3(3);
    if (wg < t_dd) slot_put(srs, A_SO * SLOT_ARR + (unsigned)wg * SLOT_STRIDE, pack, R * 4, tg + 3, tid);
    w4_issue(wsml, mkrs(ly.w_cq, wb_dd), d, d, wg * 4, wave, lane); bias_issue(biasl, ly.b_cq, d, wg * 4, 4, wave, lane);
    ln_issue(lnp, ly.ln_self_g, ly.ln_self_b, d, wave, lane);
    // ---- P4: post-LN of the self-attention block, cross-attention query
    phase = li * 8 + 3;
    gather_cols<6>(srs, A_SO * SLOT_ARR, t_dd, R, 2, tg + 3, xs, Kd, tid, pc);
    SYNC_OR_QUIT();
    TR3(1);
    ln_inplace(xs, Kd, hres, lnp, R, d, p.eps, wave, lane);
    __syncthreads();
    TR3(2);
    mfma_finish<false>(mfma_share<false>(wsml, xs, Kd, R, d, wave, lane), red, biasl, VACNIC_ACT_NONE, pack, R, tid, wave, lane);
    TR3(3);
    if (wg < t_dd) slot_put(srs, A_CQ * SLOT_ARR + (unsigned)wg * SLOT_STRIDE, pack, R * 4, tg + 4, tid);
    w4_issue(wsml, mkrs(ly.w_co, wb_dd), d, d, wg * 4, wave, lane); bias_issue(biasl, ly.b_co, d, wg * 4, 4, wave, lane);
    // ---- P5: cross-attention of pair wg over the encoder K/V
    phase = li * 8 + 4;
    if (wg < npair) {
      const int r = wg / H, h = wg - r * H;
      if (tid < 16) gather_piece<6>(srs, A_CQ * SLOT_ARR, h * 64, R, r, 2, tg + 4, qkvn + 128, tid, pc);
      SYNC_OR_QUIT();
      TR3(1);
      const unsigned kv_rows = ly.cross_bs == 0 ? 1u : (unsigned)R;
      const rsrc_t rs_kv = mkrs(ly.cross_kv, kv_rows * (unsigned)p.S * (unsigned)(2 * d) * 2u);
      const unsigned kb = (unsigned)r * (unsigned)ly.cross_bs * 2u + (unsigned)h * 128u;
      const AttnNew an = {qkvn + 128, nullptr, nullptr, -1, true};
      float res[8];
      attn_pair<0>(rs_kv, 0u, rs_kv, kb, kb + (unsigned)d * 2u, (unsigned)(2 * d) * 2u, p.S, p.enc_mask ? p.enc_mask + (size_t)r * p.S : nullptr,
                   p.scale, rs_kv, 0u, probs, wave, lane, an, res);
      if (wave == 0 && lane < 8)
        *(u32x4*)(pack + lane * 8) = (u32x4){pack2bf(res[0], res[1]), pack2bf(res[2], res[3]), pack2bf(res[4], res[5]), pack2bf(res[6], res[7])};
      __syncthreads();
      TR3(3);
      slot_put(srs, CTX_ARR0 + CTX_ARR + (unsigned)wg * CTX_STRIDE, pack, 64, tg + 5, tid);
    }
    // ---- P6: cross-attention output projection
    phase = li * 8 + 5;
    gather_ctx(srs, CTX_ARR0 + CTX_ARR, npair, H, tg + 5, xs, Kd, tid, pc);
    SYNC_OR_QUIT();
    TR3(1);
    mfma_finish<false>(mfma_share<false>(wsml, xs, Kd, R, d, wave, lane), red, biasl, VACNIC_ACT_NONE, pack, R, tid, wave, lane);
    TR3(3);
    if (wg < t_dd) slot_put(srs, A_CO * SLOT_ARR + (unsigned)wg * SLOT_STRIDE, pack, R * 4, tg + 6, tid);
    w16_issue(wbig, mkrs(ly.w_fc1, wb_fd), F, d, wg * 16, wave, lane); bias_issue(biasl, ly.b_fc1, F, wg * 16, 16, wave, lane);
    ln_issue(lnp, ly.ln_cross_g, ly.ln_cross_b, d, wave, lane);
    // ---- P7: post-LN of the cross-attention block, fc1 + GELU
    phase = li * 8 + 6;
    gather_cols<6>(srs, A_CO * SLOT_ARR, t_dd, R, 2, tg + 6, xs, Kd, tid, pc);
    SYNC_OR_QUIT();
    TR3(1);
    ln_inplace(xs, Kd, hres, lnp, R, d, p.eps, wave, lane);
    __syncthreads();
    TR3(2);
    mfma_finish<true>(mfma_share<true>(wbig, xs, Kd, R, d, wave, lane), red, biasl, VACNIC_ACT_GELU, pack, R, tid, wave, lane);
    TR3(3);
    if (wg < t_f) slot_put(srs, FC1L_ARR0 + (unsigned)li * SLOT_ARR + (unsigned)wg * SLOT_STRIDE, pack, R * 16, tg + 7, tid);
    w4_issue(wbig, mkrs(ly.w_fc2, wb_fd), d, F, wg * 4, wave, lane); bias_issue(biasl, ly.b_fc2, d, wg * 4, 4, wave, lane);
    // ---- P8: fc2
    phase = li * 8 + 7;
    gather_cols<22, true>(srs, FC1L_ARR0 + (unsigned)li * SLOT_ARR, t_f, R, 4, tg + 7, xs, Kf, tid, pc);
    SYNC_OR_QUIT();
    TR3(1);
    mfma_finish<false>(mfma_share<false>(wbig, xs, Kf, R, F, wave, lane), red, biasl, VACNIC_ACT_NONE, pack, R, tid, wave, lane);
    TR3(3);
    if (wg < t_dd) slot_put(srs, A_FC2 * SLOT_ARR + (unsigned)wg * SLOT_STRIDE, pack, R * 4, tg + 8, tid);
    ln_issue(lnp, ly.ln_final_g, ly.ln_final_b, d, wave, lane);
    if (li + 1 < p.L) {
      const vacnic_decoder_layer nx = p.layers[li + 1];
      w16_issue(wbig, mkrs(nx.w_kvq, 3 * wb_dd), 3 * d, d, wg * 16, wave, lane); bias_issue(biasl, nx.b_kvq, 3 * d, wg * 16, 16, wave, lane);
    }
  }
  if (p.trace && wg == p.trace_wg && tid == 0) { p.trace[p.L * 64 + 6] = wall_clock64(); p.trace[p.L * 64 + 7] = clock64(); }
  // workgroup 0 applies the last layer's final LayerNorm, hands the hidden rows to the LM-head kernel and bumps the launch nonce
  if (wg == 0) {
    phase = p.L * 8;
    gather_cols<6>(srs, A_FC2 * SLOT_ARR, t_dd, R, 2, tag0 + (unsigned)(p.L - 1) * 8u + 8u, xs, Kd, tid, pc);
    SYNC_OR_QUIT();
    ln_inplace(xs, Kd, hres, lnp, R, d, p.eps, wave, lane);
    __syncthreads();
    TR3(2);
    for (int c = tid; c < R * (d >> 3); c += NTHR) *(u32x4*)(p.o + (size_t)c * 8) = *(const u32x4*)(hres + (size_t)c * 8);
    if (tid == 0) __hip_atomic_store(p.bar + BAR_NONCE, (tag0 >> 10) + 1u, __ATOMIC_RELAXED, AGENT);
  }
}

}  // namespace

extern "C" int64_t vacnic_decoder_step_sync_bytes(void) { return (int64_t)(BAR_NONCE + 32) * 4; }
extern "C" int64_t vacnic_decoder_step_slots_bytes(int64_t L) { return (int64_t)FC1L_ARR0 + (L < 1 ? 1 : L) * (int64_t)SLOT_ARR; }

extern "C" int vacnic_decoder_step(const vacnic_decoder_step_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_decoder_step, a, stream);
  VCHECK(a && a->layers && a->cache && a->h0 && a->hbuf[0] && a->hbuf[1] && a->obuf && a->ctx && a->qbuf && a->fbuf && a->sync,
         VACNIC_BAD_SHAPE, "decoder_step: null operand");
  VCHECK(a->L >= 1 && a->R >= 1 && a->R <= MR, VACNIC_UNSUPPORTED, "decoder_step: 1 <= R <= 8 rows (beams x batch), L >= 1");
  VCHECK(a->d >= 64 && a->d <= 1024 && (a->d & 7) == 0 && a->H * 64 == a->d, VACNIC_UNSUPPORTED,
         "decoder_step: d_model <= 1024, multiple of 8, heads of 64");
  VCHECK(a->F >= 8 && a->F <= 4096 && (a->F & 7) == 0, VACNIC_UNSUPPORTED, "decoder_step: ffn_dim <= 4096, multiple of 8");
  VCHECK(a->S >= 1 && a->S <= 8192 && a->t >= 0 && a->t < a->Tmax && a->Tmax <= 8191, VACNIC_BAD_SHAPE, "decoder_step: 0 <= t < Tmax, 1 <= S <= 8192");
  const int64_t cache_bytes = a->L * a->R * (a->Tmax + 1) * 2 * a->d * 2;
  VCHECK(cache_bytes < ((int64_t)1 << 31) && a->R * a->S * 2 * a->d * 2 < ((int64_t)1 << 31), VACNIC_UNSUPPORTED, "decoder_step: cache beyond 2 GiB");
  VCHECK(aligned16(a->cache) && aligned16(a->h0) && aligned16(a->hbuf[0]) && aligned16(a->hbuf[1]) && aligned16(a->obuf) && aligned16(a->ctx) &&
         aligned16(a->qbuf) && aligned16(a->fbuf), VACNIC_MISALIGNED, "decoder_step: 16-byte aligned buffers");
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) {
      vacnic_set_error("decoder_step: cannot query the CU count");
      return VACNIC_HIP_ERROR;
    }
    n_cu = v;
  }
  DecP p;
  p.layers = a->layers; p.cache = (bf16_t*)a->cache; p.h0 = (const bf16_t*)a->h0;
  p.hb0 = (bf16_t*)a->hbuf[0]; p.hb1 = (bf16_t*)a->hbuf[1]; p.o = (bf16_t*)a->obuf; p.ctx = (bf16_t*)a->ctx; p.q = (bf16_t*)a->qbuf;
  p.f = (bf16_t*)a->fbuf; p.enc_mask = a->enc_mask; p.bar = a->sync;
  p.L = (int)a->L; p.R = (int)a->R; p.d = (int)a->d; p.H = (int)a->H; p.F = (int)a->F; p.S = (int)a->S; p.t = (int)a->t;
  p.Tstride = (int)((a->Tmax + 1) * 2 * a->d);
  p.cache_bytes = (unsigned)cache_bytes;
  const int64_t kmax = a->d > a->F ? a->d : a->F;
  p.xs_bytes = (unsigned)(a->R * kmax * 2);
  p.eps = a->eps; p.scale = a->scale;
  p.trace = (unsigned long long*)a->trace; p.trace_wg = (int)a->trace_wg;
  p.slots = (char*)a->slots;
  const int64_t tkmax = a->S > a->t + 1 ? a->S : a->t + 1;
  const bool use_slots = a->slots != nullptr;
  size_t lds = (size_t)p.xs_bytes + (size_t)NWAVE * (WBIG + WSMALL) + (size_t)(2048 + 64) * 4 + (size_t)(tkmax + NWAVE * 64 + 8) * 4;
  if (use_slots) {
    p.xs_bytes = (unsigned)(a->R * (kmax + 8) * 2);
    lds = (size_t)p.xs_bytes + (size_t)NWAVE * (WBIG + WSMALL) + (size_t)(2048 + 64) * 4 + (size_t)MR * 1024 * 2 + 160 * 2 + 192 * 2 + 1024 * 4 +
          (size_t)(tkmax + NWAVE * 64 + 8) * 4;
  }
  VCHECK(lds <= 150 * 1024, VACNIC_UNSUPPORTED, "decoder_step: LDS budget");
  static bool lds_set[2] = {false, false};
  if (lds > 65536 - 1024 && !lds_set[use_slots]) {
    const void* fn = use_slots ? (const void*)decoder_step_slots_kernel : (const void*)decoder_step_kernel;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024)) != hipSuccess) {
      vacnic_set_error("decoder_step: cannot raise the dynamic LDS limit");
      return VACNIC_HIP_ERROR;
    }
    lds_set[use_slots] = true;
  }
  if (use_slots) {
    // one workgroup per 4 columns of a d-wide projection / 16 columns of the FFN: every launched workgroup produces in the last
    // layer's fc1 or fc2, both consumed by workgroup 0 before it bumps the launch nonce
    VCHECK((a->d & 15) == 0 && (a->F & 15) == 0 && a->L <= MAX_L, VACNIC_UNSUPPORTED, "decoder_step (slots): d, F multiples of 16, L <= 120");
    const int64_t G = a->d / 4 > a->F / 16 ? a->d / 4 : a->F / 16;
    VCHECK(G <= 256 && G <= n_cu && a->R * a->H <= 128 && a->R * a->H <= G, VACNIC_UNSUPPORTED,
           "decoder_step (slots): needs max(d / 4, ffn / 16) <= min(256, CUs) co-resident workgroups");
    VCHECK(aligned16(a->slots), VACNIC_MISALIGNED, "decoder_step: slots must be 16-byte aligned");
    hipLaunchKernelGGL(decoder_step_slots_kernel, dim3((unsigned)G), dim3(NTHR), lds, (hipStream_t)stream, p);
  } else {
    const int G = n_cu < 256 ? n_cu : 256;
    hipLaunchKernelGGL(decoder_step_kernel, dim3(G), dim3(NTHR), lds, (hipStream_t)stream, p);
  }
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
