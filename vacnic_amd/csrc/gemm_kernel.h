// gemm_kernel.h — the MFMA GEMM kernel template of libvacnic_hip.so (see gemm.hip for the design notes and the host-side
// dispatch).  Included by gemm.hip and by the gemm_t*.hip translation units, each of which instantiates ONE tile
// configuration (four operand layouts) so that the configurations compile in parallel.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace vacgemm {


constexpr int BK = 64;
constexpr int OOB = 0x7ffffff0;               // voffset beyond any (<2 GiB) buffer -> load returns 0

struct GemmP {
  const bf16_t* x; const bf16_t* w; const float* bias;
  void* out; bf16_t* preact; const bf16_t* dact_src; const bf16_t* residual;
  float* xsum;          // optional f32 [M]: xsum[m] += sum_k X(m,k)  (bias gradient fused into the weight-gradient GEMM)
  int M, N, K;
  int ldx, ldw, ldo;
  int act, out_mode, split_k, k_per_split;
  float alpha;
  unsigned x_bytes, w_bytes;
  int tiles_m, tiles_n;
  int ce_col0;          // out_mode 3 / 4: global vocabulary index of this launch's column 0 (targets are global ids)
  // activation dropout fused into the epilogue (after the activation / after act'): common.h "activation dropout"; thr 0 = off.
  // The output must be a contiguous [M, N] bf16 array with N % 16 == 0 (element index m * N + n).
  unsigned drop_thr; float drop_inv; unsigned long long drop_seed; const unsigned long long* drop_seed_dev;
  float* ws;            // split-K with ordered fix-up (see gemm_tile): fp32 partial tiles [tile][split][BM*BN]; nullptr = atomics
  unsigned* cnt;        // one arrival counter per output tile (zero before the launch, zero again after it)
  int debug;            // profiling aid (tile_hint >= 1000): bit0 skip the global stores, bit1 skip the K loop, bit2 skip the epilogue, bit3 return at once, bit4 fp32 staged epilogue, bit5 no LDS-DMA in the loop, bit7 no fragment reads
};

// f(k) of the K-strided swizzle: distinct for the 8 k-rows one tr-read half touches.
__device__ __forceinline__ int fk(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Issue this wave's LDS-DMA loads for one operand tile of ROWS rows x BKT k (BKT = 64 or 32).
//   KS=false, BKT=64: tile [ROWS][64 k], 128-B LDS rows, chunk' = chunk ^ (row & 7); one 1-KiB piece = 8 rows
//   KS=false, BKT=32: two 64-B global rows share one 128-B LDS row R = row/2 (chunk = (row&1)*4 + kchunk),
//                     chunk' = chunk ^ (R & 7); one piece = 16 rows
//   KS=true : tile [BKT k][ROWS], 2*ROWS-B rows, chunk' = chunk ^ swz(k); one piece = 512/ROWS k-rows
template <bool KS, int ROWS, int BKT, int NWAVE>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int r0, int R,
                                           int k0, int kend, int ld, int wave, int lane) {
  constexpr int PIECES = ROWS * BKT * 2 / 1024 / NWAVE;      // 1-KiB pieces per wave
  static_assert(PIECES >= 1, "tile too small for this wave count");
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int blk = wave * PIECES + i;
    int voff;
    if (!KS) {
      int row, kch;
      if (BKT == 64) {
        row = blk * 8 + (lane >> 3);
        kch = (lane & 7) ^ (row & 7);
      } else {
        const int Rl = blk * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ (Rl & 7);
        row = 2 * Rl + (lc >> 2);
        kch = lc & 3;
      }
      const int gr = r0 + row, gk = k0 + kch * 8;
      voff = (gr < R && gk < kend) ? (int)(((unsigned)gr * (unsigned)ld + (unsigned)gk) * 2u) : OOB;
    } else {
      constexpr int LPR = ROWS / 8;             // lanes (16-B chunks) per k-row
      const int k = blk * (64 / LPR) + lane / LPR;
      const int lc = (lane % LPR) ^ ((fk(k) << 1) & (LPR - 1));   // 64-row tiles have only 8 chunks per k-row
      const int gk = k0 + k, gr = r0 + lc * 8;
      voff = (gk < kend && gr < R) ? (int)(((unsigned)gk * (unsigned)ld + (unsigned)gr) * 2u) : OOB;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + blk * 1024), 16, voff, 0, 0, 0);
  }
}

// Fragment for MFMA 16x16x32: lane l gets element (row = rbase + (l&15), k = kk*32 + 8*(l>>4) + j), j=0..7.
template <bool KS, int ROWS, int BKT>
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int rbase, int kk, int lane) {
  if (!KS) {
    const int row = rbase + (lane & 15);
    if (BKT == 64) {
      const int chunk = kk * 4 + (lane >> 4);
      return *(const bf16x8*)(lds_tile + row * 128 + ((chunk ^ (row & 7)) << 4));
    } else {
      const int Rl = row >> 1;
      const int chunk = (row & 1) * 4 + (lane >> 4);
      return *(const bf16x8*)(lds_tile + Rl * 128 + ((chunk ^ (Rl & 7)) << 4));
    }
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int chunk = (rbase >> 3) + (p >> 1);
    bf16x8 r;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int k = kk * 32 + 8 * g + 4 * hh + q;
      const int phys = chunk ^ ((fk(k) << 1) & (ROWS / 8 - 1));
      const char* a = lds_tile + k * (ROWS * 2) + phys * 16 + (p & 1) * 8;
      bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)LDS_PTR(a));
      r[4 * hh + 0] = t[0]; r[4 * hh + 1] = t[1]; r[4 * hh + 2] = t[2]; r[4 * hh + 3] = t[3];
    }
    return r;
  }
}

// Epilogue on 8 consecutive outputs of one row (read back from the LDS-staged C tile): bias, saved
// pre-activation, activation or fused activation-backward, residual, then a 16-byte (bf16) /
// 2x16-byte (f32) store or 8 f32 atomics on 32 contiguous bytes.
__device__ __forceinline__ void load8bf(const bf16_t* p, float v[8]) {
  u32x4 r = *(const u32x4*)p;
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
}
__device__ __forceinline__ void store8bf(bf16_t* p, const float v[8]) {
  *(u32x4*)p = (u32x4){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
}

__device__ __forceinline__ void unpack8bf(u32x4 r, float v[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
}

// 8 consecutive, fully in-range, 16-byte-aligned outputs of one row.  Bias and the raw residual / activation-source words
// were loaded by the caller (batched over all of a thread's chunks, so their latency overlaps).
template <bool DR = false>
__device__ __forceinline__ void epilogue8_vec(const GemmP& p, float v[8], size_t off, const float bia[8], u32x4 rraw, u32x4 draw,
                                              uint32_t mlo = 0, uint32_t mhi = 0) {
  if (p.alpha != 1.0f) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] += bia[j];
  if (p.preact) store8bf(p.preact + off, v);
  if (p.dact_src) {
    float d[8];
    unpack8bf(draw, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= act_bwd(p.act, d[j]);
  } else if (p.act != VACNIC_ACT_NONE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = act_fwd(p.act, v[j]);
  }
  if constexpr (DR) {
    float m[8];
    actdrop_factors8(mlo, mhi, p.drop_thr, p.drop_inv, m);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= m[j];
  }
  if (p.residual) {
    float d[8];
    unpack8bf(rraw, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += d[j];
  }
  if (p.debug & 1) {
    if (v[0] == 12345.678f) ((bf16_t*)p.out)[off] = 0;
  } else if (p.out_mode == 0) {
    store8bf((bf16_t*)p.out + off, v);
  } else if (p.out_mode == 1) {
    float* o = (float*)p.out + off;
    *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
    *(f32x4*)(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
  } else {
    float* o = (float*)p.out + off;
    f32x4 a0 = *(f32x4*)o, a1 = *(f32x4*)(o + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { a0[j] += v[j]; a1[j] += v[4 + j]; }
    *(f32x4*)o = a0; *(f32x4*)(o + 4) = a1;
  }
}

__device__ __forceinline__ void epilogue8(const GemmP& p, float v[8], int m, int n, bool add_bias, bool vec_ok) {
  const size_t off = (size_t)m * p.ldo + n;
  const int nv = min(8, p.N - n);
  const bool vec = vec_ok && nv == 8;
  if (p.alpha != 1.0f) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
  }
  if (p.bias && add_bias) {
    if (nv == 8) {
      const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);   // arena slots are 16-byte aligned, n % 8 == 0
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
    } else {
      for (int j = 0; j < nv; ++j) v[j] += p.bias[n + j];
    }
  }
  if (p.preact) {
    if (vec) store8bf(p.preact + off, v);
    else for (int j = 0; j < nv; ++j) p.preact[off + j] = f2bf(v[j]);
  }
  if (p.dact_src) {
    float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (vec) load8bf(p.dact_src + off, d);
    else for (int j = 0; j < nv; ++j) d[j] = bf2f(p.dact_src[off + j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= act_bwd(p.act, d[j]);
  } else if (p.act != VACNIC_ACT_NONE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = act_fwd(p.act, v[j]);
  }
  if (p.residual) {
    float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (vec) load8bf(p.residual + off, d);
    else for (int j = 0; j < nv; ++j) d[j] = bf2f(p.residual[off + j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += d[j];
  }
  if (p.debug & 1) {
    if (v[0] == 12345.678f) ((bf16_t*)p.out)[off] = 0;      // keeps the values live, never true in practice
  } else if (p.out_mode == 0) {
    bf16_t* o = (bf16_t*)p.out + off;
    if (vec) store8bf(o, v);
    else for (int j = 0; j < nv; ++j) o[j] = f2bf(v[j]);
  } else if (p.out_mode == 1) {
    float* o = (float*)p.out + off;
    if (nv == 8 && (p.ldo & 3) == 0) {
      *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
      *(f32x4*)(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else {
      for (int j = 0; j < nv; ++j) o[j] = v[j];
    }
  } else {
    // accumulate, single K-split: no other workgroup of this launch touches these outputs -> vector read-modify-write
    float* o = (float*)p.out + off;
    if (nv == 8 && (p.ldo & 3) == 0) {
      f32x4 a0 = *(f32x4*)o, a1 = *(f32x4*)(o + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a0[j] += v[j]; a1[j] += v[4 + j]; }
      *(f32x4*)o = a0; *(f32x4*)(o + 4) = a1;
    } else {
      for (int j = 0; j < nv; ++j) o[j] += v[j];
    }
  }
}

template <int V> struct IC { static constexpr int value = V; };

// Workgroup barrier that orders LDS traffic only (this wave's LDS operations retired; global stores stay in flight)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}



template <int N>
__device__ __forceinline__ void wait_vm_lgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// DR: compile the activation-dropout epilogues (GemmP::drop_*) — their own instantiations (gemm_t*d.hip) for the same reason as CE:
// inside the general kernels the extra epilogue code cost the 128 x 128 tile 18 VGPRs (occupancy 3 -> 2) and the 256 x 256 tile
// scratch.  A DR kernel always applies dropout (the host launches it only with drop_thr != 0).
// CE: compile the fused LM-head cross-entropy epilogues (out_mode 3 / 4) — their own instantiation (gemm_t256ce.hip): inside the
// general kernels their code raised the register allocation of EVERY epilogue path (128x128 tiles: 181 -> 256 VGPRs + scratch).
// One output tile: rows [m0, m0 + BM) x columns [n0, n0 + BN), reduction slice `zsplit`.  `tn` = this tile's column index inside
// its row panel and p.tiles_n the number of tiles that share the panel (the xsum K-steps are dealt round-robin over them).
template <int BM, int BN, int WM, int WN, int BKT, int NSTAGE, bool PIPE, bool XKS, bool WKS, bool CE, bool DR = false>
__device__ __forceinline__ void gemm_tile(const GemmP& p, const int m0, const int n0, const int tn, const int zsplit, const int tile_id = 0) {
  constexpr int NWAVE = WM * WN, NTHR = 64 * NWAVE;
  constexpr int TM = BM / WM, TN = BN / WN;             // per-wave output sub-tile
  constexpr int FA = TM / 16, FB = TN / 16;             // MFMA tiles per wave along m / n
  constexpr int XT = BM * BKT * 2, WT = BN * BKT * 2, STAGE = XT + WT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int kbeg = zsplit * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int ntile = (p.debug & 2) ? 0 : (kend - kbeg + BKT - 1) / BKT;

  __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  // K-strided operands may read up to round_up(R, 8) columns of a row (host guarantees ld covers it)
  const int RX = XKS ? ((p.M + 7) & ~7) : p.M;
  const int RW = WKS ? ((p.N + 7) & ~7) : p.N;

  f32x4 acc[FB][FA];
#pragma unroll
  for (int b = 0; b < FB; ++b)
#pragma unroll
    for (int a = 0; a < FA; ++a) acc[b][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int XSN = FA / WN;                           // row-sum accumulators per wave (xsum)
  static_assert(FA % WN == 0 && FA / WN <= 4, "xsum: row blocks split over the waves of a row group, <= 4 each");
  // xsum[m] += sum_k X(m,k) (the bias gradient of a weight-gradient GEMM: X = dY; compiled into the both-K-strided layout
  // only).  One extra MFMA per 16 rows re-uses the X fragments that are in registers anyway: the A operand is a row
  // selector (row i all ones, the other rows zero), so the XSN row blocks a wave is responsible for land in rows 0..XSN-1
  // of ONE extra accumulator tile.  The K-steps are dealt round-robin to the tiles_n workgroups that share this row panel
  // and the FA row blocks to the WN waves that hold the same fragments, so the extra matrix work is 1/(WN*FB*tiles_n) of
  // the tile's — and the separate pass over dY is gone.
  constexpr bool XS = XKS && WKS;
  const bool xs_on = XS && p.xsum != nullptr;
  int xs_next = tn;
  f32x4 accs = (f32x4){0.f, 0.f, 0.f, 0.f};
#define VAC_XSUM_MFMA(XF)                                                                   \
  if constexpr (XS) {                                                                       \
    _Pragma("unroll") for (int w_ = 0; w_ < WN; ++w_)                                       \
      if (wn == w_) {                                                                       \
        _Pragma("unroll") for (int i = 0; i < XSN; ++i) {                                   \
          const short o_ = (lane & 15) == i ? (short)0x3F80 : (short)0;                              \
          const bf16x8 sel_ = {o_, o_, o_, o_, o_, o_, o_, o_};                             \
          accs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sel_, XF[w_ * XSN + i], accs, 0, 0, 0); \
        }                                                                                   \
      }                                                                                     \
  }
#define VAC_XSUM(XF, STEP)                                                                  \
  if (XS && xs_on && (STEP) == xs_next) {                                                   \
    VAC_XSUM_MFMA(XF)                                                                       \
    xs_next += p.tiles_n;                                                                   \
  }

  // LDS: NSTAGE-deep ring, stage s at smem + s*STAGE = {X tile, W tile}.  Tiles t+1 .. t+NSTAGE-1 are in flight
  // while tile t is multiplied; one barrier per K-tile.  Loads are issued unconditionally (a tile past kend is
  // all out-of-range -> zero fill, never read) so the counted vmcnt below is a compile-time constant.
  constexpr int LOADS = (BM + BN) * BKT * 2 / 1024 / NWAVE;   // LDS-DMA instructions per wave per K-tile
  static_assert(LOADS * (NSTAGE - 2) <= 63, "vmcnt immediate");
  constexpr bool PIPED = PIPE && BKT == 64 && NSTAGE == 2;
  constexpr int PRO = PIPED ? 2 : NSTAGE - 1;            // tiles staged before the loop
#pragma unroll
  for (int s = 0; s < PRO; ++s) {
    stage_tile<XKS, BM, BKT, NWAVE>(xs, smem + s * STAGE, m0, RX, kbeg + s * BKT, kend, p.ldx, wave, lane);
    stage_tile<WKS, BN, BKT, NWAVE>(ws, smem + s * STAGE + XT, n0, RW, kbeg + s * BKT, kend, p.ldw, wave, lane);
  }
  wait_vm_lgkm<LOADS * (PRO - 1)>();                     // tile 0 landed
  __builtin_amdgcn_s_barrier();
  int cur = 0, nxt = NSTAGE - 1;
  if constexpr (PIPE && BKT == 32 && NSTAGE == 4) {
    // Ping-pong K loop (8 waves, two per SIMD; 32-wide K stages in a 4-slot ring).  The waves of a workgroup form two
    // groups, A = waves 0..3 and B = waves 4..7 (SIMD partners), that run the same sequence one interval apart:
    //     A:  MEM(0) | COMP(0) | MEM(1) | COMP(1) | ...
    //     B:    -    | MEM(0)  | COMP(0)| MEM(1)  | ...          ('|' = workgroup barrier)
    // MEM(h)  = 12 ds_read_b128 (the fragments of stage h) + this wave's 4 LDS-DMA pieces of stage h+3 + counted wait,
    // COMP(h) = 32 back-to-back MFMAs.  In every interval one wave per SIMD owns the matrix pipe while its partner owns the
    // LDS / vector-memory issue ports, so neither the fragment reads nor the ~60-100-cycle issue cost of an LDS-DMA piece
    // ever stalls the MFMA stream (measured before: MFMA-only loop 1.0 us per 64-K, +0.27 us for the LDS reads, +0.34 us
    // for the DMA issue when both partners do the same thing at the same time).
    // Ring safety: stage j is read by A in interval 2j and by B in interval 2j+1; MEM(j+1) (intervals 2j+2 / 2j+3) refills
    // its slot with stage j+4.  A wave leaves MEM(h) only when its own pieces of stage h+1 have landed (vmcnt(8): stages
    // h+2, h+3 may fly), and a barrier separates that from every later reader.
    static_assert(NWAVE == 8 && LOADS >= 2, "ping-pong loop is written for 8 waves (two per SIMD)");
    // (Issuing part of the DMA pieces in the middle of COMP(h) instead, or staggering the partners' issue points in a 64-wide
    // pipelined loop, measured the same within noise: profiles/r1_gemm_overhead.txt.)
    const bool grp_b = wave >= NWAVE / 2;
    const int nst = ntile;
    bf16x8 xf[FA], wf[FB];
    if (grp_b) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    int rd = 0, wr = 3;
    for (int h = 0; h < nst; ++h) {
      const char* xr = smem + rd * STAGE;
      char* xw = smem + wr * STAGE;
      if (!(p.debug & 128) || h == 0) {
#pragma unroll
        for (int a = 0; a < FA; ++a) xf[a] = read_frag<XKS, BM, BKT>(xr, wm * TM + a * 16, 0, lane);
#pragma unroll
        for (int b = 0; b < FB; ++b) wf[b] = read_frag<WKS, BN, BKT>(xr + XT, wn * TN + b * 16, 0, lane);
      }
      if (!(p.debug & 32)) {
        stage_tile<XKS, BM, BKT, NWAVE>(xs, xw, m0, RX, kbeg + (h + 3) * BKT, kend, p.ldx, wave, lane);
        stage_tile<WKS, BN, BKT, NWAVE>(ws, xw + XT, n0, RW, kbeg + (h + 3) * BKT, kend, p.ldw, wave, lane);
      }
      // stage h+1 must have landed (this wave's pieces); newer ones may fly
      wait_vm_lgkm<2 * LOADS>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int a = 0; a < FA; ++a)
          acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[b][a], 0, 0, 0);
      VAC_XSUM(xf, h)
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      rd = (rd + 1) & 3;
      wr = (wr + 1) & 3;
    }
    if (!grp_b) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
  } else if constexpr (PIPED) {
    // Software-pipelined K loop, two LDS slots.  Per tile t:
    //   read frags (t, kk=1) | MFMA (t, kk=0) | wait tile t+1 landed + BARRIER | issue LDS-DMA of tile t+2 into slot t |
    //   read frags (t+1, kk=0) | MFMA (t, kk=1)
    // - the barrier sits in the middle of the tile's MFMA work: after it the LDS reads of the next tile (and the barrier
    //   skew of the 8 waves) are covered by the 32..64 MFMAs of (t, kk=1);
    // - at the barrier every fragment of tile t is already in registers, so slot t is free: the loads of tile t+2 are
    //   issued right behind it and have a FULL iteration to land (a 64 KiB tile needs ~0.9 us at the per-CU L2->LDS rate
    //   plus latency; a load issued half an iteration before its wait stalls the whole workgroup);
    bf16x8 xf0[FA], wf0[FB], xf1[FA], wf1[FB];
#pragma unroll
    for (int a = 0; a < FA; ++a) xf0[a] = read_frag<XKS, BM, BKT>(smem, wm * TM + a * 16, 0, lane);
#pragma unroll
    for (int b = 0; b < FB; ++b) wf0[b] = read_frag<WKS, BN, BKT>(smem + XT, wn * TN + b * 16, 0, lane);
    for (int t = 0; t < ntile; ++t) {
      char* xcur = smem + cur * STAGE;
      char* wcur = xcur + XT;
#pragma unroll
      for (int a = 0; a < FA; ++a) xf1[a] = read_frag<XKS, BM, BKT>(xcur, wm * TM + a * 16, 1, lane);
#pragma unroll
      for (int b = 0; b < FB; ++b) wf1[b] = read_frag<WKS, BN, BKT>(wcur, wn * TN + b * 16, 1, lane);
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int a = 0; a < FA; ++a)
          acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[b], xf0[a], acc[b][a], 0, 0, 0);
      if (XS && xs_on && t == xs_next) { VAC_XSUM_MFMA(xf0) }
      // tile t+1 landed (all of this wave's loads), every LDS read of slot t retired
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stage_tile<XKS, BM, BKT, NWAVE>(xs, xcur, m0, RX, kbeg + (t + 2) * BKT, kend, p.ldx, wave, lane);
      stage_tile<WKS, BN, BKT, NWAVE>(ws, wcur, n0, RW, kbeg + (t + 2) * BKT, kend, p.ldw, wave, lane);
      cur ^= 1;
      {
        const char* xn = smem + cur * STAGE;           // tile t+1 (zero-filled past the end: harmless)
#pragma unroll
        for (int a = 0; a < FA; ++a) xf0[a] = read_frag<XKS, BM, BKT>(xn, wm * TM + a * 16, 0, lane);
#pragma unroll
        for (int b = 0; b < FB; ++b) wf0[b] = read_frag<WKS, BN, BKT>(xn + XT, wn * TN + b * 16, 0, lane);
      }
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int a = 0; a < FA; ++a)
          acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[b], xf1[a], acc[b][a], 0, 0, 0);
      VAC_XSUM(xf1, t)
    }
  } else {
  bf16x8 xf[FA], wf[FB];
  for (int t = 0; t < ntile; ++t) {
    char* xcur = smem + cur * STAGE;
    char* wcur = xcur + XT;
    if (!(p.debug & 32)) {
      // ring slot `nxt` was last read in iteration t-1 and every wave has passed that iteration's barrier
      char* xnext = smem + nxt * STAGE;
      stage_tile<XKS, BM, BKT, NWAVE>(xs, xnext, m0, RX, kbeg + (t + NSTAGE - 1) * BKT, kend, p.ldx, wave, lane);
      stage_tile<WKS, BN, BKT, NWAVE>(ws, xnext + XT, n0, RW, kbeg + (t + NSTAGE - 1) * BKT, kend, p.ldw, wave, lane);
    }
    if (!(p.debug & 64))
#pragma unroll
    for (int kk = 0; kk < BKT / 32; ++kk) {
      if (!(p.debug & 128) || t == 0) {
#pragma unroll
        for (int a = 0; a < FA; ++a) xf[a] = read_frag<XKS, BM, BKT>(xcur, wm * TM + a * 16, kk, lane);
#pragma unroll
        for (int b = 0; b < FB; ++b) wf[b] = read_frag<WKS, BN, BKT>(wcur, wn * TN + b * 16, kk, lane);
      }
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int a = 0; a < FA; ++a)
          acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[b][a], 0, 0, 0);
      if (XS && xs_on && t == xs_next) { VAC_XSUM_MFMA(xf) }
    }
    if (XS && xs_on && t == xs_next) xs_next += p.tiles_n;
    // tile t+1 landed (this wave's loads; newer tiles may still fly), LDS reads of this slot retired; then everyone's
    wait_vm_lgkm<LOADS * (NSTAGE - 2)>();
    __builtin_amdgcn_s_barrier();
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the (zero-fill) tail loads before LDS is reused
  __builtin_amdgcn_s_barrier();
#undef VAC_XSUM
#undef VAC_XSUM_MFMA
  if (XS && xs_on && lane < 16) {
    // C^T layout: lanes 0..15 hold result rows 0..3 (element j = row), column m = lane
#pragma unroll
    for (int i = 0; i < XSN; ++i) {
      const int m = m0 + wm * TM + (wn * XSN + i) * 16 + lane;
      if (m < p.M) atomicAdd(p.xsum + m, accs[i]);
    }
  }

  if (p.debug & 4) { if (acc[0][0][0] == 12345.678f) ((float*)p.out)[0] = 0.f; return; }
  // ---- split-K with an ORDERED FIX-UP instead of atomics (p.ws != nullptr): every K-slice workgroup of a tile deposits its fp32
  // partial tile in the workspace and takes a ticket; the LAST one to arrive sums all slices in slice order 0 .. S-1 — a fixed
  // order whoever comes last, so the result is bitwise reproducible — and runs the ordinary epilogue ONCE (bias, activation,
  // residual, bf16 or fp32 output, accumulate as a plain read-modify-write).  Nobody waits: the other workgroups leave after their
  // deposit.  The partials cross XCDs, so they travel as sc1 stores / loads (write-through, miss-always in the XCD's L2; complete
  // when vmcnt says so — the decoder-step kernel's hand-over, decstep.hip) and the ticket is a relaxed agent-scope atomic: no
  // cache write-back / invalidate (20-35 us apiece on this part).  Layout: a thread's own accumulator registers, 16 bytes per
  // lane per store — fully coalesced, and the summing workgroup's threads read back exactly the elements they own.
  const bool fix = p.ws != nullptr && p.split_k > 1;
  if (fix) {
    constexpr int NACC = FA * FB, COH = 16;
    constexpr unsigned TILE_BYTES = (unsigned)BM * BN * 4u;
    float* wtile = p.ws + (size_t)tile_id * p.split_k * ((size_t)BM * BN);
    {
      __amdgpu_buffer_rsrc_t ps = __builtin_amdgcn_make_buffer_rsrc((void*)(wtile + (size_t)zsplit * ((size_t)BM * BN)), 0, TILE_BYTES, 0x00020000);
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int a = 0; a < FA; ++a)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[b][a]), ps, (unsigned)(((b * FA + a) * NTHR + tid) * 16), 0, COH);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this thread's deposit is in memory
    __syncthreads();                                     // ... and everybody's
    int* s_ticket = (int*)smem;
    if (tid == 0) *s_ticket = (int)__hip_atomic_fetch_add(p.cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = *s_ticket;
    if (ticket != p.split_k - 1) return;
    if (tid == 0) __hip_atomic_store(p.cnt + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every slice has arrived: clean for the next launch
    constexpr int BATCH = NACC < 8 ? NACC : 8;
#pragma unroll 1
    for (int z = 0; z < p.split_k; ++z) {
      __amdgpu_buffer_rsrc_t ps = __builtin_amdgcn_make_buffer_rsrc((void*)(wtile + (size_t)z * ((size_t)BM * BN)), 0, TILE_BYTES, 0x00020000);
#pragma unroll
      for (int i0 = 0; i0 < NACC; i0 += BATCH) {
        u32x4 v[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(ps, (unsigned)(((i0 + i) * NTHR + tid) * 16), 0, COH);
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
          const f32x4 f = __builtin_bit_cast(f32x4, v[i]);
          const int b = (i0 + i) / FA, a = (i0 + i) % FA;
          if (z == 0) acc[b][a] = f; else acc[b][a] += f;
        }
      }
    }
    __syncthreads();                                     // the ticket word's readers are done before the epilogue reuses the LDS
  }
  const bool split_atomic = p.split_k > 1 && !fix;        // K slices meet in fp32 atomics on the output
  // ---- direct epilogue for the common plain case (bf16 out, bias + activation only): each lane owns 4 consecutive n of
  // one m per accumulator tile -> bias as one 16-byte load, pack with v_cvt_pk_bf16_f32, one 8-byte store.  No LDS
  // round trip, no barriers; the 32-byte row pieces of the four n-groups are merged by the L2.
  if (BM <= 128 && p.out_mode == 0 && !p.preact && !p.dact_src && !p.residual && (p.ldo & 3) == 0 && !(p.debug & 16) && !DR) {
    const int lm_ = lane & 15, ln4_ = (lane >> 4) * 4;
    const bool add_bias_ = p.bias != nullptr;
#pragma unroll
    for (int a = 0; a < FA; ++a) {
      const int m = m0 + wm * TM + a * 16 + lm_;
      if (m >= p.M) continue;
      bf16_t* orow = (bf16_t*)p.out + (size_t)m * p.ldo;
#pragma unroll
      for (int b = 0; b < FB; ++b) {
        const int n = n0 + wn * TN + b * 16 + ln4_;
        if (n + 4 <= p.N) {
          f32x4 v = acc[b][a];
          if (p.alpha != 1.0f) v *= p.alpha;
          if (add_bias_) v += *(const f32x4*)(p.bias + n);
          if (p.act != VACNIC_ACT_NONE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act_fwd(p.act, v[j]);
          }
          *(u32x2*)(orow + n) = (u32x2){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        } else if (n < p.N) {
          for (int j = 0; j < 4 && n + j < p.N; ++j) {
            float v = acc[b][a][j] * p.alpha + (add_bias_ ? p.bias[n + j] : 0.f);
            orow[n + j] = f2bf(act_fwd(p.act, v));
          }
        }
      }
    }
    return;
  }
  // bf16 epilogue geometry (256-row tiles): 16-byte chunks per staged row, rows per sweep of the workgroup, sweeps (= 16-byte
  // global stores per thread) per tile
  constexpr int ROWB = BN * 2, NCH = ROWB / 16, RSTEP = NTHR / NCH, NIT = BM / RSTEP;
  const int lm = lane & 15, ln4 = (lane >> 4) * 4;
  // ---- bf16 epilogue of the 256-row tiles (bf16 output, 16-byte output rows): bias (+ activation when nothing else
  // needs the pre-activation) on the fp32 accumulators in registers, round ONCE to bf16 and transpose the whole C tile
  // through LDS in a single pass (BM x BN x 2 B <= 128 KiB, XOR-swizzled 16-byte chunks: conflict-free 8-byte writes
  // from the C^T fragments, conflict-free 16-byte row reads), so that every global access is a 16-byte row-contiguous
  // one.  One barrier per tile instead of four fp32 passes.  Saved
  // pre-activation, fused activation-backward and residual are applied on the bf16 value after the transposition — the
  // arithmetic of a bf16 autocast Linear followed by a bf16 elementwise op.
  if constexpr (BM == 256 && BN >= 128) {
    if (p.out_mode == 0 && (p.ldo & 7) == 0 && (p.N & 7) == 0 && !(p.debug & 16)) {
      static_assert(BM * ROWB <= NSTAGE * STAGE, "bf16 C tile must fit in the operand ring");
      const bool act_in_regs = p.act != VACNIC_ACT_NONE && !p.preact && !p.dact_src;
      f32x4 bia[FB];
#pragma unroll
      for (int b = 0; b < FB; ++b) {
        const int n = n0 + wn * TN + b * 16 + ln4;
        bia[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
          if (n + 4 <= p.N) bia[b] = *(const f32x4*)(p.bias + n);
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (n + j < p.N) bia[b][j] = p.bias[n + j];
          }
        }
      }
      auto deposit = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int a = 0; a < FA; ++a) {
          const int row = wm * TM + a * 16 + lm;
          char* rp = smem + row * ROWB;
#pragma unroll
          for (int b = 0; b < FB; ++b) {
            f32x4 v = acc[b][a];
            if (p.alpha != 1.0f) v *= p.alpha;
            v += bia[b];
            if (ACT != VACNIC_ACT_NONE) {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = act_fwd(ACT, v[j]);
            }
            const int nl = wn * TN + b * 16 + ln4;
            *(u32x2*)(rp + ((((nl >> 3) ^ (row & 15)) << 4) | (((nl >> 2) & 1) << 3))) = (u32x2){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          }
        }
      };
      const int act_regs = act_in_regs ? p.act : VACNIC_ACT_NONE;
      if (act_regs == VACNIC_ACT_GELU) deposit(IC<VACNIC_ACT_GELU>{});
      else if (act_regs == VACNIC_ACT_TANH) deposit(IC<VACNIC_ACT_TANH>{});
      else if (act_regs == VACNIC_ACT_QUICKGELU) deposit(IC<VACNIC_ACT_QUICKGELU>{});
      else deposit(IC<VACNIC_ACT_NONE>{});
      lds_barrier();
      static_assert(NTHR % NCH == 0 && BM % RSTEP == 0 && NIT % 8 == 0, "epilogue sweep mapping");
      const int c = tid % NCH, rb = tid / NCH;
      u32x4 cv[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int row = rb + k * RSTEP;
        cv[k] = *(const u32x4*)(smem + row * ROWB + ((c ^ (row & 15)) << 4));
      }
      // Global traffic of the epilogue through buffer instructions: rows >= M (and, with debug bit 0, everything) fall
      // outside the descriptor's range — loads return 0, stores are dropped — so there is no branch around any access
      // (a branch per load makes hipcc wait for each load separately) and every wave issues exactly NIT stores per output.
      const unsigned span = (unsigned)((((size_t)p.M - 1) * p.ldo + p.N) * 2);
      const int n = n0 + c * 8;
      const unsigned rstride = (unsigned)(RSTEP * p.ldo * 2);
      unsigned voff = (n + 8 <= p.N && !(p.debug & 1)) ? (unsigned)(((size_t)(m0 + rb) * p.ldo + n) * 2) : (unsigned)OOB;
      const unsigned vstep = (n + 8 <= p.N && !(p.debug & 1)) ? rstride : 0u;
      __amdgpu_buffer_rsrc_t so = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, span, 0x00020000);
      const bool plain = !p.preact && !p.dact_src && !p.residual;
      // activation dropout (host: contiguous output, N % 16 == 0, never on the plain path): Philox block of this thread's column
      // group in row m0 + rb, and the block stride between two of its rows
      unsigned long long dseed = 0, dblk_step = 0, dblk0 = 0;
      if constexpr (DR) {
        dseed = p.drop_seed_dev ? p.drop_seed ^ (*p.drop_seed_dev * 0x9E3779B97F4A7C15ull) : p.drop_seed;
        dblk_step = (unsigned long long)RSTEP * (unsigned)(p.N >> 4);
        dblk0 = (unsigned long long)(m0 + rb) * (unsigned)(p.N >> 4) + (unsigned)(n >> 4);
      }
      // exactly one of residual / preact / dact_src may ride on this path (the host sends combinations to the fp32 path);
      // each gets its own straight-line instance: MODE 1 residual, 2 saved pre-activation (+ activation), 3 activation backward
      auto tail = [&](auto mode_c, auto act_c) {
        constexpr int MODE = decltype(mode_c)::value, ACT = decltype(act_c)::value;
        const void* eptr = MODE == 1 ? (const void*)p.residual : MODE == 2 ? (const void*)p.preact : (const void*)p.dact_src;
        __amdgpu_buffer_rsrc_t se = __builtin_amdgcn_make_buffer_rsrc((void*)eptr, 0, span, 0x00020000);
#pragma unroll
        for (int k0 = 0; k0 < NIT; k0 += 8) {
          u32x4 ex[8];
          uint32_t mlo[8], mhi[8];
          __builtin_amdgcn_sched_barrier(0);          // one batch of 8 rows at a time (the C tile already holds 64 registers)
          if (MODE != 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) ex[k] = __builtin_amdgcn_raw_buffer_load_b128(se, voff + (k0 + k) * vstep, 0, 0);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float v[8], d[8];
            if constexpr (DR && MODE != 1) {            // activation dropout: masks of a row pair from one Philox block per lane
              if ((k & 1) == 0) {
                const unsigned long long ba = dblk0 + (unsigned long long)(k0 + k) * dblk_step;
                actdrop_pair(dseed, ba, ba + dblk_step, lane, mlo[k], mhi[k], mlo[k + 1], mhi[k + 1]);
              }
            }
            unpack8bf(cv[k0 + k], v);
            if (MODE == 2) {
              __builtin_amdgcn_raw_buffer_store_b128(cv[k0 + k], se, voff + (k0 + k) * vstep, 0, 0);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = act_fwd(ACT, v[j]);
            } else {
              unpack8bf(ex[k], d);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = MODE == 1 ? v[j] + d[j] : v[j] * act_bwd(ACT, d[j]);
            }
            if constexpr (DR && MODE != 1) {
              float m[8];
              actdrop_factors8(mlo[k], mhi[k], p.drop_thr, p.drop_inv, m);
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] *= m[j];
            }
            const u32x4 o = (u32x4){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
            __builtin_amdgcn_raw_buffer_store_b128(o, so, voff + (k0 + k) * vstep, 0, 0);
          }
        }
      };
      if (plain) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) __builtin_amdgcn_raw_buffer_store_b128(cv[k], so, voff + k * vstep, 0, 0);
      } else if (p.residual) {
        tail(IC<1>{}, IC<VACNIC_ACT_NONE>{});
      } else if (p.preact) {
        if (p.act == VACNIC_ACT_GELU) tail(IC<2>{}, IC<VACNIC_ACT_GELU>{});
        else if (p.act == VACNIC_ACT_TANH) tail(IC<2>{}, IC<VACNIC_ACT_TANH>{});
        else if (p.act == VACNIC_ACT_QUICKGELU) tail(IC<2>{}, IC<VACNIC_ACT_QUICKGELU>{});
        else tail(IC<2>{}, IC<VACNIC_ACT_NONE>{});
      } else {
        if (p.act == VACNIC_ACT_GELU) tail(IC<3>{}, IC<VACNIC_ACT_GELU>{});
        else if (p.act == VACNIC_ACT_TANH) tail(IC<3>{}, IC<VACNIC_ACT_TANH>{});
        else if (p.act == VACNIC_ACT_QUICKGELU) tail(IC<3>{}, IC<VACNIC_ACT_QUICKGELU>{});
        else tail(IC<3>{}, IC<VACNIC_ACT_NONE>{});
      }
      return;
    }
  }
  // ---- epilogue: stage the fp32 C tile through LDS (the operand buffers are free now) in 64-row
  // passes, then every thread handles 8 consecutive n of one row: 16-byte coalesced traffic for the
  // output, the saved pre-activation, the activation-backward source and the residual.
  float* sc = (float*)smem;                 // [64][CLD] f32
  constexpr int CLD = BN + 4;
  constexpr int PASSES = (BM + 63) / 64;
  const bool vec_ok = (p.ldo & 7) == 0;
  const bool add_bias = zsplit == 0 || fix;
  float bia[8];
  // A pass stages 64 tile rows: RPWM = 64/WM rows from EACH wave row-group, so that every wave deposits in every pass (the
  // LDS store path has two halves, SIMDs {0,1} and {2,3}; a pass fed by the waves of one wm only ran it at half rate).
  // LDS row r of pass p holds tile row (r / RPWM) * TM + p * RPWM + r % RPWM.
  constexpr int RPWM = 64 / WM;
  static_assert((RPWM % 16 == 0 && TM % RPWM == 0) || BM < 64, "epilogue pass mapping");
  const bool old_map = (p.debug & 256) != 0;      // A/B: one wave row-group per pass (the previous mapping)
  auto tile_row = [&](int pass, int r) { return old_map ? pass * 64 + r : (r / RPWM) * TM + pass * RPWM + r % RPWM; };
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
#pragma unroll
    for (int a = 0; a < FA; ++a) {
      if (!old_map && (a * 16) / RPWM == pass) {
#pragma unroll
        for (int b = 0; b < FB; ++b)
          *(f32x4*)(sc + (wm * RPWM + (a * 16) % RPWM + lm) * CLD + wn * TN + b * 16 + ln4) = acc[b][a];
      }
      if (old_map && (wm * TM + a * 16) / 64 == pass) {
#pragma unroll
        for (int b = 0; b < FB; ++b)
          *(f32x4*)(sc + (((wm * TM + a * 16) & 63) + lm) * CLD + wn * TN + b * 16 + ln4) = acc[b][a];
      }
    }
    __syncthreads();
    if constexpr (CE) {
    if (p.out_mode >= 3) {
      // ---- fused LM-head cross-entropy epilogues (MFULL:1997 + TRAIN:287): the logits tile never leaves the chip.
      //   out_mode 3 (forward): per row of this tile, the online-softmax pair {max, sum exp(. - max)} over the tile's valid
      //                columns -> ce_part[m][tn]; the target column's logit -> ce_tl[m] (by the one tile that holds it)
      //   out_mode 4 (backward, logits recomputed): dlogit = (exp(logit - lse[m]) - [n == target[m]]) * coef[m] -> bf16 out
      // (targets: p.dact_src as int64; forward: part = p.preact, tl = p.xsum; backward: {lse, coef} pairs = p.residual.)
      constexpr int TPR = NTHR / 64, CPT = BN / TPR;          // threads per staged row, columns per thread
      static_assert(CPT % 8 == 0, "column groups of 8");
      const int row = tid / TPR, seg = tid % TPR;
      const int m = m0 + tile_row(pass, row);
      const bool mok = m < p.M;
      const int nb = n0 + seg * CPT;
      const float* src = sc + row * CLD + seg * CPT;
      const long long tgt = mok ? ((const long long*)p.dact_src)[m] - p.ce_col0 : -1;      // column inside this launch's N range
      if (p.out_mode == 3) {
        float mx = -INFINITY, sm = 0.f;
#pragma unroll
        for (int j0 = 0; j0 < CPT; j0 += 4) {
          const f32x4 v4 = *(const f32x4*)(src + j0);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int n = nb + j0 + j;
            if (n < p.N) {
              float v = v4[j] * p.alpha;
              if (p.bias) v += p.bias[n];
              if (n == tgt) p.xsum[m] = v;
              const float nm = fmaxf(mx, v);
              sm = sm * __expf(mx - nm) + __expf(v - nm);
              mx = nm;
            }
          }
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) {
          const float omx = __shfl_xor(mx, o, 64), osm = __shfl_xor(sm, o, 64);
          const float nm = fmaxf(mx, omx);
          sm = (mx == -INFINITY ? 0.f : sm * __expf(mx - nm)) + (omx == -INFINITY ? 0.f : osm * __expf(omx - nm));
          mx = nm;
        }
        if (seg == 0 && mok) {
          float* part = (float*)p.preact + ((size_t)m * p.tiles_n + tn) * 2;
          part[0] = mx; part[1] = sm;
        }
      } else if (mok) {
        const float lse = ((const float*)p.residual)[2 * m], coef = ((const float*)p.residual)[2 * m + 1];
        bf16_t* orow = (bf16_t*)p.out + (size_t)m * p.ldo;
#pragma unroll
        for (int j0 = 0; j0 < CPT; j0 += 8) {
          const int n = nb + j0;
          if (n >= ((p.N + 7) & ~7)) continue;
          const f32x4 a4 = *(const f32x4*)(src + j0), b4 = *(const f32x4*)(src + j0 + 4);
          float d[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float v = (j < 4 ? a4[j & 3] : b4[j & 3]) * p.alpha;
            if (p.bias && n + j < p.N) v += p.bias[n + j];
            d[j] = n + j < p.N ? (__expf(v - lse) - (n + j == tgt ? 1.f : 0.f)) * coef : 0.f;     // pad columns: exact zeros
          }
          store8bf(orow + n, d);                            // ldo % 8 == 0 and 16-byte rows are checked on the host
        }
      }
      __syncthreads();
      continue;
    }
    }
    if (p.out_mode == 2 && split_atomic) {
      // split-K accumulate: f32 atomics shaped as 256 contiguous bytes per wave-instruction (one row, 64
      // consecutive columns) — the shape the memory-side atomic units run at full rate on
      constexpr int RPW = 64 / NWAVE;
#pragma unroll 1
      for (int rr = 0; rr < RPW; ++rr) {
        const int row = wave * RPW + rr;
        const int m = m0 + tile_row(pass, row);
        if (m >= p.M) continue;
#pragma unroll
        for (int h = 0; h < BN / 64; ++h) {
          const int n = n0 + h * 64 + lane;
          if (n < p.N) {
            float v = sc[row * CLD + h * 64 + lane] * p.alpha;
            if (p.bias && add_bias) v += p.bias[n];
            atomicAdd((float*)p.out + (size_t)m * p.ldo + n, v);
          }
        }
      }
      __syncthreads();
      continue;
    }
    // every thread owns ONE 8-wide column group (NTHR is a multiple of the chunks per row) and RPT rows of the pass: the LDS
    // reads and the residual / activation-source loads of all RPT chunks are issued before any of them is consumed, and the
    // bias is loaded once per kernel — a chunk-at-a-time loop exposed one L2 round trip per chunk (7-9 us per tile).
    constexpr int CPR = BN / 8;               // 8-wide chunks per row
    constexpr int RPT = 64 * CPR / NTHR;      // chunks per thread per pass
    constexpr int RSTEP = NTHR / CPR;
    static_assert((64 * CPR) % NTHR == 0 && NTHR % CPR == 0, "epilogue chunk mapping");
    const int c8 = (tid % CPR) * 8, rbase = tid / CPR;
    const int n = n0 + c8;
    const bool fast = vec_ok && n + 8 <= p.N && (p.out_mode == 0 || (p.ldo & 3) == 0);
    if (fast) {
      if (pass == 0) {
        if (p.bias && add_bias) {
          const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { bia[j] = b0[j]; bia[4 + j] = b1[j]; }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) bia[j] = 0.f;
        }
      }
      float v[RPT][8];
      u32x4 rraw[RPT], draw[RPT];
      uint32_t mlo[RPT], mhi[RPT];
      if constexpr (DR) {                               // activation dropout masks of this thread's rows (one Philox block per row pair)
        static_assert(RPT % 2 == 0, "row pairs");
        const unsigned long long dseed = p.drop_seed_dev ? p.drop_seed ^ (*p.drop_seed_dev * 0x9E3779B97F4A7C15ull) : p.drop_seed;
#pragma unroll
        for (int k = 0; k < RPT; k += 2) {
          const unsigned long long ba = (unsigned long long)(m0 + tile_row(pass, rbase + k * RSTEP)) * (unsigned)(p.N >> 4) + (unsigned)(n >> 4);
          const unsigned long long bb = (unsigned long long)(m0 + tile_row(pass, rbase + (k + 1) * RSTEP)) * (unsigned)(p.N >> 4) + (unsigned)(n >> 4);
          actdrop_pair(dseed, ba, bb, lane, mlo[k], mhi[k], mlo[k + 1], mhi[k + 1]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < RPT; ++k) { mlo[k] = 0; mhi[k] = 0; }
      }
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int row = rbase + k * RSTEP;
        const int m = m0 + tile_row(pass, row);
        const f32x4 v0 = *(const f32x4*)(sc + row * CLD + c8), v1 = *(const f32x4*)(sc + row * CLD + c8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[k][j] = v0[j]; v[k][4 + j] = v1[j]; }
        rraw[k] = (u32x4){0, 0, 0, 0}; draw[k] = (u32x4){0, 0, 0, 0};
        if (m < p.M) {
          const size_t off = (size_t)m * p.ldo + n;
          if (p.residual) rraw[k] = *(const u32x4*)(p.residual + off);
          if (p.dact_src) draw[k] = *(const u32x4*)(p.dact_src + off);
        }
      }
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int m = m0 + tile_row(pass, rbase + k * RSTEP);
        if (m < p.M) epilogue8_vec<DR>(p, v[k], (size_t)m * p.ldo + n, bia, rraw[k], draw[k], mlo[k], mhi[k]);
      }
    } else if (n < p.N) {
#pragma unroll 1
      for (int k = 0; k < RPT; ++k) {
        const int row = rbase + k * RSTEP;
        const int m = m0 + tile_row(pass, row);
        if (m < p.M) {
          float v[8];
          const f32x4 v0 = *(const f32x4*)(sc + row * CLD + c8), v1 = *(const f32x4*)(sc + row * CLD + c8 + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
          epilogue8(p, v, m, n, add_bias, vec_ok);
        }
      }
    }
    __syncthreads();
  }
}


template <int BM, int BN, int WM, int WN, int BKT, int NSTAGE, bool PIPE, bool XKS, bool WKS, bool CE = false, bool DR = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_kernel(GemmP p) {
  if (p.debug & 8) return;
  // XCD-aware work order.  Workgroup L of the 1-D grid runs on XCD L % 8, each with its own 4 MiB L2.
  //  * split-K launches (weight gradients): split z = L % nsplit, so one XCD (or nsplit/8 .. 8/nsplit of them) owns a whole
  //    K-slice and every row of dY / X in it is fetched once — all tiles of a slice run concurrently on that XCD
  //    (+2 % on the wgrad GEMMs; with the splits in blockIdx.z every XCD touched every K-slice).
  //  * otherwise each XCD gets a contiguous run of tiles (bijective for any tile count), n fastest so neighbours reuse the
  //    same X panel in their L2.  (Walking 4-column strips inside a run — an 8 x 4 block of tiles in flight instead of
  //    2 x 16 — measured no gain: the 256 MiB memory-side cache already absorbs the W re-reads.)
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x, zsplit = 0;
  if (p.split_k > 1) {
    zsplit = bid % p.split_k;
    bid = bid / p.split_k;
  } else {
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  gemm_tile<BM, BN, WM, WN, BKT, NSTAGE, PIPE, XKS, WKS, CE, DR>(p, tm * BM, tn * BN, tn, zsplit, bid);
}

// ---------------------------------------------------------------------------------------------------------------------
// Grouped weight gradients: several independent dW[N,K] += dY[M,N]^T X[M,K] problems in ONE launch, no split-K.
// A UNIT is a block of up to UT x UT output tiles of one problem with the FULL reduction; unit u runs on XCD u % 8 (workgroup L
// runs on XCD L % 8 and the dispatcher hands them out in order), whose 32 CUs hold all UT*UT = 64 tiles at two workgroups per
// CU.  The tiles of a unit march through the reduction together, so every dY / X row block is fetched from HBM once and
// re-read by its 8 consumers from that XCD's L2 — the traffic of the K-slice-per-XCD split-K mapping without its fp32
// atomics (32 MB of memory-side atomics for a 1024 x 1024 gradient at split 8: ~25 of the launch's 60 us), and the sums are
// bitwise reproducible: every output element has ONE writer and a fixed summation order.
// Optionally (VACNIC_WGRAD_PHASE_ROWS) a long reduction is cut into PHASES, one launch each over the same grid (dW read-modify-
// written once per phase): shorter-lived workgroups.  Measured, it does not change how the encoder-sized groups delay the compute
// stream's chain (67.4 vs 67.4 ms/step at 4096 rows, 68.5 at 2048: profiles/r3_step_ab_wgrad.txt), so it is off by default; what
// the step time wants is decided in ops._groupable (only reductions of <= 4096 rows are grouped).
constexpr int GROUP_UT = 8;                 // tiles per unit side (128-wide tiles: 1024 x 1024 outputs per unit)
constexpr int GROUP_MAX_UNITS = 16;
struct GroupUnit {
  const bf16_t* x; const bf16_t* w; float* out; float* xsum;
  int M, N, K, ldx, ldw, ldo;
  int um0, un0;                             // first output row / column of the unit
  unsigned x_bytes, w_bytes;
};
struct GroupP { GroupUnit u[GROUP_MAX_UNITS]; int nunits; int debug; int kphase; int k_per_phase; };   // this launch reduces rows [kphase * k_per_phase, +k_per_phase) of every unit

template <int BM, int BN, int WM, int WN, int BKT, int NSTAGE, bool PIPE>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_group_kernel(GroupP g) {
  constexpr int TPU = GROUP_UT * GROUP_UT;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int unit = (j / TPU) * 8 + xcd;
  if (unit >= g.nunits) return;
  const int t = j % TPU;
  const int tm = t / GROUP_UT, tn = t % GROUP_UT;
  const GroupUnit& u = g.u[unit];
  const int m0 = u.um0 + tm * BM, n0 = u.un0 + tn * BN;
  if (m0 >= u.M || n0 >= u.N || g.kphase * g.k_per_phase >= u.K) return;
  GemmP p;
  p.x = u.x; p.w = u.w; p.bias = nullptr; p.out = u.out; p.preact = nullptr; p.dact_src = nullptr; p.residual = nullptr;
  p.xsum = u.xsum;
  p.M = u.M; p.N = u.N; p.K = u.K; p.ldx = u.ldx; p.ldw = u.ldw; p.ldo = u.ldo;
  p.act = VACNIC_ACT_NONE; p.out_mode = 2; p.split_k = 1; p.k_per_split = g.k_per_phase;
  p.alpha = 1.0f; p.x_bytes = u.x_bytes; p.w_bytes = u.w_bytes;
  p.tiles_m = GROUP_UT; p.tiles_n = min(GROUP_UT, (u.N - u.un0 + BN - 1) / BN);
  p.ce_col0 = 0; p.debug = g.debug; p.ws = nullptr; p.cnt = nullptr; p.drop_thr = 0;
  gemm_tile<BM, BN, WM, WN, BKT, NSTAGE, PIPE, true, true, false>(p, m0, n0, tn, g.kphase);
}

template <int BM, int BN, int WM, int WN, int BKT, int NSTAGE, bool PIPE>
int launch_gemm_group(const GroupP& g, hipStream_t s) {
  static_assert(BM == 128 && BN == 128, "units are GROUP_UT x GROUP_UT tiles of 128 x 128");
  const int rounds = (g.nunits + 7) / 8;
  dim3 grid(rounds * 8 * GROUP_UT * GROUP_UT), block(64 * WM * WN);
  constexpr size_t lds = NSTAGE * (BM + BN) * BKT * 2;
  auto kern = gemm_group_kernel<BM, BN, WM, WN, BKT, NSTAGE, PIPE>;
  if (lds > 65536) {
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; }
  }
  hipLaunchKernelGGL(kern, grid, block, lds, s, g);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}


template <int BM, int BN, int WM, int WN, int BKT, int NSTAGE, bool PIPE = false, bool CE = false, bool DR = false>
int launch_gemm(const GemmP& p0, bool xks, bool wks, int zsplits, hipStream_t s) {
  GemmP p = p0;
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
  const int nwg = p.tiles_m * p.tiles_n * zsplits;
  dim3 grid(nwg), block(64 * WM * WN);
  constexpr size_t lds = NSTAGE * (BM + BN) * BKT * 2;
  static_assert(lds >= 64 * (BN + 4) * 4, "epilogue staging must fit in the operand buffers");
#define VAC_LAUNCH(XK, WK)                                                                            \
  do {                                                                                                \
    auto kern = gemm_kernel<BM, BN, WM, WN, BKT, NSTAGE, PIPE, XK, WK, CE, DR>;                                          \
    if (lds > 65536) {                                                                                \
      static bool once = false;                                                                       \
      if (!once) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; } \
    }                                                                                                 \
    hipLaunchKernelGGL(kern, grid, block, lds, s, p);                                                 \
  } while (0)
  if constexpr (CE) {
    if (xks || wks) { vacnic_set_error("gemm: the cross-entropy epilogues are built for the forward layout only"); return VACNIC_UNSUPPORTED; }
    VAC_LAUNCH(false, false);
  } else if constexpr (DR) {         // activation dropout: the forward Linear and the dgrad that carries act' (K-contiguous X)
    if (xks) { vacnic_set_error("gemm: fused activation dropout is built for K-contiguous X (forward / dgrad layouts)"); return VACNIC_UNSUPPORTED; }
    if (wks) VAC_LAUNCH(false, true); else VAC_LAUNCH(false, false);
  } else {
    if (!xks && !wks) VAC_LAUNCH(false, false);
    else if (!xks && wks) VAC_LAUNCH(false, true);
    else if (xks && wks) VAC_LAUNCH(true, true);
    else VAC_LAUNCH(true, false);
  }
#undef VAC_LAUNCH
  VLAUNCH_CHECK();
  return VACNIC_OK;
}


// one entry per tile configuration (defined in gemm_t*.hip)
int launch_t256(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s);   // 256x256 ping-pong
int launch_t264(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s);   // 256x128 ping-pong
int launch_t128(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s);   // 128x128 software-pipelined
int launch_t64(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s);    // 64x128, 4-deep ring
int launch_t260(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s);   // A/B baselines kept for the ablations in profiles/
int launch_t261(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s);
int launch_t262(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s);
int launch_t256ce(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s); // 256x256 ping-pong + LM-head cross-entropy epilogues
int launch_t256d(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s);  // the same tiles with the activation-dropout epilogues (DR)
int launch_t264d(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s);
int launch_t64d(const GemmP& p, bool xks, bool wks, int zsplits, hipStream_t s);
int launch_group128(const GroupP& g, hipStream_t s);                                // grouped weight gradients, 128x128 tiles (gemm_tgroup.hip)

}  // namespace vacgemm
