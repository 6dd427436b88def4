// bf16 MFMA GEMM for gfx950:  out[m][n] = epi(alpha * sum_k X(m,k) W(n,k) + bias[n]).
//
// Replaces every nn.Linear on the VACNIC path (q/k/v/out_proj MFULL:449-452, fc1/fc2 MFULL:581-582,
// _linear_1up/down :587-588, _face_up/down :607-608, ner_map_up/down :594-595, prompt_mlp :1136,
// visual_map :1144, lm_head :1885) and their dgrad / wgrad.
//
// Design (CDNA4): 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave 64x64 as 4x4
// v_mfma_f32_16x16x32_bf16 tiles.  Operands go HBM -> LDS with `buffer_load_dwordx4 ... lds`
// (LDS-DMA, no VGPR round trip; the SRD range check zero-fills M/N/K edges).  The LDS image is
// lane-linear per wave-instruction, so the bank-conflict XOR swizzle is applied to the per-lane
// SOURCE address and again on the fragment read (guide rule 21).  An operand whose reduction index
// is the strided one in memory (dgrad's W, wgrad's dY and X) is staged as [k][row] and read with
// ds_read_b64_tr_b16 (hardware transpose), so no transposed copies are ever materialised.
// MFMA A <- W rows, B <- X rows, i.e. the wave computes the C^T tile: each lane then owns 4
// consecutive n of one m and stores 8/16 contiguous bytes.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;      // 16 KiB, both layouts
constexpr int OOB = 0x7ffffff0;               // voffset beyond any (<2 GiB) buffer -> load returns 0

struct GemmP {
  const bf16_t* x; const bf16_t* w; const float* bias;
  void* out; bf16_t* preact; const bf16_t* dact_src; const bf16_t* residual;
  int M, N, K;
  int ldx, ldw, ldo;
  int act, out_mode, split_k, k_per_split;
  float alpha;
  unsigned x_bytes, w_bytes;
  int tiles_m, tiles_n;
};

// f(k) of the K-strided swizzle: distinct for the 8 k-rows one tr-read half touches.
__device__ __forceinline__ int fk(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Issue the 4 LDS-DMA loads of this thread for one operand tile.
//   KS=false: tile [128 rows][64 k], 128-B rows, chunk' = chunk ^ (row & 7)
//   KS=true : tile [64 k][128 rows], 256-B rows, chunk' = chunk ^ (fk(k) << 1)
template <bool KS>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int r0, int R,
                                           int k0, int kend, int ld, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int blk = wave * 4 + i;
    int voff;
    if (!KS) {
      const int row = blk * 8 + (lane >> 3);
      const int lc = (lane & 7) ^ (row & 7);
      const int gr = r0 + row, gk = k0 + lc * 8;
      voff = (gr < R && gk < kend) ? (int)(((unsigned)gr * (unsigned)ld + (unsigned)gk) * 2u) : OOB;
    } else {
      const int k = blk * 4 + (lane >> 4);
      const int lc = (lane & 15) ^ (fk(k) << 1);
      const int gk = k0 + k, gr = r0 + lc * 8;
      voff = (gk < kend && gr < R) ? (int)(((unsigned)gk * (unsigned)ld + (unsigned)gr) * 2u) : OOB;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + blk * 1024), 16, voff, 0, 0, 0);
  }
}

// Fragment for MFMA 16x16x32: lane l gets element (row = rbase + (l&15), k = kk*32 + 8*(l>>4) + j), j=0..7.
template <bool KS>
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int rbase, int kk, int lane) {
  if (!KS) {
    const int row = rbase + (lane & 15);
    const int chunk = kk * 4 + (lane >> 4);
    return *(const bf16x8*)(lds_tile + row * 128 + ((chunk ^ (row & 7)) << 4));
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int chunk = (rbase >> 3) + (p >> 1);
    bf16x8 r;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int k = kk * 32 + 8 * g + 4 * hh + q;
      const int phys = chunk ^ (fk(k) << 1);
      const char* a = lds_tile + k * 256 + phys * 16 + (p & 1) * 8;
      bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)LDS_PTR(a));
      r[4 * hh + 0] = t[0]; r[4 * hh + 1] = t[1]; r[4 * hh + 2] = t[2]; r[4 * hh + 3] = t[3];
    }
    return r;
  }
}

__device__ __forceinline__ void store_quad(const GemmP& p, f32x4 c, int m, int n, bool add_bias) {
  const size_t off = (size_t)m * p.ldo + n;
  const int nv = p.N - n;               // >= 1; 4 or more means the whole quad is in range
  float v0 = c[0] * p.alpha, v1 = c[1] * p.alpha, v2 = c[2] * p.alpha, v3 = c[3] * p.alpha;
  if (p.bias && add_bias) {
    v0 += p.bias[n];
    if (nv > 1) v1 += p.bias[n + 1];
    if (nv > 2) v2 += p.bias[n + 2];
    if (nv > 3) v3 += p.bias[n + 3];
  }
  const bool vec = (nv >= 4) && ((p.ldo & 3) == 0);
  if (p.preact) {
    if (vec) {
      *(u32x2*)(p.preact + off) = (u32x2){pack2bf(v0, v1), pack2bf(v2, v3)};
    } else {
      p.preact[off] = f2bf(v0);
      if (nv > 1) p.preact[off + 1] = f2bf(v1);
      if (nv > 2) p.preact[off + 2] = f2bf(v2);
      if (nv > 3) p.preact[off + 3] = f2bf(v3);
    }
  }
  if (p.dact_src) {
    float d0, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    if (vec) {
      u32x2 d = *(const u32x2*)(p.dact_src + off);
      d0 = __uint_as_float(d[0] << 16); d1 = __uint_as_float(d[0] & 0xffff0000u);
      d2 = __uint_as_float(d[1] << 16); d3 = __uint_as_float(d[1] & 0xffff0000u);
    } else {
      d0 = bf2f(p.dact_src[off]);
      if (nv > 1) d1 = bf2f(p.dact_src[off + 1]);
      if (nv > 2) d2 = bf2f(p.dact_src[off + 2]);
      if (nv > 3) d3 = bf2f(p.dact_src[off + 3]);
    }
    v0 *= act_bwd(p.act, d0); v1 *= act_bwd(p.act, d1); v2 *= act_bwd(p.act, d2); v3 *= act_bwd(p.act, d3);
  } else if (p.act != VACNIC_ACT_NONE) {
    v0 = act_fwd(p.act, v0); v1 = act_fwd(p.act, v1); v2 = act_fwd(p.act, v2); v3 = act_fwd(p.act, v3);
  }
  if (p.residual) {
    if (vec) {
      u32x2 d = *(const u32x2*)(p.residual + off);
      v0 += __uint_as_float(d[0] << 16); v1 += __uint_as_float(d[0] & 0xffff0000u);
      v2 += __uint_as_float(d[1] << 16); v3 += __uint_as_float(d[1] & 0xffff0000u);
    } else {
      v0 += bf2f(p.residual[off]);
      if (nv > 1) v1 += bf2f(p.residual[off + 1]);
      if (nv > 2) v2 += bf2f(p.residual[off + 2]);
      if (nv > 3) v3 += bf2f(p.residual[off + 3]);
    }
  }
  if (p.out_mode == 0) {
    bf16_t* o = (bf16_t*)p.out + off;
    if (vec) {
      *(u32x2*)o = (u32x2){pack2bf(v0, v1), pack2bf(v2, v3)};
    } else {
      o[0] = f2bf(v0);
      if (nv > 1) o[1] = f2bf(v1);
      if (nv > 2) o[2] = f2bf(v2);
      if (nv > 3) o[3] = f2bf(v3);
    }
  } else if (p.out_mode == 1) {
    float* o = (float*)p.out + off;
    if (vec) {
      *(f32x4*)o = (f32x4){v0, v1, v2, v3};
    } else {
      o[0] = v0;
      if (nv > 1) o[1] = v1;
      if (nv > 2) o[2] = v2;
      if (nv > 3) o[3] = v3;
    }
  } else {
    float* o = (float*)p.out + off;
    atomicAdd(o, v0);
    if (nv > 1) atomicAdd(o + 1, v1);
    if (nv > 2) atomicAdd(o + 2, v2);
    if (nv > 3) atomicAdd(o + 3, v3);
  }
}

template <bool XKS, bool WKS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;

  // XCD-aware tile order: blocks b, b+8, ... share an XCD; give each XCD a contiguous run of tiles
  // (bijective for any tile count), n fastest so neighbours reuse the same X panel in their L2.
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int ntile = (kend - kbeg + BK - 1) / BK;

  __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  // K-strided operands may read up to round_up(R, 8) columns of a row (host guarantees ld covers it)
  const int RX = XKS ? ((p.M + 7) & ~7) : p.M;
  const int RW = WKS ? ((p.N + 7) & ~7) : p.N;

  f32x4 acc[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[b][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // LDS: buffer b at smem + b*2*TILE_BYTES = {X tile, W tile}
  if (ntile > 0) {
    stage_tile<XKS>(xs, smem, m0, RX, kbeg, kend, p.ldx, wave, lane);
    stage_tile<WKS>(ws, smem + TILE_BYTES, n0, RW, kbeg, kend, p.ldw, wave, lane);
  }
  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    char* xcur = smem + cur * (2 * TILE_BYTES);
    char* wcur = xcur + TILE_BYTES;
    if (t + 1 < ntile) {
      // buffer cur^1 was last read in iteration t-1; every wave has passed that iteration's
      // trailing barrier, so it is free to overwrite.
      char* xnext = smem + (cur ^ 1) * (2 * TILE_BYTES);
      stage_tile<XKS>(xs, xnext, m0, RX, kbeg + (t + 1) * BK, kend, p.ldx, wave, lane);
      stage_tile<WKS>(ws, xnext + TILE_BYTES, n0, RW, kbeg + (t + 1) * BK, kend, p.ldw, wave, lane);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile t landed (this wave's 8 loads)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                         // ... and every other wave's
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 xf[4], wf[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) xf[a] = read_frag<XKS>(xcur, wm * 64 + a * 16, kk, lane);
#pragma unroll
      for (int b = 0; b < 4; ++b) wf[b] = read_frag<WKS>(wcur, wn * 64 + b * 16, kk, lane);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a)
          acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[b][a], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // all reads of buffer cur done
  }

  // ---- epilogue: lane owns out[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4 ----
  const int lm = lane & 15, ln4 = (lane >> 4) * 4;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int m = m0 + wm * 64 + a * 16 + lm;
      const int n = n0 + wn * 64 + b * 16 + ln4;
      if (m < p.M && n < p.N) store_quad(p, acc[b][a], m, n, blockIdx.z == 0);
    }
  }
}

}  // namespace

extern "C" int vacnic_gemm_bf16(const vacnic_gemm_args* a, void* stream) {
  VCHECK(a && a->x && a->w && a->out, VACNIC_BAD_SHAPE, "gemm: null operand");
  VCHECK(a->M > 0 && a->N > 0 && a->K > 0, VACNIC_BAD_SHAPE, "gemm: empty problem M=%ld N=%ld K=%ld",
         (long)a->M, (long)a->N, (long)a->K);
  VCHECK((a->ldx & 7) == 0 && (a->ldw & 7) == 0, VACNIC_MISALIGNED, "gemm: ldx/ldw must be multiples of 8");
  VCHECK(aligned16(a->x) && aligned16(a->w), VACNIC_MISALIGNED, "gemm: x/w must be 16-byte aligned");
  VCHECK(a->out_mode >= 0 && a->out_mode <= 2, VACNIC_BAD_DTYPE, "gemm: bad out_mode %d", a->out_mode);
  const int split = a->split_k < 1 ? 1 : a->split_k;
  VCHECK(split == 1 || a->out_mode == 2, VACNIC_UNSUPPORTED, "gemm: split_k needs out_mode 2");
  VCHECK(a->ldo >= a->N, VACNIC_BAD_SHAPE, "gemm: ldo < N");
  if (a->x_kstrided) VCHECK(a->ldx >= ((a->M + 7) & ~7LL), VACNIC_BAD_SHAPE, "gemm: ldx too small for K-strided X");
  else VCHECK(a->ldx >= ((a->K + 7) & ~7LL), VACNIC_BAD_SHAPE, "gemm: ldx < round_up(K, 8) for K-contiguous X");
  if (a->w_kstrided) VCHECK(a->ldw >= ((a->N + 7) & ~7LL), VACNIC_BAD_SHAPE, "gemm: ldw too small for K-strided W");
  else VCHECK(a->ldw >= ((a->K + 7) & ~7LL), VACNIC_BAD_SHAPE, "gemm: ldw < round_up(K, 8) for K-contiguous W");

  auto span = [](int64_t rows, int64_t cols, int64_t ld) { return ((rows - 1) * ld + cols) * 2; };
  const int64_t K8 = (a->K + 7) & ~7LL;
  const int64_t xb = a->x_kstrided ? span(a->K, (a->M + 7) & ~7LL, a->ldx) : span(a->M, K8, a->ldx);
  const int64_t wb = a->w_kstrided ? span(a->K, (a->N + 7) & ~7LL, a->ldw) : span(a->N, K8, a->ldw);
  VCHECK(xb < 0x7ffffff0LL && wb < 0x7ffffff0LL, VACNIC_UNSUPPORTED, "gemm: operand larger than 2 GiB");

  GemmP p;
  p.x = (const bf16_t*)a->x; p.w = (const bf16_t*)a->w; p.bias = a->bias;
  p.out = a->out; p.preact = (bf16_t*)a->preact; p.dact_src = (const bf16_t*)a->dact_src;
  p.residual = (const bf16_t*)a->residual;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
  p.ldx = (int)a->ldx; p.ldw = (int)a->ldw; p.ldo = (int)a->ldo;
  p.act = a->act; p.out_mode = a->out_mode; p.split_k = split;
  int kps = (int)((a->K + split - 1) / split);
  kps = (kps + BK - 1) / BK * BK;
  p.k_per_split = kps;
  const int zsplits = (int)((a->K + kps - 1) / kps);
  p.alpha = a->alpha;
  p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
  p.tiles_m = (int)((a->M + BM - 1) / BM); p.tiles_n = (int)((a->N + BN - 1) / BN);
  dim3 grid(p.tiles_m * p.tiles_n, 1, zsplits), block(256);
  const size_t lds = 4 * TILE_BYTES;
  hipStream_t s = (hipStream_t)stream;
  if (!a->x_kstrided && !a->w_kstrided) hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, lds, s, p);
  else if (!a->x_kstrided && a->w_kstrided) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, lds, s, p);
  else if (a->x_kstrided && a->w_kstrided) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, lds, s, p);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
