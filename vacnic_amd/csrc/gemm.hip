// bf16 MFMA GEMM for gfx950:  out[m][n] = epi(alpha * sum_k X(m,k) W(n,k) + bias[n]).
//
// Replaces every nn.Linear on the VACNIC path (q/k/v/out_proj MFULL:449-452, fc1/fc2 MFULL:581-582,
// _linear_1up/down :587-588, _face_up/down :607-608, ner_map_up/down :594-595, prompt_mlp :1136,
// visual_map :1144, lm_head :1885) and their dgrad / wgrad.
//
// Design (CDNA4): block tile BMxBN with WMxWN waves, each wave a (BM/WM)x(BN/WN) sub-tile of v_mfma_f32_16x16x32_bf16
// tiles.  Configurations (chosen per launch by the host cost model below, or forced by tile_hint):
//   256x256, 8 waves (2x4, 128x64 per wave), 32-wide K stages in a 4-slot LDS ring (128 KiB, one block per CU), PING-PONG
//            K loop: the two waves of every SIMD alternate MFMA and LDS/DMA phases — the large training GEMMs;
//   128x128, 4 waves, 64-wide K tiles, two blocks per CU, software-pipelined loop (barrier in the middle of the tile's MFMAs),
//            register->global epilogue — when that fills the chip better;
//   64x128,  4 waves, 4-deep ring — small latency-bound problems;   256x128 ping-pong (hint 264);
//   skinny   (M <= 8): W-streaming kernel without LDS — the single-token decoder.
// Operands go HBM -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA, no VGPR round trip; the SRD range check zero-fills
// M/N/K edges) with counted vmcnt waits.  The LDS image is lane-linear per wave-instruction,
// so the bank-conflict XOR swizzle is applied to the per-lane SOURCE address and again on the
// fragment read (guide rule 21).  An operand whose reduction index is the strided one in memory
// (dgrad's W, wgrad's dY and X) is staged as [k][row] and read with ds_read_b64_tr_b16 (hardware
// transpose), so no transposed copies are ever materialised.  MFMA A <- W rows, B <- X rows, i.e. the
// wave computes the C^T tile; the fp32 C tile is then staged through LDS so that every global access
// of the epilogue (output, saved pre-activation, activation-backward source, residual) is a 16-byte
// row-contiguous access and split-K atomics are 256 contiguous bytes per wave-instruction.
#include "gemm_kernel.h"
#include <vector>

using namespace vacgemm;

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Skinny-M GEMM (M <= 8): the single-token decoder of caption generation (MFULL:474-501, beams*batch rows) and any other
// GEMM whose X is a handful of rows.  HBM-bound: W [N,K] is streamed exactly once; X (<= 8 x K bf16) is re-read from
// L1/L2.  One wave owns CW consecutive output columns; its 64 lanes split K in 16-byte chunks, accumulate MR x CW partial
// dot products in fp32 and meet in a wave reduction.  No LDS, no barriers.
template <int MR, int CW, int KW>
__global__ __launch_bounds__(64 * KW) void gemm_skinny_kernel(GemmP p) {
  // one workgroup per CW output columns: N / CW independent workgroups spread over every CU (N = 1024 -> 256), each limited
  // only by one round trip of its loads.  KW waves split K (long reductions, e.g. fc2 with K = 4096) and meet through LDS.
  const int lane = threadIdx.x & 63;
  const int kw = KW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.x * CW;
  constexpr int NV = MR * CW;                        // 32 partial dot products per lane
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.f;
  const int nchunk = p.K >> 3;                       // host guarantees K % 8 == 0
#pragma unroll 4
  for (int ch = lane + 64 * kw; ch < nchunk; ch += 64 * KW) {
    float wv[CW][8];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      if (n0 + c < p.N) load8bf(p.w + (size_t)(n0 + c) * p.ldw + ch * 8, wv[c]);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[c][j] = 0.f;
      }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < p.M) {
        float xv[8];
        load8bf(p.x + (size_t)m * p.ldx + ch * 8, xv);
#pragma unroll
        for (int c = 0; c < CW; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[m * CW + c] += xv[j] * wv[c][j];
      }
    }
  }
  // reduce-scatter butterfly: at offset 32,16,8,4,2 every lane hands the half of its values its partner keeps and adds the
  // half it keeps (16+8+4+2+1 exchanges), then one last exchange at offset 1 -> 32 cross-lane moves for 32 sums instead of 192.
  static_assert(NV == 32, "butterfly written for 32 values per lane");
#define VAC_BFLY(OFF, HALF)                                                      \
  {                                                                              \
    const bool up = (lane & OFF) != 0;                                           \
    _Pragma("unroll") for (int i = 0; i < HALF; ++i) {                           \
      const float send = up ? acc[i] : acc[HALF + i];                            \
      const float keep = up ? acc[HALF + i] : acc[i];                            \
      acc[i] = keep + __shfl_xor(send, OFF, 64);                                 \
    }                                                                            \
  }
  VAC_BFLY(32, 16) VAC_BFLY(16, 8) VAC_BFLY(8, 4) VAC_BFLY(4, 2) VAC_BFLY(2, 1)
#undef VAC_BFLY
  float v = acc[0] + __shfl_xor(acc[0], 1, 64);
  const int idx = (((lane >> 5) & 1) << 4) | (((lane >> 4) & 1) << 3) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 1) | ((lane >> 1) & 1);
  if constexpr (KW > 1) {
    __shared__ float part[KW][NV];
    if ((lane & 1) == 0) part[kw][idx] = v;
    __syncthreads();
    if (kw != 0) return;
    v = 0.f;
#pragma unroll
    for (int w = 0; w < KW; ++w) v += part[w][idx];
  }
  if ((lane & 1) == 0) {
    const int m = idx / CW, c = idx % CW;
    const int n = n0 + c;
    if (m < p.M && n < p.N) {
      v *= p.alpha;
      if (p.bias) v += p.bias[n];
      v = act_fwd(p.act, v);
      const size_t off = (size_t)m * p.ldo + n;
      if (p.out_mode == 0) ((bf16_t*)p.out)[off] = f2bf(v);
      else if (p.out_mode == 1) ((float*)p.out)[off] = v;
      else ((float*)p.out)[off] += v;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Skinny GEMM with a residual + LayerNorm PROLOGUE (single-token decoder): y = epi( LN(x + res) . W^T + b ), and the
// normalised rows LN(x + res) are written once (by workgroup 0) because the next block needs them as ITS residual.
// Replaces {add_ln_fwd, gemm_skinny} pairs of the decode step (36 of its 142 launches per token): every wave recomputes the
// statistics of the <= 8 rows (10 KB from L2, two wave reductions per row) WHILE its weight loads — issued first, they do not
// depend on x — are still in flight, so the prologue costs no latency of its own.  The arithmetic is add_ln_fwd_kernel's, in the
// same order on the same lanes (chunk c = lane + 64 i, fp32 sums, wave butterfly), and the normalised values are rounded to
// bf16 before the dot products exactly as if they had been stored and re-loaded: the logits are bit-identical to the
// unfused chain.
struct GemvLnP {
  const bf16_t* x; const bf16_t* res; const float* gamma; const float* beta; bf16_t* ln_out;
  const bf16_t* w; const float* bias; void* out;
  int M, N, K, ldw, ldo, act, out_mode;
  float eps;
};

template <int MR, int CW, int NCH>
__global__ __launch_bounds__(64) void gemv_ln_kernel(GemvLnP p) {
  const int lane = threadIdx.x & 63;
  const int n0 = blockIdx.x * CW;
  const int nchunk = p.K >> 3;
  // 1. activations, residual and LayerNorm parameters FIRST, weights second: vmcnt retires loads in issue order, so the
  //    statistics can start as soon as the (L2-resident) rows are back while the weight stream from HBM is still in flight
  u32x4 xraw[MR][NCH], rraw[MR][NCH];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = lane + 64 * i;
      xraw[m][i] = (u32x4){0, 0, 0, 0}; rraw[m][i] = (u32x4){0, 0, 0, 0};
      if (m < p.M && ch < nchunk) {
        xraw[m][i] = *(const u32x4*)(p.x + (size_t)m * p.K + ch * 8);
        rraw[m][i] = *(const u32x4*)(p.res + (size_t)m * p.K + ch * 8);
      }
    }
  float gam[NCH][8], bet[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
#pragma unroll
    for (int j = 0; j < 8; ++j) { gam[i][j] = 0.f; bet[i][j] = 0.f; }
    if (ch < nchunk) {
      const f32x4 g0 = *(const f32x4*)(p.gamma + ch * 8), g1 = *(const f32x4*)(p.gamma + ch * 8 + 4);
      const f32x4 b0 = *(const f32x4*)(p.beta + ch * 8), b1 = *(const f32x4*)(p.beta + ch * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { gam[i][j] = g0[j]; gam[i][4 + j] = g1[j]; bet[i][j] = b0[j]; bet[i][4 + j] = b1[j]; }
    }
  }
  u32x4 wraw[CW][NCH];
#pragma unroll
  for (int c = 0; c < CW; ++c)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = lane + 64 * i;
      wraw[c][i] = (u32x4){0, 0, 0, 0};
      if (ch < nchunk && n0 + c < p.N) wraw[c][i] = *(const u32x4*)(p.w + (size_t)(n0 + c) * p.ldw + ch * 8);
    }
  // 2. h = x + res, LayerNorm statistics per row (add_ln_fwd_kernel's order of operations)
  float h[MR][NCH][8];
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = lane + 64 * i;
      if (m < p.M && ch < nchunk) {
        float xv[8], rv[8];
        unpack8bf(xraw[m][i], xv);
        unpack8bf(rraw[m][i], rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) { xv[j] += rv[j]; h[m][i][j] = xv[j]; s += xv[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) h[m][i][j] = 0.f;
      }
    }
    if (m < p.M) {
      const float mean = wave_sum(s) / p.K;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (lane + 64 * i < nchunk) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float d = h[m][i][j] - mean; q += d * d; }
        }
      }
      const float rstd = rsqrtf(wave_sum(q) / p.K + p.eps);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nchunk) {
          float y[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) y[j] = (h[m][i][j] - mean) * rstd * gam[i][j] + bet[i][j];
          const u32x4 packed = (u32x4){pack2bf(y[0], y[1]), pack2bf(y[2], y[3]), pack2bf(y[4], y[5]), pack2bf(y[6], y[7])};
          if (blockIdx.x == 0 && p.ln_out) *(u32x4*)(p.ln_out + (size_t)m * p.K + ch * 8) = packed;
          unpack8bf(packed, h[m][i]);                    // the bf16-rounded values, as the unfused chain would re-load them
        }
      }
    }
  }
  // 3. dot products + the skinny kernel's reduce-scatter butterfly and epilogue
  constexpr int NV = MR * CW;
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      float wv[8];
      unpack8bf(wraw[c][i], wv);
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[m * CW + c] += h[m][i][j] * wv[j];
    }
  static_assert(NV == 32, "butterfly written for 32 values per lane");
#define VAC_BFLY(OFF, HALF)                                                      \
  {                                                                              \
    const bool up = (lane & OFF) != 0;                                           \
    _Pragma("unroll") for (int i = 0; i < HALF; ++i) {                           \
      const float send = up ? acc[i] : acc[HALF + i];                            \
      const float keep = up ? acc[HALF + i] : acc[i];                            \
      acc[i] = keep + __shfl_xor(send, OFF, 64);                                 \
    }                                                                            \
  }
  VAC_BFLY(32, 16) VAC_BFLY(16, 8) VAC_BFLY(8, 4) VAC_BFLY(4, 2) VAC_BFLY(2, 1)
#undef VAC_BFLY
  float v = acc[0] + __shfl_xor(acc[0], 1, 64);
  const int idx = (((lane >> 5) & 1) << 4) | (((lane >> 4) & 1) << 3) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 1) | ((lane >> 1) & 1);
  if ((lane & 1) == 0) {
    const int m = idx / CW, c = idx % CW;
    const int n = n0 + c;
    if (m < p.M && n < p.N) {
      if (p.bias) v += p.bias[n];
      v = act_fwd(p.act, v);
      const size_t off = (size_t)m * p.ldo + n;
      if (p.out_mode == 0) ((bf16_t*)p.out)[off] = f2bf(v);
      else ((float*)p.out)[off] = v;
    }
  }
}

}  // namespace

extern "C" int vacnic_gemv_ln_bf16(const vacnic_gemv_ln_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_gemv_ln_bf16, a, stream);
  VCHECK(a && a->x && a->residual && a->gamma && a->beta && a->w && a->out, VACNIC_BAD_SHAPE, "gemv_ln: null operand");
  VCHECK(a->M >= 1 && a->M <= 8 && a->N > 0, VACNIC_UNSUPPORTED, "gemv_ln: 1 <= M <= 8 rows");
  VCHECK(a->K > 0 && (a->K & 7) == 0 && a->K <= 1024, VACNIC_UNSUPPORTED, "gemv_ln: K %% 8 == 0 and K <= 1024 (one LayerNorm row per wave)");
  VCHECK((a->ldw & 7) == 0 && a->ldw >= a->K && aligned16(a->x) && aligned16(a->residual) && aligned16(a->w), VACNIC_MISALIGNED,
         "gemv_ln: 16-byte aligned operands, ldw %% 8 == 0");
  VCHECK(a->out_mode == 0 || a->out_mode == 1, VACNIC_BAD_DTYPE, "gemv_ln: out_mode 0 (bf16) or 1 (f32)");
  VCHECK(a->ldo >= a->N, VACNIC_BAD_SHAPE, "gemv_ln: ldo < N");
  GemvLnP p;
  p.x = (const bf16_t*)a->x; p.res = (const bf16_t*)a->residual; p.gamma = a->gamma; p.beta = a->beta; p.ln_out = (bf16_t*)a->ln_out;
  p.w = (const bf16_t*)a->w; p.bias = a->bias; p.out = a->out;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K; p.ldw = (int)a->ldw; p.ldo = (int)a->ldo; p.act = a->act; p.out_mode = a->out_mode;
  p.eps = a->eps;
  constexpr int CW = 4;
  const dim3 grid((unsigned)((a->N + CW - 1) / CW));
  if (a->K <= 512) hipLaunchKernelGGL((gemv_ln_kernel<8, CW, 1>), grid, dim3(64), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((gemv_ln_kernel<8, CW, 2>), grid, dim3(64), 0, (hipStream_t)stream, p);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

// ---- host-side tile selection --------------------------------------------------------------------------------
// Cost model in "MFMA work units" (bm*bn*k elements): a launch needs ceil(tiles / resident slots) rounds; a round
// costs the time of one tile with the CU shared by `per_cu` co-resident workgroups, plus a fixed ~10 us of launch /
// prologue / epilogue (measured: profiles/r1_gemm_overhead.txt), expressed in the same units.
struct TileCfg { int bm, bn, slots, per_cu; double eff; int hint; };
static const TileCfg kCfgs[3] = {
    {256, 256, 256, 1, 1.00, 256},     // 8 waves, 128 KiB LDS, one workgroup per CU
    {128, 128, 512, 2, 0.80, 128},     // 4 waves, 64 KiB LDS, two per CU
    {64, 128, 256, 1, 0.42, 64},       // 4 waves, 4-deep ring (96 KiB): latency-bound small problems
};
static const double kFixedUnits = 27e6;

static double cfg_cost(const TileCfg& c, int64_t M, int64_t N, int64_t kper, int64_t z) {
  const int64_t tiles = ((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn) * z;
  const int64_t rounds = (tiles + c.slots - 1) / c.slots;
  return (double)rounds * ((double)c.per_cu * c.bm * c.bn * (double)kper / c.eff + kFixedUnits);
}

static int gemm_one(const vacnic_gemm_args* a, int hint, void* stream, int ce_mode = 0, int ce_col0 = 0);

// fix-up buffers: partial tiles are padded to whole tiles; the bound covers every configuration (64-row tiles pad M the least,
// 256 x 256 the most), so the caller need not know which one a launch picks
extern "C" int64_t vacnic_gemm_workspace_bytes(int64_t M, int64_t N, int64_t split_k) {
  if (M <= 0 || N <= 0 || split_k <= 1) return 0;
  const int64_t mp = (M + 255) / 256 * 256, np = (N + 255) / 256 * 256;
  return mp * np * 4 * split_k;
}
extern "C" int64_t vacnic_gemm_counters(int64_t M, int64_t N) {
  if (M <= 0 || N <= 0) return 0;
  return ((M + 63) / 64) * ((N + 127) / 128);
}

extern "C" int vacnic_gemm_bf16(const vacnic_gemm_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_gemm_bf16, a, stream);
  VCHECK(a && a->x && a->w && a->out, VACNIC_BAD_SHAPE, "gemm: null operand");
  VCHECK(a->M > 0 && a->N > 0 && a->K > 0, VACNIC_BAD_SHAPE, "gemm: empty problem M=%ld N=%ld K=%ld",
         (long)a->M, (long)a->N, (long)a->K);
  if (a->tile_hint != 0) return gemm_one(a, a->tile_hint, stream);
  if (a->M <= 8 && !a->x_kstrided && !a->w_kstrided && a->split_k <= 1 && !a->preact && !a->dact_src && !a->residual && !a->xsum &&
      (a->K & 7) == 0)
    return gemm_one(a, 8, stream);
  const int split = a->split_k < 1 ? 1 : a->split_k;
  int64_t kper = (a->K + split - 1) / split;
  kper = (kper + BK - 1) / BK * BK;
  const int64_t z = (a->K + kper - 1) / kper;
  // best single launch
  int best = 0; double best_cost = 1e300;
  for (int i = 0; i < 3; ++i) {
    if (i == 0 && (a->M < 256 || a->N < 256)) continue;
    const double c = cfg_cost(kCfgs[i], a->M, a->N, kper, z);
    if (c < best_cost) { best_cost = c; best = i; }
  }
  // M-split: rows are independent, so a small row remainder that would open an extra (almost empty) round is cut off
  // and run as its own small launch (e.g. the ViT's M = 32*257 = 8224 = 8192 + 32)
  int split_cfg = -1; int64_t m_main = 0; double split_cost = best_cost;
  for (int i = 0; i < 2; ++i) {
    const int64_t rem = a->M % kCfgs[i].bm;
    if (rem == 0 || a->M - rem < kCfgs[i].bm || (i == 0 && a->N < 256)) continue;
    const double c = cfg_cost(kCfgs[i], a->M - rem, a->N, kper, z) + cfg_cost(kCfgs[2], rem, a->N, kper, z);
    if (c < 0.85 * split_cost) { split_cost = c; split_cfg = i; m_main = a->M - rem; }
  }
  if (split_cfg < 0) return gemm_one(a, kCfgs[best].hint, stream);
  vacnic_gemm_args main_a = *a, tail_a = *a;
  main_a.M = m_main;
  tail_a.M = a->M - m_main;
  const int64_t xoff = a->x_kstrided ? m_main : m_main * a->ldx;
  tail_a.x = (const char*)a->x + xoff * 2;
  const int64_t osz = a->out_mode == 0 ? 2 : 4;
  tail_a.out = (char*)a->out + m_main * a->ldo * osz;
  if (a->preact) tail_a.preact = (char*)a->preact + m_main * a->ldo * 2;
  if (a->dact_src) tail_a.dact_src = (const char*)a->dact_src + m_main * a->ldo * 2;
  if (a->residual) tail_a.residual = (const char*)a->residual + m_main * a->ldo * 2;
  if (a->xsum) tail_a.xsum = a->xsum + m_main;
  if (int e = gemm_one(&main_a, kCfgs[split_cfg].hint, stream)) return e;
  return gemm_one(&tail_a, kCfgs[2].hint, stream);
}

static int gemm_one(const vacnic_gemm_args* a, int tile_hint, void* stream, int ce_mode, int ce_col0) {
  VCHECK((a->ldx & 7) == 0 && (a->ldw & 7) == 0, VACNIC_MISALIGNED, "gemm: ldx/ldw must be multiples of 8");
  VCHECK(aligned16(a->x) && aligned16(a->w), VACNIC_MISALIGNED, "gemm: x/w must be 16-byte aligned");
  VCHECK(a->out_mode >= 0 && a->out_mode <= 2, VACNIC_BAD_DTYPE, "gemm: bad out_mode %d", a->out_mode);
  const int split = a->split_k < 1 ? 1 : a->split_k;
  VCHECK(split == 1 || a->out_mode == 2 || a->workspace, VACNIC_UNSUPPORTED, "gemm: split_k needs out_mode 2 or a fix-up workspace");
  if (a->workspace && split > 1) {
    VCHECK(a->counters && aligned16(a->workspace), VACNIC_BAD_SHAPE, "gemm: the split-K fix-up needs `counters` and a 16-byte aligned workspace");
    VCHECK(a->workspace_bytes >= vacnic_gemm_workspace_bytes(a->M, a->N, split) && a->counters_len >= vacnic_gemm_counters(a->M, a->N), VACNIC_BAD_SHAPE,
           "gemm: fix-up workspace %ld B / %ld counters, need %ld B / %ld", (long)a->workspace_bytes, (long)a->counters_len,
           (long)vacnic_gemm_workspace_bytes(a->M, a->N, split), (long)vacnic_gemm_counters(a->M, a->N));
    VCHECK(tile_hint != 8 && !ce_mode, VACNIC_UNSUPPORTED, "gemm: no split-K fix-up in the skinny / cross-entropy kernels");
  }
  VCHECK(a->bias == nullptr || aligned16(a->bias), VACNIC_MISALIGNED, "gemm: bias must be 16-byte aligned");
  VCHECK(a->ldo >= a->N, VACNIC_BAD_SHAPE, "gemm: ldo < N");
  VCHECK(!a->xsum || ce_mode || (a->x_kstrided && a->w_kstrided), VACNIC_UNSUPPORTED,
         "gemm: xsum (row sums of X) is implemented for the weight-gradient layout only (x_kstrided and w_kstrided)");
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
  p.xsum = a->xsum;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
  p.ldx = (int)a->ldx; p.ldw = (int)a->ldw; p.ldo = (int)a->ldo;
  p.act = a->act; p.out_mode = ce_mode ? ce_mode : a->out_mode; p.split_k = split;
  p.ce_col0 = ce_col0;
  int kps = (int)((a->K + split - 1) / split);
  kps = (kps + BK - 1) / BK * BK;               // multiple of 64: valid for both K-tile depths
  p.k_per_split = kps;
  const int zsplits = (int)((a->K + kps - 1) / kps);
  p.split_k = zsplits;
  p.alpha = a->alpha;
  p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
  p.tiles_m = p.tiles_n = 0;
  p.ws = (a->workspace && zsplits > 1) ? (float*)a->workspace : nullptr;
  p.cnt = p.ws ? a->counters : nullptr;
  p.drop_thr = 0; p.drop_inv = 1.f; p.drop_seed = a->drop_seed; p.drop_seed_dev = (const unsigned long long*)a->drop_seed_dev;
  if (a->drop_p != 0.f) {
    VCHECK(a->drop_p > 0.f && a->drop_p < 1.f, VACNIC_BAD_SHAPE, "gemm: drop_p must be in [0, 1)");
    VCHECK(a->out_mode == 0 && a->ldo == a->N && (a->N & 15) == 0 && !ce_mode && tile_hint != 8, VACNIC_UNSUPPORTED,
           "gemm: fused activation dropout needs a contiguous bf16 output with N %% 16 == 0 (use vacnic_dropout_bf16)");
    VCHECK((zsplits == 1 || p.ws) , VACNIC_UNSUPPORTED, "gemm: fused activation dropout with split-K needs the fix-up workspace");
    unsigned thr = (unsigned)(a->drop_p * 256.f + 0.5f);
    if (thr > 255) thr = 255;
    p.drop_thr = thr;                                  // (p < 1/512 rounds to "off")
    p.drop_inv = 256.f / (256.f - (float)thr);
  }
  hipStream_t s = (hipStream_t)stream;
  if (tile_hint == 8) {     // skinny-M kernel (chosen by vacnic_gemm_bf16 for M <= 8, or forced by the caller)
    VCHECK(a->M <= 8 && !a->x_kstrided && !a->w_kstrided && zsplits == 1 && !a->preact && !a->dact_src && !a->residual && !a->xsum &&
           (a->K & 7) == 0, VACNIC_UNSUPPORTED, "gemm: the skinny kernel needs M <= 8, K-contiguous operands, K %% 8 == 0 and a plain epilogue");
    constexpr int CW = 4;
    dim3 grid((unsigned)((a->N + CW - 1) / CW));
    if (a->K >= 2048 && a->N <= 8192) hipLaunchKernelGGL((gemm_skinny_kernel<8, CW, 4>), grid, dim3(256), 0, s, p);   // long reduction, few columns: 4 waves split K
    else hipLaunchKernelGGL((gemm_skinny_kernel<8, CW, 1>), grid, dim3(64), 0, s, p);
    VLAUNCH_CHECK();
    return VACNIC_OK;
  }
  const int force = tile_hint % 1000;
  p.debug = tile_hint / 1000;
  {
    // A/B aid for whole-step measurements on ONE box (device-to-device spread is larger than most kernel deltas):
    // VACNIC_GEMM_DEBUG=<bits> is OR-ed into the debug mask of every launch (e.g. 80 = fp32 epilogue + one workgroup per tile)
    static int env_debug = -1;
    if (env_debug < 0) { const char* e = getenv("VACNIC_GEMM_DEBUG"); env_debug = e ? atoi(e) : 0; }
    p.debug |= env_debug;
  }
  // the bf16 epilogue addresses its outputs through 32-bit buffer offsets: larger outputs take the fp32-staged path
  if (((a->M - 1) * a->ldo + a->N) * 2 >= 0x7ffffff0LL) p.debug |= 16;
  if ((a->preact != nullptr) + (a->dact_src != nullptr) + (a->residual != nullptr) > 1) p.debug |= 16;   // bf16 epilogue: one extra operand
  if (p.drop_thr && !a->preact && !a->dact_src) p.debug |= 16;     // dropout rides on the saved-pre-activation / act' instances of the bf16 epilogue
  const bool big = force == 256;
  const bool mid = force == 128;
  if (p.drop_thr) {                 // activation dropout: 256 x 256, 256 x 128 and 64 x 128 tiles carry the DR epilogues
    if (big) return launch_t256d(p, a->x_kstrided, a->w_kstrided, zsplits, s);
    if (force == 264) return launch_t264d(p, a->x_kstrided, a->w_kstrided, zsplits, s);
    return launch_t64d(p, a->x_kstrided, a->w_kstrided, zsplits, s);
  }
  // A/B baselines kept for the ablations in profiles/: plain K loops and the 64-wide software-pipelined loop
  if (force == 260) return launch_t260(p, a->x_kstrided, a->w_kstrided, zsplits, s);
  if (force == 261) return launch_t261(p, a->x_kstrided, a->w_kstrided, zsplits, s);
  if (force == 262) return launch_t262(p, a->x_kstrided, a->w_kstrided, zsplits, s);
  if (force == 264) return launch_t264(p, a->x_kstrided, a->w_kstrided, zsplits, s);   // ping-pong, half-width tile
  if (ce_mode) return launch_t256ce(p, a->x_kstrided, a->w_kstrided, zsplits, s);
  if (big) return launch_t256(p, a->x_kstrided, a->w_kstrided, zsplits, s);   // ping-pong loop
  if (mid) return launch_t128(p, a->x_kstrided, a->w_kstrided, zsplits, s);
  return launch_t64(p, a->x_kstrided, a->w_kstrided, zsplits, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// Grouped weight gradients (nn.Linear backward w.r.t. weight and bias, MFULL:449-452, for several layers at once):
// dw_j[N_j, K_j] += dy_j[M_j, N_j]^T x_j[M_j, K_j],  dbias_j[N_j] += column sums of dy_j.   See gemm_group_kernel.
extern "C" int vacnic_wgrad_group(const vacnic_wgrad_job* jobs, int64_t njobs, void* stream) {
  VCHECK(jobs && njobs > 0, VACNIC_BAD_SHAPE, "wgrad_group: no jobs");
  vplan::Guard vplan_guard__;
  if (vplan::outermost()) {
    std::vector<vacnic_wgrad_job> copy__(jobs, jobs + njobs);          // the job table is host memory of the caller: freeze it
    vplan::push([copy__, stream]() { return vacnic_wgrad_group(copy__.data(), (int64_t)copy__.size(), stream); });
  }
  constexpr int64_t US = (int64_t)GROUP_UT * 128;
  GroupP g;
  g.nunits = 0;
  g.debug = 0;
  {
    static int env_debug = -1;
    if (env_debug < 0) { const char* e = getenv("VACNIC_GEMM_DEBUG"); env_debug = e ? atoi(e) : 0; }
    g.debug = env_debug & ~8;
  }
  static int phase_rows = -1;            // rows of the reduction per launch (VACNIC_WGRAD_PHASE_ROWS; 0 = whole reduction in one launch)
  if (phase_rows < 0) { const char* e = getenv("VACNIC_WGRAD_PHASE_ROWS"); phase_rows = e ? atoi(e) : 0; phase_rows = (phase_rows + BK - 1) / BK * BK; }
  auto flush = [&]() -> int {
    if (g.nunits == 0) return VACNIC_OK;
    int kmax = 0;
    for (int i = 0; i < g.nunits; ++i) kmax = g.u[i].K > kmax ? g.u[i].K : kmax;
    g.k_per_phase = phase_rows > 0 ? phase_rows : (kmax + BK - 1) / BK * BK;
    const int phases = (kmax + g.k_per_phase - 1) / g.k_per_phase;
    for (g.kphase = 0; g.kphase < phases; ++g.kphase)
      if (int e = launch_group128(g, (hipStream_t)stream)) { g.nunits = 0; return e; }
    g.nunits = 0;
    return VACNIC_OK;
  };
  auto span = [](int64_t rows, int64_t cols, int64_t ld) { return ((rows - 1) * ld + cols) * 2; };
  for (int64_t j = 0; j < njobs; ++j) {
    const vacnic_wgrad_job& a = jobs[j];
    VCHECK(a.dy && a.x && a.dw, VACNIC_BAD_SHAPE, "wgrad_group: job %ld has a null operand", (long)j);
    VCHECK(a.M > 0 && a.N > 0 && a.K > 0, VACNIC_BAD_SHAPE, "wgrad_group: job %ld is empty (M=%ld N=%ld K=%ld)", (long)j, (long)a.M, (long)a.N, (long)a.K);
    VCHECK((a.lddy & 7) == 0 && (a.ldx & 7) == 0 && aligned16(a.dy) && aligned16(a.x), VACNIC_MISALIGNED,
           "wgrad_group: job %ld: dy / x rows must be 16-byte aligned (ld %% 8 == 0)", (long)j);
    VCHECK(a.lddy >= ((a.N + 7) & ~7LL) && a.ldx >= ((a.K + 7) & ~7LL) && a.lddw >= a.K, VACNIC_BAD_SHAPE,
           "wgrad_group: job %ld: leading dimension too small", (long)j);
    VCHECK((a.lddw & 3) == 0 && aligned16(a.dw), VACNIC_MISALIGNED, "wgrad_group: job %ld: dw rows must be 16-byte aligned", (long)j);
    const int64_t xb = span(a.M, (a.N + 7) & ~7LL, a.lddy), wb = span(a.M, (a.K + 7) & ~7LL, a.ldx);
    VCHECK(xb < 0x7ffffff0LL && wb < 0x7ffffff0LL, VACNIC_UNSUPPORTED, "wgrad_group: job %ld: operand larger than 2 GiB", (long)j);
    for (int64_t um0 = 0; um0 < a.N; um0 += US)
      for (int64_t un0 = 0; un0 < a.K; un0 += US) {
        GroupUnit& u = g.u[g.nunits++];
        u.x = (const bf16_t*)a.dy; u.w = (const bf16_t*)a.x; u.out = a.dw;
        u.xsum = un0 == 0 ? a.dbias : nullptr;        // the bias gradient is formed by the first column block's workgroups only
        u.M = (int)a.N; u.N = (int)a.K; u.K = (int)a.M;
        u.ldx = (int)a.lddy; u.ldw = (int)a.ldx; u.ldo = (int)a.lddw;
        u.um0 = (int)um0; u.un0 = (int)un0;
        u.x_bytes = (unsigned)xb; u.w_bytes = (unsigned)wb;
        if (g.nunits == GROUP_MAX_UNITS) { if (int e = flush()) return e; }
      }
  }
  return flush();
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused LM head + cross entropy (MFULL:1885,1997 lm_head; TRAIN:287,816 CrossEntropyLoss(ignore_index=pad)): the
// [R, V] logits are never written.  Forward = one GEMM whose epilogue reduces every 256-column tile of a row to an
// online-softmax pair and picks the target's logit, plus a small combine; backward recomputes the logits one vocabulary
// chunk at a time straight into bf16 dlogits (vacnic_lmhead_ce_dlogits), which the caller feeds to the dh / dE GEMMs.
namespace {
__global__ __launch_bounds__(256) void ce_combine_kernel(const float* __restrict__ part, const float* __restrict__ tl,
                                                          const int64_t* __restrict__ targets, float* __restrict__ row_lse,
                                                          float* __restrict__ loss_sum, float* __restrict__ count, int R, int tiles,
                                                          int64_t ignore) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  float loss = 0.f, cnt = 0.f;
  if (r < R) {
    float mx = -INFINITY, sm = 0.f;
    for (int t = lane; t < tiles; t += 64) {
      const float pm = part[((size_t)r * tiles + t) * 2], ps = part[((size_t)r * tiles + t) * 2 + 1];
      if (pm > -INFINITY) {
        const float nm = fmaxf(mx, pm);
        sm = (mx == -INFINITY ? 0.f : sm * __expf(mx - nm)) + ps * __expf(pm - nm);
        mx = nm;
      }
    }
    const float gm = wave_max(mx);
    const float gs = wave_sum(mx == -INFINITY ? 0.f : sm * __expf(mx - gm));
    const float lse = gm + __logf(gs);
    if (lane == 0) {
      row_lse[r] = lse;
      if (targets[r] != ignore) { loss = lse - tl[r]; cnt = 1.f; }
    }
  }
  __shared__ float red[2][4];
  if (lane == 0) { red[0][wave] = loss; red[1][wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float l = red[0][0] + red[0][1] + red[0][2] + red[0][3], c = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (c > 0.f) { atomicAdd(loss_sum, l); atomicAdd(count, c); }
  }
}

__global__ void ce_rowp_kernel(const float* __restrict__ row_lse, const int64_t* __restrict__ targets, const float* __restrict__ count,
                               const float* __restrict__ grad_out, float grad_scale, float* __restrict__ rowp, int R, int64_t ignore) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float c = fmaxf(*count, 1.f);
  const float g = (grad_out ? *grad_out : 1.f) * grad_scale / c;
  rowp[2 * r] = row_lse[r];
  rowp[2 * r + 1] = targets[r] != ignore ? g : 0.f;
}
}  // namespace

static void fill_lmhead_gemm(vacnic_gemm_args& g, const vacnic_lmhead_ce_args* a, const void* emb, int64_t N) {
  g = vacnic_gemm_args{};
  g.x = a->h; g.w = emb; g.bias = a->bias;
  g.M = a->R; g.N = N; g.K = a->D;
  g.ldx = a->ldh; g.ldw = a->lde;
  g.alpha = 1.0f; g.split_k = 1;
}

extern "C" int vacnic_lmhead_ce_fwd(const vacnic_lmhead_ce_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_lmhead_ce_fwd, a, stream);
  VCHECK(a && a->h && a->emb && a->targets && a->part && a->tl && a->row_lse && a->loss_sum && a->count, VACNIC_BAD_SHAPE,
         "lmhead_ce_fwd: null operand");
  VCHECK(a->R > 0 && a->V > 0 && a->D > 0, VACNIC_BAD_SHAPE, "lmhead_ce_fwd: empty problem");
  const int64_t tiles = (a->V + 255) / 256;
  VCHECK(a->part_tiles >= tiles, VACNIC_BAD_SHAPE, "lmhead_ce_fwd: part holds %ld tiles per row, %ld needed", (long)a->part_tiles, (long)tiles);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(a->loss_sum, 0, sizeof(float), st) != hipSuccess || hipMemsetAsync(a->count, 0, sizeof(float), st) != hipSuccess) {
    vacnic_set_error("lmhead_ce_fwd: memset failed");
    return VACNIC_HIP_ERROR;
  }
  vacnic_gemm_args g;
  fill_lmhead_gemm(g, a, a->emb, a->V);
  g.out = a->part;                       // never written through `out`; the epilogue uses preact / xsum / dact_src
  g.ldo = a->V;
  g.preact = a->part; g.xsum = a->tl; g.dact_src = a->targets;
  if (a->part_tiles != tiles) {          // the kernel indexes part[m][tiles_n] with ITS tile count
    vacnic_set_error("lmhead_ce_fwd: part_tiles must equal ceil(V / 256) = %ld", (long)tiles);
    return VACNIC_BAD_SHAPE;
  }
  if (int e = gemm_one(&g, 256 + 64000, stream, 3, 0)) return e;
  hipLaunchKernelGGL(ce_combine_kernel, dim3((unsigned)((a->R + 3) / 4)), dim3(256), 0, st, a->part, a->tl, a->targets, a->row_lse,
                     a->loss_sum, a->count, (int)a->R, (int)tiles, a->ignore_index);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_lmhead_ce_rowp(const float* row_lse, const int64_t* targets, const float* count, const float* grad_out,
                                     float grad_scale, float* rowp, int64_t R, int64_t ignore_index, void* stream) {
  VPLAN_REC(vacnic_lmhead_ce_rowp, row_lse, targets, count, grad_out, grad_scale, rowp, R, ignore_index, stream);
  VCHECK(row_lse && targets && count && rowp && R > 0, VACNIC_BAD_SHAPE, "lmhead_ce_rowp: bad operand");
  hipLaunchKernelGGL(ce_rowp_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, row_lse, targets, count,
                     grad_out, grad_scale, rowp, (int)R, ignore_index);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_lmhead_ce_dlogits(const vacnic_lmhead_ce_args* a, int64_t col0, int64_t ncols, void* dl, int64_t lddl,
                                        const float* rowp, void* stream) {
  VPLAN_REC_STRUCT(vacnic_lmhead_ce_dlogits, a, col0, ncols, dl, lddl, rowp, stream);
  VCHECK(a && a->h && a->emb && a->targets && dl && rowp, VACNIC_BAD_SHAPE, "lmhead_ce_dlogits: null operand");
  VCHECK(col0 >= 0 && ncols > 0 && col0 + ncols <= a->V, VACNIC_BAD_SHAPE, "lmhead_ce_dlogits: chunk [%ld, +%ld) outside V=%ld",
         (long)col0, (long)ncols, (long)a->V);
  VCHECK((lddl & 7) == 0 && lddl >= ((ncols + 7) & ~7LL) && aligned16(dl), VACNIC_MISALIGNED, "lmhead_ce_dlogits: dl rows must be 16-byte aligned and hold round_up(ncols, 8)");
  vacnic_gemm_args g;
  fill_lmhead_gemm(g, a, (const char*)a->emb + col0 * a->lde * 2, ncols);
  if (g.bias) g.bias = a->bias + col0;
  g.out = dl; g.ldo = lddl;
  g.dact_src = a->targets; g.residual = rowp;
  return gemm_one(&g, 256 + 64000, stream, 4, (int)col0);
}
