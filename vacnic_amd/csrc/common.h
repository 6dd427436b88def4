// Shared device/host helpers for the gfx950 kernels of libvacnic_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/vacnic_hip.h"

typedef unsigned short bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---- error plumbing (host) -------------------------------------------------------------------
void vacnic_set_error(const char* fmt, ...);
#define VCHECK(cond, code, ...)                      \
  do {                                               \
    if (!(cond)) {                                   \
      vacnic_set_error(__VA_ARGS__);                 \
      return (code);                                 \
    }                                                \
  } while (0)
#define VLAUNCH_CHECK()                                                        \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      vacnic_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,          \
                       hipGetErrorString(e__));                                \
      return VACNIC_HIP_ERROR;                                                 \
    }                                                                          \
  } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- launch plans (capi.hip): every C-ABI entry point can be RECORDED — its arguments frozen in a closure — while it runs, and
// the recorded sequence replayed later from C++ with one call (vacnic_plan_replay): the host cost of a training step drops from
// ~1400 Python -> ctypes round trips to one.  Entry points call VPLAN_REC / VPLAN_REC_STRUCT first; a nested entry point (one
// C-ABI function calling another) is recorded once, by the outermost call.
#include <functional>
namespace vplan {
bool active();                                   // a plan is being recorded BY THIS THREAD (the one that called vacnic_plan_begin)
void push(std::function<int()> f);
size_t size();                                   // commands recorded so far
void truncate(size_t n);                         // forget the commands recorded after the first n
extern thread_local int depth;
extern thread_local bool failed;                 // vacnic_set_error was called inside the outermost entry point in progress
// An entry point records its closure on entry and runs; if it then reports an error (every error path goes through
// vacnic_set_error), the outermost Guard takes the command back, so a plan never holds a call that failed while recording.
struct Guard {
  size_t mark;
  Guard() : mark(0) { if (++depth == 1) { failed = false; mark = active() ? size() : 0; } }
  ~Guard() { if (--depth == 0 && failed && active()) truncate(mark); }
};
inline bool outermost() { return depth == 1 && active(); }
}  // namespace vplan
#define VPLAN_REC(fn, ...)                                                   \
  vplan::Guard vplan_guard__;                                               \
  if (vplan::outermost()) vplan::push([=]() { return fn(__VA_ARGS__); })
#define VPLAN_REC_STRUCT(fn, a, ...)                                         \
  vplan::Guard vplan_guard__;                                               \
  if (vplan::outermost() && (a)) {                                          \
    auto copy__ = *(a);                                                     \
    vplan::push([=]() { return fn(&copy__, __VA_ARGS__); });                \
  }

// ---- bf16 <-> f32 ------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// f32 -> bf16, round-to-nearest-even, NaN stays NaN: ONE v_cvt_pk_bf16_f32 per two values (gfx950).  The integer
// rounding idiom costs ~7 VALU instructions per element, which made the GEMM's LDS-staged epilogue take ~8 us per 256x256
// tile (profiles/r1_gemm_overhead.txt) — more than the first-tile load and the launch together.
// (written as a __bf16 vector conversion, not inline asm: the compiler then also pads the MFMA-result -> VALU hazards)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_native;
typedef __attribute__((ext_vector_type(2))) float f32x2_native;
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  const f32x2_native f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_native));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

// ---- wave64 reductions ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- activations -----------------------------------------------------------------------------------
// Epilogue-grade transcendental forms (v_exp_f32 / v_rcp_f32 based, no libm calls): the GEMM epilogue
// applies them to every output element, so a libm erff/tanhf there costs more than the MFMA main loop
// at K = 1024.  erf: Abramowitz-Stegun 7.1.26, |abs err| <= 1.5e-7 (three orders below bf16 rounding).
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// returns erf(|u|) and e = exp(-u*u) (shared with the Gaussian pdf of the GELU derivative)
__device__ __forceinline__ float erf_abs(float au, float& e) {
  const float t = fast_rcp(1.0f + 0.3275911f * au);
  e = __expf(-au * au);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  return 1.0f - poly * e;
}
__device__ __forceinline__ float gelu_fwd(float x) {
  float e;
  const float er = erf_abs(fabsf(x) * 0.70710678118654752f, e);
  return 0.5f * x * (1.0f + copysignf(er, x));
}
__device__ __forceinline__ float gelu_bwd(float x) {
  float e;                                            // e = exp(-x^2/2)
  const float er = erf_abs(fabsf(x) * 0.70710678118654752f, e);
  return 0.5f * (1.0f + copysignf(er, x)) + x * 0.3989422804014327f * e;
}
__device__ __forceinline__ float tanh_fast(float x) {
  const float e2 = __expf(-2.0f * fabsf(x));          // in (0, 1]: no overflow
  return copysignf((1.0f - e2) * fast_rcp(1.0f + e2), x);
}
__device__ __forceinline__ float act_fwd(int act, float x) {
  switch (act) {
    case VACNIC_ACT_GELU: return gelu_fwd(x);
    case VACNIC_ACT_TANH: return tanh_fast(x);
    case VACNIC_ACT_QUICKGELU: return x * fast_rcp(1.0f + __expf(-1.702f * x));
    default: return x;
  }
}
// derivative w.r.t. the pre-activation x
__device__ __forceinline__ float act_bwd(int act, float x) {
  switch (act) {
    case VACNIC_ACT_GELU: return gelu_bwd(x);
    case VACNIC_ACT_TANH: { float t = tanh_fast(x); return 1.0f - t * t; }
    case VACNIC_ACT_QUICKGELU: {
      float s = fast_rcp(1.0f + __expf(-1.702f * x));
      return s + 1.702f * x * s * (1.0f - s);
    }
    default: return 1.0f;
  }
}

// ---- Philox4x32-10 counter RNG (dropout) -------------------------------------------------------------
// keep-mask for element index idx is derived from philox(key=seed, ctr=idx/4)[idx%4]; fwd and bwd
// regenerate the same bits, nothing is stored.
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// 4 consecutive elements starting at idx4*4; returns 4 uniform uint32
__device__ __forceinline__ void dropout_bits4(uint64_t seed, uint64_t idx4, uint32_t out[4]) {
  philox4x32((uint32_t)idx4, (uint32_t)(idx4 >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), out);
}
// ---- activation dropout (hidden states of the FFN blocks, MFULL:649,660,684,740,874) ------------------------------------------
// Element e of a flat contiguous array takes byte e & 15 of Philox block e >> 4 (16 keep decisions per block); keep iff
// byte >= thr, p quantised to 1/256 like the attention-probability dropout.  The GEMM epilogues (forward: after the activation;
// backward: after act') and vacnic_dropout_bf16 evaluate the same function, so nothing is stored.
__device__ __forceinline__ void actdrop_block(uint64_t seed, uint64_t blk, uint32_t w[4]) {
  philox4x32((uint32_t)blk, (uint32_t)(blk >> 32), 0x41435444u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
}
__device__ __forceinline__ void actdrop_factors8(uint32_t lo, uint32_t hi, unsigned thr, float inv, float m[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    m[e] = ((lo >> (8 * e)) & 0xffu) >= thr ? inv : 0.f;
    m[4 + e] = ((hi >> (8 * e)) & 0xffu) >= thr ? inv : 0.f;
  }
}
// Two adjacent lanes own the two 8-element halves of the same 16-element blocks, row by row: for a PAIR of rows the even lane
// generates the first row's block and the odd lane the second's, and each hands the partner its half with two quad_perm DPP
// moves — one Philox block per 16 decisions and lane.  blk_a / blk_b: block index of the first / second row (this lane's column
// group); returns the mask words {lo, hi} of this lane's 8 elements in both rows.  All lanes of a pair must call it together.
__device__ __forceinline__ void actdrop_pair(uint64_t seed, uint64_t blk_a, uint64_t blk_b, int lane, uint32_t& a_lo, uint32_t& a_hi,
                                             uint32_t& b_lo, uint32_t& b_hi) {
  const bool odd = (lane & 1) != 0;
  uint32_t w[4];
  actdrop_block(seed, odd ? blk_b : blk_a, w);
  const uint32_t s0 = odd ? w[0] : w[2], s1 = odd ? w[1] : w[3];          // the half the partner owns
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xf, 0xf, true);
  a_lo = odd ? r0 : w[0]; a_hi = odd ? r1 : w[1];
  b_lo = odd ? w[2] : r0; b_hi = odd ? w[3] : r1;
}

__device__ __forceinline__ uint32_t dropout_threshold(float p) {
  // keep iff bits >= thr  (P(keep) = 1-p)
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
}
