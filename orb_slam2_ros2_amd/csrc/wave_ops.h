// wave_ops.h -- 64-lane wave reductions / scans with DPP (gfx9 row shifts + row broadcasts), no LDS traffic.
// All 64 lanes must be active at the call site.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbfe {

#define ORBFE_DPP_ROW_SHR(n) (0x110 + (n))
#define ORBFE_DPP_ROW_BCAST15 0x142
#define ORBFE_DPP_ROW_BCAST31 0x143
#define ORBFE_DPP_WAVE_SHR1 0x138
#define ORBFE_DPP_WAVE_SHL1 0x130

// NOTE on the inline-asm helpers below: the compiler's hazard recogniser does not look inside an asm statement.  Their INPUTS must
// not come straight out of a v_dot* instruction (gfx950: three wait states before another VALU instruction may read a dot result),
// and their RESULT must not be the direct input of a DPP / readlane instruction (two wait states): pass such values through a
// compiler-generated instruction first.  (k_ic_moments: sums of v_dot4 fed to an asm v_mad_i32_i24 gave garbage moments.)
// Full-rate 24-bit integer multiplies.  hipcc lowers an int product whose operand ranges it cannot prove to the quarter-rate
// v_mul_lo_u32 (and __mul24 back to a plain product): where the operands are known to fit 24 bits, spell the instruction out.
__device__ __forceinline__ int mul24u(int a, int b) {
  int d;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ int mad24u(int a, int b, int c) {
  int d;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// signed variant (|a|, |b| < 2^23)
__device__ __forceinline__ int mad24s(int a, int b, int c) {
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
struct OpMinI {
  static __device__ __forceinline__ int id() { return 2147483647; }
  static __device__ __forceinline__ int f(int a, int b) { return a < b ? a : b; }
};
struct OpMaxI {
  static __device__ __forceinline__ int id() { return -2147483647 - 1; }
  static __device__ __forceinline__ int f(int a, int b) { return a > b ? a : b; }
};
struct OpAddI {
  static __device__ __forceinline__ int id() { return 0; }
  static __device__ __forceinline__ int f(int a, int b) { return a + b; }
};

template <class Op, int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_step(int v) {
  // lanes without a source lane keep the identity
  return Op::f(v, __builtin_amdgcn_update_dpp(Op::id(), v, CTRL, ROW_MASK, 0xf, false));
}

// inclusive scan over the 64 lanes; lane 63 holds the reduction of the whole wave
template <class Op>
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
  v = dpp_step<Op, ORBFE_DPP_ROW_SHR(1), 0xf>(v);
  v = dpp_step<Op, ORBFE_DPP_ROW_SHR(2), 0xf>(v);
  v = dpp_step<Op, ORBFE_DPP_ROW_SHR(4), 0xf>(v);
  v = dpp_step<Op, ORBFE_DPP_ROW_SHR(8), 0xf>(v);
  v = dpp_step<Op, ORBFE_DPP_ROW_BCAST15, 0xa>(v);
  v = dpp_step<Op, ORBFE_DPP_ROW_BCAST31, 0xc>(v);
  return v;
}
template <class Op>
__device__ __forceinline__ int wave_reduce_dpp(int v) {
  return __builtin_amdgcn_readlane(wave_incl_scan_dpp<Op>(v), 63);
}

// Sum of a double over the 64 lanes in a FIXED tree (deterministic), the same value returned to every lane.  Data-parallel-primitive moves
// instead of __shfl_xor: a shuffle of a double is two ds_bpermute_b32 through the LDS crossbar; here a step is two full-rate register moves and
// an add.  Tree: an inclusive scan inside each row of 16 (row_shr 1, 2, 4, 8: lane 15 of a row = pairwise tree over its lanes), then
// (row 1 + row 0), (row 3 + row 2), and their sum in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (unsigned long long)(uint32_t)lo));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<ORBFE_DPP_ROW_SHR(1), 0xf>(v);  // (lanes without a source add the +0.0 of `old`)
  v += dpp_f64<ORBFE_DPP_ROW_SHR(2), 0xf>(v);
  v += dpp_f64<ORBFE_DPP_ROW_SHR(4), 0xf>(v);
  v += dpp_f64<ORBFE_DPP_ROW_SHR(8), 0xf>(v);
  v += dpp_f64<ORBFE_DPP_ROW_BCAST15, 0xa>(v);
  v += dpp_f64<ORBFE_DPP_ROW_BCAST31, 0xc>(v);
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

}  // namespace orbfe
