// k_lmbig.hip -- the reduced system of the local BA for windows past the register-resident solver of k_lm.hip (more than
// LM_CHOL_MAX_NB free keyframes): a blocked, multi-workgroup Cholesky with the trailing updates on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64), gated by the device-side Levenberg-Marquardt state exactly like k_lm_chol, so that a window of any size
// keeps ONE enqueue and ONE synchronisation per call.
//
// The reference puts every covisible keyframe into the free set (Optimizer.cc:232-233, no bound) and solves the reduced system with
// g2o's BlockSolver_6_3 over a sparse Cholesky (Optimizer.h:49-54).  Here the reduced system is dense: at 300 free keyframes an
// 1800 x 1800 factorisation, ~1.9 GFLOP per trial -- the dense contraction north_star reserves the matrix cores for.
//
// Layout: M = (ld + 48) x ld doubles, row-major, ld = 6 nf rounded up to 48.  Rows / columns [6 nf, ld) are padding (unit diagonal);
// row ld carries the right-hand side, so the forward substitution is part of the factorisation (an extra tile row), the other rows of
// that tile row are zero.  Only the lower triangle is read.
//
// One launch per tile column (48 x 48 tiles, k = -1 .. KT - 2): workgroup (i, j), k < j <= i, does
//   A_ij -= L_ik L_jk^T                                  three waves, a 16-row strip each, 3 x 12 MFMA 16x16x4 per strip
// and the workgroups of column j = k + 1 go on to finish THAT column: each updates and factorises the column's diagonal tile in
// 16-column blocks (one wave factorises and inverts a 16 x 16 diagonal block, rows in lanes, the column exchanged by v_readlane; the
// blocks below it and the rank-16 updates run on the matrix cores) and solves X L^T = A for its own 48 rows by blocked substitution,
// X_b = (A_b - sum X_c L_bc^T) inv_bb^T, every product an MFMA (a column-at-a-time substitution by one wave was 14 us a tile, the
// 48-column factorisation 24 us: tools/exp/lmb_bench.hip).  No workgroup waits for another.  After the last launch L and y are
// complete; k_lmb_back_mw solves L^T x = y with one workgroup per tile column (k_lmb_back: the one-workgroup version, kept for tools/exp/lmb_bench.hip).
// Fixed summation order everywhere: run-to-run identical.  A non-positive pivot clears LmState::ok (g2o: the linear solver fails, the
// trial is rejected) and x = 0.
#include <hip/hip_runtime.h>

#include "orbfe_internal.h"

namespace orbfe {

#define LMB_T 48
#define LMB_LS 49  // LDS row pitch of a tile in doubles (odd: the 16 rows a wave's operand read touches fall on distinct banks)

typedef double lmb_d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool lmb_gate(const LmState* st) { return *(const volatile int32_t*)&st->run_step != 0; }

// pad diagonal of M (once per call, after the zero fill)
__global__ __launch_bounds__(64) void k_lmb_init(double* __restrict__ M, int n, int ld) {
  const int i = n + (int)threadIdx.x;
  if (i < ld) M[(size_t)i * ld + i] = 1.0;
}

#define LMB_IS 17  // pitch of a 16 x 16 inverse in LDS
struct LmbShared {
  double Li[LMB_T * LMB_LS];  // L_ik, later the factorised diagonal tile of the column for the panel solves
  double Lj[LMB_T * LMB_LS];  // L_jk
  double C[LMB_T * LMB_LS];   // the updated tile on its way to the factorisation / solve
  double Inv[3][16 * LMB_IS]; // inverses of the three 16 x 16 diagonal blocks of the diagonal tile (lower triangular)
  int fail;
};

// ---- 16 x 16 fragments on the matrix core --------------------------------------------------------------------------------------------
// v_mfma_f64_16x16x4_f64: A operand lane l = A[l & 15][l >> 4], B operand lane l = B[l >> 4][l & 15], C / D register rg of lane l =
// element (row (l >> 4) + 4 rg, column l & 15).  All products here are X Y^T with X and Y row-major in LDS (k along the row), so both
// operands are read the same way.
__device__ __forceinline__ lmb_d4 lmb_frag_load(const double* P, int pitch, int lane) {
  lmb_d4 f;
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) f[rg] = P[((lane >> 4) + 4 * rg) * pitch + (lane & 15)];
  return f;
}
__device__ __forceinline__ void lmb_frag_store(double* P, int pitch, int lane, lmb_d4 f) {
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) P[((lane >> 4) + 4 * rg) * pitch + (lane & 15)] = f[rg];
}
// acc +- X[0..15][0..15] Y[0..15][0..15]^T
template <bool NEG>
__device__ __forceinline__ lmb_d4 lmb_mm_abt(lmb_d4 acc, const double* X, int px, const double* Y, int py, int lane) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const double a = X[(lane & 15) * px + 4 * s + (lane >> 4)];
    const double b = Y[(lane & 15) * py + 4 * s + (lane >> 4)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -a : a, b, acc, 0, 0, 0);
  }
  return acc;
}
__device__ __forceinline__ void lmb_wave_sync() {  // a wave's own LDS traffic executes in order: this only pins the compiler
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Cholesky of the 16 x 16 block at D (pitch LMB_LS, lower triangle, in place) and its inverse -> inv (pitch LMB_IS), by ONE wave: lanes
// 0..15 own a row each, the finished column travels by v_readlane (fifteen scalar pairs live at most).  Pivot: 1 / sqrt by v_rsq_f64
// and one third-order correction, L_cc = d y, as in k_lm_chol.  Returns false on a bad pivot.
__device__ __forceinline__ bool lmb_potrf16(double* D, double* inv, int lane) {
#pragma clang fp contract(fast)
  const int r = lane & 15;
  double a[16], y[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = D[r * LMB_LS + c];
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[c]), c), __builtin_amdgcn_readlane(__double2loint(a[c]), c));
    if (!(piv > 0.0)) ok = false;  // (uniform)
    const double d = piv > 0.0 ? piv : 1.0;
    const double y0 = __builtin_amdgcn_rsq(d);
    const double e = fma(-(d * y0), y0, 1.0);
    y[c] = fma(y0 * e, fma(0.375, e, 0.5), y0);
    const double l = a[c] * y[c];  // column c of L, lane = row: lane c's a[c] IS the pivot (rows above c: garbage, never read)
    a[c] = l;
#pragma unroll
    for (int j = c + 1; j < 16; ++j) {
      const double lj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(l), j), __builtin_amdgcn_readlane(__double2loint(l), j));
      a[j] = fma(-l, lj, a[j]);  // (lanes r < j update the upper triangle: never read)
    }
  }
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; ++c) D[r * LMB_LS + c] = c <= r ? a[c] : 0.0;
  }
  lmb_wave_sync();
  // inverse: lane j solves L x = e_j (column j of the inverse), L read back as wave-uniform (broadcast) LDS reads
  double x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = i == r ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double xk = x[k] * y[k];
    x[k] = xk;
#pragma unroll
    for (int i = k + 1; i < 16; ++i) x[i] = fma(-D[i * LMB_LS + k], xk, x[i]);
  }
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) inv[i * LMB_IS + r] = i >= r ? x[i] : 0.0;
  }
  return ok;
}

// The diagonal tile in sh.C (lower triangle) -> L in place, the inverses of its three diagonal blocks in sh.Inv.  Three waves, 16-column
// blocks: factor the block (one wave), L_below = A_below inv^T and the rank-16 updates of what is left on the matrix cores.
__device__ __forceinline__ void lmb_potrf_tile(LmbShared& sh, int wv, int lane) {
  double* C = sh.C;
  // block column 0
  if (wv == 0 && !lmb_potrf16(C, sh.Inv[0], lane)) sh.fail = 1;
  __syncthreads();
  if (wv > 0) {  // L_w0 = A_w0 inv11^T
    lmb_d4 f = {0, 0, 0, 0};
    f = lmb_mm_abt<false>(f, C + 16 * wv * LMB_LS, LMB_LS, sh.Inv[0], LMB_IS, lane);
    lmb_wave_sync();
    lmb_frag_store(C + 16 * wv * LMB_LS, LMB_LS, lane, f);
  }
  __syncthreads();
  if (wv == 1) {  // A22 -= L21 L21^T, then its factorisation
    lmb_d4 f = lmb_frag_load(C + 16 * LMB_LS + 16, LMB_LS, lane);
    f = lmb_mm_abt<true>(f, C + 16 * LMB_LS, LMB_LS, C + 16 * LMB_LS, LMB_LS, lane);
    lmb_frag_store(C + 16 * LMB_LS + 16, LMB_LS, lane, f);
    lmb_wave_sync();
    if (!lmb_potrf16(C + 16 * LMB_LS + 16, sh.Inv[1], lane)) sh.fail = 1;
  } else if (wv == 2) {  // A32 -= L31 L21^T, A33 -= L31 L31^T
    lmb_d4 f = lmb_frag_load(C + 32 * LMB_LS + 16, LMB_LS, lane);
    f = lmb_mm_abt<true>(f, C + 32 * LMB_LS, LMB_LS, C + 16 * LMB_LS, LMB_LS, lane);
    lmb_d4 g = lmb_frag_load(C + 32 * LMB_LS + 32, LMB_LS, lane);
    g = lmb_mm_abt<true>(g, C + 32 * LMB_LS, LMB_LS, C + 32 * LMB_LS, LMB_LS, lane);
    lmb_frag_store(C + 32 * LMB_LS + 16, LMB_LS, lane, f);
    lmb_frag_store(C + 32 * LMB_LS + 32, LMB_LS, lane, g);
  }
  __syncthreads();
  if (wv == 2) {  // L32 = A32 inv22^T, A33 -= L32 L32^T, factor
    lmb_d4 f = {0, 0, 0, 0};
    f = lmb_mm_abt<false>(f, C + 32 * LMB_LS + 16, LMB_LS, sh.Inv[1], LMB_IS, lane);
    lmb_wave_sync();
    lmb_frag_store(C + 32 * LMB_LS + 16, LMB_LS, lane, f);
    lmb_wave_sync();
    lmb_d4 g = lmb_frag_load(C + 32 * LMB_LS + 32, LMB_LS, lane);
    g = lmb_mm_abt<true>(g, C + 32 * LMB_LS + 16, LMB_LS, C + 32 * LMB_LS + 16, LMB_LS, lane);
    lmb_frag_store(C + 32 * LMB_LS + 32, LMB_LS, lane, g);
    lmb_wave_sync();
    if (!lmb_potrf16(C + 32 * LMB_LS + 32, sh.Inv[2], lane)) sh.fail = 1;
  }
  __syncthreads();
}

// X L^T = A for the 16 rows at R (pitch LMB_LS), in place: L = the column's factorised diagonal tile (row-major, pitch LMB_LS), the inverses
// of its diagonal blocks in sh.Inv.  Blocked forward substitution, every product on the matrix core; a wave touches its own rows only.
__device__ __forceinline__ void lmb_trsm_strip(double* R, const double* L, LmbShared& sh, int lane) {
  lmb_d4 x = {0, 0, 0, 0};
  x = lmb_mm_abt<false>(x, R, LMB_LS, sh.Inv[0], LMB_IS, lane);  // X1 = A1 inv11^T
  lmb_d4 a2 = lmb_frag_load(R + 16, LMB_LS, lane), a3 = lmb_frag_load(R + 32, LMB_LS, lane);
  lmb_wave_sync();
  lmb_frag_store(R, LMB_LS, lane, x);
  lmb_wave_sync();
  a2 = lmb_mm_abt<true>(a2, R, LMB_LS, L + 16 * LMB_LS, LMB_LS, lane);  // A2 -= X1 L21^T
  a3 = lmb_mm_abt<true>(a3, R, LMB_LS, L + 32 * LMB_LS, LMB_LS, lane);  // A3 -= X1 L31^T
  lmb_frag_store(R + 16, LMB_LS, lane, a2);
  lmb_wave_sync();
  x = lmb_d4{0, 0, 0, 0};
  x = lmb_mm_abt<false>(x, R + 16, LMB_LS, sh.Inv[1], LMB_IS, lane);  // X2 = A2 inv22^T
  lmb_wave_sync();
  lmb_frag_store(R + 16, LMB_LS, lane, x);
  lmb_wave_sync();
  a3 = lmb_mm_abt<true>(a3, R + 16, LMB_LS, L + 32 * LMB_LS + 16, LMB_LS, lane);  // A3 -= X2 L32^T
  lmb_frag_store(R + 32, LMB_LS, lane, a3);
  lmb_wave_sync();
  x = lmb_d4{0, 0, 0, 0};
  x = lmb_mm_abt<false>(x, R + 32, LMB_LS, sh.Inv[2], LMB_IS, lane);  // X3 = A3 inv33^T
  lmb_wave_sync();
  lmb_frag_store(R + 32, LMB_LS, lane, x);
}

// One tile column.  k: the column whose panel is final (-1: none yet); the launch finishes column k + 1.
// EVERY workgroup of column k + 1 updates and factorises the column's diagonal tile itself (the same instructions on the same inputs:
// the same bits) instead of waiting for the one workgroup that owns it: the owner's store -> flag -> poll -> load chain was ~9 us of
// global round trips per column on top of the 10 us factorisation, the other CUs are idle anyway, and nothing in the kernel waits on
// another workgroup.  Only the owner writes the tile and its block inverses back (the back substitution reads them).
__global__ __launch_bounds__(192) void k_lmb_step(int k, int KT, int ld, double* __restrict__ M, double* __restrict__ Linv, LmState* __restrict__ st,
                                                  int32_t* __restrict__ flags) {
  __shared__ LmbShared sh;
  if (!lmb_gate(st)) return;
  // an earlier column of THIS factorisation hit a bad pivot.  (Not looked at by the first launch, k = -1: there is no earlier column, and
  // the flag may still be up from the PREVIOUS factorisation -- it is cleared below, in stream order, by the one workgroup that may raise
  // it again in this launch.)
  if (k >= 0 && *(volatile int32_t*)&flags[KT] != 0) return;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (k < 0 && blockIdx.x == 0) {
    // the state of the PREVIOUS factorisation, cleared here and nowhere else: the back substitution's column flags (k_lmb_back_mw polls
    // them) and the bad-pivot flag (k_lmb_back_mw's workgroups read it at their start; one of them clearing it for the others was a
    // window for a late workgroup to take the normal path and wait for columns that never publish -- ADVICE r4).  Workgroup 0 is also
    // the owner of column 0's diagonal tile, the only one that can raise the flag in this launch: same thread, program order.
    for (int q = t; q < KT; q += 192) flags[KT + 1 + q] = 0;
    if (t == 0) flags[KT] = 0;
  }
  // blockIdx.x -> (i, j): column-major over the trailing tiles, column k + 1 first
  const int m = KT - 1 - k;  // tile columns left; tile rows k + 1 .. KT (KT: the right-hand-side row)
  int jq = 0, base = 0;
  while (jq < m - 1 && base + (m + 1 - jq) <= (int)blockIdx.x) {
    base += m + 1 - jq;
    ++jq;
  }
  const int j = k + 1 + jq, i = j + ((int)blockIdx.x - base);
  double* Aij = M + (size_t)i * LMB_T * ld + (size_t)j * LMB_T;
  double* Ajj = M + (size_t)j * LMB_T * ld + (size_t)j * LMB_T;
  const bool diag = i == j, column = jq == 0;
  lmb_d4 acc[3], dacc[3];
  // C fragment of strip wv, sub-tile b: row = 16 wv + (lane >> 4) + 4 reg, col = 16 b + (lane & 15)
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const size_t o = (size_t)(16 * wv + (lane >> 4) + 4 * rg) * ld + 16 * b + (lane & 15);
      acc[b][rg] = Aij[o];
      dacc[b][rg] = (column && !diag) ? Ajj[o] : 0.0;
    }
  if (k >= 0) {
    const double* Lik = M + (size_t)i * LMB_T * ld + (size_t)k * LMB_T;
    const double* Ljk = M + (size_t)j * LMB_T * ld + (size_t)k * LMB_T;
    for (int e = t; e < LMB_T * LMB_T; e += 192) {
      const int r = e / LMB_T, c = e - r * LMB_T;
      sh.Li[r * LMB_LS + c] = Lik[(size_t)r * ld + c];
      sh.Lj[r * LMB_LS + c] = Ljk[(size_t)r * ld + c];
    }
    __syncthreads();
    const int row = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int s = 0; s < LMB_T / 4; ++s) {
      const double a = -sh.Li[(16 * wv + row) * LMB_LS + 4 * s + kq];
      const double aj = -sh.Lj[(16 * wv + row) * LMB_LS + 4 * s + kq];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const double bv = sh.Lj[(16 * b + row) * LMB_LS + 4 * s + kq];
        if (!(diag && b > wv)) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[b], 0, 0, 0);  // (only the lower triangle of a diagonal tile is ever read)
        if (column && !diag && b <= wv) dacc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj, bv, dacc[b], 0, 0, 0);
      }
    }
  }
  if (!column) {  // an interior tile of the trailing matrix: done
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) Aij[(size_t)(16 * wv + (lane >> 4) + 4 * rg) * ld + 16 * b + (lane & 15)] = acc[b][rg];
    return;
  }
  // ---- column k + 1: the diagonal tile -> sh.C (factorised there), the own tile of a workgroup below it -> sh.Li ----
  __syncthreads();  // (the operand tiles are dead)
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int o = (16 * wv + (lane >> 4) + 4 * rg) * LMB_LS + 16 * b + (lane & 15);
      sh.C[o] = diag ? acc[b][rg] : dacc[b][rg];
      if (!diag) sh.Li[o] = acc[b][rg];
    }
  if (t == 0) sh.fail = 0;
  __syncthreads();
  lmb_potrf_tile(sh, wv, lane);
  if (diag) {
    for (int e = t; e < LMB_T * LMB_T; e += 192) {
      const int r = e / LMB_T, c = e - r * LMB_T;
      Aij[(size_t)r * ld + c] = sh.C[r * LMB_LS + c];
    }
    for (int e = t; e < 3 * 256; e += 192) Linv[(size_t)j * 768 + e] = sh.Inv[e >> 8][((e & 255) >> 4) * LMB_IS + (e & 15)];
    if (t == 0 && sh.fail) {
      st->ok = 0;
      flags[KT] = 1;
    }
    return;
  }
  if (sh.fail) return;
  lmb_trsm_strip(sh.Li + 16 * wv * LMB_LS, sh.C, sh, lane);
  __syncthreads();
  for (int e = t; e < LMB_T * LMB_T; e += 192) {
    const int r = e / LMB_T, c = e - r * LMB_T;
    Aij[(size_t)r * ld + c] = sh.Li[r * LMB_LS + c];
  }
}

// L^T x = y (y = row ld of M after the factorisation), one workgroup; resets the column flags for the next trial.
// Per tile column, from the last: x_k = L_kk^-T y_k by blocked backward substitution with the three 16 x 16 inverses the factorisation
// left behind (five 16 x 16 matrix-vector products, sixteen lanes each), then y_j -= L_kj^T x_k for every column left of the tile
// (one column per thread, its 48 loads in flight at once); the next diagonal tile is requested before the panel update and lands under it.
// (First version: a 48-step substitution straight from global memory, 1.75 ms at 300 keyframes; from LDS 0.40 ms.)
__global__ __launch_bounds__(1024) void k_lmb_back(int n, int KT, int ld, const double* __restrict__ M, const double* __restrict__ Linv,
                                                   LmState* __restrict__ st, int32_t* __restrict__ flags, double* __restrict__ x) {
#pragma clang fp contract(fast)
  extern __shared__ double lmb_y[];  // [ld] y -> x | [48] x of the current tile | [48][49] its diagonal tile | [3][16][17] the inverses
  if (!lmb_gate(st)) return;
  const int t = threadIdx.x;
  const bool failed = *(volatile int32_t*)&flags[KT] != 0;
  if (failed) {
    for (int c = t; c < n; c += 1024) x[c] = 0.0;
    __syncthreads();
    for (int q = t; q <= KT; q += 1024) flags[q] = 0;
    return;
  }
  double* xs = lmb_y + ld;
  double* Lt = xs + LMB_T;
  double* Iv = Lt + LMB_T * LMB_LS;
  for (int c = t; c < ld; c += 1024) lmb_y[c] = M[(size_t)ld * ld + c];
  // (tile elements of thread t: e = t, t + 1024, t + 2048 of 2304; inverse elements e = t of 768)
  auto tile_fetch = [&](int kt, double (&v)[3], double& iv) {
    const double* Lkk = M + (size_t)kt * LMB_T * ld + (size_t)kt * LMB_T;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = t + 1024 * u;
      v[u] = e < LMB_T * LMB_T ? Lkk[(size_t)(e / LMB_T) * ld + e % LMB_T] : 0.0;
    }
    iv = t < 768 ? Linv[(size_t)kt * 768 + t] : 0.0;
  };
  auto tile_store = [&](const double (&v)[3], double iv) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e = t + 1024 * u;
      if (e < LMB_T * LMB_T) Lt[(e / LMB_T) * LMB_LS + e % LMB_T] = v[u];
    }
    if (t < 768) Iv[(t >> 8) * 16 * LMB_IS + ((t & 255) >> 4) * LMB_IS + (t & 15)] = iv;
  };
  double tv[3], tiv;
  tile_fetch(KT - 1, tv, tiv);
  tile_store(tv, tiv);
  __syncthreads();
  for (int kt = KT - 1; kt >= 0; --kt) {
    if (kt > 0) tile_fetch(kt - 1, tv, tiv);  // lands under this step's work
    if (t < 64) {
      // out[i] = sum_k Mx[k][i] v[k] over a 16 x 16 block (transposed product), lanes 0..15
      const int i = t & 15;
      double* yk = lmb_y + kt * LMB_T;
      auto mv_t = [&](const double* Mx, int pitch, const double* v) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < 16; k2 += 2) {
          s0 = fma(Mx[k2 * pitch + i], v[k2], s0);
          s1 = fma(Mx[(k2 + 1) * pitch + i], v[k2 + 1], s1);
        }
        return s0 + s1;
      };
      const double x3 = mv_t(Iv + 2 * 16 * LMB_IS, LMB_IS, yk + 32);
      if (t < 16) xs[32 + i] = x3;
      lmb_wave_sync();
      const double y2 = yk[16 + i] - mv_t(Lt + 32 * LMB_LS + 16, LMB_LS, xs + 32);
      lmb_wave_sync();
      if (t < 16) yk[16 + i] = y2;
      lmb_wave_sync();
      const double x2 = mv_t(Iv + 16 * LMB_IS, LMB_IS, yk + 16);
      if (t < 16) xs[16 + i] = x2;
      lmb_wave_sync();
      const double y1 = yk[i] - (mv_t(Lt + 32 * LMB_LS, LMB_LS, xs + 32) + mv_t(Lt + 16 * LMB_LS, LMB_LS, xs + 16));
      lmb_wave_sync();
      if (t < 16) yk[i] = y1;
      lmb_wave_sync();
      const double x1 = mv_t(Iv, LMB_IS, yk);
      if (t < 16) {
        xs[i] = x1;
        yk[i] = x1, yk[16 + i] = x2, yk[32 + i] = x3;
      }
    }
    __syncthreads();
    // y_j -= L_kj^T x_k for every column left of the tile: row panel kt of L, one column per thread
    for (int c = t; c < kt * LMB_T; c += 1024) {
      const double* col = M + (size_t)kt * LMB_T * ld + c;
      double lv[LMB_T];
#pragma unroll
      for (int r = 0; r < LMB_T; ++r) lv[r] = col[(size_t)r * ld];  // (all requested before the first is used: one round trip)
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int r = 0; r < LMB_T; r += 4) {
        s0 = fma(lv[r], xs[r], s0);
        s1 = fma(lv[r + 1], xs[r + 1], s1);
        s2 = fma(lv[r + 2], xs[r + 2], s2);
        s3 = fma(lv[r + 3], xs[r + 3], s3);
      }
      lmb_y[c] -= (s0 + s1) + (s2 + s3);
    }
    if (kt > 0) tile_store(tv, tiv);  // (the tile solve of this step is over: the barrier above)
    __syncthreads();
  }
  for (int c = t; c < n; c += 1024) x[c] = lmb_y[c];
  for (int q = t; q <= KT; q += 1024) flags[q] = 0;
}

// The same over MANY CUs: one workgroup per tile column, column KT - 1 first (workgroup 0).  Workgroup j owns y_j: it subtracts
// L_kj^T x_k for k = KT - 1 ... j + 1 as the x_k become available (the 48 x 48 tile of the next k is requested BEFORE the wait, so a
// column's critical path is flag -> 48 doubles of x -> one tile's matrix-vector product -> the 16-wide block solves -> publish), solves
// its own tile and raises xready[j].  A workgroup waits only for workgroups of SMALLER index (dispatched before it; workgroup 0 waits for
// nobody), so the waits cannot deadlock.  xready[] is zeroed by the first launch of the NEXT factorisation (k_lmb_step, k = -1): no
// workgroup of this kernel may reset a flag another one is still polling.  The one-workgroup version moved the whole 13 MB of an
// 1800-row factor through one CU: 0.44 ms; this one 0.15.
__global__ __launch_bounds__(192) void k_lmb_back_mw(int n, int KT, int ld, const double* __restrict__ M, const double* __restrict__ Linv,
                                                     LmState* __restrict__ st, int32_t* __restrict__ flags, int32_t* __restrict__ xready,
                                                     double* __restrict__ x) {
#pragma clang fp contract(fast)
  __shared__ double Lt[LMB_T * LMB_LS];
  __shared__ double Iv[3 * 16 * LMB_IS];
  __shared__ double yj[LMB_T], xk[LMB_T], part[4][LMB_T];
  if (!lmb_gate(st)) return;
  const int t = threadIdx.x;
  const int j = KT - 1 - (int)blockIdx.x;
  if (*(volatile int32_t*)&flags[KT] != 0) {  // a bad pivot: x = 0 (every workgroup its own rows; nobody waits)
    for (int c = t; c < LMB_T; c += 192)
      if (j * LMB_T + c < n) x[j * LMB_T + c] = 0.0;
    return;  // (the flag stays up until the next factorisation's first launch clears it: every workgroup of THIS launch must see it)
  }
  const int c = t % LMB_T, pr = t / LMB_T;  // column of the tile, row part (12 rows each)
  if (t < LMB_T) yj[t] = M[(size_t)ld * ld + (size_t)j * LMB_T + t];
  {  // this column's diagonal tile and block inverses (needed last: they travel under everything else)
    const double* Ljj = M + (size_t)j * LMB_T * ld + (size_t)j * LMB_T;
    for (int e = t; e < LMB_T * LMB_T; e += 192) Lt[(e / LMB_T) * LMB_LS + e % LMB_T] = Ljj[(size_t)(e / LMB_T) * ld + e % LMB_T];
    for (int e = t; e < 3 * 256; e += 192) Iv[(e >> 8) * 16 * LMB_IS + ((e & 255) >> 4) * LMB_IS + (e & 15)] = Linv[(size_t)j * 768 + e];
  }
  auto tile_rows = [&](int kt, double (&lv)[12]) {
    const double* T = M + ((size_t)kt * LMB_T + 12 * pr) * ld + (size_t)j * LMB_T + c;
#pragma unroll
    for (int r = 0; r < 12; ++r) lv[r] = T[(size_t)r * ld];
  };
  double lv[12];
  if (KT - 1 > j) tile_rows(KT - 1, lv);
  __syncthreads();
  for (int kt = KT - 1; kt > j; --kt) {
    if (t == 0) {
      while (__hip_atomic_load(&xready[kt], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (t < LMB_T) xk[t] = __builtin_nontemporal_load(&x[(size_t)kt * LMB_T + t]);
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int r = 0; r < 12; r += 2) {
      s0 = fma(lv[r], xk[12 * pr + r], s0);
      s1 = fma(lv[r + 1], xk[12 * pr + r + 1], s1);
    }
    part[pr][c] = s0 + s1;
    if (kt - 1 > j) tile_rows(kt - 1, lv);  // the next tile travels while this one is folded in
    __syncthreads();
    if (t < LMB_T) yj[t] -= (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    __syncthreads();
  }
  if (t < 64) {  // x_j = L_jj^-T y_j by the block inverses (as k_lmb_back)
    const int i = t & 15;
    auto mv_t = [&](const double* Mx, int pitch, const double* v) {
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int k2 = 0; k2 < 16; k2 += 2) {
        a0 = fma(Mx[k2 * pitch + i], v[k2], a0);
        a1 = fma(Mx[(k2 + 1) * pitch + i], v[k2 + 1], a1);
      }
      return a0 + a1;
    };
    const double x3 = mv_t(Iv + 2 * 16 * LMB_IS, LMB_IS, yj + 32);
    if (t < 16) xk[32 + i] = x3;
    lmb_wave_sync();
    const double y2 = yj[16 + i] - mv_t(Lt + 32 * LMB_LS + 16, LMB_LS, xk + 32);
    lmb_wave_sync();
    if (t < 16) yj[16 + i] = y2;
    lmb_wave_sync();
    const double x2 = mv_t(Iv + 16 * LMB_IS, LMB_IS, yj + 16);
    if (t < 16) xk[16 + i] = x2;
    lmb_wave_sync();
    const double y1 = yj[i] - (mv_t(Lt + 32 * LMB_LS, LMB_LS, xk + 32) + mv_t(Lt + 16 * LMB_LS, LMB_LS, xk + 16));
    lmb_wave_sync();
    if (t < 16) yj[i] = y1;
    lmb_wave_sync();
    const double x1 = mv_t(Iv, LMB_IS, yj);
    if (t < 16) {
      double* xo = x + (size_t)j * LMB_T;  // (x has ld entries: the padding rows come out 0)
      xo[i] = x1, xo[16 + i] = x2, xo[32 + i] = x3;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    lmb_wave_sync();
    if (t == 0) __hip_atomic_store(&xready[j], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

size_t lm_big_inv_bytes(int nf) { return (((size_t)6 * nf + LMB_T - 1) / LMB_T) * 768 * sizeof(double); }
size_t lm_big_bytes(int nf) {
  const size_t ld = ((size_t)6 * nf + LMB_T - 1) / LMB_T * LMB_T;
  return (ld + LMB_T) * ld * sizeof(double);
}
int lm_big_ld(int nf) { return (int)(((size_t)6 * nf + LMB_T - 1) / LMB_T * LMB_T); }

void launch_lm_big_init(hipStream_t s, const LmLaunch& L) {  // M is zero-filled by the caller
  if (L.ld > 6 * L.nf) hipLaunchKernelGGL(k_lmb_init, dim3(1), dim3(64), 0, s, L.M, 6 * L.nf, L.ld);
}

void launch_lm_chol_big(hipStream_t s, const LmLaunch& L) {
  const int KT = L.ld / LMB_T;
  for (int k = -1; k <= KT - 2; ++k) {
    const int m = KT - 1 - k;
    const int grid = k < 0 ? m + 1 : m * (m + 3) / 2;
    hipLaunchKernelGGL(k_lmb_step, dim3(grid), dim3(192), 0, s, k, KT, L.ld, L.M, L.lmb_inv, L.state, L.lmb_flags);
  }
  // x gets ld entries (the padding rows come out 0); flags: [0, KT) unused, [KT] bad pivot, [KT + 1, 2 KT + 1) the back substitution's column flags
  hipLaunchKernelGGL(k_lmb_back_mw, dim3(KT), dim3(192), 0, s, 6 * L.nf, KT, L.ld, L.M, L.lmb_inv, L.state, L.lmb_flags, L.lmb_flags + KT + 1, L.x);
}

}  // namespace orbfe
