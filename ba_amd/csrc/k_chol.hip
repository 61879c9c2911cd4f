// FP64 solve of the reduced camera system S delta = rhs on gfx950.
//
// Replaces CalculateGn (BundleAdjuster.cpp:748-833): the reference converts the dense s_ to
// a sparse view and runs Eigen::SimplicialLDLT<Upper> (or dense LDLT<Upper>).  Here: blocked
// right-looking L D L^T, D = diag(+-1), on the LOWER storage (row-major, leading dimension ld, a
// multiple of 64), tile-sparse at 64x64 granularity, three blocking levels, with look-ahead:
//
//   for every outer panel J of KOUT tile columns (4; 8 for n >= 16k; 16 for n >= 25k)
//     for every sub-panel of 4 (8 for n >= 32k) tile columns, for every 64-column tile jj  [stream s0]
//       k_trsm_op      rows below the diagonal tile: X = A L_jj^-T D (blocked substitution on
//                      the matrix cores from the "factor packet" of tile jj)
//       k_step_update  update of the sub-panel's remaining tile columns with column jj (K = 64)
//                      [n >= 32k, left-looking: of column jj + 1 with all earlier columns of the
//                      sub-panel]; its workgroup (0,0) also factorises the next diagonal tile in
//                      LDS and publishes that tile's factor packet
//     k_step_update    after a sub-panel: the rest of the outer panel with its columns
//     k_step_update (+ k_update128<true> for the rows below the panel when there are >= 128)
//                      (a) the next panel's columns, K = 64 KOUT, + its first diagonal tile
//     k_update128<false> / k_update2   (b) everything right of the next panel, K = 64 KOUT
//                      concurrently with the next panel's serial chain               [stream s1]
//   k_linvT, k_backward2  L^T x = y, two tile rows per launch
//
// All products run on the FP64 matrix cores (v_mfma_f64_16x16x4_f64 — the one true dense
// contraction of the path).  L carries sqrt|pivot| and D the pivot signs: for SPD systems it
// IS the Cholesky factor, and like the reference's un-pivoted LDL^T it does not break down on
// the indefinite / nearly singular systems that ill-posed gauges produce
// (BundleAdjuster.cpp:752-761).  The right-hand side rides along as one extra row below the
// matrix, so the forward substitution L y = b is a by-product.  Masked parameters carry 1e6
// on the diagonal; a zero / NaN / Inf pivot raises the status flag (-> FactorizationError,
// BundleAdjuster.cpp:756-759).  With the collectives hook the factorisation is distributed
// over the ranks (cholesky_solve_dist).
#include "engine.h"
#include <cstdlib>
#include <algorithm>
#include <vector>

namespace bae {

static const int NB = 64;        // tile size
static const int LDT = NB + 2;   // LDS row stride (doubles): conflict-free MFMA operand reads

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

static const int LDM = 18;  // LDS row stride of the 16x16 diagonal inverses

// ---------------------------------------------------------------------------------
// Trailing update: tile (i, c) -= sum over the panel's tile columns kb of X_i,kb D_kb X_c,kb^T,
// one 64x64 output tile per workgroup (4 waves x 32x32 = 2x2 MFMA tiles each):
//   * K is consumed in chunks of 16 through double-buffered LDS stages (stride 18: the
//     16-lane phases of the operand reads hit distinct banks): while the matrix cores work
//     on chunk k, chunk k+1 moves registers -> LDS and chunks k+2, k+3 global -> registers
//     (two register sets); one barrier per chunk;
//   * the accumulators start at zero and the C tile is fetched after the loop;
//   * 37 KB of LDS per workgroup, four workgroups per CU.
static const int KC2 = 16;
#ifndef BAE_LDK2            // (measurement builds: LDS row stride of the operand stages)
#define BAE_LDK2 (KC2 + 2)
#endif
static const int LDK2 = BAE_LDK2;

// Fast path of update_tile — a full 64-row tile and no negative pivot in the K range (every tile
// of an SPD system except the rhs row).  Hand-scheduled:
//   * the active tile columns come from a wave-uniform bit mask (one pattern byte per lane,
//     __ballot) and are walked with scalar bit operations: no LDS list, no serial loads;
//   * one loop iteration = one tile column = four 16-wide chunks with static LDS stages and
//     static prefetch sets; the last tile column is a second copy of the body that fetches the C
//     tile instead of operands;
//   * operand fragments of k-step s + 1 are read from LDS while the four MFMAs of k-step s run
//     (two fragment sets); the single barrier of a chunk sits between k-steps 2 and 3, so the
//     first fragments of the next chunk are read under the last MFMAs of this one.
//     sched_barrier pins that order (the compiler otherwise issues each fragment read directly
//     in front of the MFMAs that need it).
__device__ __forceinline__ void update_tile_fast(double* __restrict__ A, uint32_t ld, uint32_t i, uint32_t c,
                                                 uint32_t kb0, uint64_t mask, double (*X)[NB][LDK2],
                                                 double (*Y)[NB][LDK2]) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int rb = 32 * (wave >> 1), cb = 32 * (wave & 1);
  double* Aic = A + ((size_t)i * NB) * ld + (size_t)c * NB;
  double4_t acc[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int sr = tid >> 3, sc = (tid & 7) * 2;
  const double* Xg0 = A + ((size_t)i * NB + sr) * ld + sc;
  const double* Xg1 = Xg0 + (size_t)32 * ld;
  const double* Yg0 = A + ((size_t)c * NB + sr) * ld + sc;
  const double* Yg1 = Yg0 + (size_t)32 * ld;
  double2 px0[2], px1[2], py0[2], py1[2];
  double F0[4], F1[4];
  double4_t cv[2][2];
#define BAE_SB __builtin_amdgcn_sched_barrier(0)
#define BAE_GLOAD(S, K0)                                        \
  {                                                             \
    px0[S] = *reinterpret_cast<const double2*>(Xg0 + (K0));     \
    px1[S] = *reinterpret_cast<const double2*>(Xg1 + (K0));     \
    py0[S] = *reinterpret_cast<const double2*>(Yg0 + (K0));     \
    py1[S] = *reinterpret_cast<const double2*>(Yg1 + (K0));     \
  }
#define BAE_SSTORE(B, S)                                                    \
  {                                                                         \
    X[B][sr][sc] = px0[S].x; X[B][sr][sc + 1] = px0[S].y;                   \
    X[B][sr + 32][sc] = px1[S].x; X[B][sr + 32][sc + 1] = px1[S].y;         \
    Y[B][sr][sc] = py0[S].x; Y[B][sr][sc + 1] = py0[S].y;                   \
    Y[B][sr + 32][sc] = py1[S].x; Y[B][sr + 32][sc + 1] = py1[S].y;         \
  }
#define BAE_LDF(F, B, KS)                                                              \
  {                                                                                    \
    F[0] = X[B][rb + li][4 * (KS) + lk]; F[1] = X[B][rb + 16 + li][4 * (KS) + lk];     \
    F[2] = Y[B][cb + li][4 * (KS) + lk]; F[3] = Y[B][cb + 16 + li][4 * (KS) + lk];     \
  }
#define BAE_MM(F)                                                                    \
  {                                                                                  \
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[0], F[2], acc[0][0], 0, 0, 0); \
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[0], F[3], acc[0][1], 0, 0, 0); \
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[1], F[2], acc[1][0], 0, 0, 0); \
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[1], F[3], acc[1][1], 0, 0, 0); \
  }
// chunk j of a tile column, operands in LDS stage (j & 1), ks = 0 fragments already in F0
#define BAE_CHUNK(J, STORE, LOADK, NEXT)                          \
  {                                                               \
    if (STORE) BAE_SSTORE(((J) + 1) & 1, ((J) + 1) & 1);          \
    LOADK;                                                        \
    BAE_SB; BAE_LDF(F1, (J) & 1, 1); BAE_SB; BAE_MM(F0); BAE_SB;  \
    BAE_LDF(F0, (J) & 1, 2); BAE_SB; BAE_MM(F1); BAE_SB;          \
    BAE_LDF(F1, (J) & 1, 3); BAE_SB; BAE_MM(F0); BAE_SB;          \
    __syncthreads(); BAE_SB;                                      \
    if (NEXT) BAE_LDF(F0, ((J) + 1) & 1, 0);                      \
    BAE_SB; BAE_MM(F1); BAE_SB;                                   \
  }
  // scalar walk over the set bits of the mask
  uint32_t kcur = (kb0 + (uint32_t)__builtin_ctzll(mask)) * NB;
  mask &= mask - 1;
  BAE_GLOAD(0, kcur);
  BAE_GLOAD(1, kcur + KC2);
  BAE_SSTORE(0, 0);
  BAE_GLOAD(0, kcur + 2 * KC2);
  __syncthreads();
  BAE_LDF(F0, 0, 0);
  while (mask) {
    const uint32_t knext = (kb0 + (uint32_t)__builtin_ctzll(mask)) * NB;
    mask &= mask - 1;
    // stage 0 holds chunk 0 of this column, set 1 chunk 1, set 0 (in flight) chunk 2
    BAE_CHUNK(0, true, BAE_GLOAD(1, kcur + 3 * KC2), true);
    BAE_CHUNK(1, true, BAE_GLOAD(0, knext), true);
    BAE_CHUNK(2, true, BAE_GLOAD(1, knext + KC2), true);
    BAE_CHUNK(3, true, BAE_GLOAD(0, knext + 2 * KC2), true);
    kcur = knext;
  }
  // the last tile column: the C tile is fetched under its MFMAs once a prefetch set is dead
  BAE_CHUNK(0, true, BAE_GLOAD(1, kcur + 3 * KC2), true);
  BAE_CHUNK(1, true, , true);
#define BAE_CLOAD                                                                              \
  _Pragma("unroll") for (int ti = 0; ti < 2; ++ti)                                             \
    _Pragma("unroll") for (int tj = 0; tj < 2; ++tj)                                           \
      _Pragma("unroll") for (int reg = 0; reg < 4; ++reg)                                      \
        cv[ti][tj][reg] = Aic[(size_t)(rb + 16 * ti + lk + 4 * reg) * ld + cb + 16 * tj + li];
  BAE_CHUNK(2, true, BAE_CLOAD, true);
#undef BAE_CLOAD
  BAE_CHUNK(3, false, , false);
#undef BAE_SB
#undef BAE_GLOAD
#undef BAE_SSTORE
#undef BAE_LDF
#undef BAE_MM
#undef BAE_CHUNK
  const bool diag = (i == c);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = rb + 16 * ti + lk + 4 * reg;
        const int cc = cb + 16 * tj + li;
        if (!diag || cc <= r) Aic[(size_t)r * ld + cc] = cv[ti][tj][reg] - acc[ti][tj][reg];
      }
}

// BULK = true (the look-ahead's background updates on small systems) pads the LDS footprint
// to 56 KB: at most two such workgroups fit on a CU, which always leaves the 45 KB + one wave
// per SIMD that a k_step_update / k_trsm_op workgroup of the concurrent serial chain needs.
// One 64x64 output tile (i, c): A_ic -= sum over the active tile columns kb of [kb0, kb1) of
// X_i,kb D_kb X_c,kb^T.  Called by all 256 threads of a workgroup; X, Y, klist are its LDS.
__device__ __forceinline__ void update_tile(double* __restrict__ A, uint32_t ld, uint32_t nblk, uint32_t i,
                                            uint32_t c, uint32_t kb0, uint32_t kb1,
                                            const double* __restrict__ dsgn, const int* __restrict__ colneg,
                                            const uint8_t* __restrict__ nz, double (*X)[NB][LDK2],
                                            double (*Y)[NB][LDK2], uint32_t* klist) {
  const int rows = (i == nblk) ? 1 : NB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int rb = 32 * (wave >> 1), cb = 32 * (wave & 1);
  double* Aic = A + ((size_t)i * NB) * ld + (size_t)c * NB;
  // The accumulators start at zero and the C tile is fetched after the K loop: no MFMA of the
  // loop depends on a global load (a wait for the C tile, needed by the first MFMA, would be
  // re-executed in every iteration and serialise the operand prefetch), and the registers go
  // to a second prefetch set instead.
  double4_t acc[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = (double4_t){0.0, 0.0, 0.0, 0.0};
  // staging: thread t moves rows sr and sr + 32, columns sc, sc + 1 of a 64 x 16 chunk
  const int sr = tid >> 3, sc = (tid & 7) * 2;
  const double* Xg0 = A + ((size_t)i * NB + (sr < rows ? sr : 0)) * ld + sc;
  const double* Xg1 = A + ((size_t)i * NB + (sr + 32 < rows ? sr + 32 : 0)) * ld + sc;
  const double* Yg0 = A + ((size_t)c * NB + sr) * ld + sc;
  const double* Yg1 = Yg0 + (size_t)32 * ld;
  const double xm0 = sr < rows ? 1.0 : 0.0, xm1 = sr + 32 < rows ? 1.0 : 0.0;
  // tile-sparse factor: column kb contributes only where both operand tiles are structurally
  // nonzero (nz = tile pattern of L, nblk x nblk bytes; the rhs row is dense)
#ifndef BAE_NO_FAST
  if (rows == NB && kb1 - kb0 <= 64u) {
    // wave-uniform pattern mask of the K range: lane l looks at tile column kb0 + l
    const uint32_t kl = kb0 + (uint32_t)lane;
    bool on = kl < kb1;
    bool neg = false;
    if (on) {
      neg = colneg[kl] != 0;
      if (nz) on = nz[(size_t)i * nblk + kl] && nz[(size_t)c * nblk + kl];
    }
    const uint64_t mask = __ballot(on);
    if (__ballot(neg) == 0) {
      if (mask == 0) return;
      update_tile_fast(A, ld, i, c, kb0, mask, X, Y);
      return;
    }
  }
#endif
  uint32_t nact = 0;
  for (uint32_t kb = kb0; kb < kb1; ++kb) {
    const bool on = !nz || ((i == nblk || nz[(size_t)i * nblk + kb]) && nz[(size_t)c * nblk + kb]);
    if (on) {
      if (threadIdx.x == 0) klist[nact] = kb * NB;
      ++nact;
    }
  }
  if (nact == 0) return;  // nothing to subtract from this tile (uniform over the workgroup)
  __syncthreads();
  const int nchunk = (int)nact * (NB / KC2);  // a multiple of 4
  // two register sets: a chunk is loaded two iterations (~2 x 4096 MFMA-pipe cycles at four
  // waves per SIMD) before it is written to LDS
  double2 px0[2], px1[2], py0[2], py1[2], ps[2] = {make_double2(1.0, 1.0), make_double2(1.0, 1.0)};
  // `plain`: a full tile and no negative pivot in the K range (every SPD system) — the staged
  // operands are then plain copies, no per-element multiplies in the loop.  The accumulators
  // collect +X D Y^T; the epilogue subtracts.
  bool plain = rows == NB;
  for (uint32_t kb = kb0; kb < kb1; ++kb) plain = plain && (colneg[kb] == 0);
#define BAE_GLOAD(S, CH)                                                          \
  {                                                                               \
    const uint32_t k0_ = klist[(CH) >> 2] + (uint32_t)((CH) & 3) * KC2;           \
    px0[S] = *reinterpret_cast<const double2*>(Xg0 + k0_);                        \
    px1[S] = *reinterpret_cast<const double2*>(Xg1 + k0_);                        \
    py0[S] = *reinterpret_cast<const double2*>(Yg0 + k0_);                        \
    py1[S] = *reinterpret_cast<const double2*>(Yg1 + k0_);                        \
    if (!plain) ps[S] = *reinterpret_cast<const double2*>(dsgn + k0_ + sc);       \
  }
#define BAE_SSTORE(B, S)                                                          \
  if (plain) {                                                                    \
    X[B][sr][sc] = px0[S].x; X[B][sr][sc + 1] = px0[S].y;                         \
    X[B][sr + 32][sc] = px1[S].x; X[B][sr + 32][sc + 1] = px1[S].y;               \
    Y[B][sr][sc] = py0[S].x; Y[B][sr][sc + 1] = py0[S].y;                         \
    Y[B][sr + 32][sc] = py1[S].x; Y[B][sr + 32][sc + 1] = py1[S].y;               \
  } else {                                                                        \
    X[B][sr][sc] = px0[S].x * xm0; X[B][sr][sc + 1] = px0[S].y * xm0;             \
    X[B][sr + 32][sc] = px1[S].x * xm1; X[B][sr + 32][sc + 1] = px1[S].y * xm1;   \
    Y[B][sr][sc] = ps[S].x * py0[S].x; Y[B][sr][sc + 1] = ps[S].y * py0[S].y;     \
    Y[B][sr + 32][sc] = ps[S].x * py1[S].x; Y[B][sr + 32][sc + 1] = ps[S].y * py1[S].y; \
  }
#define BAE_MMA(B)                                                                \
  _Pragma("unroll") for (int ks = 0; ks < KC2 / 4; ++ks) {                        \
    const double a0 = X[B][rb + li][4 * ks + lk];                                 \
    const double a1 = X[B][rb + 16 + li][4 * ks + lk];                            \
    const double b0 = Y[B][cb + li][4 * ks + lk];                                 \
    const double b1 = Y[B][cb + 16 + li][4 * ks + lk];                            \
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0); \
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0); \
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0); \
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0); \
  }
  BAE_GLOAD(0, 0);
  BAE_GLOAD(1, 1);
  BAE_SSTORE(0, 0);
  BAE_GLOAD(0, 2);
  __syncthreads();
  for (int kc = 0; kc + 2 < nchunk; kc += 2) {
    // LDS stage 0 holds chunk kc, set 1 chunk kc + 1, set 0 (in flight) chunk kc + 2
    BAE_SSTORE(1, 1);
    if (kc + 3 < nchunk) BAE_GLOAD(1, kc + 3);
    BAE_MMA(0);
    __syncthreads();
    BAE_SSTORE(0, 0);
    if (kc + 4 < nchunk) BAE_GLOAD(0, kc + 4);
    BAE_MMA(1);
    __syncthreads();
  }
  // the last two chunks, peeled: both prefetch sets are dead after the next store, so the C
  // tile is fetched here — its latency hides under the last 32 MFMAs of the wave
  BAE_SSTORE(1, 1);
  double4_t cv[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = rb + 16 * ti + lk + 4 * reg;
        cv[ti][tj][reg] = Aic[(size_t)(r < rows ? r : 0) * ld + cb + 16 * tj + li];
      }
  BAE_MMA(0);
  __syncthreads();
  BAE_MMA(1);
#undef BAE_GLOAD
#undef BAE_SSTORE
#undef BAE_MMA
  // epilogue: C - X D Y^T  (rows past `rows` are not touched)
  const bool diag = (i == c);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = rb + 16 * ti + lk + 4 * reg;
        const int cc = cb + 16 * tj + li;
        if (r < rows && (!diag || cc <= r)) Aic[(size_t)r * ld + cc] = cv[ti][tj][reg] - acc[ti][tj][reg];
      }
}

template <bool BULK>
__global__ void __launch_bounds__(256, BULK ? 2 : 4)
k_update2(double* __restrict__ A, uint32_t ld, uint32_t nblk, uint32_t c0, uint32_t kb0,
          uint32_t kb1, const double* __restrict__ dsgn, const int* __restrict__ colneg, int swz,
          const uint8_t* __restrict__ nz, const OwnMap own) {
  __shared__ double X[2][NB][LDK2];
  __shared__ double Y[2][NB][LDK2];
  __shared__ uint32_t klist[32];  // the tile columns of [kb0, kb1) with a structurally nonzero product
  __shared__ double pad_[BULK ? 2432 : 1];
  if (BULK && kb0 == 0xffffffffu) pad_[threadIdx.x] = 0.0;  // keeps the padding allocated
  if (!BULK) __builtin_amdgcn_s_setprio(2);  // critical-path launches outrank the bulk waves
  uint32_t c, i;
  if (swz & 1) {
    // XCD-aware 1-D launch (the big trailing updates): workgroups whose ids agree mod 8 share
    // an XCD and its L2, so each XCD walks its own 8x8 super-blocks of output tiles — the
    // 8 + 8 operand slices of a super-block (2 MB at K = 256) stay L2-resident instead of
    // being re-fetched for every tile.  Super-blocks enumerate the lower triangle row by row.
    const uint32_t sbl = (uint32_t)swz >> 8;  // log2 of the super-block edge (3: 8x8 tiles)
    const uint32_t b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
    const uint32_t t = (slot >> (2 * sbl)) * 8u + xcd, within = slot & ((1u << (2 * sbl)) - 1u);
    uint32_t sr = (uint32_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((uint64_t)(sr + 1) * (sr + 2) / 2 <= t) ++sr;
    while ((uint64_t)sr * (sr + 1) / 2 > t) --sr;
    const uint32_t sc = t - (uint32_t)((uint64_t)sr * (sr + 1) / 2);
    const uint32_t R = (sr << sbl) + (within >> sbl), C = (sc << sbl) + (within & ((1u << sbl) - 1u));
    if (C > R) return;
    c = c0 + C;
    i = c0 + R;
    if (i > nblk || c >= nblk) return;
  } else {
    c = c0 + blockIdx.y;
    i = c + blockIdx.x;
    if (i > nblk) return;
  }
  // distributed solve: a rank only updates the tiles it owns (dist_plan.h)
  if (!own_tile(own, i, c, nblk)) return;
  update_tile(A, ld, nblk, i, c, kb0, kb1, dsgn, colneg, nz, X, Y, klist);
}

// ---------------------------------------------------------------------------------
// 128x128 trailing update for the big look-ahead launches: one workgroup forms the 2x2 block of
// 64-tiles (i, i+1) x (c, c+1), four waves x 64x64 (4x4 MFMA tiles each).  Per MFMA it moves half
// the LDS fragments, half the LDS stores and half the L2 -> register operand bytes of the 64x64
// kernel.  That is what counts on this chip: with real data the 64x64 kernel runs into the
// package power limit (1.36 kW, sclk 2.40 -> 2.23 GHz measured; all-zero operands run 13 % faster),
// so the energy spent on moving operands is paid for in MFMA clock.
//   * same summation order per element as the 64x64 kernel (results are bitwise identical): tile
//     columns of [kb0, kb1) where either row tile and either column tile is structurally nonzero
//     are formed for the whole block — a structurally zero operand tile holds exact zeros;
//   * one-deep register prefetch (the 128 accumulator registers leave room for one set), chunks of
//     16, two LDS stages of 2 x 128 x 16, fragments of k-step s + 1 read under the 16 MFMAs of
//     k-step s, one barrier per chunk (as update_tile_fast);
//   * the C block is read and written in the epilogue, one row of MFMA tiles at a time;
//   * a negative pivot in the K range (indefinite system) sends the block through update_tile,
//     one 64-tile at a time.
// The blocks cover the even part of the trailing matrix; the rhs row and an odd last tile row are
// 64-tiles taken by the first workgroups of the same launch.
static const int NB2 = 128;
#ifdef BAE_TIME128  // measurement build: every 128-block of the bulk launches leaves a record (scratch/gpu_r03_time128.sh)
__device__ unsigned long long* g_t128_buf = nullptr;
__device__ unsigned int g_t128_n = 0;
static const unsigned int T128_CAP = 1u << 21, T128_REC = 6;
#endif
template <bool RECT, bool LIST = false>  // RECT: the rectangle variant (separate kernel name in profiles); LIST: own.pairs
__global__ void __launch_bounds__(256, 2)
k_update128(double* __restrict__ A, uint32_t ld, uint32_t nblk, uint32_t c0, uint32_t m2, uint32_t kb0,
            uint32_t kb1, const double* __restrict__ dsgn, const int* __restrict__ colneg, uint32_t sbl,
            const uint8_t* __restrict__ nz, const OwnMap own,
            uint32_t nrow64, uint32_t r0, uint32_t rect_cols) {
  __shared__ double X[2][NB2][LDK2];
  __shared__ double Y[2][NB2][LDK2];
  __shared__ uint32_t klist[32];
  // The first nrow64 workgroups take the 64-tiles the 128-blocks leave over: the rhs row (row
  // nblk) and, when the trailing matrix has an odd number of tile rows, its last tile row.
  // Two shapes: the lower triangle of the trailing matrix from tile c0 on (rect_cols = 0, r0 = c0),
  // or the rectangle rows >= r0 x the rect_cols 128-column blocks from c0 on (the next panel's
  // columns below the panel itself).
  if (blockIdx.x < nrow64) {
    const uint32_t mcols = RECT ? 2 * rect_cols : nblk - c0, y = blockIdx.x / mcols;
    const uint32_t c = c0 + blockIdx.x % mcols, i = nblk - y;
    if (y > ((nblk - r0) & 1u) || c > i || c >= nblk) return;  // (nrow64 is rounded up to a multiple of 8)
    if (!own_tile(own, i, c, nblk)) return;
    update_tile(A, ld, nblk, i, c, kb0, kb1, dsgn, colneg, nz, reinterpret_cast<double(*)[NB][LDK2]>(&X[0][0][0]),
                reinterpret_cast<double(*)[NB][LDK2]>(&Y[0][0][0]), klist);
    return;
  }
  const uint32_t b = blockIdx.x - nrow64;
  uint32_t R, C;
  if constexpr (LIST) {
    // distributed bulk update: workgroup -> (owned block pair, 128-block inside it); consecutive workgroups (= the
    // XCDs) take the column blocks of one row of the pair, as in the rectangle mode
    const uint32_t gb = own.G >> 1, per = gb * gb;
    const uint32_t pr = b / per, within = b - pr * per;
    const uint2* pairs = reinterpret_cast<const uint2*>(((unsigned long long)own.pairs_hi << 32) | own.pairs_lo);
    const uint2 bp = pairs[pr];
    const uint32_t base = c0 / own.G;
    R = (bp.x - base) * gb + within / gb;
    C = (bp.y - base) * gb + within % gb;
    if (C > R || R >= m2) return;
  } else if (RECT) {
    // consecutive workgroups (= the XCDs, round-robin) take the column blocks of one row block:
    // an XCD keeps "its" column operands in L2 for all rows
    R = b / rect_cols;
    C = b % rect_cols;
    if (R >= m2) return;
  } else {
    // XCD-aware mapping as in k_update2, over the m2 x m2 grid of 128-blocks
    const uint32_t xcd = b & 7u, slot = b >> 3;
    const uint32_t t = (slot >> (2 * sbl)) * 8u + xcd, within = slot & ((1u << (2 * sbl)) - 1u);
    uint32_t sr_ = (uint32_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((uint64_t)(sr_ + 1) * (sr_ + 2) / 2 <= t) ++sr_;
    while ((uint64_t)sr_ * (sr_ + 1) / 2 > t) --sr_;
    const uint32_t sc_ = t - (uint32_t)((uint64_t)sr_ * (sr_ + 1) / 2);
    R = (sr_ << sbl) + (within >> sbl);
    C = (sc_ << sbl) + (within & ((1u << sbl) - 1u));
    if (C > R || R >= m2) return;
  }
  const uint32_t c = c0 + 2 * C, i = r0 + 2 * R;
  if (!own_tile(own, i, c, nblk)) return;  // (a 128-block never straddles an ownership block: G is even)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // pattern mask of the K range (lane l: tile column kb0 + l) and pivot-sign check
  const uint32_t kl = kb0 + (uint32_t)lane;
  bool on = kl < kb1, neg = false;
  if (on) {
    neg = colneg[kl] != 0;
    if (nz)
      on = (nz[(size_t)i * nblk + kl] | nz[(size_t)(i + 1) * nblk + kl]) &&
           (nz[(size_t)c * nblk + kl] | nz[(size_t)(c + 1) * nblk + kl]);
  }
  uint64_t mask = __ballot(on);
  if (mask == 0) return;
  if (__ballot(neg) != 0 || kb1 - kb0 > 64u) {
    double(*X64)[NB][LDK2] = reinterpret_cast<double(*)[NB][LDK2]>(&X[0][0][0]);
    double(*Y64)[NB][LDK2] = reinterpret_cast<double(*)[NB][LDK2]>(&Y[0][0][0]);
    for (uint32_t q = 0; q < 4; ++q) {
      const uint32_t ii = i + (q >> 1), cc = c + (q & 1);
      if (cc > ii) continue;
      __syncthreads();
      update_tile(A, ld, nblk, ii, cc, kb0, kb1, dsgn, colneg, nz, X64, Y64, klist);
    }
    return;
  }
  const bool diag = (i == c);
  const int li = lane & 15, lk = lane >> 4;
  const int rb = 64 * (wave >> 1), cb = 64 * (wave & 1);
  double4_t acc[4][4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int sr = tid >> 3, sc = (tid & 7) * 2;
  const double* Xg = A + ((size_t)i * NB + sr) * ld + sc;
  const double* Yg = A + ((size_t)c * NB + sr) * ld + sc;
  const size_t ld32 = (size_t)32 * ld;
  double2 px[4], py[4];
  double FA0[4], FB0[4], FA1[4], FB1[4];
#ifdef BAE_NO_SCHED_PINS_128   // (measurement builds: the compiler schedules the 128x128 loop freely)
#define BAE_SB
#else
#define BAE_SB __builtin_amdgcn_sched_barrier(0)
#endif
// (BAE_KMASK: measurement builds only — 63 makes every operand chunk come from tile column 0, i.e. from the L2 /
// Infinity Cache: wrong numbers, the kernel's rate without HBM misses; scratch/gpu_r03_ceiling.sh)
#ifndef BAE_KMASK
#define BAE_KMASK 0xffffffffu
#endif
#define BAE_GLOAD(K0)                                                                \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                    \
    px[u] = *reinterpret_cast<const double2*>(Xg + u * ld32 + ((K0) & BAE_KMASK));   \
    py[u] = *reinterpret_cast<const double2*>(Yg + u * ld32 + ((K0) & BAE_KMASK));   \
  }
#define BAE_SSTORE(B)                                                                \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                    \
    X[B][sr + 32 * u][sc] = px[u].x; X[B][sr + 32 * u][sc + 1] = px[u].y;            \
    Y[B][sr + 32 * u][sc] = py[u].x; Y[B][sr + 32 * u][sc + 1] = py[u].y;            \
  }
#define BAE_LDF(FA, FB, B, KS)                                                       \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                    \
    FA[q] = X[B][rb + 16 * q + li][4 * (KS) + lk];                                   \
    FB[q] = Y[B][cb + 16 * q + li][4 * (KS) + lk];                                   \
  }
#define BAE_MM(FA, FB)                                                               \
  _Pragma("unroll") for (int ti = 0; ti < 4; ++ti)                                   \
    _Pragma("unroll") for (int tj = 0; tj < 4; ++tj)                                 \
      acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[ti], FB[tj], acc[ti][tj], 0, 0, 0);
#define BAE_CHUNK(J, STORE, LOADK, NEXT)                                             \
  {                                                                                  \
    if (STORE) BAE_SSTORE(((J) + 1) & 1);                                            \
    LOADK;                                                                           \
    BAE_SB; BAE_LDF(FA1, FB1, (J) & 1, 1); BAE_SB; BAE_MM(FA0, FB0); BAE_SB;         \
    BAE_LDF(FA0, FB0, (J) & 1, 2); BAE_SB; BAE_MM(FA1, FB1); BAE_SB;                 \
    BAE_LDF(FA1, FB1, (J) & 1, 3); BAE_SB; BAE_MM(FA0, FB0); BAE_SB;                 \
    __syncthreads(); BAE_SB;                                                         \
    if (NEXT) BAE_LDF(FA0, FB0, ((J) + 1) & 1, 0);                                   \
    BAE_SB; BAE_MM(FA1, FB1); BAE_SB;                                                \
  }
#ifdef BAE_TIME128   // measurement build: cycle stamps of one workgroup in 257 (scratch/gpu_r03_time128.sh)
  const unsigned long long tt0 = __builtin_readcyclecounter();
  const unsigned long long tw0 = wall_clock64();
  const int tt_cols = __builtin_popcountll(mask);
#endif
  uint32_t kcur = (kb0 + (uint32_t)__builtin_ctzll(mask)) * NB;
  mask &= mask - 1;
  BAE_GLOAD(kcur);
  BAE_SSTORE(0);
  BAE_GLOAD(kcur + KC2);
  __syncthreads();
  BAE_LDF(FA0, FB0, 0, 0);
#ifdef BAE_TIME128
  const unsigned long long tt1 = __builtin_readcyclecounter();
  const unsigned long long tw1 = wall_clock64();  // constant 100 MHz: the tick rate of the cycle counter follows
#endif
  while (mask) {
    const uint32_t knext = (kb0 + (uint32_t)__builtin_ctzll(mask)) * NB;
    mask &= mask - 1;
    // stage 0 holds chunk 0 of this tile column, the register set (in flight) chunk 1
    BAE_CHUNK(0, true, BAE_GLOAD(kcur + 2 * KC2), true);
    BAE_CHUNK(1, true, BAE_GLOAD(kcur + 3 * KC2), true);
    BAE_CHUNK(2, true, BAE_GLOAD(knext), true);
    BAE_CHUNK(3, true, BAE_GLOAD(knext + KC2), true);
    kcur = knext;
  }
  // the last tile column; the C block is read and written one row of four MFMA tiles (16 doubles
  // per lane) at a time, two rows in flight: row 0 is fetched under the last chunk (the prefetch
  // set is dead by then), row t + 2 when row t has been stored
  double* Aic = A + ((size_t)i * NB) * ld + (size_t)c * NB;
  double4_t cva[4], cvb[4];
#define BAE_CLOAD(CV, TI)                                                            \
  _Pragma("unroll") for (int tj = 0; tj < 4; ++tj)                                   \
    _Pragma("unroll") for (int reg = 0; reg < 4; ++reg)                              \
      CV[tj][reg] = Aic[(size_t)(rb + 16 * (TI) + lk + 4 * reg) * ld + cb + 16 * tj + li];
#define BAE_CSTORE(CV, TI)                                                           \
  _Pragma("unroll") for (int tj = 0; tj < 4; ++tj)                                   \
    _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) {                            \
      const int r = rb + 16 * (TI) + lk + 4 * reg;                                   \
      const int cc = cb + 16 * tj + li;                                              \
      if (!diag || cc <= r) Aic[(size_t)r * ld + cc] = CV[tj][reg] - acc[TI][tj][reg]; \
    }
  BAE_CHUNK(0, true, BAE_GLOAD(kcur + 2 * KC2), true);
  BAE_CHUNK(1, true, BAE_GLOAD(kcur + 3 * KC2), true);
  BAE_CHUNK(2, true, , true);
  BAE_CHUNK(3, false, BAE_CLOAD(cva, 0), false);
#ifdef BAE_TIME128
  const unsigned long long tt2 = __builtin_readcyclecounter();
  const unsigned long long tw2 = wall_clock64();
#endif
  if (diag && wave == 1) return;  // the strictly upper 64-tile of a diagonal block (computed, not stored)
  BAE_SB;
  BAE_CLOAD(cvb, 1);
  BAE_SB;
  BAE_CSTORE(cva, 0);
  BAE_SB;
  BAE_CLOAD(cva, 2);
  BAE_SB;
  BAE_CSTORE(cvb, 1);
  BAE_SB;
  BAE_CLOAD(cvb, 3);
  BAE_SB;
  BAE_CSTORE(cva, 2);
  BAE_CSTORE(cvb, 3);
#ifdef BAE_TIME128
  if (tid == 0 && (blockIdx.x % 257u) == 100u && !RECT) {
    const unsigned long long tt3 = __builtin_readcyclecounter();
    printf("T128 cols %d prologue %llu loop %llu epilogue %llu per_chunk %llu loop_100MHz %llu\n", tt_cols, tt1 - tt0,
           tt2 - tt1, tt3 - tt2, (tt2 - tt1) / (unsigned long long)(4 * tt_cols), tw2 - tw1);
  }
  if (tid == 0 && !RECT && g_t128_buf) {  // one record per 128-block: scratch/analyze_t128.py
    const unsigned int idx = atomicAdd(&g_t128_n, 1u);
    if (idx < T128_CAP) {
      unsigned long long* r = g_t128_buf + (size_t)idx * T128_REC;
      r[0] = tw0; r[1] = tw1; r[2] = tw2; r[3] = wall_clock64();
      r[4] = ((unsigned long long)__builtin_amdgcn_s_getreg(6164) << 32) | (unsigned int)__builtin_amdgcn_s_getreg(63492);
      r[5] = (unsigned long long)blockIdx.x | ((unsigned long long)tt_cols << 32) | ((unsigned long long)kb0 << 40);
    }
  }
#endif
#undef BAE_CLOAD
#undef BAE_CSTORE
#undef BAE_SB
#undef BAE_GLOAD
#undef BAE_SSTORE
#undef BAE_LDF
#undef BAE_MM
#undef BAE_CHUNK
}

// ---------------------------------------------------------------------------------
// Pipelined panel step: the pivot chain of tile d + 1 runs INSIDE the update launch of step d
// instead of after it:
//
//   k_step_update  = the k_update2 tile update, except that workgroup (0,0) — the diagonal
//                    tile (c0,c0), dispatched first — keeps its updated tile in LDS,
//                    factorises it (factor_tile_wave0), and publishes the
//                    "factor packet" of step c0: the 40 A-operand vectors of the blocked
//                    substitution (signed, in fragment order: plain coalesced loads for the
//                    consumer), the pivot signs and L^-T.  The chain (~12 us) overlaps the
//                    other tiles' updates of the same launch instead of following them.
//   k_trsm_op      = rows below the diagonal tile: X = A L^-T D from the factor packet —
//                    no LDS, no barrier: 16 + 40 loads, 40 MFMAs, 16 stores per wave.
//
// Serial path per 64 columns: k_trsm_op (~8 us) + the diagonal workgroup (~17 us).
static const int NOPV = 40;  // operand vectors per factor packet

struct TileLds {
  double T[NB][LDT];        // the diagonal tile: A_dd -> L_dd (upper part zero)
  double Md[4][16][LDM];    // M_cc = L_cc^-1, c = 0..3
  double dv[NB];            // 1 / L[j][j]
  double sg[NB];            // pivot signs d_j
  __attribute__((aligned(16))) double colbuf[2][NB];  // column j of L, for broadcast reads
  int bad;
  int neg;                  // any negative pivot in this tile
};

// Wave 0 factorises sh.T in place, then forms the four 16x16 diagonal inverses.
// 16-column panels: a row per lane (16 columns in registers); software-pipelined right-looking
// elimination — iteration j runs the pivot chain of column j (v_rsq_f64 + one third-order
// correction, 1.4e-16 measured; no division) while the rank-1 update of step j-1 on the columns
// right of j + 1, independent of that chain, is issued; only the update of column j + 1 by step
// j sits on the serial path.  A lone wave issues one VALU instruction every ~6 cycles, so the
// loop is kept lean: no pivot tests (a zero / NaN pivot poisons the diagonal, which is checked
// once at the end), the pivot sign is applied with an integer xor and collected in a scalar
// bit mask, the multipliers of the deferred updates come back from LDS as wave-uniform broadcast
// reads issued one serial section ahead of their use.  sched_barrier pins the interleaving (the
// compiler would otherwise re-serialise the updates into dot products in front of every
// pivot).  The rest of the tile is updated on the matrix cores straight out of LDS.
// Diagonal inverses: lane = (block, column), column-oriented forward substitution.
__device__ __forceinline__ void factor_tile_wave0(TileLds& sh, int lane, const double* __restrict__ floor_tile) {
  const int li = lane & 15, lk = lane >> 4;
  unsigned long long negmask = 0;
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    const int c0 = 16 * pb;
    double p[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) p[c] = (c0 + c <= lane) ? sh.T[lane][c0 + c] : 0.0;
    double sa_prev = 0.0;
    double lb[16];
    double d = readlane_f64(p[0], c0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int J = c0 + j;
      const int dhi = __double2hiint(d);
      const int sbit = dhi & (int)0x80000000;
      negmask |= (unsigned long long)((unsigned)dhi >> 31) << J;
      const double ad = fabs(d);
#define BAE_UPD(n)                                      \
  {                                                     \
    const int c_ = j + 1 + (n);                         \
    if (j > 0 && c_ < 16) p[c_] -= sa_prev * lb[c_];    \
  }
#define BAE_SB __builtin_amdgcn_sched_barrier(0)
      double y = __builtin_amdgcn_rsq(ad);
      BAE_SB;
      double t = ad * y;
      BAE_SB;
      double h = fma(-t, y, 1.0);
      BAE_UPD(0) BAE_SB;
      double q = fma(0.375, h, 0.5);
      double yh = y * h;
      BAE_UPD(1) BAE_UPD(2) BAE_UPD(3) BAE_SB;
      y = fma(yh, q, y);
      BAE_UPD(4) BAE_UPD(5) BAE_UPD(6) BAE_UPD(7) BAE_UPD(8) BAE_SB;
      const double sa = p[j] * y;
      BAE_UPD(9) BAE_UPD(10) BAE_SB;
      const double l = __hiloint2double(__double2hiint(sa) ^ sbit, __double2loint(sa));
      sh.colbuf[j & 1][lane] = l;
      p[j] = l;
      BAE_UPD(11) BAE_UPD(12) BAE_UPD(13) BAE_UPD(14) BAE_SB;
#pragma unroll
      for (int c = j + 2; c < 16; ++c) lb[c] = sh.colbuf[j & 1][c0 + c];
      BAE_SB;
      if (j < 15) {
        p[j + 1] -= sa * readlane_f64(l, J + 1);
        d = readlane_f64(p[j + 1], J + 1);
      }
      BAE_SB;
#undef BAE_UPD
#undef BAE_SB
      sa_prev = sa;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) sh.T[lane][c0 + c] = (c0 + c <= lane) ? p[c] : 0.0;
    sh.sg[lane] = ((negmask >> lane) & 1ull) ? -1.0 : 1.0;
    if (pb < 3) {
      double la[4][3], nb[4][3];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int k = c0 + 4 * ks + lk;
        const double sk = -sh.sg[k];
#pragma unroll
        for (int b = pb + 1; b < 4; ++b) {
          la[ks][b - 1] = sh.T[16 * b + li][k];
          nb[ks][b - 1] = sk * la[ks][b - 1];
        }
      }
#pragma unroll
      for (int rbk = pb + 1; rbk < 4; ++rbk)
#pragma unroll
        for (int cbk = pb + 1; cbk <= rbk; ++cbk) {
          double4_t acc;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) acc[reg] = sh.T[16 * rbk + lk + 4 * reg][16 * cbk + li];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(la[ks][rbk - 1], nb[ks][cbk - 1], acc, 0, 0, 0);
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) sh.T[16 * rbk + lk + 4 * reg][16 * cbk + li] = acc[reg];
        }
    }
  }
  if (lane == 0) sh.neg = negmask != 0 ? 1 : 0;
  {
    const double q = sh.T[lane][lane];
    if (!(q > 0.0 && q < 1e150)) sh.bad = 1;  // zero, NaN or Inf pivot
    // optional rank-deficiency guard (ba_hip_options::pivot_rel_tolerance): |d_j| = q^2 against
    // tol * |S_jj| of the matrix before elimination
    if (floor_tile && q * q < floor_tile[lane]) sh.bad = 1;
    sh.dv[lane] = 1.0 / q;
  }
  {
    const int base = 16 * lk, c = li;
    double sacc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const double xk = sacc[k] * sh.dv[base + k];
      sacc[k] = xk;
#pragma unroll
      for (int r = k + 1; r < 16; ++r) sacc[r] -= sh.T[base + r][base + k] * xk;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sh.Md[lk][r][c] = sacc[r];
  }
}

// Blocked substitution X = A L^-T D on the TRANSPOSE, never leaving registers:
//     R_c^T = A_c^T - sum_{k<c} (L_ck D_k) Z_k ,   Z_c = X_c^T = (D_c M_cc) R_c^T
// — the C/D fragment of v_mfma_f64_16x16x4_f64 (row = (lane>>4) + 4 reg, col = lane&15) is
// exactly its B fragment for k-step `reg`, so Z_k and R_c^T feed the next product directly;
// only L and M_cc (A operands) come from memory.  Operand vector v for this lane:
//   v < 16:  aM[c][ks] =  d_(16c+li) M_cc[li][4ks+lk]           c = v / 4, ks = v % 4
//   v >= 16: aL[c][k][ks] = -d_kk L[16c+li][kk], kk = 16k+4ks+lk, (c,k) = (1,0) (2,0) (2,1) (3,0) (3,1) (3,2)
// `sgn` false: all signs +1 (the inverse L^-T)
__device__ __forceinline__ double subst_operand(const TileLds& sh, int v, int li, int lk, bool sgn) {
  if (v < 16) {
    const int c = v >> 2, ks = v & 3;
    return (sgn ? sh.sg[16 * c + li] : 1.0) * sh.Md[c][li][4 * ks + lk];
  }
  const int pi = (v - 16) >> 2, ks = v & 3;
  const int c = pi == 0 ? 1 : (pi < 3 ? 2 : 3);
  const int k = pi == 0 ? 0 : (pi < 3 ? pi - 1 : pi - 3);
  const int kk = 16 * k + 4 * ks + lk;
  return -(sgn ? sh.sg[kk] : 1.0) * sh.T[16 * c + li][kk];
}

// R[c] (c = 0..3) holds A^T fragments of 16 rows; on return X^T.  op[v]: operand vectors.
__device__ __forceinline__ void subst_rows(double4_t R[4], const double op[NOPV]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double4_t Zc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) Zc = __builtin_amdgcn_mfma_f64_16x16x4f64(op[4 * c + ks], R[c][ks], Zc, 0, 0, 0);
#pragma unroll
    for (int c2 = c + 1; c2 < 4; ++c2) {
      const int pi = (c2 == 1 ? 0 : (c2 == 2 ? 1 : 3)) + c;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        R[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[16 + 4 * pi + ks], Zc[ks], R[c2], 0, 0, 0);
    }
    R[c] = Zc;
  }
}

// FACTOR = false: every tile takes the plain update (the diagonal tiles are factorised by k_square)
template <bool FACTOR>
__global__ void __launch_bounds__(256, 2)
k_step_update(double* __restrict__ A, uint32_t ld, uint32_t nblk, uint32_t c0, uint32_t kb0,
              uint32_t kb1, double* __restrict__ dsgn, double* __restrict__ opbuf,
              int* __restrict__ colneg, int* __restrict__ status, const uint8_t* __restrict__ nz,
              uint32_t row_end, const uint32_t* __restrict__ rowlist) {
  struct UpdLds { double X[2][NB][LDK2]; double Y[2][NB][LDK2]; };
  __shared__ uint32_t klist[32];  // active tile columns (tile-sparse factor, see k_update2)
  constexpr size_t kLds = sizeof(TileLds) > sizeof(UpdLds) ? sizeof(TileLds) : sizeof(UpdLds);
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLds];
  UpdLds& u = *reinterpret_cast<UpdLds*>(smem);
  TileLds& sh = *reinterpret_cast<TileLds*>(smem);
  __builtin_amdgcn_s_setprio(2);
  const uint32_t c = c0 + blockIdx.y;
  // rowlist (distributed solve): the row tiles this rank works on, ascending, instead of all rows from c on
  const uint32_t i = rowlist ? rowlist[blockIdx.x] : c + blockIdx.x;
  if (i > nblk || i >= row_end || i < c) return;  // row_end: rows from there on belong to a k_update128 launch
  const bool special = FACTOR && (i == c0 && c == c0);  // the diagonal tile (c0,c0): first workgroup of the launch
  if (!special) {  // every other tile: the plain trailing update
    update_tile(A, ld, nblk, i, c, kb0, kb1, dsgn, colneg, nz, u.X, u.Y, klist);
    return;
  }
  const int rows = (i == nblk) ? 1 : NB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const int rb = 32 * (wave >> 1), cb = 32 * (wave & 1);
  uint32_t nact = 0;
  for (uint32_t kb = kb0; kb < kb1; ++kb) {
    const bool on = !nz || ((i == nblk || nz[(size_t)i * nblk + kb]) && nz[(size_t)c * nblk + kb]);
    if (on) {
      if (tid == 0) klist[nact] = kb * NB;
      ++nact;
    }
  }
  __syncthreads();
  double* Aic = A + ((size_t)i * NB) * ld + (size_t)c * NB;
  double4_t acc[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = rb + 16 * ti + lk + 4 * reg;
        acc[ti][tj][reg] = Aic[(size_t)(r < rows ? r : 0) * ld + cb + 16 * tj + li];
      }
  const int nchunk = (int)nact * (NB / KC2);
  if (nchunk > 0) {
    const int sr = tid >> 3, sc = (tid & 7) * 2;
    const double* Xg0 = A + ((size_t)i * NB + (sr < rows ? sr : 0)) * ld + sc;
    const double* Xg1 = A + ((size_t)i * NB + (sr + 32 < rows ? sr + 32 : 0)) * ld + sc;
    const double* Yg0 = A + ((size_t)c * NB + sr) * ld + sc;
    const double* Yg1 = Yg0 + (size_t)32 * ld;
    const double xm0 = sr < rows ? 1.0 : 0.0, xm1 = sr + 32 < rows ? 1.0 : 0.0;
    double2 px0, px1, py0, py1, ps;
    auto gload = [&](uint32_t k0) {
      px0 = *reinterpret_cast<const double2*>(Xg0 + k0);
      px1 = *reinterpret_cast<const double2*>(Xg1 + k0);
      py0 = *reinterpret_cast<const double2*>(Yg0 + k0);
      py1 = *reinterpret_cast<const double2*>(Yg1 + k0);
      ps = *reinterpret_cast<const double2*>(dsgn + k0 + sc);
    };
    auto sstore = [&](int b) {
      u.X[b][sr][sc] = px0.x * xm0; u.X[b][sr][sc + 1] = px0.y * xm0;
      u.X[b][sr + 32][sc] = px1.x * xm1; u.X[b][sr + 32][sc + 1] = px1.y * xm1;
      u.Y[b][sr][sc] = -ps.x * py0.x; u.Y[b][sr][sc + 1] = -ps.y * py0.y;
      u.Y[b][sr + 32][sc] = -ps.x * py1.x; u.Y[b][sr + 32][sc + 1] = -ps.y * py1.y;
    };
    auto kof = [&](int ch) { return klist[ch >> 2] + (uint32_t)(ch & 3) * KC2; };
    gload(kof(0));
    sstore(0);
    if (nchunk > 1) gload(kof(1));
    __syncthreads();
    for (int kc = 0; kc < nchunk; ++kc) {
      const int b = kc & 1;
      if (kc + 1 < nchunk) sstore(b ^ 1);
      if (kc + 2 < nchunk) gload(kof(kc + 2));
#pragma unroll
      for (int ks = 0; ks < KC2 / 4; ++ks) {
        const double a0 = u.X[b][rb + li][4 * ks + lk];
        const double a1 = u.X[b][rb + 16 + li][4 * ks + lk];
        const double b0 = u.Y[b][cb + li][4 * ks + lk];
        const double b1 = u.Y[b][cb + 16 + li][4 * ks + lk];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  // ---- the diagonal tile: factorise, publish the factor packet of step c0 -------------
  __builtin_amdgcn_s_setprio(3);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = rb + 16 * ti + lk + 4 * reg;
        const int cc = cb + 16 * tj + li;
        sh.T[r][cc] = (cc <= r) ? acc[ti][tj][reg] : 0.0;
      }
  if (tid == 0) sh.bad = 0;
  __syncthreads();
  // the per-row pivot floors, if any, are published through the status block (ints 2..3 hold the
  // device pointer): no extra kernel argument on the many launch sites
  const double* floors = *reinterpret_cast<const double* const*>(status + 2);
  if (wave == 0) factor_tile_wave0(sh, lane, floors ? floors + (size_t)c0 * NB : nullptr);
  __syncthreads();
  {
    double* ob = opbuf + (size_t)c0 * NOPV * 64;
#pragma unroll
    for (int q = 0; q < NOPV / 4; ++q) {
      const int v = wave * (NOPV / 4) + q;
      ob[(size_t)v * 64 + lane] = subst_operand(sh, v, li, lk, true);
    }
    if (tid < NB) dsgn[(size_t)c0 * NB + tid] = sh.sg[tid];
    if (tid == 0) colneg[c0] = sh.neg;
    if (tid == 0 && sh.bad) atomicExch(status, 1);
  }
}

// rows below the diagonal tile d (row tiles d+1 .., the last block is the rhs row)
__global__ void __launch_bounds__(256)
k_trsm_op(double* __restrict__ A, uint32_t ld, uint32_t d, uint32_t nblk,
          const double* __restrict__ opbuf, const uint8_t* __restrict__ nz, const uint32_t* __restrict__ rowlist) {
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const uint32_t i = rowlist ? rowlist[blockIdx.x] : d + 1 + blockIdx.x;
  if (nz && i < nblk && !nz[(size_t)i * nblk + d]) return;  // structurally zero tile: stays zero
  const int rows = (i == nblk) ? 1 : NB;
  const int myrow = 16 * wave + li;
  double* Xrow = A + ((size_t)i * NB + (myrow < rows ? myrow : 0)) * ld + (size_t)d * NB;
  double4_t R[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) R[c][reg] = Xrow[16 * c + lk + 4 * reg];
  double op[NOPV];
  const double* ob = opbuf + (size_t)d * NOPV * 64 + lane;
#pragma unroll
  for (int v = 0; v < NOPV; ++v) op[v] = ob[(size_t)v * 64];
  subst_rows(R, op);
  if (myrow < rows) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Xrow[16 * c + lk + 4 * reg] = R[c][reg];
  }
}

// ---------------------------------------------------------------------------------
// Small systems (chain-bound: 94 tile columns at configs[1]): the diagonal SQUARE of a sub-panel — w <= 4
// tile columns, the lower triangle of w x w tiles — is factorised by ONE workgroup that never leaves the CU
// between its tile columns: no launch and no packet round trip between the factorisation of tile d, the
// substitution of the tiles below it and the update that completes tile d + 1.  Per tile column d = J + j:
//   factor   wave 0 factorises the diagonal tile in LDS (factor_tile_wave0); the other waves have their
//            strips of column d in flight meanwhile;
//   phase A  the tiles (r, d) below it inside the square: X = A L^-T D, operands straight from LDS, a 16-row
//            strip per task; X goes to global memory (the factor) and stays in LDS for phase B;
//   phase B  right-looking: every remaining tile (r, c), d < c <= r, takes the K = 64 update with column d,
//            operands from LDS, a 16-row strip per task, in the TRANSPOSED fragment layout of the
//            substitution (target strip = C/D fragments, X(r, d) strip = B fragments, X(c, d) = A operands);
//            the next diagonal tile (d+1, d+1) lands in LDS for its factorisation.
// A strip (tile, quarter) belongs to ONE wave for the whole kernel (sq_owner): its read-modify-write sequence
// through global memory is program order of that wave, so no barrier waits for a store to be acknowledged
// (lds_barrier: LDS traffic only) and loads are issued a phase ahead of their use.
// The factor packets, pivot signs and flags are published exactly as k_step_update does (k_rowpanel,
// k_linvT and the trailing updates read them).  The tiles of the square must carry every earlier
// panel's update when the kernel starts (k_step_update<false>).
struct SquareLds {
  TileLds t;
  double X[3][NB][LDT];  // X(d+1.., d) of the current tile column
};

// workgroup barrier that orders LDS traffic only: global stores stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// substitution with the operands computed from the factorised tile in LDS (subst_rows reads a packet)
__device__ __forceinline__ void subst_rows_lds(double4_t R[4], const TileLds& t, int li, int lk) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double4_t Zc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      Zc = __builtin_amdgcn_mfma_f64_16x16x4f64(subst_operand(t, 4 * c + ks, li, lk, true), R[c][ks], Zc, 0, 0, 0);
#pragma unroll
    for (int c2 = c + 1; c2 < 4; ++c2) {
      const int pi = (c2 == 1 ? 0 : (c2 == 2 ? 1 : 3)) + c;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        R[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(subst_operand(t, 16 + 4 * pi + ks, li, lk, true), Zc[ks], R[c2], 0, 0, 0);
    }
    R[c] = Zc;
  }
}

// Strip ownership: entry = r | c << 2 | q << 4 (tile (r, c) of the square, relative indices, quarter q), 255 = none.
// SQ_A[wave][j]: the strips of column j below the diagonal tile (phase A of step j) — waves 1..7 only: wave 0
// factorises meanwhile and holds nothing across the factorisation; SQ_B[wave][j]: the strips right of column j
// (phase B of step j; the diagonal tiles' strips included).  At most two / three strips per wave and phase.
__constant__ uint8_t SQ_A[8][4][2] = {{{255, 255}, {255, 255}, {255, 255}, {255, 255}}, {{1, 50}, {6, 55}, {255, 255}, {255, 255}}, {{17, 3}, {22, 255}, {11, 255}, {255, 255}}, {{33, 19}, {38, 255}, {27, 255}, {255, 255}}, {{49, 35}, {54, 255}, {43, 255}, {255, 255}}, {{2, 51}, {7, 255}, {59, 255}, {255, 255}}, {{18, 255}, {23, 255}, {255, 255}, {255, 255}}, {{34, 255}, {39, 255}, {255, 255}, {255, 255}}};
__constant__ uint8_t SQ_B[8][4][3] = {{{5, 10, 15}, {10, 15, 255}, {15, 255, 255}, {255, 255, 255}}, {{6, 55, 58}, {58, 255, 255}, {255, 255, 255}, {255, 255, 255}}, {{22, 11, 63}, {11, 63, 255}, {63, 255, 255}, {255, 255, 255}}, {{21, 38, 27}, {27, 255, 255}, {255, 255, 255}, {255, 255, 255}}, {{37, 54, 43}, {43, 255, 255}, {255, 255, 255}, {255, 255, 255}}, {{53, 7, 59}, {59, 255, 255}, {255, 255, 255}, {255, 255, 255}}, {{23, 26, 31}, {26, 31, 255}, {31, 255, 255}, {255, 255, 255}}, {{39, 42, 47}, {42, 47, 255}, {47, 255, 255}, {255, 255, 255}}};

__global__ void __launch_bounds__(512)
k_square(double* __restrict__ A, uint32_t ld, uint32_t nblk, uint32_t J, uint32_t w, double* __restrict__ dsgn,
         double* __restrict__ opbuf, int* __restrict__ colneg, int* __restrict__ status,
         const uint8_t* __restrict__ nz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sq_smem[];
  SquareLds& sh = *reinterpret_cast<SquareLds*>(sq_smem);
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* floors = *reinterpret_cast<const double* const*>(status + 2);
#ifdef BAE_TIME_SQ
  const unsigned long long tsq_start = __builtin_readcyclecounter();
#endif
  {
    const double* T0 = A + ((size_t)J * NB) * ld + (size_t)J * NB;
    for (int el = tid; el < NB * NB; el += 512) {
      const int r = el >> 6, c = el & 63;
      sh.t.T[r][c] = (c <= r) ? T0[(size_t)r * ld + c] : 0.0;
    }
    if (tid == 0) sh.t.bad = 0;
  }
#ifdef BAE_TIME_SQ  // measurement build: cycle stamps of the phases (scratch/gpu_r03_square_time.sh)
  __shared__ unsigned long long tsq[16];
#define BAE_TSQ(k) if (tid == 0) tsq[k] = __builtin_readcyclecounter()
#else
#define BAE_TSQ(k)
#endif
  lds_barrier();
#ifdef BAE_TIME_SQ
  if (tid == 0) { tsq[0] = tsq_start; tsq[1] = __builtin_readcyclecounter(); }
#endif
#define BAE_SLOAD(R, ptr)                          \
  _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) \
  _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) R[c_][g_] = (ptr)[16 * c_ + 4 * g_]
#define BAE_SSTORE_G(R, ptr)                       \
  _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) \
  _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) (ptr)[16 * c_ + 4 * g_] = R[c_][g_]
  for (uint32_t j = 0; j < w; ++j) {
    const uint32_t d = J + j, nb = w - 1 - j;  // nb: tiles below (d, d) inside the square
    // (the lane index is made opaque per step: hoisted out of the loop, the address and predicate
    // arithmetic of all four phases — 64 column predicates of the factorisation alone — would spill)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, lk = lane >> 4;
    // strip `e` (table entry) in the fragment layout of the substitution: this lane's base address
    auto strip_ptr = [&](uint32_t e) {
      const uint32_t r = e & 3u, c = (e >> 2) & 3u, q = e >> 4;
      return A + ((size_t)(J + r) * NB + 16 * q + li) * ld + (size_t)(J + c) * NB + lk;
    };
    // a table entry names a strip of THIS square (w < 4: the table of the 4-column square, filtered) whose
    // tile is structurally nonzero
    auto strip_on = [&](uint32_t e) {
      if (e == 255u || (e & 3u) >= w) return false;
      return !nz || nz[(size_t)(J + (e & 3u)) * nblk + J + ((e >> 2) & 3u)] != 0;
    };
    double4_t Ra[2][4];
    uint32_t ea[2];
    bool ona[2] = {false, false};
    if (wave == 0) {
      factor_tile_wave0(sh.t, lane, floors ? floors + (size_t)d * NB : nullptr);
    } else {
      // this wave's strips of column j: in flight while wave 0 factorises
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        ea[a] = SQ_A[wave][j][a];
        ona[a] = strip_on(ea[a]);
        if (ona[a]) {
          const double* ptr = strip_ptr(ea[a]);
          BAE_SLOAD(Ra[a], ptr);
        }
      }
    }
    lds_barrier();
    BAE_TSQ(2 + 3 * j);
    if (wave < 4) {  // the factor packet of tile column d
      double* ob = opbuf + (size_t)d * NOPV * 64;
#pragma unroll
      for (int q = 0; q < NOPV / 4; ++q) {
        const int v = wave * (NOPV / 4) + q;
        ob[(size_t)v * 64 + lane] = subst_operand(sh.t, v, li, lk, true);
      }
    } else if (wave == 4) {
      dsgn[(size_t)d * NB + lane] = sh.t.sg[lane];
      if (lane == 0) colneg[d] = sh.t.neg;
    }
    if (nb == 0) break;  // (uniform)
    // phase B's first strip: loaded now, the others one strip ahead of their update.  A strip whose tile is
    // structurally zero is skipped, except the next diagonal tile's (it has to reach LDS).
    // (w < 4: the entries of the 4-column table that lie outside the square are skipped)
    int bslot = 0;
    auto b_next = [&]() -> uint32_t {
      while (bslot < 3) {
        const uint32_t e = SQ_B[wave][j][bslot++];
        if (e != 255u && (e & 3u) < w) return e;
      }
      return 255u;
    };
    double4_t Rb[4];
    uint32_t eb = b_next();
    if (eb != 255u) {
      const double* ptr = strip_ptr(eb);
      BAE_SLOAD(Rb, ptr);
    }
    // ---- phase A: substitution of this wave's strips of column d ----
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if (ona[a]) {
        const uint32_t t = (ea[a] & 3u) - j - 1, q = ea[a] >> 4;
        subst_rows_lds(Ra[a], sh.t, li, lk);
        double* ptr = strip_ptr(ea[a]);
        BAE_SSTORE_G(Ra[a], ptr);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) sh.X[t][16 * q + li][16 * c + lk + 4 * reg] = Ra[a][c][reg];
      }
    }
    lds_barrier();
    BAE_TSQ(3 + 3 * j);
    // ---- phase B: column d applied to this wave's strips of the rest of the square ----
    while (eb != 255u) {
      const uint32_t en = b_next();
      double4_t Rn[4];
      if (en != 255u) {  // next strip: its loads are issued before this one's MFMAs
        const double* ptr = strip_ptr(en);
        BAE_SLOAD(Rn, ptr);
      }
      const uint32_t rr = eb & 3u, cr = (eb >> 2) & 3u, q = eb >> 4;
      const uint32_t tr = rr - j - 1, tc = cr - j - 1;
      const bool next_diag = rr == j + 1;  // (then cr == j + 1 too)
      const bool on = !nz || (nz[(size_t)(J + rr) * nblk + d] && nz[(size_t)(J + cr) * nblk + d]);
      if (on) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int kk = 16 * cc + 4 * reg + lk;
            const double b = sh.X[tr][16 * q + li][kk];
            const double nsg = -sh.t.sg[kk];
#pragma unroll
            for (int cp = 0; cp < 4; ++cp)
              Rb[cp] = __builtin_amdgcn_mfma_f64_16x16x4f64(nsg * sh.X[tc][16 * cp + li][kk], b, Rb[cp], 0, 0, 0);
          }
      }
      if (next_diag) {
        const int row = 16 * q + li;
#pragma unroll
        for (int cp = 0; cp < 4; ++cp)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int col = 16 * cp + lk + 4 * reg;
            sh.t.T[row][col] = (col <= row) ? Rb[cp][reg] : 0.0;
          }
      } else if (on) {
        double* ptr = strip_ptr(eb);
        BAE_SSTORE_G(Rb, ptr);
      }
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) Rb[cp] = Rn[cp];
      eb = en;
    }
    lds_barrier();
    BAE_TSQ(4 + 3 * j);
  }
  __syncthreads();
#ifdef BAE_TIME_SQ
  if (tid == 0 && J == 40 && w == 4) {
    const unsigned long long te = __builtin_readcyclecounter();
    printf("TSQ load %llu | f0 %llu a0 %llu b0 %llu | f1 %llu a1 %llu b1 %llu | f2 %llu a2 %llu b2 %llu | f3 %llu | total %llu\n",
           tsq[1] - tsq[0], tsq[2] - tsq[1], tsq[3] - tsq[2], tsq[4] - tsq[3], tsq[5] - tsq[4], tsq[6] - tsq[5],
           tsq[7] - tsq[6], tsq[8] - tsq[7], tsq[9] - tsq[8], tsq[10] - tsq[9], tsq[11] - tsq[10], te - tsq[0]);
  }
#endif
#undef BAE_TSQ
#undef BAE_SLOAD
#undef BAE_SSTORE_G
  if (tid == 0 && sh.t.bad) atomicExch(status, 1);
}

// The rows below a square (row tiles from row0 on; the last block is the rhs row): one workgroup per row
// tile, a 16-row strip per wave, LEFT-looking over the w tile columns of the square — the strip of column
// jj takes the updates with the columns before it (its own earlier strips from registers as B fragments,
// the square's tiles X(J+jj, J+k), signed and staged through LDS once per workgroup, as A operands) and then
// the substitution with packet jj (staged through LDS as well).  No workgroup depends on another one: every
// operand outside the strip was finished by k_square.
struct RowPanelLds {
  double S[2][NB][LDT];    // -D_k X(J+jj, J+k), double-buffered over the (jj, k) pairs
  double pk[NOPV * 64];    // factor packet jj
};

__global__ void __launch_bounds__(256)
k_rowpanel(double* __restrict__ A, uint32_t ld, uint32_t nblk, uint32_t J, uint32_t w, uint32_t row0,
           const double* __restrict__ opbuf, const double* __restrict__ dsgn, const uint8_t* __restrict__ nz,
           const uint32_t* __restrict__ rowlist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rp_smem[];
  RowPanelLds& sh = *reinterpret_cast<RowPanelLds*>(rp_smem);
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const uint32_t i = rowlist ? rowlist[blockIdx.x] : row0 + blockIdx.x;
  if (i > nblk) return;
  const int rows = (i == nblk) ? 1 : NB;
  const int myrow = 16 * wave + li;
  // staging of a square tile: thread t moves columns sc, sc + 1 of the rows sr + 8 it
  const int sr = tid >> 5, sc = (tid & 31) * 2;
  double2 P[8];
  double2 psg;
  auto tile_load = [&](uint32_t jj, uint32_t k) {
    const double* T = A + ((size_t)(J + jj) * NB + sr) * ld + (size_t)(J + k) * NB + sc;
#pragma unroll
    for (int it = 0; it < 8; ++it) P[it] = *reinterpret_cast<const double2*>(T + (size_t)(8 * it) * ld);
    psg = *reinterpret_cast<const double2*>(dsgn + (size_t)(J + k) * NB + sc);
  };
  auto tile_store = [&](int b) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      sh.S[b][sr + 8 * it][sc] = -psg.x * P[it].x;
      sh.S[b][sr + 8 * it][sc + 1] = -psg.y * P[it].y;
    }
  };
  double pkr[NOPV * 64 / 256];
  auto packet_load = [&](uint32_t jj) {
    const double* ob = opbuf + (size_t)(J + jj) * NOPV * 64 + tid;
#pragma unroll
    for (int it = 0; it < NOPV * 64 / 256; ++it) pkr[it] = ob[256 * it];
  };
  auto packet_store = [&]() {
#pragma unroll
    for (int it = 0; it < NOPV * 64 / 256; ++it) sh.pk[256 * it + tid] = pkr[it];
  };
  double4_t Rk[4][4];
  bool onk[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    onk[jj] = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) Rk[jj][c] = (double4_t){0.0, 0.0, 0.0, 0.0};
  }
  packet_load(0);
  if (w > 1) tile_load(1, 0);
  int np = 0;  // pairs staged so far
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    if ((uint32_t)jj < w) {
      const uint32_t d = J + jj;
      const bool on = !nz || i == nblk || nz[(size_t)i * nblk + d];
      onk[jj] = on;
      double* Xrow = A + ((size_t)i * NB + (myrow < rows ? myrow : 0)) * ld + (size_t)d * NB + lk;
      double4_t R[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) R[c][reg] = on ? Xrow[16 * c + 4 * reg] : 0.0;
#pragma unroll
      for (int k = 0; k < jj; ++k) {
        const int b = np & 1;
        ++np;
        tile_store(b);
        lds_barrier();
        // the next pair's tile: in flight during this pair's MFMAs
        if (k + 1 < jj) tile_load(jj, k + 1);
        else if ((uint32_t)jj + 1 < w) tile_load(jj + 1, 0);
        if (on && onk[k] && (!nz || nz[(size_t)d * nblk + J + k])) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const int kk = 16 * cc + 4 * reg + lk;
              const double bf = Rk[k][cc][reg];
#pragma unroll
              for (int cp = 0; cp < 4; ++cp)
                R[cp] = __builtin_amdgcn_mfma_f64_16x16x4f64(sh.S[b][16 * cp + li][kk], bf, R[cp], 0, 0, 0);
            }
        }
      }
      packet_store();
      lds_barrier();
      if ((uint32_t)jj + 1 < w) packet_load(jj + 1);
      if (on) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double4_t Zc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            Zc = __builtin_amdgcn_mfma_f64_16x16x4f64(sh.pk[(4 * c + ks) * 64 + lane], R[c][ks], Zc, 0, 0, 0);
#pragma unroll
          for (int c2 = c + 1; c2 < 4; ++c2) {
            const int pi = (c2 == 1 ? 0 : (c2 == 2 ? 1 : 3)) + c;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
              R[c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(sh.pk[(16 + 4 * pi + ks) * 64 + lane], Zc[ks], R[c2], 0, 0, 0);
          }
          R[c] = Zc;
        }
        if (myrow < rows) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Xrow[16 * c + 4 * reg] = R[c][reg];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) Rk[jj][c] = R[c];
      // (the packet buffer is rewritten after the next column's first pair barrier, or — w == 1 — never)
    }
  }
}

// L_dd^-T of every diagonal tile, for the backward substitution: the blocked substitution on
// the identity with the signs taken back out of the factor packets (d = +-1).  One launch
// for all tiles after the factorisation — off the serial chain.
__global__ void __launch_bounds__(256)
k_linvT(const double* __restrict__ opbuf, const double* __restrict__ dsgn, double* __restrict__ linvT) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  const uint32_t d = blockIdx.x;
  const double* sg = dsgn + (size_t)d * NB;
  const double* ob = opbuf + (size_t)d * NOPV * 64 + lane;
  double op[NOPV];
#pragma unroll
  for (int v = 0; v < NOPV; ++v) {
    double s;
    if (v < 16) {
      s = sg[16 * (v >> 2) + li];
    } else {
      const int pi = (v - 16) >> 2;
      const int k = pi == 0 ? 0 : (pi < 3 ? pi - 1 : pi - 3);
      s = sg[16 * k + 4 * (v & 3) + lk];
    }
    op[v] = ob[(size_t)v * 64] * s;
  }
  const int myrow = 16 * wave + li;
  double4_t R[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) R[c][reg] = (16 * c + lk + 4 * reg == myrow) ? 1.0 : 0.0;
  subst_rows(R, op);
  double* Xrow = linvT + (size_t)d * NB * NB + (size_t)myrow * NB;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Xrow[16 * c + lk + 4 * reg] = R[c][reg];
}

// Backward substitution, block row i (from the last to the first):
//   x_i = L_ii^-T y_i ;  y[0 : i*NB] -= L[i-block, 0:i*NB]^T x_i
// y lives in the rhs row of A.  linvT[i] holds L_ii^-T (row-major), so x_i is a 64x64
// mat-vec (4 partial sums per row, fixed order); then each thread updates one column.
// Workgroup 0 also stores x_i.
__global__ void __launch_bounds__(256)
k_backward(double* __restrict__ A, uint32_t ld, uint32_t i, uint32_t nblk,
           const double* __restrict__ linvT, double* __restrict__ x, const uint8_t* __restrict__ nz, uint32_t col0) {
  __shared__ double xi[NB];
  __shared__ double part[4][NB];
  const int tid = threadIdx.x;
  const double* y = A + ((size_t)nblk * NB) * ld + (size_t)i * NB;
  {
    const int r = tid & 63, q = tid >> 6;
    const double* Xr = linvT + (size_t)i * NB * NB + (size_t)r * NB;
    double s = 0.0;
#pragma unroll
    for (int c = q * 16; c < q * 16 + 16; ++c) s += Xr[c] * y[c];
    part[q][r] = s;
  }
  __syncthreads();
  if (tid < NB) {
    const double v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    xi[tid] = v;
    if (blockIdx.x == 0) x[(size_t)i * NB + tid] = v;
  }
  __syncthreads();
  const uint32_t col = col0 + blockIdx.x * 256 + tid;  // col0: first column updated (distributed solve: the panel's)
  if (col < i * NB && (!nz || nz[(size_t)i * nblk + (col >> 6)])) {  // structurally zero tiles of L: nothing to subtract
    const double* Li = A + ((size_t)i * NB) * ld + col;
    double s = 0.0;
#pragma unroll 16
    for (int r = 0; r < NB; ++r) s += Li[(size_t)r * ld] * xi[r];
    A[((size_t)nblk * NB) * ld + col] -= s;
  }
}

// Two block rows per launch (i + 1, then i): every workgroup solves the 2x2 block triangle itself
// (three 64x64 mat-vecs, redundantly — there is no dependency between workgroups inside a
// launch) and then updates its columns with both x vectors.  Halves the number of dependent
// launches of the backward substitution (each ~6 us of launch latency on small systems).
__device__ __forceinline__ void matvec64(const double* __restrict__ M, size_t stride_r, size_t stride_c,
                                         const double* v, double (*part)[NB], int tid) {
  // part[q][r] = sum over the q-th quarter of c of M[r][c] * v[c]
  const int r = tid & 63, q = tid >> 6;
  double s = 0.0;
#pragma unroll
  for (int c = q * 16; c < q * 16 + 16; ++c) s += M[(size_t)r * stride_r + (size_t)c * stride_c] * v[c];
  part[q][r] = s;
}
__global__ void __launch_bounds__(256)
k_backward2(double* __restrict__ A, uint32_t ld, uint32_t i, uint32_t nblk,
            const double* __restrict__ linvT, double* __restrict__ x, const uint8_t* __restrict__ nz, uint32_t col0) {
  __shared__ double x1[NB], x0[NB], y0[NB], y1[NB];
  __shared__ double part[4][NB];
  const int tid = threadIdx.x, r = tid & 63, q = tid >> 6;
  const double* yrow = A + ((size_t)nblk * NB) * ld;
  const uint32_t col = col0 + blockIdx.x * 256 + tid;
  const bool incol = col < i * NB;
  // A launch of this kernel is a chain of memory round trips (~1-2 us each) around a few hundred flops.  Everything
  // that depends on nothing computed here is requested at once: the operands of the first two mat-vecs, both y
  // blocks, the pattern flags and the rows of block row i + 1 in this thread's column; the operands of the third
  // mat-vec and the rows of block row i follow as soon as registers are free, each a full phase ahead of its use.
  // Same sums in the same order as the staged version (k_backward): results are bitwise unchanged.
  bool on1 = false, on0 = false;
  if (incol) {
    on1 = !nz || nz[(size_t)(i + 1) * nblk + (col >> 6)];
    on0 = !nz || nz[(size_t)i * nblk + (col >> 6)];
  }
  double m1[16], m2[16];
  {
    const double* M1 = linvT + (size_t)(i + 1) * NB * NB + (size_t)r * NB + 16 * q;           // L_(i+1)(i+1)^-T, row r
    const double* M2 = A + ((size_t)(i + 1) * NB + 16 * q) * ld + (size_t)i * NB + r;         // L_(i+1)i, column r
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      m1[k] = M1[k];
      m2[k] = M2[(size_t)k * ld];
    }
  }
  double yi1 = 0.0, yi0 = 0.0;
  if (tid < NB) {
    yi1 = yrow[(size_t)(i + 1) * NB + tid];
    yi0 = yrow[(size_t)i * NB + tid];
  }
  double lr[NB];
  if (on1) {
    const double* L1 = A + ((size_t)(i + 1) * NB) * ld + col;
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) lr[rr] = L1[(size_t)rr * ld];
  }
  if (tid < NB) y1[tid] = yi1;
  __syncthreads();
  // x1 = L_(i+1)(i+1)^-T y_(i+1)
  {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += m1[k] * y1[16 * q + k];
    part[q][r] = s;
  }
  __syncthreads();
  if (tid < NB) {
    const double v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    x1[tid] = v;
    if (blockIdx.x == 0) x[(size_t)(i + 1) * NB + tid] = v;
  }
  double m3[16];
  {
    const double* M3 = linvT + (size_t)i * NB * NB + (size_t)r * NB + 16 * q;                 // L_ii^-T, row r
#pragma unroll
    for (int k = 0; k < 16; ++k) m3[k] = M3[k];
  }
  __syncthreads();
  // y0 = y_i - L_(i+1)i^T x1
  {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += m2[k] * x1[16 * q + k];
    part[q][r] = s;
  }
  double s1 = 0.0, s0 = 0.0;
  if (on1) {
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) s1 += lr[rr] * x1[rr];
  }
  if (on0) {
    const double* L0 = A + ((size_t)i * NB) * ld + col;
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) lr[rr] = L0[(size_t)rr * ld];
  }
  __syncthreads();
  if (tid < NB) y0[tid] = yi0 - ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
  __syncthreads();
  // x0 = L_ii^-T y0
  {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += m3[k] * y0[16 * q + k];
    part[q][r] = s;
  }
  __syncthreads();
  if (tid < NB) {
    const double v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    x0[tid] = v;
    if (blockIdx.x == 0) x[(size_t)i * NB + tid] = v;
  }
  __syncthreads();
  if (on0) {
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) s0 += lr[rr] * x0[rr];
  }
  if (on1 || on0) A[((size_t)nblk * NB) * ld + col] -= (s1 + s0);
}

// the backward substitution over the block rows [lo, hi), last first, updating the columns from tile lo on:
// pairs of block rows, a single one first if the count is odd.  (0, nblk) is the whole substitution.
static void launch_backward(hipStream_t s, double* dA, uint32_t ld, uint32_t nblk, const double* linvT, double* dx,
                            const uint8_t* nz, uint32_t lo = 0, uint32_t hi = 0xffffffffu) {
  uint32_t ii = std::min(hi, nblk);
  const uint32_t col0 = lo * NB;
  if ((ii - lo) & 1u) {
    --ii;
    const uint32_t cols = (ii - lo) * NB, grid = cols == 0 ? 1 : (cols + 255) / 256;
    hipLaunchKernelGGL(k_backward, dim3(grid), dim3(256), 0, s, dA, ld, ii, nblk, linvT, dx, nz, col0);
  }
  while (ii >= lo + 2) {
    ii -= 2;
    const uint32_t cols = (ii - lo) * NB, grid = cols == 0 ? 1 : (cols + 255) / 256;
    hipLaunchKernelGGL(k_backward2, dim3(grid), dim3(256), 0, s, dA, ld, ii, nblk, linvT, dx, nz, col0);
  }
}

// Tile pattern of the factor L: the union of the shards' S patterns (all-reduce of the tile
// map), then symbolic elimination in natural order at 64x64-tile granularity:
// L(i,j) becomes nonzero when L(i,k) and L(j,k) are, k < j <= i.  The trailing updates,
// triangular solves and the look-ahead skip every tile product with a structurally zero
// operand — the reference reaches the same saving through Eigen::SimplicialLDLT on
// s_.sparseView() (BundleAdjuster.cpp:792-799).  BA_HIP_DENSE=1 keeps everything dense.
int factor_tile_pattern(Engine* e) {
  const Structure& st = e->st;
  const uint32_t nt = st.ld / 64;
  std::vector<uint8_t> nz(st.tile_nz);
  if (nz.size() != (size_t)nt * nt) nz.assign((size_t)nt * nt, 1);
  static const bool dense = getenv("BA_HIP_DENSE") != nullptr;
  if (dense) std::fill(nz.begin(), nz.end(), 1);
  if (e->sharded() && !dense) {
    std::vector<double> cnt(nz.begin(), nz.end());
    DBuf<double> d;
    BAE_HIP(d.alloc(cnt.size()));
    BAE_HIP(hipMemcpy(d.p, cnt.data(), cnt.size() * sizeof(double), hipMemcpyHostToDevice));
    if (shard_allreduce(e, d.p, cnt.size(), 0) != 0) { d.release(); return e->fail_msg("allreduce hook failed"); }
    BAE_HIP(hipMemcpy(cnt.data(), d.p, cnt.size() * sizeof(double), hipMemcpyDeviceToHost));
    d.release();
    for (size_t i = 0; i < nz.size(); ++i) nz[i] = cnt[i] > 0.5 ? 1 : 0;
  }
  e->nzS_host = nz;
  // symbolic right-looking elimination on the lower triangle; the working copy is column-major
  // (c[k * nt + i] = L(i,k)) so that both the scan of column k and the fill of column j are
  // contiguous
  std::vector<uint8_t> c(nz);  // symmetric on input
  std::vector<uint32_t> rows;
  for (uint32_t k = 0; k < nt; ++k) {
    rows.clear();
    const uint8_t* ck = &c[(size_t)k * nt];
    for (uint32_t i = k + 1; i < nt; ++i)
      if (ck[i]) rows.push_back(i);
    for (size_t a = 0; a < rows.size(); ++a) {
      uint8_t* cj = &c[(size_t)rows[a] * nt];
      for (size_t b = a; b < rows.size(); ++b) cj[rows[b]] = 1;
    }
  }
  for (uint32_t i = 0; i < nt; ++i)
    for (uint32_t k = 0; k < nt; ++k) nz[(size_t)i * nt + k] = (k <= i) ? c[(size_t)k * nt + i] : 0;
  BAE_HIP(e->nzL.alloc(nz.size()));
  BAE_HIP(hipMemcpy(e->nzL.p, nz.data(), nz.size(), hipMemcpyHostToDevice));
  e->nzL_host = nz;
  e->nzL_version++;
  e->nzL_valid = true;
  return 0;
}

uint32_t choose_kout(uint32_t nblk) {
  const char* ke = getenv("BA_HIP_KOUT");  // (read on every call: tests vary it inside one process)
  const uint32_t kout_env = ke ? (uint32_t)atoi(ke) : 0;
  return kout_env ? kout_env : (nblk >= 400 ? 16u : nblk >= 256 ? 8u : 4u);  // measured at 94 / 282 / 469 / 938 tiles
}

// ---------------------------------------------------------------------------------
// Distributed reduced solve (replaces CalculateGn,
// /root/reference/src/BundleAdjuster.cpp:748-833, across the GPUs of a node).  Ownership of the tile blocks
// and the message plan: dist_plan.h.  Per iteration:
//   * reduce-scatter: the partial S of every landmark shard is summed onto the owners of its tile blocks
//     (dist_reduce_scatter_S), instead of an all-reduce of the whole matrix;
//   * panel J, four streams per rank:
//       chain  (e->stream,  chain communicator)  the owner of block (J, J) factorises the SQUARE and broadcasts it
//              with its factor packets; the owner of block (J+1, J) substitutes those rows and sends them to the
//              ranks that multiply their class (URGENT: the next square needs them); the owner of block
//              (J+1, J+1) applies panel J to it and factorises its first diagonal tile in the same launch;
//       panel  (e->stream2) substitution + in-panel updates of the other row tiles this rank owns in panel J,
//              packing of their messages, then panel J applied to the tiles it owns in column block J+1;
//       side   (e->stream4, side communicator)   those messages, point to point, and their unpacking;
//       bulk   (e->stream3) panel J applied to the tiles it owns right of column block J+1 (k_update128).
//     Only the square and one block row per panel travel on the chain stream; everything else has until the
//     step its block row reaches the diagonal.
//   * forward substitution rides along (the rhs row is a row of every square); the backward substitution
//     walks the panels from the last: partial sums over the own tiles of the panel, one small all-reduce,
//     then the square's own rows (every rank holds every square) — the step is bitwise equal on all ranks.
// Enabled when the caller installed the collectives hook (ba_hip_set_collectives) or the native
// communicator, more than one rank takes part, and the reduced system need not stay readable
// (keep_reduced_system).  BA_HIP_DIST_LAYOUT = tri | grid | col | row overrides the layout.
bool dist_solve_enabled(const Engine* e) {
  static const bool off = getenv("BA_HIP_NO_DIST_SOLVE") != nullptr;
  return e->coll && e->sharded() && !e->opt.keep_reduced_system && !off;
}

// a list of 64-row tiles (then, if the grid says so, the rhs row = row n_pad) x columns [c0, c0 + w) of A
// <->  dense row-major block in buf: messages carry no structurally zero tiles
__global__ void __launch_bounds__(256)
k_copy_panel_rows(double* __restrict__ A, uint32_t ld, const uint32_t* __restrict__ tiles, uint32_t ntiles,
                  uint32_t n_pad, uint32_t c0, uint32_t w, double* __restrict__ buf, int unpack) {
  const uint32_t r = blockIdx.x;
  const uint32_t arow = (r < ntiles * NB) ? tiles[r >> 6] * NB + (r & 63u) : n_pad;
  double* row = A + (size_t)arow * ld + c0;
  double* brow = buf + (size_t)r * w;
  for (uint32_t cc = threadIdx.x; cc < w; cc += 256) {
    if (unpack) row[cc] = brow[cc];
    else brow[cc] = row[cc];
  }
}

// The serial chain of one outer panel [J, Jend) on stream s.
// Two-level inside the panel: sub-panels of KIN tile columns are factorised right-looking, one
// K = 64 update of the rest of the sub-panel per tile column; at the end of a sub-panel the
// remaining columns of the outer panel get ONE update with all KIN columns (K = 256) — a third of
// the tile updates of a flat right-looking panel at KOUT = 16, and mostly four times as deep.
// Every update launch also factorises the next diagonal tile (k_step_update).
// Distributed solve: `rl` lists the row tiles to work on (ascending, rl_n of them) instead of every row
// from the panel down; rl_square: the list starts with the panel's own tiles J .. Jend-1 (the square — its
// diagonal tiles are factorised here); otherwise all its rows lie below the panel and the factor packets
// of the diagonal tiles are already there (received with the square).
static const uint32_t KIN = 4;
static void launch_panel_chain(hipStream_t s, double* dA, uint32_t ld, uint32_t nblk, uint32_t J, uint32_t Jend,
                               double* dsgn, double* opbuf, int* colneg, int* flags, const uint8_t* nz,
                               const uint32_t* rl = nullptr, uint32_t rl_n = 0, bool rl_square = false,
                               bool sq = false) {
  // Large systems (the chain's tile updates are bandwidth-bound there and cost the bulk update their
  // full duration): sub-panels of 8, left-looking inside — every column is read and written once
  // per sub-panel instead of once per earlier column.  Small systems are latency-bound on the
  // diagonal tile: right-looking keeps its update shallow (K = 64).
  const bool left = nblk >= 512;
  const uint32_t KINv = left ? 2 * KIN : KIN;
  // rows from tile `from` on: the whole range, or the tail of the list
  auto rows_from = [&](uint32_t from, const uint32_t** lp, uint32_t* cnt) {
    if (!rl) { *lp = nullptr; *cnt = nblk - from + 1; return; }
    const uint32_t skip = rl_square ? std::min(from - J, rl_n) : 0u;
    *lp = rl + skip; *cnt = rl_n - skip;
  };
  if (rl && rl_n == 0) return;
  if (sq) {
    // small systems: the square of every sub-panel in one workgroup (k_square), the rows below it
    // left-looking (k_rowpanel), then the sub-panel applied to the rest of the outer panel
    const char* swe = getenv("BA_HIP_SQ_W");
    const uint32_t SW = swe ? std::min(4u, std::max(1u, (uint32_t)atoi(swe))) : KIN;
    for (uint32_t sub = J; sub < Jend; sub += SW) {
      const uint32_t sub_end = std::min(sub + SW, Jend);
      hipLaunchKernelGGL(k_square, dim3(1), dim3(512), sizeof(SquareLds), s, dA, ld, nblk, sub, sub_end - sub, dsgn,
                         opbuf, colneg, flags, nz);
      const uint32_t* lp; uint32_t cnt;
      rows_from(sub_end, &lp, &cnt);
      if (!cnt) continue;
      hipLaunchKernelGGL(k_rowpanel, dim3(cnt), dim3(256), sizeof(RowPanelLds), s, dA, ld, nblk, sub, sub_end - sub, sub_end,
                         (const double*)opbuf, (const double*)dsgn, nz, lp);
      if (sub_end < Jend)
        hipLaunchKernelGGL(k_step_update<false>, dim3(cnt, Jend - sub_end), dim3(256), 0, s, dA, ld, nblk, sub_end,
                           sub, sub_end, dsgn, opbuf, colneg, flags, nz, nblk + 1u, lp);
    }
    return;
  }
  for (uint32_t sub = J; sub < Jend; sub += KINv) {
    const uint32_t sub_end = std::min(sub + KINv, Jend);
    for (uint32_t jj = sub; jj < sub_end; ++jj) {
      const uint32_t* lp; uint32_t cnt;
      rows_from(jj + 1, &lp, &cnt);
      if (cnt) hipLaunchKernelGGL(k_trsm_op, dim3(cnt), dim3(256), 0, s, dA, ld, jj, nblk, (const double*)opbuf, nz, lp);
      if (jj + 1 < sub_end && cnt) {
        if (left)  // column jj + 1 alone, with every earlier column of the sub-panel at once
          hipLaunchKernelGGL(k_step_update<true>, dim3(cnt, 1), dim3(256), 0, s, dA, ld, nblk, jj + 1, sub,
                             jj + 1, dsgn, opbuf, colneg, flags, nz, nblk + 1u, lp);
        else       // the rest of the sub-panel with column jj
          hipLaunchKernelGGL(k_step_update<true>, dim3(cnt, sub_end - (jj + 1)), dim3(256), 0, s, dA, ld,
                             nblk, jj + 1, jj, jj + 1, dsgn, opbuf, colneg, flags, nz, nblk + 1u, lp);
      }
    }
    if (sub_end < Jend) {
      const uint32_t* lp; uint32_t cnt;
      rows_from(sub_end, &lp, &cnt);
      if (cnt) hipLaunchKernelGGL(k_step_update<true>, dim3(cnt, Jend - sub_end), dim3(256), 0, s, dA, ld, nblk,
                                  sub_end, sub, sub_end, dsgn, opbuf, colneg, flags, nz, nblk + 1u, lp);
    }
  }
}

// Update of the NEXT panel's tile columns [c0, c0 + ncols) with the tile columns [kb0, kb1) on the
// chain stream: the panel's own triangle — with the factorisation of tile (c0, c0) in workgroup
// (0,0) — stays a k_step_update launch; the rectangle below the panel goes to 128x128 blocks when
// it is big enough.
static void launch_next_panel_update(hipStream_t s, double* dA, uint32_t ld, uint32_t nblk, uint32_t c0,
                                     uint32_t ncols, uint32_t kb0, uint32_t kb1, double* dsgn, double* opbuf,
                                     int* colneg, int* flags, const uint8_t* nz, bool factor = true) {
  static const bool no128 = getenv("BA_HIP_NO128") != nullptr;
  const uint32_t r0 = c0 + ncols;
  auto step_update = factor ? k_step_update<true> : k_step_update<false>;
  // (a second launch on the serial chain: only where the chain is not the bottleneck)
  static const uint32_t min_rows =
      getenv("BA_HIP_BULK_FULL_M") ? (uint32_t)std::max(16, atoi(getenv("BA_HIP_BULK_FULL_M"))) : 128u;
  if (!no128 && r0 + min_rows <= nblk && ncols % 2u == 0 && c0 % 2u == 0) {
    const uint32_t m = nblk - r0, m2 = m / 2, rect_cols = ncols / 2;
    hipLaunchKernelGGL(step_update, dim3(ncols, ncols), dim3(256), 0, s, dA, ld, nblk, c0, kb0, kb1, dsgn, opbuf,
                       colneg, flags, nz, r0, (const uint32_t*)nullptr);
    const uint32_t nrow64 = (ncols * (1 + (m & 1u)) + 7) / 8 * 8;
    hipLaunchKernelGGL(k_update128<true>, dim3(nrow64 + m2 * rect_cols), dim3(256), 0, s, dA, ld, nblk, c0, m2, kb0, kb1,
                       (const double*)dsgn, (const int*)colneg, 0u, nz, own_map_single(), nrow64, r0, rect_cols);
    return;
  }
  hipLaunchKernelGGL(step_update, dim3(nblk - c0 + 1, ncols), dim3(256), 0, s, dA, ld, nblk, c0, kb0, kb1, dsgn,
                     opbuf, colneg, flags, nz, nblk + 1u, (const uint32_t*)nullptr);
}

// Bulk trailing update of the tile rows / columns >= a_end (and the rhs row) with the tile columns
// [J, Jend).  Big trailing matrices: 128x128 blocks over the even part (k_update128, XCD-aware 4x4
// super-blocks = the footprint of the 64-tile kernel's 8x8; the rhs row and an odd last tile row
// ride along as 64-tiles); otherwise the 64-tile kernel, capped (`full` = false) to leave the
// serial chain room.  `own`: the tiles this rank updates (distributed solve).
// One launch, bracketed by the profiling events.
static void launch_bulk_update(Engine* e, hipStream_t s, double* dA, uint32_t ld, uint32_t nblk, uint32_t a_end,
                                   uint32_t J, uint32_t Jend, const double* dsgn, const int* colneg,
                                   const uint8_t* nz, bool full, const OwnMap& own, const uint2* pairs = nullptr,
                                   uint32_t npairs = 0) {
  static const bool no128 = getenv("BA_HIP_NO128") != nullptr;  // A/B switch
  static const uint32_t sbl = getenv("BA_HIP_SBL") ? (uint32_t)atoi(getenv("BA_HIP_SBL")) : 3u;
  const uint32_t m = nblk - a_end;
  if (full && !no128 && m >= 16 && (a_end % 2u) == 0 && (own.n <= 1 || own.G % 2u == 0)) {
    const uint32_t m2 = m / 2, sbl2 = 2, sbe2 = 1u << sbl2;
    const uint32_t nsr = (m2 + sbe2 - 1) / sbe2, nsb = nsr * (nsr + 1) / 2;
    // leftover 64-tiles first (a multiple of 8 workgroups keeps the XCD phase of the blocks)
    const uint32_t nrow64 = (m * (1 + (m & 1u)) + 7) / 8 * 8;
    if (e) e->prof_begin(e->ev_syrk, s);
    if (pairs && own.n > 1 && a_end % own.G == 0) {
      // only the block pairs this rank owns (distributed solve)
      OwnMap ol = own;
      ol.pairs_lo = (uint32_t)(reinterpret_cast<unsigned long long>(pairs) & 0xffffffffull);
      ol.pairs_hi = (uint32_t)(reinterpret_cast<unsigned long long>(pairs) >> 32);
      const uint32_t per = (own.G / 2) * (own.G / 2);
      hipLaunchKernelGGL((k_update128<false, true>), dim3(nrow64 + npairs * per), dim3(256), 0, s, dA, ld, nblk,
                         a_end, m2, J, Jend, dsgn, colneg, sbl2, nz, ol, nrow64, a_end, 0u);
    } else {
      hipLaunchKernelGGL(k_update128<false>, dim3(nrow64 + ((nsb + 7) / 8) * 8 * sbe2 * sbe2), dim3(256), 0, s, dA, ld, nblk,
                         a_end, m2, J, Jend, dsgn, colneg, sbl2, nz, own, nrow64, a_end, 0u);
    }
    if (e) e->prof_end(e->ev_syrk, s);
    return;
  }
  // 1-D XCD-aware launch over the 8x8 super-blocks of the (m + 1) x m lower-triangular tile region
  const uint32_t sbe = 1u << sbl;
  const uint32_t nsr = (m + 1 + sbe - 1) / sbe, nsb = nsr * (nsr + 1) / 2;
  const uint32_t grid1 = ((nsb + 7) / 8) * 8 * sbe * sbe;
  const int swzf = 1 | (int)(sbl << 8);
  if (e) e->prof_begin(e->ev_syrk, s);
  if (full)
    hipLaunchKernelGGL(k_update2<false>, dim3(grid1), dim3(256), 0, s, dA, ld, nblk, a_end, J, Jend, dsgn, colneg,
                       swzf, nz, own);
  else
    hipLaunchKernelGGL(k_update2<true>, dim3(grid1), dim3(256), 0, s, dA, ld, nblk, a_end, J, Jend, dsgn, colneg,
                       swzf, nz, own);
  if (e) e->prof_end(e->ev_syrk, s);
}

// ---- the plan of the distributed solve for the current pattern / communicator ------------------------
static int ensure_dist_plan(Engine* e, uint32_t nblk, bool pat) {
  const uint32_t KOUT = choose_kout(nblk);
  const char* lay_env = getenv("BA_HIP_DIST_LAYOUT");
  const uint64_t want = (pat ? e->nzL_version : 0) * 1024 + KOUT * 4 + (pat ? 1 : 0);
  if (e->dist_plan_version == want && e->dist_plan.nblk == nblk && e->dist_plan.map.n == (uint32_t)e->nranks &&
      e->dist_plan.map.rank == (uint32_t)e->rank)
    return 0;
  OwnMap map;
  std::string why;
  if (!build_own_map((uint32_t)e->nranks, lay_env, KOUT, &map, &e->dist_layout_name, &why)) {
    e->err = "distributed solve: " + why;
    return -1;
  }
  map.rank = (uint32_t)e->rank;
  // one rank with the native communicator (comm_force): the rank sends its rows to itself — every pack /
  // transfer / unpack path runs on a one-GPU box
  e->dist_plan = build_dist_plan(nblk, map, pat ? e->nzL_host.data() : nullptr, e->nranks == 1);
  BAE_HIP(e->dist_tiles.alloc(std::max<size_t>(e->dist_plan.tiles.size(), 1)));
  if (!e->dist_plan.tiles.empty())
    BAE_HIP(hipMemcpy(e->dist_tiles.p, e->dist_plan.tiles.data(), e->dist_plan.tiles.size() * sizeof(uint32_t),
                      hipMemcpyHostToDevice));
  {
    // block pairs (bi >= bc) this rank owns, by block column then block row; pair_first[J] = first pair with
    // bc >= J + 2 (what the bulk update of panel J touches)
    const DistPlan& pl = e->dist_plan;
    std::vector<uint2> pairs;
    e->dist_pair_first.assign(pl.nb + 1, 0);
    for (uint32_t bc = 0; bc < pl.nb; ++bc) {
      if (bc >= 2) e->dist_pair_first[bc - 2] = (uint32_t)pairs.size();
      for (uint32_t bi = bc; bi < pl.nb; ++bi)
        if (map.n <= 1 || map.tbl[bi % map.T][bc % map.T] == map.rank) pairs.push_back(make_uint2(bi, bc));
    }
    for (uint32_t J = (pl.nb >= 2 ? pl.nb - 2 : 0); J <= pl.nb; ++J) e->dist_pair_first[J] = (uint32_t)pairs.size();
    e->dist_npairs = (uint32_t)pairs.size();
    BAE_HIP(e->dist_pairs.alloc(std::max<size_t>(pairs.size(), 1)));
    if (!pairs.empty())
      BAE_HIP(hipMemcpy(e->dist_pairs.p, pairs.data(), pairs.size() * sizeof(uint2), hipMemcpyHostToDevice));
  }
  {
    std::vector<uint32_t> h;
    for (const DistPanel& pn : e->dist_plan.panels) {
      for (uint32_t t = pn.c0; t < pn.c1; ++t) h.push_back(t);
      h.push_back(nblk);
    }
    BAE_HIP(e->dist_sq_list.alloc(std::max<size_t>(h.size(), 1)));
    BAE_HIP(hipMemcpy(e->dist_sq_list.p, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  e->dist_plan_version = want;
  e->dist_srows_version = ~0ull;
  e->dist_sp_version = ~0ull;
  return 0;
}

// Reduce-scatter of S onto the owners of its tile blocks.  Only the row tiles of a panel that hold a
// structurally nonzero tile of S on SOME shard travel (the union pattern of factor_tile_pattern, identical
// on every rank; S itself is half as dense as its factor): the other tiles are zero on every rank already.
// Chunk of destination rank r: for every panel, the row tiles of that panel whose block r owns.
int dist_scatter_S_sparse(Engine* e);
int dist_reduce_scatter_S(Engine* e) {
  const uint32_t ld = e->st.ld, nblk = ld / NB, N = (uint32_t)e->nranks, rank = (uint32_t)e->rank;
  // default: the sparse point-to-point exchange (below); BA_HIP_DENSE_SCATTER=1: one reduce-scatter of the union pattern
  if (!getenv("BA_HIP_DENSE_SCATTER") && e->nranks > 1) return dist_scatter_S_sparse(e);
  const bool pat = e->nzS_host.size() == (size_t)nblk * nblk;
  { const int prc = ensure_dist_plan(e, nblk, pat && e->nzL_host.size() == (size_t)nblk * nblk); if (prc) return prc; }
  const DistPlan& pl = e->dist_plan;
  const uint32_t npanels = pl.nb;
  const uint64_t want = e->dist_plan_version;
  if (e->dist_srows_version != want || e->dist_srows_off.size() != (size_t)npanels * N + 1) {
    std::vector<uint32_t> list;
    e->dist_srows_off.assign((size_t)npanels * N + 1, 0);
    for (uint32_t p = 0; p < npanels; ++p) {
      const uint32_t J = pl.panels[p].c0, Jend = pl.panels[p].c1;
      for (uint32_t r = 0; r < N; ++r) {
        OwnMap m = pl.map;
        m.rank = r;
        for (uint32_t i = J; i < nblk; ++i) {
          if (!own_tile(m, i, J, nblk)) continue;
          bool on = !pat || i < Jend;
          for (uint32_t kb = J; kb < Jend && !on; ++kb) on = e->nzS_host[(size_t)i * nblk + kb] != 0;
          if (on) list.push_back(i);
        }
        e->dist_srows_off[(size_t)p * N + r + 1] = (uint32_t)list.size();
      }
    }
    BAE_HIP(e->dist_srows.alloc(std::max<size_t>(list.size(), 1)));
    if (!list.empty())
      BAE_HIP(hipMemcpy(e->dist_srows.p, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    e->dist_srows_version = want;
  }
  // offsets inside the chunks: panels in order
  std::vector<size_t> off((size_t)npanels * N, 0), used(N, 0);
  for (uint32_t p = 0; p < npanels; ++p)
    for (uint32_t r = 0; r < N; ++r) {
      const size_t k = (size_t)p * N + r;
      const size_t rows = (size_t)(e->dist_srows_off[k + 1] - e->dist_srows_off[k]) * NB;
      off[k] = used[r];
      used[r] += rows * (size_t)(pl.panels[p].c1 - pl.panels[p].c0) * NB;
    }
  size_t chunk = 0;
  for (size_t u : used) chunk = std::max(chunk, u);
  chunk = (chunk + 63) / 64 * 64;
  BAE_HIP(e->packed.alloc(chunk * N));
  // the padding behind the shorter chunks is summed too: keep it finite
  BAE_HIP(hipMemsetAsync(e->packed.p, 0, chunk * N * sizeof(double), e->stream));
  for (uint32_t p = 0; p < npanels; ++p)
    for (uint32_t r = 0; r < N; ++r) {
      const size_t k = (size_t)p * N + r;
      const uint32_t ntl = e->dist_srows_off[k + 1] - e->dist_srows_off[k];
      if (ntl == 0) continue;
      hipLaunchKernelGGL(k_copy_panel_rows, dim3(ntl * NB), dim3(256), 0, e->stream, e->A.p, ld,
                         (const uint32_t*)(e->dist_srows.p + e->dist_srows_off[k]), ntl, nblk * NB, pl.panels[p].c0 * NB,
                         (pl.panels[p].c1 - pl.panels[p].c0) * NB, e->packed.p + (size_t)r * chunk + off[k], 0);
    }
  BAE_HIP(hipGetLastError());
  BAE_HIP(hipStreamSynchronize(e->stream));
  e->cstats.reduce_scatter_bytes += 8.0 * (double)chunk * N;
  if (e->coll(e->coll_ctx, 2, e->packed.p, chunk, 0) != 0) return e->fail_msg("reduce-scatter hook failed");
  for (uint32_t p = 0; p < npanels; ++p) {
    const size_t k = (size_t)p * N + rank;
    const uint32_t ntl = e->dist_srows_off[k + 1] - e->dist_srows_off[k];
    if (ntl == 0) continue;
    hipLaunchKernelGGL(k_copy_panel_rows, dim3(ntl * NB), dim3(256), 0, e->stream, e->A.p, ld,
                       (const uint32_t*)(e->dist_srows.p + e->dist_srows_off[k]), ntl, nblk * NB, pl.panels[p].c0 * NB,
                       (pl.panels[p].c1 - pl.panels[p].c0) * NB, e->packed.p + (size_t)rank * chunk + off[k], 1);
  }
  BAE_HIP(hipGetLastError());
  return 0;
}

// ---- sparse exchange of S: only the tiles a shard really has ------------------------------------------------------
// rects[b >> 6] = (row tile, first column in doubles, columns, offset of the 64 x columns block in buf);
// mode 0: buf <- A (pack), mode 1: A += buf (sum at the owner)
__global__ void __launch_bounds__(256)
k_copy_rects(double* __restrict__ A, uint32_t ld, const uint4* __restrict__ rects, double* __restrict__ buf, int mode) {
  const uint4 rc = rects[blockIdx.x >> 6];
  const uint32_t r = blockIdx.x & 63u;
  double* row = A + ((size_t)rc.x * NB + r) * ld + rc.y;
  double* brow = buf + (size_t)rc.w + (size_t)r * rc.z;
  for (uint32_t cc = threadIdx.x; cc < rc.z; cc += 256) {
    if (mode) row[cc] += brow[cc];
    else brow[cc] = row[cc];
  }
}

// Every rank holds a partial S over its landmark shard; the owner of a tile block needs the SUM.  The dense
// reduce-scatter (dist_reduce_scatter_S) moves every tile of the UNION pattern from every rank.  Here a rank sends a
// (row tile x panel) rectangle to its owner only if its OWN pattern (Structure::tile_nz: the tiles its shard, its
// pose-pose residuals and the diagonal touch) has a tile in it; the owner adds what arrives in ascending rank order
// (deterministic).  With shards dealt along the trajectory a shard touches a band of S and sends a fraction of what
// the dense exchange does; with shards that touch everything the volume is the dense one.  The local patterns are
// all-gathered once per structure (one u64 all-reduce with every rank's bits in its own slot).
int dist_scatter_S_sparse(Engine* e) {
  const uint32_t ld = e->st.ld, nblk = ld / NB, N = (uint32_t)e->nranks, rank = (uint32_t)e->rank;
  { const int prc = ensure_dist_plan(e, nblk, e->nzL_host.size() == (size_t)nblk * nblk); if (prc) return prc; }
  const DistPlan& pl = e->dist_plan;
  if (e->st.tile_nz.size() != (size_t)nblk * nblk) return e->fail_msg("sparse exchange of S: no local tile pattern");
  if (e->dist_sp_version != e->dist_plan_version) {
    // --- all-gather of the local patterns (lower triangle, bit per tile)
    const size_t bits = (size_t)nblk * nblk, words = (bits + 63) / 64;
    std::vector<unsigned long long> all(words * N, 0ull);
    for (uint32_t i = 0; i < nblk; ++i)
      for (uint32_t k = 0; k <= i; ++k)
        if (e->st.tile_nz[(size_t)i * nblk + k]) {
          const size_t b = (size_t)i * nblk + k;
          all[(size_t)rank * words + (b >> 6)] |= 1ull << (b & 63);
        }
    {
      DBuf<unsigned long long> d;
      BAE_HIP(d.alloc(all.size()));
      hipError_t err = hipMemcpy(d.p, all.data(), all.size() * 8, hipMemcpyHostToDevice);
      int rc = 0;
      if (err != hipSuccess) rc = e->fail(err, "hipMemcpy");
      if (!rc && shard_allreduce(e, d.p, all.size(), 1) != 0) rc = e->fail_msg("allreduce hook failed");
      if (!rc && (err = hipMemcpy(all.data(), d.p, all.size() * 8, hipMemcpyDeviceToHost)) != hipSuccess) rc = e->fail(err, "hipMemcpy");
      d.release();
      if (rc) return rc;
    }
    auto has = [&](uint32_t s_, uint32_t i, uint32_t c0, uint32_t c1) {
      for (uint32_t kb = c0; kb < c1 && kb <= i; ++kb) {
        const size_t b = (size_t)i * nblk + kb;
        if (all[(size_t)s_ * words + (b >> 6)] >> (b & 63) & 1ull) return true;
      }
      return false;
    };
    // --- rectangles this rank sends (by destination) and receives (by source), enumerated identically on both sides
    std::vector<uint4> srect, rrect;
    e->dist_sp_send.assign(N + 1, 0); e->dist_sp_recv.assign(N + 1, 0);
    e->dist_sp_send_off.assign(N + 1, 0); e->dist_sp_recv_off.assign(N + 1, 0);
    size_t soff = 0, roff = 0;
    bool too_big = false;
    for (uint32_t peer = 0; peer < N; ++peer) {
      e->dist_sp_send[peer] = (uint32_t)srect.size(); e->dist_sp_recv[peer] = (uint32_t)rrect.size();
      e->dist_sp_send_off[peer] = soff; e->dist_sp_recv_off[peer] = roff;
      if (peer == rank) continue;
      OwnMap mp = pl.map, mr = pl.map;
      mp.rank = peer; mr.rank = rank;
      for (uint32_t p = 0; p < pl.nb; ++p) {
        const uint32_t c0 = pl.panels[p].c0, c1 = pl.panels[p].c1, w = (c1 - c0) * NB;
        for (uint32_t i = c0; i < nblk; ++i) {
          if (own_tile(mp, i, c0, nblk) && has(rank, i, c0, c1)) {   // mine -> peer
            too_big |= soff > 0xffffffffull - (size_t)NB * w;
            srect.push_back(make_uint4(i, c0 * NB, w, (uint32_t)soff));
            soff += (size_t)NB * w;
          }
          if (own_tile(mr, i, c0, nblk) && has(peer, i, c0, c1)) {   // peer -> me
            too_big |= roff > 0xffffffffull - (size_t)NB * w;
            rrect.push_back(make_uint4(i, c0 * NB, w, (uint32_t)roff));
            roff += (size_t)NB * w;
          }
        }
      }
    }
    e->dist_sp_send[N] = (uint32_t)srect.size(); e->dist_sp_recv[N] = (uint32_t)rrect.size();
    e->dist_sp_send_off[N] = soff; e->dist_sp_recv_off[N] = roff;
    if (too_big) return e->fail_msg("sparse exchange of S: more than 2^32 doubles to one side (use BA_HIP_DENSE_SCATTER=1)");
    BAE_HIP(e->dist_sp_srect.alloc(std::max<size_t>(srect.size(), 1)));
    BAE_HIP(e->dist_sp_rrect.alloc(std::max<size_t>(rrect.size(), 1)));
    if (!srect.empty()) BAE_HIP(hipMemcpy(e->dist_sp_srect.p, srect.data(), srect.size() * sizeof(uint4), hipMemcpyHostToDevice));
    if (!rrect.empty()) BAE_HIP(hipMemcpy(e->dist_sp_rrect.p, rrect.data(), rrect.size() * sizeof(uint4), hipMemcpyHostToDevice));
    e->dist_sp_version = e->dist_plan_version;
  }
  const size_t stot = e->dist_sp_send_off[N], rtot = e->dist_sp_recv_off[N];
  // staging: the send half and the receive half of `packed`
  BAE_HIP(e->packed.alloc(std::max<size_t>(stot + rtot, 1)));
  double* sbuf = e->packed.p;
  double* rbuf = e->packed.p + stot;
  std::vector<DistXfer> xf;
  for (uint32_t peer = 0; peer < N; ++peer) {
    if (peer == rank) continue;
    const uint32_t ns = e->dist_sp_send[peer + 1] - e->dist_sp_send[peer];
    const size_t slen = e->dist_sp_send_off[peer + 1] - e->dist_sp_send_off[peer];
    if (ns) {
      hipLaunchKernelGGL(k_copy_rects, dim3(ns * 64), dim3(256), 0, e->stream, e->A.p, ld,
                         (const uint4*)(e->dist_sp_srect.p + e->dist_sp_send[peer]), sbuf, 0);
      xf.push_back({sbuf + e->dist_sp_send_off[peer], slen, (int)peer, true});
    }
    const size_t rlen = e->dist_sp_recv_off[peer + 1] - e->dist_sp_recv_off[peer];
    if (rlen) xf.push_back({rbuf + e->dist_sp_recv_off[peer], rlen, (int)peer, false});
  }
  BAE_HIP(hipGetLastError());
  e->cstats.reduce_scatter_bytes += 8.0 * (double)stot;
  {
    // (counted as the S exchange, not as chain traffic: undo dist_exchange's own bookkeeping)
    const ba_hip_comm_stats keep = e->cstats;
    const int xrc = dist_exchange(e, xf, false, e->stream);
    const double rs = e->cstats.reduce_scatter_bytes;
    e->cstats = keep;
    e->cstats.reduce_scatter_bytes = rs;
    if (xrc) return xrc;
  }
  for (uint32_t peer = 0; peer < N; ++peer) {   // ascending rank order: the sums are reproducible
    if (peer == rank) continue;
    const uint32_t nr = e->dist_sp_recv[peer + 1] - e->dist_sp_recv[peer];
    if (nr)
      hipLaunchKernelGGL(k_copy_rects, dim3(nr * 64), dim3(256), 0, e->stream, e->A.p, ld,
                         (const uint4*)(e->dist_sp_rrect.p + e->dist_sp_recv[peer]), rbuf, 1);
  }
  BAE_HIP(hipGetLastError());
  return 0;
}

// Which tiles of the factor's pattern this rank's assembly has to write when S is exchanged sparsely: the tiles it OWNS
// (they receive the other shards' rectangles and the factorisation's updates: they must start from its own partial or
// from zero) and the tiles of every rectangle it SENDS (a rectangle travels whole: tiles of it that the shard does not
// touch must be zeros, not last iteration's leftovers).  Every other tile is neither read nor sent by this rank.  need:
// nt x nt bytes, lower triangle.  Returns 1 if the restriction applies (distributed solve with the sparse exchange).
int dist_assembly_tiles(Engine* e, std::vector<uint8_t>* need) {
  const uint32_t nt = e->st.ld / NB;
  if (!dist_solve_enabled(e) || e->nranks <= 1 || getenv("BA_HIP_DENSE_SCATTER")) return 0;
  if (e->nzL_host.size() != (size_t)nt * nt || e->st.tile_nz.size() != (size_t)nt * nt) return 0;
  if (ensure_dist_plan(e, nt, true)) return -1;
  const DistPlan& pl = e->dist_plan;
  OwnMap me = pl.map;
  me.rank = (uint32_t)e->rank;
  need->assign((size_t)nt * nt, 0);
  for (uint32_t p = 0; p < pl.nb; ++p) {
    const uint32_t c0 = pl.panels[p].c0, c1 = pl.panels[p].c1;
    for (uint32_t i = c0; i < nt; ++i) {
      bool touch = own_tile(me, i, c0, nt);
      for (uint32_t kb = c0; kb < c1 && kb <= i && !touch; ++kb) touch = e->st.tile_nz[(size_t)i * nt + kb] != 0;
      if (!touch) continue;
      for (uint32_t kb = c0; kb < c1 && kb <= i; ++kb)
        if (e->nzL_host[(size_t)i * nt + kb]) (*need)[(size_t)i * nt + kb] = 1;
    }
  }
  return 1;
}

// pivot floors tol * |A_jj| of the matrix about to be factorised (rank-deficiency guard)
__global__ void k_pivot_floor(uint32_t n_pad, uint32_t ld, const double* __restrict__ A, double tol,
                              double* __restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_pad) out[j] = tol * fabs(A[(size_t)j * ld + j]);
}

// status block: int 0 = factorisation status, ints 2..3 = device pointer of the pivot floors (or null)
static int setup_status_block(Engine* e, const double* dA, uint32_t ld, hipStream_t s0) {
  BAE_HIP(hipMemsetAsync(e->flags.p, 0, 4 * sizeof(int), s0));
  const double tol = e->opt.pivot_rel_tolerance;
  if (tol > 0.0) {
    BAE_HIP(e->pivot_floor.alloc(ld));
    hipLaunchKernelGGL(k_pivot_floor, dim3((ld + 255) / 256), dim3(256), 0, s0, ld, ld, dA, tol, e->pivot_floor.p);
    BAE_HIP(hipGetLastError());
    const double* ptr = e->pivot_floor.p;
    BAE_HIP(hipMemcpyAsync(e->flags.p + 2, &ptr, sizeof(ptr), hipMemcpyHostToDevice, s0));
    BAE_HIP(hipStreamSynchronize(s0));  // `ptr` is a stack variable
  }
  return 0;
}

// Backward substitution of the distributed solve, panel [c0, c1): partial sums over the row tiles this rank
// owns below the square,  part[g][col] = sum over its tiles t = g, g + NG, ... of  L[t-rows, col]^T x[t-rows].
__global__ void __launch_bounds__(256)
k_back_partial(const double* __restrict__ A, uint32_t ld, uint32_t nblk, const uint32_t* __restrict__ tiles,
               uint32_t ntiles, uint32_t c0, uint32_t w, const double* __restrict__ x,
               const uint8_t* __restrict__ nz, double* __restrict__ part) {
  __shared__ double xs[NB];
  const uint32_t col = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y, ng = gridDim.y;
  double s = 0.0;
  for (uint32_t t = g; t < ntiles; t += ng) {
    const uint32_t i = tiles[t];
    __syncthreads();
    if (threadIdx.x < NB) xs[threadIdx.x] = x[(size_t)i * NB + threadIdx.x];
    __syncthreads();
    if (col < w && (!nz || nz[(size_t)i * nblk + c0 + (col >> 6)])) {
      const double* L = A + ((size_t)i * NB) * ld + (size_t)c0 * NB + col;
#pragma unroll 16
      for (int r = 0; r < NB; ++r) s += L[(size_t)r * ld] * xs[r];
    }
  }
  if (col < w) part[(size_t)g * w + col] = s;
}

// out[col] = sum over g (fixed order) of part[g][col]
__global__ void k_back_sum(const double* __restrict__ part, uint32_t ng, uint32_t w, double* __restrict__ out) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= w) return;
  double s = 0.0;
  for (uint32_t g = 0; g < ng; ++g) s += part[(size_t)g * w + col];
  out[col] = s;
}

// y[c0*64 + col] -= p[col]   (y = the rhs row of A)
__global__ void k_back_apply(double* __restrict__ yrow, const double* __restrict__ p, uint32_t w) {
  const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col < w) yrow[col] -= p[col];
}

int cholesky_solve_dist(Engine* e, double* dA, uint32_t ld, double* dx, int* status, const uint8_t* nz) {
  const uint32_t nblk = ld / NB, rank = (uint32_t)e->rank;
  BAE_HIP(e->invdiag.alloc((size_t)nblk * NB + (size_t)nblk * NB * NB + (size_t)nblk * NOPV * 64 +
                           (nblk + 1) / 2));
  double* dsgn = e->invdiag.p;
  double* linvT = dsgn + (size_t)nblk * NB;
  double* opbuf = linvT + (size_t)nblk * NB * NB;
  int* colneg = reinterpret_cast<int*>(opbuf + (size_t)nblk * NOPV * 64);
  const bool pat = nz && e->nzL_host.size() == (size_t)nblk * nblk;
  { const int prc = ensure_dist_plan(e, nblk, pat); if (prc) return prc; }
  const DistPlan& pl = e->dist_plan;
  const OwnMap& own = pl.map;
  const uint32_t KOUT = own.G, npanels = pl.nb, n_pad = nblk * NB;
  const uint32_t* dtiles = e->dist_tiles.p;
  // streams: chain, panel, bulk, side
  if (!e->stream3) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    BAE_HIP(hipStreamCreateWithPriority(&e->stream3, hipStreamNonBlocking, lo));
    BAE_HIP(hipStreamCreateWithPriority(&e->stream4, hipStreamNonBlocking, hi));
  }
  // (a native communicator without its duplicate — librccl lacks ncclCommSplit, or BA_HIP_ONE_COMM — carries the
  // side transfers too: they are then ordered into the chain stream, one communicator is never used on two streams)
  hipStream_t s0 = e->stream, s1 = e->stream2, s2 = e->stream3, s3 = (e->comm && !e->comm2) ? e->stream : e->stream4;
  // staging buffers: the square message; urgent / side messages of the busiest panel
  const size_t wmax = (size_t)KOUT * NB;
  BAE_HIP(e->dist_msg.alloc(dist_square_doubles(KOUT)));
  size_t cap_us = 1, cap_ur = 1, cap_ss = 1, cap_sr = 1;
  for (const DistPanel& pn : pl.panels) {
    size_t us = 0, ur = 0, ss = 0, sr = 0;
    const size_t w = (size_t)(pn.c1 - pn.c0) * NB;
    for (uint32_t mi : pn.msgs) {
      const DistMsg& m = pl.msgs[mi];
      const size_t cnt = (size_t)m.count * NB * w;
      const bool recv = std::find(m.dst.begin(), m.dst.end(), rank) != m.dst.end();
      if (m.src == rank) (m.urgent ? us : ss) += cnt;
      if (recv) (m.urgent ? ur : sr) += cnt;
    }
    cap_us = std::max(cap_us, us); cap_ur = std::max(cap_ur, ur);
    cap_ss = std::max(cap_ss, ss); cap_sr = std::max(cap_sr, sr);
  }
  BAE_HIP(e->dist_usend.alloc(cap_us)); BAE_HIP(e->dist_urecv.alloc(cap_ur));
  BAE_HIP(e->dist_ssend.alloc(cap_ss)); BAE_HIP(e->dist_srecv.alloc(cap_sr));
  const uint32_t NG = 32;
  BAE_HIP(e->dist_back.alloc((size_t)(NG + 1) * wmax));
  while (e->ev_dist.size() < (size_t)npanels * 5) {
    hipEvent_t a;
    BAE_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
    e->ev_dist.push_back(a);
  }
  auto EV = [&](uint32_t J, int k) { return e->ev_dist[(size_t)J * 5 + k]; };  // 0 urgent 1 packed 2 side 3 next 4 bulk
  e->cstats.factorisations++;
  { const int src = setup_status_block(e, dA, ld, s0); if (src) return src; }
  // everything queued on the chain stream so far (assembly, reduce-scatter unpack) precedes the other streams' work
  BAE_HIP(hipEventRecord(e->ev_dist[1], s0));
  BAE_HIP(hipStreamWaitEvent(s1, e->ev_dist[1], 0));
  BAE_HIP(hipStreamWaitEvent(s2, e->ev_dist[1], 0));
  BAE_HIP(hipStreamWaitEvent(s3, e->ev_dist[1], 0));
  if (rank == pl.panels[0].diag_owner)  // factor packet of tile 0
    hipLaunchKernelGGL(k_step_update<true>, dim3(1, 1), dim3(256), 0, s0, dA, ld, nblk, 0u, 0u, 0u, dsgn, opbuf,
                       colneg, e->flags.p, nz, nblk + 1u, (const uint32_t*)nullptr);
  // row list of a square: its tiles, then the rhs row — one small device array per plan (ensure_dist_plan)
  DBuf<uint32_t>& sq_list = e->dist_sq_list;
  static const bool no128 = getenv("BA_HIP_NO128") != nullptr;
  int rc = 0;
  size_t sq_off = 0;
  for (uint32_t J = 0; J < npanels && !rc; ++J) {
    const DistPanel& pn = pl.panels[J];
    const uint32_t c0 = pn.c0, c1 = pn.c1, wt = c1 - c0, w = wt * NB;
    const uint32_t* sq = sq_list.p + sq_off;
    sq_off += wt + 1;
    const bool last = c1 >= nblk;
    // ---- chain stream: the square ----------------------------------------------------------------------
    double* msg = e->dist_msg.p;
    const size_t n_cols = (size_t)(w + 1) * w;
    double* m_sgn = msg + n_cols;
    double* m_neg = m_sgn + w;
    double* m_op = m_neg + wt;
    const size_t msg_len = n_cols + w + wt + (size_t)wt * NOPV * 64;
    if (rank == pn.diag_owner) {
      // (its updates by the earlier panels: bulk of steps <= J-2 and the chain-stream update of step J-1,
      // both ordered before this point by the waits of step J-1)
      launch_panel_chain(s0, dA, ld, nblk, c0, c1, dsgn, opbuf, colneg, e->flags.p, nz, sq, wt + 1, true);
      hipLaunchKernelGGL(k_copy_panel_rows, dim3(w + 1), dim3(256), 0, s0, dA, ld, sq, wt, n_pad, c0 * NB, w, msg, 0);
      BAE_HIP(hipMemcpyAsync(m_sgn, dsgn + (size_t)c0 * NB, w * sizeof(double), hipMemcpyDeviceToDevice, s0));
      BAE_HIP(hipMemcpyAsync(m_neg, colneg + c0, wt * sizeof(int), hipMemcpyDeviceToDevice, s0));
      BAE_HIP(hipMemcpyAsync(m_op, opbuf + (size_t)c0 * NOPV * 64, (size_t)wt * NOPV * 64 * sizeof(double),
                             hipMemcpyDeviceToDevice, s0));
    }
    BAE_HIP(hipGetLastError());
    if ((rc = dist_broadcast(e, msg, msg_len, (int)pn.diag_owner, s0))) break;
    if (rank != pn.diag_owner) {
      hipLaunchKernelGGL(k_copy_panel_rows, dim3(w + 1), dim3(256), 0, s0, dA, ld, sq, wt, n_pad, c0 * NB, w, msg, 1);
      BAE_HIP(hipMemcpyAsync(dsgn + (size_t)c0 * NB, m_sgn, w * sizeof(double), hipMemcpyDeviceToDevice, s0));
      BAE_HIP(hipMemcpyAsync(colneg + c0, m_neg, wt * sizeof(int), hipMemcpyDeviceToDevice, s0));
      BAE_HIP(hipMemcpyAsync(opbuf + (size_t)c0 * NOPV * 64, m_op, (size_t)wt * NOPV * 64 * sizeof(double),
                             hipMemcpyDeviceToDevice, s0));
    }
    if (last) break;
    // ---- chain stream: the urgent block row (block J+1 of panel J) -------------------------------------
    // its tiles took the updates of the panels < J on the panel stream (step J-1: column block J)
    if (pn.own_u_count[rank]) {
      if (J > 0) BAE_HIP(hipStreamWaitEvent(s0, EV(J - 1, 3), 0));
      launch_panel_chain(s0, dA, ld, nblk, c0, c1, dsgn, opbuf, colneg, e->flags.p, nz, dtiles + pn.own_u_first[rank],
                         pn.own_u_count[rank], false);
    }
    auto run_messages = [&](bool urgent, hipStream_t sp, hipStream_t sc, double* sendbuf, double* recvbuf,
                            hipEvent_t packed_ev) -> int {
      // pack on `sp` (the stream that computed the rows), transfer + unpack on `sc`
      std::vector<DistXfer> xf;
      struct Unp { const uint32_t* tl; uint32_t n; double* buf; };
      std::vector<Unp> unp;
      size_t so = 0, ro = 0;
      for (uint32_t mi : pn.msgs) {
        const DistMsg& m = pl.msgs[mi];
        if (m.urgent != urgent) continue;
        const size_t cnt = (size_t)m.count * NB * w;
        if (m.src == rank) {
          hipLaunchKernelGGL(k_copy_panel_rows, dim3(m.count * NB), dim3(256), 0, sp, dA, ld, dtiles + m.first, m.count,
                             n_pad, c0 * NB, w, sendbuf + so, 0);
          for (uint32_t q : m.dst) xf.push_back({sendbuf + so, cnt, (int)q, true});
          so += cnt;
        }
        if (std::find(m.dst.begin(), m.dst.end(), rank) != m.dst.end()) {
          xf.push_back({recvbuf + ro, cnt, (int)m.src, false});
          unp.push_back({dtiles + m.first, m.count, recvbuf + ro});
          ro += cnt;
        }
      }
      BAE_HIP(hipGetLastError());
      if (sp != sc) {
        BAE_HIP(hipEventRecord(packed_ev, sp));
        BAE_HIP(hipStreamWaitEvent(sc, packed_ev, 0));
      }
      const int xrc = dist_exchange(e, xf, !urgent, sc);
      if (xrc) return xrc;
      for (const Unp& u : unp)
        hipLaunchKernelGGL(k_copy_panel_rows, dim3(u.n * NB), dim3(256), 0, sc, dA, ld, u.tl, u.n, n_pad, c0 * NB, w,
                           u.buf, 1);
      BAE_HIP(hipGetLastError());
      return 0;
    };
    if ((rc = run_messages(true, s0, s0, e->dist_usend.p, e->dist_urecv.p, nullptr))) break;
    BAE_HIP(hipEventRecord(EV(J, 0), s0));
    // ---- chain stream: panel J applied to the next square (and its rhs segment) + its first diagonal tile
    const uint32_t n1 = std::min(c1 + KOUT, nblk);  // end of column block J+1
    if (rank == pl.panels[J + 1].diag_owner) {
      if (J > 0) BAE_HIP(hipStreamWaitEvent(s0, EV(J - 1, 4), 0));  // the bulk updates of its tiles by the panels < J
      hipLaunchKernelGGL(k_step_update<true>, dim3(n1 - c1 + 1, n1 - c1), dim3(256), 0, s0, dA, ld, nblk, c1, c0, c1, dsgn, opbuf,
                         colneg, e->flags.p, nz, nblk + 1u, (const uint32_t*)(sq_list.p + sq_off));
    }
    // ---- panel stream: the other own rows of panel J, their messages -----------------------------------
    BAE_HIP(hipStreamWaitEvent(s1, EV(J, 0), 0));
    if (J > 0) BAE_HIP(hipStreamWaitEvent(s1, EV(J - 1, 2), 0));  // the side staging buffers are free again
    if (pn.own_r_count[rank])
      launch_panel_chain(s1, dA, ld, nblk, c0, c1, dsgn, opbuf, colneg, e->flags.p, nz, dtiles + pn.own_r_first[rank],
                         pn.own_r_count[rank], false);
    if ((rc = run_messages(false, s1, s3, e->dist_ssend.p, e->dist_srecv.p, EV(J, 1)))) break;
    BAE_HIP(hipEventRecord(EV(J, 2), s3));
    // ---- panel stream: panel J applied to the own tiles of column block J+1 below its square ------------
    if (pn.next_needs_side[rank] || own.n <= 1) BAE_HIP(hipStreamWaitEvent(s1, EV(J, 2), 0));  // rows received on the side stream
    if (J > 0) BAE_HIP(hipStreamWaitEvent(s1, EV(J - 1, 4), 0));
    if (n1 < nblk) {
      const uint32_t ncols = n1 - c1, m = nblk - n1;
      OwnMap o1 = own;
      o1.skip_rhs = 1;  // the rhs segment of column block J+1 went with the square's update on the chain stream
      if (!no128 && m >= 16 && ncols % 2u == 0 && c1 % 2u == 0 && own.G % 2u == 0) {
        const uint32_t m2 = m / 2, rect_cols = ncols / 2;
        const uint32_t nrow64 = (ncols * (1 + (m & 1u)) + 7) / 8 * 8;
        hipLaunchKernelGGL(k_update128<true>, dim3(nrow64 + m2 * rect_cols), dim3(256), 0, s1, dA, ld, nblk, c1, m2, c0, c1,
                           (const double*)dsgn, (const int*)colneg, 0u, nz, o1, nrow64, n1, rect_cols);
      } else {
        // 64-tiles: exactly the row tiles this rank owns below the square of panel J+1 (its two row lists are
        // adjacent) — a structurally zero tile of L takes no update, so the lists are complete
        const DistPanel& pq = pl.panels[J + 1];
        const uint32_t cnt = pq.own_u_count[rank] + pq.own_r_count[rank];
        if (cnt)
          hipLaunchKernelGGL(k_step_update<true>, dim3(cnt, ncols), dim3(256), 0, s1, dA, ld, nblk, c1, c0, c1, dsgn, opbuf,
                             colneg, e->flags.p, nz, nblk + 1u, dtiles + pq.own_u_first[rank]);
      }
    }
    BAE_HIP(hipEventRecord(EV(J, 3), s1));
    // ---- bulk stream: panel J applied to the own tiles right of column block J+1 -----------------------
    if (n1 < nblk) {
      BAE_HIP(hipStreamWaitEvent(s2, EV(J, 2), 0));
      BAE_HIP(hipStreamWaitEvent(s2, EV(J, 0), 0));
      launch_bulk_update(e, s2, dA, ld, nblk, n1, c0, c1, dsgn, colneg, nz, true, own,
                         e->dist_pairs.p + e->dist_pair_first[J], e->dist_npairs - e->dist_pair_first[J]);
      if (e->profiling) {
        // tile products formed by THIS rank: tiles (i, c), i >= c >= n1, it owns, both operand tiles of the
        // panel column structurally nonzero; per row class a suffix count of the nonzero tiles of the column
        double tiles = 0.0;
        std::vector<double> suffix((size_t)own.T * (nblk + 1));
        for (uint32_t kb = c0; kb < c1; ++kb) {
          for (uint32_t a = 0; a < own.T; ++a) suffix[(size_t)a * (nblk + 1) + nblk] = 0.0;
          for (uint32_t r = nblk; r-- > n1;)
            for (uint32_t a = 0; a < own.T; ++a)
              suffix[(size_t)a * (nblk + 1) + r] = suffix[(size_t)a * (nblk + 1) + r + 1] +
                  (((r / own.G) % own.T == a && (!pat || e->nzL_host[(size_t)r * nblk + kb])) ? 1.0 : 0.0);
          for (uint32_t c = n1; c < nblk; ++c) {
            if (pat && !e->nzL_host[(size_t)c * nblk + kb]) continue;
            const uint32_t cc = (c / own.G) % own.T;
            for (uint32_t a = 0; a < own.T; ++a)
              if (own.n <= 1 || own.tbl[a][cc] == rank) tiles += suffix[(size_t)a * (nblk + 1) + c];
            if (own.n <= 1 || own.tbl[cc][cc] == rank) tiles += 1.0;  // the rhs row
          }
        }
        e->kstats.syrk_flops += tiles * 2.0 * NB * NB * NB;
      }
    }
    BAE_HIP(hipEventRecord(EV(J, 4), s2));
  }
  if (rc) return rc;
  BAE_HIP(hipGetLastError());
  // every stream has drained into the chain stream's view before the backward substitution
  BAE_HIP(hipStreamSynchronize(s1));
  BAE_HIP(hipStreamSynchronize(s2));
  BAE_HIP(hipStreamSynchronize(s3));
  hipLaunchKernelGGL(k_linvT, dim3(nblk), dim3(256), 0, s0, (const double*)opbuf, (const double*)dsgn, linvT);
  // ---- backward substitution, panels from the last -------------------------------------------------------
  double* yrow = dA + (size_t)n_pad * ld;
  double* part = e->dist_back.p;
  double* psum = part + (size_t)NG * wmax;
  for (uint32_t J = npanels; J-- > 0 && !rc;) {
    const DistPanel& pn = pl.panels[J];
    const uint32_t c0 = pn.c0, c1 = pn.c1, w = (c1 - c0) * NB;
    if (c1 < nblk) {
      const uint32_t nt = pn.own_u_count[rank] + pn.own_r_count[rank];  // (the two lists are adjacent)
      const uint32_t ng = std::min(NG, std::max(nt, 1u));
      if (nt) {
        hipLaunchKernelGGL(k_back_partial, dim3((w + 255) / 256, ng), dim3(256), 0, s0, (const double*)dA, ld, nblk,
                           dtiles + pn.own_u_first[rank], nt, c0, w, (const double*)dx, nz, part);
        hipLaunchKernelGGL(k_back_sum, dim3((w + 255) / 256), dim3(256), 0, s0, (const double*)part, ng, w, psum);
      } else {
        BAE_HIP(hipMemsetAsync(psum, 0, (size_t)w * sizeof(double), s0));
      }
      BAE_HIP(hipGetLastError());
      if ((rc = dist_allreduce_stream(e, psum, w, s0))) break;
      hipLaunchKernelGGL(k_back_apply, dim3((w + 255) / 256), dim3(256), 0, s0, yrow + (size_t)c0 * NB, (const double*)psum, w);
    }
    launch_backward(s0, dA, ld, nblk, (const double*)linvT, dx, nz, c0, c1);
  }
  if (rc) return rc;
  BAE_HIP(hipGetLastError());
  // the pivot status of every diagonal owner
  int st = 0;
  BAE_HIP(hipMemcpyAsync(&st, e->flags.p, sizeof(int), hipMemcpyDeviceToHost, s0));
  BAE_HIP(hipStreamSynchronize(s0));
  double sd = (double)st;
  BAE_HIP(hipMemcpy(e->scalars_out.p, &sd, sizeof(double), hipMemcpyHostToDevice));
  if (shard_allreduce(e, e->scalars_out.p, 1, 0) != 0) return e->fail_msg("allreduce hook failed");
  BAE_HIP(hipMemcpy(&sd, e->scalars_out.p, sizeof(double), hipMemcpyDeviceToHost));
  *status = sd > 0.5 ? 1 : 0;
  return 0;
}

// Solve on the padded lower storage dA ((n_pad + 1) x ld, n_pad = ld multiple of 64; the
// rhs is row n_pad).  dx receives n_pad doubles (the first n are the solution).
// Marginal covariance of the trailing K unknowns from the factor left by cholesky_solve:
// (S^-1)_kk = G^T D_k G with G = (L_kk)^-1, the inverse of the K x K diagonal block of L (the rows
// behind it are padding and decoupled).  The diagonal tiles of L exist as their inverse-transposes
// (linvT); when the K rows straddle a tile boundary, G = [[Ga, 0], [-Gb L_ba Ga, Gb]] with the
// off-diagonal piece L_ba read from the factorised A.  Replaces the K extra solves with unit vectors
// of BundleAdjuster.cpp:771-784 (same numbers: the unit vectors only excite these rows).
int trailing_marginals(Engine* e, const double* dA, uint32_t ld, uint32_t first, uint32_t K, double* cov) {
  const uint32_t nblk = ld / NB;
  if (!e->invdiag.p || first + K > ld) return e->fail_msg("no factor to read the marginals from");
  const double* dsgn = e->invdiag.p;
  const double* linvT = dsgn + (size_t)nblk * NB;
  BAE_HIP(hipStreamSynchronize(e->stream));
  const uint32_t t = first / NB, o = first % NB;
  const uint32_t a = std::min(K, NB - o), b = K - a;  // rows in tile t / in tile t + 1
  std::vector<double> G((size_t)K * K, 0.0), sg(K), tile((size_t)NB * NB);
  BAE_HIP(hipMemcpy(sg.data(), dsgn + first, K * sizeof(double), hipMemcpyDeviceToHost));
  BAE_HIP(hipMemcpy(tile.data(), linvT + (size_t)t * NB * NB, tile.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (uint32_t r = 0; r < a; ++r)
    for (uint32_t c = 0; c <= r; ++c) G[(size_t)r * K + c] = tile[(size_t)(o + c) * NB + (o + r)];  // L^-1[r][c] = L^-T[c][r]
  if (b) {
    BAE_HIP(hipMemcpy(tile.data(), linvT + (size_t)(t + 1) * NB * NB, tile.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < b; ++r)
      for (uint32_t c = 0; c <= r; ++c) G[(size_t)(a + r) * K + a + c] = tile[(size_t)c * NB + r];
    std::vector<double> Lba((size_t)b * a);
    BAE_HIP(hipMemcpy2D(Lba.data(), a * sizeof(double), dA + (size_t)(t + 1) * NB * ld + (size_t)t * NB + o,
                        ld * sizeof(double), a * sizeof(double), b, hipMemcpyDeviceToHost));
    // -Gb Lba Ga
    for (uint32_t r = 0; r < b; ++r)
      for (uint32_t c = 0; c < a; ++c) {
        double s = 0.0;
        for (uint32_t x = 0; x < b; ++x)
          for (uint32_t y = 0; y < a; ++y) s += G[(size_t)(a + r) * K + a + x] * Lba[(size_t)x * a + y] * G[(size_t)y * K + c];
        G[(size_t)(a + r) * K + c] = -s;
      }
  }
  for (uint32_t r = 0; r < K; ++r)
    for (uint32_t c = 0; c < K; ++c) {
      double s = 0.0;
      for (uint32_t j = 0; j < K; ++j) s += G[(size_t)j * K + r] * sg[j] * G[(size_t)j * K + c];
      cov[(size_t)r * K + c] = s;
    }
  return 0;
}

int cholesky_solve(Engine* e, double* dA, uint32_t n, uint32_t ld, double* dx, int* status,
                   const uint8_t* nz) {
  (void)n;
  const uint32_t nblk = ld / NB;
  BAE_HIP(e->invdiag.alloc((size_t)nblk * NB + (size_t)nblk * NB * NB + (size_t)nblk * NOPV * 64 +
                           (nblk + 1) / 2));
  double* dsgn = e->invdiag.p;                                  // pivot signs
  double* linvT = dsgn + (size_t)nblk * NB;                     // inverse-transposed diagonal tiles
  double* opbuf = linvT + (size_t)nblk * NB * NB;               // factor packets (k_step_update)
  int* colneg = reinterpret_cast<int*>(opbuf + (size_t)nblk * NOPV * 64);  // per tile column: any d_k < 0
  static const bool no_lookahead = getenv("BA_HIP_NO_LOOKAHEAD") != nullptr;  // A/B switches
  static const bool bulk_full = getenv("BA_HIP_BULK_FULL") != nullptr;
  static const uint32_t bulk_full_m =
      getenv("BA_HIP_BULK_FULL_M") ? (uint32_t)atoi(getenv("BA_HIP_BULK_FULL_M")) : 128u;
  hipStream_t s0 = e->stream, s1 = no_lookahead ? e->stream : e->stream2;
  // Tiles per outer panel: every panel costs one read-modify-write pass over the trailing
  // matrix (8 n^3 / (3 * 64 KOUT) bytes in total), the serial chain inside a panel grows with
  // KOUT.  Small systems are chain-bound (KOUT = 4), large ones HBM-bound on the C tiles.
  const uint32_t KOUT = choose_kout(nblk);
  const uint32_t npanels = (nblk + KOUT - 1) / KOUT;
  while (e->ev_panel.size() < npanels) {
    hipEvent_t a, b;
    BAE_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
    BAE_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
    e->ev_panel.push_back(a);
    e->ev_bulk.push_back(b);
  }
  { const int src = setup_status_block(e, dA, ld, s0); if (src) return src; }
#ifdef BAE_TIME128
  static unsigned long long* t128_dev = nullptr;
  if (getenv("BA_HIP_TRACE_FILE")) {
    if (!t128_dev) {
      BAE_HIP(hipMalloc(&t128_dev, (size_t)T128_CAP * T128_REC * sizeof(unsigned long long)));
      BAE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_t128_buf), &t128_dev, sizeof(t128_dev)));
    }
    const unsigned int zero = 0;
    BAE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_t128_n), &zero, sizeof(zero)));
  }
#endif
  // BA_HIP_SQUARE=1 (measurement switch, read on every call): the diagonal squares of the sub-panels go to
  // k_square / k_rowpanel — one workgroup per square instead of two launches per tile column.  Measured
  // SLOWER than the per-column chain at configs[1] (5.07 / 4.38 ms at widths 4 / 2 against 3.73 ms: the chain
  // overlaps every factorisation with the other tiles' updates of the same launch, a single workgroup
  // serialises them on one CU's matrix pipes — profiles/r03_square_kernel.txt).
  const char* sqe = getenv("BA_HIP_SQUARE");
  const bool sq = nblk < 512 && sqe && atoi(sqe) != 0;
  if (sq) {
    if (!e->square_attr_set) {  // (per engine = per device: function attributes belong to the device)
      BAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_square), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)sizeof(SquareLds)));
      BAE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowpanel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)sizeof(RowPanelLds)));
      e->square_attr_set = true;
    }
  } else {
    // factor packet of tile 0 (nothing to update: one workgroup)
    hipLaunchKernelGGL(k_step_update<true>, dim3(1, 1), dim3(256), 0, s0, dA, ld, nblk, 0u, 0u, 0u, dsgn, opbuf,
                       colneg, e->flags.p, nz, nblk + 1u, (const uint32_t*)nullptr);
  }
  // Look-ahead: the trailing update of panel J is split into (a) the columns of the NEXT
  // panel — on the critical path, stream s0 — and (b) everything right of it — stream s1,
  // overlapping the serial factorisation of the next panel.
  int prev_bulk = -1;  // index of the last recorded ev_bulk
  uint32_t pj = 0;
  for (uint32_t J = 0; J < nblk; J += KOUT, ++pj) {
    const uint32_t Jend = J + KOUT < nblk ? J + KOUT : nblk;
    launch_panel_chain(s0, dA, ld, nblk, J, Jend, dsgn, opbuf, colneg, e->flags.p, nz, nullptr, 0, false, sq);
    if (Jend >= nblk) break;
    BAE_HIP(hipEventRecord(e->ev_panel[pj], s0));
    const uint32_t a_end = Jend + KOUT < nblk ? Jend + KOUT : nblk;  // columns of the next panel
    // (a) next panel's columns: needs every earlier bulk update of those columns
    if (prev_bulk >= 0) BAE_HIP(hipStreamWaitEvent(s0, e->ev_bulk[prev_bulk], 0));
    launch_next_panel_update(s0, dA, ld, nblk, Jend, a_end - Jend, J, Jend, dsgn, opbuf, colneg, e->flags.p, nz, !sq);
    // (b) the rest, concurrently with the next panel's factorisation
    if (a_end < nblk) {
      BAE_HIP(hipStreamWaitEvent(s1, e->ev_panel[pj], 0));
      // large trailing matrices (the serial chain is negligible beside them): full occupancy;
      // otherwise the capped variant leaves the chain room
      const uint32_t m = nblk - a_end;
      launch_bulk_update(e, s1, dA, ld, nblk, a_end, J, Jend, dsgn, colneg, nz,
                         no_lookahead || bulk_full || m >= bulk_full_m, own_map_single());
      BAE_HIP(hipEventRecord(e->ev_bulk[pj], s1));
      prev_bulk = (int)pj;
      if (e->profiling) {
        // algorithmic flops of this launch: per panel column kb the tiles (i, c), i >= c >= a_end,
        // whose two operand tiles are structurally nonzero, plus the rhs row
        double tiles = 0.0;
        for (uint32_t kb = J; kb < Jend; ++kb) {
          double mk = (double)(nblk - a_end);
          if (nz && e->nzL_host.size() == (size_t)nblk * nblk) {
            mk = 0.0;
            for (uint32_t r = a_end; r < nblk; ++r) mk += e->nzL_host[(size_t)r * nblk + kb];
          }
          tiles += mk * (mk + 1) / 2 + mk;
        }
        e->kstats.syrk_flops += tiles * 2.0 * NB * NB * NB;
      }
    }
  }
  BAE_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_linvT, dim3(nblk), dim3(256), 0, s0, (const double*)opbuf, (const double*)dsgn,
                       linvT);
  launch_backward(s0, dA, ld, nblk, (const double*)linvT, dx, nz);
  BAE_HIP(hipGetLastError());
  int st = 0;
  BAE_HIP(hipMemcpyAsync(&st, e->flags.p, sizeof(int), hipMemcpyDeviceToHost, s0));
  BAE_HIP(hipStreamSynchronize(s0));
  *status = st;
#ifdef BAE_TIME128
  if (t128_dev && getenv("BA_HIP_TRACE_FILE")) {
    BAE_HIP(hipDeviceSynchronize());
    unsigned int n = 0;
    BAE_HIP(hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_t128_n), sizeof(n)));
    n = std::min(n, T128_CAP);
    std::vector<unsigned long long> rec((size_t)n * T128_REC);
    if (n) BAE_HIP(hipMemcpy(rec.data(), t128_dev, rec.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(getenv("BA_HIP_TRACE_FILE"), "wb")) {
      fwrite(rec.data(), sizeof(unsigned long long), rec.size(), f);
      fclose(f);
    }
  }
#endif
  return 0;
}

}  // namespace bae
