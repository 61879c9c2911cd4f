// Dense FP64 solve of the reduced camera system S delta = rhs on gfx950.
//
// Replaces CalculateGn (BundleAdjuster.cpp:748-833): the reference converts the dense
// s_ to a sparse view and runs Eigen::SimplicialLDLT<Upper> (or dense LDLT<Upper>).
// Here: two-level blocked right-looking Cholesky on the LOWER storage (row-major,
// leading dimension ld, multiple of 64):
//
//   for every outer panel J of KOUT = 4 tile columns (256 columns)
//     for every 64-column tile jj of the panel
//       k_potrf64   one wavefront factorises the 64x64 diagonal tile in registers
//                   (row per lane, pivots broadcast with v_readlane: no barriers)
//       k_trsm64    rows below: X L_jj^T = A, one thread per row (forward substitution,
//                   L_jj broadcast from LDS)
//       k_update    in-panel update of the remaining tile columns of the panel (K = 64)
//     k_update      trailing update of everything right of the panel with K = 256
//
// The products run on the FP64 matrix cores (v_mfma_f64_16x16x4_f64 — the one true dense
// contraction of the path); accumulating 256 columns per pass over the trailing matrix
// quarters the HBM read-modify-write traffic of the C tiles compared with K = 64.
// The factorisation is L D L^T with D = diag(+-1) (L carries sqrt|pivot|): for SPD
// systems it IS the Cholesky factor, and like the reference's un-pivoted LDL^T it does
// not break down on the indefinite / nearly singular systems that ill-posed gauges
// produce (BundleAdjuster.cpp:752-761).
// The right-hand side rides along as one extra row below the matrix, so the forward
// substitution L y = b is a by-product; k_backward then solves L^T x = y block row by
// block row.  The system is SPD for well-posed problems (masked parameters carry 1e6 on
// the diagonal); a non-positive pivot raises the status flag (-> FactorizationError,
// BundleAdjuster.cpp:756-759).
#include "engine.h"

namespace bae {

static const int NB = 64;        // tile size
static const int KOUT = 4;       // tiles per outer panel
static const int LDT = NB + 2;   // LDS row stride (doubles): conflict-free MFMA operand reads
static const int LDP = NB + 1;   // LDS row stride for row-per-lane access

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------
// 64x64 Cholesky of the diagonal tile k by ONE wavefront.  Lane r keeps row r in
// registers.  Step j: the pivot is broadcast by v_readlane, 1/sqrt(pivot) comes from
// v_rsq_f64 refined by two Newton steps (no division on the serial chain), every lane
// scales its column-j entry and publishes it to an LDS column buffer, then updates its
// columns right of j with the column entries read back as wave-uniform (broadcast) LDS
// reads.  Fully unrolled: every register index is static; no barriers (single wave, LDS
// operations of one wave execute in order).
// Also stores 1/L[j][j] for the triangular solves.
__global__ void __launch_bounds__(64)
k_potrf64(double* __restrict__ A, uint32_t ld, uint32_t k, double* __restrict__ dinv_out,
          double* __restrict__ dsgn_out, int* __restrict__ status) {
  __shared__ double T[NB][LDP];
  __shared__ double colbuf[2][NB];
  const int lane = threadIdx.x;
  double* Akk = A + ((size_t)k * NB) * ld + (size_t)k * NB;
  {
    double tmp[NB];  // all 64 row loads in flight before the first use
#pragma unroll
    for (int r = 0; r < NB; ++r) tmp[r] = Akk[(size_t)r * ld + lane];  // coalesced rows
#pragma unroll
    for (int r = 0; r < NB; ++r) T[r][lane] = tmp[r];
  }
  __syncthreads();
  double a[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) a[c] = (c <= lane) ? T[lane][c] : 0.0;
  int bad = 0;
  double my_dinv = 0.0, my_sgn = 1.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const double d = readlane_f64(a[j], j);
    const double ad = fabs(d);
    if (!(ad > 0.0) || !(ad < 1e300)) bad = 1;  // zero / NaN / Inf pivot
    const double sgn = d < 0.0 ? -1.0 : 1.0;
    const double dd = (ad > 0.0 && ad < 1e300) ? ad : 1.0;
    // y ~ 1/sqrt(dd): hardware estimate + 2 Newton-Raphson steps y <- y (1.5 - 0.5 dd y^2)
    double y = __builtin_amdgcn_rsq(dd);
    y = y * fma(-0.5 * dd * y, y, 1.5);
    y = y * fma(-0.5 * dd * y, y, 1.5);
    if (lane == j) { my_dinv = y; my_sgn = sgn; }
    a[j] *= y * sgn;  // lanes below: L[r][j] = d_j A[r][j] / sqrt|d|
    if (lane == j) a[j] = fabs(a[j]);  // L[j][j] = sqrt|d|
    colbuf[j & 1][lane] = a[j];
    const double sa = sgn * a[j];
#pragma unroll
    for (int c = j + 1; c < NB; ++c) a[c] -= sa * colbuf[j & 1][c];  // d_j L[r][j] L[c][j]
  }
#pragma unroll
  for (int c = 0; c < NB; ++c) T[lane][c] = a[c];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NB; ++r)
    if (lane <= r) Akk[(size_t)r * ld + lane] = T[r][lane];
  dinv_out[(size_t)k * NB + lane] = my_dinv;
  dsgn_out[(size_t)k * NB + lane] = my_sgn;
  if (lane == 0 && bad) atomicExch(status, 1);
}

// ---------------------------------------------------------------------------------
// Rows below the diagonal tile: X L_kk^T = A_ik, one thread per row, column-oriented
// (right-looking) forward substitution: once x_p is known every remaining entry of the
// row is updated independently, s_j -= x_p L[j][p], so the 2016 FMAs per row have no
// serial dependence beyond one FMA per step.  L_kk is staged TRANSPOSED in LDS so that
// the entries needed at step p (L[p+1..63][p]) are contiguous: wave-uniform 16-byte LDS
// reads (broadcast).  One wavefront per 64-row block; the last block is the rhs row.
__global__ void __launch_bounds__(64)
k_trsm64(const double* __restrict__ Lkk, const double* __restrict__ dinv,
         const double* __restrict__ dsgn, double* __restrict__ Apanel, uint32_t ld,
         uint32_t nrowblk) {
  __shared__ double Xs[NB][LDP];
  __shared__ __attribute__((aligned(16))) double LsT[NB][LDT];  // LsT[p][j] = L[j][p] / L[j][j]
  __shared__ double dv[NB];
  const int lane = threadIdx.x;
  const int rows = (blockIdx.x + 1 == nrowblk) ? 1 : NB;  // the last block is the rhs row
  double* Aik = Apanel + ((size_t)blockIdx.x * NB) * ld;
  {
    double ta[NB], tl[NB];  // all loads in flight before the first use
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      ta[r] = (r < rows) ? Aik[(size_t)r * ld + lane] : 0.0;
      tl[r] = Lkk[(size_t)r * ld + lane];
    }
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      Xs[r][lane] = ta[r];
      // row r of L, element (r, lane), pre-scaled by 1/L[r][r]: with t_j = s_j / L[j][j]
      // the substitution needs no multiply on its serial chain (x_p = t_p)
      // (column sign d_lane, row scale d_r / L[r][r]:  x_j = d_j (a_j - sum x_p d_p L[j][p]) / L[j][j])
      LsT[lane][r] = tl[r] * dinv[r] * dsgn[r] * dsgn[lane];
    }
  }
  dv[lane] = dinv[lane] * dsgn[lane];
  __syncthreads();
  double s[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) s[j] = Xs[lane][j] * dv[j];
  // software pipeline: the L entries of step p+1 are fetched (wave-uniform LDS reads)
  // while step p computes; sched_barrier keeps the compiler from sinking the loads
  double cur[NB], nxt[NB];
#pragma unroll
  for (int j = 1; j < NB; ++j) cur[j] = LsT[0][j];
#pragma unroll
  for (int p = 0; p < NB; ++p) {
    if (p + 1 < NB) {
#pragma unroll
      for (int j = p + 2; j < NB; ++j) nxt[j] = LsT[p + 1][j];
    }
    __builtin_amdgcn_sched_barrier(0);
    const double x = s[p];
#pragma unroll
    for (int j = p + 1; j < NB; ++j) s[j] -= x * cur[j];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = p + 2; j < NB; ++j) cur[j] = nxt[j];
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) Xs[lane][j] = s[j];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NB; ++r)
    if (r < rows) Aik[(size_t)r * ld + lane] = Xs[r][lane];
}

// ---------------------------------------------------------------------------------
// C(64x64) += X(64x64) * Y(64x64)^T on the FP64 matrix cores; X, Y in LDS (stride LDT).
// 4 waves: wave w owns rows [32*(w>>1), +32) x cols [32*(w&1), +32) as 2x2 MFMA tiles.
// Fragment maps of v_mfma_f64_16x16x4_f64: A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15]; C/D: col = lane&15, row = (lane>>4) + 4*reg.
__device__ __forceinline__ void tile_mma(const double (*X)[LDT], const double (*Y)[LDT],
                                         int wave, int lane, double4_t acc[2][2]) {
  const int rb = 32 * (wave >> 1), cb = 32 * (wave & 1);
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll 4
  for (int k0 = 0; k0 < NB; k0 += 4) {
    const double a0 = X[rb + li][k0 + lk];
    const double a1 = X[rb + 16 + li][k0 + lk];
    const double b0 = Y[cb + li][k0 + lk];
    const double b1 = Y[cb + 16 + li][k0 + lk];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
}

// load a 64-row x 64-col tile (rows beyond `rows` are zero) into LDS
__device__ __forceinline__ void load_tile(double (*T)[LDT], const double* __restrict__ src,
                                          uint32_t ld, int rows, int tid) {
  for (int idx = tid; idx < NB * (NB / 2); idx += 256) {
    const int r = idx / (NB / 2), c2 = (idx % (NB / 2)) * 2;
    double2 v = make_double2(0.0, 0.0);
    if (r < rows) v = *reinterpret_cast<const double2*>(src + (size_t)r * ld + c2);
    T[r][c2] = v.x;
    T[r][c2 + 1] = v.y;
  }
}

// same, columns scaled by the pivot signs d_k = +-1 (C -= X D Y^T)
__device__ __forceinline__ void load_tile_signed(double (*T)[LDT], const double* __restrict__ src,
                                                 uint32_t ld, const double* __restrict__ sg, int tid) {
  for (int idx = tid; idx < NB * (NB / 2); idx += 256) {
    const int r = idx / (NB / 2), c2 = (idx % (NB / 2)) * 2;
    const double2 v = *reinterpret_cast<const double2*>(src + (size_t)r * ld + c2);
    T[r][c2] = v.x * sg[c2];
    T[r][c2 + 1] = v.y * sg[c2 + 1];
  }
}

// Update of the tiles (i, c), c in [c0, c0 + gridDim.y), i in [c, nblk] (i == nblk: the
// rhs row) with the tile columns [kb0, kb1):
//     A_ic -= sum_kb A_i,kb * A_c,kb^T
// grid = (nblk - c0 + 1, number of tile columns); blockIdx.x counts rows from c.
__global__ void __launch_bounds__(256)
k_update(double* __restrict__ A, uint32_t ld, uint32_t nblk, uint32_t c0, uint32_t kb0,
         uint32_t kb1, const double* __restrict__ dsgn) {
  __shared__ double X[NB][LDT];
  __shared__ double Y[NB][LDT];
  const uint32_t c = c0 + blockIdx.y;
  const uint32_t i = c + blockIdx.x;
  if (i > nblk) return;
  const int rows = (i == nblk) ? 1 : NB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (uint32_t kb = kb0; kb < kb1; ++kb) {
    const double* Aik = A + ((size_t)i * NB) * ld + (size_t)kb * NB;
    const double* Ack = A + ((size_t)c * NB) * ld + (size_t)kb * NB;
    if (kb != kb0) __syncthreads();
    load_tile(X, Aik, ld, rows, tid);
    load_tile_signed(Y, Ack, ld, dsgn + (size_t)kb * NB, tid);
    __syncthreads();
    tile_mma(X, Y, wave, lane, acc);
  }
  double* Aic = A + ((size_t)i * NB) * ld + (size_t)c * NB;
  const int rb = 32 * (wave >> 1), cb = 32 * (wave & 1);
  const bool diag = (i == c);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = rb + 16 * ti + (lane >> 4) + 4 * reg;
        const int cc = cb + 16 * tj + (lane & 15);
        if (r < rows && (!diag || cc <= r)) Aic[(size_t)r * ld + cc] -= acc[ti][tj][reg];
      }
}

// Backward substitution, block row i (from the last to the first):
//   L_ii^T x_i = y_i ;  y[0 : i*NB] -= L[i-block, 0:i*NB]^T x_i
// y lives in the rhs row of A.  Wave 0 of every workgroup solves the 64x64 triangular
// system (column sweep, pivots broadcast by readlane); then each thread updates one
// column.  Workgroup 0 also stores x_i.
__global__ void __launch_bounds__(256)
k_backward(double* __restrict__ A, uint32_t ld, uint32_t i, uint32_t nblk,
           const double* __restrict__ dinv, double* __restrict__ x) {
  __shared__ double Ls[NB][LDP];
  __shared__ double xi[NB];
  const int tid = threadIdx.x;
  const double* Aii = A + ((size_t)i * NB) * ld + (size_t)i * NB;
  {
    double tmp[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) tmp[u] = Aii[(size_t)(4 * u + (tid >> 6)) * ld + (tid & 63)];
#pragma unroll
    for (int u = 0; u < 16; ++u) Ls[4 * u + (tid >> 6)][tid & 63] = tmp[u];
  }
  __syncthreads();
  if (tid < NB) {
    double yv = A[((size_t)nblk * NB) * ld + (size_t)i * NB + tid];
    // for j = 63..0: x_j = y_j / L[j][j]; y_r -= L[j][r] x_j for r < j
    for (int j = NB - 1; j >= 0; --j) {
      const double xj = readlane_f64(yv, j) * dinv[(size_t)i * NB + j];
      if (tid == j) yv = xj;
      else if (tid < j) yv -= Ls[j][tid] * xj;
    }
    xi[tid] = yv;
    if (blockIdx.x == 0) x[(size_t)i * NB + tid] = yv;
  }
  __syncthreads();
  const uint32_t col = blockIdx.x * 256 + tid;
  if (col < i * NB) {
    const double* Li = A + ((size_t)i * NB) * ld + col;
    double s = 0.0;
#pragma unroll 8
    for (int r = 0; r < NB; ++r) s += Li[(size_t)r * ld] * xi[r];
    A[((size_t)nblk * NB) * ld + col] -= s;
  }
}

// Solve on the padded lower storage dA ((n_pad + 1) x ld, n_pad = ld multiple of 64; the
// rhs is row n_pad).  dx receives n_pad doubles (the first n are the solution).
int cholesky_solve(Engine* e, double* dA, uint32_t n, uint32_t ld, double* dx, int* status) {
  (void)n;
  const uint32_t nblk = ld / NB;
  BAE_HIP(e->invdiag.alloc((size_t)2 * nblk * NB));  // 1 / diag(L), then the pivot signs
  double* dsgn = e->invdiag.p + (size_t)nblk * NB;
  BAE_HIP(hipMemsetAsync(e->flags.p, 0, sizeof(int), e->stream));
  for (uint32_t J = 0; J < nblk; J += KOUT) {
    const uint32_t Jend = J + KOUT < nblk ? J + KOUT : nblk;
    for (uint32_t jj = J; jj < Jend; ++jj) {
      hipLaunchKernelGGL(k_potrf64, dim3(1), dim3(64), 0, e->stream, dA, ld, jj, e->invdiag.p,
                         dsgn, e->flags.p);
      hipLaunchKernelGGL(k_trsm64, dim3(nblk - jj), dim3(64), 0, e->stream,
                         (const double*)(dA + ((size_t)jj * NB) * ld + (size_t)jj * NB),
                         (const double*)(e->invdiag.p + (size_t)jj * NB),
                         (const double*)(dsgn + (size_t)jj * NB),
                         dA + ((size_t)(jj + 1) * NB) * ld + (size_t)jj * NB, ld, nblk - jj);
      if (jj + 1 < Jend) {
        // in-panel update of the panel's remaining tile columns with tile column jj
        hipLaunchKernelGGL(k_update, dim3(nblk - (jj + 1) + 1, Jend - (jj + 1)), dim3(256), 0,
                           e->stream, dA, ld, nblk, jj + 1, jj, jj + 1, (const double*)dsgn);
      }
    }
    if (Jend < nblk) {
      // trailing update right of the panel, K = 64 * (Jend - J)
      e->prof_begin(e->ev_syrk);
      hipLaunchKernelGGL(k_update, dim3(nblk - Jend + 1, nblk - Jend), dim3(256), 0, e->stream, dA,
                         ld, nblk, Jend, J, Jend, (const double*)dsgn);
      e->prof_end(e->ev_syrk);
      if (e->profiling) {
        const double m = (double)(nblk - Jend);
        e->kstats.syrk_flops += (m * (m + 1) / 2 + m) * 2.0 * NB * NB * NB * (Jend - J);
      }
    }
  }
  BAE_HIP(hipGetLastError());
  for (uint32_t ii = nblk; ii-- > 0;) {
    const uint32_t cols = ii * NB;
    const uint32_t grid = cols == 0 ? 1 : (cols + 255) / 256;
    hipLaunchKernelGGL(k_backward, dim3(grid), dim3(256), 0, e->stream, dA, ld, ii, nblk,
                       (const double*)e->invdiag.p, dx);
  }
  BAE_HIP(hipGetLastError());
  int st = 0;
  BAE_HIP(hipMemcpyAsync(&st, e->flags.p, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  BAE_HIP(hipStreamSynchronize(e->stream));
  *status = st;
  return 0;
}

}  // namespace bae
