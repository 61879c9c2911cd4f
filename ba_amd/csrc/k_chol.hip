// Dense FP64 solve of the reduced camera system S delta = rhs on gfx950.
//
// Replaces CalculateGn (BundleAdjuster.cpp:748-833): the reference converts the dense
// s_ to a sparse view and runs Eigen::SimplicialLDLT<Upper> (or dense LDLT<Upper>).
// Here: blocked right-looking Cholesky on the LOWER storage (row-major, leading
// dimension ld, multiple of 64) with
//   * k_potrf_inv   — 64x64 diagonal tile factorised in LDS, plus its inverse, so that
//   * k_panel       — the panel solve A_ik L_kk^-T becomes a matrix product,
//   * k_syrk        — trailing update A_ij -= A_ik A_jk^T,
// both products on the FP64 matrix cores (v_mfma_f64_16x16x4_f64: the one true dense
// contraction of the path).  The right-hand side rides along as one extra row below
// the matrix, so the forward substitution L y = b is a by-product of the panel steps;
// k_backward then solves L^T x = y block row by block row using the stored inverse
// diagonal tiles.
// The system is SPD for well-posed problems (masked parameters carry 1e6 on the
// diagonal); a non-positive pivot raises the status flag (-> FactorizationError,
// BundleAdjuster.cpp:756-759).
#include "engine.h"

namespace bae {

static const int NB = 64;        // tile size
static const int LDT = NB + 2;   // LDS row stride in doubles: conflict-free MFMA operand reads

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------
// Factorise the diagonal tile k in place (lower), compute inv(L_kk) (lower) into
// invd[k], one workgroup of 256 threads.  status != 0 on a non-positive pivot.
__global__ void __launch_bounds__(256)
k_potrf_inv(double* __restrict__ A, uint32_t ld, uint32_t k, double* __restrict__ invd,
            int* __restrict__ status) {
  // lower triangle + diagonal: L; strict upper triangle: inv(L)^T; dinv: diagonal of inv(L)
  __shared__ double T[NB][LDT];
  __shared__ double dinv[NB];
  __shared__ int bad;
  const int tid = threadIdx.x;
  double* Akk = A + ((size_t)k * NB) * ld + (size_t)k * NB;
  if (tid == 0) bad = 0;
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx / NB, c = idx % NB;
    T[r][c] = (c <= r) ? Akk[(size_t)r * ld + c] : 0.0;
  }
  __syncthreads();
  // right-looking unblocked Cholesky on the lower triangle
  for (int j = 0; j < NB; ++j) {
    if (tid == 0) {
      const double d = T[j][j];
      if (!(d > 0.0)) bad = 1;
      T[j][j] = sqrt(d > 0.0 ? d : 1.0);
    }
    __syncthreads();
    const double dj = T[j][j];
    if (tid > j && tid < NB) T[tid][j] = T[tid][j] / dj;
    __syncthreads();
    const int m = NB - 1 - j;  // trailing size
    for (int idx = tid; idx < m * m; idx += 256) {
      const int rr = idx / m, cc = idx % m;
      if (cc <= rr) {
        const int r = j + 1 + rr, c = j + 1 + cc;
        T[r][c] -= T[r][j] * T[c][j];
      }
    }
    __syncthreads();
  }
  // inverse of the lower-triangular tile, column c solves L x = e_c by forward
  // substitution; 4 threads per column split each dot product.  x_r for r > c is kept
  // at T[c][r] (the unused upper triangle), x_c in dinv[c].
  {
    const int c = tid >> 2, part = tid & 3;
    for (int r = 0; r < NB; ++r) {
      double s = 0.0;
      if (r > c) {
        for (int p = c + 1 + part; p < r; p += 4) s += T[r][p] * T[c][p];
        if (part == 0) s += T[r][c] * dinv[c];
      }
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if (part == 0) {
        if (r == c) dinv[c] = 1.0 / T[r][r];
        else if (r > c) T[c][r] = -s / T[r][r];
      }
      __syncthreads();
    }
  }
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx / NB, c = idx % NB;
    if (c <= r) Akk[(size_t)r * ld + c] = T[r][c];
    invd[(size_t)k * NB * NB + idx] = (r < c) ? 0.0 : (r == c ? dinv[r] : T[c][r]);
  }
  if (tid == 0 && bad) atomicExch(status, 1);
}

// ---------------------------------------------------------------------------------
// C(64x64) = X(64xNB) * Y(64xNB)^T on the FP64 matrix cores; X, Y in LDS (stride LDT).
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

// Panel: for every row block i > k (the last one is the single rhs row):
// A_ik <- A_ik * inv(L_kk)^T.
__global__ void __launch_bounds__(256)
k_panel(double* __restrict__ A, uint32_t ld, uint32_t k, uint32_t nblk /* matrix row blocks */,
        const double* __restrict__ invd) {
  __shared__ double X[NB][LDT];
  __shared__ double Y[NB][LDT];
  const uint32_t i = k + 1 + blockIdx.x;       // row block; i == nblk is the rhs row
  const int rows = (i == nblk) ? 1 : NB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double* Aik = A + ((size_t)i * NB) * ld + (size_t)k * NB;
  load_tile(X, Aik, ld, rows, tid);
  load_tile(Y, invd + (size_t)k * NB * NB, NB, NB, tid);
  __syncthreads();
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
  tile_mma(X, Y, wave, lane, acc);
  const int rb = 32 * (wave >> 1), cb = 32 * (wave & 1);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = rb + 16 * ti + (lane >> 4) + 4 * reg;
        const int c = cb + 16 * tj + (lane & 15);
        if (r < rows) Aik[(size_t)r * ld + c] = acc[ti][tj][reg];
      }
}

// Trailing update: for k < j <= i: A_ij -= A_ik A_jk^T (lower tiles; i may be the rhs row,
// for which j < nblk).  Linear block index -> (i,j) over the trailing triangle.
__global__ void __launch_bounds__(256)
k_syrk(double* __restrict__ A, uint32_t ld, uint32_t k, uint32_t nblk) {
  __shared__ double X[NB][LDT];
  __shared__ double Y[NB][LDT];
  const uint32_t m = nblk - k - 1;  // trailing matrix row blocks
  // tiles: triangle of m rows (t*(t+1)/2 indexing) followed by the rhs row (m tiles)
  const uint32_t tri = m * (m + 1) / 2;
  uint32_t bi, bj;
  const uint32_t b = blockIdx.x;
  if (b < tri) {
    // bi = largest t with t(t+1)/2 <= b
    uint32_t t = (uint32_t)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((t + 1) * (t + 2) / 2 <= b) ++t;
    while (t * (t + 1) / 2 > b) --t;
    bi = t;
    bj = b - t * (t + 1) / 2;
  } else {
    bi = m;  // rhs row
    bj = b - tri;
  }
  const uint32_t i = k + 1 + bi, j = k + 1 + bj;
  const int rows = (i == nblk) ? 1 : NB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const double* Aik = A + ((size_t)i * NB) * ld + (size_t)k * NB;
  const double* Ajk = A + ((size_t)j * NB) * ld + (size_t)k * NB;
  double* Aij = A + ((size_t)i * NB) * ld + (size_t)j * NB;
  load_tile(X, Aik, ld, rows, tid);
  load_tile(Y, Ajk, ld, NB, tid);
  __syncthreads();
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) acc[a][bb] = (double4_t){0.0, 0.0, 0.0, 0.0};
  tile_mma(X, Y, wave, lane, acc);
  const int rb = 32 * (wave >> 1), cb = 32 * (wave & 1);
  const bool diag = (i == j);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = rb + 16 * ti + (lane >> 4) + 4 * reg;
        const int c = cb + 16 * tj + (lane & 15);
        if (r < rows && (!diag || c <= r)) Aij[(size_t)r * ld + c] -= acc[ti][tj][reg];
      }
}

// Backward substitution, block row i (from the last to the first):
//   x_i = inv(L_ii)^T y_i ;  y[0 : i*NB] -= L[i-block, 0:i*NB]^T x_i
// y lives in the rhs row of A.  Every workgroup recomputes x_i (64x64 mat-vec) and
// updates its own 256 columns; workgroup 0 also stores x_i.
__global__ void __launch_bounds__(256)
k_backward(double* __restrict__ A, uint32_t ld, uint32_t i, uint32_t nblk,
           const double* __restrict__ invd, double* __restrict__ x) {
  __shared__ double xi[NB];
  __shared__ double part[4][NB];
  const int tid = threadIdx.x;
  const double* y = A + ((size_t)nblk * NB) * ld;
  {
    // x_i[c] = sum_r Linv[r][c] * y_i[r]; 4 partial sums per column
    const int c = tid & 63, q = tid >> 6;
    const double* Li = invd + (size_t)i * NB * NB;
    double s = 0.0;
    for (int r = q * 16; r < q * 16 + 16; ++r) s += Li[r * NB + c] * y[(size_t)i * NB + r];
    part[q][c] = s;
  }
  __syncthreads();
  if (tid < NB) {
    const double v = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    xi[tid] = v;
    if (blockIdx.x == 0) x[(size_t)i * NB + tid] = v;
  }
  __syncthreads();
  const uint32_t col = blockIdx.x * 256 + tid;
  if (col < i * NB) {
    const double* Li = A + ((size_t)i * NB) * ld + col;
    double s = 0.0;
#pragma unroll 8
    for (int r = 0; r < NB; ++r) s += Li[(size_t)r * ld] * xi[r];
    double* yy = A + ((size_t)nblk * NB) * ld + col;
    *yy -= s;
  }
}

// Solve on the padded lower storage dA ((n_pad + 1) x ld, n_pad = ld multiple of 64; the
// rhs is row n_pad).  dx receives n_pad doubles (the first n are the solution).
int cholesky_solve(Engine* e, double* dA, uint32_t n, uint32_t ld, double* dx, int* status) {
  (void)n;
  const uint32_t nblk = ld / NB;
  BAE_HIP(e->invdiag.alloc((size_t)nblk * NB * NB));
  BAE_HIP(hipMemsetAsync(e->flags.p, 0, sizeof(int), e->stream));
  for (uint32_t k = 0; k < nblk; ++k) {
    hipLaunchKernelGGL(k_potrf_inv, dim3(1), dim3(256), 0, e->stream, dA, ld, k, e->invdiag.p,
                       e->flags.p);
    const uint32_t below = nblk - k;  // row blocks k+1..nblk (incl. the rhs row)
    hipLaunchKernelGGL(k_panel, dim3(below), dim3(256), 0, e->stream, dA, ld, k, nblk,
                       e->invdiag.p);
    const uint32_t m = nblk - k - 1;
    const uint32_t tiles = m * (m + 1) / 2 + m;
    if (tiles > 0)
      hipLaunchKernelGGL(k_syrk, dim3(tiles), dim3(256), 0, e->stream, dA, ld, k, nblk);
  }
  BAE_HIP(hipGetLastError());
  for (uint32_t ii = nblk; ii-- > 0;) {
    const uint32_t cols = ii * NB;
    const uint32_t grid = cols == 0 ? 1 : (cols + 255) / 256;
    hipLaunchKernelGGL(k_backward, dim3(grid), dim3(256), 0, e->stream, dA, ld, ii, nblk,
                       e->invdiag.p, dx);
  }
  BAE_HIP(hipGetLastError());
  int st = 0;
  BAE_HIP(hipMemcpyAsync(&st, e->flags.p, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  BAE_HIP(hipStreamSynchronize(e->stream));
  *status = st;
  return 0;
}

}  // namespace bae
