// gfx950 kernels that assemble the reduced camera system and the scalar reductions:
//   * k_gather_S / k_gather_rhs — deterministic gathers that replace the reference's
//     block-sparse products jt_pr*j_pr, jt_pr*j_l*vi*jt_l*j_pr and the dense write of
//     S = U - W V^-1 W^T (BundleAdjuster.cpp:337-354, 448-485; SparseBlockMatrixOps.h:
//     182-254, 318-364).  Every 6x6 block of S is the sum of rank-1 products of "factor
//     rows" emitted by k_landmarks; the list of products per pose pair is static across
//     Gauss-Newton iterations and is built once per Solve() on the host.  No atomics: S
//     and rhs are bitwise reproducible.
//   * exact k-th element selection for the Huber sigma (std::nth_element,
//     BundleAdjuster.cpp:1356-1358)
//   * step composition, norms and the dogleg scalars (BundleAdjuster.cpp:858-1017)
#include "engine.h"
#include <cstring>

namespace bae {

// ---------------------------------------------------------------------------------
// Off-diagonal blocks of S from the static gather lists.  Lane (r, c0): r = row of the 6x6 block,
// c0 = 0 or 3; the factor rows are 48-byte gathers served by L2 / Infinity Cache (rows are laid
// out pose-major, so one pair touches two contiguous row ranges).
// Output: lower storage of the symmetric S, row-major with leading dimension ld: block (i,j),
// i<j, is written transposed at rows j*D.., cols i*D..  Masked parameters: S(idx,idx) = 1e6
// (BundleAdjuster.cpp:587-598) is written by k_gather_S_diag — their rows and columns are
// already zero because k_landmarks zeroed the Jacobian columns.
// Five pose pairs per wavefront, one per 12-lane slot: a lane owns 3 elements (r, c0..c0+2) of
// its pair's 6x6 block and walks that pair's entry list alone — no partial blocks, no LDS, no
// barrier.  Short lists (configs[3]: 2.3 entries per pair on average) no longer leave four of the
// five slots idle, and for long lists the five pairs of a wave (neighbours in the sorted order:
// similar co-visibility, similar length) keep all slots busy just as the five-way split of one
// list did.  The terms of a block are added in entry order.
__global__ void __launch_bounds__(256)
k_gather_S(uint32_t npairs, const uint32_t* __restrict__ pair_ptr,
            const uint2* __restrict__ pair_ij, const uint2* __restrict__ pair_ent,
            const double* __restrict__ frow, int D, uint32_t ld, double* __restrict__ A) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane >= 60) return;
  const int slot = lane / 12, t = lane - 12 * slot;
  const uint32_t pair = (blockIdx.x * 4 + w) * 5 + slot;
  if (pair >= npairs) return;
  const int r = t >> 1, c0 = (t & 1) * 3;
  const uint32_t e0 = pair_ptr[pair], e1 = pair_ptr[pair + 1];
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  uint32_t e = e0;
  for (; e + 1 < e1; e += 2) {
    const uint2 p0 = pair_ent[e], p1 = pair_ent[e + 1];
    const double a0 = frow[(size_t)p0.x * kRow + r], a1 = frow[(size_t)p1.x * kRow + r];
    const double* rb0 = frow + (size_t)p0.y * kRow + c0;
    const double* rb1 = frow + (size_t)p1.y * kRow + c0;
    const double b00 = rb0[0], b01 = rb0[1], b02 = rb0[2], b10 = rb1[0], b11 = rb1[1], b12 = rb1[2];
    acc0 += a0 * b00; acc1 += a0 * b01; acc2 += a0 * b02;
    acc0 += a1 * b10; acc1 += a1 * b11; acc2 += a1 * b12;
  }
  if (e < e1) {
    const uint2 p0 = pair_ent[e];
    const double a0 = frow[(size_t)p0.x * kRow + r];
    const double* rb0 = frow + (size_t)p0.y * kRow + c0;
    acc0 += a0 * rb0[0]; acc1 += a0 * rb0[1]; acc2 += a0 * rb0[2];
  }
  const uint2 ij = pair_ij[pair];
  const uint32_t i = ij.x, j = ij.y;
  if (i == j) {
    // cross terms of observations whose two sides sit on the same pose; the block itself was
    // written by k_gather_S_diag (stream order), single writer
    double* o = A + ((size_t)i * D + r) * ld + (size_t)i * D + c0;
    o[0] += acc0; o[1] += acc1; o[2] += acc2;
  } else {
    double* o = A + ((size_t)j * D + c0) * ld + (size_t)i * D + r;
    o[0] = acc0; o[ld] = acc1; o[2 * (size_t)ld] = acc2;
  }
}

// ---------------------------------------------------------------------------------
// Diagonal blocks (i,i): one workgroup per active pose streams the pose's own factor
// rows — contiguous in the pose-major numbering — instead of walking an entry list:
//   sum over its J rows of row row^T                       (jt_pr j_pr, diagonal part)
//   sum over its incidences of (-W V^-1)_k W_k^T           (the Schur term of the pose with itself)
// A pose of the 1M-residual benchmark owns ~2000 + 2*1000 rows; as list entries of ONE
// wavefront they were the critical path of the whole gather.  Fixed reduction order
// (strided rows per thread, xor-butterfly per wave, waves in order): bitwise reproducible.
// Also writes the fixed entries of the block: 1e6 on masked parameters (shard 0 only).
template <int LM>
__global__ void __launch_bounds__(256)
k_gather_S_diag(const uint32_t* __restrict__ pose_rows, uint32_t npose, uint32_t jbase,
                const double* __restrict__ frow, int D, uint32_t ld,
                const uint16_t* __restrict__ mask_opt, int write_fixed, double* __restrict__ A) {
  __shared__ double red[4][36];
  const uint32_t i = blockIdx.x;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
  {
    const uint32_t r0 = jbase + 2 * pose_rows[i], r1 = jbase + 2 * pose_rows[i + 1];
    for (uint32_t row = r0 + tid; row < r1; row += 256) {
      const double* v = frow + (size_t)row * kRow;
      const double v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3], v4 = v[4], v5 = v[5];
      const double vv[6] = {v0, v1, v2, v3, v4, v5};
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) acc[a * 6 + b] += vv[a] * vv[b];
    }
  }
  if (LM > 0) {
    const uint32_t* pinc = pose_rows + (npose + 1);
    const uint32_t q0 = pinc[i], q1 = pinc[i + 1];
    const int lm = LM > 0 ? LM : 1;
    for (uint32_t q = q0 + tid; q < q1; q += 256) {
      const double* base = frow + (size_t)q * 2 * lm * kRow;
#pragma unroll
      for (int k = 0; k < lm; ++k) {
        const double* a = base + (size_t)(lm + k) * kRow;  // (-W V^-1) row k
        const double* b = base + (size_t)k * kRow;         // W row k
        const double av[6] = {a[0], a[1], a[2], a[3], a[4], a[5]};
        const double bv[6] = {b[0], b[1], b[2], b[3], b[4], b[5]};
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
          for (int y = 0; y < 6; ++y) acc[x * 6 + y] += av[x] * bv[y];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 36; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[w][k] = v;
  }
  __syncthreads();
  if (tid < 36) {
    const int rr = tid / 6, cc = tid - 6 * rr;
    double v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    const uint16_t m = mask_opt[i];
    if (rr == cc && (m & (1u << rr))) v = write_fixed ? 1e6 : 0.0;
    A[((size_t)i * D + rr) * ld + (size_t)i * D + cc] = v;
  } else if ((int)tid - 36 < D - 6) {
    const uint16_t m = mask_opt[i];
    const int k = 6 + (tid - 36);
    if ((m & (1u << k)) && write_fixed) A[((size_t)i * D + k) * ld + (size_t)i * D + k] = 1e6;
  }
}

// rows n..n_pad-1 of the padded system: identity
__global__ void k_pad_diag(uint32_t n, uint32_t n_pad, uint32_t ld, double* __restrict__ A) {
  const uint32_t k = n + blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_pad) A[(size_t)k * ld + k] = 1.0;
}

// One workgroup per active pose: rhs_p[i] = sum F[row] * scal[idx] over the pose's
// observation rows (jt_pr * r_pr, BundleAdjuster.cpp:348-353), then minus W V^-1 rhs_l
// over its incidences (:480-484).  The factor rows of a pose are contiguous (pose-major),
// so consecutive threads read consecutive 48-byte rows; the scalars are 8-byte gathers.
// Fixed reduction order (strided entries per thread, xor-butterfly, waves in order).
__global__ void __launch_bounds__(256)
k_gather_rhs(uint32_t npose, const uint32_t* __restrict__ prhs_ptr,
             const uint32_t* __restrict__ prhs_mid, const uint2* __restrict__ prhs_ent,
             const double* __restrict__ frow, const double* __restrict__ scal, int D,
             double* __restrict__ rhs_p, double* __restrict__ rhs_sc) {
  __shared__ double red[4][12];
  const uint32_t i = blockIdx.x;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const uint32_t e0 = prhs_ptr[i], em = prhs_mid[i], e1 = prhs_ptr[i + 1];
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.0;
  for (uint32_t e = e0 + tid; e < em; e += 256) {
    const uint2 p = prhs_ent[e];
    const double* v = frow + (size_t)p.x * kRow;
    const double sc = scal[p.y];
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[c] += v[c] * sc;
  }
  for (uint32_t e = em + tid; e < e1; e += 256) {
    const uint2 p = prhs_ent[e];
    const double* v = frow + (size_t)p.x * kRow;
    const double sc = scal[p.y];
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[6 + c] += v[c] * sc;
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[w][k] = v;
  }
  __syncthreads();
  if (tid < 6) {
    const double sa = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    const double sb = (red[0][6 + tid] + red[1][6 + tid]) + (red[2][6 + tid] + red[3][6 + tid]);
    rhs_p[(size_t)i * D + tid] = sa;
    rhs_sc[(size_t)i * D + tid] = sa + sb;
  }
  (void)npose;
}

// Lower storage of S at 64-tile granularity (row r: columns 0 .. 64 (r/64 + 1) - 1, i.e. the
// diagonal tiles in full — the 6x6 diagonal blocks are stored with both triangles) + the rhs
// row  <->  packed array: halves the all-reduce message of the landmark-sharded path.
// Row r of tile row t = r / 64 sits at 64^2 t (t+1) / 2 + (r % 64) 64 (t+1); the rhs row
// (r == n_pad) behind the matrix.  grid = (column chunks of 1024, n_pad + 1 rows).
__global__ void __launch_bounds__(256)
k_pack_lower(double* __restrict__ A, uint32_t ld, uint32_t n_pad, double* __restrict__ packed, int unpack) {
  const uint32_t r = blockIdx.y;
  const uint32_t t = r >> 6;
  const uint32_t len = (r == n_pad) ? n_pad : 64u * (t + 1);
  const uint32_t c0 = blockIdx.x * 1024u;
  if (c0 >= len) return;
  double* row = A + (size_t)r * ld;
  double* prow = packed + (size_t)4096 * ((size_t)t * (t + 1) / 2) + (size_t)(r & 63u) * 64u * (t + 1);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t c = c0 + k * 256u + threadIdx.x;
    if (c < len) {
      if (unpack) row[c] = prow[c];
      else prow[c] = row[c];
    }
  }
}

size_t packed_lower_count(uint32_t n_pad) {
  const size_t nblk = n_pad / 64;
  return (size_t)4096 * (nblk * (nblk + 1) / 2) + n_pad;
}

int launch_pack_lower(Engine* e, int unpack) {
  const uint32_t n_pad = e->st.ld;
  hipLaunchKernelGGL(k_pack_lower, dim3((n_pad + 1023) / 1024, n_pad + 1), dim3(256), 0, e->stream, e->A.p,
                     e->st.ld, n_pad, e->packed.p, unpack);
  BAE_HIP(hipGetLastError());
  return 0;
}

// Zeroes the 64x64 tiles of the lower triangle (tile row >= tile column) of the row-major
// storage: half the bytes of clearing the square.  Nothing reads or writes the strictly upper
// tiles except the corner of a diagonal D x D block that straddles a tile boundary, and that
// corner is rewritten in full by k_gather_S_diag every iteration.
__global__ void __launch_bounds__(256)
k_zero_lower_tiles(double* __restrict__ A, uint32_t ld, uint32_t nblk) {
  const uint32_t t = blockIdx.x;
  uint32_t i = (uint32_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((uint64_t)(i + 1) * (i + 2) / 2 <= t) ++i;
  while ((uint64_t)i * (i + 1) / 2 > t) --i;
  const uint32_t c = t - (uint32_t)((uint64_t)i * (i + 1) / 2);
  if (i >= nblk) return;
  double2* base = reinterpret_cast<double2*>(A + ((size_t)i * 64) * ld + (size_t)c * 64);
  const int tid = threadIdx.x, col2 = tid & 31, r0 = tid >> 5;  // 32 double2 per row, 8 rows per pass
#pragma unroll
  for (int r = r0; r < 64; r += 8) base[(size_t)r * (ld / 2) + col2] = make_double2(0.0, 0.0);
}

int launch_gather_S(Engine* e) {
  const Structure& st = e->st;
  const uint32_t n = st.n, ld = st.ld, n_pad = ld;
  // zero the lower storage + rhs row (the whole square once per structure, so that the unused
  // upper tiles never hold garbage), then identity on the padding
  if (e->A_cleared != e->A.p) {
    BAE_HIP(hipMemsetAsync(e->A.p, 0, (size_t)(n_pad + 1) * ld * sizeof(double), e->stream));
    e->A_cleared = e->A.p;
  } else {
    const uint32_t nblk = n_pad / 64;
    hipLaunchKernelGGL(k_zero_lower_tiles, dim3(nblk * (nblk + 1) / 2), dim3(256), 0, e->stream, e->A.p, ld, nblk);
    BAE_HIP(hipGetLastError());
    BAE_HIP(hipMemsetAsync(e->A.p + (size_t)n_pad * ld, 0, (size_t)ld * sizeof(double), e->stream));
  }
  // fixed entries (padding identity, 1e6 on masked parameters) are written by shard 0
  // only, so that the cross-shard sum of S leaves them exact
  const int write_fixed = (e->rank == 0) ? 1 : 0;
  if (n_pad > n && write_fixed) {
    hipLaunchKernelGGL(k_pad_diag, dim3((n_pad - n + 255) / 256), dim3(256), 0, e->stream, n, n_pad,
                       ld, e->A.p);
    BAE_HIP(hipGetLastError());
  }
  BAE_HIP(hipMemsetAsync(e->rhs_p.p, 0, e->rhs_p.bytes(), e->stream));
  BAE_HIP(hipMemsetAsync(e->rhs_sc.p, 0, e->rhs_sc.bytes(), e->stream));
  if (st.Pact > 0) {
    const uint16_t* masks = e->pose_mask.p + st.P;  // masks by opt id live after the by-id masks
#define BAE_DIAG(LMV)                                                                              \
  hipLaunchKernelGGL(k_gather_S_diag<LMV>, dim3(st.Pact), dim3(256), 0, e->stream,                 \
                     (const uint32_t*)e->pose_rows.p, st.Pact, st.jbase, (const double*)e->frow.p, \
                     e->pose_dim, ld, masks, write_fixed, e->A.p)
    if (e->lm_dim == 0) BAE_DIAG(0); else if (e->lm_dim == 1) BAE_DIAG(1); else BAE_DIAG(3);
#undef BAE_DIAG
    BAE_HIP(hipGetLastError());
  }
  if (st.n_pairs > 0) {
    e->prof_begin(e->ev_gather);
    hipLaunchKernelGGL(k_gather_S, dim3((st.n_pairs + 19) / 20), dim3(256), 0, e->stream, st.n_pairs,
                       e->pair_ptr.p, e->pair_ij.p, e->pair_ent.p, e->frow.p, e->pose_dim, ld, e->A.p);
    e->prof_end(e->ev_gather);
    BAE_HIP(hipGetLastError());
  }
  if (st.Pact > 0 && st.O > 0) {
    hipLaunchKernelGGL(k_gather_rhs, dim3(st.Pact), dim3(256), 0, e->stream, st.Pact,
                       e->prhs_ptr.p, e->prhs_ptr.p + (st.Pact + 1), e->prhs_ent.p, e->frow.p,
                       e->scal.p, e->pose_dim, e->rhs_p.p, e->rhs_sc.p);
    BAE_HIP(hipGetLastError());
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// Test tap (ba_hip_check_solve): y = S x for the symmetric S kept in lower storage (the copy taken
// before the in-place factorisation), without downloading it — S is 28.8 GB at configs[3].
//   k_symv_rows: y[r]  = sum_{c <= r} A[r][c] x[c]      one workgroup per row, coalesced row read
//   k_symv_cols: y[c] += sum_{r >  c} A[r][c] x[r]      one workgroup per 64-column tile; a wave
//                reads 64 consecutive doubles of one row per step, lane = column; fixed order
__global__ void __launch_bounds__(256)
k_symv_rows(uint32_t n, uint32_t ld, const double* __restrict__ A, const double* __restrict__ x,
            double* __restrict__ y) {
  __shared__ double red[4];
  const uint32_t r = blockIdx.x;
  const double* row = A + (size_t)r * ld;
  double s = 0.0;
  for (uint32_t c = threadIdx.x; c <= r; c += 256) s += row[c] * x[c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) y[r] = (red[0] + red[1]) + (red[2] + red[3]);
  (void)n;
}
__global__ void __launch_bounds__(256)
k_symv_cols(uint32_t n, uint32_t ld, const double* __restrict__ A, const double* __restrict__ x,
            double* __restrict__ y) {
  __shared__ double red[4][64];
  const uint32_t c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  double s = 0.0;
  if (c < n)
    for (uint32_t r = blockIdx.x * 64 + w; r < n; r += 4)
      if (r > c) s += A[(size_t)r * ld + c] * x[r];
  red[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < n) y[c] += (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// partial sums of |y - b|^2 and |b|^2
__global__ void __launch_bounds__(256)
k_resid_norms(uint32_t n, const double* __restrict__ y, const double* __restrict__ b, double* __restrict__ partials,
              uint32_t nparts) {
  __shared__ double red[2][256];
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  double d = 0, bb = 0;
  if (i < n) { d = y[i] - b[i]; bb = b[i]; }
  red[0][threadIdx.x] = d * d; red[1][threadIdx.x] = bb * bb;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) { red[0][threadIdx.x] += red[0][threadIdx.x + k]; red[1][threadIdx.x] += red[1][threadIdx.x + k]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partials[blockIdx.x] = red[0][0]; partials[nparts + blockIdx.x] = red[1][0]; }
}

int sum_partials(Engine* e, uint32_t nparts, uint32_t ncomp, double* host_out, bool cross_shard);

// |S x - b| and |b| for the kept copy of S (lower storage, ld), x and b device vectors of n doubles
int check_solve_residual(Engine* e, const double* dS, const double* dx, const double* db, double* out2) {
  const uint32_t n = e->st.n, ld = e->st.ld;
  out2[0] = out2[1] = 0.0;
  if (n == 0) return 0;
  DBuf<double> y;
  BAE_HIP(y.alloc(n));
  hipLaunchKernelGGL(k_symv_rows, dim3(n), dim3(256), 0, e->stream, n, ld, dS, dx, y.p);
  hipLaunchKernelGGL(k_symv_cols, dim3((n + 63) / 64), dim3(256), 0, e->stream, n, ld, dS, dx, y.p);
  const uint32_t nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_resid_norms, dim3(nb), dim3(256), 0, e->stream, n, (const double*)y.p, db, e->partials.p, nb);
  hipError_t err = hipGetLastError();
  int rc = err == hipSuccess ? sum_partials(e, nb, 2, out2, false) : e->fail(err, "check_solve kernels");
  y.release();
  if (rc) return rc;
  out2[0] = sqrt(out2[0]); out2[1] = sqrt(out2[1]);
  return 0;
}

// ---------------------------------------------------------------------------------
// Fixed-order sum of `ncomp` rows of `nparts` partial sums (one block).
__global__ void k_sum_partials(uint32_t nparts, uint32_t ncomp, const double* __restrict__ parts,
                               double* __restrict__ out) {
  __shared__ double red[256];
  for (uint32_t c = 0; c < ncomp; ++c) {
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < nparts; i += 256) s += parts[(size_t)c * nparts + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
      if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = red[0];
    __syncthreads();
  }
}

int sum_partials(Engine* e, uint32_t nparts, uint32_t ncomp, double* host_out, bool cross_shard) {
  if (nparts == 0) {
    for (uint32_t c = 0; c < ncomp; ++c) host_out[c] = 0.0;
  } else {
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, e->stream, nparts, ncomp,
                       e->partials.p, e->scalars_out.p);
    BAE_HIP(hipGetLastError());
    BAE_HIP(hipMemcpyAsync(host_out, e->scalars_out.p, ncomp * sizeof(double),
                           hipMemcpyDeviceToHost, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
  }
  if (cross_shard && e->allreduce && e->nranks > 1) {
    // cross-shard sum of the scalars
    BAE_HIP(hipMemcpyAsync(e->scalars_out.p, host_out, ncomp * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
    if (e->allreduce(e->allreduce_ctx, e->scalars_out.p, ncomp, 0) != 0)
      return e->fail_msg("allreduce hook failed");
    BAE_HIP(hipMemcpy(host_out, e->scalars_out.p, ncomp * sizeof(double), hipMemcpyDeviceToHost));
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// Exact selection of the k-th smallest of non-negative doubles by most-significant-
// digit radix passes over their bit patterns (monotone for x >= 0): six passes of 11
// bits; per pass an LDS-privatised 2048-bin histogram of the values whose higher bits
// equal the prefix found so far.  The histogram is the only thing that crosses shards
// (summed by the all-reduce hook), so the multi-GPU median is exact too.
__global__ void __launch_bounds__(256)
k_select_hist(uint32_t n, const double* __restrict__ v, unsigned long long prefix,
              unsigned long long prefix_mask, int shift, unsigned int digit_mask,
              unsigned long long* __restrict__ hist) {
  __shared__ unsigned int h[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v[i]);
    if ((b & prefix_mask) == prefix) atomicAdd(&h[(b >> shift) & digit_mask], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256)
    if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

// Single-shard variant: the whole selection stays on the device.  The running state
// {prefix, mask, remaining rank, error} lives in sel[0..3]; k_select_hist_dev reads it,
// k_select_scan (one wavefront) walks the 2048 bins with a wave-wide prefix sum and advances
// it.  Six passes are enqueued back to back; one 8-byte copy returns the result.
__global__ void __launch_bounds__(256)
k_select_hist_dev(uint32_t n, const double* __restrict__ v, const unsigned long long* __restrict__ sel,
                  int shift, unsigned int digit_mask, unsigned long long* __restrict__ hist) {
  __shared__ unsigned int h[2048];
  const unsigned long long prefix = sel[0], prefix_mask = sel[1];
  for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v[i]);
    if ((b & prefix_mask) == prefix) atomicAdd(&h[(b >> shift) & digit_mask], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256)
    if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

__global__ void __launch_bounds__(64)
k_select_scan(unsigned long long* __restrict__ sel, unsigned long long* __restrict__ hist, int shift,
              int nbins, unsigned long long digit_mask) {
  const int lane = threadIdx.x;
  unsigned long long k = sel[2];
  // lane owns bins [32 lane, 32 lane + 32): local sum, exclusive wave scan, then a local walk
  unsigned long long loc = 0;
  for (int b = 0; b < 32; ++b) loc += hist[32 * lane + b];
  unsigned long long incl = loc;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  const unsigned long long excl = incl - loc;
  const bool mine = (k >= excl) && (k < incl) && (32 * lane < nbins);
  if (mine) {
    unsigned long long r = k - excl;
    int b = 0;
    for (; b < 32; ++b) {
      const unsigned long long hb = hist[32 * lane + b];
      if (r < hb) break;
      r -= hb;
    }
    sel[0] |= ((unsigned long long)(32 * lane + b)) << shift;
    sel[1] |= digit_mask << shift;
    sel[2] = r;
  }
  const unsigned long long found = __ballot(mine);
  if (lane == 0 && found == 0) sel[3] = 1;  // rank out of range
  __syncthreads();
  for (int b = 0; b < 32; ++b) hist[32 * lane + b] = 0;  // ready for the next pass
}

static int select_kth_device(Engine* e, const double* d_values, uint32_t n_local, uint64_t k, double* out) {
  static const int shifts[6] = {53, 42, 31, 20, 9, 0};
  unsigned long long init[4] = {0ull, 0ull, (unsigned long long)k, 0ull};
  unsigned long long* sel = e->hist.p + 2048;
  BAE_HIP(hipMemsetAsync(e->hist.p, 0, 2048 * sizeof(unsigned long long), e->stream));
  BAE_HIP(hipMemcpyAsync(sel, init, sizeof(init), hipMemcpyHostToDevice, e->stream));
  uint32_t grid = (n_local + 255) / 256;
  if (grid > 2048) grid = 2048;
  for (int pass = 0; pass < 6; ++pass) {
    hipLaunchKernelGGL(k_select_hist_dev, dim3(grid), dim3(256), 0, e->stream, n_local, d_values,
                       (const unsigned long long*)sel, shifts[pass], pass == 5 ? 511u : 2047u, e->hist.p);
    hipLaunchKernelGGL(k_select_scan, dim3(1), dim3(64), 0, e->stream, sel, e->hist.p, shifts[pass],
                       pass == 5 ? 512 : 2048, pass == 5 ? 511ull : 2047ull);
  }
  BAE_HIP(hipGetLastError());
  unsigned long long res[4];
  BAE_HIP(hipMemcpyAsync(res, sel, sizeof(res), hipMemcpyDeviceToHost, e->stream));
  BAE_HIP(hipStreamSynchronize(e->stream));
  if (res[3]) return e->fail_msg("select_kth: rank out of range");
  double r;
  memcpy(&r, &res[0], sizeof(double));
  *out = r;
  return 0;
}

int select_kth(Engine* e, const double* d_values, uint32_t n_local, uint64_t k, double* out) {
  if (!(e->allreduce && e->nranks > 1) && n_local > 0) return select_kth_device(e, d_values, n_local, k, out);
  // digits from the top: bits [63:53] [52:42] [41:31] [30:20] [19:9] [8:0](9 bits, shift 0)
  static const int shifts[6] = {53, 42, 31, 20, 9, 0};
  unsigned long long prefix = 0, mask = 0;
  std::vector<unsigned long long> hh(2048);
  for (int pass = 0; pass < 6; ++pass) {
    const int shift = shifts[pass];
    BAE_HIP(hipMemsetAsync(e->hist.p, 0, 2048 * sizeof(unsigned long long), e->stream));
    if (n_local > 0) {
      uint32_t grid = (n_local + 255) / 256;
      if (grid > 2048) grid = 2048;
      hipLaunchKernelGGL(k_select_hist, dim3(grid), dim3(256), 0, e->stream, n_local, d_values,
                         prefix, mask, shift, pass == 5 ? 511u : 2047u, e->hist.p);
      BAE_HIP(hipGetLastError());
    }
    BAE_HIP(hipStreamSynchronize(e->stream));
    if (e->allreduce && e->nranks > 1)
      if (e->allreduce(e->allreduce_ctx, e->hist.p, 2048, 1) != 0)
        return e->fail_msg("allreduce hook failed");
    BAE_HIP(hipMemcpy(hh.data(), e->hist.p, 2048 * sizeof(unsigned long long),
                      hipMemcpyDeviceToHost));
    const int nbins = pass == 5 ? 512 : 2048;
    int bin = 0;
    for (; bin < nbins; ++bin) {
      if (k < hh[bin]) break;
      k -= hh[bin];
    }
    if (bin >= nbins) return e->fail_msg("select_kth: rank out of range");
    const unsigned long long digit_mask = (pass == 5 ? 511ull : 2047ull) << shift;
    prefix |= ((unsigned long long)bin << shift);
    mask |= digit_mask;
  }
  double r;
  memcpy(&r, &prefix, sizeof(double));
  *out = r;
  return 0;
}

// ---------------------------------------------------------------------------------
// step = coef_rhs * rhs + coef_gn * gn for poses (by opt id) and landmarks (rhs_l is
// stored by landmark id, steps by opt id); partial squared norms per block.
__global__ void k_compose(uint32_t n, double a, double b, const double* __restrict__ rhs,
                          const double* __restrict__ gn, double* __restrict__ step,
                          double* __restrict__ partials) {
  __shared__ double red[256];
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  double v = 0.0;
  if (i < n) {
    v = a * rhs[i] + (b != 0.0 ? b * gn[i] : 0.0);
    step[i] = v;
  }
  red[threadIdx.x] = v * v;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}
__global__ void k_compose_lm(uint32_t L, int LM, double a, double b,
                             const int32_t* __restrict__ lm_opt, const double* __restrict__ bl,
                             const double* __restrict__ gn, double* __restrict__ step,
                             double* __restrict__ partials) {
  __shared__ double red[256];
  const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
  double sq = 0.0;
  if (l < L) {
    const int lo = lm_opt[l];
    if (lo >= 0) {
      for (int k = 0; k < LM; ++k) {
        const double v = a * bl[l * LM + k] + (b != 0.0 ? b * gn[(size_t)lo * LM + k] : 0.0);
        step[(size_t)lo * LM + k] = v;
        sq += v * v;
      }
    }
  }
  red[threadIdx.x] = sq;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

int launch_compose_step(Engine* e, double coef_rhs, double coef_gn, double* norms2_host) {
  const Structure& st = e->st;
  norms2_host[0] = norms2_host[1] = 0.0;
  if (st.n > 0) {
    const uint32_t nb = (st.n + 255) / 256;
    hipLaunchKernelGGL(k_compose, dim3(nb), dim3(256), 0, e->stream, st.n, coef_rhs, coef_gn,
                       e->rhs_p.p, e->gn_p.p, e->step_p.p, e->partials.p);
    BAE_HIP(hipGetLastError());
    // the pose step is replicated on every shard: no cross-shard sum
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, e->stream, nb, 1u, e->partials.p,
                       e->scalars_out.p);
    BAE_HIP(hipGetLastError());
    BAE_HIP(hipMemcpyAsync(&norms2_host[0], e->scalars_out.p, sizeof(double),
                           hipMemcpyDeviceToHost, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
  }
  if (st.L > 0 && e->lm_dim > 0 && st.Lact > 0) {
    const uint32_t nb = (st.L + 255) / 256;
    hipLaunchKernelGGL(k_compose_lm, dim3(nb), dim3(256), 0, e->stream, st.L, e->lm_dim, coef_rhs,
                       coef_gn, e->lm_opt.p, e->lm_bl.p, e->gn_l.p, e->step_l.p, e->partials.p);
    BAE_HIP(hipGetLastError());
    int rc = sum_partials(e, nb, 1, &norms2_host[1]);  // landmarks are sharded: summed
    if (rc) return rc;
  } else if (e->allreduce && e->nranks > 1) {
    int rc = sum_partials(e, 0, 1, &norms2_host[1]);
    if (rc) return rc;
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// Dogleg scalars.  Pose-part dots (replicated across shards) and landmark-part dots
// (sharded), plus ||J_pr rhs_p + J_l rhs_l||^2 over the observations
// (BundleAdjuster.cpp:881-910).
__global__ void k_dots_pose(uint32_t n, int gn_ok, const double* __restrict__ rhs,
                            const double* __restrict__ gn, double* __restrict__ partials,
                            uint32_t nparts) {
  __shared__ double red[3][256];
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  double r = 0, g = 0;
  if (i < n) { r = rhs[i]; g = gn_ok ? gn[i] : 0.0; }
  red[0][threadIdx.x] = r * r; red[1][threadIdx.x] = g * g; red[2][threadIdx.x] = r * g;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k)
      for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int c = 0; c < 3; ++c) partials[(size_t)c * nparts + blockIdx.x] = red[c][0];
}
__global__ void k_dots_lm(uint32_t L, int LM, int gn_ok, const int32_t* __restrict__ lm_opt,
                          const double* __restrict__ bl, const double* __restrict__ gn,
                          double* __restrict__ partials, uint32_t nparts) {
  __shared__ double red[3][256];
  const size_t l = (size_t)blockIdx.x * 256 + threadIdx.x;
  double rr = 0, gg = 0, rg = 0;
  if (l < L) {
    const int lo = lm_opt[l];
    if (lo >= 0)
      for (int k = 0; k < LM; ++k) {
        const double r = bl[l * LM + k], g = gn_ok ? gn[(size_t)lo * LM + k] : 0.0;
        rr += r * r; gg += g * g; rg += r * g;
      }
  }
  red[0][threadIdx.x] = rr; red[1][threadIdx.x] = gg; red[2][threadIdx.x] = rg;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k)
      for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int c = 0; c < 3; ++c) partials[(size_t)c * nparts + blockIdx.x] = red[c][0];
}
// per observation: v = sqrt(w) (Jm rhs_p[m] + Jr rhs_p[r] + Jl rhs_l[l]); sum |v|^2
__global__ void k_jrhs(uint32_t O, int D, int LM, const int32_t* __restrict__ obs_jrow_m,
                       const int32_t* __restrict__ obs_jrow_r, const uint32_t* __restrict__ obs_pose,
                       const uint32_t* __restrict__ obs_lm, const uint32_t* __restrict__ lm_ref_pose,
                       const int32_t* __restrict__ pose_opt, const int32_t* __restrict__ lm_opt,
                       const double* __restrict__ frow, const double* __restrict__ obs_jl,
                       const double* __restrict__ rhs_p, const double* __restrict__ bl,
                       double* __restrict__ partials) {
  __shared__ double red[256];
  const size_t a = (size_t)blockIdx.x * 256 + threadIdx.x;
  double sq = 0.0;
  if (a < O) {
    double v0 = 0, v1 = 0;
    const int jm = obs_jrow_m[a], jr = obs_jrow_r[a];
    const uint32_t l = obs_lm[a];
    if (jm >= 0) {
      const double* g = rhs_p + (size_t)pose_opt[obs_pose[a]] * D;
      const double* r0 = frow + (size_t)jm * kRow;
      for (int c = 0; c < 6; ++c) { v0 += r0[c] * g[c]; v1 += r0[kRow + c] * g[c]; }
    }
    if (jr >= 0) {
      const double* g = rhs_p + (size_t)pose_opt[lm_ref_pose[l]] * D;
      const double* r0 = frow + (size_t)jr * kRow;
      for (int c = 0; c < 6; ++c) { v0 += r0[c] * g[c]; v1 += r0[kRow + c] * g[c]; }
    }
    if (lm_opt[l] >= 0)
      for (int k = 0; k < LM; ++k) {
        v0 += obs_jl[a * 2 * LM + k] * bl[(size_t)l * LM + k];
        v1 += obs_jl[a * 2 * LM + LM + k] * bl[(size_t)l * LM + k];
      }
    sq = v0 * v0 + v1 * v1;
  }
  red[threadIdx.x] = sq;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

int launch_dogleg(Engine* e, int gn_available, ba_hip_dogleg_scalars* out) {
  const Structure& st = e->st;
  memset(out, 0, sizeof(*out));
  double h[3];
  int rc;
  if (st.n > 0) {  // pose parts: replicated on every shard
    const uint32_t nb = (st.n + 255) / 256;
    hipLaunchKernelGGL(k_dots_pose, dim3(nb), dim3(256), 0, e->stream, st.n, gn_available,
                       e->rhs_p.p, e->gn_p.p, e->partials.p, nb);
    BAE_HIP(hipGetLastError());
    if ((rc = sum_partials(e, nb, 3, h, false))) return rc;
    out->rhs_p_sq = h[0]; out->gn_p_sq = h[1]; out->rhs_gn_p = h[2];
  }
  {  // landmark parts: sharded
    const uint32_t nb = (st.L > 0 && e->lm_dim > 0) ? (st.L + 255) / 256 : 0;
    if (nb) {
      hipLaunchKernelGGL(k_dots_lm, dim3(nb), dim3(256), 0, e->stream, st.L, e->lm_dim, gn_available,
                         e->lm_opt.p, e->lm_bl.p, e->gn_l.p, e->partials.p, nb);
      BAE_HIP(hipGetLastError());
    }
    if ((rc = sum_partials(e, nb, 3, h, true))) return rc;
    out->rhs_l_sq = h[0]; out->gn_l_sq = h[1]; out->rhs_gn_l = h[2];
  }
  {  // || J_pr rhs_p + J_l rhs_l ||^2 over the observations (BundleAdjuster.cpp:881-906)
    const uint32_t nb = st.O > 0 ? (st.O + 255) / 256 : 0;
    if (nb) {
      hipLaunchKernelGGL(k_jrhs, dim3(nb), dim3(256), 0, e->stream, st.O, e->pose_dim, e->lm_dim,
                         e->obs_jrow_m.p, e->obs_jrow_r.p, e->obs_pose.p, e->obs_lm.p,
                         e->lm_ref_pose.p, e->pose_opt.p, e->lm_opt.p, e->frow.p, e->obs_jl.p,
                         e->rhs_p.p, e->lm_bl.p, e->partials.p);
      BAE_HIP(hipGetLastError());
    }
    if ((rc = sum_partials(e, nb, 1, h, true))) return rc;
    out->j_rhs_sq = h[0];
  }
  return 0;
}

}  // namespace bae
