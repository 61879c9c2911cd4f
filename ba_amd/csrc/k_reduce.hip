// gfx950 kernels that assemble the reduced camera system and the scalar reductions:
//   * k_assemble_tiles / k_pose_blocks — deterministic gathers that replace the reference's
//     block-sparse products jt_pr*j_pr, jt_pr*j_l*vi*jt_l*j_pr and the dense write of
//     S = U - W V^-1 W^T (BundleAdjuster.cpp:337-354, 448-485; SparseBlockMatrixOps.h:
//     182-254, 318-364).  Every 6x6 block of S is the sum of rank-1 products of "factor
//     rows" emitted by k_linearize; the list of products per block is static across
//     Gauss-Newton iterations and Solve() calls and is built once per graph (structure.h).
//     No atomics: S and rhs are bitwise reproducible.
//   * exact k-th element selection for the Huber sigma (std::nth_element,
//     BundleAdjuster.cpp:1356-1358)
//   * step composition, norms and the dogleg scalars (BundleAdjuster.cpp:858-1017)
#include "engine.h"
#include <cstring>

namespace bae {

// ---------------------------------------------------------------------------------
// Assembly of the reduced system from the factor rows of k_linearize (structure.h).
//
// k_assemble_tiles — one workgroup per 64x64 tile of the lower storage.  The tile is formed in LDS:
// zeroed, then every pose-pair block that overlaps it is summed by ONE thread from its list of
// rank-1 terms (rowA (x) rowB, two 48-byte row gathers per term; a block has 2.3 terms on average
// at configs[3]) and dropped into the tile, then the tile leaves as 64 full 512-byte rows.  Every
// byte of the lower triangle is written exactly once per iteration, coalesced: the separate
// zero-fill pass and the 8-byte column scatters of the first design are gone.  Blocks that straddle
// a tile boundary are listed by both tiles and clipped.  No atomics: bitwise reproducible.
// VAR 0: one thread per block (36 accumulators), terms one by one
// VAR 1: six threads per block (thread x forms row x of a (x) b), terms one by one
// VAR 2: six threads per block, terms four at a time — the four (rowA, rowB) index pairs are loaded
//        first, then the eight row pieces, so that a block's dependent-load chain is
//        2 x ceil(terms / 4) memory latencies instead of 2 x terms
// VAR 6 (round-3 experiment, ba_hip_debug_set key 1; VAR 5 is what runs): a test of the hypothesis that the kernel
//        is bound by the LATENCY of its dependent loads (a tile lives ~20 us in a workgroup, at most 4.7 tiles fit
//        a CU's LDS): half the threads (a tile has at most ~145 blocks), the launch entry carries the tile's list
//        range and coordinates (tile_desc: one load instead of tile_order -> tile_ptr), a block's terms go four at
//        a time with the index pairs of the NEXT four loaded beside the eight rows of the current four — four
//        dependent round trips instead of eight.  Bitwise the S of VAR 5, and NOT faster: 6.81 against 6.66 ms at
//        configs[3], 0.54 against 0.55 ms at configs[1] (profiles/r03_assemble_variants.log).  The kernel moves
//        ~33 GB (PMC) at ~5 TB/s: it is bound by the 64-byte sectors its 48-byte row gathers drag in, not by the
//        length of the dependency chain.
template <int VAR>
__global__ void __launch_bounds__(VAR == 6 ? 128 : 256)
k_assemble_tiles(uint32_t nt, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ tile_ptr,
                 const uint2* __restrict__ tile_ref, const uint2* __restrict__ pair_ent,
                 const double* __restrict__ frow, uint32_t ld, double* __restrict__ A,
                 const uint4* __restrict__ tile_desc) {
  constexpr int TS = 66;  // LDS row stride (doubles): even, so that rows can be read 16 bytes at a time
  constexpr int NT = VAR == 6 ? 128 : 256;
  __shared__ __attribute__((aligned(16))) double T[64 * TS];
  // launch order: see build_tile_order (engine.hip); padding entries are 0xffffffff
  uint32_t t, tr, tc, q0, q1;
  if constexpr (VAR == 6) {
    const uint4 d = tile_desc[blockIdx.x];
    t = d.x;
    if (t == 0xffffffffu) return;
    q0 = d.y; q1 = d.z; tr = d.w >> 16; tc = d.w & 0xffffu;
  } else {
    t = tile_order[blockIdx.x];
    if (t == 0xffffffffu) return;
    tr = (uint32_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((uint64_t)(tr + 1) * (tr + 2) / 2 <= t) ++tr;
    while ((uint64_t)tr * (tr + 1) / 2 > t) --tr;
    tc = t - (uint32_t)((uint64_t)tr * (tr + 1) / 2);
    q0 = tile_ptr[t]; q1 = tile_ptr[t + 1];
  }
  if (tr >= nt) return;
  const int tid = threadIdx.x;
  if (q1 > q0) {
    for (int i = tid; i < 64 * TS / 2; i += NT) reinterpret_cast<double2*>(T)[i] = make_double2(0.0, 0.0);
    __syncthreads();
    if constexpr (VAR == 3) {
      // experiment: no gather at all (floor of the write phase)
    } else if constexpr (VAR == 6) {
      for (uint32_t q = q0 + tid; q < q1; q += NT) {
        const uint2 ref = tile_ref[q];
        const uint32_t cnt = ref.y >> 14;
        const int ro = (int)((ref.y >> 7) & 127u) - kRefBias, co = (int)(ref.y & 127u) - kRefBias;
        double acc[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = 0.0;
        const uint32_t e1 = ref.x + cnt;
        uint2 en[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) en[u] = pair_ent[min(ref.x + (uint32_t)u, e1 - 1)];
        for (uint32_t e = ref.x; e < e1; e += 4) {
          double2 ra[4][3], rb[4][3];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double2* pa = reinterpret_cast<const double2*>(frow + (size_t)en[u].x * kRow);
            const double2* pb = reinterpret_cast<const double2*>(frow + (size_t)en[u].y * kRow);
            ra[u][0] = pa[0]; ra[u][1] = pa[1]; ra[u][2] = pa[2];
            rb[u][0] = pb[0]; rb[u][1] = pb[1]; rb[u][2] = pb[2];
          }
          if (e + 4 < e1) {  // the next four index pairs travel beside these rows
#pragma unroll
            for (int u = 0; u < 4; ++u) en[u] = pair_ent[min(e + 4 + (uint32_t)u, e1 - 1)];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (e + (uint32_t)u < e1) {
              const double a[6] = {ra[u][0].x, ra[u][0].y, ra[u][1].x, ra[u][1].y, ra[u][2].x, ra[u][2].y};
              const double b[6] = {rb[u][0].x, rb[u][0].y, rb[u][1].x, rb[u][1].y, rb[u][2].x, rb[u][2].y};
#pragma unroll
              for (int x = 0; x < 6; ++x)
#pragma unroll
                for (int y = 0; y < 6; ++y) acc[x * 6 + y] += a[x] * b[y];
            }
          }
        }
#pragma unroll
        for (int y = 0; y < 6; ++y) {
          const int rr = ro + y;
          if (rr < 0 || rr >= 64) continue;
#pragma unroll
          for (int x = 0; x < 6; ++x) {
            const int cc = co + x;
            if (cc >= 0 && cc < 64) T[rr * TS + cc] = acc[x * 6 + y];
          }
        }
      }
    } else if constexpr (VAR == 0 || VAR == 4 || VAR == 5) {
      for (uint32_t q = q0 + tid; q < q1; q += 256) {
        const uint2 ref = tile_ref[q];
        const uint32_t cnt = ref.y >> 14;
        const int ro = (int)((ref.y >> 7) & 127u) - kRefBias, co = (int)(ref.y & 127u) - kRefBias;
        double acc[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = 0.0;
        uint32_t e = ref.x;
        if constexpr (VAR == 5) {
          // two terms at a time: both index pairs first, then the four rows (independent loads)
          for (; e + 2 <= ref.x + cnt; e += 2) {
            const uint2 en0 = pair_ent[e], en1 = pair_ent[e + 1];
            const double2* pa0 = reinterpret_cast<const double2*>(frow + (size_t)en0.x * kRow);
            const double2* pb0 = reinterpret_cast<const double2*>(frow + (size_t)en0.y * kRow);
            const double2* pa1 = reinterpret_cast<const double2*>(frow + (size_t)en1.x * kRow);
            const double2* pb1 = reinterpret_cast<const double2*>(frow + (size_t)en1.y * kRow);
            const double2 a00 = pa0[0], a01 = pa0[1], a02 = pa0[2], b00 = pb0[0], b01 = pb0[1], b02 = pb0[2];
            const double2 a10 = pa1[0], a11 = pa1[1], a12 = pa1[2], b10 = pb1[0], b11 = pb1[1], b12 = pb1[2];
            const double a0[6] = {a00.x, a00.y, a01.x, a01.y, a02.x, a02.y};
            const double b0[6] = {b00.x, b00.y, b01.x, b01.y, b02.x, b02.y};
            const double a1[6] = {a10.x, a10.y, a11.x, a11.y, a12.x, a12.y};
            const double b1[6] = {b10.x, b10.y, b11.x, b11.y, b12.x, b12.y};
#pragma unroll
            for (int x = 0; x < 6; ++x)
#pragma unroll
              for (int y = 0; y < 6; ++y) acc[x * 6 + y] += a0[x] * b0[y];
#pragma unroll
            for (int x = 0; x < 6; ++x)
#pragma unroll
              for (int y = 0; y < 6; ++y) acc[x * 6 + y] += a1[x] * b1[y];
          }
        }
        for (; e < ref.x + cnt; ++e) {
          const uint2 en = pair_ent[e];
          const double2* pa = reinterpret_cast<const double2*>(frow + (size_t)en.x * kRow);
          const double2* pb = reinterpret_cast<const double2*>(frow + (size_t)en.y * kRow);
          const double2 a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
          const double a[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
          const double b[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
#pragma unroll
          for (int x = 0; x < 6; ++x)
#pragma unroll
            for (int y = 0; y < 6; ++y) acc[x * 6 + y] += a[x] * b[y];
        }
        // block (i, j), i < j, lives transposed in the lower storage: row = j D + y, column = i D + x
#pragma unroll
        for (int y = 0; y < 6; ++y) {
          const int rr = ro + y;
          if (rr < 0 || rr >= 64) continue;
#pragma unroll
          for (int x = 0; x < 6; ++x) {
            const int cc = co + x;
            if (cc >= 0 && cc < 64) T[rr * TS + cc] = acc[x * 6 + y];
          }
        }
      }
    } else {
      // work item = (block, x): the six threads of a block each form one row x of a (x) b — they read
      // the same rowB (one request) and consecutive doubles of rowA
      const uint32_t items = (q1 - q0) * 6;
      for (uint32_t it = tid; it < items; it += NT) {
        const uint32_t q = q0 + it / 6, x = it - (it / 6) * 6;
        const uint2 ref = tile_ref[q];
        const uint32_t cnt = ref.y >> 14;
        const int ro = (int)((ref.y >> 7) & 127u) - kRefBias, co = (int)(ref.y & 127u) - kRefBias;
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        uint32_t e = ref.x;
        const uint32_t e1 = ref.x + cnt;
        if constexpr (VAR == 2) {
          for (; e + 4 <= e1; e += 4) {
            uint2 en[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) en[u] = pair_ent[e + u];
            double ax[4];
            double2 b[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              ax[u] = frow[(size_t)en[u].x * kRow + x];
              const double2* pb = reinterpret_cast<const double2*>(frow + (size_t)en[u].y * kRow);
              b[u][0] = pb[0]; b[u][1] = pb[1]; b[u][2] = pb[2];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              acc[0] += ax[u] * b[u][0].x; acc[1] += ax[u] * b[u][0].y; acc[2] += ax[u] * b[u][1].x;
              acc[3] += ax[u] * b[u][1].y; acc[4] += ax[u] * b[u][2].x; acc[5] += ax[u] * b[u][2].y;
            }
          }
          if (e < e1) {  // 1..3 terms left: same shape, out-of-range slots read the last term with weight 0
            uint2 en[3];
            double wgt[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              const uint32_t ee = e + u < e1 ? e + u : e1 - 1;
              en[u] = pair_ent[ee];
              wgt[u] = e + u < e1 ? 1.0 : 0.0;
            }
            double ax[3];
            double2 b[3][3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              ax[u] = frow[(size_t)en[u].x * kRow + x] * wgt[u];
              const double2* pb = reinterpret_cast<const double2*>(frow + (size_t)en[u].y * kRow);
              b[u][0] = pb[0]; b[u][1] = pb[1]; b[u][2] = pb[2];
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              if (wgt[u] != 0.0) {
                acc[0] += ax[u] * b[u][0].x; acc[1] += ax[u] * b[u][0].y; acc[2] += ax[u] * b[u][1].x;
                acc[3] += ax[u] * b[u][1].y; acc[4] += ax[u] * b[u][2].x; acc[5] += ax[u] * b[u][2].y;
              }
            }
          }
        } else {
          for (; e < e1; ++e) {
            const uint2 en = pair_ent[e];
            const double ax = frow[(size_t)en.x * kRow + x];
            const double2* pb = reinterpret_cast<const double2*>(frow + (size_t)en.y * kRow);
            const double2 b0 = pb[0], b1 = pb[1], b2 = pb[2];
            acc[0] += ax * b0.x; acc[1] += ax * b0.y; acc[2] += ax * b1.x;
            acc[3] += ax * b1.y; acc[4] += ax * b2.x; acc[5] += ax * b2.y;
          }
        }
        const int cc = co + (int)x;
        if (cc >= 0 && cc < 64) {
#pragma unroll
          for (int y = 0; y < 6; ++y) {
            const int rr = ro + y;
            if (rr >= 0 && rr < 64) T[rr * TS + cc] = acc[y];
          }
        }
      }
    }
    __syncthreads();
  }
  // 64 rows of 64 doubles: a half-wave writes one 512-byte row per store
  double* base = A + ((size_t)tr * 64) * ld + (size_t)tc * 64;
  const int c2 = tid & 31;
#pragma unroll
  for (int r = tid >> 5; r < 64; r += NT / 32) {
    const double2 v = q1 > q0 ? *reinterpret_cast<const double2*>(T + r * TS + 2 * c2) : make_double2(0.0, 0.0);
    if (VAR == 4 && r >= 8) break;  // experiment: gather only (the tile is NOT written in full: wrong results)
    // 14 GB of tiles nobody re-reads before the factorisation reaches them: streaming stores (-2 %)
    typedef double d2_t __attribute__((ext_vector_type(2)));
    d2_t v2; v2.x = v.x; v2.y = v.y;
    __builtin_nontemporal_store(v2, reinterpret_cast<d2_t*>(base + (size_t)r * ld + 2 * c2));
  }
}

// k_pose_blocks — one workgroup per active pose: its diagonal block and its right-hand sides from
// the pose's term list (structure.h: pose_ent).  Terms before `mid` are the J rows of the pose,
//   U_ii += row row^T,   rhs_p_i += row * sqrt(w) r                      (jt_pr j_pr, jt_pr r_pr: :337-353)
// terms after it the Schur part,
//   S_ii += (-W V^-1)_k W_k^T,   rhs_sc_i += (-W V^-1)_k b_l             (:468-484)
// Fixed reduction order (strided terms per thread, xor-butterfly per wave, waves in order): bitwise
// reproducible.  The block goes to a side buffer (k_write_diag places it after the tile assembly):
// the kernel depends on the factor rows only and runs on the engine's second stream, concurrently
// with k_assemble_tiles.
__global__ void __launch_bounds__(256)
k_pose_blocks(const uint32_t* __restrict__ pose_ptr, const uint32_t* __restrict__ pose_mid,
              const uint32_t* __restrict__ pose_ent, const double* __restrict__ frow,
              const double* __restrict__ scal, int D, double* __restrict__ diag_out,
              double* __restrict__ rhs_p, double* __restrict__ rhs_sc) {
  __shared__ double red[4][48];
  const uint32_t i = blockIdx.x;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const uint32_t e0 = pose_ptr[i], em = pose_mid[i], e1 = pose_ptr[i + 1];
  double acc[48];  // 36 block entries | 6 rhs_p | 6 Schur part of the rhs
#pragma unroll
  for (int k = 0; k < 48; ++k) acc[k] = 0.0;
  for (uint32_t e = e0 + tid; e < e1; e += 256) {
    const uint32_t ra = pose_ent[3 * (size_t)e], rb = pose_ent[3 * (size_t)e + 1], si = pose_ent[3 * (size_t)e + 2];
    const double2* pa = reinterpret_cast<const double2*>(frow + (size_t)ra * kRow);
    const double2* pb = reinterpret_cast<const double2*>(frow + (size_t)rb * kRow);
    const double2 a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
    const double a[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
    const double b[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
    const double sc = scal[si];
    const double s1 = e < em ? sc : 0.0, s2 = e < em ? 0.0 : sc;
#pragma unroll
    for (int x = 0; x < 6; ++x) {
#pragma unroll
      for (int y = 0; y < 6; ++y) acc[x * 6 + y] += a[x] * b[y];
      acc[36 + x] += a[x] * s1;
      acc[42 + x] += a[x] * s2;
    }
  }
  // (two terms per step with all index loads up front measured 3.55 against 3.41 ms at configs[3]: not kept)
#pragma unroll
  for (int k = 0; k < 48; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[w][k] = v;
  }
  __syncthreads();
  if (tid < 48) {
    double v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    if (tid < 36) {
      diag_out[(size_t)i * 36 + tid] = v;
    } else if (tid < 42) {
      rhs_p[(size_t)i * D + (tid - 36)] = v;
      red[0][tid] = v;  // (own slot: read back below by the thread that owns the Schur part)
    }
  }
  __syncthreads();
  if (tid >= 42 && tid < 48) {
    const double sb = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    rhs_sc[(size_t)i * D + (tid - 42)] = red[0][tid - 6] + sb;
  }
}

// Places the diagonal blocks formed by k_pose_blocks into S — after k_assemble_tiles, which zeroed
// their place — and writes the fixed entries: 1e6 on masked parameters (BundleAdjuster.cpp:587-598;
// shard 0 only, so that the cross-shard sum leaves them exact).  One wavefront per pose.
__global__ void __launch_bounds__(64)
k_write_diag(const double* __restrict__ diag, int D, uint32_t ld, const uint16_t* __restrict__ mask_opt,
             int write_fixed, double* __restrict__ A) {
  const uint32_t i = blockIdx.x;
  const int tid = threadIdx.x;
  const uint16_t m = mask_opt[i];
  if (tid < 36) {
    const int rr = tid / 6, cc = tid - 6 * rr;
    double v = diag[(size_t)i * 36 + tid];
    if (rr == cc && (m & (1u << rr))) v = write_fixed ? 1e6 : 0.0;
    A[((size_t)i * D + rr) * ld + (size_t)i * D + cc] = v;
  } else if (tid - 36 < D - 6) {
    const int k = 6 + (tid - 36);
    if ((m & (1u << k)) && write_fixed) A[((size_t)i * D + k) * ld + (size_t)i * D + k] = 1e6;
  }
}

// ---- calibration border (T_vs columns of the DoTvs instantiations, BundleAdjuster.cpp:493-583) ----
// k_pose_border — the 6 x 6 block S_pk of one active pose from the SAME term list as its diagonal
// block: a term (rowA, rowB, scalar index si) contributes rowA (x) crow[si], because the calibration
// rows are indexed like the scalars —
//   J terms      si = 2a + k      crow = sqrt(w) dz_dtvs row k of observation a:   J_p^T J_k      (:501-518)
//   Schur terms  si = 2O + l      crow = E_l = sum w J_l^T J_k:                 -(W V^-1) E_l    (:538-556)
//   same-pose cross terms carry the zero scalar, whose calibration row is zero.
__global__ void __launch_bounds__(256)
k_pose_border(const uint32_t* __restrict__ pose_ptr, const uint32_t* __restrict__ pose_ent,
              const double* __restrict__ frow, const double* __restrict__ crow, double* __restrict__ out) {
  __shared__ double red[4][36];
  const uint32_t i = blockIdx.x;
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const uint32_t e0 = pose_ptr[i], e1 = pose_ptr[i + 1];
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
  for (uint32_t e = e0 + tid; e < e1; e += 256) {
    const uint32_t ra = pose_ent[3 * (size_t)e], si = pose_ent[3 * (size_t)e + 2];
    const double2* pa = reinterpret_cast<const double2*>(frow + (size_t)ra * kRow);
    const double2* pc = reinterpret_cast<const double2*>(crow + (size_t)si * kRow);
    const double2 a0 = pa[0], a1 = pa[1], a2 = pa[2], c0 = pc[0], c1 = pc[1], c2 = pc[2];
    const double a[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
    const double c[6] = {c0.x, c0.y, c1.x, c1.y, c2.x, c2.y};
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int y = 0; y < 6; ++y) acc[x * 6 + y] += a[x] * c[y];
  }
#pragma unroll
  for (int k = 0; k < 36; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[w][k] = v;
  }
  __syncthreads();
  if (tid < 36) out[(size_t)i * 36 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// k_calib_reduce — S_kk and rhs_k: one pass over the calibration rows (fixed grid, strided items,
// fixed trees: reproducible).  33 sums: the 21 unique entries of S_kk, J_k^T r (6), and the Schur part
// of rhs_k (6):
//   observation rows:  S_kk += c c^T,          rhs_k += c * sqrt(w) r                     (:494-499, 520-522)
//   landmark rows:     S_kk -= V^-1 E E^T,     rhs_k_sc -= E V^-1 b_l                     (:558-582)
__global__ void __launch_bounds__(256)
k_calib_reduce(uint32_t n_obs_rows, uint32_t L, const double* __restrict__ crow, const double* __restrict__ scal,
               const double* __restrict__ lm_vinv, const int32_t* __restrict__ lm_opt,
               double* __restrict__ partials, uint32_t nparts) {
  __shared__ double red[4][33];
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  double acc[33];
#pragma unroll
  for (int k = 0; k < 33; ++k) acc[k] = 0.0;
  const size_t total = (size_t)n_obs_rows + L;
  for (size_t idx = (size_t)blockIdx.x * 256 + tid; idx < total; idx += (size_t)gridDim.x * 256) {
    const double2* pc = reinterpret_cast<const double2*>(crow + idx * kRow);
    const double2 c0 = pc[0], c1 = pc[1], c2 = pc[2];
    const double c[6] = {c0.x, c0.y, c1.x, c1.y, c2.x, c2.y};
    double f = 1.0, s_raw = scal[idx], s_sc = 0.0;
    if (idx >= n_obs_rows) {
      const size_t l = idx - n_obs_rows;
      const double vi = lm_opt[l] >= 0 ? lm_vinv[l] : 0.0;
      f = -vi; s_sc = -vi * s_raw; s_raw = 0.0;
    }
    int k = 0;
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
      for (int y = x; y < 6; ++y) acc[k++] += f * c[x] * c[y];
#pragma unroll
    for (int x = 0; x < 6; ++x) { acc[21 + x] += c[x] * s_raw; acc[27 + x] += c[x] * s_sc; }
  }
#pragma unroll
  for (int k = 0; k < 33; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[w][k] = v;
  }
  __syncthreads();
  if (tid < 33) partials[(size_t)tid * nparts + blockIdx.x] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// Sums the block partials of k_calib_reduce in block order and places S_kk (both triangles of the
// 6 x 6 block, like the pose blocks), rhs_k (unreduced, tail of rhs_p) and its reduced form (tail of
// rhs_sc); k_write_border places the S_pk blocks as rows np .. np+5 of the lower storage.
__global__ void __launch_bounds__(64)
k_calib_finish(uint32_t nparts, const double* __restrict__ partials, uint32_t np, uint32_t ld, int K,
               double* __restrict__ A, double* __restrict__ rhs_p, double* __restrict__ rhs_sc) {
  __shared__ double tot[33];
  const int tid = threadIdx.x;
  if (tid < 33) {
    double s = 0.0;
    for (uint32_t b = 0; b < nparts; ++b) s += partials[(size_t)tid * nparts + b];
    tot[tid] = s;
  }
  __syncthreads();
  if (tid < 36) {
    const int r = tid / 6, c = tid - 6 * r, x = r < c ? r : c, y = r < c ? c : r;
    if (r < K && c < K) A[((size_t)np + r) * ld + np + c] = tot[x * 6 - x * (x - 1) / 2 + (y - x)];
  } else if (tid < 42) {
    const int k = tid - 36;
    if (k < K) {
      rhs_p[np + k] = tot[21 + k];
      rhs_sc[np + k] = tot[21 + k] + tot[27 + k];
    }
  }
}
__global__ void __launch_bounds__(64)
k_write_border(const double* __restrict__ border, int D, uint32_t np, uint32_t ld, int K, double* __restrict__ A) {
  const uint32_t i = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid < 36) {
    const int r = tid / 6, c = tid - 6 * r;
    if (c < K) A[((size_t)np + c) * ld + (size_t)i * D + r] = border[(size_t)i * 36 + tid];
  }
}

// rows n..n_pad-1 of the padded system: identity
__global__ void k_pad_diag(uint32_t n, uint32_t n_pad, uint32_t ld, double* __restrict__ A) {
  const uint32_t k = n + blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_pad) A[(size_t)k * ld + k] = 1.0;
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

// launch entry of the tile assembly: (tile id, first / end block reference, tile row << 16 | tile column)
__global__ void k_tile_desc(uint32_t n, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ tile_ptr,
                            uint4* __restrict__ desc) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  const uint32_t t = tile_order[b];
  if (t == 0xffffffffu) { desc[b] = make_uint4(t, 0u, 0u, 0u); return; }
  uint32_t tr = (uint32_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((uint64_t)(tr + 1) * (tr + 2) / 2 <= t) ++tr;
  while ((uint64_t)tr * (tr + 1) / 2 > t) --tr;
  const uint32_t tc = t - (uint32_t)((uint64_t)tr * (tr + 1) / 2);
  desc[b] = make_uint4(t, tile_ptr[t], tile_ptr[t + 1], (tr << 16) | tc);
}

int build_tile_desc(Engine* e) {
  BAE_HIP(e->tile_desc.alloc(std::max<uint32_t>(e->n_tile_order, 1)));
  if (e->n_tile_order == 0) return 0;
  hipLaunchKernelGGL(k_tile_desc, dim3((e->n_tile_order + 255) / 256), dim3(256), 0, e->stream, e->n_tile_order,
                     (const uint32_t*)e->tile_order.p, (const uint32_t*)e->tile_ptr.p, e->tile_desc.p);
  BAE_HIP(hipGetLastError());
  return 0;
}

int launch_gather_S(Engine* e) {
  const Structure& st = e->st;
  const uint32_t n = st.n, ld = st.ld, n_pad = ld;
  // the whole square + rhs row is zeroed once per structure (the tiles above the diagonal are
  // never written afterwards); every iteration k_assemble_tiles rewrites the lower triangle
  if (e->A_cleared != e->A.p) {
    BAE_HIP(hipMemsetAsync(e->A.p, 0, (size_t)(n_pad + 1) * ld * sizeof(double), e->stream));
    e->A_cleared = e->A.p;
  }
  const uint32_t nt = n_pad / 64;
  {
    int rc = build_tile_order(e);
    if (rc) return rc;
  }
  BAE_HIP(hipMemsetAsync(e->rhs_p.p, 0, e->rhs_p.bytes(), e->stream));
  BAE_HIP(hipMemsetAsync(e->rhs_sc.p, 0, e->rhs_sc.bytes(), e->stream));
  // fork: the per-pose blocks and right-hand sides (they read the factor rows only) on the second
  // stream, concurrently with the tile assembly
  if (st.Pact > 0) {
    if (!e->ev_fork) {
      BAE_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
      BAE_HIP(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    }
    BAE_HIP(e->diag_blocks.alloc((size_t)st.Pact * 36));
    BAE_HIP(hipEventRecord(e->ev_fork, e->stream));
    BAE_HIP(hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
    e->prof_begin(e->ev_pose, e->stream2);
    hipLaunchKernelGGL(k_pose_blocks, dim3(st.Pact), dim3(256), 0, e->stream2, e->pose_ptr.p, e->pose_mid.p,
                       e->pose_ent.p, e->frow.p, e->scal.p, e->pose_dim, e->diag_blocks.p, e->rhs_p.p, e->rhs_sc.p);
    e->prof_end(e->ev_pose, e->stream2);
    BAE_HIP(hipGetLastError());
    if (st.K) {
      hipLaunchKernelGGL(k_pose_border, dim3(st.Pact), dim3(256), 0, e->stream2, e->pose_ptr.p, e->pose_ent.p,
                         e->frow.p, e->crow.p, e->border_blocks.p);
      BAE_HIP(hipGetLastError());
    }
    BAE_HIP(hipEventRecord(e->ev_join, e->stream2));
  }
  e->prof_begin(e->ev_gather);
#define BAE_ASM(V)                                                                                       \
  hipLaunchKernelGGL(k_assemble_tiles<V>, dim3(e->n_tile_order), dim3(V == 6 ? 128 : 256), 0, e->stream, nt, e->tile_order.p, \
                     e->tile_ptr.p, e->tile_ref.p, e->pair_ent.p, e->frow.p, ld, e->A.p, (const uint4*)e->tile_desc.p)
  switch (e->dbg_assemble_variant) {
    case 0: BAE_ASM(0); break; case 1: BAE_ASM(1); break; case 2: BAE_ASM(2); break;
    case 3: BAE_ASM(3); break; case 4: BAE_ASM(4); break; case 5: BAE_ASM(5); break; default: BAE_ASM(6); break;
  }
#undef BAE_ASM
  e->prof_end(e->ev_gather);
  BAE_HIP(hipGetLastError());
  BAE_HIP(hipMemsetAsync(e->A.p + (size_t)n_pad * ld, 0, (size_t)ld * sizeof(double), e->stream));
  // fixed entries (padding identity, 1e6 on masked parameters) are written by shard 0
  // only, so that the cross-shard sum of S leaves them exact
  const int write_fixed = (e->rank == 0) ? 1 : 0;
  if (n_pad > n && write_fixed) {
    hipLaunchKernelGGL(k_pad_diag, dim3((n_pad - n + 255) / 256), dim3(256), 0, e->stream, n, n_pad,
                       ld, e->A.p);
    BAE_HIP(hipGetLastError());
  }
  if (st.Pact > 0) {
    // join: diagonal blocks into the assembled matrix
    BAE_HIP(hipStreamWaitEvent(e->stream, e->ev_join, 0));
    const uint16_t* masks = e->pose_mask.p + st.P;  // masks by opt id live after the by-id masks
    hipLaunchKernelGGL(k_write_diag, dim3(st.Pact), dim3(64), 0, e->stream, (const double*)e->diag_blocks.p,
                       e->pose_dim, ld, masks, write_fixed, e->A.p);
    BAE_HIP(hipGetLastError());
    if (st.K) {
      hipLaunchKernelGGL(k_write_border, dim3(st.Pact), dim3(64), 0, e->stream, (const double*)e->border_blocks.p,
                         e->pose_dim, st.np, ld, (int)st.K, e->A.p);
      BAE_HIP(hipGetLastError());
    }
  }
  if (st.K) {
    int rc = launch_calib_border(e);
    if (rc) return rc;
  }
  return 0;
}

// S_kk and rhs_k (k_calib_reduce, k_calib_finish): after the tile assembly zeroed the border tiles
int launch_calib_border(Engine* e) {
  const Structure& st = e->st;
  const size_t total = 2 * (size_t)st.O + st.L;
  const uint32_t nb = (uint32_t)std::min<size_t>(1024, std::max<size_t>((total + 255) / 256, 1));
  DBuf<double>& part = e->calib_partials;
  BAE_HIP(part.alloc(33 * 1024));
  hipLaunchKernelGGL(k_calib_reduce, dim3(nb), dim3(256), 0, e->stream, 2 * st.O, st.L, e->crow.p, e->scal.p,
                     e->lm_vinv.p, e->lm_opt.p, part.p, nb);
  BAE_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_calib_finish, dim3(1), dim3(64), 0, e->stream, nb, (const double*)part.p, st.np, st.ld,
                     (int)st.K, e->A.p, e->rhs_p.p, e->rhs_sc.p);
  BAE_HIP(hipGetLastError());
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

// first level for long partial lists (one partial per linearisation wave: 167k at configs[3]): 64
// blocks each sum a contiguous slice in fixed order into partials2
__global__ void k_sum_partials_l1(uint32_t nparts, uint32_t nblk, const double* __restrict__ parts,
                                  double* __restrict__ out) {
  __shared__ double red[256];
  const uint32_t per = (nparts + nblk - 1) / nblk;
  const uint32_t i0 = blockIdx.x * per, i1 = min(nparts, i0 + per);
  double s = 0.0;
  for (uint32_t i = i0 + threadIdx.x; i < i1; i += 256) s += parts[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

void defer_begin(Engine* e) {
  e->defer_active = !e->sharded();
  e->defer_n = 0;
}
int defer_flush(Engine* e) {
  const bool was = e->defer_active;
  const int n = e->defer_n;
  e->defer_active = false;
  e->defer_n = 0;
  if (!was || n == 0) return 0;
  double host[40];
  BAE_HIP(hipMemcpyAsync(host, e->scalars_out.p + 16, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  BAE_HIP(hipStreamSynchronize(e->stream));
  for (int i = 0; i < n; ++i) *e->defer_host[i] = host[i];
  return 0;
}

int sum_partials(Engine* e, uint32_t nparts, uint32_t ncomp, double* host_out, bool cross_shard) {
  if (e->defer_active && nparts != 0 && e->defer_n + (int)ncomp <= 40) {
    // deferred: the sum stays on the device until defer_flush (never reached by a sharded engine)
    const double* src = e->partials.p;
    if (ncomp == 1 && nparts > 8192) {
      hipLaunchKernelGGL(k_sum_partials_l1, dim3(64), dim3(256), 0, e->stream, nparts, 64u, (const double*)e->partials.p,
                         e->scalars_out.p + 64);
      BAE_HIP(hipGetLastError());
      src = e->scalars_out.p + 64;
      nparts = 64;
    }
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, e->stream, nparts, ncomp, src,
                       e->scalars_out.p + 16 + e->defer_n);
    BAE_HIP(hipGetLastError());
    for (uint32_t c = 0; c < ncomp; ++c) e->defer_host[e->defer_n + c] = host_out + c;
    e->defer_n += (int)ncomp;
    return 0;
  }
  if (nparts == 0) {
    for (uint32_t c = 0; c < ncomp; ++c) host_out[c] = 0.0;
  } else {
    const double* src = e->partials.p;
    if (ncomp == 1 && nparts > 8192) {
      hipLaunchKernelGGL(k_sum_partials_l1, dim3(64), dim3(256), 0, e->stream, nparts, 64u, (const double*)e->partials.p,
                         e->scalars_out.p + 64);
      BAE_HIP(hipGetLastError());
      src = e->scalars_out.p + 64;
      nparts = 64;
    }
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, e->stream, nparts, ncomp,
                       src, e->scalars_out.p);
    BAE_HIP(hipGetLastError());
    BAE_HIP(hipMemcpyAsync(host_out, e->scalars_out.p, ncomp * sizeof(double),
                           hipMemcpyDeviceToHost, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
  }
  if (cross_shard && e->sharded()) {
    // cross-shard sum of the scalars
    BAE_HIP(hipMemcpyAsync(e->scalars_out.p, host_out, ncomp * sizeof(double),
                           hipMemcpyHostToDevice, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
    if (shard_allreduce(e, e->scalars_out.p, ncomp, 0) != 0)
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

// Small inputs (a sliding window's residuals): the six passes in ONE workgroup — histogram in LDS, the
// walk over the bins by the first wavefront exactly as k_select_scan does it — instead of twelve launches.
__global__ void __launch_bounds__(1024)
k_select_small(uint32_t n, const double* __restrict__ v, unsigned long long k0, unsigned long long* __restrict__ sel) {
  __shared__ unsigned int h[2048];
  __shared__ unsigned long long st[4];  // prefix, mask, remaining rank, error
  const int tid = threadIdx.x;
  if (tid == 0) { st[0] = 0; st[1] = 0; st[2] = k0; st[3] = 0; }
  for (int pass = 0; pass < 6; ++pass) {
    const int shift = pass == 5 ? 0 : 53 - 11 * pass;
    const unsigned int dmask = pass == 5 ? 511u : 2047u;
    const int nbins = pass == 5 ? 512 : 2048;
    for (int i = tid; i < 2048; i += 1024) h[i] = 0;
    __syncthreads();
    const unsigned long long prefix = st[0], pmask = st[1];
    for (uint32_t i = tid; i < n; i += 1024) {
      const unsigned long long b = (unsigned long long)__double_as_longlong(v[i]);
      if ((b & pmask) == prefix) atomicAdd(&h[(b >> shift) & dmask], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      const int lane = tid;
      const unsigned long long k = st[2];
      unsigned long long loc = 0;
      for (int b = 0; b < 32; ++b) loc += h[32 * lane + b];
      unsigned long long incl = loc;
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
      }
      const unsigned long long excl = incl - loc;
      const bool mine = (k >= excl) && (k < incl) && (32 * lane < nbins);
      unsigned long long r = k - excl;
      int b = 0;
      if (mine)
        for (; b < 32; ++b) {
          const unsigned long long hb = h[32 * lane + b];
          if (r < hb) break;
          r -= hb;
        }
      const unsigned long long found = __ballot(mine);
      if (mine) {
        st[0] = prefix | (((unsigned long long)(32 * lane + b)) << shift);
        st[1] = pmask | ((unsigned long long)dmask << shift);
        st[2] = r;
      }
      if (lane == 0 && found == 0) st[3] = 1;  // rank out of range
    }
    __syncthreads();
  }
  if (tid < 4) sel[tid] = st[tid];
}

static int select_kth_device(Engine* e, const double* d_values, uint32_t n_local, uint64_t k, double* out) {
  static const int shifts[6] = {53, 42, 31, 20, 9, 0};
  unsigned long long init[4] = {0ull, 0ull, (unsigned long long)k, 0ull};
  unsigned long long* sel = e->hist.p + 2048;
  if (n_local <= 65536u) {
    hipLaunchKernelGGL(k_select_small, dim3(1), dim3(1024), 0, e->stream, n_local, d_values, (unsigned long long)k, sel);
    BAE_HIP(hipGetLastError());
    unsigned long long res[4];
    BAE_HIP(hipMemcpyAsync(res, sel, sizeof(res), hipMemcpyDeviceToHost, e->stream));
    BAE_HIP(hipStreamSynchronize(e->stream));
    if (res[3]) return e->fail_msg("select_kth: rank out of range");
    memcpy(out, &res[0], sizeof(double));
    return 0;
  }
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
  if (!(e->sharded()) && n_local > 0) return select_kth_device(e, d_values, n_local, k, out);
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
    if (e->sharded())
      if (shard_allreduce(e, e->hist.p, 2048, 1) != 0)
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
  if (st.K) {  // delta_k: composed like the pose part, left out of the norm (BundleAdjuster.cpp:26)
    hipLaunchKernelGGL(k_compose, dim3(1), dim3(256), 0, e->stream, st.K, coef_rhs, coef_gn, e->rhs_p.p + st.np,
                       e->gn_p.p + st.np, e->step_p.p + st.np, e->partials.p);
    BAE_HIP(hipGetLastError());
  }
  if (st.np > 0) {
    const uint32_t nb = (st.np + 255) / 256;
    hipLaunchKernelGGL(k_compose, dim3(nb), dim3(256), 0, e->stream, st.np, coef_rhs, coef_gn,
                       e->rhs_p.p, e->gn_p.p, e->step_p.p, e->partials.p);
    BAE_HIP(hipGetLastError());
    // the pose step is replicated on every shard: no cross-shard sum
    int rc = sum_partials(e, nb, 1, &norms2_host[0], false);
    if (rc) return rc;
  }
  if (st.L > 0 && e->lm_dim > 0 && st.Lact > 0) {
    const uint32_t nb = (st.L + 255) / 256;
    hipLaunchKernelGGL(k_compose_lm, dim3(nb), dim3(256), 0, e->stream, st.L, e->lm_dim, coef_rhs,
                       coef_gn, e->lm_opt.p, e->lm_bl.p, e->gn_l.p, e->step_l.p, e->partials.p);
    BAE_HIP(hipGetLastError());
    int rc = sum_partials(e, nb, 1, &norms2_host[1]);  // landmarks are sharded: summed
    if (rc) return rc;
  } else if (e->sharded()) {
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
// per observation: v = sqrt(w) (Jm rhs_p[m] + Jr rhs_p[r] + Jl rhs_l[l]); sum |v|^2.  The J rows of
// observation a are rows a R, a R + 1 (measuring pose) and a R + 2, a R + 3 (reference pose, LM 1)
// of the factor rows; a side counts when the observation is listed and the pose active.
__global__ void k_jrhs(uint32_t O, int D, int LM, const uint32_t* __restrict__ obs_pose,
                       const uint32_t* __restrict__ obs_lm, const uint32_t* __restrict__ lm_ref_pose,
                       const int32_t* __restrict__ pose_opt, const int32_t* __restrict__ lm_opt,
                       const double* __restrict__ frow, const double* __restrict__ obs_jl,
                       const double* __restrict__ rhs_p, const double* __restrict__ bl,
                       double* __restrict__ partials) {
  __shared__ double red[256];
  const size_t a = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int R = LM == 1 ? 6 : 8;
  double sq = 0.0;
  if (a < O) {
    double v0 = 0, v1 = 0;
    const uint32_t l = obs_lm[a], pm = obs_pose[a], rp = lm_ref_pose[l];
    const bool listed = LM != 1 || pm != rp;
    const int om = listed ? pose_opt[pm] : -1, orr = (LM == 1 && listed) ? pose_opt[rp] : -1;
    if (om >= 0) {
      const double* g = rhs_p + (size_t)om * D;
      const double* r0 = frow + (a * R) * kRow;
      for (int c = 0; c < 6; ++c) { v0 += r0[c] * g[c]; v1 += r0[kRow + c] * g[c]; }
    }
    if (orr >= 0) {
      const double* g = rhs_p + (size_t)orr * D;
      const double* r0 = frow + (a * R + 2) * kRow;
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

// The seven sums land in h7 = (rhs_p_sq, gn_p_sq, rhs_gn_p | rhs_l_sq, gn_l_sq, rhs_gn_l | j_rhs_sq): a buffer
// that outlives a deferred flush (Engine::dog_h); ba_hip_dogleg_terms moves them into the result.
// skip_jrhs: the denominator term is known from an earlier call at the same linearisation (h7[6] untouched)
int launch_dogleg(Engine* e, int gn_available, double* h7, bool skip_jrhs) {
  const Structure& st = e->st;
  for (int i = 0; i < (skip_jrhs ? 6 : 7); ++i) h7[i] = 0.0;
  int rc;
  if (st.np > 0) {  // pose parts: replicated on every shard
    const uint32_t nb = (st.np + 255) / 256;
    hipLaunchKernelGGL(k_dots_pose, dim3(nb), dim3(256), 0, e->stream, st.np, gn_available,
                       e->rhs_p.p, e->gn_p.p, e->partials.p, nb);
    BAE_HIP(hipGetLastError());
    if ((rc = sum_partials(e, nb, 3, h7, false))) return rc;
  }
  {  // landmark parts: sharded
    const uint32_t nb = (st.L > 0 && e->lm_dim > 0) ? (st.L + 255) / 256 : 0;
    if (nb) {
      hipLaunchKernelGGL(k_dots_lm, dim3(nb), dim3(256), 0, e->stream, st.L, e->lm_dim, gn_available,
                         e->lm_opt.p, e->lm_bl.p, e->gn_l.p, e->partials.p, nb);
      BAE_HIP(hipGetLastError());
    }
    if ((rc = sum_partials(e, nb, 3, h7 + 3, true))) return rc;
  }
  if (!skip_jrhs) {  // || J_pr rhs_p + J_l rhs_l ||^2 over the observations (BundleAdjuster.cpp:881-906)
    const uint32_t nb = st.O > 0 ? (st.O + 255) / 256 : 0;
    if (nb) {
      hipLaunchKernelGGL(k_jrhs, dim3(nb), dim3(256), 0, e->stream, st.O, e->pose_dim, e->lm_dim,
                         e->obs_pose.p, e->obs_lm.p, e->lm_ref_pose.p, e->pose_opt.p, e->lm_opt.p, e->frow.p,
                         e->obs_jl.p, e->rhs_p.p, e->lm_bl.p, e->partials.p);
      BAE_HIP(hipGetLastError());
    }
    if ((rc = sum_partials(e, nb, 1, h7 + 6, true))) return rc;
  }
  return 0;
}

// Calibration part of the dogleg scalars (BundleAdjuster.cpp:858-859, 883-886, 906-910, 971-1002):
// squared norms / dot of rhs_k and the Gauss-Newton delta_k (replicated: tails of rhs_p / gn_p), and
// || J_k rhs_k ||^2 over the observations, which the reference adds to the denominator as a term of
// its own.
__global__ void __launch_bounds__(256)
k_jk_rhs(uint32_t O, int K, const double* __restrict__ crow, const double* __restrict__ rhs_k, double* __restrict__ partials) {
  __shared__ double red[256];
  const size_t a = (size_t)blockIdx.x * 256 + threadIdx.x;
  double sq = 0.0;
  if (a < O) {
    double u0 = 0, u1 = 0;
    const double* r0 = crow + 2 * a * kRow;
    for (int c = 0; c < K; ++c) { u0 += r0[c] * rhs_k[c]; u1 += r0[kRow + c] * rhs_k[c]; }
    sq = u0 * u0 + u1 * u1;
  }
  red[threadIdx.x] = sq;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

int launch_calib_dogleg(Engine* e, int gn_available, ba_hip_dogleg_scalars* out) {
  const Structure& st = e->st;
  double rk[6] = {0, 0, 0, 0, 0, 0}, gk[6] = {0, 0, 0, 0, 0, 0};
  BAE_HIP(hipMemcpyAsync(rk, e->rhs_p.p + st.np, st.K * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (gn_available) BAE_HIP(hipMemcpyAsync(gk, e->gn_p.p + st.np, st.K * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  BAE_HIP(hipStreamSynchronize(e->stream));
  for (int i = 0; i < 6; ++i) { out->rhs_k_sq += rk[i] * rk[i]; out->gn_k_sq += gk[i] * gk[i]; out->rhs_gn_k += rk[i] * gk[i]; }
  const uint32_t nb = (st.O > 0 && st.Pact > 0) ? (st.O + 255) / 256 : 0;  // :881-886: only with active poses
  if (nb) {
    hipLaunchKernelGGL(k_jk_rhs, dim3(nb), dim3(256), 0, e->stream, st.O, (int)st.K, (const double*)e->crow.p,
                       (const double*)(e->rhs_p.p + st.np), e->partials.p);
    BAE_HIP(hipGetLastError());
  }
  double h = 0.0;
  int rc = sum_partials(e, nb, 1, &h, true);
  if (rc) return rc;
  out->j_rhs_sq += h;
  return 0;
}

}  // namespace bae
