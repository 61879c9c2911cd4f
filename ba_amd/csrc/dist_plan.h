// Ownership map and message plan of the distributed reduced solve — plain
// host C++, no device code: the engine (k_chol.hip) executes the plan, ba_hip_dist_plan_stats and the CPU
// tests only count its bytes.  Replaces nothing in the reference (arpg/ba is single-process); what is
// distributed is CalculateGn, /root/reference/src/BundleAdjuster.cpp:748-833.
//
// The lower triangle of S is cut into BLOCKS of G x G 64-tiles (G = the outer panel width KOUT of the
// single-GPU factorisation, so a column of blocks is one outer panel).  Block row / column b has the
// class b mod T, and the owner of block (bi, bc), bi >= bc, is tbl[class(bi)][class(bc)]:
//
//   tri   T(T-1)/2 "cross" ranks own the blocks of one unordered class pair {a, b}, T/2 "diagonal" ranks
//         the blocks (a, a) of two classes — N = T^2/2 ranks (2, 8, 18): every rank multiplies rows of only
//         TWO classes, so it receives 2/T of the factor (half of it at N = 8), and the blocks of a pair are
//         balanced to one unit of work per rank;
//   grid  Pr x Pc block-cyclic, tbl[a][b] = (a mod Pr) Pc + (b mod Pc), T = lcm(Pr, Pc);
//   col   tbl[a][b] = b (T = N): round-robin column panels — every rank receives the whole factor;
//   row   tbl[a][b] = a (T = N): round-robin block rows — the whole factor again, but a block row never
//         changes hands, so no bulk message sits between two consecutive panels of a row.
//
// Per panel J (tile columns [c0, c1)):
//   1. the owner of block (J, J) factorises the SQUARE (its G diagonal tiles, the tiles between them and
//      the rhs row) and broadcasts it with its factor packets — the only message every rank waits for;
//   2. every rank runs the panel's substitution / in-panel updates on the row tiles it owns below the
//      square; block J+1 first — its rows are the URGENT message (the next square needs them);
//   3. the rows of a block of class a go to the ranks that multiply class a (needs[q][a]), point to
//      point over xGMI: urgent ones on the chain communicator, the rest on the side communicator;
//   4. trailing updates of the tiles a rank owns.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

namespace bae {

static const uint32_t kOwnMaxT = 16;

// Passed by value to the update kernels (plain data).
struct OwnMap {
  uint32_t n, rank, G, T, skip_rhs;
  uint8_t tbl[kOwnMaxT][kOwnMaxT];
  // k_update128<false, true> only (distributed bulk updates): the (block row, block column) pairs this rank owns
  // right of the next panel, G/2 x G/2 128-blocks each — the launch holds just these instead of the whole trailing
  // triangle with 1 - 1/N of its workgroups leaving at the ownership test.  Two 32-bit halves of the device address
  // (plain data for the host-only builds of this header).
  uint32_t pairs_lo, pairs_hi;
};

// Does this launch touch tile (i, c)?  i == nblk is the rhs row; its segment over block column bc rides
// with the diagonal block (bc, bc).  n <= 1: everything (single GPU).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline bool own_tile(const OwnMap& m, uint32_t i, uint32_t c, uint32_t nblk) {
  if (i == nblk && m.skip_rhs) return false;
  if (m.n <= 1) return true;
  const uint32_t cc = (c / m.G) % m.T;
  const uint32_t rc = (i == nblk) ? cc : (i / m.G) % m.T;
  return m.tbl[rc][cc] == m.rank;
}

inline OwnMap own_map_single() {
  OwnMap m = {};
  m.n = 1; m.rank = 0; m.G = 1; m.T = 1; m.skip_rhs = 0;
  return m;
}

// layout: "auto" | "tri" | "grid" | "col" | "row".  False (with *why) if it does not exist for nranks.
inline bool build_own_map(uint32_t nranks, const char* layout, uint32_t G, OwnMap* out, std::string* name,
                          std::string* why = nullptr) {
  OwnMap m = {};
  m.n = nranks; m.rank = 0; m.G = G ? G : 1; m.skip_rhs = 0;
  std::string lay = layout && *layout ? layout : "auto";
  auto tri_T = [&](uint32_t n) -> uint32_t {
    for (uint32_t t = 2; t <= kOwnMaxT; t += 2)
      if (t * t / 2 == n) return t;
    return 0;
  };
  if (nranks <= 1) {
    *out = own_map_single();
    out->G = m.G;
    if (name) *name = "single";
    return true;
  }
  if (lay == "auto") lay = tri_T(nranks) ? "tri" : "grid";
  if (lay == "tri") {
    const uint32_t T = tri_T(nranks);
    if (!T) { if (why) *why = "layout tri needs T*T/2 ranks (2, 8, 18, ...)"; return false; }
    m.T = T;
    uint32_t r = 0;
    for (uint32_t a = 0; a < T; ++a)
      for (uint32_t b = a + 1; b < T; ++b) { m.tbl[a][b] = m.tbl[b][a] = (uint8_t)r; ++r; }
    for (uint32_t k = 0; k < T / 2; ++k) { m.tbl[2 * k][2 * k] = m.tbl[2 * k + 1][2 * k + 1] = (uint8_t)(r + k); }
  } else if (lay == "grid") {
    uint32_t pr = 1;
    for (uint32_t d = 1; d * d <= nranks; ++d)
      if (nranks % d == 0) pr = d;
    const uint32_t pc = nranks / pr;
    uint32_t T = pc;
    while (T % pr) T += pc;
    if (T > kOwnMaxT) { if (why) *why = "layout grid: lcm(Pr, Pc) exceeds the class limit"; return false; }
    m.T = T;
    for (uint32_t a = 0; a < T; ++a)
      for (uint32_t b = 0; b < T; ++b) m.tbl[a][b] = (uint8_t)((a % pr) * pc + (b % pc));
  } else if (lay == "col" || lay == "row") {
    if (nranks > kOwnMaxT) { if (why) *why = "more ranks than classes"; return false; }
    m.T = nranks;
    for (uint32_t a = 0; a < m.T; ++a)
      for (uint32_t b = 0; b < m.T; ++b) m.tbl[a][b] = (uint8_t)(lay == "col" ? b : a);
  } else {
    if (why) *why = "unknown layout " + lay;
    return false;
  }
  *out = m;
  if (name) *name = lay;
  return true;
}

struct DistMsg {
  uint32_t panel, src;
  uint32_t first, count;        // tile list: tiles[first, first + count)
  std::vector<uint32_t> dst;    // receivers (never src, except the single-rank self test)
  bool urgent;
};

struct DistPanel {
  uint32_t c0, c1, diag_owner;
  // per rank: the row tiles it owns below the square — block J+1 (urgent) / the others — as ranges of `tiles`
  std::vector<uint32_t> own_u_first, own_u_count, own_r_first, own_r_count;
  std::vector<uint32_t> msgs;   // indices into DistPlan::msgs; urgent ones first
  // per rank: does its update of column block J+1 with this panel use a row another rank computed and sent on the side
  // stream?  (never in the row layout: a block row never changes hands — the panel stream then runs ahead of the side
  // transfers)
  std::vector<uint8_t> next_needs_side;
};

struct DistPlan {
  OwnMap map;
  uint32_t nblk = 0, nb = 0;
  std::vector<uint32_t> tiles;
  std::vector<DistPanel> panels;
  std::vector<DistMsg> msgs;
  std::vector<uint8_t> needs;   // [n][T]: rank q multiplies rows of class a
  bool need(uint32_t q, uint32_t a) const { return needs[(size_t)q * map.T + a] != 0; }
};

// nzL: nblk x nblk bytes, lower pattern of the factor (row-major, nz[i * nblk + k]); nullptr = dense.
// self_messages: single-rank self test — the one rank sends its rows to itself (exercises every code path).
inline DistPlan build_dist_plan(uint32_t nblk, const OwnMap& map, const uint8_t* nzL, bool self_messages = false) {
  DistPlan p;
  p.map = map;
  p.nblk = nblk;
  const uint32_t G = map.G, T = map.T, n = std::max(map.n, 1u);
  p.nb = (nblk + G - 1) / G;
  p.needs.assign((size_t)n * T, 0);
  for (uint32_t a = 0; a < T; ++a)
    for (uint32_t b = 0; b < T; ++b) {
      const uint32_t r = n > 1 ? map.tbl[a][b] : 0;
      p.needs[(size_t)r * T + a] = 1;   // row operand of a tile it owns
      p.needs[(size_t)r * T + b] = 1;   // column operand
    }
  p.panels.resize(p.nb);
  std::vector<std::vector<uint32_t>> own_u(n), own_r(n);
  for (uint32_t J = 0; J < p.nb; ++J) {
    DistPanel& pn = p.panels[J];
    pn.c0 = J * G;
    pn.c1 = std::min(pn.c0 + G, nblk);
    pn.diag_owner = n > 1 ? map.tbl[J % T][J % T] : 0;
    for (uint32_t r = 0; r < n; ++r) { own_u[r].clear(); own_r[r].clear(); }
    for (uint32_t i = pn.c1; i < nblk; ++i) {
      bool on = nzL == nullptr;
      for (uint32_t kb = pn.c0; kb < pn.c1 && !on; ++kb) on = nzL[(size_t)i * nblk + kb] != 0;
      if (!on) continue;
      const uint32_t r = n > 1 ? map.tbl[(i / G) % T][J % T] : 0;
      (i / G == J + 1 ? own_u[r] : own_r[r]).push_back(i);
    }
    pn.next_needs_side.assign(n, 0);
    for (uint32_t i = std::min(pn.c1 + G, nblk); i < nblk; ++i) {
      if (n <= 1) break;
      const uint32_t upd = map.tbl[(i / G) % T][(J + 1) % T], src = map.tbl[(i / G) % T][J % T];
      if (upd != src) pn.next_needs_side[upd] = 1;
    }
    pn.own_u_first.assign(n, 0); pn.own_u_count.assign(n, 0);
    pn.own_r_first.assign(n, 0); pn.own_r_count.assign(n, 0);
    for (uint32_t r = 0; r < n; ++r) {
      pn.own_u_first[r] = (uint32_t)p.tiles.size(); pn.own_u_count[r] = (uint32_t)own_u[r].size();
      p.tiles.insert(p.tiles.end(), own_u[r].begin(), own_u[r].end());
      pn.own_r_first[r] = (uint32_t)p.tiles.size(); pn.own_r_count[r] = (uint32_t)own_r[r].size();
      p.tiles.insert(p.tiles.end(), own_r[r].begin(), own_r[r].end());
    }
    // messages: the tiles of a sender grouped by the set of ranks that need their class
    for (int urgent = 1; urgent >= 0; --urgent)
      for (uint32_t s = 0; s < n; ++s) {
        const std::vector<uint32_t>& mine = urgent ? own_u[s] : own_r[s];
        if (mine.empty()) continue;
        std::map<std::vector<uint32_t>, std::vector<uint32_t>> groups;  // receivers -> tiles
        for (uint32_t i : mine) {
          const uint32_t a = (i / G) % T;
          std::vector<uint32_t> dst;
          for (uint32_t q = 0; q < n; ++q)
            if ((q != s || self_messages) && p.need(q, a)) dst.push_back(q);
          if (!dst.empty()) groups[dst].push_back(i);
        }
        for (auto& g : groups) {
          DistMsg m;
          m.panel = J; m.src = s; m.urgent = urgent != 0;
          m.first = (uint32_t)p.tiles.size(); m.count = (uint32_t)g.second.size();
          p.tiles.insert(p.tiles.end(), g.second.begin(), g.second.end());
          m.dst = g.first;
          pn.msgs.push_back((uint32_t)p.msgs.size());
          p.msgs.push_back(std::move(m));
        }
      }
  }
  return p;
}

// doubles of the square message of a panel of w tile columns: (w*64 + 1) x (w*64) factor entries (square +
// rhs row), the pivot signs, the per-tile-column sign flags, the factor packets (NOPV = 40 vectors of 64)
inline size_t dist_square_doubles(uint32_t w_tiles) {
  const size_t w = (size_t)w_tiles * 64;
  return (w + 1) * w + w + w_tiles + (size_t)w_tiles * 40 * 64;
}

struct DistPlanStats {
  double factor_bytes;            // all of L below the squares + the squares (what the 1-D panel broadcast moves)
  double chain_recv_max, chain_recv_total;   // per-rank maximum / sum over ranks, bytes per factorisation
  double side_recv_max, side_recv_total;
  double chain_sent_total, side_sent_total;  // every (message, receiver) pair once
  double recv_max;                           // chain + side of the busiest receiver
  double backward_allreduce_bytes;           // nb all-reduces of one panel width
  uint32_t messages_chain, messages_side, panels, ranks, classes;
};

inline DistPlanStats dist_plan_stats(const DistPlan& p) {
  DistPlanStats s = {};
  const uint32_t n = std::max(p.map.n, 1u);
  std::vector<double> chain(n, 0.0), side(n, 0.0);
  s.panels = p.nb; s.ranks = n; s.classes = p.map.T;
  for (uint32_t J = 0; J < p.nb; ++J) {
    const DistPanel& pn = p.panels[J];
    const uint32_t wt = pn.c1 - pn.c0;
    const double sq = 8.0 * (double)dist_square_doubles(wt);
    s.factor_bytes += sq;
    for (uint32_t r = 0; r < n; ++r) {
      if (r != pn.diag_owner) { chain[r] += sq; s.chain_sent_total += sq; }
      s.factor_bytes += 8.0 * 64.0 * 64.0 * wt * (double)(pn.own_u_count[r] + pn.own_r_count[r]);
    }
    s.backward_allreduce_bytes += 8.0 * 64.0 * wt;
    for (uint32_t mi : pn.msgs) {
      const DistMsg& m = p.msgs[mi];
      const double b = 8.0 * 64.0 * 64.0 * wt * (double)m.count;
      for (uint32_t q : m.dst) {
        if (q == m.src) continue;
        (m.urgent ? chain : side)[q] += b;
        (m.urgent ? s.chain_sent_total : s.side_sent_total) += b;
      }
      (m.urgent ? s.messages_chain : s.messages_side) += 1;
    }
  }
  for (uint32_t r = 0; r < n; ++r) {
    s.chain_recv_max = std::max(s.chain_recv_max, chain[r]); s.chain_recv_total += chain[r];
    s.side_recv_max = std::max(s.side_recv_max, side[r]); s.side_recv_total += side[r];
    s.recv_max = std::max(s.recv_max, chain[r] + side[r]);
  }
  return s;
}

}  // namespace bae
