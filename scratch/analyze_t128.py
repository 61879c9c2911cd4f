"""Per-block trace of the bulk launches (measurement build -DBAE_TIME128, BA_HIP_TRACE_FILE): how full the CUs are.
Record: start, loop start, loop end, end (100 MHz wall clock), xcc << 32 | HW_ID, blockIdx | cols << 32 | kb0 << 40."""
import sys
import numpy as np

r = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 6)
t0, l0, l1, t1 = (r[:, i].astype(np.int64) for i in range(4))
hw = r[:, 4]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
cu = (hwid >> 8) & 15
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 7
simd = (hwid >> 4) & 3
cols = ((r[:, 5] >> np.uint64(32)) & np.uint64(255)).astype(np.int64)
kb0 = (r[:, 5] >> np.uint64(40)).astype(np.int64)
cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ncu = len(np.unique(cukey))
print("records %d, distinct CUs %d, XCDs %d, launches %d" % (len(r), ncu, len(np.unique(xcc)), len(np.unique(kb0))))
tot_life = tot_loop = tot_span = 0.0
hist_all = np.zeros(4)
rows = []
for k in np.unique(kb0):
    m = kb0 == k
    a, b = t0[m].min(), t1[m].max()
    span = float(b - a)
    life = float((t1[m] - t0[m]).sum())
    loop = float((l1[m] - l0[m]).sum())
    # per-CU concurrency, time-weighted over the launch span
    hist = np.zeros(4)
    idx = np.nonzero(m)[0]
    idx = idx[np.argsort(cukey[idx], kind="stable")]
    bounds = np.nonzero(np.diff(cukey[idx]))[0] + 1
    for g in np.split(idx, bounds):
        tt = np.concatenate([t0[g], t1[g]])
        dd = np.concatenate([np.ones(len(g), dtype=np.int64), -np.ones(len(g), dtype=np.int64)])
        o = np.lexsort((dd, tt))
        tt, dd = tt[o], dd[o]
        lvl = np.concatenate([[0], np.cumsum(dd)])           # level before each event, then after the last
        edges = np.concatenate([[a], tt, [b]])
        dur = np.diff(edges)
        np.add.at(hist, np.minimum(lvl, 3), dur)
    # per-XCD finish spread
    xend = [t1[m & (xcc == x)].max() for x in np.unique(xcc[m])]
    rows.append((k, m.sum(), span / 100.0, life / (2 * ncu * span), loop / life, hist / hist.sum(), (max(xend) - min(xend)) / 100.0,
                 float(np.median((t1[m] - t0[m])[cols[m] == 16])) / 100.0 if (cols[m] == 16).any() else 0.0))
    tot_life += life; tot_loop += loop; tot_span += span; hist_all += hist
print("launch(kb0) blocks   span_us  fill(2 slots/CU)  in-loop  CU time with 0/1/2/3+ blocks   XCD finish spread us  median 16-col block us")
for k, n, span, fill, inl, h, xs, med in rows:
    print("%5d %7d %9.1f  %6.3f  %6.3f   %5.3f %5.3f %5.3f %5.3f   %8.1f  %7.1f" % (k, n, span, fill, inl, h[0], h[1], h[2], h[3], xs, med))
h = hist_all / hist_all.sum()
print("all launches: fill %.3f  in-loop share of block life %.3f  CU time with 0/1/2/3+ resident blocks %.3f %.3f %.3f %.3f"
      % (tot_life / (2 * ncu * tot_span), tot_loop / tot_life, h[0], h[1], h[2], h[3]))
# gaps between launches
ks = np.unique(kb0)
gaps = []
for a_, b_ in zip(ks[:-1], ks[1:]):
    gaps.append((t0[kb0 == b_].min() - t1[kb0 == a_].max()) / 100.0)
if gaps:
    print("gap between the last block of a launch and the first of the next: median %.1f us, max %.1f us, sum %.1f us" % (np.median(gaps), max(gaps), sum(gaps)))
