"""Time MODEL (not a measurement) of the distributed reduced solve of DESIGN.md §6a on N GPUs: the four streams of
cholesky_solve_dist and their event dependencies, replayed per rank and per panel on the committed tile pattern of
configs[3], with rates measured on ONE MI355X and ASSUMED link figures.  It exists to compare layouts and to show which
term bounds the factorisation; it claims no scaling curve.

    python scratch/dist_time_model.py [link_GBs] [p2p_latency_us]

Measured inputs (profiles/, one GPU): bulk trailing updates 65 TFLOP/s when alone on the device; the panel chain
(substitution + in-panel updates) 92 ms for the whole matrix = 3.35 us per (row tile, panel); a 1024-column square chain
0.45 ms; rectangle updates 60 TFLOP/s.  Assumed: every pair of GPUs has its own xGMI link (full mesh) carrying
`link_GBs` one way under RCCL point-to-point; a broadcast of 9 MB costs one link time + latency (tree over the mesh);
each message costs `p2p_latency_us`."""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINK = float(sys.argv[1]) * 1e9 if len(sys.argv) > 1 else 50e9
LAT = float(sys.argv[2]) * 1e-6 if len(sys.argv) > 2 else 30e-6
BULK_RATE, RECT_RATE = 65e12, 60e12
CHAIN_PER_ROWTILE = 92e-3 / (938 * 939 / 2 / 16)     # s per (row tile, panel of 16 tile columns)
SQUARE = 0.45e-3
G, NB = 16, 64
d = np.load(os.path.join(ROOT, "tests", "golden", "config3_factor_tile_pattern.npz"))
nblk = int(d["nblk"])
nz = np.unpackbits(d["bits"])[:nblk * nblk].reshape(nblk, nblk).astype(bool)
nb = (nblk + G - 1) // G
blk = np.arange(nblk) // G


def table(layout, N):
    if layout == "tri":
        T = {2: 2, 8: 4, 18: 6}[N]
        t = np.zeros((T, T), int); r = 0
        for a in range(T):
            for b in range(a + 1, T):
                t[a, b] = t[b, a] = r; r += 1
        for k in range(T // 2):
            t[2 * k, 2 * k] = t[2 * k + 1, 2 * k + 1] = r + k
        return T, t
    if layout == "row":
        return N, np.repeat(np.arange(N)[:, None], N, 1)
    if layout == "col":
        return N, np.repeat(np.arange(N)[None, :], N, 0)
    raise ValueError(layout)


def model(layout, N):
    T, tbl = table(layout, N)
    own = lambda bi, bc: tbl[bi % T, bc % T]
    needs = np.zeros((N, T), bool)
    for a in range(T):
        for b in range(T):
            needs[tbl[a, b], a] = needs[tbl[a, b], b] = True
    # per panel: row tiles with a nonzero tile in the panel, by block row
    s0 = np.zeros(N); s1 = np.zeros(N); s2 = np.zeros(N); s3 = np.zeros(N)   # stream clocks per rank
    ev_next = np.zeros(N); ev_bulk = np.zeros(N); ev_bulk_prev = np.zeros(N)
    recv_bytes = np.zeros(N)
    for J in range(nb):
        c0, c1 = J * G, min((J + 1) * G, nblk)
        w = (c1 - c0) * NB
        rows = np.nonzero(nz[:, c0:c1].any(1) & (np.arange(nblk) >= c1))[0]
        dJ = own(J, J)
        # chain stream: the square on its owner (after its own step J-1 work), broadcast to everybody
        t_sq = max(s0[dJ], ev_bulk_prev[dJ]) + SQUARE
        t_b = t_sq + LAT + 8.0 * dist_sq(w) / LINK
        s0 = np.maximum(s0, t_b)
        if c1 >= nblk:
            break
        # urgent block row J+1
        u = rows[blk[rows] == J + 1]
        ou = own(J + 1, J)
        t_u = max(s0[ou], ev_next[ou]) + len(u) * CHAIN_PER_ROWTILE
        urecv = [q for q in range(N) if needs[q, (J + 1) % T] and q != ou]
        t_ux = t_u + (LAT + len(u) * NB * w * 8.0 / LINK if urecv else 0.0)
        s0[ou] = t_ux
        for q in urecv:
            s0[q] = max(s0[q], t_ux); recv_bytes[q] += len(u) * NB * w * 8.0
        ev_urg = s0.copy()
        # square update of panel J+1 on its owner
        dn = own(J + 1, J + 1)
        s0[dn] = max(s0[dn], ev_bulk_prev[dn]) + 2.0 * (G * NB) ** 2 * w / RECT_RATE
        # panel stream: the other own rows, their side messages
        rest = rows[blk[rows] > J + 1]
        owner = np.array([own(b, J) for b in blk[rest]]) if len(rest) else np.zeros(0, int)
        t_pack = np.zeros(N)
        for r in range(N):
            mine = int((owner == r).sum())
            s1[r] = max(s1[r], ev_urg[r]) + mine * CHAIN_PER_ROWTILE
            t_pack[r] = s1[r]
        t_side = np.zeros(N)
        for q in range(N):
            t_in = s3[q]
            for s in range(N):
                if s == q:
                    continue
                cnt = sum(1 for i, o in zip(rest, owner) if o == s and needs[q, blk[i] % T])
                if cnt:
                    byt = cnt * NB * w * 8.0
                    recv_bytes[q] += byt
                    t_in = max(t_in, max(t_pack[s], s3[q]) + LAT + byt / LINK)   # every sender on its own link
            s3[q] = t_in; t_side[q] = t_in
        # panel stream: panel J applied to the own tiles of column block J+1 (below its square)
        n1 = min(c1 + G, nblk)
        ktiles = nz[:, c0:c1]
        for r in range(N):
            rows_r = [i for i in range(n1, nblk) if own(blk[i], J + 1) == r]
            prod = 0.0
            if rows_r:
                colop = ktiles[c1:n1].astype(np.float64)            # operand tiles of the column block
                prod = float((ktiles[rows_r].astype(np.float64) @ colop.T).sum())
            # (the wait for the side transfer only where a row of it was computed elsewhere: never in the row layout)
            side_dep = any(own(blk[i], J) != r for i in rows_r)
            t = max(s1[r], t_side[r] if side_dep else 0.0, ev_bulk_prev[r]) + prod * 2.0 * NB ** 3 / RECT_RATE
            s1[r] = t; ev_next[r] = t
        # bulk stream
        if n1 < nblk:
            kt = ktiles[n1:].astype(np.float64)
            full = kt @ kt.T                                        # products per tile (i, c), i, c >= n1
            bi = blk[n1:]
            for r in range(N):
                mask = np.array([[own(a, b) == r and a >= b for b in bi] for a in bi]) if False else None
            cls = bi % T
            ownm = tbl[cls[:, None], cls[None, :]]
            low = (np.arange(n1, nblk)[:, None] >= np.arange(n1, nblk)[None, :])
            for r in range(N):
                prod = float((full * ((ownm == r) & low)).sum())
                t = max(s2[r], t_side[r], ev_urg[r]) + prod * 2.0 * NB ** 3 / BULK_RATE
                s2[r] = t
            ev_bulk_prev = ev_bulk.copy(); ev_bulk = s2.copy()
    total = max(s0.max(), s1.max(), s2.max(), s3.max())
    return total, recv_bytes.max()


def dist_sq(w):
    wt = w // NB
    return (w + 1) * w + w + wt + wt * 40 * 64


if __name__ == "__main__":
    print("time MODEL of the distributed factorisation at configs[3] (n = 59 988); link %.0f GB/s one way, %.0f us per message"
          % (LINK / 1e9, LAT * 1e6))
    one = (3.2e13 / BULK_RATE + 92e-3)
    print("one GPU (measured): %.0f ms" % (one * 1e3))
    for N, lay in ((2, "tri"), (2, "row"), (8, "tri"), (8, "row"), (8, "col")):
        t, rb = model(lay, N)
        print("N = %d  %-4s  %6.1f ms   (x%.1f)   busiest receiver %.2f GB" % (N, lay, t * 1e3, one / t, rb / 1e9))
