"""Byte accounting of the distributed reduced solve's message plan (ba_amd/csrc/dist_plan.h through
ba_hip_dist_plan_stats — pure host code of libba_hip.so, no device).  The numbers asserted here are the
ones DESIGN.md §6 quotes: at 8 ranks on the tile pattern of BASELINE configs[3] the chain stream carries a
few percent of the factor and no rank receives more than half of it.  The pattern is the committed fixture
tests/golden/config3_factor_tile_pattern.npz (written on the GPU box by tests/golden/make_config3_pattern.py:
the engine's symbolic elimination of the 10k-pose scene)."""
import os

import numpy as np
import pytest

from ba_amd import hipapi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config3_factor_tile_pattern.npz")


def config3_pattern():
    d = np.load(GOLDEN)
    nblk = int(d["nblk"])
    return nblk, np.unpackbits(d["bits"])[:nblk * nblk].reshape(nblk, nblk).astype(np.uint8)


def test_fixture_is_a_closed_lower_pattern():
    nblk, nz = config3_pattern()
    assert nblk == 938 and nz.shape == (938, 938)
    assert not np.triu(nz, 1).any() and nz.diagonal().all()
    # fill closure of a factor: L(i,k) and L(j,k) nonzero, k < j <= i  =>  L(i,j) nonzero (spot check on a stride)
    for k in range(0, nblk, 37):
        rows = np.nonzero(nz[k + 1:, k])[0] + k + 1
        assert nz[np.ix_(rows, rows)][np.tril_indices(len(rows))].all()
    assert 0.70 < nz.sum() / (nblk * (nblk + 1) / 2) < 0.80   # 0.75 of the lower tiles


def test_config3_at_8_ranks_chain_and_receive_volume():
    """chain-stream bytes <= 15 % of the 10.5 GB the 1-D panel broadcast put on the chain
    stream, per-rank receive <= 50 % of it."""
    nblk, nz = config3_pattern()
    s = hipapi.dist_plan_stats(nblk, nz, 8, "auto")
    assert s["classes"] == 4 and s["panels"] == 59 and s["kout"] == 16     # "tri": 4 classes on 8 ranks
    factor = s["factor_bytes"]
    assert 10.0e9 < factor < 11.5e9
    assert s["chain_recv_max"] <= 0.15 * 10.5e9 and s["chain_recv_max"] <= 0.15 * factor
    assert s["recv_max"] <= 0.50 * 10.5e9 and s["recv_max"] <= 0.50 * factor
    # what the round-2 design did: every rank receives every panel
    col = hipapi.dist_plan_stats(nblk, nz, 8, "col")
    assert col["recv_max"] >= 0.85 * factor
    # block rows that never change hands: the same volume as "col", but none of it between two panels of a row
    row = hipapi.dist_plan_stats(nblk, nz, 8, "row")
    assert row["recv_max"] >= 0.85 * factor and row["chain_recv_max"] <= 0.15 * factor
    # the backward substitution adds one small all-reduce per panel
    assert s["backward_allreduce_bytes"] == 8.0 * 64 * nblk


@pytest.mark.parametrize("nranks,layout,classes", [(2, "auto", 2), (8, "auto", 4), (18, "tri", 6), (3, "auto", 3), (4, "auto", 2),
                                                   (6, "grid", 6), (8, "grid", 4), (8, "row", 8), (8, "col", 8), (1, "auto", 1)])
def test_layouts_conserve_the_factor(nranks, layout, classes):
    """Whatever the layout: every row tile below a square is computed by exactly one rank (factor_bytes does
    not depend on the layout), and nobody receives more than the whole factor."""
    nblk = 120
    i, j = np.indices((nblk, nblk))
    band = (np.abs(i - j) <= 40) | (np.abs(i - j) >= nblk - 10)
    nz = np.tril(band).astype(np.uint8)
    for k in range(nblk):   # symbolic elimination (dense numpy, small)
        rows = np.nonzero(nz[k + 1:, k])[0] + k + 1
        sub = nz[np.ix_(rows, rows)]
        sub[np.tril_indices(len(rows))] = 1
        nz[np.ix_(rows, rows)] = sub
    ref = hipapi.dist_plan_stats(nblk, nz, 1, "auto", 4)
    s = hipapi.dist_plan_stats(nblk, nz, nranks, layout, 4)
    assert s["classes"] == classes and s["ranks"] == nranks
    assert s["factor_bytes"] == ref["factor_bytes"]
    assert s["recv_max"] <= s["factor_bytes"]
    if nranks == 1:
        assert s["chain_recv_total"] == 0 and s["side_recv_total"] == 0
    else:
        assert s["chain_recv_total"] > 0
    # received == sent, stream by stream
    assert s["chain_recv_total"] == s["chain_sent_total"] and s["side_recv_total"] == s["side_sent_total"]


def test_tri_layout_halves_the_receive_volume_only_where_it_exists():
    with pytest.raises(ValueError):
        hipapi.dist_plan_stats(64, None, 4, "tri")        # 4 is not T*T/2
    with pytest.raises(ValueError):
        hipapi.dist_plan_stats(64, None, 8, "diagonal")   # unknown layout
    dense8 = hipapi.dist_plan_stats(256, None, 8, "tri", 8)
    grid8 = hipapi.dist_plan_stats(256, None, 8, "grid", 8)
    col8 = hipapi.dist_plan_stats(256, None, 8, "col", 8)
    assert dense8["recv_max"] < 0.5 * dense8["factor_bytes"] < grid8["recv_max"] < col8["recv_max"]
