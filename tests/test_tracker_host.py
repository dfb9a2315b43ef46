"""Host logic of the regression-tracking receiver (no GPU): `ofdm_mi355x.tracker.SyncPointerTracker` must place every window
where the reference does.  The device primitives are replaced by the oracle's fp64 trial (`TrackerOracle._trial`); the pointer
table is compared entry for entry with `TrackerOracle.work` (itself pinned to recorded reference runs, test_oracle_tracker.py)
over three consecutive calls on the same buffer -- the table persists across calls and feeds the spacing rule."""
import warnings

import numpy as np
import pytest

from oracle import ofdm_oracle as orc
from ofdm_mi355x.tracker import SyncPointerTracker


def _buffer(seed, lead, sigma, n_sym):
    rng = np.random.default_rng(seed)
    n_data = sum(1 for s in range(n_sym) if s % 4 >= 1)
    bits = rng.integers(0, 2, n_data * 120)
    tx = orc.channel_apply(orc.tx_modulate(bits, 64, 16, 62, 60, n_sym, synch_dat=(1, 3), zc_root=23), orc.REF_TAPS, 64)
    tx = tx[:n_sym * 80 + 8] + sigma * (rng.standard_normal(n_sym * 80 + 8) + 1j * rng.standard_normal(n_sym * 80 + 8))
    return np.concatenate([np.zeros(lead), tx, np.zeros(50)]).astype(np.complex64)


def _run_tracker(o, iq, table, rows_sync):
    trk = SyncPointerTracker(o.NFFT, o.len_CP, 4, 0.5 * o.MM, table)

    def trial(window):
        _, _, peak, lag = o._trial(iq, window)
        return float(peak), int(lag)

    accepted = []

    def on_sync(row, window, lag):
        if row >= rows_sync:
            raise IndexError("row %d" % row)
        accepted.append((row, window, lag))

    n_scan = trk.scan_length(len(iq))
    scan = [trial(int(trk.scan_position(i))) for i in range(n_scan)]
    trk.run(len(iq), lambda i: scan[i], trial, on_sync)
    return trk, accepted


@pytest.mark.parametrize("seed,lead,sigma,n_sym", [(1, 0, 0.0, 48), (2, 13, 0.05, 40), (3, 29, 0.01, 48), (4, 5, 0.0, 20),
                                                   (5, 37, 0.02, 48), (6, 3, 0.3, 44)])
def test_pointer_table_matches_the_oracle(seed, lead, sigma, n_sym):
    iq = _buffer(seed, lead, sigma, n_sym)
    o = orc.TrackerOracle(0)
    table = np.zeros((250, 3))
    warnings.simplefilter("ignore")
    for _ in range(3):
        o.work(iq, np.zeros(len(iq), np.complex64))
        trk, accepted = _run_tracker(o, iq, table, o.lmax_s)
        assert trk.n_found - 1 == o.corr_obs
        assert np.array_equal(table, o.time_synch_ref[0])
        assert [a[0] for a in accepted] == list(range(trk.n_found))
        # the window handed to the device is the one that was evaluated, not the nudged pointer
        for row, window, lag in accepted:
            assert window <= int(table[row, 0]) and lag == int(table[row, 1])


def test_index_error_leaves_the_reference_state():
    rng = np.random.default_rng(9)
    bits = rng.integers(0, 2, 45 * 120)
    iq = np.concatenate([orc.tx_modulate(bits, 64, 16, 62, 60, 60, synch_dat=(1, 3), zc_root=23), np.zeros(40)]).astype(np.complex64)
    o = orc.TrackerOracle(0)
    warnings.simplefilter("ignore")
    with pytest.raises(IndexError):
        o.work(iq, np.zeros(len(iq), np.complex64))
    table2 = np.zeros((250, 3))
    trk2 = SyncPointerTracker(o.NFFT, o.len_CP, 4, 0.5 * o.MM, table2)

    def trial(window):
        _, _, peak, lag = o._trial(iq, window)
        return float(peak), int(lag)

    def on_sync(row, window, lag):
        if row >= o.lmax_s:
            raise IndexError

    scan = [trial(int(trk2.scan_position(i))) for i in range(trk2.scan_length(len(iq)))]
    with pytest.raises(IndexError):
        trk2.run(len(iq), lambda i: scan[i], trial, on_sync)
    assert trk2.n_found - 1 == o.corr_obs == 12
    assert np.array_equal(table2, o.time_synch_ref[0])
