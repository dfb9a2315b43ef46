"""Oracle of the decomposed transmitter (no GPU).  The reference has no code for these stages (block names only,
LEGACY/gr-ofdm-tx/grc/RXtransmit_6.grc:701-975), so the oracle's stage functions are pinned to what the reference DOES hold:
chained without pilots they must be `tx_modulate` -- itself pinned to the reference's `tx_data_online` fixture -- and
reproduce that fixture from the reference's bit fixture.  The counter-based bit source is pinned to the published
Philox4x32-10 known-answer vectors (Random123 kat_vectors)."""
import numpy as np
import pytest

from oracle import ofdm_oracle as orc


def _chain(bits, N, cp, Ks, Kd, mod="QPSK", root=23, every=3):
    sym = orc.map_bits(bits, mod)
    rows = orc.tx_stage_cp(orc.tx_stage_ifft(orc.tx_stage_grid(sym, N, Kd)), cp)
    return orc.tx_stage_mux(rows, N, cp, root, every, Ks).ravel()


def test_chain_reproduces_the_reference_fixture(golden):
    fx = golden("ref_fixtures.npz")
    got = _chain(fx["tx_bits"][0], 64, 16, 62, 60)
    assert got.shape == fx["tx_online"][0].shape
    assert np.max(np.abs(got - fx["tx_online"][0])) < 1e-12


@pytest.mark.parametrize("N,cp,Kd,mod,n_sym", [(64, 16, 60, "QPSK", 24), (256, 64, 180, "16QAM", 8), (2048, 144, 1200, "64QAM", 4),
                                               (128, 32, 100, "BPSK", 10)])
def test_chain_is_tx_modulate(N, cp, Kd, mod, n_sym):
    rng = np.random.default_rng(N)
    n_data = sum(1 for s in range(n_sym) if s % 4 >= 1)
    bits = rng.integers(0, 2, n_data * Kd * orc.BITS_PER_SYMBOL[mod])
    ref = orc.tx_modulate(bits, N, cp, N - 2, Kd, n_sym, modulation=mod)
    assert np.array_equal(_chain(bits, N, cp, N - 2, Kd, mod), ref)


def test_partial_last_group_keeps_its_sync_symbol():
    rows = np.arange(5 * 80).reshape(5, 80).astype(complex)
    out = orc.tx_stage_mux(rows, 64, 16, 47, 3, 62)
    assert out.shape == (7, 80)
    assert np.array_equal(out[0], out[4]) and np.array_equal(out[1:4], rows[:3]) and np.array_equal(out[5:], rows[3:])


def test_pilots_sit_on_their_offsets_and_data_keep_list_order():
    sym = np.arange(1, 2 * 48 + 1).reshape(2, 48).astype(complex)
    g = orc.tx_stage_grid(sym, 64, 48, [-21, -7, 7, 21], 1j)
    assert g.shape == (2, 64) and g[0, 0] == 0 and g[0, 32] == 0
    for off in (-21, -7, 7, 21):
        assert g[0, off % 64] == 1j and g[1, off % 64] == 1j
    occ = orc.bins_p(52, 64)
    data = [b for b in occ if b not in {43, 57, 7, 21}]
    assert np.array_equal(g[0, data], sym[0]) and np.array_equal(g[1, data], sym[1])
    unused = sorted(set(range(64)) - set(occ.tolist()))
    assert not g[:, unused].any()


def test_philox_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = orc.philox4x32_10(*[[c] for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


def test_random_bits_are_a_function_of_seed_and_index():
    a = orc.random_bits(20260101, 0, 4096)
    assert set(np.unique(a)) == {0, 1} and abs(a.mean() - 0.5) < 0.05
    b = np.concatenate([orc.random_bits(20260101, 0, 100), orc.random_bits(20260101, 100, 29), orc.random_bits(20260101, 129, 3967)])
    assert np.array_equal(a, b)
    assert not np.array_equal(a, orc.random_bits(20260102, 0, 4096))
    w = orc.philox4x32_10([3], [0], [0], [0], 20260101 & 0xFFFFFFFF, 20260101 >> 32)
    assert a[3 * 128 + 32 + 5] == (int(w[1][0]) >> 5) & 1
