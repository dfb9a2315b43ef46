#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ .

Run ONLY in the build container (needs /root/reference, which never travels to the GPU box):

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

What it does
 1. Extracts the reference's five data fixtures (.pckl files holding one ndarray each) WITHOUT
    unpickling them: `ofdm_mi355x.safe_pickle` disassembles the opcode stream and rebuilds the array
    from shape/dtype/raw bytes.  -> ref_fixtures.npz
 2. Imports the reference's Python (TX classes, gr-utsa_ofdm RX block, BitRecovery) with in-process
    shims (np.complex/np.product aliases, Agg backend, a stub `gnuradio.gr.sync_block`) and records
    input/output arrays for the hot path.  -> ref_rx_*.npz, ref_tx_*.npz, ref_bitrecovery.npz
Only arrays are written: no reference source text is copied anywhere.
"""
import os
import sys
import types
import importlib.util

import numpy as np

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "lte-gnu-radio-code_amd"))
from ofdm_mi355x.safe_pickle import load_ndarray  # noqa: E402

G = "/root/reference/GNU-Radio-Repositories/"

# ---------------------------------------------------------------- shims
np.complex = complex
np.product = np.prod
_gr = types.ModuleType("gnuradio")
_grgr = types.ModuleType("gnuradio.gr")


class _sync_block:
    def __init__(self, name=None, in_sig=None, out_sig=None):
        pass


_grgr.sync_block = _sync_block
_gr.gr = _grgr
sys.modules["gnuradio"] = _gr
sys.modules["gnuradio.gr"] = _grgr
sys.path += [G + "LEGACY/gr-ofdm-rx/python/txrx_mod"]


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


from OFDM import OFDM  # noqa: E402
from SynchSignal import SynchSignal  # noqa: E402
from MultiAntennaSystem import MultiAntennaSystem  # noqa: E402

REF_RX = _load(G + "gr-utsa_ofdm/python/SynchAndChanEst.py", "ref_utsa_rx").SynchAndChanEst
REF_BR = _load(G + "LEGACY/gr-ofdm-rx/python/BitRecovery.py", "ref_bitrecovery").BitRecovery


def ref_tx(bits, N, cp, Kd, n_sym, channel="Fading"):
    """bits (1, n) -> (grid (n_sym*N), tx iq, rx iq after the reference channel (no noise))."""
    all_bins = np.array(list(range(-Kd // 2, 0)) + list(range(1, Kd // 2 + 1)))
    ofdm = OFDM(cp, Kd, "QPSK", N, 15e3)
    pat = np.tile(np.array([0, 1, 1, 1]), n_sym // 4)
    mas = MultiAntennaSystem(ofdm, 1, "SpMult", all_bins, n_sym, pat, 15e3 * N, "LTE-TU", 0, channel, 1,
                             all_bins, np.array([], dtype=int))
    caz = SynchSignal(cp, N - 2, 1, N, np.array([1, 3]))
    mas.multi_ant_binary_map(caz, bits, np.array([1, 3]))
    mas.multi_ant_symb_gen(n_sym)
    mas.rx_signal_gen()
    return mas.buffer_data_tx[0].copy(), mas.buffer_data_tx_time[0].copy(), mas.buffer_data_rx_time[0].copy()


def ref_rx_run(iq64, n_sym, N, cp, Kd, snr=100, gate=0.7, calls=1):
    blk = REF_RX(n_sym, N, cp, N - 2, [1, 3], Kd, snr, gate, "/tmp/x", "y", 0, 0, "Fading")
    outs = []
    states = []
    for _ in range(calls):
        out = np.zeros(len(iq64), np.complex64)
        blk.work([iq64], [out])
        outs.append(out)
        states.append(dict(tsr=blk.time_synch_ref.copy(), H=blk.est_chan_freq_P[0].copy(),
                           htime=blk.est_chan_time[0].copy(), edf=blk.est_data_freq.copy(),
                           esf=blk.est_synch_freq[0].copy(), eq_gain=np.asarray(blk.eq_gain).copy()))
    return outs, states


def main():
    # ------------------------------------------------------------ 1. reference data fixtures
    D = G + "TEST/GNU_RADIO_OFFLINE/"
    fx = dict(
        tx_bits=load_ndarray(D + "Data/tx_bit_data_chan_type_Fading_SNR_100.pckl"),
        tx_online=load_ndarray(D + "Data/tx_data_online_chan_type_Fading_SNR_100.pckl"),
        tx_offline=load_ndarray(D + "Data/tx_data_offline_chan_type_Fading_SNR_100.pckl"),
        chan_est_tim_ideal=load_ndarray(D + "Output/_output_data.pckl"),
        legacy_tx_data_0=load_ndarray(G + "LEGACY/gr-ofdm-tx/python/tx_data_0.pckl"),
    )
    for k, v in fx.items():
        print("fixture", k, v.dtype, v.shape)
    np.savez_compressed(os.path.join(HERE, "ref_fixtures.npz"), **fx)

    # ------------------------------------------------------------ 2a. reference RX on its own fixtures
    N, cp, Kd, n_sym = 64, 16, 60, 240
    res = {}
    for tag, arr in (("offline", fx["tx_offline"]), ("online", fx["tx_online"])):
        iq64 = arr[0].astype(np.complex64)
        outs, st = ref_rx_run(iq64, n_sym, N, cp, Kd, calls=2)
        res[tag + "_tsr"] = st[0]["tsr"]
        res[tag + "_H"] = st[0]["H"]
        res[tag + "_htime"] = st[0]["htime"]
        res[tag + "_edf"] = st[0]["edf"]
        res[tag + "_esf"] = st[0]["esf"]
        res[tag + "_eq_gain"] = st[0]["eq_gain"]
        res[tag + "_out_call1"] = outs[0]
        res[tag + "_out_call2"] = outs[1]
        res[tag + "_tsr_call2"] = st[1]["tsr"]
        res[tag + "_edf_call2"] = st[1]["edf"]
        print(tag, "tsr", st[0]["tsr"], "call2 tsr", st[1]["tsr"])
    np.savez_compressed(os.path.join(HERE, "ref_rx_fixture64.npz"), **res)

    # ------------------------------------------------------------ 2b. reference TX+RX on synthetic cases
    cases = [
        # tag, N, cp, Kd, n_sym, lead (samples of zeros before the frame), channel, snr, gate
        ("n64_lead5", 64, 16, 60, 8, 5, "Fading", 100, 0.7),
        ("n256", 256, 18, 150, 8, 0, "Fading", 100, 0.7),
        ("n1024_lead3", 1024, 72, 600, 8, 3, "Fading", 100, 0.7),
        ("n2048", 2048, 144, 1200, 8, 0, "Fading", 100, 0.7),
        ("n2048_snr30", 2048, 144, 1200, 8, 0, "Fading", 30, 0.7),
        ("n4096", 4096, 288, 2400, 4, 0, "Fading", 100, 0.7),
    ]
    syn = {}
    rng = np.random.default_rng(20260101)
    for tag, N, cp, Kd, n_sym, lead, chan, snr, gate in cases:
        bits = rng.integers(0, 2, (1, (n_sym // 4) * 3 * Kd * 2)).astype(np.int32)
        grid, tx, rx = ref_tx(bits, N, cp, Kd, n_sym, chan)
        rx = rx[:n_sym * (N + cp) + 2 * cp]          # keep a short channel tail only
        iq = np.concatenate([np.zeros(lead, complex), rx]).astype(np.complex64)
        outs, st = ref_rx_run(iq, n_sym, N, cp, Kd, snr=snr, gate=gate, calls=1)
        s = st[0]
        hb = np.stack([s["edf"].real < 0, s["edf"].imag < 0], -1)
        rows = [r for r in range(n_sym) if r % 4 != 3]
        nerr = int((hb[rows].reshape(-1) != bits[0]).sum())
        print(tag, "tsr", s["tsr"], "bit errors", nerr, "/", bits.size)
        syn[tag + "_cfg"] = np.array([N, cp, Kd, n_sym, lead, snr], dtype=np.float64)
        syn[tag + "_gate"] = np.array([gate])
        syn[tag + "_bits"] = bits
        syn[tag + "_tx"] = tx.astype(np.complex128) if N <= 256 else tx[:2 * (N + cp)].astype(np.complex128)
        syn[tag + "_grid01"] = grid[:2 * N]
        syn[tag + "_iq"] = iq
        syn[tag + "_tsr"] = s["tsr"]
        syn[tag + "_H"] = s["H"]
        syn[tag + "_htime"] = s["htime"]
        syn[tag + "_edf"] = s["edf"]
        syn[tag + "_esf"] = s["esf"]
    np.savez_compressed(os.path.join(HERE, "ref_rx_synth.npz"), **syn)

    # ------------------------------------------------------------ 2c. BitRecovery
    rng = np.random.default_rng(7)
    nb = 4000
    bits = rng.integers(0, 2, 2 * nb)
    pts = np.exp(1j * 2 * np.pi / 8 * np.array([1.0, -1.0, 3.0, 5.0]))
    z = pts[2 * bits[0::2] + bits[1::2]] + 0.25 * (rng.standard_normal(nb) + 1j * rng.standard_normal(nb))
    z[:4] = [0.0 + 0.3j, -0.2 + 0.0j, 0.0 + 0.0j, 0.5 - 0.0j]      # axis/origin ties
    z64 = z.astype(np.complex64)
    captured = {}

    def tracer(frame, event, arg):
        if frame.f_code.co_name == "work":
            def local(fr, ev, a):
                if ev == "return":
                    captured.update({k: np.array(v) for k, v in fr.f_locals.items()
                                     if k in ("hardbit", "softbit0", "softbit1", "dmin", "dminind")})
                return local
            return local
        return None

    br = REF_BR("QPSK", "/tmp/", 0)
    import io
    import contextlib
    sys.settrace(tracer)
    with contextlib.redirect_stdout(io.StringIO()):
        br.work([z64], [None])
    sys.settrace(None)
    print("bitrecovery captured", {k: v.shape for k, v in captured.items()})
    np.savez_compressed(os.path.join(HERE, "ref_bitrecovery.npz"), z=z64, tx_bits=bits.astype(np.int32),
                        hardbit=captured["hardbit"].reshape(-1).astype(np.int32),
                        softbit0=captured["softbit0"], softbit1=captured["softbit1"])


if __name__ == "__main__":
    main()
