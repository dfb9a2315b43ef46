"""Drop-in GNU Radio blocks: same class names, constructor orders and work() contract as the reference.

  utsa_ofdm.SynchAndChanEst      gr-utsa_ofdm/python/SynchAndChanEst.py:16-262
  utsa_ofdm.TxSignalTransmitter  gr-utsa_ofdm/python/TxSignalTransmitter.py:15-29
  RXOFDM.synch_and_chan_est      gr-RXOFDM/python/synch_and_chan_est.py:16-266  (ctor/constants only, see below)
  TXOFDM.tx_signal_transmitter   gr-TXOFDM/python/tx_signal_transmitter.py:13-27
  OFDMReceiver.BitRecovery       LEGACY/gr-ofdm-rx/python/BitRecovery.py:32-189

All signal processing happens in libofdm_mi355x.so (HIP kernels); these classes only marshal NumPy views.
"""
from __future__ import annotations

import os
import pickle

import numpy as np

from . import _lib
from .engine import DeviceBuffer, FoEngine, RxEngine, TrkEngine, bins_p, zadoff_chu
from .gr_compat import sync_block
from .safe_pickle import load_ndarray
from .tracker import SyncPointerTracker


def _device() -> int:
    return int(os.environ.get("OFDM_MI355X_DEVICE", "0"))


class SynchAndChanEst(sync_block):
    """utsa_ofdm.SynchAndChanEst -- ZC timing sync + LS channel estimate + FFT/equalise (stream block)."""

    _COMPAT = _lib.COMPAT_UTSA
    _NAME = "SynchAndChanEst"

    def __init__(self, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr,
                 scale_factor_gate, directory_name, file_name_cest, diagnostics, genie, channel="Fading"):
        # GRC's make template passes 12 arguments (grc/utsa_ofdm_SynchAndChanEst.block.yml:7): `channel` defaults.
        sync_block.__init__(self, name=self._NAME, in_sig=[np.complex64], out_sig=[np.complex64])
        self.num_ofdm_symb = num_ofdm_symb
        self.nfft = nfft
        self.channel = channel
        self.cp_len = cp_len
        self.genie = genie
        self.scale_factor_gate = scale_factor_gate
        self.num_synch_bins = num_synch_bins
        self.synch_dat = synch_dat
        self.num_data_bins = num_data_bins
        self.num_data_symbs_blk = synch_dat[1]
        self.synch_bins_used_P = list(bins_p(num_synch_bins, nfft))            # :38-41
        self.bins_used_P = list(bins_p(num_data_bins, nfft))                    # :66-70
        self.L_synch = len(self.synch_bins_used_P)
        self.M = [synch_dat[0], num_synch_bins]                                 # :48
        self.MM = int(np.prod(self.M))
        self.SNR = snr
        if self._COMPAT == _lib.COMPAT_UTSA:
            self.p = 23                                                         # :52
            self.stride_val = 1                                                 # :77
            self.SNR_lin = 10 ** (snr / 20)                                     # :99
            self.zadoff_chu = zadoff_chu(self.MM, self.p)
        else:
            self.p = 37                                                         # gr-RXOFDM :54
            self.stride_val = cp_len - 1                                        # :81
            self.SNR_lin = snr
            self.zadoff_chu = zadoff_chu(self.MM, self.p, parity_of=num_synch_bins)
        self.start_samp = cp_len                                                # :78
        self.rx_b_len = nfft + cp_len                                           # :79
        self.diagnostic = diagnostics
        self.directory_name = directory_name
        self.file_name_cest = file_name_cest
        self.num_ant_txrx = 1
        self._engine = RxEngine(num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr,
                                scale_factor_gate, compat=self._COMPAT, device=_device())
        self.time_synch_ref = np.zeros(3)                                       # :83
        self.count = 0                                                          # :100
        self.corr_obs = -1                                                      # :72

    # -- attributes the reference keeps as NumPy arrays; here they live in HBM and are fetched on access
    def _rows(self, key, width):
        out = np.zeros((self.num_ofdm_symb, width), dtype=complex)
        for row in range(min(2, self.num_ofdm_symb)):
            out[row] = self._engine.state(row)[key]
        return out

    @property
    def est_chan_freq_P(self):
        return self._rows("chan_freq", self.nfft)

    @property
    def est_chan_time(self):
        return self._rows("chan_time", self.nfft)

    @property
    def est_synch_freq(self):
        return self._rows("synch_freq", self.MM)

    @property
    def est_data_freq(self):
        return self._engine.state(0)["data_freq"].astype(complex)

    @property
    def eq_gain(self):
        return self._engine.state(0)["eq_gain"].astype(complex)

    def work(self, input_items, output_items):
        in0 = input_items[0]
        out = output_items[0]
        rc = self._engine.work(in0, out)
        rep = self._engine.report
        # the reference updates these before any later IndexError / ValueError of the same call
        self.time_synch_ref = np.array(rep.time_synch_ref[:], dtype=float)
        self.count = rep.count
        self.corr_obs = rep.corr_obs
        n = _lib.check(rc)
        if self.diagnostic == 1 and rep.detected:                               # :204-210 channel-estimate dump
            row = 0 if self.count == 1 else min(1, self.num_ofdm_symb - 1)
            chan_est_tim = self._engine.state(row)["chan_time"].astype(complex)[np.newaxis, :]
            with open(str(self.directory_name) + "_" + str(self.file_name_cest), "wb") as f:
                pickle.dump(chan_est_tim, f, protocol=2)
        return n


class synch_and_chan_est(SynchAndChanEst):  # noqa: N801  (reference class name)
    """RXOFDM.synch_and_chan_est -- the class `ofdm_chain.py:83` instantiates.

    The reference implementation of this generation raises AttributeError/TypeError on its first
    detection under Python 3 (gr-RXOFDM/python/synch_and_chan_est.py:194,253); what is kept is its
    constructor signature and constants (ZC root 37, search stride cp-1, gate 0.4, linear snr) on the
    gr-utsa_ofdm control flow.  ``table_mode=True`` selects the control flow its own work() spells out instead
    (sync table, one data symbol per sync): see `synch_and_chan_est_table`.
    """

    _COMPAT = _lib.COMPAT_RXOFDM

    def __new__(cls, *args, table_mode=False, **kw):
        if table_mode and cls is synch_and_chan_est:
            return synch_and_chan_est_table(*args, **kw)        # another class: __init__ below is not run again
        return super().__new__(cls)

    def __init__(self, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr,
                 directory_name, file_name_cest, diagnostics, genie, table_mode=False):
        SynchAndChanEst.__init__(self, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr,
                                 0.4, directory_name, file_name_cest, diagnostics, genie)
        self.diagnostics = diagnostics


# numerology of SynchEstAndFO's `case` argument (LEGACY/gr-ofdm-rx/python/SynchEstAndFO.py:36-135):
# case -> (num_ofdm_symb, fs, nfft, synch_dat, num_data_bins); cp_len = nfft/4, num_synch_bins = nfft-2, SNR = 1e8
_FO_CASES = {
    0: (48, 960000, 64, [1, 1], 12), 1: (48, 960000, 64, [1, 1], 36), 2: (48, 960000, 64, [1, 1], 48),
    3: (48, 960000, 64, [2, 1], 48), 4: (48, 960000, 64, [3, 1], 24), 5: (48, 960000, 64, [2, 1], 24),
    6: (24, 1920000, 128, [3, 1], 24), 7: (24, 1920000, 128, [5, 1], 100), 8: (12, 3840000, 256, [5, 1], 36),
    9: (12, 3840000, 256, [2, 1], 180),
}


class _SyncTableBlock(sync_block):
    """Shared body of the legacy-generation receivers that keep a TABLE of up to 100 syncs per work() call and demodulate
    one data symbol per sync (SynchEstAndFO.py:232-369 == gr-RXOFDM synch_and_chan_est.py:136-266 minus the rotators)."""

    def _configure(self, name, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr, rotators,
                   directory_name, file_name_cest, diagnostics, spread_code=None):
        sync_block.__init__(self, name=name, in_sig=[np.complex64], out_sig=[np.complex64])
        self.num_ofdm_symb, self.nfft, self.cp_len = num_ofdm_symb, nfft, cp_len
        self.num_synch_bins, self.num_data_bins = num_synch_bins, num_data_bins
        self.synch_dat = list(synch_dat)
        self.SNR = snr
        self.synch_bins_used_P = list(bins_p(num_synch_bins, nfft))             # FO:155-158 / RXc:42-45
        self.bins_used_P = list(bins_p(num_data_bins, nfft))                    # FO:185-187 / RXc:70-72
        self.L_synch = len(self.synch_bins_used_P)
        self.M = [self.synch_dat[0], num_synch_bins]
        self.MM = int(np.prod(self.M))
        self.p = 37                                                             # FO:167 / RXc:54
        self.zadoff_chu = zadoff_chu(self.MM, self.p, parity_of=num_synch_bins)
        self.stride_val = cp_len - 1                                            # FO:196 / RXc:81
        self.start_samp = cp_len
        self.rx_b_len = nfft + cp_len
        self.max_num_corr = _lib.FO_MAX_SYNC                                    # FO:200 / RXc:85
        self.cor_obs = -1
        self.count = 0
        self.directory_name = directory_name
        self.file_name_cest = file_name_cest
        self.diagnostics = diagnostics
        self._engine = FoEngine(num_ofdm_symb, nfft, cp_len, num_synch_bins, self.synch_dat, num_data_bins, snr, rotators,
                                device=_device(), spread_code=spread_code)
        self._n_sync = 0

    def _state(self, key):
        return self._engine.state()[key]

    @property
    def time_synch_ref(self):
        return self._state("time_synch_ref")

    @property
    def est_chan_freq_P(self):
        return self._state("chan_freq").astype(complex)

    @property
    def est_chan_time(self):
        return self._state("chan_time").astype(complex)

    @property
    def est_synch_freq(self):
        return self._state("synch_freq").astype(complex)

    @property
    def est_data_freq(self):
        return self._state("data_freq").astype(complex)

    @property
    def eq_gain(self):
        return self._state("eq_gain").astype(complex)

    def _dump_name(self, date_time):
        return str(self.directory_name) + str(self.file_name_cest) + date_time + '.pckl'         # FO:312

    def work(self, input_items, output_items):
        in0 = input_items[0]
        out = output_items[0]
        rc = self._engine.work(in0, out)
        rep = self._engine.report
        self.count = rep.count
        self._n_sync = rep.n_sync
        if rep.dmax_tmp_ind >= 0:
            self.dmax_tmp_ind = rep.dmax_tmp_ind                                # FO:283 (absent until a trial has run)
        if rc >= 0:
            self.cor_obs = 0                                                    # FO:369
        n = _lib.check(rc)
        if self.diagnostics == 1 and rep.n_sync > 0:                            # FO:308-314: dump of the latest estimate
            import datetime
            chan_est_tim = self.est_chan_time[rep.n_sync - 1][np.newaxis, :]
            date_time = datetime.datetime.now().strftime('%Y_%m_%d_%Hh_%Mm')
            with open(self._dump_name(date_time), 'wb') as f:
                pickle.dump(chan_est_tim, f, protocol=2)
        return n


class synch_and_chan_est_table(_SyncTableBlock):  # noqa: N801
    """gr-RXOFDM's synch_and_chan_est as its work() is written (synch_and_chan_est.py:136-266): every gate-passing trial
    that is far enough from the previous sync enters a 100-row table, each sync is followed by ONE equalised data symbol,
    and the first num_ofdm_symb/(S+D) rows are emitted from the second call on.  (As shipped the reference class dies on its
    first detection -- :194 reads `self.diagnostic`, which the constructor never sets; the goldens were recorded with that
    attribute supplied.)  Selected with ``RXOFDM.synch_and_chan_est(..., table_mode=True)``."""

    def __init__(self, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr,
                 directory_name, file_name_cest, diagnostics, genie):
        self.genie = genie
        self._configure("SynchAndChanEst", num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, snr, None,
                        directory_name, file_name_cest, diagnostics)

    def _dump_name(self, date_time):
        return str(self.directory_name) + str(date_time) + str(self.file_name_cest) + '.pckl'    # RXc:206-207


class LegacySynchAndChanEst(synch_and_chan_est_table):
    """OFDMReceiver.SynchAndChanEst(num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, SNR, directory_name,
    file_name_cest, diagnostics) -- LEGACY/gr-ofdm-rx/python/SynchAndChanEst.py:28-243: the same work() as gr-RXOFDM's
    block (sync table, one data symbol per sync) without its broken genie branch; it runs unmodified under Python 2 and is
    bit-identical to the recorded gr-RXOFDM runs (tests/golden/gen_golden_rxofdm_table.py)."""

    def __init__(self, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, SNR, directory_name,
                 file_name_cest, diagnostics):
        synch_and_chan_est_table.__init__(self, num_ofdm_symb, nfft, cp_len, num_synch_bins, synch_dat, num_data_bins, SNR,
                                          directory_name, file_name_cest, diagnostics, 0)


class SynchEstAndFO(_SyncTableBlock):
    """OFDMReceiver.SynchEstAndFO(case, fo_range, directory_name, file_name_cest, diagnostics) -- ZC timing sync with a
    brute-force carrier-offset search, a table of up to 100 syncs per call, one equalised data symbol per sync.

    The reference file is Python-2 code (`self.cp_len = self.nfft/4` is used as an index, :39,193): its `/` on ints is a
    floor division, which also makes `(1/self.fs)` in :192 equal 0 -- every candidate rotator is exp(0) = 1 and the
    search always picks index 0.  `py2_rotators=True` (default) keeps exactly that; `py2_rotators=False` builds the
    evidently intended rotators exp(j 2 pi fo n / fs) (no reference output exists for it).
    """

    _CASES = _FO_CASES
    _NAME = "SynchEstAndFO"

    def __init__(self, case, fo_range, directory_name, file_name_cest, diagnostics, py2_rotators=True):
        self.case = case
        if case not in self._CASES:
            # :136 prints "Error: Case Out of Bounds" and then fails on the first missing attribute
            print("Error: Case Out of Bounds")
            raise AttributeError("'%s' object has no attribute 'num_synch_bins'" % self._NAME)
        num_ofdm_symb, self.fs, nfft, sd, num_data_bins = self._CASES[case][:5]
        self.fo_range = fo_range
        inv_fs = (1 // self.fs) if py2_rotators else (1.0 / self.fs)
        self.cfo = np.exp(1j * 2 * np.pi * inv_fs * np.outer(list(fo_range), np.arange(nfft)))   # :192
        self.p = 37
        self._configure(self._NAME, num_ofdm_symb, nfft, nfft // 4, nfft - 2, sd, num_data_bins, 100000000, self.cfo,
                        directory_name, file_name_cest, diagnostics, spread_code=self._spread_code())

    def _spread_code(self):
        return None


# SynchEstFOAndDSSS's case table (LEGACY/gr-ofdm-rx/python/SynchEstFOAndDSSS.py:37-157): ... + DSSS spreading factor
_DSSS_CASES = {
    0: (48, 960000, 64, [2, 1], 12, 1), 1: (48, 960000, 64, [3, 1], 36, 6), 2: (45, 960000, 64, [4, 1], 48, 6),
    3: (45, 960000, 64, [4, 1], 48, 12), 4: (24, 1920000, 128, [3, 1], 32, 8), 5: (20, 1920000, 128, [4, 1], 84, 12),
    6: (20, 1920000, 128, [4, 1], 96, 16), 7: (24, 1920000, 128, [5, 1], 120, 24), 8: (12, 3840000, 256, [3, 1], 168, 12),
    9: (10, 3840000, 256, [4, 1], 192, 16), 10: (10, 3840000, 256, [4, 1], 240, 24),
}


class SynchEstFOAndDSSS(SynchEstAndFO):
    """OFDMReceiver.SynchEstFOAndDSSS(case, fo_range, directory_name, file_name_cest, diagnostics) -- SynchEstAndFO with its
    own numerology table, followed by despreading of every equalised data symbol over groups of DSSS consecutive bins
    (SynchEstFOAndDSSS.py:391-399).  The despread rows are emitted on every call (the count gate is commented out, :405-407)."""

    _CASES = _DSSS_CASES
    _NAME = "SynchEstFOAndDSSS"

    def _spread_code(self):
        self.DSSS = self._CASES[self.case][5]
        dd = self.DSSS
        t0 = np.arange(dd, dtype=np.float64)
        xx = t0 * t0 if dd % 2 == 0 else t0 * (t0 + 1)                           # :253-259
        self.SC = np.exp((-1j * (2 * np.pi / dd) * self.p / 2.0) * xx)          # :261-262
        return self.SC

    @property
    def est_data_freq_d(self):
        return self._engine.despread().astype(complex)


# SDR_profile of SynchronizeAndEstimate (LEGACY/gr-ofdm-rx/python/SynchronizeAndEstimate.py:33-60), the fields the block uses
_TRACKER_PROFILES = {
    0: dict(system_scenario='4G5GSISO-TU', diagnostic=1, wireless_channel='Fading', channel_band=0.97 * 960e3, bin_spacing=15e3,
            channel_profile='LTE-TU', CP_type='Normal', num_ant_txrx=1, param_est='Estimated', MIMO_method='SpMult', SNR=100,
            ebno_db=[100] * 9, num_symbols=[48] + [1000] * 8, stream_size=1),
    1: dict(system_scenario='WIFIMIMOSM-A', diagnostic=0, wireless_channel='Fading', channel_band=0.9 * 20e6, bin_spacing=312.5e3,
            channel_profile='Indoor A', CP_type='Extended', num_ant_txrx=2, param_est='Ideal', MIMO_method='SpMult', SNR=50,
            ebno_db=[6, 7, 8, 9, 10, 14, 16, 20, 24], num_symbols=[10] * 9, stream_size=2),
}


class SynchronizeAndEstimate(sync_block):
    """OFDMReceiver.SynchronizeAndEstimate(case) -- acquisition by a strided ZC search, then a pointer tracker that follows
    one sync symbol per [1,3] pattern (fixed advance for the first five, a least-squares line through the last five
    (position + lag) observations afterwards), LS channel estimate per sync and three equalised data symbols per sync.

    The scalar pointer logic lives in `tracker.SyncPointerTracker` (an acquire / coast / predict state machine pinned to the
    recorded reference runs); every array computation -- windows, FFTs, lag correlation, channel estimate, equaliser,
    renormalisation -- runs on the GPU through `TrkEngine`.
    """

    def __init__(self, case):
        sync_block.__init__(self, name="SynchronizeAndEstimate", in_sig=[np.complex64], out_sig=[np.complex64])
        self.case = case
        for k, v in _TRACKER_PROFILES[case].items():                              # KeyError for other cases, like :61
            setattr(self, k, v)
        self.synch_data = np.array([1, 3])                                        # :78
        self.NFFT = int(2 ** (np.ceil(np.log2(round(self.channel_band / self.bin_spacing)))))      # :83
        self.fs = self.bin_spacing * self.NFFT
        self.len_CP = int(round(self.NFFT / 4))
        num_bins1 = 4 * np.floor(np.floor(self.channel_band / self.bin_spacing) / 4)                # :87-90
        all_bins = np.array(list(range(-int(num_bins1 / 2), 0)) + list(range(1, int(num_bins1 / 2) + 1)))
        self.num_data_bins = len(all_bins)
        self.used_bins_data = ((self.NFFT + all_bins) % self.NFFT).astype(int)                      # :102
        n_pat = int(np.ceil(self.num_symbols[0] / sum(self.synch_data)))                            # :104
        self.symbol_pattern = np.tile(np.concatenate((np.zeros(1), np.ones(3))), n_pat)
        self._lmax_s, self._lmax_d = n_pat, 3 * n_pat                                               # :143-144
        self.rx_buff_len = self.NFFT + self.len_CP
        self.num_synch_bins = self.NFFT - 2
        self.M = np.array([1, self.num_synch_bins])
        self.MM = int(np.prod(self.M))
        self.prime = 23
        self.ZChu0 = zadoff_chu(self.MM, self.prime)                                                # :123-130
        self.synch_ref = self.ZChu0
        self.used_bins_synch = np.asarray(bins_p(self.num_synch_bins, self.NFFT)).astype(int)       # :134-136
        if list(self.used_bins_data) != list(bins_p(self.num_data_bins, self.NFFT)):
            raise ValueError("data bins are not the symmetric +-K/2 list the device kernels use")
        self.time_synch_ref = np.zeros((self.num_ant_txrx, 250, 3))                                 # :179
        self.corr_obs = None
        self.stride_val = None
        self.start_samp = None
        self._engine = TrkEngine(self.NFFT, self.len_CP, self.num_synch_bins, self.num_data_bins, 3, self._lmax_s,
                                 self._lmax_d if self.num_ant_txrx == 1 else 0, self.SNR, zc_root=self.prime, device=_device())

    # -- arrays kept in HBM, fetched on access (antenna axis of length num_ant_txrx; only antenna 0 is ever written, :215)
    def _ant(self, arr):
        out = np.zeros((self.num_ant_txrx,) + arr.shape, dtype=complex)
        out[0] = arr
        return out

    @property
    def est_chan_freq_p(self):
        return self._ant(self._engine.state()["chan_freq"])

    @property
    def est_chan_freq_n(self):
        return self._ant(self._engine.state()["chan_freq"][:, self.used_bins_synch])

    @property
    def est_chan_impulse(self):
        return self._ant(self._engine.state()["chan_impulse"])

    @property
    def est_synch_freq(self):
        return self._ant(self._engine.state()["synch_freq"])

    @property
    def est_data_freq(self):
        if self.num_ant_txrx != 1:
            raise AttributeError("'SynchronizeAndEstimate' object has no attribute 'est_data_freq'")   # :160-165
        return self._ant(self._engine.state()["data_freq"])

    def work(self, input_items, output_items):
        in0 = input_items[0]
        out = output_items[0]
        eng = self._engine
        n_in = in0.shape[0]
        N = self.NFFT
        eng.load(in0)
        trk = SyncPointerTracker(N, self.len_CP, int(sum(self.synch_data)), 0.5 * self.MM, self.time_synch_ref[0])
        self.stride_val, self.start_samp = trk.scan_step, trk.scan_origin           # :209,219
        # the acquisition windows are independent of each other: one batched launch for the whole scan
        scan_peak, scan_lag = eng.trials(int(trk.scan_origin), int(trk.scan_step), trk.scan_length(n_in))
        seen = {}

        def probe(window):
            if window not in seen:                       # the late-lag branch can ask for the same window twice
                pk, lg = eng.trials(window, 1, 1)
                seen[window] = (float(pk[0]), int(lg[0]))
            return seen[window]

        def accept(row, window, lag):
            # LS estimate on the device (:344-378); a lag of -1 selects the LAST phase column there
            eng.accept(row, window, lag if lag >= 0 else self.len_CP, lag)

        self.corr_obs = -1                                                           # :216
        try:
            trk.run(n_in, lambda i: (float(scan_peak[i]), int(scan_lag[i])), probe, accept)
        finally:
            self.corr_obs = trk.n_found - 1          # also when the reference's IndexError ends the call early
        if self.num_ant_txrx == 1:                                                   # :397
            rows = self.time_synch_ref[0, :trk.n_found]
            ptrs = [int(r[0]) for r in rows]
            guards = [bool(sum(r) + N < n_in) for r in rows]                         # :401 (pointer + lag + PEAK)
            last_row, last = eng.demod(ptrs, guards)
            if last_row >= 0:
                out[0:self.num_data_bins] = last                                     # :438-440
        return len(output_items[0])


def _load_iq_file(path: str) -> np.ndarray:
    """The reference unpickles the file (TxSignalTransmitter.py:22-24); here ndarray pickles are parsed
    without executing them, and .npy files are accepted too."""
    if path.endswith(".npy"):
        arr = np.load(path, allow_pickle=False)
    else:
        arr = load_ndarray(path)
    arr = np.asarray(arr)
    if arr.ndim == 1:
        arr = arr[np.newaxis, :]
    return arr


class TxSignalTransmitter(sync_block):
    """utsa_ofdm.TxSignalTransmitter -- replays a stored IQ buffer into the output stream."""

    def __init__(self, pickle_directory, pickle_file):
        sync_block.__init__(self, name="SimpleTx", in_sig=None, out_sig=[np.complex64])
        self.tx_data = _load_iq_file(str(pickle_directory) + str(pickle_file))

    def work(self, input_items, output_items):
        out = output_items[0]
        out[0:self.tx_data.shape[1]] = self.tx_data[0, :]                       # :28 (complex128 -> complex64)
        return len(output_items[0])                                             # :29


class tx_signal_transmitter(TxSignalTransmitter):  # noqa: N801
    """TXOFDM.tx_signal_transmitter(case, pickle_directory, pickle_file); `case` is ignored as in the reference."""

    def __init__(self, case, pickle_directory, pickle_file):
        TxSignalTransmitter.__init__(self, pickle_directory, pickle_file)
        self.case = case


class BitRecovery(sync_block):
    """OFDMReceiver.BitRecovery -- hard decisions + max-log soft metrics (sink block).  QPSK follows the reference
    literally; "16QAM" / "64QAM" use the same metric on the TS 36.211 maps (extension, no reference code)."""

    def __init__(self, modulation, directory_name, diagnostics):
        sync_block.__init__(self, name="BitRecovery", in_sig=[np.complex64], out_sig=None)
        self.modulation = modulation
        self.directory_name = directory_name
        self.diagnostics = diagnostics
        self.K = 1.414213562373095
        self.count = 0
        self._engine = RxEngine(1, 64, 16, 62, (1, 3), 60, 100, device=_device())
        self.hardbit = None
        self.softbit0 = None
        self.softbit1 = None

    def work(self, input_items, output_items):
        in0 = np.ascontiguousarray(input_items[0], dtype=np.complex64)
        n = in0.size
        bps = _lib.MODULATION_BITS.get(str(self.modulation).upper().replace("-", ""), 2)
        dev = self._engine.cfg.device
        d_sym = DeviceBuffer(max(8, in0.nbytes), dev).upload(in0)
        d_hard = DeviceBuffer(max(8, n * bps), dev)
        soft = bps in (2, 4, 6)
        d_s0 = DeviceBuffer(max(8, n * bps * 4), dev) if soft else None
        d_s1 = DeviceBuffer(max(8, n * bps * 4), dev) if soft else None
        self._engine.demap(d_sym, n, bps, d_hard, d_s0, d_s1)
        _lib.check(self._engine.lib.ofdm_device_synchronize(dev))
        self.hardbit = d_hard.download(np.uint8, n * bps).astype(int)[:, np.newaxis]     # :155-157
        if soft:
            self.softbit0 = d_s0.download(np.float32, n * bps).astype(float)             # :147
            self.softbit1 = d_s1.download(np.float32, n * bps).astype(float)             # :148
        if self.diagnostics == 1 and soft:                                               # :167-184
            import csv
            import datetime
            date_time = datetime.datetime.now().strftime("%Y_%m_%d_%Hh_%Mm")
            for tag, arr in (("softbit0_", self.softbit0), ("softbit1_", self.softbit1)):
                with open(self.directory_name + tag + date_time + ".pckl", "wb") as f:
                    pickle.dump(arr, f, protocol=2)
            with open(self.directory_name + "rx_data.csv", "a") as f:
                csv.writer(f).writerows(self.hardbit)
        self.count += 1                                                                  # :188
        return len(input_items[0])                                                       # :189
