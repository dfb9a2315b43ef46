"""Diagnostic: per-row differences of the CFO receiver vs the oracle (GPU box)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "lte-gnu-radio-code_amd"))
import numpy as np
from oracle import ofdm_oracle as orc
import test_gpu_fo as T

case, fo_range, cfo_hz, fading = 1, [-30000, -15000, 0, 15000, 30000], 15000.0, False
iq, _ = T._make_input(case, cfo_hz, 4, fading, 7 + case)
o = orc.FoOracle(case, fo_range, py2_rotators=False)
blk = T._block(case, fo_range, py2_rotators=False)
ro, rb = np.zeros(len(iq), np.complex64), np.zeros(len(iq), np.complex64)
o.work(iq, ro)
blk.work([iq], [rb])
a, b = blk.est_data_freq, o.est_data_freq
H = o.est_chan_freq_P
bins = o.bins_used_P
for r in range(26):
    d = np.abs(a[r] - b[r])
    print(r, o.time_synch_ref[r].tolist(), "max|b|=%.3g maxdiff=%.3g at %d  min|H|=%.3g" % (np.abs(b[r]).max(), d.max(), d.argmax(), np.abs(H[r][bins]).min()),
          "guard", o.time_synch_ref[r][0] + 80 + 63 <= len(iq))
print(len(iq), blk.dmax_tmp_ind, o.dmax_tmp_ind)
