"""Host-side pointer logic of the regression-tracking receiver, as a small state machine.

Behaviour contract: `OFDMReceiver.SynchronizeAndEstimate.work`, LEGACY/gr-ofdm-rx/python/SynchronizeAndEstimate.py:209-343
(the recorded reference runs in tests/golden/ref_tracker.npz pin every pointer and lag it produces).

The receiver looks for one sync symbol per [S, D] pattern.  Where it looks next depends on how many it has found:

  ACQUIRE  (nothing found yet)   a strided scan over the buffer; positions are known up front, so the block evaluates them
                                 in ONE batched device launch and hands the table in as `scan`.
  COAST    (syncs 1..5)          the previous position plus one pattern length.
  PREDICT  (from the 6th sync)   a straight line fitted through the last HISTORY (pattern time, position + lag) observations,
                                 evaluated one pattern ahead, minus cp/4, rounded up.

Everything numeric that decides a pointer is evaluated with the same NumPy routines and operand shapes the reference uses
(`np.linalg.lstsq(..., rcond=-1)`, a (1,2)x(2,) `np.dot`, `np.ceil`): the predicted pointer goes through a ceil, so a closed
form that differs in the last bit could move a window by one sample.
"""
from __future__ import annotations

import numpy as np


class SyncPointerTracker:
    HISTORY = 5          # observations the line is fitted through (SE:325 `tap_delay`)
    COAST_SYNCS = 5      # syncs placed by the fixed advance before the line takes over (SE:234)

    def __init__(self, nfft: int, cp: int, pattern_len: int, gate: float, table: np.ndarray):
        """`table`: the block's persistent [rows][3] (position, lag, peak) array -- rows written by an EARLIER call are
        still read by the spacing rule of this call (SE:311), so it is shared, not owned."""
        self.nfft, self.cp, self.pattern_len, self.gate = nfft, cp, pattern_len, gate
        self.table = table
        self.scan_step = np.ceil(cp / 2)                  # SE:209
        self.scan_origin = (cp - 4) - 1                   # SE:219
        self.late_lag = np.ceil(0.75 * cp)                # SE:281
        self.nudge = np.ceil(0.5 * cp)                    # SE:283,297
        self.min_spacing = 2 * cp + nfft                  # SE:313
        self.n_found = 0                                  # == corr_obs + 1

    # ---- acquisition scan geometry
    def scan_steps(self, n_in: int) -> int:
        return int(np.ceil(n_in / self.scan_step))        # SE:222 (the loop runs steps 0..this inclusive)

    def scan_position(self, step: int):
        return step * self.scan_step + self.scan_origin

    def scan_length(self, n_in: int) -> int:
        """Leading scan positions whose window lies inside the buffer (SE:240); the only ones that can be evaluated."""
        n = 0
        while n <= self.scan_steps(n_in) and self.nfft + self.scan_position(n) < n_in:
            n += 1
        return n

    # ---- the loop
    def run(self, n_in: int, scan, probe, on_sync):
        """scan(step) -> (peak, lag) of acquisition window `step`; probe(window) -> (peak, lag) of one tracked window;
        on_sync(row, window, lag) is called for every accepted sync (row = its index in `table`), after `table[row]` and
        `n_found` have been updated -- it may raise (the reference's IndexError past the estimate rows) and the state stays
        as the reference leaves it."""
        H = self.HISTORY
        when = np.zeros(H)            # pattern time of the kept observations (ring, SE:323)
        where = np.zeros(H)           # their position + lag                   (ring, SE:321)
        line = 0                      # (intercept, slope) once fitted
        ahead = []                    # [[1, pattern time of the NEXT sync]]
        pos = 0
        self.n_found = 0
        for step in range(self.scan_steps(n_in) + 1):
            tracking = self.n_found > 0
            if not tracking:
                pos = self.scan_position(step)
            elif self.n_found <= self.COAST_SYNCS:
                pos = pos + self.pattern_len * (self.nfft + self.cp)
            else:
                pos = (np.ceil(np.dot(ahead, line) - self.cp / 4))[0]                       # SE:237
            if not (self.nfft + pos < n_in):                                                # SE:240
                continue
            window = int(pos)
            if tracking and window < 0:
                raise IndexError("window pointer %d before the buffer" % window)
            peak, lag = probe(window) if tracking else scan(step)
            lag = lag - 1                                                                   # SE:275
            if not (peak > self.gate or tracking):                                          # SE:279
                continue
            if lag > self.late_lag and self.n_found <= self.COAST_SYNCS:
                pos = pos + self.nudge        # the pointer moves, the window that was evaluated does not (SE:281-309)
            anchor = self.table[max(self.n_found - 1, 1), 0]                                # SE:311
            if tracking and not (pos - anchor > self.min_spacing):                          # SE:313
                continue
            row = self.n_found
            self.n_found = row + 1
            self.table[row] = [pos, lag, peak]                                              # SE:316-318
            when[row % H] = row * self.pattern_len
            where[row % H] = self.table[row, 0] + self.table[row, 1]
            kept = min(row, H)                 # observations the fit may use: all EARLIER ones, at most HISTORY (SE:326,336)
            ahead = np.zeros((kept + 1, 2))[-1:]
            ahead[0, 0] = 1
            ahead[0, 1] = (row + 1) * self.pattern_len
            if row > 3:                                                                     # SE:333-341
                A = np.zeros((kept, 2))
                A[:, 0] = 1
                A[:, 1] = when[0:kept]
                line = np.linalg.lstsq(A, where[0:kept], rcond=-1)[0]
            on_sync(row, window, lag)
