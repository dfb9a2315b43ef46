/* ofdm_mi355x.h -- C ABI of the MI355X-native OFDM TX/RX hot path (libofdm_mi355x.so).
 *
 * Drop-in boundary for tayloreisman16/LTE-GNU-Radio-Code's `ofdm_chain.py` path.  The reference
 * has no native code and no FFI (its blocks are pure Python/NumPy), so every entry point below
 * cites the PYTHON interface it replaces; the GNU Radio blocks in
 * `lte-gnu-radio-code_amd/{utsa_ofdm,RXOFDM,TXOFDM}` bind these symbols with ctypes
 * (INTEGRATION.md shows the stub).  G/ = GNU-Radio-Repositories/ in the reference tree.
 *
 * Conventions
 *   - plain C types only; complex samples are interleaved float32 pairs (numpy complex64,
 *     GNU Radio gr_complex); caller allocates every output; no exceptions cross the boundary.
 *   - every function returns OFDM_OK (0) / a non-negative count, or a negative ofdm_status;
 *     ofdm_last_error() gives the message of the calling thread's last failure.
 *   - `d_` arguments are DEVICE pointers (HBM of the handle's GPU), `h_` arguments are HOST pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the handle's own NON-BLOCKING stream; mind that
 *     the legacy default stream also has handle 0, so it cannot be named here).  Batch entry
 *     points are asynchronous on that stream and allocate nothing.
 *   - a handle owns one HIP stream and its device tables; no global mutable state in the library,
 *     so different block instances (GNU Radio: one thread per block) may run concurrently.
 */
#ifndef OFDM_MI355X_H
#define OFDM_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFDM_ABI_VERSION 1

typedef enum {
    OFDM_OK = 0,
    OFDM_ERR_INVALID = -1,      /* bad argument / unsupported configuration                        */
    OFDM_ERR_HIP = -2,          /* HIP runtime error (no device, launch failure, ...)              */
    OFDM_ERR_INDEX = -3,        /* the reference would raise IndexError (row >= num_ofdm_symb)     */
    OFDM_ERR_SHAPE = -4,        /* the reference would raise ValueError (reshape / assignment)     */
    OFDM_ERR_NOMEM = -5,
    OFDM_ERR_UNBOUND = -6       /* the reference would raise UnboundLocalError (SynchEstFOAndDSSS.py:392) */
} ofdm_status;

typedef enum { OFDM_MOD_BPSK = 1, OFDM_MOD_QPSK = 2, OFDM_MOD_16QAM = 4, OFDM_MOD_64QAM = 6 } ofdm_modulation;

/* Which generation of the reference RX block supplies the constants (SURVEY.md section 2 notes). */
typedef enum {
    OFDM_COMPAT_UTSA = 0,   /* G/gr-utsa_ofdm/python/SynchAndChanEst.py: ZC root 23, stride 1, gate arg, SNR_lin=10^(snr/20) */
    OFDM_COMPAT_RXOFDM = 1  /* G/gr-RXOFDM/python/synch_and_chan_est.py:54,81,170,184: root 37, stride cp-1, gate 0.4, linear snr */
} ofdm_compat;

typedef enum { OFDM_BITS_NONE = 0, OFDM_BITS_PACKED = 1, OFDM_BITS_UNPACKED = 2 } ofdm_bits_mode;

/* ------------------------------------------------------------------------------------------ RX
 * ctor arguments of utsa_ofdm.SynchAndChanEst (G/gr-utsa_ofdm/python/SynchAndChanEst.py:17-19). */
typedef struct {
    int32_t num_ofdm_symb;     /* rows of est_data_freq kept by the stream block (:88)            */
    int32_t nfft;              /* 64,128,...,4096                                                  */
    int32_t cp_len;
    int32_t num_synch_bins;    /* even, <= nfft                                                    */
    int32_t synch_S;           /* synch_dat[0]                                                     */
    int32_t synch_D;           /* synch_dat[1]                                                     */
    int32_t num_data_bins;     /* even, <= nfft                                                    */
    double snr;                /* the block's `snr` argument (dB-like, see :99,:214)               */
    double scale_factor_gate;  /* :166 (ignored for OFDM_COMPAT_RXOFDM: 0.4)                       */
    int32_t compat;            /* ofdm_compat                                                      */
    int32_t modulation;        /* ofdm_modulation used by the fused / standalone de-mapper         */
    int32_t device;            /* HIP device ordinal                                               */
    int32_t reserved;
} ofdm_rx_cfg;

typedef struct ofdm_rx ofdm_rx;

/* what work() leaves in the block's inspectable attributes (:83-91,:173-175,:260-261) */
typedef struct {
    double time_synch_ref[3];  /* [P*stride+cp, argmax lag, int(max|corr|)]                        */
    int32_t detected;          /* a sync trial was accepted during THIS call                        */
    int32_t trials_run;        /* sync trials evaluated during this call                            */
    int32_t count;             /* number of completed work() calls                                  */
    int32_t corr_obs;
    int64_t n_data_items;      /* n_data_symb * num_data_bins values packed at the head of `out`    */
} ofdm_rx_report;

int ofdm_rx_create(const ofdm_rx_cfg* cfg, ofdm_rx** out);
int ofdm_rx_destroy(ofdm_rx* h);

/* Replaces SynchAndChanEst.work(input_items, output_items) (:135-262), host buffers, including
 * the block's call-to-call state (count / corr_obs gating, persistent est_data_freq rows).
 * `h_in`: n_in complex64 items; `h_out`: n_out complex64 items (GNU Radio sync block: n_out == n_in).
 * Returns n_out (the reference returns len(output_items[0]), :262) or a negative ofdm_status. */
int64_t ofdm_rx_work(ofdm_rx* h, const float* h_in, int64_t n_in, float* h_out, int64_t n_out,
                     ofdm_rx_report* rep);

/* Copies block state to host (complex64 interleaved): what the reference exposes as attributes.
 * `row` selects the corr_obs row the reference wrote: 0 = estimate of the first detection (the one
 * the data equaliser keeps using, :242), 1 = estimate of the latest later-call detection (:171,188).
 * Any pointer may be NULL.  h_chan_freq[nfft]=est_chan_freq_P[row], h_chan_time[nfft]=est_chan_time[row],
 * h_synch_freq[S*Ks]=est_synch_freq[row], h_eq_gain[Ks]=eq_gain (latest), h_data_freq[num_ofdm_symb*Kd]=est_data_freq. */
int ofdm_rx_get_state(ofdm_rx* h, int32_t row, float* h_chan_freq, float* h_chan_time, float* h_synch_freq,
                      float* h_eq_gain, float* h_data_freq);

/* Frame-batched device fast path (benchmarks, multi-GPU shards): every frame is one reference
 * work() buffer processed with FRESH-instance semantics (own sync search from P=0, own channel
 * estimate; :143-248).  Frame f occupies d_iq[f*frame_stride .. +frame_len) (complex64 items).
 * n_pat = floor(floor(frame_len/L)/(S+D)) patterns are demodulated; outputs per frame:
 *   d_eq   [n_pat*D][Kd] complex64 equalised symbols (rows 3,7,.. of the reference already dropped), may be NULL
 *   d_bits hard bits of those symbols: packed MSB-first [n_pat*D*Kd*bps/8] bytes or unpacked [..*bps] bytes, may be NULL
 *   d_tsr  [4] int32: time_synch_ref[0..2], detected flag, may be NULL
 * Asynchronous on `stream`.  Returns n_pat*D (data symbols per frame) or a negative ofdm_status. */
int64_t ofdm_rx_demod_frames(ofdm_rx* h, const float* d_iq, int64_t n_frames, int64_t frame_stride,
                             int64_t frame_len, float* d_eq, uint8_t* d_bits, int32_t bits_mode,
                             int32_t* d_tsr, void* stream);

/* Per-frame state of the LAST ofdm_rx_demod_frames call, copied to host (synchronises the stream):
 * h_chan_freq[nfft], h_gain[Kd] (equaliser gain incl. lag de-rotation), h_chan_time[nfft]. NULL = skip. */
int ofdm_rx_get_frame_state(ofdm_rx* h, int64_t frame, float* h_chan_freq, float* h_gain, float* h_chan_time);

/* Optional HIP-event timing of the two kernels ofdm_rx_demod_frames launches (bench.py's roofline leg).
 * enable != 0 records events around the sync kernel and the demod kernel on the launch stream (a ring of 32
 * calls, so a timed loop needs no host synchronisation) and resets the ring; ofdm_rx_get_kernel_ms waits for the
 * recorded events and returns the MEAN durations over the calls recorded since (at most the last 32). */
int ofdm_rx_set_profiling(ofdm_rx* h, int32_t enable);
int ofdm_rx_get_kernel_ms(ofdm_rx* h, float* sync_ms, float* demod_ms);

/* Upper bound on sync trials per frame in the batch path (0 = none: scan the whole frame like the
 * reference, :143).  A frame with no sync costs one FFT pair per sample, so hosts may cap it. */
int ofdm_rx_set_max_trials(ofdm_rx* h, int32_t max_trials);

/* Sync search of the batch path and of ofdm_rx_work.  The reference tries the windows P = 0, 1, 2, ... one by one (:143-169);
 * exhaustive != 0 does exactly that (batch: one workgroup per frame; ofdm_rx_work: a trial table in windows, then a finalize
 * launch).  exhaustive == 0 (default) uses the screened search where its preconditions hold
 * (synch_dat[0] == 1, stride 1, num_synch_bins == nfft - 2): the trials between exactly evaluated anchor trials are screened
 * with an O(cp) sliding recurrence of the lag correlations and only flagged trials are evaluated exactly -- the accepted trial,
 * its lag and every output are those of the exhaustive search; ofdm_rx_work then searches its one
 * buffer in parallel segments and accepts and finalizes the first hit in the same launch.  Returns 1 if the screened search is
 * active for this handle afterwards, 0 if the exhaustive one is, or a negative ofdm_status. */
int ofdm_rx_set_sync_search(ofdm_rx* h, int32_t exhaustive);

/* Bytes of device workspace the batch path needs for n_frames (allocated lazily, grown on demand
 * OUTSIDE the asynchronous section: call ofdm_rx_reserve before capturing into a hipGraph). */
int ofdm_rx_reserve(ofdm_rx* h, int64_t n_frames);

/* ------------------------------------------------------------------------------------------ de-map
 * Replaces BitRecovery.work (G/LEGACY/gr-ofdm-rx/python/BitRecovery.py:66-189) on device buffers.
 * d_sym: n complex64; d_hard: n*bps bytes (one bit per byte, order [b0,b1,..] per symbol), may be NULL;
 * d_soft0/d_soft1: n*bps float32 max-log metrics llrp0/llrp1, may be NULL: QPSK literally as :105-125; 16/64-QAM
 * (modulation 4/6) by the same rule per axis -- -0.5/sigma^2 * distance to the nearest PAM level carrying bit value
 * 0 / 1, sigma = 0.7071*mean(dmin) over the buffer (:88,102) -- an extension the reference does not have. */
int ofdm_demap(ofdm_rx* h, const float* d_sym, int64_t n, int32_t modulation, uint8_t* d_hard,
               float* d_soft0, float* d_soft1, void* stream);

/* ------------------------------------------------------------------------------------------ TX
 * Replaces MultiAntennaSystem.multi_ant_binary_map + multi_ant_symb_gen (single antenna)
 * (G/LEGACY/gr-ofdm-rx/python/txrx_mod/MultiAntennaSystem.py:113-218) and SynchSignal (:13-30). */
typedef struct {
    int32_t nfft, cp_len, num_synch_bins, num_data_bins, synch_S, synch_D;
    int32_t modulation;        /* ofdm_modulation */
    int32_t zc_root;           /* 23 (SynchSignal.py:23) */
    int32_t device;
    int32_t reserved;
} ofdm_tx_cfg;

typedef struct ofdm_tx ofdm_tx;

int ofdm_tx_create(const ofdm_tx_cfg* cfg, ofdm_tx** out);
int ofdm_tx_destroy(ofdm_tx* h);

/* bits -> time-domain IQ.  Each frame has n_sym symbols laid out back to back (symbol s is a sync
 * symbol iff s % (S+D) < S); frame f is written at d_iq[f*frame_stride ..] (n_sym*(nfft+cp) items).
 * d_bits: per frame n_data_sym*Kd*bps bits, one per byte (bits_mode UNPACKED) or MSB-first packed. */
int ofdm_tx_modulate_frames(ofdm_tx* h, const uint8_t* d_bits, int32_t bits_mode, int64_t n_frames,
                            int32_t n_sym, float* d_iq, int64_t frame_stride, void* stream);

/* ---- decomposed transmitter stages.  The reference names these blocks only in a flowgraph --
 * txOFDM_random_bit_source -> txOFDM_ConstellationModulation(modulation) -> txOFDM_OFDM_Modulation(fft_size, pilot_locations)
 * -> txOFDM_IFFT(fft_size) -> txOFDM_CyclicPrefix(fft_size, cp_size) -> txOFDM_SynchDataMux(fft_size, cp_size, prime_no,
 * synch_every, synch_length)  (G/LEGACY/gr-ofdm-tx/grc/RXtransmit_6.grc:701-975, connections :1819-1854) -- and holds no
 * code for them.  Each stage is the corresponding slice of MultiAntennaSystem.multi_ant_binary_map / multi_ant_symb_gen
 * (:150-218) on device buffers; chained they equal ofdm_tx_modulate_frames bit for bit (the kernels share their device
 * functions).  The handle's cfg supplies the numerology: modulation (map), nfft / num_data_bins (grid), cp_len (CP),
 * num_synch_bins / zc_root / synch_S / synch_D (mux: S sync symbols before every D = synch_every data symbols). */
/* random_bit_source: n_bits bits, one per byte, of a counter-based stream: bit k = bit (k%32) of word (k/32)%4 of
 * Philox4x32-10(counter = k/128, key = seed); the window [offset, offset+n_bits) is produced (a stream block advances offset). */
int ofdm_tx_random_bits(ofdm_tx* h, uint64_t seed, uint64_t offset, uint8_t* d_bits, int64_t n_bits, void* stream);
/* ConstellationModulation: n_symbols * bps bits (MSB first per symbol; bits_mode as above) -> n_symbols complex64 (:150-178). */
int ofdm_tx_map(ofdm_tx* h, const uint8_t* d_bits, int32_t bits_mode, int64_t n_symbols, float* d_sym, void* stream);
/* pilot_locations of OFDM_Modulation: signed bin offsets (e.g. -21,-7,7,21) inside the occupied span
 * binsP(num_data_bins + n_pilots); those bins carry pilot_re + j pilot_im, the data symbols fill the other occupied bins in
 * list order.  n_pilots = 0 (default) gives the reference grid (:182-183).  Host array, copied; synchronises the device. */
int ofdm_tx_set_pilots(ofdm_tx* h, const int32_t* h_locations, int32_t n_pilots, float pilot_re, float pilot_im);
/* OFDM_Modulation: d_sym [n_rows][num_data_bins] -> d_grid [n_rows][nfft] (unused bins and DC zero) (:135-183). */
int ofdm_tx_grid(ofdm_tx* h, const float* d_sym, int64_t n_rows, float* d_grid, void* stream);
/* IFFT (do_ifft=1, add_cp=0): [n_rows][nfft] grid rows -> [n_rows][nfft] time samples, numpy.fft.ifft scaling (:199).
 * CyclicPrefix (0,1): [n_rows][nfft] time samples -> [n_rows][nfft+cp] CP-extended symbols, power-normalised as :200-218.
 * (1,1): both in one launch. */
int ofdm_tx_ifft_cp(ofdm_tx* h, const float* d_in, int64_t n_rows, int32_t do_ifft, int32_t add_cp, float* d_out, void* stream);
/* SynchDataMux: d_data [n_data_sym][nfft+cp] -> d_out: synch_S Zadoff-Chu sync symbols in front of every synch_D data symbols
 * (a trailing partial pattern keeps its sync symbols).  Returns the number of OUTPUT symbols
 * (n/D*(S+D) + (n%D ? S + n%D : 0)) or a negative ofdm_status; d_out must hold that many symbols. */
int64_t ofdm_tx_mux(ofdm_tx* h, const float* d_data, int64_t n_data_sym, float* d_out, void* stream);
/* the sync symbol(s) the mux inserts: h_sync_time [synch_S][nfft+cp] complex64 */
int ofdm_tx_get_sync_symbol(ofdm_tx* h, float* h_sync_time);

/* Loop-back channel (MultiAntennaSystem.py:221-260): y = x (*) taps (taps used as given; the
 * reference normalises to unit norm first, :86) + complex AWGN of variance noise_var (Philox
 * counter-based, reproducible from `seed`).  Per frame: in_len input samples, out_len outputs
 * (out_len <= in_len + n_taps - 1, the rest of the convolution tail is dropped).
 * d_taps: [n_taps] complex64, or [n_frames][n_taps] when per_frame_taps != 0. */
int ofdm_channel_apply(ofdm_tx* h, const float* d_in, int64_t n_frames, int64_t in_stride, int64_t in_len,
                       const float* d_taps, int32_t n_taps, int32_t per_frame_taps, float noise_var,
                       uint64_t seed, float* d_out, int64_t out_stride, int64_t out_len, void* stream);

/* ------------------------------------------------------- CFO-search receiver */
/* Replaces OFDMReceiver.SynchEstAndFO (G/LEGACY/gr-ofdm-rx/python/SynchEstAndFO.py:28-369): the
 * gr-RXOFDM receiver (root-37 ZC, stride cp-1, gate 0.4, linear SNR) plus a brute-force carrier
 * offset search -- every sync trial is evaluated once per candidate rotator (:258-282) -- and a
 * table of up to OFDM_FO_MAX_SYNC syncs per work() call, each followed by ONE equalised data
 * symbol (:332-358).  The candidate rotators are handed over as a table so that the caller decides
 * how `cfo` (:192) is formed (under the file's Python-2 semantics 1/fs is an integer division and
 * every rotator is 1). */
#define OFDM_FO_MAX_SYNC 100          /* rows of time_synch_ref / est_chan_freq_P / est_data_freq (:197,202-206) */
typedef struct ofdm_fo ofdm_fo;
typedef struct ofdm_fo_cfg {
    int32_t num_ofdm_symb;     /* :37  only sizes the output: corr_size = num_ofdm_symb / (S+D) rows (:362)      */
    int32_t nfft;
    int32_t cp_len;
    int32_t num_synch_bins;
    int32_t synch_S, synch_D;
    int32_t num_data_bins;
    int32_t n_fo;              /* len(fo_range) >= 1                                                           */
    double snr;                /* self.SNR (:150), linear                                                      */
    const float* rotators;     /* host, [n_fo][nfft] complex64 interleaved: self.cfo (:192); copied.  NULL with
                                * n_fo == 1: no rotation at all = work() of gr-RXOFDM's synch_and_chan_est
                                * (gr-RXOFDM/python/synch_and_chan_est.py:136-266), short data slices zero-padded (:228-230) */
    int32_t device;
    int32_t dsss;              /* 0: SynchEstAndFO.  >= 1: SynchEstFOAndDSSS with spreading factor self.DSSS     */
    const float* spread_code;  /* host, [dsss] complex64 interleaved: self.SC (SynchEstFOAndDSSS.py:253-262); copied */
} ofdm_fo_cfg;

typedef struct ofdm_fo_report {
    int32_t n_sync;            /* cor_obs + 1 before the end-of-call reset (:369)                               */
    int32_t count;
    int32_t dmax_tmp_ind;      /* candidate index of the LAST trial evaluated (:283), -1 if none ever was         */
    int32_t trials_run;
    int64_t n_data_items;      /* corr_size * num_data_bins values packed at the head of `out` (when count > 0) */
} ofdm_fo_report;

int ofdm_fo_create(const ofdm_fo_cfg* cfg, ofdm_fo** out);
int ofdm_fo_destroy(ofdm_fo* h);
/* SynchEstAndFO.work (:232-369), host buffers; returns n_out or a negative ofdm_status
 * (OFDM_ERR_INDEX: a 101st sync, :294-296; OFDM_ERR_SHAPE: short data slice :338-339 / reshape :364). */
int64_t ofdm_fo_work(ofdm_fo* h, const float* h_in, int64_t n_in, float* h_out, int64_t n_out, ofdm_fo_report* rep);
/* Block attributes, any pointer may be NULL: h_tsr[100][3] doubles (time_synch_ref), complex64 interleaved
 * h_chan_freq[100][nfft], h_chan_time[100][nfft], h_synch_freq[100][S*Ks], h_data_freq[100][Kd], h_eq_gain[Ks]. */
int ofdm_fo_get_state(ofdm_fo* h, double* h_tsr, float* h_chan_freq, float* h_chan_time, float* h_synch_freq,
                      float* h_data_freq, float* h_eq_gain);
/* DSSS variant (cfg.dsss >= 1; SynchEstFOAndDSSS.py:28-413, grc/OFDMReceiver_SynchEstFOAndDSSS.block.yml): after the
 * equaliser every row is despread -- est_data_freq_d[P][i] = mean_SF(est_data_freq[P][SF + i*DSSS] * conj(SC[SF])),
 * i < floor(Kd/DSSS) (:391-399) -- and ofdm_fo_work emits the first corr_size despread rows on EVERY call (:405-407).
 * OFDM_ERR_UNBOUND: row 0 of an earlier call fails the data guard before any row passed (:392).
 * h_data_freq_d[100][floor(Kd/DSSS)] complex64 interleaved. */
int ofdm_fo_get_despread(ofdm_fo* h, float* h_data_freq_d);

/* ------------------------------------------------ regression-tracking receiver */
/* Device primitives behind OFDMReceiver.SynchronizeAndEstimate (G/LEGACY/gr-ofdm-rx/python/SynchronizeAndEstimate.py:25-442).
 * The block's control flow is a strictly sequential pointer tracker (each window position depends on the outcome of the
 * previous one, from the sixth sync on through a least-squares line, :230-350); that scalar logic stays in the host-side
 * block mirror, which calls these entry points for every array computation: the strided acquisition search and the
 * per-sync trial (:236-276), the LS estimate (:344-378) and the data stage (:397-440).  [1,3]-type patterns with ONE sync
 * symbol per pattern only (the reference's reshape at :351 does not work for more). */
typedef struct ofdm_trk ofdm_trk;
typedef struct ofdm_trk_cfg {
    int32_t nfft, cp_len;
    int32_t num_synch_bins;    /* nfft - 2 (:122)                                                               */
    int32_t num_data_bins;
    int32_t synch_D;           /* data symbols per pattern (3)                                                   */
    int32_t rows_sync;         /* lmax_s: rows of est_chan_freq_p / est_synch_freq / est_chan_impulse (:143)     */
    int32_t rows_data;         /* lmax_d: rows of est_data_freq (:144)                                           */
    int32_t zc_root;           /* 23 (:125)                                                                      */
    double snr;                /* linear (:349,377,425)                                                          */
    int32_t device;
    int32_t reserved;
} ofdm_trk_cfg;
int ofdm_trk_create(const ofdm_trk_cfg* cfg, ofdm_trk** out);
int ofdm_trk_destroy(ofdm_trk* h);
/* work() buffer -> HBM; kept until the next load */
int ofdm_trk_load(ofdm_trk* h, const float* h_in, int64_t n_in);
/* `count` windows at first_ptr + i*step: h_peak[i] = max_d |del_mat[d]|, h_lag[i] = argmax d in 0..cp (:236-274).
 * The caller vouches that every window lies inside the buffer (:240). */
int ofdm_trk_trials(ofdm_trk* h, int64_t first_ptr, int32_t step, int32_t count, float* h_peak, int32_t* h_lag);
/* LS estimate of the window at window_ptr into row `row` (:344-378): est_chan_freq_p, est_chan_impulse, est_synch_freq
 * with the phase column lag_sync, and the data equaliser of that row de-rotated by lag_data (:421, may be -1).
 * row >= rows_sync -> OFDM_ERR_INDEX (the reference raises IndexError at :358). */
int ofdm_trk_accept(ofdm_trk* h, int32_t row, int64_t window_ptr, int32_t lag_sync, int32_t lag_data);
/* Data stage (:397-440) for syncs 0..n_sync-1: h_ptr[p] = time_synch_ref[p][0]; h_guard[p] != 0 iff :401 holds.  Rows
 * p*D+n of est_data_freq are equalised and renormalised in loop order; h_last[Kd] receives the last row processed (what
 * lands in out[0:Kd], :438-440), *last_row its index or -1.  A row past rows_data -> OFDM_ERR_INDEX; short and even empty
 * data slices are zero-padded like np.fft.fft(x, N) does. */
int ofdm_trk_demod(ofdm_trk* h, int32_t n_sync, const int64_t* h_ptr, const uint8_t* h_guard, float* h_last, int32_t* last_row);
/* complex64 interleaved, any pointer may be NULL: h_chan_freq[rows_sync][nfft], h_chan_impulse[rows_sync][nfft],
 * h_synch_freq[rows_sync][Ks], h_data_freq[rows_data][Kd] */
int ofdm_trk_get_state(ofdm_trk* h, float* h_chan_freq, float* h_chan_impulse, float* h_synch_freq, float* h_data_freq);

/* ------------------------------------------------------------------------------------------ misc */
/* *d_count += number of differing bits of two device byte strings (packed bit-streams): the BER numerator without moving
 * the streams -- with frames sharded over GPUs the ranks then exchange 8 bytes instead of their bits.  The
 * reference's idiom is bitwise_xor(a, b).sum() (TEST/GNU_RADIO_OFFLINE/pls_aio.py:131).  d_count is a DEVICE uint64 the
 * caller zeroes; asynchronous on `stream` (NULL: the default stream).  n_bytes <= 2^40. */
int ofdm_count_bit_errors(int32_t device, const uint8_t* d_a, const uint8_t* d_b, int64_t n_bytes, uint64_t* d_count, void* stream);

/* Measurement aid for bench.py: mode 0 = float4 device copy of `bytes` (achievable HBM rate of this chip, same run);
 * mode 1 = the demod kernel's access pattern without arithmetic (per symbol: skip gap_bytes, read sym_in_bytes, write
 * sym_out_bytes).  Asynchronous on `stream`. */
int ofdm_bandwidth_probe(int32_t device, const void* d_in, void* d_out, int64_t bytes, int32_t mode, int32_t sym_in_bytes,
                         int32_t gap_bytes, int32_t sym_out_bytes, int64_t n_sym, void* stream);
/* Frame partition of the N-GPU path: frames are independent (each is one reference work() buffer: own sync, own estimate,
 * SynchAndChanEst.py:135-262 keeps no state between the buffers the batch entry takes), so rank `rank` of `world` owns the
 * contiguous frames [*first, *first + *count) and calls ofdm_rx_demod_frames on them with its own handle on its own device; the
 * only exchange is the caller's all-gather of the packed bits.  The same rule as ofdm_mi355x.dist.shard_frames and bench.py, for
 * hosts that are not Python.  n_frames_total must be a multiple of world (an all-gather needs equal counts: pad the batch), else
 * OFDM_ERR_INVALID.  Host arithmetic only: no device is touched. */
int ofdm_shard_frames(int64_t n_frames_total, int32_t world, int32_t rank, int64_t* first, int64_t* count);
int ofdm_abi_version(void);
const char* ofdm_last_error(void);
/* plain device memory helpers so hosts without torch can drive the batch path */
int ofdm_device_malloc(int32_t device, void** d_ptr, int64_t bytes);
int ofdm_device_free(int32_t device, void* d_ptr);
int ofdm_memcpy_h2d(int32_t device, void* d_dst, const void* h_src, int64_t bytes);
int ofdm_memcpy_d2h(int32_t device, void* h_dst, const void* d_src, int64_t bytes);
int ofdm_device_synchronize(int32_t device);

#ifdef __cplusplus
}
#endif
#endif /* OFDM_MI355X_H */
