// ofdm_launch.hpp -- host-visible argument structs + launch wrappers of the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "ofdm_device.hpp"

namespace ofdm {

// ---- RX data demod (reference: SynchAndChanEst.py:221-248, "Loop B") --------------------------
struct DemodArgs {
    const cf* iq;            // frames, frame f at iq + f*frame_stride
    int64_t frame_stride;
    int64_t frame_len;       // valid samples per frame (windows past the end read zeros, like fft(x, N))
    int n_frames;
    const int* tsr;          // [n_frames][4]  {time_synch_ref[0], lag, int(max), detected}
    const cf* gain;          // [n_frames][Kd] conj(H)/(|H|^2+1/snr) * exp(j 2pi lag k/N)
    cf* eq;                  // [n_frames][rows_per_frame][Kd] or null
    uint8_t* bits;           // hard bits or null
    int bits_mode;           // ofdm_bits_mode
    int mod;                 // bits per symbol of the de-mapper (1,2,4,6)
    int n_dsym;              // data symbols visited per frame (= n_pat * D)
    int spc;                 // data symbols per chunk (one chunk = one symbol slot's loop)
    int chunks_per_frame;
    int row_stride_pat;      // output row = p*row_stride_pat + n   (S+D in stream mode, D in batch mode)
    int rows_per_frame;
    int zero_skipped;        // write zeros for patterns whose guard fails (batch mode)
    int variant;             // kernel tuning variant (0 = default)
    unsigned* stamps;        // diagnostic variant 9: [workgroups][waves][8] phase cycle sums, or null
    const cf* rot;           // per-sample rotator e^{j 2pi fo n/fs} [nfft] applied to every window (CFO receiver), or null
    int host_guard;          // 1: a frame is demodulated iff tsr[frame][3] != 0 (the host applied the reference's own guard)
    unsigned* work;          // [2] device words {next chunk, workgroups done}, both 0 between launches: work queue (batch path), or null
};

// ---- RX sync search + LS estimate (reference: SynchAndChanEst.py:143-219, "Loop A") -----------
struct SyncArgs {
    const cf* iq;
    int64_t frame_stride;
    int64_t frame_len;
    int n_frames;
    int mode;                // 0: per-frame sequential search + finalize; 1: trial table only (frame 0)
    int p_begin;
    int p_count;             // mode 0: max trials (<=0: unbounded); mode 1: number of trials in the table
    int force_accept;        // mode 0: accept trial p_begin regardless of the gate (host already decided)
    int* tsr;                // [n_frames][4]
    cf* H;                   // [n_frames][N]   est_chan_freq_P row
    const cf* H_for_gain;    // stream block, calls after the first: the data equaliser keeps using row 0 (:242); null = use H
    cf* gain;                // [n_frames][Kd]
    cf* htime;               // [n_frames][N]   est_chan_time row, or null
    cf* esf;                 // [n_frames][MM]  est_synch_freq row, or null
    cf* eqg;                 // [n_frames][Ks]  eq_gain, or null
    cf* yscratch;            // [n_frames][MM]  raw sync-bin values of the current trial (needed for esf), or null
    float* trial_m;          // mode 1: [n_rot*p_count] max|corr| (-1 = trial not valid), candidate-major
    int* trial_d;            // mode 1: [n_rot*p_count] argmax lag
    const cf* rot;           // carrier-offset rotators [n_rot][nfft] (mode 1) / the one rotator of the finalize (mode 0), or null
    int n_rot;               // mode 1: candidates per trial (0 or 1 = plain)
    int off_delta;           // added to the window start P*stride + cp (+L*LL); 0 = the reference layout of SynchAndChanEst
    int host_valid;          // 1: the host already applied the block's own window-validity rule
    int gain_lag_set;        // 1: the data gains are de-rotated with gain_lag instead of the trial's lag (may be negative)
    int gain_lag;
    int force_dhat_p1;       // mode 0: lag+1 that replaces the trial's own arg-max lag (SynchEstAndFO.py:285,300); 0 = off
    int scan_block;          // mode 0, > 0: screened search (rx_sync_scan_kernel) with blocks of this many trials
    const cf* scan_g;        // [nfft + 2] G[m] = sum_k e^{j 2pi m k/N} conj(zc_k), G[nfft] = G[0], then {max |G|, 0}
    unsigned* stamps;        // OFDM_EXPERIMENTS build only: [workgroups][8] cycle sums per phase of the scan kernel, or null
    int keep_on_miss;        // mode 0: a frame without an accepted trial leaves every output row untouched (stream block: the old
                             // estimate stays in force, SynchAndChanEst.py:166-219 only writes on detection) except tsr[3] = 0
    int* tsr_host;           // mode 0: optional second copy of the frame's tsr words in device-visible HOST memory (stream block: the
                             // host reads them after its one synchronisation without a copy in the stream), or null
    int n_seg;               // screened search of ONE long buffer (n_frames == 1): > 0 = the trials p_begin .. are cut into n_seg
    int seg_len;             //   segments of seg_len trials searched in parallel; the first accepted trial overall is finalized
    int* seg_state;          //   [2] device words {first hit so far = INT_MAX, finalized early = 0}; the last finalize launch re-arms them
    int seg_base;            //   this launch covers the segments seg_base .. seg_base + seg_launch - 1 (seg_launch == 0: launch_rx_sync
    int seg_launch;          //   stages the search itself: the first SYNC_STAGE_SEGS segments, then the rest behind them)
    int seg_final;           //   the one-workgroup launch the launcher adds behind the search launches: 1 = the search ends here
                             //   (finalize the first hit unless an early finalize did, report a miss, re-arm seg_state); 2 = early
                             //   finalize behind a first stage the caller runs on its own (a hit there is the first hit: finalize it
                             //   and mark seg_state[1]; nothing found: leave everything to the later stages); 0 = none
};

// Segments of the first stage of a staged segment search: a continuing stream finds its sync a few symbols into the buffer, so
// the later segments, launched behind the first stage, see the published hit and leave at once.
constexpr int SYNC_STAGE_SEGS = 64;

struct DemapArgs {
    const cf* sym;
    int64_t n;
    int mod;
    uint8_t* hard;           // n*bps bytes or null
    float* soft0;            // n*bps or null (QPSK)
    float* soft1;
    double* partial;         // [DEMAP_PARTIALS] scratch for the dmin mean
};
constexpr int DEMAP_PARTIALS = 256;

// ---- TX (reference: MultiAntennaSystem.py:113-218) ---------------------------------------------
struct TxDev {
    int nfft, cp, L, Ks, Kd, S, D, bps;
    const cf* tw;
    const cf* zc;            // [S*Ks]
};
struct ModArgs {
    const uint8_t* bits;
    int bits_mode;
    int64_t bits_stride;     // bytes per frame in `bits`
    int n_frames;
    int n_sym;               // symbols per frame
    cf* iq;
    int64_t frame_stride;
    const cf* sync_time;     // [S][L] the finished sync symbol(s) (synthesised once per handle by the same device functions), or null
};
// decomposed TX stages
struct GridArgs {
    const cf* sym;           // [n_rows][Kd]
    int64_t n_rows;
    const int* pilots;       // [n_pilots] ascending list indices into binsP(Kd + n_pilots), or null
    int n_pilots;
    cf pilot_value;
    cf* grid;                // [n_rows][nfft]
};
struct TimeArgs {
    const cf* in;            // do_ifft: [n_rows][nfft] grid rows; else [n_rows][nfft] time samples
    int64_t n_rows;
    int do_ifft, do_cp;
    cf* out;                 // do_cp: [n_rows][L]; else [n_rows][nfft]
};
struct MuxArgs {
    const cf* sync_time;     // [S][L]
    const cf* data;          // [n_data_sym][L]
    int64_t n_out_sym;
    cf* out;                 // [n_out_sym][L]
};
struct ChanArgs {
    const cf* in;
    int64_t in_stride, in_len;
    int n_frames;
    const cf* taps;
    int n_taps;
    int per_frame_taps;
    float noise_std;         // per real component: sqrt(noise_var/2)
    uint64_t seed;
    cf* out;
    int64_t out_stride, out_len;
};

hipError_t launch_rx_demod(const RxDev& rx, const DemodArgs& a, hipStream_t s);
hipError_t launch_rx_sync(const RxDev& rx, const SyncArgs& a, hipStream_t s);
// htime[r] = ifft(H[r]) for n_rows rows of nfft bins (est_chan_time on demand)
hipError_t launch_rx_chan_time(const RxDev& rx, const cf* H, cf* htime, int n_rows, hipStream_t s);
hipError_t launch_demap(const DemapArgs& a, hipStream_t s);
hipError_t launch_bit_errors(const uint8_t* a, const uint8_t* b, int64_t n, unsigned long long* count, hipStream_t s);
// out[row][i] = mean_SF( in[row][SF + i*dsss] * conj(code[SF]) ), i < n_spread  (SynchEstFOAndDSSS.py:391-399)
// rows visited in order; row r of frame f = f*D + n is divided by sqrt(mean |row f|^2) (SynchronizeAndEstimate.py:431-434)
hipError_t launch_row_renorm(cf* eq, int Kd, int D, int n_frames, const int* tsr, hipStream_t s);
// out = est_data_freq rows minus rows 3, 3+SD, ... (SynchAndChanEst.py:249-255), row-major
hipError_t launch_pack_rows(const cf* edf, int rows, int Kd, int SD, cf* out, hipStream_t s);
hipError_t launch_despread(const cf* in, int in_row_stride, const cf* code, int dsss, int n_spread, int rows, cf* out, hipStream_t s);
hipError_t launch_tx_modulate(const TxDev& tx, const ModArgs& a, hipStream_t s);
hipError_t launch_channel(const ChanArgs& a, hipStream_t s);
hipError_t launch_tx_random_bits(uint64_t seed, uint64_t offset, uint8_t* out, int64_t n, hipStream_t s);
hipError_t launch_tx_map(const uint8_t* bits, int bits_mode, int bps, int64_t n_sym, cf* out, hipStream_t s);
hipError_t launch_tx_grid(const TxDev& tx, const GridArgs& a, hipStream_t s);
hipError_t launch_tx_sync_grid(const TxDev& tx, cf* grid, hipStream_t s);
hipError_t launch_tx_time(const TxDev& tx, const TimeArgs& a, hipStream_t s);
hipError_t launch_tx_mux(const TxDev& tx, const MuxArgs& a, hipStream_t s);
size_t rx_lds_bytes(int nfft);
// RxDev::zcp for a host copy of the Zadoff-Chu sequence zc[S*Ks]: [S][nfft] entries in lane / register-slot order
std::vector<cf> rx_zc_lane_table(int nfft, int Ks, int S, const cf* zc);
// block length of the screened sync search for this numerology (0: the preconditions do not hold, use the sequential search)
int rx_sync_scan_block(const RxDev& rx);
hipError_t launch_probe(const void* in, void* out, int64_t n16, int mode, int sym_in16, int gap16, int sym_out16, int64_t n_sym,
                        hipStream_t s);

}  // namespace ofdm
