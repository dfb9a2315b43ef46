// capi.hip -- C ABI of libofdm_mi355x.so (declared in include/ofdm_mi355x.h).
// Host-side logic only: handle/state management, the reference's stream-block control flow
// (SynchAndChanEst.work, gr-utsa_ofdm/python/SynchAndChanEst.py:135-262) and kernel launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/ofdm_mi355x.h"
#ifdef OFDM_EXPERIMENTS
#include "../../tools/experiments/ofdm_experiments.h"
#endif
#include "ofdm_launch.hpp"

using namespace ofdm;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(OFDM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

bool supported_nfft(int n) { return n == 64 || n == 128 || n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096; }

std::vector<cf> make_twiddles(int n) {
    std::vector<cf> t(n);
    for (int j = 0; j < n; ++j) {
        const double a = -2.0 * M_PI * double(j) / double(n);
        t[j] = cf{float(std::cos(a)), float(std::sin(a))};
    }
    return t;
}

// SynchAndChanEst.py:52-59 / SynchSignal.py:23-30 (root 23) ; synch_and_chan_est.py:54-64 (root 37)
std::vector<cf> make_zc(int mm, int root, int parity_of) {
    std::vector<cf> z(mm);
    for (int n = 0; n < mm; ++n) {
        const double x0 = double(n), x1 = double(n + 1);
        const double q = (parity_of % 2 == 0) ? (x0 * x0 / 2.0) : (x0 * x1 / 2.0);
        const double a = -(2.0 * M_PI / double(mm)) * double(root) * q;
        z[n] = cf{float(std::cos(a)), float(std::sin(a))};
    }
    return z;
}

// G[m] = sum_i e^{+j 2pi m k_i / N} conj(zc_i), m = 0..N (G[N] = G[0]): the kernel of the screened sync search's recurrence
// (rx_sync_scan_kernel).  Unnormalised inverse DFT of the sync symbol's conjugated grid row, iterative radix-2 in double.
std::vector<cf> make_scan_table(int N, int Ks, const std::vector<cf>& zc) {
    std::vector<std::complex<double>> g(size_t(N), {0.0, 0.0});
    const int h = Ks / 2;
    for (int i = 0; i < Ks; ++i) {
        const int k = i < h ? N - h + i : i - h + 1;                    // binsP(Ks) (SynchAndChanEst.py:38-41)
        g[size_t(k)] = std::conj(std::complex<double>(zc[size_t(i)].x, zc[size_t(i)].y));
    }
    for (int i = 1, j = 0; i < N; ++i) {                                // bit reversal
        int bit = N >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(g[size_t(i)], g[size_t(j)]);
    }
    for (int len = 2; len <= N; len <<= 1) {
        const double ang = 2.0 * M_PI / double(len);                    // e^{+j..}: inverse transform
        for (int i = 0; i < N; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const std::complex<double> w(std::cos(ang * k), std::sin(ang * k));
                const auto x = g[size_t(i + k)], y = g[size_t(i + k + len / 2)] * w;
                g[size_t(i + k)] = x + y;
                g[size_t(i + k + len / 2)] = x - y;
            }
    }
    std::vector<cf> out(size_t(N) + 2);
    double gmax = 0.0;
    for (int m = 0; m <= N; ++m) {
        out[size_t(m)] = cf{float(g[size_t(m % N)].real()), float(g[size_t(m % N)].imag())};
        gmax = std::max(gmax, std::abs(g[size_t(m % N)]));
    }
    out[size_t(N) + 1] = cf{float(gmax * (1.0 + 1e-6)), 0.f};          // max |G[m]|: the checkpoint bound of the screened search
    return out;
}

template <class T>
int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) return OFDM_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
    if (e != hipSuccess) return fail(OFDM_ERR_NOMEM, "hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e));
    return OFDM_OK;
}

// Derived constants of a receiver configuration; returns the Zadoff-Chu root.
int fill_rxdev(const ofdm_rx_cfg& cfg, RxDev& d) {
    const ofdm_rx_cfg* c = &cfg;
    const int N = c->nfft, Ks = c->num_synch_bins, Kd = c->num_data_bins, S = c->synch_S;
    const int MM = S * Ks;
    d.nfft = N;
    d.cp = c->cp_len;
    d.L = N + c->cp_len;
    d.Ks = Ks;
    d.Kd = Kd;
    d.S = S;
    d.D = c->synch_D;
    d.MM = MM;
    d.bps = c->modulation;
    double snr_ls, snr_eq, snr_data, gate;
    int root;
    if (c->compat == OFDM_COMPAT_UTSA) {
        const double snr_lin = std::pow(10.0, c->snr / 20.0);     // SynchAndChanEst.py:99 (sic: /20)
        snr_ls = snr_lin;                                          // :180
        snr_eq = c->snr;                                           // :214 uses the raw argument
        snr_data = snr_lin;                                        // :245
        gate = c->scale_factor_gate;                               // :166
        d.stride = 1;                                              // :77
        root = 23;                                                 // :52
    } else {
        snr_ls = snr_eq = snr_data = c->snr;                       // synch_and_chan_est.py:184,217,247
        gate = 0.4;                                                // :170
        d.stride = c->cp_len - 1;                                  // :81
        root = 37;                                                 // :54
    }
    d.gate_mm = float(gate * double(MM));
    d.inv_ls = float(1.0 / (double(S) * (1.0 + 1.0 / snr_ls)));
    d.inv_snr_data = float(1.0 / snr_data);
    d.inv_snr_eqsync = float(1.0 / snr_eq);
    return root;
}

}  // namespace

struct ofdm_rx {
    ofdm_rx_cfg cfg{};
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;       // stream block: the first stage of the sync search runs here, under the rest of the upload
    hipEvent_t ev_up = nullptr, ev_s1 = nullptr;
    RxDev dev{};
    cf* d_tw = nullptr;
    cf* d_zc = nullptr;
    // ---- stream block state (SynchAndChanEst.py:72,83-100)
    int count = 0;
    int corr_obs = -1;
    double tsr[3] = {0, 0, 0};
    cf* d_in = nullptr;
    int64_t in_cap = 0;
    cf* d_edf = nullptr;                 // est_data_freq [num_ofdm_symb][Kd]
    cf* d_pack = nullptr;                // the same rows without the deleted ones (:249-255), one contiguous copy to the host
    int* pin_tsr = nullptr;              // [4] pinned host copy of s_tsr, written by the search kernel itself (tsr_host)
    int* pin_tsr_dev = nullptr;          //     its device address
    // GNU Radio sized buffers (a few symbols): a copy in the stream costs more than it moves (5-10 us of engine latency plus a
    // 10-20 us bubble next to the kernels), so buffers up to PIN_IN_BYTES / PIN_OUT_BYTES go through pinned host memory that the kernels read and
    // write in place over the link; the host's share is a memcpy of a few tens of KB
    // (thresholds measured: input in place pays up to ~1 MB -- a 32-symbol buffer 0.107 -> 0.098 ms -- and loses at 4 MB, 0.21 ->
    // 0.26-0.55 ms; output in place loses at 1.7 MB, 0.21 -> 0.275 ms)
    static constexpr size_t PIN_IN_BYTES = size_t(1024) << 10, PIN_OUT_BYTES = size_t(256) << 10;
    cf* pin_in = nullptr;
    cf* pin_in_dev = nullptr;
    cf* pin_out = nullptr;
    cf* pin_out_dev = nullptr;
    int* s_tsr = nullptr;                // [4]
    cf* s_H = nullptr;                   // [2][N]   rows 0 / 1 of est_chan_freq_P
    cf* s_htime = nullptr;               // [2][N]
    cf* s_esf = nullptr;                 // [2][MM]
    cf* s_eqg = nullptr;                 // [Ks]
    cf* s_gain = nullptr;                // [Kd]
    cf* s_ysc = nullptr;                 // [MM]
    float* d_trial_m = nullptr;
    int* d_trial_d = nullptr;
    static constexpr int TRIAL_CAP = 1024;
    double* d_partial = nullptr;
    // ---- batch (frame) workspace
    int64_t cap_frames = 0;
    int* f_tsr = nullptr;
    cf* f_H = nullptr;
    cf* f_gain = nullptr;
    cf* f_htime = nullptr;
    int max_trials = 0;
    int scan_block = 0;                  // > 0: the batch path's sync search is screened in blocks of this many trials
    cf* d_scan_g = nullptr;              // [N + 2] recurrence kernel G, then {max |G|, 0}
    int* d_seg_state = nullptr;          // [2] {first hit, segments done} of the stream block's segment-parallel search
    unsigned* d_work = nullptr;          // [2] work queue of the batch demod launch {next chunk, workgroups done}
    bool use_queue = true;
    bool seg_armed = false;              // the kernel re-arms the two words itself; false after a launch that did not complete
    int variant = 0;
    unsigned* d_stamps = nullptr;
    bool profiling = false;
    static constexpr int PROF_RING = 32;       // per-call event triples: no host sync inside a timed loop
    hipEvent_t ev[3 * PROF_RING] = {};
    int64_t prof_calls = 0;
};

struct ofdm_fo {
    ofdm_fo_cfg cfg{};
    hipStream_t stream = nullptr;
    RxDev dev{};
    cf* d_tw = nullptr;
    cf* d_zc = nullptr;
    cf* d_rot = nullptr;                 // [n_fo][N]  self.cfo (SynchEstAndFO.py:192)
    // ---- block state (SynchEstAndFO.py:197-222)
    int count = 0;
    int cor_obs = -1;
    int dmax_tmp_ind = -1;
    double tsr[OFDM_FO_MAX_SYNC][3] = {};
    cf* d_in = nullptr;
    int64_t in_cap = 0;
    int* t_tsr = nullptr;                // [100][4]   device copy used by the kernels
    cf* t_H = nullptr;                   // [100][N]
    cf* t_htime = nullptr;               // [100][N]
    cf* t_esf = nullptr;                 // [100][MM]
    cf* t_gain = nullptr;                // [100][Kd]
    cf* t_edf = nullptr;                 // [100][Kd]  est_data_freq
    cf* d_code = nullptr;                // [dsss]     self.SC      (DSSS variant only)
    cf* t_edfd = nullptr;                // [100][Kd/dsss] est_data_freq_d
    int n_spread = 0;
    cf* s_eqg = nullptr;                 // [Ks]
    cf* s_ysc = nullptr;                 // [MM]
    float* d_trial_m = nullptr;          // [n_fo * TRIAL_WIN]
    int* d_trial_d = nullptr;
    static constexpr int TRIAL_WIN = 256;
};

struct ofdm_trk {
    ofdm_trk_cfg cfg{};
    hipStream_t stream = nullptr;
    RxDev dev{};
    cf* d_tw = nullptr;
    cf* d_zc = nullptr;
    cf* d_in = nullptr;
    int64_t in_cap = 0;
    int64_t n_in = 0;
    int* t_tsr = nullptr;                // [rows_sync][4]
    cf* t_H = nullptr;                   // [rows_sync][N]
    cf* t_imp = nullptr;                 // [rows_sync][N]
    cf* t_esf = nullptr;                 // [rows_sync][Ks]
    cf* t_gain = nullptr;                // [rows_sync][Kd]
    cf* t_edf = nullptr;                 // [rows_data][Kd]
    cf* s_ysc = nullptr;                 // [Ks]
    float* d_trial_m = nullptr;
    int* d_trial_d = nullptr;
    static constexpr int TRIAL_CAP = 4096;
};

struct ofdm_tx {
    ofdm_tx_cfg cfg{};
    hipStream_t stream = nullptr;
    TxDev dev{};
    cf* d_tw = nullptr;
    cf* d_zc = nullptr;
    // decomposed stages
    cf* d_sync_time = nullptr;           // [S][L] the sync symbol(s) SynchDataMux inserts, synthesised once at creation
    int* d_pilots = nullptr;             // ascending list indices into binsP(Kd + n_pilots)
    int n_pilots = 0;
    cf pilot_value = cf{1.f, 0.f};
};

extern "C" {

int ofdm_abi_version(void) { return OFDM_ABI_VERSION; }

int ofdm_shard_frames(int64_t n_frames_total, int32_t world, int32_t rank, int64_t* first, int64_t* count) {
    if (!first || !count || world < 1 || rank < 0 || rank >= world || n_frames_total < 0)
        return fail(OFDM_ERR_INVALID, "ofdm_shard_frames: bad argument");
    if (n_frames_total % world)
        return fail(OFDM_ERR_INVALID, "n_frames_total=%lld is not a multiple of world=%d", (long long)n_frames_total, int(world));
    *count = n_frames_total / world;
    *first = int64_t(rank) * *count;
    return OFDM_OK;
}
const char* ofdm_last_error(void) { return g_last_error.c_str(); }

int ofdm_device_malloc(int32_t device, void** d_ptr, int64_t bytes) {
    if (!d_ptr || bytes < 0) return fail(OFDM_ERR_INVALID, "ofdm_device_malloc: bad argument");
    HIP_TRY(hipSetDevice(device));
    hipError_t e = hipMalloc(d_ptr, size_t(bytes));
    if (e != hipSuccess) return fail(OFDM_ERR_NOMEM, "hipMalloc(%lld): %s", (long long)bytes, hipGetErrorString(e));
    return OFDM_OK;
}
int ofdm_device_free(int32_t device, void* d_ptr) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(d_ptr));
    return OFDM_OK;
}
int ofdm_memcpy_h2d(int32_t device, void* d_dst, const void* h_src, int64_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(d_dst, h_src, size_t(bytes), hipMemcpyHostToDevice));
    return OFDM_OK;
}
int ofdm_memcpy_d2h(int32_t device, void* h_dst, const void* d_src, int64_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(h_dst, d_src, size_t(bytes), hipMemcpyDeviceToHost));
    return OFDM_OK;
}
int ofdm_device_synchronize(int32_t device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return OFDM_OK;
}

// ------------------------------------------------------------------------------------------ RX
int ofdm_rx_destroy(ofdm_rx* h) {
    if (!h) return OFDM_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->pin_tsr) (void)hipHostFree(h->pin_tsr);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    void* ptrs[] = {h->d_pack, h->d_tw,    h->d_zc,  h->d_in,  h->d_edf,     h->s_tsr,     h->s_H,       h->s_htime, h->s_esf, h->s_eqg,
                    h->s_gain,  h->s_ysc, h->d_trial_m, h->d_trial_d, h->d_partial, h->f_tsr, h->f_H, h->f_gain, h->f_htime,
                    h->d_scan_g, h->d_seg_state, h->d_work};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : h->ev)
        if (e) (void)hipEventDestroy(e);
    if (h->ev_up) (void)hipEventDestroy(h->ev_up);
    if (h->ev_s1) (void)hipEventDestroy(h->ev_s1);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return OFDM_OK;
}

int ofdm_rx_set_profiling(ofdm_rx* h, int32_t enable) {
    if (!h) return fail(OFDM_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (enable && !h->ev[0])
        for (auto& e : h->ev) HIP_TRY(hipEventCreate(&e));
    h->profiling = enable != 0;
    h->prof_calls = 0;
    return OFDM_OK;
}

int ofdm_rx_get_kernel_ms(ofdm_rx* h, float* sync_ms, float* demod_ms) {
    if (!h || !h->ev[0] || h->prof_calls == 0) return fail(OFDM_ERR_INVALID, "profiling was not enabled / no call recorded");
    HIP_TRY(hipSetDevice(h->cfg.device));
    // mean over the calls recorded since ofdm_rx_set_profiling (at most the last PROF_RING of them)
    const int64_t n = h->prof_calls < ofdm_rx::PROF_RING ? h->prof_calls : ofdm_rx::PROF_RING;
    double ssum = 0, dsum = 0;
    for (int64_t c = h->prof_calls - n; c < h->prof_calls; ++c) {
        hipEvent_t* e = h->ev + 3 * (c % ofdm_rx::PROF_RING);
        HIP_TRY(hipEventSynchronize(e[2]));
        float a = 0, b = 0;
        HIP_TRY(hipEventElapsedTime(&a, e[0], e[1]));
        HIP_TRY(hipEventElapsedTime(&b, e[1], e[2]));
        ssum += a;
        dsum += b;
    }
    if (sync_ms) *sync_ms = float(ssum / double(n));
    if (demod_ms) *demod_ms = float(dsum / double(n));
    return OFDM_OK;
}

#ifdef OFDM_EXPERIMENTS
// Bench-only build (tools/experiments/ofdm_experiments.h): kernel tuning variants and the s_memtime-stamped diagnostic.
// Not part of the product ABI and not compiled into libofdm_mi355x.so.
int ofdm_exp_set_variant(ofdm_rx* h, int32_t variant) {
    if (!h || variant < 0) return fail(OFDM_ERR_INVALID, "bad argument");
    h->variant = variant;
    return OFDM_OK;
}

int ofdm_exp_set_stamp_buffer(ofdm_rx* h, void* d_stamps) {
    if (!h) return fail(OFDM_ERR_INVALID, "null handle");
    h->d_stamps = static_cast<unsigned*>(d_stamps);
    return OFDM_OK;
}
#endif

int ofdm_rx_set_sync_search(ofdm_rx* h, int32_t exhaustive) {
    if (!h) return fail(OFDM_ERR_INVALID, "null handle");
    h->scan_block = exhaustive ? 0 : rx_sync_scan_block(h->dev);
    if (h->scan_block > 0 && !h->d_scan_g) h->scan_block = 0;
    return h->scan_block > 0 ? 1 : 0;
}

int ofdm_rx_set_max_trials(ofdm_rx* h, int32_t max_trials) {
    if (!h || max_trials < 0) return fail(OFDM_ERR_INVALID, "bad argument");
    h->max_trials = max_trials;
    return OFDM_OK;
}

int ofdm_rx_create(const ofdm_rx_cfg* c, ofdm_rx** out) {
    if (!c || !out) return fail(OFDM_ERR_INVALID, "ofdm_rx_create: null argument");
    *out = nullptr;
    if (!supported_nfft(c->nfft)) return fail(OFDM_ERR_INVALID, "nfft=%d unsupported (64,128,...,4096)", c->nfft);
    if (c->cp_len < 0 || c->cp_len >= c->nfft) return fail(OFDM_ERR_INVALID, "cp_len=%d out of range", c->cp_len);
    if (c->num_synch_bins < 2 || c->num_synch_bins > c->nfft || (c->num_synch_bins & 1))
        return fail(OFDM_ERR_INVALID, "num_synch_bins=%d must be even and in [2, nfft]", c->num_synch_bins);
    if (c->num_data_bins < 2 || c->num_data_bins > c->nfft || (c->num_data_bins & 1))
        return fail(OFDM_ERR_INVALID, "num_data_bins=%d must be even and in [2, nfft]", c->num_data_bins);
    if (c->synch_S < 1 || c->synch_D < 1) return fail(OFDM_ERR_INVALID, "synch_dat must be [>=1, >=1]");
    if (c->num_ofdm_symb < 1) return fail(OFDM_ERR_INVALID, "num_ofdm_symb must be >= 1");
    if (c->modulation != 1 && c->modulation != 2 && c->modulation != 4 && c->modulation != 6)
        return fail(OFDM_ERR_INVALID, "modulation must be 1, 2, 4 or 6 bits per symbol");
    if (c->compat == OFDM_COMPAT_RXOFDM && c->cp_len < 2)
        return fail(OFDM_ERR_INVALID, "gr-RXOFDM search stride is cp_len-1: cp_len must be >= 2");
    if (c->compat != OFDM_COMPAT_UTSA && c->compat != OFDM_COMPAT_RXOFDM) return fail(OFDM_ERR_INVALID, "bad compat");

    HIP_TRY(hipSetDevice(c->device));
    ofdm_rx* h = new (std::nothrow) ofdm_rx();
    if (!h) return fail(OFDM_ERR_NOMEM, "out of host memory");
    h->cfg = *c;
    const int N = c->nfft, Ks = c->num_synch_bins, Kd = c->num_data_bins, S = c->synch_S;
    const int MM = S * Ks;
    RxDev& d = h->dev;
    const int root = fill_rxdev(*c, d);

    int rc = OFDM_OK;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        rc = fail(OFDM_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    if (rc == OFDM_OK && (hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess ||
                          hipEventCreateWithFlags(&h->ev_up, hipEventDisableTiming) != hipSuccess ||
                          hipEventCreateWithFlags(&h->ev_s1, hipEventDisableTiming) != hipSuccess))
        rc = fail(OFDM_ERR_HIP, "second stream / events of the stream block");
    auto tw = make_twiddles(N);
    // utsa: parity of MM decides the ZC form (:56); gr-RXOFDM: parity of num_synch_bins (synch_and_chan_est.py:56-61)
    auto zc = make_zc(MM, root, c->compat == OFDM_COMPAT_UTSA ? MM : Ks);
    const size_t rows = size_t(c->num_ofdm_symb);
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_tw, size_t(N));
    auto zcp = rx_zc_lane_table(N, Ks, S, zc.data());                  // lane-order copy, stored behind the sequence itself
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_zc, size_t(MM) + zcp.size());
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_edf, rows * Kd);
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_pack, rows * Kd);
    if (rc == OFDM_OK && (hipHostMalloc(reinterpret_cast<void**>(&h->pin_tsr), 4 * sizeof(int), hipHostMallocMapped) != hipSuccess ||
                          hipHostMalloc(reinterpret_cast<void**>(&h->pin_in), ofdm_rx::PIN_IN_BYTES, hipHostMallocMapped) != hipSuccess ||
                          hipHostMalloc(reinterpret_cast<void**>(&h->pin_out), ofdm_rx::PIN_OUT_BYTES, hipHostMallocMapped) != hipSuccess ||
                          hipHostGetDevicePointer(reinterpret_cast<void**>(&h->pin_tsr_dev), h->pin_tsr, 0) != hipSuccess ||
                          hipHostGetDevicePointer(reinterpret_cast<void**>(&h->pin_in_dev), h->pin_in, 0) != hipSuccess ||
                          hipHostGetDevicePointer(reinterpret_cast<void**>(&h->pin_out_dev), h->pin_out, 0) != hipSuccess))
        rc = fail(OFDM_ERR_NOMEM, "pinned host allocation failed");
    if (rc == OFDM_OK) rc = dev_alloc(&h->s_tsr, 4);
    if (rc == OFDM_OK) rc = dev_alloc(&h->s_H, size_t(2) * N);
    if (rc == OFDM_OK) rc = dev_alloc(&h->s_htime, size_t(2) * N);
    if (rc == OFDM_OK) rc = dev_alloc(&h->s_esf, size_t(2) * MM);
    if (rc == OFDM_OK) rc = dev_alloc(&h->s_eqg, size_t(Ks));
    if (rc == OFDM_OK) rc = dev_alloc(&h->s_gain, size_t(Kd));
    if (rc == OFDM_OK) rc = dev_alloc(&h->s_ysc, size_t(MM));
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_trial_m, size_t(ofdm_rx::TRIAL_CAP));
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_trial_d, size_t(ofdm_rx::TRIAL_CAP));
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_partial, size_t(DEMAP_PARTIALS));
    if (rc == OFDM_OK) {
        bool ok = hipMemcpy(h->d_tw, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(h->d_zc, zc.data(), zc.size() * sizeof(cf), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(h->d_zc + MM, zcp.data(), zcp.size() * sizeof(cf), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemset(h->d_edf, 0, rows * Kd * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->s_tsr, 0, 4 * sizeof(int)) == hipSuccess &&
                  hipMemset(h->s_H, 0, size_t(2) * N * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->s_htime, 0, size_t(2) * N * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->s_esf, 0, size_t(2) * MM * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->s_eqg, 0, size_t(Ks) * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->s_gain, 0, size_t(Kd) * sizeof(cf)) == hipSuccess;
        if (!ok) rc = fail(OFDM_ERR_HIP, "device table initialisation failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (rc != OFDM_OK) {
        std::string keep = g_last_error;
        ofdm_rx_destroy(h);
        g_last_error = keep;
        return rc;
    }
    d.tw = h->d_tw;
    d.zc = h->d_zc;
    d.zcp = h->d_zc + MM;
#ifdef OFDM_EXPERIMENTS
    if (const char* ev = std::getenv("OFDM_EXP_VARIANT")) h->variant = std::atoi(ev);   // run a whole test suite on one variant
#endif
    {   // work queue of the batch demod launch: two words, zero between launches (the kernel re-arms them itself)
        int rcq = dev_alloc(&h->d_work, 2);
        if (rcq == OFDM_OK && hipMemset(h->d_work, 0, 2 * sizeof(unsigned)) != hipSuccess) rcq = fail(OFDM_ERR_HIP, "work queue init failed");
        if (rcq != OFDM_OK) {
            std::string keep = g_last_error;
            ofdm_rx_destroy(h);
            g_last_error = keep;
            return rcq;
        }
        if (const char* e = std::getenv("OFDM_MI355X_DEMOD_QUEUE")) h->use_queue = std::atoi(e) != 0;   // 0: one chunk per workgroup (A/B)
    }
    h->scan_block = rx_sync_scan_block(d);
    if (h->scan_block > 0) {
        auto g = make_scan_table(N, Ks, zc);
        int rc2 = dev_alloc(&h->d_scan_g, g.size());
        if (rc2 == OFDM_OK && hipMemcpy(h->d_scan_g, g.data(), g.size() * sizeof(cf), hipMemcpyHostToDevice) != hipSuccess)
            rc2 = fail(OFDM_ERR_HIP, "scan table upload failed");
        const int seg0[2] = {0x7fffffff, 0};
        if (rc2 == OFDM_OK) rc2 = dev_alloc(&h->d_seg_state, 2);
        if (rc2 == OFDM_OK && hipMemcpy(h->d_seg_state, seg0, sizeof seg0, hipMemcpyHostToDevice) != hipSuccess)
            rc2 = fail(OFDM_ERR_HIP, "segment state upload failed");
        h->seg_armed = rc2 == OFDM_OK;
        if (rc2 != OFDM_OK) {
            std::string keep = g_last_error;
            ofdm_rx_destroy(h);
            g_last_error = keep;
            return rc2;
        }
    }
    *out = h;
    return OFDM_OK;
}

int ofdm_rx_reserve(ofdm_rx* h, int64_t n_frames) {
    if (!h || n_frames < 0) return fail(OFDM_ERR_INVALID, "ofdm_rx_reserve: bad argument");
    if (n_frames <= h->cap_frames) return OFDM_OK;
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipDeviceSynchronize());
    for (void* p : {(void*)h->f_tsr, (void*)h->f_H, (void*)h->f_gain, (void*)h->f_htime})
        if (p) (void)hipFree(p);
    h->f_tsr = nullptr;
    h->f_H = h->f_gain = h->f_htime = nullptr;
    h->cap_frames = 0;
    int rc = dev_alloc(&h->f_tsr, size_t(n_frames) * 4);
    if (rc == OFDM_OK) rc = dev_alloc(&h->f_H, size_t(n_frames) * h->dev.nfft);
    if (rc == OFDM_OK) rc = dev_alloc(&h->f_gain, size_t(n_frames) * h->dev.Kd);
    if (rc == OFDM_OK) rc = dev_alloc(&h->f_htime, size_t(n_frames) * h->dev.nfft);
    if (rc != OFDM_OK) return rc;
    h->cap_frames = n_frames;
    return OFDM_OK;
}

int64_t ofdm_rx_demod_frames(ofdm_rx* h, const float* d_iq, int64_t n_frames, int64_t frame_stride,
                             int64_t frame_len, float* d_eq, uint8_t* d_bits, int32_t bits_mode, int32_t* d_tsr,
                             void* stream) {
    if (!h || !d_iq || n_frames < 0 || frame_len < 0 || frame_stride < frame_len)
        return fail(OFDM_ERR_INVALID, "ofdm_rx_demod_frames: bad argument");
    const RxDev& d = h->dev;
    const int SD = d.S + d.D;
    const int64_t n_unique = frame_len / d.L;
    const int64_t n_pat = n_unique / SD;
    const int64_t n_dsym = n_pat * d.D;
    if (n_frames > INT32_MAX / 8 || n_dsym > INT32_MAX / 8) return fail(OFDM_ERR_INVALID, "batch too large");
    if (d_bits) {
        if (bits_mode != OFDM_BITS_PACKED && bits_mode != OFDM_BITS_UNPACKED)
            return fail(OFDM_ERR_INVALID, "bits_mode must be OFDM_BITS_PACKED or OFDM_BITS_UNPACKED");
        if (bits_mode == OFDM_BITS_PACKED && ((d.Kd & 3) || (d.bps & 1)))
            return fail(OFDM_ERR_INVALID, "packed bits need num_data_bins %% 4 == 0 and an even number of bits per symbol");
    }
    if (n_frames == 0) return n_dsym;
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (n_frames > h->cap_frames) {
        int rc = ofdm_rx_reserve(h, n_frames);
        if (rc != OFDM_OK) return rc;
    }
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;

    SyncArgs sa{};
    sa.iq = reinterpret_cast<const cf*>(d_iq);
    sa.frame_stride = frame_stride;
    sa.frame_len = frame_len;
    sa.n_frames = int(n_frames);
    sa.mode = 0;
    sa.p_begin = 0;
    sa.p_count = h->max_trials;
    sa.force_accept = 0;
    sa.tsr = h->f_tsr;
    sa.H = h->f_H;
    sa.H_for_gain = nullptr;
    sa.gain = h->f_gain;
    sa.htime = nullptr;                  // est_chan_time is computed on demand (ofdm_rx_get_frame_state)
    sa.scan_block = h->scan_block;       // screened search where its preconditions hold (same outcome as the exhaustive one)
    sa.scan_g = h->d_scan_g;
#ifdef OFDM_EXPERIMENTS
    sa.stamps = h->d_stamps;
#endif
    hipEvent_t* pev = h->ev + 3 * (h->prof_calls % ofdm_rx::PROF_RING);
    if (h->profiling) HIP_TRY(hipEventRecord(pev[0], s));
    HIP_TRY(launch_rx_sync(d, sa, s));
    if (h->profiling) HIP_TRY(hipEventRecord(pev[1], s));

    if (n_dsym > 0 && (d_eq || d_bits)) {
        DemodArgs da{};
        da.iq = sa.iq;
        da.frame_stride = frame_stride;
        da.frame_len = frame_len;
        da.n_frames = int(n_frames);
        da.tsr = h->f_tsr;
        da.gain = h->f_gain;
        da.eq = reinterpret_cast<cf*>(d_eq);
        da.bits = d_bits;
        da.bits_mode = bits_mode;
        da.mod = d.bps;
        da.n_dsym = int(n_dsym);
        da.spc = 0;                 // launcher picks the chunking (multiples of its slots per workgroup)
        da.chunks_per_frame = 0;
        da.row_stride_pat = d.D;
        da.rows_per_frame = int(n_dsym);
        da.zero_skipped = 1;
        da.variant = h->variant;
        da.stamps = h->d_stamps;
        da.work = h->use_queue ? h->d_work : nullptr;
        HIP_TRY(launch_rx_demod(d, da, s));
    }
    if (h->profiling) {
        HIP_TRY(hipEventRecord(pev[2], s));
        h->prof_calls += 1;
    }
    if (d_tsr) HIP_TRY(hipMemcpyAsync(d_tsr, h->f_tsr, size_t(n_frames) * 4 * sizeof(int), hipMemcpyDeviceToDevice, s));
    return n_dsym;
}

int ofdm_rx_get_frame_state(ofdm_rx* h, int64_t frame, float* h_chan_freq, float* h_gain, float* h_chan_time) {
    if (!h || frame < 0 || frame >= h->cap_frames) return fail(OFDM_ERR_INVALID, "ofdm_rx_get_frame_state: bad frame");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    const int N = h->dev.nfft, Kd = h->dev.Kd;
    if (h_chan_freq) HIP_TRY(hipMemcpy(h_chan_freq, h->f_H + frame * N, size_t(N) * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_gain) HIP_TRY(hipMemcpy(h_gain, h->f_gain + frame * Kd, size_t(Kd) * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_chan_time) {                   // ifft of the frame's est_chan_freq_P row, now (SynchAndChanEst.py:202,212)
        HIP_TRY(launch_rx_chan_time(h->dev, h->f_H + frame * N, h->f_htime + frame * N, 1, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy(h_chan_time, h->f_htime + frame * N, size_t(N) * sizeof(cf), hipMemcpyDeviceToHost));
    }
    return OFDM_OK;
}

int ofdm_rx_get_state(ofdm_rx* h, int32_t row, float* h_chan_freq, float* h_chan_time, float* h_synch_freq,
                      float* h_eq_gain, float* h_data_freq) {
    if (!h || row < 0 || row > 1) return fail(OFDM_ERR_INVALID, "ofdm_rx_get_state: bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int N = h->dev.nfft, Kd = h->dev.Kd, Ks = h->dev.Ks, MM = h->dev.MM;
    if (h_chan_freq) HIP_TRY(hipMemcpy(h_chan_freq, h->s_H + size_t(row) * N, size_t(N) * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_chan_time) HIP_TRY(hipMemcpy(h_chan_time, h->s_htime + size_t(row) * N, size_t(N) * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_synch_freq) HIP_TRY(hipMemcpy(h_synch_freq, h->s_esf + size_t(row) * MM, size_t(MM) * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_eq_gain) HIP_TRY(hipMemcpy(h_eq_gain, h->s_eqg, size_t(Ks) * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_data_freq)
        HIP_TRY(hipMemcpy(h_data_freq, h->d_edf, size_t(h->cfg.num_ofdm_symb) * Kd * sizeof(cf), hipMemcpyDeviceToHost));
    return OFDM_OK;
}

int64_t ofdm_rx_work(ofdm_rx* h, const float* h_in, int64_t n_in, float* h_out, int64_t n_out, ofdm_rx_report* rep) {
    if (!h || (!h_in && n_in > 0) || (!h_out && n_out > 0) || n_in < 0 || n_out < 0)
        return fail(OFDM_ERR_INVALID, "ofdm_rx_work: bad argument");
    const RxDev& d = h->dev;
    const int N = d.nfft, L = d.L, S = d.S, D = d.D, Kd = d.Kd, SD = S + D;
    HIP_TRY(hipSetDevice(h->cfg.device));
    hipStream_t s = h->stream;

    if (n_in > h->in_cap) {
        HIP_TRY(hipStreamSynchronize(s));
        if (h->d_in) (void)hipFree(h->d_in);
        h->d_in = nullptr;
        h->in_cap = 0;
        const int64_t cap = n_in + n_in / 4 + 1024;
        int rc = dev_alloc(&h->d_in, size_t(cap));
        if (rc != OFDM_OK) return rc;
        h->in_cap = cap;
    }
    // The upload is issued where the search is set up (below): the one-synchronisation path sends the head of the buffer first and
    // runs the first stage of the search under the rest of it.
    bool uploaded = n_in == 0;
    auto upload = [&](int64_t from, int64_t to) -> hipError_t {
        return hipMemcpyAsync(h->d_in + from, h_in + 2 * from, size_t(to - from) * sizeof(cf), hipMemcpyHostToDevice, s);
    };

    const int64_t n_unique = n_in / L;                                       // :140
    const int64_t n_data_symb = int64_t(double(n_unique) * (double(D) / double(SD)));   // :141 int(n * (D/(S+D)))

    // ---------------- Loop A: sliding sync search, first accepted trial wins (:143-219)
    int detected = 0, trials_run = 0;
    {
        // trial P is evaluated iff S*L + P*stride + N + cp < n_in (:144) and P < round(n_in/stride) (:139,143)
        const int64_t n_trials = int64_t(std::nearbyint(double(n_in) / double(d.stride)));
        const int64_t lim = n_in - (int64_t(S) * L + N + d.cp);             // P*stride < lim
        int64_t p_valid = lim > 0 ? (lim + d.stride - 1) / d.stride : 0;    // number of valid P: P < ceil(lim/stride)
        if (p_valid > n_trials) p_valid = n_trials;
        int64_t p0 = 0;
        if (h->corr_obs != -1) {
            // After the first call a trial can only be accepted if P*stride + cp - tsr0 > 2cp + N (:168): earlier trials
            // are evaluated by the reference but can never win, so they are skipped (identical outcome).
            const int64_t need = int64_t(h->tsr[0]) + d.cp + N;            // P*stride > need
            p0 = need >= 0 ? need / d.stride + 1 : 0;
        }
        // Screened search (same outcome as trial-by-trial, see rx_sync_scan_kernel): search, accept and finalize in ONE launch
        // straight into the state rows the accepted trial would get; a miss leaves the state as it was.  The trial range is
        // searched in parallel segments (a continuing stream finds its next sync ~3 symbols into the buffer, :168).  The row
        // past the estimate arrays (the reference's IndexError) is left to the path below, which raises it.
        if (h->scan_block > 0 && p0 < p_valid && p_valid < (int64_t(1) << 30) && h->corr_obs + 1 < h->cfg.num_ofdm_symb) {
            const int row = h->corr_obs + 1 > 1 ? 1 : h->corr_obs + 1;
            // ---- one-synchronisation path.  Nothing the host decides between the search and the output depends on the search's
            // result when (a) every data row any pattern could need exists (no IndexError whatever tsr0 turns out to be), (b) the
            // reshape of :255 and the output size are fine (both known up front): then the search, Loop B (its guard :223 is
            // evaluated on the device against the tsr the search leaves there), the row deletion and both copies are queued
            // back to back and the host waits ONCE.  Round 2 waited after the search, after Loop B and after the copy-out and
            // repacked the rows on the host: 0.37-0.40 ms per 240-symbol buffer, 0.09-0.10 ms per 4-symbol buffer.
            {
                const int rows_ = h->cfg.num_ofdm_symb;
                const int64_t n_pat_all = (n_unique + SD - 1) / SD;
                const bool rows_ok = n_pat_all == 0 || (n_pat_all - 1) * SD + D - 1 < rows_;
                int n_del_ = 0;
                for (int r = 3; r < rows_; r += SD) ++n_del_;
                const bool shape_ok = int64_t(rows_ - n_del_) == n_data_symb && (h->count == 0 || n_data_symb * Kd <= n_out);
                if (rows_ok && shape_ok && h->seg_armed) {
                    const size_t out_bytes = size_t(n_data_symb) * Kd * sizeof(cf);
                    const bool in_place_in = size_t(n_in) * sizeof(cf) <= ofdm_rx::PIN_IN_BYTES;
                    const bool in_place_out = out_bytes <= ofdm_rx::PIN_OUT_BYTES;
                    const cf* iq_dev = in_place_in ? h->pin_in_dev : h->d_in;
                    SyncArgs fa{};
                    fa.iq = iq_dev;
                    fa.frame_stride = n_in;
                    fa.frame_len = n_in;
                    fa.n_frames = 1;
                    fa.mode = 0;
                    fa.p_begin = int(p0);
                    fa.p_count = int(p_valid);
                    fa.keep_on_miss = 1;
                    fa.scan_block = h->scan_block;
                    fa.scan_g = h->d_scan_g;
                    fa.seg_len = h->scan_block;
                    fa.n_seg = int((p_valid - p0 + fa.seg_len - 1) / fa.seg_len);
                    fa.seg_state = h->d_seg_state;
                    fa.tsr = h->s_tsr;
                    fa.tsr_host = h->pin_tsr_dev;
                    fa.H = h->s_H + size_t(row) * N;
                    fa.H_for_gain = (row == 0) ? nullptr : h->s_H;                          // :242 always row 0
                    fa.gain = h->s_gain;
                    fa.htime = h->s_htime + size_t(row) * N;
                    fa.esf = h->s_esf + size_t(row) * d.MM;
                    fa.eqg = h->s_eqg;
                    fa.yscratch = h->s_ysc;
                    h->seg_armed = false;
                    // samples the first SYNC_STAGE_SEGS segments can touch: their last trial's windows and screening edges
                    const int64_t head = p0 + int64_t(SYNC_STAGE_SEGS) * fa.seg_len + int64_t(S) * L + N + d.cp + 64;
                    if (in_place_in) {
                        std::memcpy(h->pin_in, h_in, size_t(n_in) * sizeof(cf));
                        fa.seg_final = 1;
                        HIP_TRY(launch_rx_sync(d, fa, s));
                    } else if (fa.n_seg > 2 * SYNC_STAGE_SEGS && head < n_in / 2) {
                        HIP_TRY(upload(0, head));
                        HIP_TRY(hipEventRecord(h->ev_up, s));
                        HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_up, 0));
                        fa.seg_base = 0;
                        fa.seg_launch = SYNC_STAGE_SEGS;
                        fa.seg_final = 2;                                // a hit in the first stage is finalized under the upload too
                        HIP_TRY(launch_rx_sync(d, fa, h->stream2));
                        HIP_TRY(hipEventRecord(h->ev_s1, h->stream2));
                        HIP_TRY(upload(head, n_in));                     // (pageable source: the host is held here while stage 1 runs)
                        HIP_TRY(hipStreamWaitEvent(s, h->ev_s1, 0));
                        fa.seg_base = SYNC_STAGE_SEGS;
                        fa.seg_launch = fa.n_seg - SYNC_STAGE_SEGS;
                        fa.seg_final = 1;
                        HIP_TRY(launch_rx_sync(d, fa, s));
                    } else {
                        HIP_TRY(upload(0, n_in));
                        fa.seg_final = 1;
                        HIP_TRY(launch_rx_sync(d, fa, s));
                    }
                    uploaded = true;
                    const int64_t n_dsym_all = n_pat_all * D;                               // the device applies the guard per pattern
                    if (n_dsym_all > 0) {
                        DemodArgs da{};
                        da.iq = iq_dev;
                        da.frame_stride = n_in;
                        da.frame_len = n_in;
                        da.n_frames = 1;
                        da.tsr = h->s_tsr;
                        da.gain = h->s_gain;
                        da.eq = h->d_edf;
                        da.mod = d.bps;
                        da.n_dsym = int(n_dsym_all);
                        da.row_stride_pat = SD;
                        da.rows_per_frame = rows_;
                        da.zero_skipped = 0;                                                // a skipped pattern keeps its old rows (:223)
                        da.variant = h->variant;
                        HIP_TRY(launch_rx_demod(d, da, s));
                    }
                    if (h->count > 0) {                                                     // :257
                        HIP_TRY(launch_pack_rows(h->d_edf, rows_, Kd, SD, in_place_out ? h->pin_out_dev : h->d_pack, s));
                        if (!in_place_out) HIP_TRY(hipMemcpyAsync(h_out, h->d_pack, out_bytes, hipMemcpyDeviceToHost, s));
                    }
                    HIP_TRY(hipStreamSynchronize(s));                                       // the one wait of this call
                    if (h->count > 0 && in_place_out) std::memcpy(h_out, h->pin_out, out_bytes);
                    h->seg_armed = true;
                    const int* t4 = h->pin_tsr;
                    if (t4[3]) {
                        h->corr_obs += 1;                                                    // :171
                        h->tsr[0] = t4[0];                                                   // :173-175
                        h->tsr[1] = t4[1];
                        h->tsr[2] = t4[2];
                        detected = 1;
                        trials_run = int((int64_t(t4[0]) - d.cp) / d.stride - p0 + 1);
                    } else {
                        trials_run = int(p_valid - p0);
                    }
                    if (rep) {
                        rep->time_synch_ref[0] = h->tsr[0];
                        rep->time_synch_ref[1] = h->tsr[1];
                        rep->time_synch_ref[2] = h->tsr[2];
                        rep->detected = detected;
                        rep->trials_run = trials_run;
                        rep->n_data_items = n_data_symb * Kd;
                    }
                    h->count += 1;                                                           // :260
                    h->corr_obs = 0;                                                         // :261
                    if (rep) {
                        rep->count = h->count;
                        rep->corr_obs = h->corr_obs;
                    }
                    return n_out;                                                            // :262
                }
            }
            HIP_TRY(upload(0, n_in));
            uploaded = true;
            SyncArgs fa{};
            fa.iq = h->d_in;
            fa.frame_stride = n_in;
            fa.frame_len = n_in;
            fa.n_frames = 1;
            fa.mode = 0;
            fa.p_begin = int(p0);
            fa.p_count = int(p_valid);
            fa.keep_on_miss = 1;
            fa.scan_block = h->scan_block;
            fa.scan_g = h->d_scan_g;
            fa.seg_len = h->scan_block;                                                     // one screening block per workgroup
            fa.n_seg = int((p_valid - p0 + fa.seg_len - 1) / fa.seg_len);
            fa.seg_state = h->d_seg_state;
            fa.tsr = h->s_tsr;
            fa.H = h->s_H + size_t(row) * N;
            fa.H_for_gain = (row == 0) ? nullptr : h->s_H;                                  // :242 always row 0
            fa.gain = h->s_gain;
            fa.htime = h->s_htime + size_t(row) * N;
            fa.esf = h->s_esf + size_t(row) * d.MM;
            fa.eqg = h->s_eqg;
            fa.yscratch = h->s_ysc;
            if (!h->seg_armed) {                                       // an earlier search did not run to its end: start clean
                const int seg0[2] = {0x7fffffff, 0};
                HIP_TRY(hipStreamSynchronize(s));
                HIP_TRY(hipMemcpy(h->d_seg_state, seg0, sizeof seg0, hipMemcpyHostToDevice));
            }
            h->seg_armed = false;
            fa.seg_final = 1;
            HIP_TRY(launch_rx_sync(d, fa, s));
            int t4[4];
            HIP_TRY(hipMemcpyAsync(t4, h->s_tsr, sizeof(t4), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            h->seg_armed = true;
            if (t4[3]) {
                h->corr_obs += 1;                                                            // :171
                h->tsr[0] = t4[0];                                                           // :173-175
                h->tsr[1] = t4[1];
                h->tsr[2] = t4[2];
                detected = 1;
                trials_run = int((int64_t(t4[0]) - d.cp) / d.stride - p0 + 1);
            } else {
                trials_run = int(p_valid - p0);
            }
            p0 = p_valid;
        }
        if (!uploaded) {
            HIP_TRY(upload(0, n_in));
            uploaded = true;
        }
        int win = 128;
        std::vector<float> tm(ofdm_rx::TRIAL_CAP);
        std::vector<int> td(ofdm_rx::TRIAL_CAP);
        while (p0 < p_valid && !detected) {
            const int cnt = int(std::min<int64_t>(win, p_valid - p0));
            SyncArgs sa{};
            sa.iq = h->d_in;
            sa.frame_stride = n_in;
            sa.frame_len = n_in;
            sa.n_frames = 1;
            sa.mode = 1;
            sa.p_begin = int(p0);
            sa.p_count = cnt;
            sa.trial_m = h->d_trial_m;
            sa.trial_d = h->d_trial_d;
            HIP_TRY(launch_rx_sync(d, sa, s));
            HIP_TRY(hipMemcpyAsync(tm.data(), h->d_trial_m, size_t(cnt) * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(td.data(), h->d_trial_d, size_t(cnt) * sizeof(int), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            for (int w = 0; w < cnt; ++w) {
                const int64_t P = p0 + w;
                ++trials_run;
                if (tm[w] > d.gate_mm) {                                                    // :166
                    const double pos = double(P * d.stride + d.cp);
                    if (h->corr_obs == -1 || pos - h->tsr[0] > double(2 * d.cp + N)) {      // :168-169
                        // finalize this trial on the device: LS estimate, gains, est_chan_time (:171-218)
                        h->corr_obs += 1;                                                    // :171
                        const int row = h->corr_obs > 1 ? 1 : h->corr_obs;
                        if (h->corr_obs >= h->cfg.num_ofdm_symb)
                            return fail(OFDM_ERR_INDEX, "est_chan_freq_P has %d rows, corr_obs=%d (the reference raises IndexError)",
                                        h->cfg.num_ofdm_symb, h->corr_obs);
                        SyncArgs fa{};
                        fa.iq = h->d_in;
                        fa.frame_stride = n_in;
                        fa.frame_len = n_in;
                        fa.n_frames = 1;
                        fa.mode = 0;
                        fa.p_begin = int(P);
                        fa.p_count = 1;
                        fa.force_accept = 1;
                        fa.tsr = h->s_tsr;
                        fa.H = h->s_H + size_t(row) * N;
                        fa.H_for_gain = (row == 0) ? nullptr : h->s_H;                      // :242 always row 0
                        fa.gain = h->s_gain;
                        fa.htime = h->s_htime + size_t(row) * N;
                        fa.esf = h->s_esf + size_t(row) * d.MM;
                        fa.eqg = h->s_eqg;
                        fa.yscratch = h->s_ysc;
                        HIP_TRY(launch_rx_sync(d, fa, s));
                        int t4[4];
                        HIP_TRY(hipMemcpyAsync(t4, h->s_tsr, sizeof(t4), hipMemcpyDeviceToHost, s));
                        HIP_TRY(hipStreamSynchronize(s));
                        h->tsr[0] = t4[0];                                                   // :173-175
                        h->tsr[1] = t4[1];
                        h->tsr[2] = t4[2];
                        detected = 1;
                        break;                                                               // :219
                    }
                }
            }
            p0 += cnt;
            if (win < ofdm_rx::TRIAL_CAP) win *= 2;
        }
    }

    auto fill_report = [&]() {
        if (!rep) return;
        rep->time_synch_ref[0] = h->tsr[0];
        rep->time_synch_ref[1] = h->tsr[1];
        rep->time_synch_ref[2] = h->tsr[2];
        rep->detected = detected;
        rep->trials_run = trials_run;
        rep->count = h->count;
        rep->corr_obs = h->corr_obs;
        rep->n_data_items = n_data_symb * Kd;
    };
    fill_report();   // valid even if the call fails below, like the attributes the reference has already updated

    // ---------------- Loop B: data demod (:221-248)
    // The reference works symbol by symbol and raises IndexError from the first row that does not exist (:248) AFTER having
    // written every earlier row; the same rows are written here before the error is returned.  A window that starts past the
    // buffer is not an error: np.fft.fft(x, nfft) zero-pads even an empty slice (:230), the row becomes 0 * inf = NaN there
    // and here alike.
    const int64_t tsr0 = int64_t(h->tsr[0]);
    int64_t n_pat_loop = (n_unique + SD - 1) / SD;                           // range(n_unique)[::S+D]
    int64_t n_dsym_run = 0;                                                  // data symbols the reference gets through
    int loop_b_err = OFDM_OK;
    char loop_b_msg[160] = "";
    for (int64_t p = 0; p < n_pat_loop && loop_b_err == OFDM_OK; ++p) {
        const int64_t ptr = tsr0 + int64_t(S) * L * (p * SD + 1);            // :222
        if (!(ptr + N - 1 <= n_in)) continue;                                // :223 (monotonic: later patterns fail too)
        for (int n = 0; n < D; ++n) {
            if (p * SD + n >= h->cfg.num_ofdm_symb) {
                loop_b_err = OFDM_ERR_INDEX;
                snprintf(loop_b_msg, sizeof loop_b_msg, "est_data_freq has %d rows, pattern %lld needs row %lld (the reference raises IndexError)",
                         h->cfg.num_ofdm_symb, (long long)p, (long long)(p * SD + n));
                break;
            }
            n_dsym_run = p * D + n + 1;
        }
    }
    if (n_dsym_run > 0) {
        DemodArgs da{};
        da.iq = h->d_in;
        da.frame_stride = n_in;
        da.frame_len = n_in;
        da.n_frames = 1;
        da.tsr = h->s_tsr;
        da.gain = h->s_gain;
        da.eq = h->d_edf;
        da.bits = nullptr;
        da.bits_mode = 0;
        da.mod = d.bps;
        da.n_dsym = int(n_dsym_run);
        da.spc = 0;
        da.chunks_per_frame = 0;
        da.row_stride_pat = SD;
        da.rows_per_frame = h->cfg.num_ofdm_symb;
        da.zero_skipped = 0;
        da.variant = h->variant;
        HIP_TRY(launch_rx_demod(d, da, s));
    }
    if (loop_b_err != OFDM_OK) {
        HIP_TRY(hipStreamSynchronize(s));
        return fail(loop_b_err, "%s", loop_b_msg);
    }

    // ---------------- output packing (:249-262)
    const int rows = h->cfg.num_ofdm_symb;
    int n_del = 0;
    for (int r = 3; r < rows; r += SD) ++n_del;                              // :249 (literal 3)
    const int64_t kept = rows - n_del;
    if (kept * Kd != n_data_symb * Kd)                                       // :255 reshape
        return fail(OFDM_ERR_SHAPE, "cannot reshape %lld kept rows x %d bins into (1, %lld) (the reference raises ValueError)",
                    (long long)kept, Kd, (long long)(n_data_symb * Kd));
    if (h->count > 0) {                                                      // :257
        if (n_data_symb * Kd > n_out)
            return fail(OFDM_ERR_SHAPE, "output buffer holds %lld items, need %lld (the reference raises ValueError)",
                        (long long)n_out, (long long)(n_data_symb * Kd));
        std::vector<cf> host(size_t(rows) * Kd);
        HIP_TRY(hipMemcpyAsync(host.data(), h->d_edf, host.size() * sizeof(cf), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        cf* o = reinterpret_cast<cf*>(h_out);
        int64_t w = 0;
        for (int r = 0; r < rows; ++r) {
            if (r >= 3 && (r - 3) % SD == 0) continue;
            std::memcpy(o + w * Kd, host.data() + size_t(r) * Kd, size_t(Kd) * sizeof(cf));
            ++w;
        }
    } else {
        HIP_TRY(hipStreamSynchronize(s));
    }
    h->count += 1;                                                           // :260
    h->corr_obs = 0;                                                         // :261
    fill_report();
    return n_out;                                                            // :262
}

int ofdm_demap(ofdm_rx* h, const float* d_sym, int64_t n, int32_t modulation, uint8_t* d_hard, float* d_soft0,
               float* d_soft1, void* stream) {
    if (!h || (!d_sym && n > 0) || n < 0) return fail(OFDM_ERR_INVALID, "ofdm_demap: bad argument");
    if (modulation != 1 && modulation != 2 && modulation != 4 && modulation != 6)
        return fail(OFDM_ERR_INVALID, "modulation must be 1, 2, 4 or 6 bits per symbol");
    if ((d_soft0 || d_soft1) && modulation == 1)
        return fail(OFDM_ERR_INVALID, "soft metrics: QPSK (BitRecovery.py) and its 16/64-QAM extension only");
    HIP_TRY(hipSetDevice(h->cfg.device));
    DemapArgs a{};
    a.sym = reinterpret_cast<const cf*>(d_sym);
    a.n = n;
    a.mod = modulation;
    a.hard = d_hard;
    a.soft0 = d_soft0;
    a.soft1 = d_soft1;
    a.partial = h->d_partial;
    HIP_TRY(launch_demap(a, stream ? static_cast<hipStream_t>(stream) : h->stream));
    return OFDM_OK;
}

int ofdm_bandwidth_probe(int32_t device, const void* d_in, void* d_out, int64_t bytes, int32_t mode, int32_t sym_in_bytes,
                         int32_t gap_bytes, int32_t sym_out_bytes, int64_t n_sym, void* stream) {
    if (!d_in || !d_out || bytes < 0 || (bytes & 15) || (sym_in_bytes & 15) || (gap_bytes & 15) || (sym_out_bytes & 15))
        return fail(OFDM_ERR_INVALID, "ofdm_bandwidth_probe: sizes must be multiples of 16 bytes");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(launch_probe(d_in, d_out, bytes / 16, mode, sym_in_bytes / 16, gap_bytes / 16, sym_out_bytes / 16, n_sym,
                         static_cast<hipStream_t>(stream)));
    return OFDM_OK;
}

int ofdm_count_bit_errors(int32_t device, const uint8_t* d_a, const uint8_t* d_b, int64_t n_bytes, uint64_t* d_count, void* stream) {
    if (!d_count || n_bytes < 0 || (n_bytes > 0 && (!d_a || !d_b))) return fail(OFDM_ERR_INVALID, "ofdm_count_bit_errors: bad argument");
    if (n_bytes > (int64_t(1) << 40)) return fail(OFDM_ERR_INVALID, "ofdm_count_bit_errors: more than 2^40 bytes per call");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(launch_bit_errors(d_a, d_b, n_bytes, reinterpret_cast<unsigned long long*>(d_count), static_cast<hipStream_t>(stream)));
    return OFDM_OK;
}

// ------------------------------------------------------------------------------------------ TX
// ================================================================== CFO-search receiver
int ofdm_fo_destroy(ofdm_fo* h) {
    if (!h) return OFDM_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* ptrs[] = {h->d_tw, h->d_zc, h->d_rot, h->d_in, h->t_tsr, h->t_H, h->t_htime, h->t_esf, h->t_gain,
                    h->t_edf, h->d_code, h->t_edfd, h->s_eqg, h->s_ysc, h->d_trial_m, h->d_trial_d};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return OFDM_OK;
}

int ofdm_fo_create(const ofdm_fo_cfg* c, ofdm_fo** out) {
    if (!c || !out) return fail(OFDM_ERR_INVALID, "ofdm_fo_create: null argument");
    *out = nullptr;
    if (!supported_nfft(c->nfft)) return fail(OFDM_ERR_INVALID, "nfft=%d unsupported (64,128,...,4096)", c->nfft);
    if (c->cp_len < 2 || c->cp_len >= c->nfft) return fail(OFDM_ERR_INVALID, "cp_len=%d out of range (stride is cp_len-1)", c->cp_len);
    if (c->num_synch_bins < 2 || c->num_synch_bins > c->nfft || (c->num_synch_bins & 1))
        return fail(OFDM_ERR_INVALID, "num_synch_bins=%d must be even and in [2, nfft]", c->num_synch_bins);
    if (c->num_data_bins < 2 || c->num_data_bins > c->nfft || (c->num_data_bins & 1))
        return fail(OFDM_ERR_INVALID, "num_data_bins=%d must be even and in [2, nfft]", c->num_data_bins);
    if (c->synch_S < 1 || c->synch_D < 1) return fail(OFDM_ERR_INVALID, "synch_dat must be [>=1, >=1]");
    if (c->num_ofdm_symb < 1) return fail(OFDM_ERR_INVALID, "num_ofdm_symb must be >= 1");
    if (c->n_fo < 1 || (!c->rotators && c->n_fo != 1))
        return fail(OFDM_ERR_INVALID, "fo_range must hold at least one candidate (rotators may be NULL only with n_fo == 1)");
    if (!(c->snr > 0.0)) return fail(OFDM_ERR_INVALID, "snr must be > 0 (linear)");
    if (c->dsss < 0 || c->dsss > c->num_data_bins || (c->dsss > 0 && !c->spread_code))
        return fail(OFDM_ERR_INVALID, "dsss=%d must be 0 or in [1, num_data_bins] with a spreading code", c->dsss);

    HIP_TRY(hipSetDevice(c->device));
    ofdm_fo* h = new (std::nothrow) ofdm_fo();
    if (!h) return fail(OFDM_ERR_NOMEM, "out of host memory");
    h->cfg = *c;
    h->cfg.rotators = nullptr;            // the caller's tables are copied below, never kept
    h->cfg.spread_code = nullptr;
    h->n_spread = c->dsss > 0 ? c->num_data_bins / c->dsss : 0;
    ofdm_rx_cfg rc_cfg{};
    rc_cfg.num_ofdm_symb = c->num_ofdm_symb;
    rc_cfg.nfft = c->nfft;
    rc_cfg.cp_len = c->cp_len;
    rc_cfg.num_synch_bins = c->num_synch_bins;
    rc_cfg.synch_S = c->synch_S;
    rc_cfg.synch_D = c->synch_D;
    rc_cfg.num_data_bins = c->num_data_bins;
    rc_cfg.snr = c->snr;
    rc_cfg.compat = OFDM_COMPAT_RXOFDM;   // same constants: root 37 (FO:167), stride cp-1 (:196), gate 0.4 (:288), linear SNR
    rc_cfg.modulation = 2;
    RxDev& d = h->dev;
    const int root = fill_rxdev(rc_cfg, d);
    const int N = d.nfft, Ks = d.Ks, Kd = d.Kd, MM = d.MM;
    constexpr size_t R = OFDM_FO_MAX_SYNC;

    int rc = OFDM_OK;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) rc = fail(OFDM_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    auto tw = make_twiddles(N);
    auto zc = make_zc(MM, root, Ks);                                  // FO:168-176: parity of num_synch_bins
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_tw, size_t(N));
    auto zcp = rx_zc_lane_table(N, Ks, d.S, zc.data());
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_zc, size_t(MM) + zcp.size());
    if (rc == OFDM_OK && c->rotators) rc = dev_alloc(&h->d_rot, size_t(c->n_fo) * N);
    if (rc == OFDM_OK) rc = dev_alloc(&h->t_tsr, R * 4);
    if (rc == OFDM_OK) rc = dev_alloc(&h->t_H, R * N);
    if (rc == OFDM_OK) rc = dev_alloc(&h->t_htime, R * N);
    if (rc == OFDM_OK) rc = dev_alloc(&h->t_esf, R * MM);
    if (rc == OFDM_OK) rc = dev_alloc(&h->t_gain, R * Kd);
    if (rc == OFDM_OK) rc = dev_alloc(&h->t_edf, R * Kd);
    if (rc == OFDM_OK) rc = dev_alloc(&h->s_eqg, size_t(Ks));
    if (rc == OFDM_OK) rc = dev_alloc(&h->s_ysc, size_t(MM));
    if (rc == OFDM_OK && c->dsss > 0) rc = dev_alloc(&h->d_code, size_t(c->dsss));
    if (rc == OFDM_OK && c->dsss > 0) rc = dev_alloc(&h->t_edfd, R * size_t(h->n_spread));
    if (rc == OFDM_OK && c->dsss > 0) {
        bool ok = hipMemcpy(h->d_code, c->spread_code, size_t(c->dsss) * sizeof(cf), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemset(h->t_edfd, 0, R * size_t(h->n_spread) * sizeof(cf)) == hipSuccess;
        if (!ok) rc = fail(OFDM_ERR_HIP, "spreading-code upload failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_trial_m, size_t(c->n_fo) * ofdm_fo::TRIAL_WIN);
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_trial_d, size_t(c->n_fo) * ofdm_fo::TRIAL_WIN);
    if (rc == OFDM_OK) {
        bool ok = hipMemcpy(h->d_tw, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(h->d_zc, zc.data(), zc.size() * sizeof(cf), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(h->d_zc + MM, zcp.data(), zcp.size() * sizeof(cf), hipMemcpyHostToDevice) == hipSuccess &&
                  (!c->rotators ||
                   hipMemcpy(h->d_rot, c->rotators, size_t(c->n_fo) * N * sizeof(cf), hipMemcpyHostToDevice) == hipSuccess) &&
                  hipMemset(h->t_tsr, 0, R * 4 * sizeof(int)) == hipSuccess &&
                  hipMemset(h->t_H, 0, R * N * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->t_htime, 0, R * N * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->t_esf, 0, R * MM * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->t_gain, 0, R * Kd * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->t_edf, 0, R * Kd * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->s_eqg, 0, size_t(Ks) * sizeof(cf)) == hipSuccess;
        if (!ok) rc = fail(OFDM_ERR_HIP, "device table initialisation failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (rc != OFDM_OK) {
        std::string keep = g_last_error;
        ofdm_fo_destroy(h);
        g_last_error = keep;
        return rc;
    }
    d.tw = h->d_tw;
    d.zc = h->d_zc;
    d.zcp = h->d_zc + MM;
    *out = h;
    return OFDM_OK;
}

int64_t ofdm_fo_work(ofdm_fo* h, const float* h_in, int64_t n_in, float* h_out, int64_t n_out, ofdm_fo_report* rep) {
    if (!h || (!h_in && n_in > 0) || (!h_out && n_out > 0) || n_in < 0 || n_out < 0)
        return fail(OFDM_ERR_INVALID, "ofdm_fo_work: bad argument");
    const RxDev& d = h->dev;
    const int N = d.nfft, L = d.L, S = d.S, Kd = d.Kd, n_fo = h->cfg.n_fo;
    HIP_TRY(hipSetDevice(h->cfg.device));
    hipStream_t s = h->stream;

    if (n_in > h->in_cap) {
        HIP_TRY(hipStreamSynchronize(s));
        if (h->d_in) (void)hipFree(h->d_in);
        h->d_in = nullptr;
        h->in_cap = 0;
        const int64_t cap = n_in + n_in / 4 + 1024;
        int rc = dev_alloc(&h->d_in, size_t(cap));
        if (rc != OFDM_OK) return rc;
        h->in_cap = cap;
    }
    if (n_in > 0) HIP_TRY(hipMemcpyAsync(h->d_in, h_in, size_t(n_in) * sizeof(cf), hipMemcpyHostToDevice, s));

    int trials_run = 0;
    auto fill_report = [&](int n_sync) {
        if (!rep) return;
        rep->n_sync = n_sync;
        rep->count = h->count;
        rep->dmax_tmp_ind = h->dmax_tmp_ind;
        rep->trials_run = trials_run;
        rep->n_data_items = int64_t(h->cfg.num_ofdm_symb / (d.S + d.D)) * (h->cfg.dsss > 0 ? h->n_spread : Kd);
    };

    // ---------------- Loop A: every valid trial, every candidate; NO break (FO:248-329)
    {
        const int64_t n_trials = int64_t(std::nearbyint(double(n_in) / double(d.stride)));   // :246
        const int64_t lim = n_in - (int64_t(S) * L + N + d.cp);                              // :249  P*stride < lim
        int64_t p_valid = lim > 0 ? (lim + d.stride - 1) / d.stride : 0;
        if (p_valid > n_trials) p_valid = n_trials;
        std::vector<float> tm(size_t(n_fo) * ofdm_fo::TRIAL_WIN);
        std::vector<int> td(size_t(n_fo) * ofdm_fo::TRIAL_WIN);
        for (int64_t p0 = 0; p0 < p_valid; p0 += ofdm_fo::TRIAL_WIN) {
            const int cnt = int(std::min<int64_t>(ofdm_fo::TRIAL_WIN, p_valid - p0));
            SyncArgs sa{};
            sa.iq = h->d_in;
            sa.frame_stride = n_in;
            sa.frame_len = n_in;
            sa.n_frames = 1;
            sa.mode = 1;
            sa.p_begin = int(p0);
            sa.p_count = cnt;
            sa.trial_m = h->d_trial_m;
            sa.trial_d = h->d_trial_d;
            sa.rot = h->d_rot;
            sa.n_rot = n_fo;
            HIP_TRY(launch_rx_sync(d, sa, s));
            HIP_TRY(hipMemcpyAsync(tm.data(), h->d_trial_m, size_t(cnt) * n_fo * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(td.data(), h->d_trial_d, size_t(cnt) * n_fo * sizeof(int), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            for (int w = 0; w < cnt; ++w) {
                const int64_t P = p0 + w;
                ++trials_run;
                int best = 0;                                                                // :282-285 first maximum wins
                for (int fo = 1; fo < n_fo; ++fo)
                    if (tm[size_t(fo) * cnt + w] > tm[size_t(best) * cnt + w]) best = fo;
                const float dmax_val = tm[size_t(best) * cnt + w];
                const int dmax_ind = td[size_t(best) * cnt + w];
                h->dmax_tmp_ind = best;                                                      // :283 (state: the LAST trial's)
                if (!(dmax_val > d.gate_mm)) continue;                                       // :288
                const double pos = double(P * d.stride + d.cp);
                const double ref = h->tsr[h->cor_obs > 0 ? h->cor_obs : 0][0];               // :289
                if (!(pos - ref > double(2 * d.cp + N) || h->cor_obs == -1)) continue;       // :291
                h->cor_obs += 1;                                                             // :294
                if (h->cor_obs >= OFDM_FO_MAX_SYNC) {
                    fill_report(h->cor_obs);
                    return fail(OFDM_ERR_INDEX, "time_synch_ref has %d rows, cor_obs=%d (the reference raises IndexError)",
                                OFDM_FO_MAX_SYNC, h->cor_obs);
                }
                const int row = h->cor_obs;
                h->tsr[row][0] = pos;                                                        // :296-298
                h->tsr[row][1] = double(dmax_ind);
                h->tsr[row][2] = double(int(dmax_val));
                // LS estimate of this sync on the device: sync vector of the LAST candidate, lag of the best one (:300-329)
                SyncArgs fa{};
                fa.iq = h->d_in;
                fa.frame_stride = n_in;
                fa.frame_len = n_in;
                fa.n_frames = 1;
                fa.mode = 0;
                fa.p_begin = int(P);
                fa.p_count = 1;
                fa.force_accept = 1;
                fa.rot = h->d_rot ? h->d_rot + size_t(n_fo - 1) * N : nullptr;
                fa.n_rot = 1;
                fa.force_dhat_p1 = dmax_ind + 1;
                fa.tsr = h->t_tsr + size_t(row) * 4;
                fa.H = h->t_H + size_t(row) * N;
                fa.H_for_gain = nullptr;                                                     // :352 own row
                fa.gain = h->t_gain + size_t(row) * Kd;
                fa.htime = h->t_htime + size_t(row) * N;
                fa.esf = h->t_esf + size_t(row) * d.MM;
                fa.eqg = h->s_eqg;
                fa.yscratch = h->s_ysc;
                HIP_TRY(launch_rx_sync(d, fa, s));
            }
        }
    }
    const int n_sync = h->cor_obs + 1;
    fill_report(n_sync);

    // ---------------- Loop B: one data symbol per sync (FO:332-358)
    if (n_sync > 0) {
        for (int r = 0; r < n_sync; ++r) {
            const int64_t ptr = int64_t(h->tsr[r][0]) + int64_t(S) * L;                      // :335
            if (h->d_rot && ptr + N - 1 <= n_in && ptr + N > n_in)                           // :334 passes, slice has N-1 items
                return fail(OFDM_ERR_SHAPE, "data window of sync %d is one sample short (the reference raises ValueError)", r);
        }
        if (h->cfg.dsss > 0 && !(int64_t(h->tsr[0][0]) + int64_t(S) * L + N - 1 <= n_in))   // DS:362 fails for row 0 ...
            return fail(OFDM_ERR_UNBOUND, "row 0 fails the data guard before any row passed (the reference raises UnboundLocalError, "
                                          "SynchEstFOAndDSSS.py:392)");                       // ... rows >= 1 of this call always pass
        if (h->d_rot && h->dmax_tmp_ind < 0)
            return fail(OFDM_ERR_INVALID, "no trial has ever been evaluated: dmax_tmp_ind is undefined (the reference raises NameError)");
        DemodArgs da{};
        da.iq = h->d_in;
        da.frame_stride = 0;                    // every "frame" is the same buffer seen from another sync
        da.frame_len = n_in;
        da.n_frames = n_sync;
        da.tsr = h->t_tsr;
        da.gain = h->t_gain;
        da.eq = h->t_edf;
        da.bits = nullptr;
        da.bits_mode = 0;
        da.mod = 2;
        da.n_dsym = 1;
        da.spc = 0;
        da.chunks_per_frame = 0;
        da.row_stride_pat = 1;
        da.rows_per_frame = 1;
        da.zero_skipped = 0;
        da.rot = h->d_rot ? h->d_rot + size_t(h->dmax_tmp_ind) * N : nullptr;              // :339 (table mode: none)
        HIP_TRY(launch_rx_demod(d, da, s));
        if (h->cfg.dsss > 0)                                                                 // DS:391-399
            HIP_TRY(launch_despread(h->t_edf, Kd, h->d_code, h->cfg.dsss, h->n_spread, n_sync, h->t_edfd, s));
    }

    // ---------------- output (:362-367)
    const int64_t corr_size = h->cfg.num_ofdm_symb / (d.S + d.D);
    if (corr_size > OFDM_FO_MAX_SYNC)
        return fail(OFDM_ERR_SHAPE, "corr_size=%lld exceeds the %d est_data_freq rows (the reference raises ValueError)",
                    (long long)corr_size, OFDM_FO_MAX_SYNC);
    if (h->cfg.dsss > 0) {                                                                   // DS:403-407: every call
        if (corr_size * h->n_spread > n_out)
            return fail(OFDM_ERR_SHAPE, "output buffer holds %lld items, need %lld (the reference raises ValueError)",
                        (long long)n_out, (long long)(corr_size * h->n_spread));
        if (corr_size * h->n_spread > 0)
            HIP_TRY(hipMemcpyAsync(h_out, h->t_edfd, size_t(corr_size) * h->n_spread * sizeof(cf), hipMemcpyDeviceToHost, s));
    } else if (h->count > 0) {
        if (corr_size * Kd > n_out)
            return fail(OFDM_ERR_SHAPE, "output buffer holds %lld items, need %lld (the reference raises ValueError)",
                        (long long)n_out, (long long)(corr_size * Kd));
        HIP_TRY(hipMemcpyAsync(h_out, h->t_edf, size_t(corr_size) * Kd * sizeof(cf), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    h->count += 1;                                                                           // :368
    h->cor_obs = 0;                                                                          // :369
    fill_report(n_sync);
    return n_out;
}

int ofdm_fo_get_state(ofdm_fo* h, double* h_tsr, float* h_chan_freq, float* h_chan_time, float* h_synch_freq,
                      float* h_data_freq, float* h_eq_gain) {
    if (!h) return fail(OFDM_ERR_INVALID, "ofdm_fo_get_state: null handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const RxDev& d = h->dev;
    constexpr size_t R = OFDM_FO_MAX_SYNC;
    if (h_tsr) std::memcpy(h_tsr, h->tsr, sizeof(h->tsr));
    if (h_chan_freq) HIP_TRY(hipMemcpy(h_chan_freq, h->t_H, R * d.nfft * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_chan_time) HIP_TRY(hipMemcpy(h_chan_time, h->t_htime, R * d.nfft * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_synch_freq) HIP_TRY(hipMemcpy(h_synch_freq, h->t_esf, R * d.MM * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_data_freq) HIP_TRY(hipMemcpy(h_data_freq, h->t_edf, R * d.Kd * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_eq_gain) HIP_TRY(hipMemcpy(h_eq_gain, h->s_eqg, size_t(d.Ks) * sizeof(cf), hipMemcpyDeviceToHost));
    return OFDM_OK;
}

int ofdm_fo_get_despread(ofdm_fo* h, float* h_data_freq_d) {
    if (!h || !h_data_freq_d) return fail(OFDM_ERR_INVALID, "ofdm_fo_get_despread: null argument");
    if (h->cfg.dsss <= 0) return fail(OFDM_ERR_INVALID, "handle was not created with dsss >= 1");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(h_data_freq_d, h->t_edfd, size_t(OFDM_FO_MAX_SYNC) * h->n_spread * sizeof(cf), hipMemcpyDeviceToHost));
    return OFDM_OK;
}

// ================================================================== regression-tracking receiver (device primitives)
int ofdm_trk_destroy(ofdm_trk* h) {
    if (!h) return OFDM_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* ptrs[] = {h->d_tw, h->d_zc, h->d_in, h->t_tsr, h->t_H, h->t_imp, h->t_esf, h->t_gain, h->t_edf, h->s_ysc,
                    h->d_trial_m, h->d_trial_d};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return OFDM_OK;
}

int ofdm_trk_create(const ofdm_trk_cfg* c, ofdm_trk** out) {
    if (!c || !out) return fail(OFDM_ERR_INVALID, "ofdm_trk_create: null argument");
    *out = nullptr;
    if (!supported_nfft(c->nfft)) return fail(OFDM_ERR_INVALID, "nfft=%d unsupported (64,128,...,4096)", c->nfft);
    if (c->cp_len < 1 || c->cp_len >= c->nfft) return fail(OFDM_ERR_INVALID, "cp_len=%d out of range", c->cp_len);
    if (c->num_synch_bins < 2 || c->num_synch_bins > c->nfft || (c->num_synch_bins & 1))
        return fail(OFDM_ERR_INVALID, "num_synch_bins=%d must be even and in [2, nfft]", c->num_synch_bins);
    if (c->num_data_bins < 2 || c->num_data_bins > c->nfft || (c->num_data_bins & 1))
        return fail(OFDM_ERR_INVALID, "num_data_bins=%d must be even and in [2, nfft]", c->num_data_bins);
    if (c->synch_D < 1 || c->rows_sync < 1 || c->rows_data < 0) return fail(OFDM_ERR_INVALID, "bad pattern / row counts");
    if (!(c->snr > 0.0)) return fail(OFDM_ERR_INVALID, "snr must be > 0 (linear)");

    HIP_TRY(hipSetDevice(c->device));
    ofdm_trk* h = new (std::nothrow) ofdm_trk();
    if (!h) return fail(OFDM_ERR_NOMEM, "out of host memory");
    h->cfg = *c;
    RxDev& d = h->dev;
    const int N = c->nfft, Ks = c->num_synch_bins, Kd = c->num_data_bins;
    d.nfft = N;
    d.cp = c->cp_len;
    d.L = N + c->cp_len;
    d.Ks = Ks;
    d.Kd = Kd;
    d.S = 1;
    d.D = c->synch_D;
    d.MM = Ks;
    d.bps = 2;
    d.stride = 1;
    d.gate_mm = 0.f;
    d.inv_ls = float(1.0 / (1.0 + 1.0 / c->snr));                     // SynchronizeAndEstimate.py:349
    d.inv_snr_data = float(1.0 / c->snr);                              // :425
    d.inv_snr_eqsync = float(1.0 / c->snr);                            // :377
    const size_t RS = size_t(c->rows_sync), RD = size_t(c->rows_data > 0 ? c->rows_data : 1);

    int rc = OFDM_OK;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) rc = fail(OFDM_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    auto tw = make_twiddles(N);
    auto zc = make_zc(Ks, c->zc_root, Ks);                             // :123-130 parity of MM = Ks
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_tw, size_t(N));
    auto zcp = rx_zc_lane_table(N, Ks, 1, zc.data());
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_zc, size_t(Ks) + zcp.size());
    if (rc == OFDM_OK) rc = dev_alloc(&h->t_tsr, RS * 4);
    if (rc == OFDM_OK) rc = dev_alloc(&h->t_H, RS * N);
    if (rc == OFDM_OK) rc = dev_alloc(&h->t_imp, RS * N);
    if (rc == OFDM_OK) rc = dev_alloc(&h->t_esf, RS * Ks);
    if (rc == OFDM_OK) rc = dev_alloc(&h->t_gain, RS * Kd);
    if (rc == OFDM_OK) rc = dev_alloc(&h->t_edf, RD * Kd);
    if (rc == OFDM_OK) rc = dev_alloc(&h->s_ysc, size_t(Ks));
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_trial_m, size_t(ofdm_trk::TRIAL_CAP));
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_trial_d, size_t(ofdm_trk::TRIAL_CAP));
    if (rc == OFDM_OK) {
        bool ok = hipMemcpy(h->d_tw, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(h->d_zc, zc.data(), zc.size() * sizeof(cf), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(h->d_zc + Ks, zcp.data(), zcp.size() * sizeof(cf), hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemset(h->t_tsr, 0, RS * 4 * sizeof(int)) == hipSuccess &&
                  hipMemset(h->t_H, 0, RS * N * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->t_imp, 0, RS * N * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->t_esf, 0, RS * Ks * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->t_gain, 0, RS * Kd * sizeof(cf)) == hipSuccess &&
                  hipMemset(h->t_edf, 0, RD * Kd * sizeof(cf)) == hipSuccess;
        if (!ok) rc = fail(OFDM_ERR_HIP, "device table initialisation failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (rc != OFDM_OK) {
        std::string keep = g_last_error;
        ofdm_trk_destroy(h);
        g_last_error = keep;
        return rc;
    }
    d.tw = h->d_tw;
    d.zc = h->d_zc;
    d.zcp = h->d_zc + Ks;
    *out = h;
    return OFDM_OK;
}

int ofdm_trk_load(ofdm_trk* h, const float* h_in, int64_t n_in) {
    if (!h || (!h_in && n_in > 0) || n_in < 0) return fail(OFDM_ERR_INVALID, "ofdm_trk_load: bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (n_in > h->in_cap) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->d_in) (void)hipFree(h->d_in);
        h->d_in = nullptr;
        h->in_cap = 0;
        const int64_t cap = n_in + n_in / 4 + 1024;
        int rc = dev_alloc(&h->d_in, size_t(cap));
        if (rc != OFDM_OK) return rc;
        h->in_cap = cap;
    }
    if (n_in > 0) HIP_TRY(hipMemcpyAsync(h->d_in, h_in, size_t(n_in) * sizeof(cf), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));          // the caller's buffer is only valid during work()
    h->n_in = n_in;
    return OFDM_OK;
}

int ofdm_trk_trials(ofdm_trk* h, int64_t first_ptr, int32_t step, int32_t count, float* h_peak, int32_t* h_lag) {
    if (!h || !h_peak || !h_lag || count < 0 || step < 1 || first_ptr < 0)
        return fail(OFDM_ERR_INVALID, "ofdm_trk_trials: bad argument");
    if (count == 0) return OFDM_OK;
    const RxDev& d0 = h->dev;
    if (first_ptr + int64_t(count - 1) * step + d0.nfft > h->n_in)
        return fail(OFDM_ERR_INVALID, "window %lld..+%d reaches past the loaded buffer (%lld samples)",
                    (long long)(first_ptr + int64_t(count - 1) * step), d0.nfft, (long long)h->n_in);
    HIP_TRY(hipSetDevice(h->cfg.device));
    RxDev d = d0;
    d.stride = step;
    for (int32_t done = 0; done < count; done += ofdm_trk::TRIAL_CAP) {
        const int cnt = std::min<int>(ofdm_trk::TRIAL_CAP, count - done);
        SyncArgs sa{};
        sa.iq = h->d_in;
        sa.frame_stride = h->n_in;
        sa.frame_len = h->n_in;
        sa.n_frames = 1;
        sa.mode = 1;
        sa.p_begin = done;
        sa.p_count = cnt;
        sa.off_delta = int(first_ptr) - d.cp;                         // window start = first_ptr + P*step
        sa.host_valid = 1;
        sa.trial_m = h->d_trial_m;
        sa.trial_d = h->d_trial_d;
        HIP_TRY(launch_rx_sync(d, sa, h->stream));
        HIP_TRY(hipMemcpyAsync(h_peak + done, h->d_trial_m, size_t(cnt) * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(h_lag + done, h->d_trial_d, size_t(cnt) * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return OFDM_OK;
}

int ofdm_trk_accept(ofdm_trk* h, int32_t row, int64_t window_ptr, int32_t lag_sync, int32_t lag_data) {
    if (!h || row < 0 || window_ptr < 0) return fail(OFDM_ERR_INVALID, "ofdm_trk_accept: bad argument");
    const RxDev& d = h->dev;
    if (row >= h->cfg.rows_sync)
        return fail(OFDM_ERR_INDEX, "est_chan_freq_p has %d rows, corr_obs=%d (the reference raises IndexError)", h->cfg.rows_sync, row);
    if (lag_sync < 0 || lag_sync > d.cp || window_ptr + d.nfft > h->n_in) return fail(OFDM_ERR_INVALID, "ofdm_trk_accept: lag / window out of range");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int N = d.nfft;
    SyncArgs fa{};
    fa.iq = h->d_in;
    fa.frame_stride = h->n_in;
    fa.frame_len = h->n_in;
    fa.n_frames = 1;
    fa.mode = 0;
    fa.p_begin = 0;
    fa.p_count = 1;
    fa.force_accept = 1;
    fa.host_valid = 1;
    fa.off_delta = int(window_ptr) - d.cp;
    fa.force_dhat_p1 = lag_sync + 1;
    fa.gain_lag_set = 1;
    fa.gain_lag = lag_data;
    fa.tsr = h->t_tsr + size_t(row) * 4;
    fa.H = h->t_H + size_t(row) * N;
    fa.gain = h->t_gain + size_t(row) * d.Kd;
    fa.htime = h->t_imp + size_t(row) * N;
    fa.esf = h->t_esf + size_t(row) * d.MM;
    fa.yscratch = h->s_ysc;
    HIP_TRY(launch_rx_sync(d, fa, h->stream));
    return OFDM_OK;
}

int ofdm_trk_demod(ofdm_trk* h, int32_t n_sync, const int64_t* h_ptr, const uint8_t* h_guard, float* h_last, int32_t* last_row) {
    if (!h || n_sync < 0 || (n_sync > 0 && (!h_ptr || !h_guard))) return fail(OFDM_ERR_INVALID, "ofdm_trk_demod: bad argument");
    if (last_row) *last_row = -1;
    if (n_sync == 0) return OFDM_OK;
    if (n_sync > h->cfg.rows_sync) return fail(OFDM_ERR_INDEX, "%d syncs, %d rows", n_sync, h->cfg.rows_sync);
    const RxDev& d = h->dev;
    const int D = d.D, Kd = d.Kd;
    HIP_TRY(hipSetDevice(h->cfg.device));
    std::vector<int> tsr(size_t(n_sync) * 4, 0);
    int last = -1;
    for (int p = 0; p < n_sync; ++p) {
        if (h_guard[p] && (h_ptr[p] < 0 || h_ptr[p] > INT32_MAX - int64_t(d.S + D) * d.L))
            // a guarded window that starts before the buffer: the reference's slice would wrap / be empty and the block
            // mirror raises IndexError for it; never hand a negative start to the kernel
            return fail(OFDM_ERR_INDEX, "sync %d: window pointer %lld outside the buffer", p, (long long)h_ptr[p]);
        tsr[size_t(p) * 4 + 0] = h_guard[p] ? int(h_ptr[p]) : 0;
        tsr[size_t(p) * 4 + 3] = h_guard[p] ? 1 : 0;
        if (!h_guard[p]) continue;
        if (p * D + D - 1 >= h->cfg.rows_data)
            return fail(OFDM_ERR_INDEX, "est_data_freq has %d rows, sync %d needs row %d (the reference raises IndexError)",
                        h->cfg.rows_data, p, p * D + D - 1);
        last = p * D + D - 1;
    }
    HIP_TRY(hipMemcpyAsync(h->t_tsr, tsr.data(), tsr.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    DemodArgs da{};
    da.iq = h->d_in;
    da.frame_stride = 0;
    da.frame_len = h->n_in;
    da.n_frames = n_sync;
    da.tsr = h->t_tsr;
    da.gain = h->t_gain;
    da.eq = h->t_edf;
    da.bits = nullptr;
    da.bits_mode = 0;
    da.mod = 2;
    da.n_dsym = D;
    da.spc = 0;
    da.chunks_per_frame = 0;
    da.row_stride_pat = D;
    da.rows_per_frame = D;
    da.zero_skipped = 0;
    da.host_guard = 1;
    HIP_TRY(launch_rx_demod(d, da, h->stream));
    HIP_TRY(launch_row_renorm(h->t_edf, Kd, D, n_sync, h->t_tsr, h->stream));
    if (h_last && last >= 0)
        HIP_TRY(hipMemcpyAsync(h_last, h->t_edf + size_t(last) * Kd, size_t(Kd) * sizeof(cf), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));          // tsr (host vector) must stay alive until the copy has run
    if (last_row) *last_row = last;
    return OFDM_OK;
}

int ofdm_trk_get_state(ofdm_trk* h, float* h_chan_freq, float* h_chan_impulse, float* h_synch_freq, float* h_data_freq) {
    if (!h) return fail(OFDM_ERR_INVALID, "ofdm_trk_get_state: null handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const RxDev& d = h->dev;
    const size_t RS = size_t(h->cfg.rows_sync), RD = size_t(h->cfg.rows_data);
    if (h_chan_freq) HIP_TRY(hipMemcpy(h_chan_freq, h->t_H, RS * d.nfft * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_chan_impulse) HIP_TRY(hipMemcpy(h_chan_impulse, h->t_imp, RS * d.nfft * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_synch_freq) HIP_TRY(hipMemcpy(h_synch_freq, h->t_esf, RS * d.Ks * sizeof(cf), hipMemcpyDeviceToHost));
    if (h_data_freq && RD > 0) HIP_TRY(hipMemcpy(h_data_freq, h->t_edf, RD * d.Kd * sizeof(cf), hipMemcpyDeviceToHost));
    return OFDM_OK;
}

// ================================================================== TX
int ofdm_tx_destroy(ofdm_tx* h) {
    if (!h) return OFDM_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->d_tw) (void)hipFree(h->d_tw);
    if (h->d_zc) (void)hipFree(h->d_zc);
    if (h->d_sync_time) (void)hipFree(h->d_sync_time);
    if (h->d_pilots) (void)hipFree(h->d_pilots);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return OFDM_OK;
}

int ofdm_tx_create(const ofdm_tx_cfg* c, ofdm_tx** out) {
    if (!c || !out) return fail(OFDM_ERR_INVALID, "ofdm_tx_create: null argument");
    *out = nullptr;
    if (!supported_nfft(c->nfft)) return fail(OFDM_ERR_INVALID, "nfft=%d unsupported (64,128,...,4096)", c->nfft);
    if (c->cp_len < 0 || c->cp_len >= c->nfft) return fail(OFDM_ERR_INVALID, "cp_len=%d out of range", c->cp_len);
    if (c->num_synch_bins < 2 || c->num_synch_bins > c->nfft || (c->num_synch_bins & 1) || c->num_data_bins < 2 ||
        c->num_data_bins > c->nfft || (c->num_data_bins & 1))
        return fail(OFDM_ERR_INVALID, "bin counts must be even and in [2, nfft]");
    if (c->synch_S < 1 || c->synch_D < 1) return fail(OFDM_ERR_INVALID, "synch_dat must be [>=1, >=1]");
    if (c->modulation != 1 && c->modulation != 2 && c->modulation != 4 && c->modulation != 6)
        return fail(OFDM_ERR_INVALID, "modulation must be 1, 2, 4 or 6 bits per symbol");
    HIP_TRY(hipSetDevice(c->device));
    ofdm_tx* h = new (std::nothrow) ofdm_tx();
    if (!h) return fail(OFDM_ERR_NOMEM, "out of host memory");
    h->cfg = *c;
    const int N = c->nfft, MM = c->synch_S * c->num_synch_bins;
    auto tw = make_twiddles(N);
    auto zc = make_zc(MM, c->zc_root ? c->zc_root : 23, MM);
    int rc = OFDM_OK;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) rc = fail(OFDM_ERR_HIP, "hipStreamCreate failed");
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_tw, size_t(N));
    if (rc == OFDM_OK) rc = dev_alloc(&h->d_zc, size_t(MM));
    if (rc == OFDM_OK &&
        (hipMemcpy(h->d_tw, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice) != hipSuccess ||
         hipMemcpy(h->d_zc, zc.data(), zc.size() * sizeof(cf), hipMemcpyHostToDevice) != hipSuccess))
        rc = fail(OFDM_ERR_HIP, "device table initialisation failed");
    if (rc != OFDM_OK) {
        std::string keep = g_last_error;
        ofdm_tx_destroy(h);
        g_last_error = keep;
        return rc;
    }
    TxDev& d = h->dev;
    d.nfft = N;
    d.cp = c->cp_len;
    d.L = N + c->cp_len;
    d.Ks = c->num_synch_bins;
    d.Kd = c->num_data_bins;
    d.S = c->synch_S;
    d.D = c->synch_D;
    d.bps = c->modulation;
    d.tw = h->d_tw;
    d.zc = h->d_zc;
    // the sync symbol(s) of SynchDataMux: ZC grid rows -> IFFT + CP + normalise, once (same kernels as the data symbols)
    {
        cf* d_grid = nullptr;
        rc = dev_alloc(&d_grid, size_t(d.S) * N);
        if (rc == OFDM_OK) rc = dev_alloc(&h->d_sync_time, size_t(d.S) * d.L);
        if (rc == OFDM_OK) {
            TimeArgs ta{};
            ta.in = d_grid;
            ta.n_rows = d.S;
            ta.do_ifft = 1;
            ta.do_cp = 1;
            ta.out = h->d_sync_time;
            if (launch_tx_sync_grid(d, d_grid, h->stream) != hipSuccess || launch_tx_time(d, ta, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess)
                rc = fail(OFDM_ERR_HIP, "sync-symbol synthesis failed: %s", hipGetErrorString(hipGetLastError()));
        }
        if (d_grid) (void)hipFree(d_grid);
        if (rc != OFDM_OK) {
            std::string keep = g_last_error;
            ofdm_tx_destroy(h);
            g_last_error = keep;
            return rc;
        }
    }
    *out = h;
    return OFDM_OK;
}

// ---- decomposed stages
int ofdm_tx_random_bits(ofdm_tx* h, uint64_t seed, uint64_t offset, uint8_t* d_bits, int64_t n_bits, void* stream) {
    if (!h || (!d_bits && n_bits > 0) || n_bits < 0) return fail(OFDM_ERR_INVALID, "ofdm_tx_random_bits: bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(launch_tx_random_bits(seed, offset, d_bits, n_bits, stream ? static_cast<hipStream_t>(stream) : h->stream));
    return OFDM_OK;
}

int ofdm_tx_map(ofdm_tx* h, const uint8_t* d_bits, int32_t bits_mode, int64_t n_symbols, float* d_sym, void* stream) {
    if (!h || n_symbols < 0 || (n_symbols > 0 && (!d_bits || !d_sym))) return fail(OFDM_ERR_INVALID, "ofdm_tx_map: bad argument");
    if (bits_mode != OFDM_BITS_PACKED && bits_mode != OFDM_BITS_UNPACKED)
        return fail(OFDM_ERR_INVALID, "bits_mode must be OFDM_BITS_PACKED or OFDM_BITS_UNPACKED");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(launch_tx_map(d_bits, bits_mode, h->dev.bps, n_symbols, reinterpret_cast<cf*>(d_sym),
                          stream ? static_cast<hipStream_t>(stream) : h->stream));
    return OFDM_OK;
}

int ofdm_tx_set_pilots(ofdm_tx* h, const int32_t* h_locations, int32_t n_pilots, float pilot_re, float pilot_im) {
    if (!h || n_pilots < 0 || (n_pilots > 0 && !h_locations)) return fail(OFDM_ERR_INVALID, "ofdm_tx_set_pilots: bad argument");
    const int K = h->dev.Kd + n_pilots;
    if ((K & 1) || K > h->dev.nfft)
        return fail(OFDM_ERR_INVALID, "num_data_bins + pilots = %d must be even and <= nfft", K);
    std::vector<int> idx(size_t(n_pilots), 0);
    for (int p = 0; p < n_pilots; ++p) {
        const int loc = h_locations[p];
        if (loc == 0 || loc < -(K / 2) || loc > K / 2)
            return fail(OFDM_ERR_INVALID, "pilot location %d outside the occupied bins [-%d..-1, 1..%d]", loc, K / 2, K / 2);
        idx[size_t(p)] = loc < 0 ? loc + K / 2 : K / 2 + loc - 1;
    }
    std::sort(idx.begin(), idx.end());
    for (int p = 1; p < n_pilots; ++p)
        if (idx[size_t(p)] == idx[size_t(p - 1)]) return fail(OFDM_ERR_INVALID, "pilot locations must be distinct");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipDeviceSynchronize());
    if (h->d_pilots) (void)hipFree(h->d_pilots);
    h->d_pilots = nullptr;
    h->n_pilots = 0;
    if (n_pilots > 0) {
        int rc = dev_alloc(&h->d_pilots, size_t(n_pilots));
        if (rc != OFDM_OK) return rc;
        HIP_TRY(hipMemcpy(h->d_pilots, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    h->n_pilots = n_pilots;
    h->pilot_value = cf{pilot_re, pilot_im};
    return OFDM_OK;
}

int ofdm_tx_grid(ofdm_tx* h, const float* d_sym, int64_t n_rows, float* d_grid, void* stream) {
    if (!h || n_rows < 0 || (n_rows > 0 && (!d_sym || !d_grid))) return fail(OFDM_ERR_INVALID, "ofdm_tx_grid: bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    GridArgs a{};
    a.sym = reinterpret_cast<const cf*>(d_sym);
    a.n_rows = n_rows;
    a.pilots = h->d_pilots;
    a.n_pilots = h->n_pilots;
    a.pilot_value = h->pilot_value;
    a.grid = reinterpret_cast<cf*>(d_grid);
    HIP_TRY(launch_tx_grid(h->dev, a, stream ? static_cast<hipStream_t>(stream) : h->stream));
    return OFDM_OK;
}

int ofdm_tx_ifft_cp(ofdm_tx* h, const float* d_in, int64_t n_rows, int32_t do_ifft, int32_t add_cp, float* d_out, void* stream) {
    if (!h || n_rows < 0 || (n_rows > 0 && (!d_in || !d_out))) return fail(OFDM_ERR_INVALID, "ofdm_tx_ifft_cp: bad argument");
    if (!do_ifft && !add_cp) return fail(OFDM_ERR_INVALID, "ofdm_tx_ifft_cp: nothing to do (do_ifft = add_cp = 0)");
    if (n_rows > INT32_MAX) return fail(OFDM_ERR_INVALID, "batch too large");
    HIP_TRY(hipSetDevice(h->cfg.device));
    TimeArgs a{};
    a.in = reinterpret_cast<const cf*>(d_in);
    a.n_rows = n_rows;
    a.do_ifft = do_ifft != 0;
    a.do_cp = add_cp != 0;
    a.out = reinterpret_cast<cf*>(d_out);
    HIP_TRY(launch_tx_time(h->dev, a, stream ? static_cast<hipStream_t>(stream) : h->stream));
    return OFDM_OK;
}

int64_t ofdm_tx_mux(ofdm_tx* h, const float* d_data, int64_t n_data_sym, float* d_out, void* stream) {
    if (!h || n_data_sym < 0 || (n_data_sym > 0 && (!d_data || !d_out))) return fail(OFDM_ERR_INVALID, "ofdm_tx_mux: bad argument");
    const TxDev& d = h->dev;
    const int64_t full = n_data_sym / d.D, rem = n_data_sym % d.D;
    const int64_t n_out = full * (d.S + d.D) + (rem ? d.S + rem : 0);
    if (n_out == 0) return 0;
    HIP_TRY(hipSetDevice(h->cfg.device));
    MuxArgs a{};
    a.sync_time = h->d_sync_time;
    a.data = reinterpret_cast<const cf*>(d_data);
    a.n_out_sym = n_out;
    a.out = reinterpret_cast<cf*>(d_out);
    HIP_TRY(launch_tx_mux(d, a, stream ? static_cast<hipStream_t>(stream) : h->stream));
    return n_out;
}

int ofdm_tx_get_sync_symbol(ofdm_tx* h, float* h_sync_time) {
    if (!h || !h_sync_time) return fail(OFDM_ERR_INVALID, "ofdm_tx_get_sync_symbol: null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipMemcpy(h_sync_time, h->d_sync_time, size_t(h->dev.S) * h->dev.L * sizeof(cf), hipMemcpyDeviceToHost));
    return OFDM_OK;
}

int ofdm_tx_modulate_frames(ofdm_tx* h, const uint8_t* d_bits, int32_t bits_mode, int64_t n_frames, int32_t n_sym,
                            float* d_iq, int64_t frame_stride, void* stream) {
    if (!h || !d_iq || n_frames < 0 || n_sym < 0) return fail(OFDM_ERR_INVALID, "ofdm_tx_modulate_frames: bad argument");
    if (bits_mode != OFDM_BITS_PACKED && bits_mode != OFDM_BITS_UNPACKED)
        return fail(OFDM_ERR_INVALID, "bits_mode must be OFDM_BITS_PACKED or OFDM_BITS_UNPACKED");
    const TxDev& d = h->dev;
    if (frame_stride < int64_t(n_sym) * d.L) return fail(OFDM_ERR_INVALID, "frame_stride shorter than n_sym*(nfft+cp)");
    if (n_frames * int64_t(n_sym) > INT32_MAX) return fail(OFDM_ERR_INVALID, "batch too large");
    const int SD = d.S + d.D;
    int64_t n_data = int64_t(n_sym / SD) * d.D;
    const int rem = n_sym % SD;
    if (rem > d.S) n_data += rem - d.S;
    const int64_t bits_per_frame = n_data * d.Kd * d.bps;
    if (bits_mode == OFDM_BITS_PACKED && (bits_per_frame & 7))
        return fail(OFDM_ERR_INVALID, "packed bits need a whole number of bytes per frame");
    if (bits_per_frame > INT32_MAX) return fail(OFDM_ERR_INVALID, "more than 2^31 bits per frame");
    HIP_TRY(hipSetDevice(h->cfg.device));
    ModArgs a{};
    a.bits = d_bits;
    a.bits_mode = bits_mode;
    a.bits_stride = bits_mode == OFDM_BITS_PACKED ? bits_per_frame / 8 : bits_per_frame;
    a.n_frames = int(n_frames);
    a.n_sym = n_sym;
    a.iq = reinterpret_cast<cf*>(d_iq);
    a.frame_stride = frame_stride;
    a.sync_time = h->d_sync_time;
    HIP_TRY(launch_tx_modulate(d, a, stream ? static_cast<hipStream_t>(stream) : h->stream));
    return OFDM_OK;
}

int ofdm_channel_apply(ofdm_tx* h, const float* d_in, int64_t n_frames, int64_t in_stride, int64_t in_len,
                       const float* d_taps, int32_t n_taps, int32_t per_frame_taps, float noise_var, uint64_t seed,
                       float* d_out, int64_t out_stride, int64_t out_len, void* stream) {
    if (!h || !d_in || !d_out || !d_taps || n_taps < 1 || n_frames < 0 || in_len < 0 || out_len < 0 || noise_var < 0.f)
        return fail(OFDM_ERR_INVALID, "ofdm_channel_apply: bad argument");
    if (out_len > in_len + n_taps - 1) return fail(OFDM_ERR_INVALID, "out_len exceeds the convolution length");
    if (in_stride < in_len || out_stride < out_len) return fail(OFDM_ERR_INVALID, "stride shorter than length");
    if (n_frames > 65535) return fail(OFDM_ERR_INVALID, "at most 65535 frames per call");
    HIP_TRY(hipSetDevice(h->cfg.device));
    ChanArgs a{};
    a.in = reinterpret_cast<const cf*>(d_in);
    a.in_stride = in_stride;
    a.in_len = in_len;
    a.n_frames = int(n_frames);
    a.taps = reinterpret_cast<const cf*>(d_taps);
    a.n_taps = n_taps;
    a.per_frame_taps = per_frame_taps;
    a.noise_std = std::sqrt(noise_var / 2.f);
    a.seed = seed;
    a.out = reinterpret_cast<cf*>(d_out);
    a.out_stride = out_stride;
    a.out_len = out_len;
    HIP_TRY(launch_channel(a, stream ? static_cast<hipStream_t>(stream) : h->stream));
    return OFDM_OK;
}

}  // extern "C"
