// rx_kernels.hip -- gfx950 kernels of the OFDM receive path.
//
//   rx_sync_kernel   Zadoff-Chu lag-correlation timing search + LS channel estimate + equaliser gains
//                    (reference: gr-utsa_ofdm/python/SynchAndChanEst.py:143-219)
//   rx_demod_kernel  CP strip + N-point FFT + data-bin de-map + per-symbol power normalisation +
//                    lag de-rotation + one-tap MMSE equalise (+ fused hard de-map)   (:221-248)
//   demap kernels    BitRecovery hard / max-log soft outputs (LEGACY/gr-ofdm-rx/python/BitRecovery.py:66-157)
//
// Layout: one OFDM symbol is owned by T = N/16 lanes (fft_core.hpp); a workgroup holds max(T,64)
// lanes = SLOTS symbols side by side.  A symbol is read from HBM exactly once (coalesced, CP skipped
// by offset), lives in VGPRs/LDS through the FFT, and leaves as one coalesced 16 B/lane store of
// its Kd equalised bins (+ packed bits).  No MFMA: the path is FFT + elementwise, HBM-bound.
#include "rx_demod.hpp"

#ifndef OFDM_SCAN_MINW
#define OFDM_SCAN_MINW 2        // waves per SIMD the screened sync search is compiled for (3: a 168-VGPR budget)
#endif

namespace ofdm {

// ------------------------------------------------------------------------------------------ sync
// One sync trial P of one frame (SynchAndChanEst.py:145-164).  On return: Z (per-lane bins, register
// slot order) = sum over the S sync symbols of Y[k]*conj(zc), zdup = negative-half part of a bin that
// is listed twice (K == N only), p_est, m = max|del_mat|, dhat = argmax lag.
// KEEP > 0 (scan kernel, S == 1): additionally returns, per lane, the COMPLEX lag correlations c[a] for a = t + T*q, q < KEEP
// (ukeep), the in-band energy sum |Y|^2 (ekeep) and -- in red[16..19] -- the DC and Nyquist bins of the window's FFT.
template <int N, class TW, int KEEP = 0>
__device__ __forceinline__ void sync_trial(const RxDev& rx, const cf* frame_iq, int64_t frame_len, bool active,
                                           int Ptrial, cf* lds, float* red, const TW& tw, const cf* w1tab, int t,
                                           cf (&Z)[Plan<N>::P], cf& zdup, float& p_est, float& m, int& dhat,
                                           cf* yscratch, const cf* rot = nullptr, int off = 0,
                                           cf* ukeep = nullptr, float* ekeep = nullptr) {
    using PL = Plan<N>;
    constexpr int T = PL::T, P = PL::P;
    int* redi = reinterpret_cast<int*>(red) + 8;
#pragma unroll
    for (int s = 0; s < P; ++s) Z[s] = cf{0.f, 0.f};
    zdup = cf{0.f, 0.f};
    float psum = 0.f;
    for (int LL = 0; LL < rx.S; ++LL) {
        const int64_t w0 = int64_t(rx.L) * LL + int64_t(Ptrial) * rx.stride + rx.cp + off;   // :146
        cf v[P];
#pragma unroll
        for (int n0 = 0; n0 < P; ++n0) {
            const int64_t idx = w0 + t + T * n0;
            v[n0] = (active && idx < frame_len) ? frame_iq[idx] : cf{0.f, 0.f};
            if (rot) v[n0] = cmul(v[n0], rot[t + T * n0]);          // sig_with_fo = dat_time * cfo[fo]  (SynchEstAndFO.py:264)
        }
        wg_fft<N>(v, lds, tw, w1tab, t);                                                       // :152
        if constexpr (KEEP > 0) {
            if (t == 0) {                                   // bins 0 and N/2 both live in lane 0
                const cf y0 = v[out_slot<N>(0, 0)], yh = v[out_slot<N>(0, PL::RL / 2)];
                red[16] = y0.x;
                red[17] = y0.y;
                red[18] = yh.x;
                red[19] = yh.y;
            }
        }
        int Ks_ = rx.Ks;                      // opaque per segment: keeps the 32 per-slot table offsets out of long-lived VGPRs
        asm volatile("" : "+s"(Ks_));
        const cf* zcs = rx.zc + LL * Ks_;
        cf* ys = yscratch ? yscratch + LL * Ks_ : nullptr;
        // The Zadoff-Chu entry of every bin this lane holds, from the LANE-ORDER copy of the table (RxDev::zcp): entry
        // [slot*T + t], zero for bins outside the list.  One base address, sixteen independent coalesced reads in flight together.
        // (Gathering zc[list index of bin k] through per-bin index arithmetic put 48 loop-invariant VGPRs of indices and addresses
        // across the trial loop and made hipcc wait for every read on its own: 16 global-memory latencies per FFT.)
        const cf* zp = rx.zcp + LL * N + t;
        cf zt[P];
#pragma unroll
        for (int s = 0; s < P; ++s) zt[s] = zp[s * T];
        // every read issued before any is consumed (one opaque use of all of them: hipcc otherwise sinks each read to its use
        // and waits for it there)
        if constexpr (P == 16) {
            asm volatile("" : "+v"(zt[0]), "+v"(zt[1]), "+v"(zt[2]), "+v"(zt[3]), "+v"(zt[4]), "+v"(zt[5]), "+v"(zt[6]), "+v"(zt[7]),
                              "+v"(zt[8]), "+v"(zt[9]), "+v"(zt[10]), "+v"(zt[11]), "+v"(zt[12]), "+v"(zt[13]), "+v"(zt[14]), "+v"(zt[15]));
        } else {
            asm volatile("" : "+v"(zt[0]), "+v"(zt[1]), "+v"(zt[2]), "+v"(zt[3]), "+v"(zt[4]), "+v"(zt[5]), "+v"(zt[6]), "+v"(zt[7]));
        }
#pragma unroll
        for (int j = 0; j < PL::C; ++j) {
#pragma unroll
            for (int kl = 0; kl < PL::RL; ++kl) {
                const int k = (t + T * j) + PL::NC * kl;
                const int s = out_slot<N>(j, kl);
                int in_, ip_;
                const bool neg = bin_neg(k, Ks_, N, in_), pos = bin_pos(k, Ks_, ip_);   // :153-161
                const bool used = neg || pos;
                Z[s] = Z[s] + cmulc(v[s], zt[s]);
                psum += used ? cnorm2(v[s]) : 0.f;
            }
        }
        if (Ks_ == N) {                             // K == N lists bin N/2 twice (ofdm_chain.py:83 wiring); zcp holds its LAST
#pragma unroll                                      // (positive-half) entry, the first (negative-half) one is added here
            for (int j = 0; j < PL::C; ++j) {
#pragma unroll
                for (int kl = 0; kl < PL::RL; ++kl) {
                    const int k = (t + T * j) + PL::NC * kl;
                    const int s = out_slot<N>(j, kl);
                    if (k == N / 2) {
                        const cf c = cmulc(v[s], zcs[0]);                               // negative half: list index 0
                        zdup = zdup + c;
                        Z[s] = Z[s] + c;
                        psum += cnorm2(v[s]);
                    }
                }
            }
        }
        if (ys && active) {                         // raw sync-bin values for est_synch_freq, AFTER the reads above: a store
#pragma unroll                                      // between two table reads would order them (the pointers may alias)
            for (int j = 0; j < PL::C; ++j) {
#pragma unroll
                for (int kl = 0; kl < PL::RL; ++kl) {
                    const int k = (t + T * j) + PL::NC * kl;
                    int in_, ip_;
                    const bool neg = bin_neg(k, Ks_, N, in_);
                    const bool pos = bin_pos(k, Ks_, ip_);
                    if (neg) ys[in_] = v[out_slot<N>(j, kl)];
                    if (pos) ys[ip_] = v[out_slot<N>(j, kl)];
                }
            }
        }
        wg_barrier();
    }
    psum = symbol_sum<T>(psum, red, t);
    p_est = sqrtf(float(rx.MM) / psum);                                                 // :157
    if constexpr (KEEP > 0) *ekeep = psum;

    // del_mat[d] = sum_k e^{+j 2pi d k/N} Z[k]  == unnormalised inverse DFT of Z read at d = 0..cp
#pragma unroll
    for (int j = 0; j < PL::C; ++j) {
#pragma unroll
        for (int kl = 0; kl < PL::RL; ++kl) lds[(t + T * j) + PL::NC * kl] = Z[out_slot<N>(j, kl)];
    }
    wg_barrier();
    cf v[P];
#pragma unroll
    for (int n0 = 0; n0 < P; ++n0) v[n0] = cconj(lds[t + T * n0]);
    wg_barrier();
    wg_fft<N>(v, lds, tw, w1tab, t);
    float best = -1.f;
    int bi = -1;
#pragma unroll
    for (int j = 0; j < PL::C; ++j) {
#pragma unroll
        for (int kl = 0; kl < PL::RL; ++kl) {
            const int d = (t + T * j) + PL::NC * kl;
            const float m2 = cnorm2(v[out_slot<N>(j, kl)]);
            if (d <= rx.cp && (bi < 0 || m2 > best || (m2 == best && d < bi))) {        // :163 first max
                best = m2;
                bi = d;
            }
        }
    }
    if constexpr (KEEP > 0) {
        // lag a = t + T*q sits in register slot (j, kl) = (q % C, q / C): NC = T*C
#pragma unroll
        for (int q = 0; q < KEEP; ++q) ukeep[q] = cconj(v[out_slot<N>(q % PL::C, q / PL::C)]);
    }
    symbol_argmax<T>(best, bi, red, redi, t);
    m = p_est * sqrtf(best);                                                            // :164
    dhat = bi;
    wg_barrier();
}

// ---- finalize (:171-218): LS estimate on the sync bins, equaliser gains, channel impulse response.
// Shared by the sequential and the screened search kernels; Zs .. dhats describe the accepted trial (zeros if none).
template <int N, class TW>
__device__ __forceinline__ void sync_finalize(const RxDev& rx, const SyncArgs& a, int frame, bool active, bool found, int Phit,
                                              const cf (&Zs)[Plan<N>::P], cf zdups, float pests, float ms, int dhats, cf* lds,
                                              const TW& tw, const cf* w1tab, int t, cf* ysc) {
    using PL = Plan<N>;
    constexpr int T = PL::T, P = PL::P;
    const int Ks = rx.Ks, Kd = rx.Kd;
    if (active && t == 0) {
        int* o = a.tsr + int64_t(frame) * 4;
        if (found || !a.keep_on_miss) {
            o[0] = found ? Phit * rx.stride + rx.cp + a.off_delta : 0;                  // :173
            o[1] = found ? dhats : 0;                                                    // :174
            o[2] = found ? int(ms) : 0;                                                  // :175
        }
        o[3] = found ? 1 : 0;
        if (a.tsr_host) {
            int* oh = a.tsr_host + int64_t(frame) * 4;
            if (found || !a.keep_on_miss) {
                oh[0] = o[0];
                oh[1] = o[1];
                oh[2] = o[2];
            }
            oh[3] = found ? 1 : 0;
        }
    }
    active = active && (found || !a.keep_on_miss);     // from here on `active` only gates the stores
    // Z (register slot order) -> LDS in natural bin order, then a rolled loop over this lane's bins: the finalize
    // arithmetic runs once per frame, so it is kept small in registers rather than unrolled 16-fold.
#pragma unroll
    for (int j = 0; j < PL::C; ++j) {
#pragma unroll
        for (int kl = 0; kl < PL::RL; ++kl) {
            const int k = (t + T * j) + PL::NC * kl;
            lds[k] = Zs[out_slot<N>(j, kl)];
            if (Ks == N && k == N / 2) lds[N] = zdups;          // negative-half part of the bin listed twice (K == N)
        }
    }
    wg_barrier();
    const float sc = pests * rx.inv_ls;                                                 // p_est / (S (1+1/snr)) :180-184
    // the lag de-rotation e^{+j 2pi d k/N} of this lane's 16 bins (:177): sixteen independent table reads issued together
    // (one per loop iteration, each waited for on its own, was a third of the kernel's time in the aligned case)
    cf rots[P];
#pragma unroll
    for (int q = 0; q < P; ++q) rots[q] = rx.tw[(dhats * (t + T * q)) & (N - 1)];
    if constexpr (P == 16) {
        asm volatile("" : "+v"(rots[0]), "+v"(rots[1]), "+v"(rots[2]), "+v"(rots[3]), "+v"(rots[4]), "+v"(rots[5]), "+v"(rots[6]), "+v"(rots[7]),
                          "+v"(rots[8]), "+v"(rots[9]), "+v"(rots[10]), "+v"(rots[11]), "+v"(rots[12]), "+v"(rots[13]), "+v"(rots[14]), "+v"(rots[15]));
    } else {
        asm volatile("" : "+v"(rots[0]), "+v"(rots[1]), "+v"(rots[2]), "+v"(rots[3]), "+v"(rots[4]), "+v"(rots[5]), "+v"(rots[6]), "+v"(rots[7]));
    }
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int k = t + T * q;
        const cf Zk = lds[k];
        int in_, ip_;
        const bool neg = bin_neg(k, Ks, N, in_);
        const bool pos = bin_pos(k, Ks, ip_);
        const cf zd = (pos && neg) ? lds[N] : cf{0.f, 0.f};
        const cf rot = cconj(rots[q]);                                                  // e^{+j 2pi d k/N}  :177
        const cf Hn = cscale(cmul(rot, (pos && neg) ? zd : Zk), sc);
        const cf Hp = cscale(cmul(rot, (pos && neg) ? (Zk - zd) : Zk), sc);
        // chan_est1[synch_bins] = chan_est : a bin listed twice keeps its LAST (positive-half) entry :186-188
        cf Hk = cf{0.f, 0.f};
        if (found && neg) Hk = Hn;
        if (found && pos) Hk = Hp;
        lds[k] = Hk;                               // natural-order H for the est_chan_time inverse FFT below (same lane, same slot)
        if (active) {
            a.H[int64_t(frame) * N + k] = Hk;
            if (a.eqg || a.esf) {
                // eq_gain = conj(chan_est)/(|chan_est|^2 + 1/snr) (:213-216); est_synch_freq = eq_gain * r (:217-218)
                if (neg) {
                    const cf e = found ? cscale(cconj(Hn), 1.f / (cnorm2(Hn) + rx.inv_snr_eqsync)) : cf{0.f, 0.f};
                    if (a.eqg) a.eqg[int64_t(frame) * Ks + in_] = e;
                    if (a.esf && ysc)
                        for (int LL = 0; LL < rx.S; ++LL)
                            a.esf[int64_t(frame) * rx.MM + LL * Ks + in_] =
                                found ? cmul(e, cscale(cmul(rot, ysc[LL * Ks + in_]), pests)) : cf{0.f, 0.f};
                }
                if (pos) {
                    const cf e = found ? cscale(cconj(Hp), 1.f / (cnorm2(Hp) + rx.inv_snr_eqsync)) : cf{0.f, 0.f};
                    if (a.eqg) a.eqg[int64_t(frame) * Ks + ip_] = e;
                    if (a.esf && ysc)
                        for (int LL = 0; LL < rx.S; ++LL)
                            a.esf[int64_t(frame) * rx.MM + LL * Ks + ip_] =
                                found ? cmul(e, cscale(cmul(rot, ysc[LL * Ks + ip_]), pests)) : cf{0.f, 0.f};
                }
            }
            // data-bin gain: conj(Hd)/(|Hd|^2 + 1/SNR_lin) (:242-246) folded with the lag de-rotation (:237-240)
            const cf Hg = a.H_for_gain ? a.H_for_gain[int64_t(frame) * N + k] : Hk;
            const cf rot_g = a.gain_lag_set ? cconj(rx.tw[(a.gain_lag * k) & (N - 1)]) : rot;
            const cf gk = cmul(cscale(cconj(Hg), 1.f / (cnorm2(Hg) + rx.inv_snr_data)), rot_g);
            int id_;
            if (bin_neg(k, Kd, N, id_)) a.gain[int64_t(frame) * Kd + id_] = gk;
            if (bin_pos(k, Kd, id_)) a.gain[int64_t(frame) * Kd + id_] = gk;
        }
    }
    if (a.htime) {                                                                      // :202,212  ifft(chan_est1)
        wg_barrier();
        cf v[P];
#pragma unroll
        for (int n0 = 0; n0 < P; ++n0) v[n0] = cconj(lds[t + T * n0]);
        wg_barrier();
        wg_fft<N>(v, lds, tw, w1tab, t);
        if (active) {
#pragma unroll
            for (int j = 0; j < PL::C; ++j) {
#pragma unroll
                for (int kl = 0; kl < PL::RL; ++kl)
                    a.htime[int64_t(frame) * N + (t + T * j) + PL::NC * kl] =
                        cscale(cconj(v[out_slot<N>(j, kl)]), 1.f / float(N));
            }
        }
    }
}

// est_chan_time = ifft(est_chan_freq_P row) (:202,212) for rows of H, the same arithmetic as the finalize stage above.  The batch
// path computes it ON DEMAND (ofdm_rx_get_frame_state) instead of once per frame and launch: one FFT and 8 N bytes per frame that
// nothing on the data path reads.
template <int N>
__global__ void __launch_bounds__(Plan<N>::WG) rx_chan_time_kernel(RxDev rx, const cf* H, cf* htime, int n_rows) {
    using PL = Plan<N>;
    constexpr int T = PL::T, P = PL::P;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int slot = (T >= 64) ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T;
    const int t = tid % T;
    cf* smem = reinterpret_cast<cf*>(smem_raw);
    cf* lds = smem + slot * WgLds<N>::STRIDE;
    const cf* w1tab = wg_init_w1<N>(smem, rx.tw, tid);
    std::conditional_t<PL::R0 == 16, CompactTwiddles<N>, LaneTwiddles<N>> tw;
    load_twiddles(tw, rx.tw, t);
    const int64_t row = int64_t(blockIdx.x) * PL::SLOTS + slot;
    const bool active = row < n_rows;
    cf v[P];
#pragma unroll
    for (int n0 = 0; n0 < P; ++n0) v[n0] = active ? cconj(H[row * N + t + T * n0]) : cf{0.f, 0.f};
    wg_fft<N>(v, lds, tw, w1tab, t);
    if (active) {
#pragma unroll
        for (int j = 0; j < PL::C; ++j) {
#pragma unroll
            for (int kl = 0; kl < PL::RL; ++kl)
                htime[row * N + (t + T * j) + PL::NC * kl] = cscale(cconj(v[out_slot<N>(j, kl)]), 1.f / float(N));
        }
    }
}

template <int N, int MINW = 3>
__global__ void __launch_bounds__(Plan<N>::WG, MINW) rx_sync_kernel(RxDev rx, SyncArgs a) {
    using PL = Plan<N>;
    constexpr int T = PL::T, P = PL::P;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int slot = (T >= 64) ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T;    // a wave lies inside one slot: uniform
    const int t = tid % T;
    cf* smem = reinterpret_cast<cf*>(smem_raw);
    cf* lds = smem + slot * WgLds<N>::STRIDE;
    float* red = reinterpret_cast<float*>(lds + WgLds<N>::ELEMS);
    const cf* w1tab = wg_init_w1<N>(smem, rx.tw, tid);

    // radix-16 first pass: 4 base twiddles + products on the fly (8 VGPRs instead of 30) keep the 168-register build off scratch
    std::conditional_t<PL::R0 == 16, CompactTwiddles<N>, LaneTwiddles<N>> tw;
    load_twiddles(tw, rx.tw, t);

    const int64_t unit = int64_t(blockIdx.x) * PL::SLOTS + slot;

    if (a.mode == 1) {
        // ---- trial table for the stream block: unit = trial index, frame 0
        const int n_rot = a.n_rot > 1 ? a.n_rot : 1;
        const bool active = unit < int64_t(a.p_count) * n_rot;
        const int cand = active ? int(unit / a.p_count) : 0;            // candidate-major: unit = cand * p_count + w
        const int Ptrial = a.p_begin + int(unit % a.p_count);
        const cf* rot = a.rot ? a.rot + int64_t(cand) * N : nullptr;
        const bool valid = active && (a.host_valid || int64_t(rx.S) * rx.L + int64_t(Ptrial) * rx.stride + N + rx.cp < a.frame_len);  // :144
        cf Z[P];
        cf zdup;
        float p_est, m;
        int dhat;
        sync_trial<N>(rx, a.iq, a.frame_len, valid, Ptrial, lds, red, tw, w1tab, t, Z, zdup, p_est, m, dhat, nullptr, rot, a.off_delta);
        if (active && t == 0) {
            a.trial_m[unit] = valid ? m : -1.f;
            a.trial_d[unit] = valid ? dhat : 0;
        }
        return;
    }

    // ---- mode 0: sequential search per frame (first accepted trial wins, :166-219), then finalize
    const bool active = unit < a.n_frames;
    const int frame = active ? int(unit) : 0;
    const cf* frame_iq = a.iq + int64_t(frame) * a.frame_stride;
    cf* ysc = a.yscratch ? a.yscratch + int64_t(frame) * rx.MM : nullptr;

    bool found = false;
    cf Zs[P];                       // Z of the accepted trial
    cf zdups = cf{0.f, 0.f};
    float pests = 0.f, ms = 0.f;
    int dhats = 0, Phit = 0;
    if constexpr (PL::SLOTS == 1) {
        // one frame per workgroup: every decision is workgroup-uniform, so the accepted trial's registers are used
        // in place (no second copy of Z to keep alive across the search loop)
        for (int it = 0;; ++it) {
            const int Ptrial = a.p_begin + it;
            const bool valid = active && (a.p_count <= 0 || it < a.p_count) &&
                               (a.host_valid || int64_t(rx.S) * rx.L + int64_t(Ptrial) * rx.stride + N + rx.cp < a.frame_len);
            if (!valid) break;
            sync_trial<N>(rx, frame_iq, a.frame_len, valid, Ptrial, lds, red, tw, w1tab, t, Zs, zdups, pests, ms, dhats, ysc, a.rot, a.off_delta);
            if (a.force_dhat_p1 > 0) dhats = a.force_dhat_p1 - 1;
            if (a.force_accept || ms > rx.gate_mm) {                                    // :166
                found = true;
                Phit = Ptrial;
                break;
            }
        }
        if (!found) {
#pragma unroll
            for (int s = 0; s < P; ++s) Zs[s] = cf{0.f, 0.f};
            zdups = cf{0.f, 0.f};
            pests = 0.f;
            ms = 0.f;
            dhats = 0;
        }
    } else {
#pragma unroll
        for (int s = 0; s < P; ++s) Zs[s] = cf{0.f, 0.f};
        for (int it = 0;; ++it) {
            const int Ptrial = a.p_begin + it;
            const bool valid = active && !found && (a.p_count <= 0 || it < a.p_count) &&
                               (a.host_valid || int64_t(rx.S) * rx.L + int64_t(Ptrial) * rx.stride + N + rx.cp < a.frame_len);
            if (!__syncthreads_or(valid ? 1 : 0)) break;
            cf Z[P];
            cf zdup;
            float p_est, m;
            int dhat;
            sync_trial<N>(rx, frame_iq, a.frame_len, valid, Ptrial, lds, red, tw, w1tab, t, Z, zdup, p_est, m, dhat, ysc, a.rot, a.off_delta);
            if (a.force_dhat_p1 > 0) dhat = a.force_dhat_p1 - 1;
            if (valid && (a.force_accept || m > rx.gate_mm)) {                          // :166
                found = true;
#pragma unroll
                for (int s = 0; s < P; ++s) Zs[s] = Z[s];
                zdups = zdup;
                pests = p_est;
                ms = m;
                dhats = dhat;
                Phit = Ptrial;
            }
        }
    }

    sync_finalize<N>(rx, a, frame, active, found, Phit, Zs, zdups, pests, ms, dhats, lds, tw, w1tab, t, ysc);
}

// ------------------------------------------------------------------------------------------ screened sync search
// The reference tries the windows P = 0, 1, 2, ... one sample apart and, for each, forms the cp+1 lag correlations
// c_P[d] = sum_k e^{j 2pi d k/N} Y_P[k] conj(zc_k) from a fresh FFT (SynchAndChanEst.py:143-164).  Consecutive windows
// share N-1 samples: with D_P = x[w+N] - x[w] (w = first sample of window P),
//     Y_{P+1}[k] = (Y_P[k] + D_P) e^{j 2pi k/N}      =>      c_{P+1}[d] = c_P[d+1] + D_P * G[d+1],
//     G[m] = sum_k e^{j 2pi m k/N} conj(zc_k)   (a table, built once per handle)
// and, for num_synch_bins = N-2 (every bin but DC and Nyquist, the reference's convention), the in-band energy that
// normalises the correlation (:157) follows from three sliding sums:  E_P = N sum|x|^2 - |sum x|^2 - |sum (-1)^n x|^2.
// In "alignment" coordinates a = d + (P - P0) every correlation value is a running sum u[a] += D * G[a - j + 1], one
// complex multiply-add per step j and alignment: O(B + cp) per trial instead of two N-point FFTs.
//
// The recurrence runs in fp32, so it only SCREENS: a block of B trials starts from an exactly evaluated anchor trial
// (the same sync_trial as everywhere else); a trial whose screened peak exceeds (1 - 1e-3) * gate * MM -- or whose window
// energy is too small for the sliding sums to be trusted -- is re-evaluated exactly, in order, and only the exact value
// decides (:166).  Trials the screen rejects lie at least 1e-3 * gate * MM below the gate, two orders of magnitude more
// than the recurrence can drift over one block (<= 256 steps of ~6e-8 relative rounding), so the accepted trial and its
// lag are those of the exhaustive search.  Frames whose sync sits at trial 0 never enter the recurrence.
// Preconditions checked by the host: S == 1, stride == 1, Ks == N - 2, no rotator, B + cp <= SCAN_QM * T.
#ifndef OFDM_SCAN_MX_T256
#define OFDM_SCAN_MX_T256 5
#endif
template <int N>
struct ScanGeom {
    static constexpr int T = Plan<N>::T;
    static constexpr int QM = (T >= 256) ? 2 : (T >= 128) ? 3 : (T >= 64) ? 4 : (T >= 32) ? 6 : (Plan<N>::P < 12 ? Plan<N>::P : 12);
    static constexpr int BMAX = (QM * T < 256) ? QM * T : 256;        // longest block (trials per anchor) the recurrence walks
    // The cold-block test (below) needs no recurrence state, only the anchor's lag vector: where one frame owns the workgroup it
    // looks MX blocks ahead, so a frame whose sync lies far away pays one anchor per MX * B trials.
    static constexpr int MX = (T >= 256) ? OFDM_SCAN_MX_T256 : (T >= 128) ? 3 : (T >= 64) ? 2 : 1;     // (two at 1024-pt: three would spill)
    static constexpr int BX = MX * BMAX;                              // window-edge samples / thresholds held per anchor
    static constexpr int KX = MX * QM;                                // lag values per lane the anchor keeps (alignments a < KX * T)
    static_assert(KX <= Plan<N>::P, "the anchor's inverse FFT holds P lags per lane");
    static constexpr int RMAX = (BX + T - 1) / T;                     // window-edge samples per lane
    static constexpr int UNR = QM >= 8 ? 1 : (QM >= 6 ? 2 : 4);       // recurrence steps per loop iteration (register budget)
    // extra LDS per slot (cf units): dl[BX + 2 UNR] | xn[BX] | BMAX zeros, then G[0 .. QM*T] | thr[BX + 2 UNR floats] | flag
    static constexpr int DL_OFF = 0, XN_OFF = BX + 2 * UNR, GZ_OFF = XN_OFF + BX, G_OFF = GZ_OFF + BMAX,
                         THR_OFF = G_OFF + QM * T + 2;
    static constexpr int FLAG_OFF = THR_OFF + (BX + 2 * UNR + 1) / 2;
    static constexpr int CK = 16;                                     // steps per checkpoint window (a multiple of 2 UNR)
    static constexpr int NCHK = BMAX / CK + 2;
    static constexpr int CHK_OFF = FLAG_OFF + 1;                      // tchk[NCHK floats]
    static constexpr int BLK_OFF = CHK_OFF + (NCHK + 1) / 2;          // per wave {min thr, sum |D|} of the long and of the first block
    static constexpr int EXTRA = BLK_OFF + 10;                        // 16 floats of block totals + 4 of first hot lags
    static constexpr size_t BYTES = WgLds<N>::BYTES + size_t(Plan<N>::SLOTS) * EXTRA * sizeof(cf);
};

// u += d * g  (complex) in two packed FMAs: (d.x g.x + u.x, d.x g.y + u.y), then (d.y (-g.y) + u.x, d.y g.x + u.y)
__device__ __forceinline__ void cfma(cf& u, cf d, cf g) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "+v"(u)
        : "v"(d), "v"(g));
}

//
// SEG (stream block, one long buffer): the trial range is cut into segments of a.seg_len trials, one per workgroup slot, searched in
// parallel, in one or more launches.  Each segment publishes its first accepted trial with an atomicMin; a one-workgroup launch
// behind them (a.seg_final) re-evaluates the overall minimum exactly (the same code at the same trial: the same numbers) and
// finalizes it, so the search still returns the reference's "first accepted trial" and its estimate.  Segments behind an
// already published hit stop early.  (Round 2 let the segment that finished last finalize: one ticket per segment on one word,
// ~2000 serialised atomics per 240-symbol buffer -- most of that launch's 0.12 ms.)
template <int N, int MINW = 2, bool SEG = false>
__global__ void __launch_bounds__(Plan<N>::WG, MINW) rx_sync_scan_kernel(RxDev rx, SyncArgs a) {
    using PL = Plan<N>;
    using SG = ScanGeom<N>;
    constexpr int T = PL::T, P = PL::P, QM = SG::QM, SLOTS = PL::SLOTS;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int slot = (T >= 64) ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T;    // a wave lies inside one slot: uniform
    const int t = tid % T;
    cf* smem = reinterpret_cast<cf*>(smem_raw);
    cf* lds = smem + slot * WgLds<N>::STRIDE;
    float* red = reinterpret_cast<float*>(lds + WgLds<N>::ELEMS);
    int* redi = reinterpret_cast<int*>(red);
    if constexpr (SEG) {
        if (a.seg_final) {
            // (both words are stable here: every search launch whose result this launch looks at lies before it in its stream)
            const int w0 = __hip_atomic_load(a.seg_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int done = __hip_atomic_load(a.seg_state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.seg_final == 2) {
                if (w0 == 0x7fffffff) return;                  // early finalize, nothing found yet: the later stages go on
            } else if (done || (w0 == 0x7fffffff && a.keep_on_miss)) {
                // the early finalize has done the work, or the search ends without a hit (the old estimate stays in force, only
                // "not detected" is reported): re-arm, no table, no transform
                if (tid == 0) {
                    if (!done) {
                        a.tsr[3] = 0;
                        if (a.tsr_host) a.tsr_host[3] = 0;
                    }
                    __hip_atomic_store(a.seg_state, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.seg_state + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
        }
        if (!a.seg_final) {
            // A workgroup whose segments all lie behind an already published hit has nothing to do -- before any table is loaded.
            // (Voted: the waves of a workgroup may read different values of the word, which only ever decreases.)
            const int64_t first_p = int64_t(a.p_begin) + (int64_t(a.seg_base) + int64_t(blockIdx.x) * SLOTS) * a.seg_len;
            if (__syncthreads_and(__hip_atomic_load(a.seg_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < first_p ? 1 : 0)) return;
        }
    }
    const cf* w1tab = wg_init_w1<N>(smem, rx.tw, tid);
    cf* extra = smem + WgLds<N>::STRIDE * SLOTS + WgLds<N>::W1_ELEMS + slot * SG::EXTRA;
    cf* xo = extra + SG::DL_OFF;          // window-edge samples leaving; overwritten by dl[i] = xn[i] - xo[i]
    cf* xn = extra + SG::XN_OFF;          // window-edge samples entering
    cf* Gl = extra + SG::G_OFF;           // preceded by BMAX zero entries: alignments that are not needed yet add nothing
    float* thr = reinterpret_cast<float*>(extra + SG::THR_OFF);
    int* cflag = reinterpret_cast<int*>(extra + SG::FLAG_OFF);
    float* tchk = reinterpret_cast<float*>(extra + SG::CHK_OFF);
    float* blk = reinterpret_cast<float*>(extra + SG::BLK_OFF);
    const float gmax = a.scan_g[N + 1].x * (1.f + 1e-5f);             // max |G[m]| (host, fp64), rounded up

    std::conditional_t<PL::R0 == 16, CompactTwiddles<N>, LaneTwiddles<N>> tw;
    load_twiddles(tw, rx.tw, t);

    const int64_t unit = (SEG ? int64_t(a.seg_base) : 0) + int64_t(blockIdx.x) * SLOTS + slot;
    // (a staged segment search: this launch owns the segments seg_base .. seg_base + seg_launch - 1 of n_seg)
    bool active = unit < (SEG ? int64_t(min(a.n_seg, a.seg_base + a.seg_launch)) : a.n_frames);
    const int frame = (active && !SEG) ? int(unit) : 0;
    const cf* frame_iq = a.iq + int64_t(frame) * a.frame_stride;
    // (a run-time pointer on purpose: with a literal nullptr hipcc's schedule of the trial needs ~3000 spills at 168 VGPRs)
    cf* ysc = a.yscratch ? a.yscratch + int64_t(frame) * rx.MM : nullptr;
    const int B = a.scan_block, cp = rx.cp;
    // trials p_begin <= P < nvalid are searched: P valid iff S*L + P + N + cp < frame_len (:144), P < p_count
    int64_t nvalid64 = a.frame_len - (int64_t(rx.S) * rx.L + N + cp);
    if (nvalid64 < 0) nvalid64 = 0;
    if (a.p_count > 0 && nvalid64 > a.p_count) nvalid64 = a.p_count;
    int nvalid = active ? int(nvalid64 < (1 << 30) ? nvalid64 : (1 << 30)) : 0;
    int P0 = a.p_begin;
    if constexpr (SEG) {
        const int64_t first = int64_t(a.p_begin) + (active ? unit : 0) * a.seg_len;
        P0 = int(first < (1 << 30) ? first : (1 << 30));
        if (int64_t(nvalid) > first + a.seg_len) nvalid = int(first + a.seg_len);
    }

    if constexpr (SEG) {
        if (a.seg_final) {
            // the launch behind the search: ONE slot re-evaluates the published first hit exactly (the same code at the same
            // trial: the same numbers), finalizes it and re-arms the word for the next search
            int win = 0x7fffffff;
            if (unit == 0) win = __hip_atomic_load(a.seg_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            active = unit == 0;
            P0 = (active && win != 0x7fffffff) ? win : 0;
            nvalid = (active && win != 0x7fffffff) ? win + 1 : 0;
        }
    }

    // the table G[1 .. B + cp] -> LDS once; entries -BMAX .. 0 are ZERO: alignments behind the current trial add nothing, branch-free
    for (int i = t - SG::BMAX; i <= QM * T; i += T) Gl[i] = i > 0 ? a.scan_g[i] : cf{0.f, 0.f};

    // Every iteration evaluates ONE trial exactly (the anchor at P0) and then screens the trials after it; the first flagged
    // trial becomes the next anchor, so "verification" and "anchor" are the same code.  The accepted trial is always the most
    // recent anchor: with one frame per workgroup its registers are used in place.
    bool found = false;
    cf Zs[P];
    cf zdups = cf{0.f, 0.f};
    float pests = 0.f, ms = 0.f;
    int dhats = 0, Phit = 0;
    if constexpr (SLOTS > 1) {
#pragma unroll
        for (int s_ = 0; s_ < P; ++s_) Zs[s_] = cf{0.f, 0.f};
    }
    // Screening margin.  A trial is handed to the exact evaluation when its screened peak exceeds (1 - 2e-4) * gate * MM.  What the
    // margin has to cover is the fp32 drift of the recurrence over one block -- at most 256 steps of one rounding each, 1.5e-5 of
    // the peak if every rounding went the same way -- and the 1e-6 of the sliding energy sums: 2e-4 is ten times that.  (Round 2
    // used 1e-3: the correlation climbs by about MM/N per trial as the window slides into the sync symbol, so a band of 1e-3 *
    // gate * MM was ~1.4 trials wide and most frames paid an anchor for a trial that the exact evaluation then turned down.)
    const float gate_s = rx.gate_mm * (1.f - 2e-4f);
    const float thr_k = gate_s * gate_s / float(rx.MM);                // |u|^2 > thr_k * E  <=>  p_est |u| > gate_s

#ifdef OFDM_EXPERIMENTS
    unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#define SCAN_STAMP(i)                                                              \
    do {                                                                           \
        unsigned long long tn_;                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory"); \
        acc[i] += unsigned(tn_ - tprev);                                           \
        tprev = tn_;                                                               \
    } while (0)
#else
#define SCAN_STAMP(i) do { } while (0)
#endif
    {
    for (;;) {
        bool blk_on = !found && P0 < nvalid;
        if constexpr (SEG) {
            // A hit published by an earlier segment ends this one: nothing at or after P0 can be the first accepted trial.  Every
            // thread reads the word itself.  Where one segment spans the whole workgroup (SLOTS == 1) its waves may see different
            // values, so they vote: ANY wave that saw the earlier hit stops the segment (the word only ever decreases) and blk_on
            // is workgroup-uniform as the anchor trial below assumes.  Where a workgroup holds several segments (SLOTS > 1) a
            // segment lies inside one wave, whose lanes read the word in one instruction: uniform per segment without a vote.
            const bool stop = !a.seg_final && __hip_atomic_load(a.seg_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < P0;
            if constexpr (SLOTS == 1) {
                if (__syncthreads_or(stop ? 1 : 0)) blk_on = false;
            } else if (stop) {
                blk_on = false;
            }
        }
        if (!__syncthreads_or(blk_on ? 1 : 0)) break;
        SCAN_STAMP(0);
        // ---- (1) anchor: exact trial at P0
        cf ux[SG::KX];                                                   // lags a = t + T*q, q < KX (the recurrence uses the first QM)
        cf (&u)[QM] = *reinterpret_cast<cf(*)[QM]>(&ux[0]);
        float e0;
        if constexpr (SLOTS == 1) {
            sync_trial<N, decltype(tw), SG::KX>(rx, frame_iq, a.frame_len, blk_on, P0, lds, red, tw, w1tab, t, Zs, zdups, pests, ms, dhats,
                                                ysc, nullptr, 0, ux, &e0);
            if (ms > rx.gate_mm) {                                                      // :166 (blk_on is workgroup-uniform here)
                found = true;
                Phit = P0;
                break;
            }
        } else {
            cf Z[P];
            cf zdup;
            float p_est, m;
            int dhat;
            sync_trial<N, decltype(tw), SG::KX>(rx, frame_iq, a.frame_len, blk_on, P0, lds, red, tw, w1tab, t, Z, zdup, p_est, m, dhat, ysc,
                                                nullptr, 0, ux, &e0);
            if (blk_on && m > rx.gate_mm) {                                             // :166
                found = true;
                Phit = P0;
#pragma unroll
                for (int s_ = 0; s_ < P; ++s_) Zs[s_] = Z[s_];
                zdups = zdup;
                pests = p_est;
                ms = m;
                dhats = dhat;
            }
        }
        const cf y0 = cf{red[16], red[17]}, yh = cf{red[18], red[19]};
        SCAN_STAMP(1);                                                       // .. anchor trial
        // ---- (2) screen the trials P0+1 .. P0+nb-1 with the recurrence
        int nb = (blk_on && !found) ? min(B, nvalid - P0) : 0;            // trials of this block (incl. the anchor)
        int nbx = (blk_on && !found) ? min(SG::MX * B, nvalid - P0) : 0;   // ... of the long block the cold test looks over
        int cand = 0x7fffffff;
        bool long_ok = false;
        float ucold = 3.0e38f;
        if (__syncthreads_or(nb > 1 ? 1 : 0)) {
            if constexpr (SG::MX > 1) {
                // How far ahead is the long block worth looking?  From far away the sync symbol already shows in the anchor's lag
                // vector at the alignments that will see it whole (the cp + 1 lags before its own): the long block ends where the
                // first lag at half the anchor's own threshold would come within reach (a <= n - 1 + cp), so that the cold test
                // below has a chance over all of it.  Only the amount of work depends on this guess, never a decision.
                float ah = 3.0e38f, um = 0.f;                           // um: the largest lag magnitude^2 BELOW that level
                const float half2 = 0.25f * thr_k * e0;
#pragma unroll
                for (int q = 0; q < SG::KX; ++q) {
                    const float m2 = cnorm2(ux[q]);
                    if (!(m2 < half2))
                        ah = fminf(ah, float(t + T * q));
                    else
                        um = fmaxf(um, m2);
                }
                ah = wave_min(ah);
                um = -wave_min(-um);
                if ((t & 63) == 0) {
                    blk[16 + (t >> 6)] = ah;
                    blk[(t >> 6) * 4] = um;
                }
                wg_barrier();
                um = 0.f;
#pragma unroll
                for (int w = 0; w < (T + 63) / 64; ++w) {
                    ah = fminf(ah, blk[16 + w]);
                    um = fmaxf(um, blk[w * 4]);
                }
                const int a_first = ah < 1.0e9f ? int(ah) : 0x3fffffff;
                long_ok = a_first - cp >= nb;                            // the first hot lag lies outside the long block's reach
                nbx = max(nb, min(nbx, a_first - cp));
                ucold = sqrtf(um) * (1.f + 1e-5f);                      // no lag the long block can reach starts above this
                wg_barrier();                                            // blk[] is read by everyone before it is rewritten below
            }
            // window edges: xo[i] = x[w0 + i], xn[i] = x[w0 + N + i], w0 = P0 + cp;  i < nbx - 1
            const int64_t w0 = int64_t(P0) + cp;
            float dw = 0.f, wadd = 0.f, dlane = 0.f;
            cf da = cf{0.f, 0.f}, db = cf{0.f, 0.f};
#pragma unroll
            for (int r = 0; r < SG::RMAX; ++r) {
                const int i = t * SG::RMAX + r;                          // contiguous chunk per lane
                if (i < nbx - 1) {
                    const cf o = frame_iq[w0 + i], n_ = frame_iq[w0 + N + i];
                    xo[i] = o;
                    xn[i] = n_;
                    const float sgn = (i & 1) ? 1.f : -1.f;             // (-1)^(i+1)
                    dw += cnorm2(n_) - cnorm2(o);
                    wadd += cnorm2(n_);
                    da = da + (n_ - o);
                    db = db + cscale(o - n_, sgn);
                    dlane += sqrtf(cnorm2(n_ - o));
                }
            }
            SCAN_STAMP(2);                                                   // .. window-edge loads
            // inclusive scan of the per-lane totals over the T lanes of the slot (T <= 64: inside one wave; else via LDS)
            float sw = dw, swa = wadd, sd = dlane;
            cf sa = da, sb = db;
            constexpr int W = T < 64 ? T : 64;
            constexpr bool PREFIX_D = SLOTS == 1 && SG::MX > 1;          // the |D| prefix feeds the long block's cold test only
#pragma unroll
            for (int dlt = 1; dlt < W; dlt <<= 1) {
                const float o1 = __shfl_up(sw, dlt, W), o2 = __shfl_up(swa, dlt, W);
                const float o3 = __shfl_up(sa.x, dlt, W), o4 = __shfl_up(sa.y, dlt, W);
                const float o5 = __shfl_up(sb.x, dlt, W), o6 = __shfl_up(sb.y, dlt, W);
                float o7 = 0.f;
                if constexpr (PREFIX_D) o7 = __shfl_up(sd, dlt, W);
                if ((t & (W - 1)) >= dlt) {
                    sw += o1;
                    swa += o2;
                    sa = sa + cf{o3, o4};
                    sb = sb + cf{o5, o6};
                    sd += o7;
                }
            }
            float tot_add;
            if constexpr (T > 64) {
                constexpr int NW = T / 64;
                if ((t & 63) == 63) {
                    float* r6 = red + (t >> 6) * 6;                      // NW <= 4 waves x 6 floats = the 24-dword scratch
                    r6[0] = sw;
                    r6[1] = swa;
                    r6[2] = sa.x;
                    r6[3] = sa.y;
                    r6[4] = sb.x;
                    r6[5] = sb.y;
                    if constexpr (PREFIX_D) blk[(t >> 6) * 4 + 1] = sd;  // (the wave's |D| total: next to the scratch, not in it)
                }
                wg_barrier();
                tot_add = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const float* r6 = red + w * 6;
                    tot_add += r6[1];
                    if (w < (t >> 6)) {
                        sw += r6[0];
                        sa = sa + cf{r6[2], r6[3]};
                        sb = sb + cf{r6[4], r6[5]};
                        if constexpr (PREFIX_D) sd += blk[w * 4 + 1];
                    }
                }
            } else {
                tot_add = __shfl(swa, W - 1, W);
            }
            // exclusive prefix of this lane's chunk
            float pw = sw - dw;
            cf pa = sa - da, pb = sb - db;
            // anchor: E_0 = e0 (in band), N W_0 = e0 + |Y[0]|^2 + |Y[N/2]|^2, sum x = Y[0], sum (-1)^n x = Y[N/2]
            const float nw0 = e0 + cnorm2(y0) + cnorm2(yh);
            const float wbound = nw0 + float(N) * tot_add;               // >= N W_j for every j of the block
            // this lane's share of {min_j thr_j, sum_i |D_i|} of the first block, and the first trial of the long block that its
            // own margin does not keep cold (below)
            float my_tminb = 3.0e38f, my_dabsb = 0.f;
            float sdrun = sd - dlane;                                    // sum of |D_i| before this lane's chunk
            int my_fail = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < SG::RMAX; ++r) {
                const int i = t * SG::RMAX + r;
                if (i < nbx - 1) {
                    const cf o = xo[i], n_ = xn[i];
                    const float sgn = (i & 1) ? 1.f : -1.f;
                    pw += cnorm2(n_) - cnorm2(o);
                    pa = pa + (n_ - o);
                    pb = pb + cscale(o - n_, sgn);
                    // state AFTER step i, i.e. of trial j = i + 1
                    const float nwj = nw0 + float(N) * pw;
                    const float ej = nwj - cnorm2(y0 + pa) - cnorm2(yh + pb);
                    // too little energy left for the sliding sums to be trusted -> force an exact evaluation
                    const bool weak = !(nwj > 1e-4f * wbound) || !(ej > 1e-3f * nwj);
                    const float th_ = weak ? -1.f : thr_k * ej;
                    thr[i + 1] = th_;
                    xo[i] = n_ - o;                                      // dl[i] (this lane's own entry)
                    const float dm = sqrtf(cnorm2(n_ - o));
                    if (i < nb - 1) {
                        my_tminb = fminf(my_tminb, th_);
                        my_dabsb += dm;
                    }
                    if constexpr (PREFIX_D) {
                        // trial j = i + 1 stays cold if sqrt(thr_j) - max|G| * sum_{i' <= i} |D_i'| still clears every lag it can see
                        sdrun += dm;
                        const float margin = sqrtf(fmaxf(th_, 0.f)) - sdrun * gmax * (1.f + 1e-4f);
                        if ((!(th_ > 0.f) || !(margin > ucold)) && my_fail == 0x7fffffff) my_fail = i + 1;
                    }
                }
            }
            if constexpr (SLOTS == 1) {
                my_tminb = wave_min(my_tminb);
                my_dabsb = wave_sum(my_dabsb);
                const float ff = wave_min(float(min(my_fail, 1 << 24)));      // (exact in a float)
                if ((t & 63) == 0) {
                    float* b4 = blk + (t >> 6) * 4;
                    b4[0] = ff;
                    b4[2] = my_tminb;
                    b4[3] = my_dabsb;
                }
            }
            // Cold blocks.  c_{P0+j}[d] = c_{P0}[d + j] + sum_{i<j} D_i G[.] exactly, so at trial j no correlation value can lie
            // above |u_a(0)| + max|G| * sum_{i<j} |D_i|.  Where that stays below the trial's threshold for every alignment in
            // reach, the trial cannot be flagged -- let alone accepted: the thresholds sit 2e-4 below the gate -- and needs no
            // recurrence.  Over the long block (up to MX * B trials, ending before the first lag at half the anchor's threshold
            // comes within reach) this is decided TRIAL BY TRIAL with the running sum of |D| and the trial's own threshold against
            // the largest lag below that level: the trials before the first one that fails are skipped, however many they are (the
            // running sum eats the margin after ~500 trials at 2048-pt; testing whole blocks against their lowest threshold and
            // their total sum, as the first version did, skipped 240 where this skips 480-550: profiles/r03_sync_leads.txt).  If
            // that does not reach past the first block, the first block alone is tested by its totals against every lag it can
            // see.  Weak-energy trials (thr < 0) and blocks near the sync fail both and take the screened recurrence below.
            bool run_block = true;
            if constexpr (SLOTS == 1) {
                wg_barrier();                                            // blk[] and the thresholds are written
                auto cold = [&](int n_tr) {                              // the first block, by its totals
                    float tmin_b = 3.0e38f, dtot = 0.f;
#pragma unroll
                    for (int w = 0; w < (T + 63) / 64; ++w) {
                        tmin_b = fminf(tmin_b, blk[w * 4 + 2]);
                        dtot += blk[w * 4 + 3];
                    }
                    const float room = sqrtf(fmaxf(tmin_b, 0.f)) - dtot * gmax * (1.f + 1e-5f);
                    bool hot = !(tmin_b > 0.f) || !(room > 0.f);
                    const float lim2 = room * room * (1.f - 1e-5f);
#pragma unroll
                    for (int q = 0; q < SG::KX; ++q) hot |= (t + T * q <= n_tr - 1 + cp) && !(cnorm2(ux[q]) < lim2);
                    return __syncthreads_or(hot ? 1 : 0) == 0;
                };
                int n_cold = 0;                                          // trials 1 .. n_cold - 1 of the long block are proven cold
                if constexpr (SG::MX > 1) {
                    if (long_ok && nbx > nb) {
                        float ff = 3.0e38f;
#pragma unroll
                        for (int w = 0; w < (T + 63) / 64; ++w) ff = fminf(ff, blk[w * 4]);
                        n_cold = min(int(ff), nbx);
                    }
                }
                if (n_cold > nb) {
                    nb = n_cold;                                         // skip them all; the next anchor is the first trial not proven
                    run_block = false;
                } else if (cold(nb)) {
                    run_block = false;
                }
            }
            if (run_block) {
                // padding read by the unrolled loop past the block's last step: adds nothing, never flags
                for (int idx = t; idx < SG::BMAX + 2 * SG::UNR; idx += T) {
                    if (idx >= nb - 1) xo[idx] = cf{0.f, 0.f};
                    if (idx >= nb) thr[idx] = 3.0e38f;
                }
                if (t == 0) *cflag = 0x7fffffff;
                wg_barrier();
            }
            if (run_block) {
            // Checkpoints.  Window c = steps c CK + 1 .. (c+1) CK.  One step changes a correlation value by D_j G[.], at most
            // |D_j| max|G|, so no value can pass its threshold inside the window unless it starts the window within
            // sum |D| max|G| of the window's lowest threshold: |u|^2 > tchk[c] = (sqrt(min thr) - sum|D| max|G|)^2 is tested
            // once per window and only windows that pass it run the per-step tests (tchk < 0: always).
            if constexpr (T >= SG::CK) {
                // one (threshold, |D|) entry per lane, reduced over the CK lanes of a window
                for (int e0 = 0; e0 < SG::NCHK * SG::CK; e0 += T) {
                    const int e = e0 + t;
                    const bool in = e + 1 < SG::BMAX + 2 * SG::UNR;
                    float lo = in ? thr[e + 1] : 3.0e38f;
                    float dsum = in ? sqrtf(cnorm2(xo[e])) : 0.f;
#pragma unroll
                    for (int m = SG::CK >> 1; m >= 1; m >>= 1) {
                        lo = fminf(lo, __shfl_xor(lo, m, SG::CK));
                        dsum += __shfl_xor(dsum, m, SG::CK);
                    }
                    const float s_ = sqrtf(fmaxf(lo, 0.f)) - dsum * gmax;
                    if ((e & (SG::CK - 1)) == 0 && e < SG::NCHK * SG::CK) tchk[e / SG::CK] = (lo > 0.f && s_ > 0.f) ? s_ * s_ : -1.f;
                }
            } else {
                for (int c = t; c < SG::NCHK; c += T) {
                    float lo = 3.0e38f, dsum = 0.f;
#pragma unroll
                    for (int i = 0; i < SG::CK; ++i) {
                        const int j = c * SG::CK + 1 + i;
                        if (j < SG::BMAX + 2 * SG::UNR) {
                            lo = fminf(lo, thr[j]);
                            dsum += sqrtf(cnorm2(xo[j - 1]));
                        }
                    }
                    const float s_ = sqrtf(fmaxf(lo, 0.f)) - dsum * gmax;
                    tchk[c] = (lo > 0.f && s_ > 0.f) ? s_ * s_ : -1.f;
                }
            }
            wg_barrier();
            SCAN_STAMP(3);                                                   // .. prefix scan + thresholds
            // ---- recurrence over the steps j = 1 .. nb-1, UNR steps per iteration with all their LDS reads issued up front.
            // No barrier inside: lanes leave the loop on their own (first flagged trial, or the end of their frame's block).
            // The first flagged trial is published in LDS (cflag) and every lane of the frame stops there: without that, only
            // the lane that owns the flagged alignment would leave and its wave would still run to the end of the block.
            {
                const cf* dl = xo;
                cf dA[SG::UNR], gA[SG::UNR][QM], dB[SG::UNR], gB[SG::UNR][QM];
                float tA[SG::UNR], tB[SG::UNR];
                const cf* gbase = Gl + t + 1;                            // G index of alignment t + T*q at trial j: (t + 1 - j) + T*q
                auto fetch = [&](int j, cf (&dd)[SG::UNR], float (&th)[SG::UNR], cf (&gg)[SG::UNR][QM]) {
#pragma unroll
                    for (int s_ = 0; s_ < SG::UNR; ++s_) {
                        dd[s_] = dl[j + s_ - 1];
                        th[s_] = thr[j + s_];
#pragma unroll
                        for (int q = 0; q < QM; ++q) gg[s_][q] = gbase[T * q - (j + s_)];   // index >= -BMAX: zeros below 1
                    }
                };
                auto step = [&](int j, const cf (&dd)[SG::UNR], const float (&th)[SG::UNR], const cf (&gg)[SG::UNR][QM]) {
                    int first = SG::UNR;
#pragma unroll
                    for (int s_ = 0; s_ < SG::UNR; ++s_) {
                        bool h = false;
#pragma unroll
                        for (int q = 0; q < QM; ++q) {
                            cfma(u[q], dd[s_], gg[s_][q]);
                            h |= (unsigned(t + T * q - (j + s_)) <= unsigned(cp)) && (cnorm2(u[q]) > th[s_]);
                        }
                        if (h && first == SG::UNR) first = s_;
                    }
                    if (first < SG::UNR) {
                        cand = j + first;
                        atomicMin(cflag, cand);
                    }
                };
                auto advance = [&](const cf (&dd)[SG::UNR], const cf (&gg)[SG::UNR][QM]) {   // the recurrence alone
#pragma unroll
                    for (int s_ = 0; s_ < SG::UNR; ++s_) {
#pragma unroll
                        for (int q = 0; q < QM; ++q) cfma(u[q], dd[s_], gg[s_][q]);
                    }
                };
                int j = 1;
                int lim = nb;
                bool tests = true;                                       // per-step threshold tests in this window (wave-uniform)
                if (j < lim) fetch(j, dA, tA, gA);
                while (j < lim) {
                    if (((j - 1) & (SG::CK - 1)) == 0) {
                        const float tc = tchk[(j - 1) / SG::CK];
                        bool h = false;
#pragma unroll
                        for (int q = 0; q < QM; ++q)
                            h |= (unsigned(t + T * q - j) <= unsigned(cp + SG::CK - 1)) && (cnorm2(u[q]) > tc);
                        tests = __builtin_amdgcn_ballot_w64(h) != 0;
                    }
                    fetch(j + SG::UNR, dB, tB, gB);                      // reads past the block's end hit the padding
                    const int seen = *cflag;
                    if (tests) step(j, dA, tA, gA); else advance(dA, gA);
                    j += SG::UNR;
                    lim = min(lim, seen);
                    if (cand != 0x7fffffff || j >= lim) break;
                    fetch(j + SG::UNR, dA, tA, gA);
                    const int seen2 = *cflag;
                    if (tests) step(j, dB, tB, gB); else advance(dB, gB);
                    j += SG::UNR;
                    lim = min(lim, seen2);
                    if (cand != 0x7fffffff) break;
                }
            }
            SCAN_STAMP(4);                                                   // .. recurrence loop
            // first flagged trial of the slot
#pragma unroll
            for (int mk = W >> 1; mk >= 1; mk >>= 1) cand = min(cand, __shfl_xor(cand, mk, W));
            if constexpr (T > 64) {
                wg_barrier();                                            // red[] above has been read by everyone
                if ((t & 63) == 0) redi[t >> 6] = cand;
                wg_barrier();
                int c2 = redi[0];
#pragma unroll
                for (int w = 1; w < T / 64; ++w) c2 = min(c2, redi[w]);
                cand = c2;
                wg_barrier();
            }
            }   // run_block
        }
        // the first flagged trial is the next anchor (evaluated exactly there); an unflagged block is skipped whole
        if (nb > 0) P0 += (cand < nb) ? cand : nb;
        SCAN_STAMP(5);                                                       // .. candidate reduction
    }
    }
    if constexpr (SEG) {
        if (!a.seg_final) {
            // publish and leave: the finalize launch behind the search picks the overall minimum up
            if (t == 0 && active && found) atomicMin(a.seg_state, Phit);
            return;
        }
        if (tid == 0) {
            if (a.seg_final == 2) {
                __hip_atomic_store(a.seg_state + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // finalized; word 0 stays:
            } else {                                                                                   // it stops the later stages
                __hip_atomic_store(a.seg_state, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
                __hip_atomic_store(a.seg_state + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    SCAN_STAMP(6);
    sync_finalize<N>(rx, a, frame, active, found, Phit, Zs, zdups, pests, ms, dhats, lds, tw, w1tab, t, ysc);
#ifdef OFDM_EXPERIMENTS
    SCAN_STAMP(7);                                                           // .. finalize
    if (a.stamps && tid == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.stamps[int64_t(blockIdx.x) * 8 + i] = acc[i];
    }
#endif
#undef SCAN_STAMP
}

// ------------------------------------------------------------------------------------------ bit-error count
// count += popcount(a ^ b) over n bytes: the BER numerator of two packed bit-streams without moving them anywhere (SURVEY 8e:
// "gather counts instead"; the reference's idiom is bitwise_xor(a, b).sum(), TEST/GNU_RADIO_OFFLINE/pls_aio.py:131).
__global__ void __launch_bounds__(256) bit_errors_kernel(const uint8_t* a, const uint8_t* b, int64_t n, unsigned long long* count) {
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, stride = int64_t(gridDim.x) * blockDim.x;
    const bool wide = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    const int64_t n16 = wide ? n / 16 : 0;
    unsigned c = 0;                                                   // <= 128 per step: a lane would need 2^25 steps to overflow
    const uint4* a4 = reinterpret_cast<const uint4*>(a);
    const uint4* b4 = reinterpret_cast<const uint4*>(b);
    for (int64_t i = gid; i < n16; i += stride) {
        const uint4 x = a4[i], y = b4[i];
        c += __popc(x.x ^ y.x) + __popc(x.y ^ y.y) + __popc(x.z ^ y.z) + __popc(x.w ^ y.w);
    }
    for (int64_t i = n16 * 16 + gid; i < n; i += stride) c += __popc(unsigned(a[i] ^ b[i]));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, static_cast<unsigned long long>(c));
}

hipError_t launch_bit_errors(const uint8_t* a, const uint8_t* b, int64_t n, unsigned long long* count, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int64_t blocks = std::min<int64_t>((n / 16 + 255) / 256 + 1, 65536);
    hipLaunchKernelGGL(bit_errors_kernel, dim3(unsigned(blocks)), dim3(256), 0, s, a, b, n, count);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ standalone de-mapper
// Hard bits, one per byte.  A thread takes FOUR consecutive symbols (two 16 B loads) and writes their 4*MOD bytes as whole
// words (byte-by-byte stores, one symbol per thread, ran at 0.36-0.64 of the HBM rate); the last n % 4 symbols and buffers that
// are not 16-byte aligned take the plain path.
template <int MOD>
__device__ __forceinline__ void demap_hard_words(const cf (&z)[4], uint32_t (&w)[MOD]) {
#pragma unroll
    for (int k = 0; k < MOD; ++k) w[k] = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned hb = hard_bits<MOD>(z[e]);
#pragma unroll
        for (int j = 0; j < MOD; ++j) {
            const int k = e * MOD + j;                                 // byte index in the group's 4*MOD bytes
            w[k >> 2] |= ((hb >> (MOD - 1 - j)) & 1u) << (8 * (k & 3));
        }
    }
}
__device__ __forceinline__ void qpsk_nearest(cf z, bool& re_pos, bool& im_pos, cf& e) {
    // quadrant tests in the reference's order ++, -+, --, +- (BitRecovery.py:106-125)
    re_pos = (z.x > 0.f) || (z.x == 0.f && z.y >= 0.f);
    im_pos = (z.y >= 0.f);
    constexpr float c = 0.70710678118654752f;
    e = cf{z.x - (re_pos ? c : -c), z.y - (im_pos ? c : -c)};                            // :93-98
}

__device__ __forceinline__ void store_stream16(float* p, float4 v);
// 16-byte load of data this kernel reads once
__device__ __forceinline__ float4 load_stream16(const float* p) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
    return float4{v.x, v.y, v.z, v.w};
}

// pass 2: llrp0 / llrp1 (BitRecovery.py:102-125)
__global__ void __launch_bounds__(256) demap_soft_kernel(DemapArgs a) {
    __shared__ double sh_tot;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < DEMAP_PARTIALS; ++i) s += a.partial[i];
        sh_tot = s;
    }
    __syncthreads();
    const double sigma = 0.7071067811865476 * (sh_tot / double(a.n));                    // :102
    const float hf = float(-0.5 / (sigma * sigma));                                      // -0.5*dfact :103
    constexpr float K = 1.414213562373095f;                                              // :57
    auto metrics = [&](cf z, float (&m0)[2], float (&m1)[2]) {
        bool rp, ip;
        cf e;
        qpsk_nearest(z, rp, ip, e);
        const float nr = hf * fabsf(e.x), fr = hf * (K - fabsf(e.x));
        const float ni = hf * fabsf(e.y), fi = hf * (K - fabsf(e.y));
        m0[0] = rp ? nr : fr;
        m0[1] = ip ? ni : fi;
        m1[0] = rp ? fr : nr;
        m1[1] = ip ? fi : ni;
    };
    // two symbols per thread: one 16 B load, one 16 B store per metric array (buffers 16-byte aligned; else one by one)
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, stride = int64_t(gridDim.x) * blockDim.x;
    const bool wide = ((reinterpret_cast<uintptr_t>(a.sym) | reinterpret_cast<uintptr_t>(a.soft0) | reinterpret_cast<uintptr_t>(a.soft1)) & 15) == 0;
    const int64_t n2 = wide ? a.n >> 1 : 0;
    for (int64_t g = gid; g < n2; g += stride) {
        const float4 v = reinterpret_cast<const float4*>(a.sym)[g];
        float a0[2], a1[2], b0[2], b1[2];
        metrics(cf{v.x, v.y}, a0, a1);
        metrics(cf{v.z, v.w}, b0, b1);
        if (a.soft0) store_stream16(a.soft0 + 4 * g, float4{a0[0], a0[1], b0[0], b0[1]});
        if (a.soft1) store_stream16(a.soft1 + 4 * g, float4{a1[0], a1[1], b1[0], b1[1]});
    }
    for (int64_t i = n2 * 2 + gid; i < a.n; i += stride) {
        float m0[2], m1[2];
        metrics(a.sym[i], m0, m1);
        if (a.soft0) {
            a.soft0[2 * i] = m0[0];
            a.soft0[2 * i + 1] = m0[1];
        }
        if (a.soft1) {
            a.soft1[2 * i] = m1[0];
            a.soft1[2 * i + 1] = m1[1];
        }
    }
}

// ---- 16/64-QAM extension of the soft metric (no reference code: BitRecovery.py knows QPSK only).
// Same structure as BitRecovery.work: sigma = 0.7071 * mean distance to the nearest point over the buffer, per-bit
// metrics -0.5/sigma^2 * (linear distance), taken per axis to the nearest PAM level that carries bit value 0 / 1.
// Axis levels l_q = (2q - (M-1)) * u, q = 0..M-1, M = 4 (u = 1/sqrt(10)) or 8 (u = 1/sqrt(42)); TS 36.211 7.1 labels:
// bit 0 = (l < 0); 16-QAM bit 1 = (|m| == 3); 64-QAM bit 1 = (|m| > 4), bit 2 = (|m| == 1 or 7), m = 2q-(M-1).
template <int BPS>
struct Pam {
    static constexpr int M = BPS == 4 ? 4 : 8;
    static constexpr int NB = BPS / 2;
    static __device__ __forceinline__ float unit() { return BPS == 4 ? 0.31622776601683794f : 0.15430334996209191f; }
    static __device__ __forceinline__ bool label(int q, int j) {
        const int m = 2 * q - (M - 1), am = m < 0 ? -m : m;
        if (j == 0) return m < 0;
        if (BPS == 4) return am == 3;
        return j == 1 ? am > 4 : (am == 1 || am == 7);
    }
    // distances from coordinate x to the nearest level with axis bit j = 0 / 1, and to the nearest level overall
    static __device__ __forceinline__ void dist(float x, float (&d0)[NB], float (&d1)[NB], float& e) {
        const float u = unit();
        e = 3.0e38f;
#pragma unroll
        for (int j = 0; j < NB; ++j) d0[j] = d1[j] = 3.0e38f;
#pragma unroll
        for (int q = 0; q < M; ++q) {
            const float d = fabsf(x - float(2 * q - (M - 1)) * u);
            e = fminf(e, d);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (label(q, j))
                    d1[j] = fminf(d1[j], d);
                else
                    d0[j] = fminf(d0[j], d);
            }
        }
    }
};

// distance of one symbol to its nearest constellation point (BitRecovery.py:87-88; the QAM extension's own definition)
template <int MOD>
__device__ __forceinline__ float demap_dmin(cf z) {
    if constexpr (MOD == 2) {
        bool rp, ip;
        cf e;
        qpsk_nearest(z, rp, ip, e);
        return sqrtf(cnorm2(e));
    } else {
        float d0[Pam<MOD>::NB], d1[Pam<MOD>::NB], ex, ey;
        Pam<MOD>::dist(z.x, d0, d1, ex);
        Pam<MOD>::dist(z.y, d0, d1, ey);
        return sqrtf(ex * ex + ey * ey);
    }
}

// Pass 1 of the de-mapper, ONE read of the symbols: hard bits (one per byte; a thread takes FOUR consecutive symbols -- two 16 B
// loads -- and writes their 4*MOD bytes as whole words) and / or the partial sums of the nearest-point distances that sigma
// needs (BitRecovery.py:88,102: double partial sums, added atomically into DEMAP_PARTIALS slots zeroed by the launcher).
// Round 2 ran these as two kernels: hard + soft output read every symbol three times.
template <int MOD, bool HARD, bool DMIN>
__global__ void __launch_bounds__(256) demap_pass1_kernel(DemapArgs a) {
    const int64_t n = a.n, n4 = n >> 2;
    const bool wide = ((reinterpret_cast<uintptr_t>(a.sym) | (HARD ? reinterpret_cast<uintptr_t>(a.hard) : 0)) & 15) == 0;
    const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, stride = int64_t(gridDim.x) * blockDim.x;
    double acc = 0.0;
    if (wide) {
        for (int64_t g = gid; g < n4; g += stride) {
            const float4 v0 = load_stream16(reinterpret_cast<const float*>(a.sym) + 8 * g),
                         v1 = load_stream16(reinterpret_cast<const float*>(a.sym) + 8 * g + 4);
            const cf z[4] = {cf{v0.x, v0.y}, cf{v0.z, v0.w}, cf{v1.x, v1.y}, cf{v1.z, v1.w}};
            if constexpr (DMIN) {
                // four float distances summed in float (exact enough: 4 terms), the running sum in double
                acc += double((demap_dmin<MOD>(z[0]) + demap_dmin<MOD>(z[1])) + (demap_dmin<MOD>(z[2]) + demap_dmin<MOD>(z[3])));
            }
            if constexpr (HARD) {
                uint32_t w[MOD];
                demap_hard_words<MOD>(z, w);
                uint32_t* o = reinterpret_cast<uint32_t*>(a.hard + g * 4 * MOD);
                if constexpr (MOD == 4) {
                    *reinterpret_cast<uint4*>(o) = uint4{w[0], w[1], w[2], w[3]};
                } else if constexpr (MOD == 1) {
                    o[0] = w[0];
                } else {                                                   // 8 or 24 bytes, 8-byte aligned
#pragma unroll
                    for (int k = 0; k < MOD; k += 2) *reinterpret_cast<uint2*>(o + k) = uint2{w[k], w[k + 1]};
                }
            }
        }
    }
    for (int64_t i = (wide ? n4 * 4 : 0) + gid; i < n; i += stride) {
        const cf z = a.sym[i];
        if constexpr (DMIN) acc += double(demap_dmin<MOD>(z));
        if constexpr (HARD) {
            const unsigned hb = hard_bits<MOD>(z);
#pragma unroll
            for (int b = 0; b < MOD; ++b) a.hard[i * MOD + b] = uint8_t((hb >> (MOD - 1 - b)) & 1u);
        }
    }
    if constexpr (DMIN) {
        __shared__ double sh[256];
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
            __syncthreads();
        }
        // more workgroups than partial slots (256 of them cannot keep the memory system busy)
        if (threadIdx.x == 0) atomicAdd(a.partial + (blockIdx.x % DEMAP_PARTIALS), sh[0]);
    }
}

// 16-byte store of data that is written once and not read again by this kernel
__device__ __forceinline__ void store_stream16(float* p, float4 v) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(p));
}

template <int BPS>
__global__ void __launch_bounds__(256) demap_soft_qam_kernel(DemapArgs a) {
    __shared__ double sh_tot;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < DEMAP_PARTIALS; ++i) s += a.partial[i];
        sh_tot = s;
    }
    __syncthreads();
    const double sigma = 0.7071067811865476 * (sh_tot / double(a.n));
    const float hf = float(-0.5 / (sigma * sigma));
    constexpr int NB = Pam<BPS>::NB;
    const bool wide = ((reinterpret_cast<uintptr_t>(a.soft0) | reinterpret_cast<uintptr_t>(a.soft1)) & 15) == 0;
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n; i += int64_t(gridDim.x) * blockDim.x) {
        const cf z = a.sym[i];
        float r0[NB], r1[NB], i0[NB], i1[NB], e;
        Pam<BPS>::dist(z.x, r0, r1, e);
        Pam<BPS>::dist(z.y, i0, i1, e);
        float o0[BPS], o1[BPS];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            o0[2 * j] = hf * r0[j];
            o0[2 * j + 1] = hf * i0[j];
            o1[2 * j] = hf * r1[j];
            o1[2 * j + 1] = hf * i1[j];
        }
        // BPS floats per symbol = 16 B (16-QAM: one 16 B store where the buffer allows it) / 24 B (64-QAM: three 8 B stores)
        if (BPS == 4 && wide) {
            if (a.soft0) store_stream16(a.soft0 + i * BPS, float4{o0[0], o0[1], o0[2], o0[3]});
            if (a.soft1) store_stream16(a.soft1 + i * BPS, float4{o1[0], o1[1], o1[2], o1[3]});
        } else {
#pragma unroll
            for (int b = 0; b < BPS; b += 2) {
                if (a.soft0) *reinterpret_cast<float2*>(a.soft0 + i * BPS + b) = make_float2(o0[b], o0[b + 1]);
                if (a.soft1) *reinterpret_cast<float2*>(a.soft1 + i * BPS + b) = make_float2(o1[b], o1[b + 1]);
            }
        }
    }
}

// 64-QAM soft metrics with DENSE stores.  A symbol owns 6 floats per metric array, so a lane that keeps "its" symbols writes 24 B
// pieces at a 24 B stride: every store instruction of a wave touches 12+ lines partially (0.39 of the HBM rate with both arrays,
// round 2).  Here a wave takes a tile of 128 symbols (one 16 B load per lane = 2 symbols), parks each array's 768 floats in LDS
// in symbol order and writes them back as 192 pieces of 16 B, piece k*64 + lane per store instruction: 1 KB contiguous, written
// once, non-temporal.  The staging area is private to the wave (LDS is in order per wave: a counter wait, no barrier).
__global__ void __launch_bounds__(256) demap_soft_qam64_kernel(DemapArgs a) {
    constexpr int BPS = 6, NB = 3, TILE = 128;
    __shared__ __attribute__((aligned(16))) float stage[4][2][TILE * BPS];               // 4 waves x 2 arrays x 3 KB
    __shared__ double sh_tot;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < DEMAP_PARTIALS; ++i) s += a.partial[i];
        sh_tot = s;
    }
    __syncthreads();
    const double sigma = 0.7071067811865476 * (sh_tot / double(a.n));
    const float hf = float(-0.5 / (sigma * sigma));
    auto metrics = [&](cf z, float (&o0)[BPS], float (&o1)[BPS]) {
        float r0[NB], r1[NB], i0[NB], i1[NB], e;
        Pam<BPS>::dist(z.x, r0, r1, e);
        Pam<BPS>::dist(z.y, i0, i1, e);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            o0[2 * j] = hf * r0[j];
            o0[2 * j + 1] = hf * i0[j];
            o1[2 * j] = hf * r1[j];
            o1[2 * j + 1] = hf * i1[j];
        }
    };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool wide = ((reinterpret_cast<uintptr_t>(a.sym) | reinterpret_cast<uintptr_t>(a.soft0) | reinterpret_cast<uintptr_t>(a.soft1)) & 15) == 0;
    const int64_t n_tiles = wide ? a.n / TILE : 0;
    float* st0 = stage[wave][0];
    float* st1 = stage[wave][1];
    for (int64_t tile = int64_t(blockIdx.x) * 4 + wave; tile < n_tiles; tile += int64_t(gridDim.x) * 4) {
        const float4 v = load_stream16(reinterpret_cast<const float*>(a.sym) + 4 * (tile * (TILE / 2) + lane));
        float p0[BPS], p1[BPS], q0[BPS], q1[BPS];
        metrics(cf{v.x, v.y}, p0, p1);                                                    // symbol 2*lane of the tile
        metrics(cf{v.z, v.w}, q0, q1);                                                    // symbol 2*lane + 1
        float4* w0 = reinterpret_cast<float4*>(st0 + lane * 2 * BPS);
        float4* w1 = reinterpret_cast<float4*>(st1 + lane * 2 * BPS);
        w0[0] = float4{p0[0], p0[1], p0[2], p0[3]};
        w0[1] = float4{p0[4], p0[5], q0[0], q0[1]};
        w0[2] = float4{q0[2], q0[3], q0[4], q0[5]};
        w1[0] = float4{p1[0], p1[1], p1[2], p1[3]};
        w1[1] = float4{p1[4], p1[5], q1[0], q1[1]};
        w1[2] = float4{q1[2], q1[3], q1[4], q1[5]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              // wave-local exchange: in order per wave
        const int64_t obase = tile * (TILE * BPS);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int piece = k * 64 + lane;
            if (a.soft0) store_stream16(a.soft0 + obase + 4 * piece, *reinterpret_cast<const float4*>(st0 + 4 * piece));
            if (a.soft1) store_stream16(a.soft1 + obase + 4 * piece, *reinterpret_cast<const float4*>(st1 + 4 * piece));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              // reads done before the next tile overwrites
    }
    for (int64_t i = n_tiles * TILE + int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n; i += int64_t(gridDim.x) * blockDim.x) {
        float o0[BPS], o1[BPS];
        metrics(a.sym[i], o0, o1);
#pragma unroll
        for (int b = 0; b < BPS; ++b) {
            if (a.soft0) a.soft0[i * BPS + b] = o0[b];
            if (a.soft1) a.soft1[i * BPS + b] = o1[b];
        }
    }
}

// ------------------------------------------------------------------------------------------ row renormalisation
// SynchronizeAndEstimate.py:431-434: after row r = f*D + n has been equalised it is divided by sqrt(mean |row f|^2) -- row f,
// not row r -- in loop order, so row f already carries its own final scaling (f <= r; f == r only for row 0).
__global__ void __launch_bounds__(256) row_renorm_kernel(cf* eq, int Kd, int D, int n_frames, const int* tsr) {
    __shared__ float sh[256];
    for (int f = 0; f < n_frames; ++f) {
        if (tsr[f * 4 + 3] == 0) continue;                       // guard failed: row untouched
        for (int n = 0; n < D; ++n) {
            const int r = f * D + n;
            float acc = 0.f;
            for (int i = threadIdx.x; i < Kd; i += blockDim.x) acc += cnorm2(eq[int64_t(f) * Kd + i]);
            sh[threadIdx.x] = acc;
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1) {
                if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
                __syncthreads();
            }
            const float inv = 1.f / sqrtf(sh[0] / float(Kd));
            __syncthreads();
            for (int i = threadIdx.x; i < Kd; i += blockDim.x) eq[int64_t(r) * Kd + i] = cscale(eq[int64_t(r) * Kd + i], inv);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------ output packing of the stream block
// SynchAndChanEst.py:249-255: rows 3, 3 + (S+D), ... of est_data_freq are deleted (literal 3), the rest flattened row-major.
// Done on the device so that the block's output leaves in ONE contiguous device-to-host copy straight into the caller's buffer.
__global__ void __launch_bounds__(256) pack_rows_kernel(const cf* edf, int rows, int Kd, int SD, cf* out) {
    const int r = blockIdx.y;
    if (r >= 3 && (r - 3) % SD == 0) return;                                            // a deleted row
    const int w = r - (r >= 3 ? (r - 3) / SD + 1 : 0);                                  // rows kept before it
    const float4* src = reinterpret_cast<const float4*>(edf + int64_t(r) * Kd);         // Kd is even: whole 16 B pairs
    float4* dst = reinterpret_cast<float4*>(out + int64_t(w) * Kd);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (Kd >> 1); i += gridDim.x * blockDim.x) dst[i] = src[i];
}
hipError_t launch_pack_rows(const cf* edf, int rows, int Kd, int SD, cf* out, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(unsigned(((Kd >> 1) + 255) / 256), unsigned(rows)), dim3(256), 0, s, edf, rows, Kd, SD, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ DSSS despreading
__global__ void despread_kernel(const cf* in, int in_row_stride, const cf* code, int dsss, int n_spread, int rows, cf* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (i >= n_spread || row >= rows) return;
    const cf* x = in + int64_t(row) * in_row_stride + int64_t(i) * dsss;
    cf acc = cf{0.f, 0.f};
    for (int sf = 0; sf < dsss; ++sf) acc = acc + cmulc(x[sf], code[sf]);              // x * conj(SC[sf])  (:395)
    out[int64_t(row) * n_spread + i] = cscale(acc, 1.f / float(dsss));                  // np.average        (:396)
}

// ------------------------------------------------------------------------------------------ launchers
template <int N>
static int scan_block_n(const RxDev& rx) {
    using SG = ScanGeom<N>;
    if (rx.S != 1 || rx.stride != 1 || rx.Ks != N - 2) return 0;
    int B = SG::QM * SG::T - rx.cp;
    if (B > SG::BMAX) B = SG::BMAX;
    return B >= 16 ? B : 0;
}

template <int N>
static hipError_t launch_sync_n(const RxDev& rx, const SyncArgs& a, hipStream_t s) {
    const int64_t units = (a.mode == 1) ? int64_t(a.p_count) * (a.n_rot > 1 ? a.n_rot : 1) : a.n_frames;
    const unsigned grid = unsigned((units + Plan<N>::SLOTS - 1) / Plan<N>::SLOTS);
    if (grid == 0) return hipSuccess;
    if (a.mode == 0 && a.scan_block > 0) {
        if (a.scan_block != scan_block_n<N>(rx) || !a.scan_g || a.rot || a.force_accept || a.host_valid || a.off_delta || a.force_dhat_p1 ||
            a.p_begin < 0)
            return hipErrorInvalidValue;
        // register budget: 168 VGPRs (3 waves per SIMD) costs ~23 spills for one frame per workgroup (N >= 1024); the packed small
        // sizes keep a second copy of Z and get 256
        if (a.n_seg > 0) {
            if (a.n_frames != 1 || a.seg_len <= 0 || !a.seg_state) return hipErrorInvalidValue;
            // Staged: every workgroup resident when the launch starts runs its anchor before the first hit can be published, so
            // one launch over a 240-symbol buffer (~2000 segments) cost ~0.12 ms for a sync that sits in segment ~25.  The first
            // SYNC_STAGE_SEGS segments go first; the rest are launched behind them and, when the hit is already published,
            // only take their tickets.  Same decisions: the tickets and the published minimum span both launches.
            if (a.seg_base < 0 || a.seg_launch < 0 || a.seg_base % Plan<N>::SLOTS) return hipErrorInvalidValue;
            if constexpr (ScanGeom<N>::BYTES > 65536) {
                static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(&rx_sync_scan_kernel<N, OFDM_SCAN_MINW, true>),
                                                                   hipFuncAttributeMaxDynamicSharedMemorySize, int(ScanGeom<N>::BYTES));
                if (once != hipSuccess) return once;
            }
            SyncArgs st = a;
            int base = a.seg_base;
            const int end = a.seg_launch > 0 ? std::min(a.n_seg, a.seg_base + a.seg_launch) : a.n_seg;
            while (base < end) {
                int cnt = end - base;
                if (a.seg_launch == 0 && base == 0 && cnt > 2 * SYNC_STAGE_SEGS) cnt = SYNC_STAGE_SEGS;
                st.seg_base = base;
                st.seg_launch = cnt;
                st.seg_final = 0;
                const unsigned gseg = unsigned((int64_t(cnt) + Plan<N>::SLOTS - 1) / Plan<N>::SLOTS);
                hipLaunchKernelGGL((rx_sync_scan_kernel<N, OFDM_SCAN_MINW, true>), dim3(gseg), dim3(Plan<N>::WG), ScanGeom<N>::BYTES, s, rx, st);
                base += cnt;
            }
            if (a.seg_final) {                       // 1: the search ends with this part; 2: early finalize behind a first stage
                st.seg_base = 0;
                st.seg_launch = Plan<N>::SLOTS;
                st.seg_final = a.seg_final;
                hipLaunchKernelGGL((rx_sync_scan_kernel<N, OFDM_SCAN_MINW, true>), dim3(1), dim3(Plan<N>::WG), ScanGeom<N>::BYTES, s, rx, st);
            }
            return hipGetLastError();
        }
        if constexpr (ScanGeom<N>::BYTES > 65536) {                  // (a compile-time size: announced once)
            static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(&rx_sync_scan_kernel<N, OFDM_SCAN_MINW>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, int(ScanGeom<N>::BYTES));
            if (once != hipSuccess) return once;
        }
        hipLaunchKernelGGL((rx_sync_scan_kernel<N, OFDM_SCAN_MINW>), dim3(grid), dim3(Plan<N>::WG), ScanGeom<N>::BYTES, s, rx, a);
        return hipGetLastError();
    }
    // 3 waves per SIMD (168 VGPRs, a few spills off the trial path): 0.14 ms instead of 0.21 ms per 4369-frame launch; the
    // unconstrained build takes 192 VGPRs + 256 AGPRs (1 wave per SIMD), a 128-register build spills into the trial (0.20 ms)
    hipLaunchKernelGGL((rx_sync_kernel<N, 3>), dim3(grid), dim3(Plan<N>::WG), WgLds<N>::BYTES, s, rx, a);
    return hipGetLastError();
}

#define OFDM_DISPATCH_N(nfft, CALL)      \
    switch (nfft) {                      \
        case 64: return CALL(64);        \
        case 128: return CALL(128);      \
        case 256: return CALL(256);      \
        case 512: return CALL(512);      \
        case 1024: return CALL(1024);    \
        case 2048: return CALL(2048);    \
        case 4096: return CALL(4096);    \
        default: return hipErrorInvalidValue; \
    }

hipError_t launch_rx_demod(const RxDev& rx, const DemodArgs& a, hipStream_t s) {
#define CALL(n) launch_rx_demod_##n(rx, a, s)
    OFDM_DISPATCH_N(rx.nfft, CALL)
#undef CALL
}
hipError_t launch_rx_sync(const RxDev& rx, const SyncArgs& a, hipStream_t s) {
#define CALL(n) launch_sync_n<n>(rx, a, s)
    OFDM_DISPATCH_N(rx.nfft, CALL)
#undef CALL
}
template <int N>
static void zc_lane_table_n(int Ks, int S, const cf* zc, std::vector<cf>& out) {
    using PL = Plan<N>;
    const int h = Ks / 2;
    for (int LL = 0; LL < S; ++LL)
        for (int t = 0; t < PL::T; ++t)
            for (int j = 0; j < PL::C; ++j)
                for (int kl = 0; kl < PL::RL; ++kl) {
                    const int k = (t + PL::T * j) + PL::NC * kl;
                    cf z = cf{0.f, 0.f};
                    if (k >= N - h) z = zc[LL * Ks + (k - (N - h))];            // negative half of binsP(Ks)
                    if (k >= 1 && k <= h) z = zc[LL * Ks + (h + k - 1)];        // positive half (the later entry of a bin listed twice)
                    out[size_t(LL) * N + size_t(out_slot<N>(j, kl)) * PL::T + t] = z;
                }
}

std::vector<cf> rx_zc_lane_table(int nfft, int Ks, int S, const cf* zc) {
    std::vector<cf> out(size_t(S) * nfft, cf{0.f, 0.f});
    switch (nfft) {
        case 64: zc_lane_table_n<64>(Ks, S, zc, out); break;
        case 128: zc_lane_table_n<128>(Ks, S, zc, out); break;
        case 256: zc_lane_table_n<256>(Ks, S, zc, out); break;
        case 512: zc_lane_table_n<512>(Ks, S, zc, out); break;
        case 1024: zc_lane_table_n<1024>(Ks, S, zc, out); break;
        case 2048: zc_lane_table_n<2048>(Ks, S, zc, out); break;
        case 4096: zc_lane_table_n<4096>(Ks, S, zc, out); break;
    }
    return out;
}

int rx_sync_scan_block(const RxDev& rx) {
    switch (rx.nfft) {
        case 64: return scan_block_n<64>(rx);
        case 128: return scan_block_n<128>(rx);
        case 256: return scan_block_n<256>(rx);
        case 512: return scan_block_n<512>(rx);
        case 1024: return scan_block_n<1024>(rx);
        case 2048: return scan_block_n<2048>(rx);
        case 4096: return scan_block_n<4096>(rx);
    }
    return 0;
}
size_t rx_lds_bytes(int nfft) {
#define CALL(n) WgLds<n>::BYTES
    switch (nfft) {
        case 64: return CALL(64);
        case 128: return CALL(128);
        case 256: return CALL(256);
        case 512: return CALL(512);
        case 1024: return CALL(1024);
        case 2048: return CALL(2048);
        case 4096: return CALL(4096);
    }
#undef CALL
    return 0;
}

hipError_t launch_row_renorm(cf* eq, int Kd, int D, int n_frames, const int* tsr, hipStream_t s) {
    if (n_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(row_renorm_kernel, dim3(1), dim3(256), 0, s, eq, Kd, D, n_frames, tsr);
    return hipGetLastError();
}

hipError_t launch_despread(const cf* in, int in_row_stride, const cf* code, int dsss, int n_spread, int rows, cf* out, hipStream_t s) {
    if (rows <= 0 || n_spread <= 0) return hipSuccess;
    hipLaunchKernelGGL(despread_kernel, dim3(unsigned((n_spread + 63) / 64), unsigned(rows)), dim3(64), 0, s, in, in_row_stride, code,
                       dsss, n_spread, rows, out);
    return hipGetLastError();
}

template <int N>
static hipError_t launch_chan_time_n(const RxDev& rx, const cf* H, cf* htime, int n_rows, hipStream_t s) {
    const unsigned grid = unsigned((n_rows + Plan<N>::SLOTS - 1) / Plan<N>::SLOTS);
    hipLaunchKernelGGL(rx_chan_time_kernel<N>, dim3(grid), dim3(Plan<N>::WG), WgLds<N>::BYTES, s, rx, H, htime, n_rows);
    return hipGetLastError();
}
hipError_t launch_rx_chan_time(const RxDev& rx, const cf* H, cf* htime, int n_rows, hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    switch (rx.nfft) {
        case 64: return launch_chan_time_n<64>(rx, H, htime, n_rows, s);
        case 128: return launch_chan_time_n<128>(rx, H, htime, n_rows, s);
        case 256: return launch_chan_time_n<256>(rx, H, htime, n_rows, s);
        case 512: return launch_chan_time_n<512>(rx, H, htime, n_rows, s);
        case 1024: return launch_chan_time_n<1024>(rx, H, htime, n_rows, s);
        case 2048: return launch_chan_time_n<2048>(rx, H, htime, n_rows, s);
        case 4096: return launch_chan_time_n<4096>(rx, H, htime, n_rows, s);
    }
    return hipErrorInvalidValue;
}

// Grid caps of the two passes.  These kernels loop with the grid as stride; with the 4 096 workgroups (16 per CU) of rounds 1-2 they
// read 0.60-0.68 of the HBM rate, with up to 262 144 (the loop then runs once or twice) 0.68-0.82: the dispatcher keeps more loads in
// flight across many short workgroups than 16 long ones per CU do (profiles/r03_demap_grid_ab.txt).
#ifndef OFDM_DEMAP_CAP1
#define OFDM_DEMAP_CAP1 262144
#endif
#ifndef OFDM_DEMAP_CAP2
#define OFDM_DEMAP_CAP2 262144
#endif
hipError_t launch_demap(const DemapArgs& a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    const bool soft = a.soft0 || a.soft1;
    if (soft && a.mod != 2 && a.mod != 4 && a.mod != 6) return hipErrorInvalidValue;
    if (a.mod != 1 && a.mod != 2 && a.mod != 4 && a.mod != 6) return hipErrorInvalidValue;
    if (soft) {
        hipError_t e = hipMemsetAsync(a.partial, 0, DEMAP_PARTIALS * sizeof(double), s);
        if (e != hipSuccess) return e;
    }
    // pass 1 (one read of the symbols): hard bits and / or the distance sums sigma needs
    if (a.hard || soft) {
        const unsigned g1 = unsigned(std::min<int64_t>((a.n / 4 + 255) / 256 + 1, OFDM_DEMAP_CAP1));
#define OFDM_P1(M)                                                                                             \
    do {                                                                                                       \
        if (a.hard && soft)                                                                                    \
            hipLaunchKernelGGL((demap_pass1_kernel<M, true, (M != 1)>), dim3(g1), dim3(256), 0, s, a);         \
        else if (a.hard)                                                                                       \
            hipLaunchKernelGGL((demap_pass1_kernel<M, true, false>), dim3(g1), dim3(256), 0, s, a);            \
        else                                                                                                   \
            hipLaunchKernelGGL((demap_pass1_kernel<M, false, (M != 1)>), dim3(g1), dim3(256), 0, s, a);        \
    } while (0)
        switch (a.mod) {
            case 1: OFDM_P1(1); break;
            case 2: OFDM_P1(2); break;
            case 4: OFDM_P1(4); break;
            default: OFDM_P1(6); break;
        }
#undef OFDM_P1
    }
    // pass 2 (second read): the two metric arrays
    if (soft) {
        const unsigned grid = unsigned(std::min<int64_t>((a.n + 255) / 256, OFDM_DEMAP_CAP2));
        if (a.mod == 2)
            hipLaunchKernelGGL(demap_soft_kernel, dim3(grid), dim3(256), 0, s, a);
        else if (a.mod == 4)
            hipLaunchKernelGGL(demap_soft_qam_kernel<4>, dim3(grid), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(demap_soft_qam64_kernel, dim3(unsigned(std::min<int64_t>(a.n / 512 + 1, OFDM_DEMAP_CAP2))), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

}  // namespace ofdm
