// ofdm_device.hpp -- shared device-side helpers: workgroup FFT driver, bin membership, reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_core.hpp"

namespace ofdm {

// Numerology + constants shared by the RX kernels (passed by value).
struct RxDev {
    int nfft, cp, L;          // L = nfft + cp  (rx_b_len)
    int Ks, Kd;               // sync / data bin counts (even)
    int S, D;                 // synch_dat
    int MM;                   // S*Ks
    int stride;               // sync search stride
    float gate_mm;            // gate * MM
    float inv_ls;             // 1 / (S * (1 + 1/snr_ls))
    float inv_snr_data;       // 1 / snr_data
    float inv_snr_eqsync;     // 1 / snr_eqsync
    int bps;                  // bits per symbol of the fused de-mapper
    const cf* tw;             // [nfft]  exp(-2 pi i j / nfft)
    const cf* zc;             // [MM]    Zadoff-Chu reference
    const cf* zcp;            // [S][nfft] the same sequence in LANE order: zcp[LL*nfft + slot*T + t] = zc[LL*Ks + i] for the bin that
                              //           lane t holds in FFT output register `slot` (0 for bins outside the sync list); see rx_zc_lane_table
};

// Bin list of the reference: i -> k  for  binsP(K) = ([-K/2..-1, 1..K/2] + N) % N  (SynchAndChanEst.py:38-41).
// The inverse k -> i has a "negative half" and a "positive half"; they overlap only for K == N at k = N/2.
__device__ __forceinline__ bool bin_neg(int k, int K, int N, int& i) {
    i = k - (N - (K >> 1));
    return k >= N - (K >> 1);
}
__device__ __forceinline__ bool bin_pos(int k, int K, int& i) {
    i = (K >> 1) + k - 1;
    return k >= 1 && k <= (K >> 1);
}

// ------------------------------------------------------------------------------ reductions over the T lanes of a symbol
// v of lane (i with the DPP control applied); lanes of rows that ROW_MASK disables, and lanes without a source, read 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or_zero(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

// Sum over the 64 lanes of a wave, returned to every lane: six v_add_f32 with DPP operands (pairs, quads, half rows, rows, then
// the gfx9 row broadcasts carry row totals forward so that row 3 holds the wave total) and one v_readlane.  The __shfl_xor
// butterfly this replaces is six ds_bpermute_b32 plus their index arithmetic (~30 VALU and 6 LDS operations per reduction).
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_or_zero<0xB1, 0xF>(v);      // quad_perm:[1,0,3,2]
    v += dpp_or_zero<0x4E, 0xF>(v);      // quad_perm:[2,3,0,1]
    v += dpp_or_zero<0x141, 0xF>(v);     // row_half_mirror
    v += dpp_or_zero<0x140, 0xF>(v);     // row_mirror: every lane of a row holds the row's sum
    v += dpp_or_zero<0x142, 0xA>(v);     // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xC>(v);     // row_bcast:31 into rows 2 and 3: row 3 = wave total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// v of lane (i with the DPP control applied); lanes of disabled rows and lanes without a source read +inf
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or_inf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0x7f800000, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// minimum over the 64 lanes of a wave, returned to every lane (same six DPP steps as wave_sum)
__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, dpp_or_inf<0xB1, 0xF>(v));
    v = fminf(v, dpp_or_inf<0x4E, 0xF>(v));
    v = fminf(v, dpp_or_inf<0x141, 0xF>(v));
    v = fminf(v, dpp_or_inf<0x140, 0xF>(v));
    v = fminf(v, dpp_or_inf<0x142, 0xA>(v));
    v = fminf(v, dpp_or_inf<0x143, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int T>
__device__ __forceinline__ float lanes_sum(float v) {
    if constexpr (T >= 64) {
        return wave_sum(v);
    } else {
#pragma unroll
        for (int m = T >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        return v;
    }
}

// Workgroup barrier that orders LDS traffic ONLY.  Every barrier in these kernels protects an LDS exchange
// (no lane ever reads another lane's global-memory writes), so unlike __syncthreads() it does not drain
// the vector-memory queue: `s_waitcnt vmcnt(0)` before each of the 6 barriers per symbol would expose
// the latency of the symbol's own output stores (CDNA4 counts stores in vmcnt) and of any load in flight.
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS ordering inside ONE wavefront: LDS operations of a wave execute in order, so when a symbol is owned by a single wave
// (T <= 64) its exchanges need no s_barrier at all -- only the counter wait and a compiler fence.
__device__ __forceinline__ void wave_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <bool WAVE_LOCAL>
__device__ __forceinline__ void slot_sync() {
    if constexpr (WAVE_LOCAL)
        wave_fence();
    else
        wg_barrier();
}

// Sum over all T lanes of one symbol.  `red` = 8 floats of LDS scratch per symbol slot (only used when T > 64).
// Contains one workgroup barrier when T > 64 (all lanes of the workgroup must call it).
template <int T>
__device__ __forceinline__ float symbol_sum(float v, float* red, int t) {
    v = lanes_sum<T>(v);
    if constexpr (T > 64) {
        constexpr int NW = T / 64;
        if ((t & 63) == 0) red[t >> 6] = v;
        wg_barrier();
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w];
        return s;
    } else {
        return v;
    }
}

// arg-max with first-index tie-break over the T lanes of one symbol (idx < 0 means "no candidate")
template <int T>
__device__ __forceinline__ void symbol_argmax(float& val, int& idx, float* redv, int* redi, int t) {
    constexpr int W = T < 64 ? T : 64;
#pragma unroll
    for (int m = W >> 1; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(val, m, 64);
        const int oi = __shfl_xor(idx, m, 64);
        const bool take = (oi >= 0) && (idx < 0 || ov > val || (ov == val && oi < idx));
        if (take) {
            val = ov;
            idx = oi;
        }
    }
    if constexpr (T > 64) {
        constexpr int NW = T / 64;
        if ((t & 63) == 0) {
            redv[t >> 6] = val;
            redi[t >> 6] = idx;
        }
        wg_barrier();
        val = redv[0];
        idx = redi[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float ov = redv[w];
            const int oi = redi[w];
            const bool take = (oi >= 0) && (idx < 0 || ov > val || (ov == val && oi < idx));
            if (take) {
                val = ov;
                idx = oi;
            }
        }
    }
}

// ------------------------------------------------------------------------------ workgroup LDS carve
// [ SLOTS x ( exchange/staging region | 24 dwords reduction scratch ) | pass-1 twiddle table ]
template <int N>
struct WgLds {
    static constexpr int ELEMS = Plan<N>::LDS_ELEMS + (Plan<N>::LDS_ELEMS & 1);
    static constexpr int RED_CF = 12;                                 // 12 cf = 24 dwords of scratch (16 reductions + 8 spare)
    static constexpr int STRIDE = ELEMS + RED_CF;                     // cf units, even -> 16 B aligned
    static constexpr int W1_ELEMS = Plan<N>::W1_ELEMS;
    static constexpr size_t BYTES = (size_t(STRIDE) * Plan<N>::SLOTS + W1_ELEMS) * sizeof(cf);
};

// fills the workgroup's pass-1 twiddle table; the caller's first barrier (inside wg_fft) publishes it
template <int N>
__device__ __forceinline__ const cf* wg_init_w1(cf* smem, const cf* __restrict__ table, int tid) {
    cf* w1 = smem + WgLds<N>::STRIDE * Plan<N>::SLOTS;
    if constexpr (Plan<N>::THREE) {
        if (tid < WgLds<N>::W1_ELEMS) w1[tid] = w1_entry<N>(table, tid);
    }
    return w1;
}

// ------------------------------------------------------------------------------ workgroup FFT
// Forward N-point FFT of one symbol held as v[n0] = x[t + T*n0] across its T lanes.
// On return lane t holds bin k = (t + T*j) + NC*kl in v[out_slot<N>(j,kl)].
// Contains 1 (2-pass) or 3 (3-pass) workgroup barriers; the symbol's LDS region may still be read by
// other lanes on return, so the caller must barrier before overwriting it.
template <int N, class TW, bool WAVE_LOCAL = false>
__device__ __forceinline__ void wg_fft(cf (&v)[Plan<N>::P], cf* lds, const TW& tw, const cf* w1tab, int t) {
    static_assert(!WAVE_LOCAL || Plan<N>::T <= 64, "wave-local synchronisation needs one wave per symbol");
    fft_pass0_store<N>(v, lds, tw, t);
    slot_sync<WAVE_LOCAL>();
    if constexpr (Plan<N>::THREE) {
        fft_pass1_load<N>(v, lds, t);
        slot_sync<WAVE_LOCAL>();
        fft_pass1_store<N>(v, lds, w1tab, t);
        slot_sync<WAVE_LOCAL>();
    }
    fft_last_load<N>(v, lds, t);
    fft_last_dft<N>(v);
}

}  // namespace ofdm
