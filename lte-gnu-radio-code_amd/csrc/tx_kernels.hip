// tx_kernels.hip -- gfx950 kernels of the OFDM transmit path and the loop-back channel.
//
//   tx_modulate_kernel  bits -> QPSK/QAM symbols -> resource grid (ZC sync symbols) -> IFFT -> CP ->
//                       per-symbol power normalisation
//                       (reference: LEGACY/gr-ofdm-rx/python/txrx_mod/MultiAntennaSystem.py:113-218,
//                        SynchSignal.py:13-30)
//   channel_kernel      tapped-delay-line convolution + Philox/Box-Muller AWGN  (MultiAntennaSystem.py:221-260)
#include <cstdlib>
#include "ofdm_launch.hpp"

namespace ofdm {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = uint64_t(0xD2511F53u) * c0;
        const uint64_t p1 = uint64_t(0xCD9E8D57u) * c2;
        const uint32_t n0 = uint32_t(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = uint32_t(p1);
        const uint32_t n2 = uint32_t(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = uint32_t(p0);
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

__device__ __forceinline__ unsigned read_bits(const uint8_t* bits, int mode, int64_t bit0, int bps) {
    unsigned v = 0;
    if (mode == 2) {
        for (int b = 0; b < bps; ++b) v = (v << 1) | (bits[bit0 + b] & 1u);
    } else {
        for (int b = 0; b < bps; ++b) {
            const int64_t B = bit0 + b;
            v = (v << 1) | ((bits[B >> 3] >> (7 - (B & 7))) & 1u);
        }
    }
    return v;
}

// ---- the bit words of one lane's P bins.  KIND is compile-time: 2 / 4 / 6 = bits per constellation symbol of a one-bit-per-
// byte stream whose frames start 4-byte aligned (one or two wide loads per bin, ALL issued before the first is used: loads
// behind per-bin branches go out one at a time, each waiting for the one before -- 16 memory latencies per symbol);
// 10 / 12 / 14 = 8 + bits per symbol of a packed stream (8 bits per byte, MSB first: one or two byte loads per bin);
// 0 = anything else (BPSK, unaligned one-bit-per-byte streams), read bit by bit.
// Bin q reads constellation symbol base + li[q] of the frame's stream (li < 0: unused bin -> entry `base`, masked by the caller);
// 32-bit symbol numbers: a frame's stream is < 2^31 bits (checked by the host).
#define OFDM_KEEP16(r) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
                                         "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]))
#define OFDM_KEEP8(r) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]))
template <int P>
__device__ __forceinline__ void keep_loads(unsigned (&r)[P]) {
    static_assert(P == 16 || P == 8, "lane bins");
    if constexpr (P == 16) OFDM_KEEP16(r); else OFDM_KEEP8(r);
}

template <int P, int KIND>
struct TxFetch {
    unsigned w[(KIND == 6 || KIND == 14) ? 2 : 1][P];
};
__host__ __device__ __forceinline__ int tx_fetch_kind(int mode, int bps, bool al4) {
    if (bps != 2 && bps != 4 && bps != 6) return 0;
    if (mode == 2) return al4 ? bps : 0;
    return 8 + bps;
}
template <int P, int KIND>
__device__ __forceinline__ void tx_fetch_issue(const uint8_t* bits, unsigned base, const int (&li)[P], TxFetch<P, KIND>& f) {
    if constexpr (KIND == 4) {
#pragma unroll
        for (int q = 0; q < P; ++q) f.w[0][q] = *reinterpret_cast<const uint32_t*>(bits + (base + max(li[q], 0)) * 4u);
    } else if constexpr (KIND == 2) {
#pragma unroll
        for (int q = 0; q < P; ++q) f.w[0][q] = *reinterpret_cast<const uint16_t*>(bits + (base + max(li[q], 0)) * 2u);
    } else if constexpr (KIND == 6) {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            // six bytes at a 2-byte-aligned address: one (unaligned) dword + one short, kept as loaded -- any arithmetic between
            // a load and the point where all of them are in flight would make each bin wait for its own load
            const uint8_t* p = bits + (base + max(li[q], 0)) * 6u;
            uint32_t lo;
            __builtin_memcpy(&lo, p, 4);
            f.w[0][q] = lo;
            f.w[1][q] = *reinterpret_cast<const uint16_t*>(p + 4);
        }
    } else if constexpr (KIND == 12 || KIND == 10) {
        constexpr unsigned sh = KIND == 12 ? 1u : 2u;                       // symbols per byte = 1 << sh
#pragma unroll
        for (int q = 0; q < P; ++q) f.w[0][q] = bits[(base + max(li[q], 0)) >> sh];
    } else if constexpr (KIND == 14) {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            // two byte loads, the second only where the symbol reaches into the next byte (one unaligned two-byte load per bin was
            // measured 3-7 % slower: profiles/r03_tx_ab.txt)
            const unsigned b0 = (base + max(li[q], 0)) * 6u;
            f.w[0][q] = bits[b0 >> 3];
            f.w[1][q] = bits[(b0 >> 3) + ((b0 & 7u) > 2u ? 1u : 0u)];
        }
    }
    // one empty asm with every result as an operand: the loads above are all issued before the first of them is used
    if constexpr (KIND != 0) keep_loads(f.w[0]);
    if constexpr (KIND == 6 || KIND == 14) keep_loads(f.w[1]);
}
// the bps-bit value, MSB first
template <int P, int KIND>
__device__ __forceinline__ unsigned tx_fetch_value(const uint8_t* bits, int mode, int bps, unsigned base, int li, const TxFetch<P, KIND>& f,
                                                   int q) {
    if constexpr (KIND == 4) {
        const uint32_t w = f.w[0][q];
        return ((w & 1u) << 3) | ((w >> 6) & 4u) | ((w >> 15) & 2u) | ((w >> 24) & 1u);
    } else if constexpr (KIND == 2) {
        return ((f.w[0][q] & 1u) << 1) | ((f.w[0][q] >> 8) & 1u);
    } else if constexpr (KIND == 6) {
        const uint32_t lo = f.w[0][q], hi = f.w[1][q];
        return ((lo & 1u) << 5) | (((lo >> 8) & 1u) << 4) | (((lo >> 16) & 1u) << 3) | (((lo >> 24) & 1u) << 2) | ((hi & 1u) << 1) |
               ((hi >> 8) & 1u);
    } else if constexpr (KIND == 12) {
        return (f.w[0][q] >> (4u - ((base + max(li, 0)) & 1u) * 4u)) & 15u;
    } else if constexpr (KIND == 10) {
        return (f.w[0][q] >> (6u - ((base + max(li, 0)) & 3u) * 2u)) & 3u;
    } else if constexpr (KIND == 14) {
        return (((f.w[0][q] << 8) | f.w[1][q]) >> (10u - (((base + max(li, 0)) * 6u) & 7u))) & 63u;
    } else {
        return read_bits(bits, mode, int64_t(base + max(li, 0)) * bps, bps);
    }
}

// MultiAntennaSystem.py:156-178 (BPSK, QPSK); 16/64-QAM: 3GPP TS 36.211 7.1 (extension)
__device__ __forceinline__ cf map_symbol(unsigned v, int bps) {
    if (bps == 2) {
        constexpr float c = 0.70710678118654752f;
        return cf{(v & 2u) ? -c : c, (v & 1u) ? -c : c};
    }
    if (bps == 1) return cf{v ? 1.f : -1.f, 0.f};
    if (bps == 4) {
        constexpr float s = 0.31622776601683794f;   // 1/sqrt(10)
        const float si = (v & 8u) ? -1.f : 1.f, sq = (v & 4u) ? -1.f : 1.f;
        const float ai = (v & 2u) ? 3.f : 1.f, aq = (v & 1u) ? 3.f : 1.f;
        return cf{si * ai * s, sq * aq * s};
    }
    constexpr float s = 0.15430334996209191f;       // 1/sqrt(42)
    const float si = (v & 32u) ? -1.f : 1.f, sq = (v & 16u) ? -1.f : 1.f;
    // |I| = 4 - s2*(2 - s4) with s = 1-2b : (b2,b4) -> 3,1,5,7 for 00,01,10,11
    const float s2i = (v & 8u) ? -1.f : 1.f, s4i = (v & 2u) ? -1.f : 1.f;
    const float s2q = (v & 4u) ? -1.f : 1.f, s4q = (v & 1u) ? -1.f : 1.f;
    return cf{si * (4.f - s2i * (2.f - s4i)) * s, sq * (4.f - s2q * (2.f - s4q)) * s};
}

// ---- shared pieces of the fused kernel and of the decomposed stage kernels: the SAME device functions, so that the chain
// map -> grid -> IFFT -> CP equals the fused kernel bit for bit by construction.

// value of resource-grid bin n of a DATA symbol (:150-183): list index i of bin n in binsP(K), K = Kd + n_pilots; pilots sit
// at the list indices plist[0..n_pilots) (ascending), data symbols fill the remaining list entries in order.
// A bin listed twice (K == N: bin N/2) keeps its later (positive-half) entry.  Returns false for an unused bin.
__device__ __forceinline__ bool data_bin_index(int n, int K, int N, int& i) {
    bool has = bin_neg(n, K, N, i);
    int ip;
    if (bin_pos(n, K, ip)) {
        has = true;
        i = ip;
    }
    return has;
}

// time-domain samples from the FFT of the conjugated grid row: ifft(X) = conj(fft(conj(X))) / N   (:199)
template <int N>
__device__ __forceinline__ void tx_time_from_fft(cf (&v)[Plan<N>::P]) {
    const float invn = 1.f / float(N);
#pragma unroll
    for (int s = 0; s < Plan<N>::P; ++s) v[s] = cscale(cconj(v[s]), invn);
}

// x (register slot order: sample m = (t + T*j) + NC*kl in x[out_slot(j,kl)]) -> CP-extended, power-normalised symbol at `o`
// (:200-218): energy and mean over the CP-extended symbol in one pass (CP samples count twice), then one scale.
// Two halves, so that the fused kernel can issue the next symbol's loads between them (ahead of this symbol's stores).
// RAW = true: x holds the FFT of the CONJUGATED grid row as it leaves wg_fft, i.e. N * conj(time sample).  1/N is a power of two
// and the conjugation a sign: both are exact, so they are folded into the reductions' results and into the final scale -- the
// staged samples stay raw and tx_cp_store applies (scale/N, -scale/N).  Every float the staged pipeline (IFFT stage: conj and
// 1/N explicitly, then this function with RAW = false) produces is reproduced bit for bit.
// STAGE = false: nothing is staged in LDS (the caller stores from its registers, tx_cp_store_direct); only the reductions remain.
template <int N, bool RAW = false, bool STAGE = true>
__device__ __forceinline__ float tx_cp_norm_stage(const TxDev& tx, const cf (&x)[Plan<N>::P], cf* lds, float* red, int t) {
    using PL = Plan<N>;
    constexpr int T = PL::T;
    float e = 0.f, sx = 0.f, sy = 0.f;
    const int cp0 = N - tx.cp;                                            // samples m >= cp0 appear twice (CP)
#pragma unroll
    for (int j = 0; j < PL::C; ++j) {
#pragma unroll
        for (int kl = 0; kl < PL::RL; ++kl) {
            const int m = (t + T * j) + PL::NC * kl;
            const cf xv = x[out_slot<N>(j, kl)];
            if constexpr (STAGE) lds[m] = xv;
            // a register slot holds the T consecutive samples [base, base + T): the CP weight is decided per slot on the scalar
            // unit wherever the slot lies wholly on one side of the CP boundary (x2 and x1 are exact: same sums either way)
            const int base = T * j + PL::NC * kl;
            float w;
            if (T >= 64 && base >= cp0)
                w = 2.f;
            else if (T >= 64 && base + T - 1 < cp0)
                w = 1.f;
            else
                w = (m >= cp0) ? 2.f : 1.f;
            e += w * cnorm2(xv);
            sx += w * xv.x;
            sy += w * xv.y;
        }
    }
    // the three sums share one exchange (and the barrier that publishes the staged samples)
    e = lanes_sum<T>(e);
    sx = lanes_sum<T>(sx);
    sy = lanes_sum<T>(sy);
    if constexpr (T > 64) {
        if ((t & 63) == 0) {
            float* r3 = red + (t >> 6) * 3;
            r3[0] = e;
            r3[1] = sx;
            r3[2] = sy;
        }
    }
    if constexpr (STAGE || T > 64) wg_barrier();
    if constexpr (T > 64) {
        e = sx = sy = 0.f;
#pragma unroll
        for (int w = 0; w < T / 64; ++w) {
            e += red[w * 3];
            sx += red[w * 3 + 1];
            sy += red[w * 3 + 2];
        }
    }
    constexpr float invn = 1.f / float(N);
    if constexpr (RAW) {
        e *= invn * invn;
        sx *= invn;
        sy *= -invn;
    }
    // :204-218 with the hardware reciprocal square root (1 ulp; the IEEE divisions and square roots this replaces were ~60 VALU
    // per wave and symbol):  a1 = sqrt(L / e),  out = a1 / sqrt(var)
    const float Lf = float(tx.L), invL = 1.f / Lf;
    const float a1 = (e > 1e-30f) ? sqrtf(Lf) * __builtin_amdgcn_rsqf(e) : 1.f;   // :204-205
    const float mx = a1 * sx * invL, my = a1 * sy * invL;
    const float var = a1 * a1 * e * invL - (mx * mx + my * my);           // np.var(data_time) :213
    const float sc = a1 * __builtin_amdgcn_rsqf(var);                     // :218
    return RAW ? sc * invn : sc;
}

template <int N, bool CONJ = false>
__device__ __forceinline__ void tx_cp_store(const TxDev& tx, const cf* lds, float scale, int t, cf* o, bool active) {
    constexpr int T = Plan<N>::T;
    const float scale_im = CONJ ? -scale : scale;
    if (active) {
        // sample j of the symbol is x[(j - cp) mod N]; with cp and L even a pair (j, j+1) never straddles the wrap: 16 B per lane
        if (((tx.cp | tx.L) & 1) == 0 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
            const float4* l4 = reinterpret_cast<const float4*>(lds);
            float4* o4 = reinterpret_cast<float4*>(o);
            const int h = tx.cp >> 1, half = tx.L >> 1;
            for (int j0 = t; j0 < half; j0 += 4 * T) {
                float4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {                              // four reads in flight, then four stores
                    int m = min(j0 + u * T, half - 1) - h;
                    if (m < 0) m += N / 2;
                    q[u] = l4[m];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u * T;
                    q[u].x *= scale;
                    q[u].y *= scale_im;
                    q[u].z *= scale;
                    q[u].w *= scale_im;
                    if (j < half) {
                        typedef float f4 __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(f4{q[u].x, q[u].y, q[u].z, q[u].w}, reinterpret_cast<f4*>(o4 + j));   // streamed out once
                    }
                }
            }
        } else {
            for (int j = t; j < tx.L; j += T) {
                int m = j - tx.cp;
                if (m < 0) m += N;
                o[j] = cf{lds[m].x * scale, lds[m].y * scale_im};
            }
        }
    }
}

// The same samples straight from the registers the transform left them in (one symbol per workgroup, T >= 64): register slot
// (j, kl) of the T lanes is T consecutive time samples, so a wave's store instruction covers 512 contiguous bytes; the cyclic prefix
// is a second store from the one or two slots that reach into the last cp samples.  No staging copy in LDS (N writes + L reads per
// symbol) and neither of the two barriers that fenced it.  The products are those of tx_cp_store: the same floats.
template <int N>
__device__ __forceinline__ void tx_cp_store_direct(const TxDev& tx, const cf (&x)[Plan<N>::P], float scale, int t, cf* o, bool active) {
    using PL = Plan<N>;
    constexpr int T = PL::T;
    typedef float f2 __attribute__((ext_vector_type(2)));
    if (!active) return;
    const int cp0 = N - tx.cp;
    cf* body = o + tx.cp + t;
    cf* pre = o + t - cp0;
#pragma unroll
    for (int j = 0; j < PL::C; ++j) {
#pragma unroll
        for (int kl = 0; kl < PL::RL; ++kl) {
            const int base = T * j + PL::NC * kl;
            const cf xv = x[out_slot<N>(j, kl)];
            const f2 q{xv.x * scale, xv.y * -scale};
            __builtin_nontemporal_store(q, reinterpret_cast<f2*>(body + base));
            if (base + T - 1 >= cp0) {                                    // (scalar: most slots lie wholly before the prefix)
                if (base + t >= cp0) __builtin_nontemporal_store(q, reinterpret_cast<f2*>(pre + base));
            }
        }
    }
}

#ifndef OFDM_TX_DIRECT
#define OFDM_TX_DIRECT 1        // 0: the fused kernel stages every symbol in LDS like the stage kernels do (A/B in DESIGN.md 4.3)
#endif

// One workgroup slot walks symbols unit, unit + stride, ...: twiddles and the pass-1 table are set up once per workgroup.
// (Requesting the next symbol's bit words ahead of this symbol's stores -- the software pipeline of the receive kernel -- was
// measured and bought nothing here: the kernel is VALU-issue-bound at 4 waves per SIMD, not waiting for memory.)
#ifndef OFDM_TX_MINW
#define OFDM_TX_MINW 4          // waves per SIMD the fused transmit kernel is compiled for from 1024-pt up (a 128-VGPR budget: 0-2
                                // spilled values once the symbol leaves from registers; A/B in DESIGN.md 4.3).  Smaller sizes and
                                // packed 64-QAM (two byte loads per bin in flight: 13-16 spills at 128) keep 3
#endif
constexpr int TX_LUT_ELEMS = 72;         // 64 points + the zero entry, rounded up
template <int N, int KIND>
__global__ void __launch_bounds__(Plan<N>::WG, (Plan<N>::SLOTS == 1 && KIND != 14) ? OFDM_TX_MINW : 3) tx_modulate_kernel(TxDev tx, ModArgs a) {
    using PL = Plan<N>;
    constexpr int T = PL::T, P = PL::P;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int slot = (T >= 64) ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T;    // a wave lies inside one slot: uniform
    const int t = tid % T;
    cf* smem = reinterpret_cast<cf*>(smem_raw);
    cf* lds = smem + slot * WgLds<N>::STRIDE;
    float* red = reinterpret_cast<float*>(lds + WgLds<N>::ELEMS);
    const cf* w1tab = wg_init_w1<N>(smem, tx.tw, tid);

    LaneTwiddles<N> tw;                  // all 15 pass-0 twiddles of a lane in VGPRs (the compact form costs 11 complex products per symbol)
    load_twiddles(tw, tx.tw, t);
    // constellation table in LDS behind the pass-1 twiddles: entry e < 2^bps = conj(map_symbol(e)) (the grid row is transformed
    // conjugated), entry 2^bps = 0 for unused bins.  One ds_read_b64 per bin replaces ~12 VALU of sign / level arithmetic.
    cf* lut = smem + WgLds<N>::STRIDE * PL::SLOTS + WgLds<N>::W1_ELEMS;
    const int n_pts = 1 << tx.bps;
    for (int e = tid; e <= n_pts; e += PL::WG) {
        const cf X = map_symbol(unsigned(e), tx.bps);
        lut[e] = e < n_pts ? cf{X.x, -X.y} : cf{0.f, 0.f};
    }
    wg_barrier();                        // the table is read before the first FFT barrier

    const int SD = tx.S + tx.D;
    const int64_t n_units = int64_t(a.n_frames) * a.n_sym;
    // trip count of the workgroup's first slot: workgroup-uniform, so the loop needs no vote (and no wait for its stores)
    const int64_t stride = int64_t(gridDim.x) * PL::SLOTS, first = int64_t(blockIdx.x) * PL::SLOTS;
    const int n_iter = first < n_units ? int((n_units - first + stride - 1) / stride) : 0;
    // (frame, symbol) of the slot's unit, advanced by the stride without a division per symbol
    const int64_t unit0 = first + slot;
    int frame_u = int(unit0 / a.n_sym), sym_u = int(unit0 % a.n_sym);
    const int step_f = int(stride / a.n_sym), step_s = int(stride % a.n_sym);
    for (int it = 0; it < n_iter; ++it) {
        const bool active = unit0 + int64_t(it) * stride < n_units;
        const int frame = active ? frame_u : 0;
        const int sym = active ? sym_u : 0;
        frame_u += step_f;
        sym_u += step_s;
        if (sym_u >= a.n_sym) {
            sym_u -= a.n_sym;
            ++frame_u;
        }
        const int pat = sym / SD, r = sym - pat * SD;
        const bool is_sync = r < tx.S;                                    // symbol_pattern == 0  (:136)
        const unsigned base = unsigned(pat * tx.D + (r - tx.S)) * unsigned(tx.Kd);   // loop_data (:134,180) * Kd
        const uint8_t* fbits = a.bits ? a.bits + int64_t(frame) * a.bits_stride : nullptr;
        if constexpr (PL::SLOTS == 1) {
            // Every sync symbol of every frame is the same L samples (synch_state never advances, :143-147): where the
            // workgroup holds one symbol (so the branch is workgroup-uniform and skips whole barriers) it is copied from the
            // handle's finished sync symbol -- the output of these very device functions, bit for bit -- instead of being
            // transformed again: a quarter of the symbols of a [1, 3] pattern.
            if (is_sync && a.sync_time) {
                if (active) {
                    const cf* src = a.sync_time + int64_t(r) * tx.L;
                    cf* o = a.iq + int64_t(frame) * a.frame_stride + int64_t(sym) * tx.L;
                    if ((tx.L & 1) == 0 && ((reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) {
                        typedef float f4 __attribute__((ext_vector_type(4)));
                        for (int j = t; j < (tx.L >> 1); j += T)
                            __builtin_nontemporal_store(reinterpret_cast<const f4*>(src)[j], reinterpret_cast<f4*>(o) + j);
                    } else {
                        for (int j = t; j < tx.L; j += T) o[j] = src[j];
                    }
                }
                continue;
            }
        }

        // resource grid row X[n] (:135-183), conjugated: ifft(X) = conj(fft(conj(X))) / N.
        // List index of bin k in binsP(K) (-1 = unused), branch-free: positive half i = K/2 + k - 1 for 1 <= k <= K/2, negative
        // half i = k - (N - K/2) for k >= N - K/2; a bin listed twice (K == N: bin N/2) keeps its later, positive-half entry
        // (the same rule as data_bin_index).  The lane's bin numbers go through an opaque copy of t: hoisted out of the symbol
        // loop, the 16 list indices and addresses cost ~100 VGPRs and the occupancy with them.
        // Where a workgroup holds one symbol and sync symbols are copied (above), every symbol that gets here is a DATA symbol:
        // the list indices depend on the lane only and hipcc may keep them in 16 VGPRs across the loop.  Otherwise (small
        // sizes, handles without a finished sync symbol) the lane index passes through an opaque copy so that nothing is hoisted.
        int tt = t;
        if (!(PL::SLOTS == 1 && a.sync_time != nullptr)) asm volatile("" : "+v"(tt));
        const int hk = (is_sync ? tx.Ks : tx.Kd) >> 1;
        const int off_pos = hk - 1, off_neg = hk - N;
        int li[P];
#pragma unroll
        for (int n0 = 0; n0 < P; ++n0) {
            const int k = tt + T * n0;
            const bool pos = unsigned(k - 1) < unsigned(hk), neg = k >= N - hk;
            li[n0] = ((pos || neg) && active) ? k + (pos ? off_pos : off_neg) : -1;
        }
        cf v[P];
        if (is_sync) {
            // x_in = zchu[0:Ks] on used_bins_synch; synch_state never advances (:143-147).  All entries requested together.
            unsigned zr[P], zi[P];
#pragma unroll
            for (int n0 = 0; n0 < P; ++n0) {
                const cf z = tx.zc[max(li[n0], 0)];
                zr[n0] = __float_as_uint(z.x);
                zi[n0] = __float_as_uint(z.y);
            }
            keep_loads(zr);
            keep_loads(zi);
#pragma unroll
            for (int n0 = 0; n0 < P; ++n0) {
                const float use = li[n0] >= 0 ? 1.f : 0.f;
                v[n0] = cf{__uint_as_float(zr[n0]) * use, -__uint_as_float(zi[n0]) * use};
            }
        } else if (fbits) {
            TxFetch<P, KIND> f;
            tx_fetch_issue<P, KIND>(fbits, base, li, f);
#pragma unroll
            for (int n0 = 0; n0 < P; ++n0) {
                const unsigned val = tx_fetch_value<P, KIND>(fbits, a.bits_mode, tx.bps, base, li[n0], f, n0);
                v[n0] = lut[li[n0] >= 0 ? int(val) : n_pts];              // conj(map_symbol(val)), or 0 for an unused bin
            }
        } else {
#pragma unroll
            for (int n0 = 0; n0 < P; ++n0) v[n0] = cf{0.f, 0.f};
        }
        wg_fft<N>(v, lds, tw, w1tab, t);                                         // :199
        // conj and 1/N of ifft(X) = conj(fft(conj(X))) / N are exact: folded into the reductions and the final scale
        cf* o = a.iq + int64_t(frame) * a.frame_stride + int64_t(sym) * tx.L;
        if constexpr (PL::SLOTS == 1 && OFDM_TX_DIRECT) {
            const float scale = tx_cp_norm_stage<N, true, false>(tx, v, lds, red, t);
            tx_cp_store_direct<N>(tx, v, scale, t, o, active);
        } else {
            wg_barrier();
            const float scale = tx_cp_norm_stage<N, true>(tx, v, lds, red, t);
            tx_cp_store<N, true>(tx, lds, scale, t, o, active);
            wg_barrier();                                                        // the staged symbol has been read by every lane
        }
    }
}

// ------------------------------------------------------------------------------------------ decomposed stages
// (random_bit_source -> ConstellationModulation -> OFDM_Modulation -> IFFT -> CyclicPrefix -> SynchDataMux,
//  block names from LEGACY/gr-ofdm-tx/grc/RXtransmit_6.grc:701-975; the reference holds no code for them.)

// one bit per byte from a counter-based generator: bit k of the stream = bit (k % 32) of word (k / 32) % 4 of
// Philox4x32-10(counter = k / 128, key = seed).  Any [offset, offset + n) window can be produced independently.
__global__ void __launch_bounds__(256) tx_random_bits_kernel(uint64_t seed, uint64_t offset, uint8_t* out, int64_t n) {
    const uint64_t blk0 = offset >> 7;
    const int64_t nblk = int64_t(((offset + uint64_t(n) + 127) >> 7) - blk0);
    for (int64_t b = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; b < nblk; b += int64_t(gridDim.x) * blockDim.x) {
        const uint64_t c = blk0 + uint64_t(b);
        uint32_t w[4];
        philox4x32_10(uint32_t(c), uint32_t(c >> 32), 0u, 0u, uint32_t(seed), uint32_t(seed >> 32), w);
        for (int j = 0; j < 128; ++j) {
            const uint64_t k = (c << 7) + uint64_t(j);
            if (k >= offset && k < offset + uint64_t(n)) out[k - offset] = uint8_t((w[j >> 5] >> (j & 31)) & 1u);
        }
    }
}

// ConstellationModulation: bits -> symbols, MSB-first per symbol (:156-178 / TS 36.211 for 16/64-QAM)
__global__ void __launch_bounds__(256) tx_map_kernel(const uint8_t* bits, int bits_mode, int bps, int64_t n_sym, cf* out) {
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_sym; i += int64_t(gridDim.x) * blockDim.x)
        out[i] = map_symbol(read_bits(bits, bits_mode, i * bps, bps), bps);
}

// OFDM_Modulation: rows of Kd symbols -> rows of N grid bins (data on binsP(Kd + n_pilots) minus the pilot entries)
__global__ void __launch_bounds__(256) tx_grid_kernel(TxDev tx, GridArgs a) {
    const int N = tx.nfft, K = tx.Kd + a.n_pilots;
    const int64_t total = a.n_rows * N;
    for (int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; g < total; g += int64_t(gridDim.x) * blockDim.x) {
        const int64_t row = g / N;
        const int n = int(g - row * N);
        cf X = cf{0.f, 0.f};
        int i;
        if (data_bin_index(n, K, N, i)) {
            int before = 0;
            bool is_pilot = false;
            for (int p = 0; p < a.n_pilots; ++p) {
                const int pi = a.pilots[p];
                before += pi < i;
                is_pilot |= pi == i;
            }
            X = is_pilot ? a.pilot_value : a.sym[row * tx.Kd + (i - before)];
        }
        a.grid[g] = X;
    }
}

// rows [S][N] of the sync symbol's resource grid: zchu[0:Ks] on binsP(Ks), the same segment for every sync symbol (:143-147)
__global__ void __launch_bounds__(256) tx_sync_grid_kernel(TxDev tx, cf* grid) {
    const int N = tx.nfft;
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < tx.S * N; g += gridDim.x * blockDim.x) {
        const int n = g % N;
        int i;
        grid[g] = data_bin_index(n, tx.Ks, N, i) ? tx.zc[i] : cf{0.f, 0.f};
    }
}

// IFFT and/or CyclicPrefix on rows:  IFFT: [N] grid bins -> [N] time samples;  CP: [N] time samples -> [L] CP-extended,
// power-normalised;  both: the fused symbol synthesis.
template <int N, bool DO_IFFT, bool DO_CP>
__global__ void __launch_bounds__(Plan<N>::WG) tx_time_kernel(TxDev tx, TimeArgs a) {
    using PL = Plan<N>;
    constexpr int T = PL::T, P = PL::P;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    const int slot = (T >= 64) ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T;    // a wave lies inside one slot: uniform
    const int t = tid % T;
    cf* smem = reinterpret_cast<cf*>(smem_raw);
    cf* lds = smem + slot * WgLds<N>::STRIDE;
    float* red = reinterpret_cast<float*>(lds + WgLds<N>::ELEMS);
    const cf* w1tab = wg_init_w1<N>(smem, tx.tw, tid);
    const int64_t row = int64_t(blockIdx.x) * PL::SLOTS + slot;
    const bool active = row < a.n_rows;
    const cf* in = a.in + (active ? row : 0) * N;
    cf v[P];
    if constexpr (DO_IFFT) {
        LaneTwiddles<N> tw;                                                            // as the fused kernel: same numbers
        load_twiddles<N>(tw, tx.tw, t);
#pragma unroll
        for (int n0 = 0; n0 < P; ++n0) v[n0] = active ? cconj(in[t + T * n0]) : cf{0.f, 0.f};
        wg_fft<N>(v, lds, tw, w1tab, t);
        wg_barrier();
        tx_time_from_fft<N>(v);
    } else {
#pragma unroll
        for (int j = 0; j < PL::C; ++j) {
#pragma unroll
            for (int kl = 0; kl < PL::RL; ++kl)
                v[out_slot<N>(j, kl)] = active ? in[(t + T * j) + PL::NC * kl] : cf{0.f, 0.f};
        }
    }
    if constexpr (DO_CP) {
        const float scale = tx_cp_norm_stage<N>(tx, v, lds, red, t);
        tx_cp_store<N>(tx, lds, scale, t, a.out + (active ? row : 0) * tx.L, active);
    } else {
        if (active) {
#pragma unroll
            for (int j = 0; j < PL::C; ++j) {
#pragma unroll
                for (int kl = 0; kl < PL::RL; ++kl) a.out[row * N + (t + T * j) + PL::NC * kl] = v[out_slot<N>(j, kl)];
            }
        }
    }
}

// SynchDataMux: S sync symbols in front of every D data symbols (symbols are L samples)
__global__ void __launch_bounds__(256) tx_mux_kernel(TxDev tx, MuxArgs a) {
    const int L = tx.L, SD = tx.S + tx.D;
    // output symbols by a grid-stride loop over blockIdx.y: any pattern length and any symbol count fit one launch
    for (int64_t s_out = blockIdx.y; s_out < a.n_out_sym; s_out += gridDim.y) {
        const int64_t pat = s_out / SD;
        const int r = int(s_out - pat * SD);
        const cf* src = r < tx.S ? a.sync_time + int64_t(r) * L : a.data + (pat * tx.D + (r - tx.S)) * L;
        cf* dst = a.out + s_out * L;
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < L; j += gridDim.x * blockDim.x) dst[j] = src[j];
    }
}

// ------------------------------------------------------------------------------------------ channel
// Two consecutive output samples per thread: one Philox4x32-10 block (four 32-bit words) feeds both Box-Muller pairs, and the
// pair leaves in one 16-byte store where the row is aligned.  Noise is N(0, noise_std^2) per component from the hardware
// transcendentals (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32 take their argument in revolutions): a statistical model of
// MultiAntennaSystem.py:258, not a bit-pinned one.
__global__ void __launch_bounds__(256) channel_kernel(ChanArgs a) {
    const int frame = blockIdx.y;
    const cf* in = a.in + int64_t(frame) * a.in_stride;
    const cf* taps = a.taps + (a.per_frame_taps ? int64_t(frame) * a.n_taps : 0);
    cf* out = a.out + int64_t(frame) * a.out_stride;
    const bool wide = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const int64_t n_pairs = (a.out_len + 1) >> 1;
    for (int64_t pr = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; pr < n_pairs; pr += int64_t(gridDim.x) * blockDim.x) {
        const int64_t n = pr * 2;
        cf acc0 = cf{0.f, 0.f}, acc1 = cf{0.f, 0.f};
        for (int l = 0; l < a.n_taps; ++l) {                          // np.convolve(tx_sig, chan) :227
            const cf h = taps[l];
            const int64_t j = n - l;
            if (j >= 0 && j < a.in_len) acc0 = acc0 + cmul(h, in[j]);
            if (j + 1 >= 0 && j + 1 < a.in_len) acc1 = acc1 + cmul(h, in[j + 1]);
        }
        if (a.noise_std > 0.f) {                                      // :258
            uint32_t rnd[4];
            philox4x32_10(uint32_t(pr), uint32_t(uint64_t(pr) >> 32), uint32_t(frame), 0u, uint32_t(a.seed), uint32_t(a.seed >> 32), rnd);
            constexpr float k = 2.3283064365386963e-10f;              // 2^-32
            const float r0 = a.noise_std * __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf((float(rnd[0]) + 0.5f) * k));
            const float r1 = a.noise_std * __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf((float(rnd[2]) + 0.5f) * k));
            const float u0 = (float(rnd[1]) + 0.5f) * k, u1 = (float(rnd[3]) + 0.5f) * k;     // sqrt(-2 ln u) = sqrt(-2 ln2 log2 u)
            acc0.x += r0 * __builtin_amdgcn_cosf(u0);
            acc0.y += r0 * __builtin_amdgcn_sinf(u0);
            acc1.x += r1 * __builtin_amdgcn_cosf(u1);
            acc1.y += r1 * __builtin_amdgcn_sinf(u1);
        }
        if (wide && n + 1 < a.out_len) {
            *reinterpret_cast<float4*>(out + n) = float4{acc0.x, acc0.y, acc1.x, acc1.y};   // (a non-temporal hint here: 8 % slower)
        } else {
            out[n] = acc0;
            if (n + 1 < a.out_len) out[n + 1] = acc1;
        }
    }
}

// ------------------------------------------------------------------------------------------ bandwidth probes
// mode 0: float4 copy (the chip's achievable HBM rate for bench.py's roofline context)
// mode 1: the demod kernel's access pattern with no arithmetic: per symbol read N*8 B (skipping cp*8 B), write Kd*8 B
__global__ void __launch_bounds__(256) probe_kernel(const float4* __restrict__ in, float4* __restrict__ out, int64_t n16, int mode,
                                                     int sym_in16, int gap16, int sym_out16, int64_t n_sym) {
    if (mode == 0) {
        // four independent non-temporal 16 B loads in flight per lane, 16 Ki workgroups: the best plain-copy shape found on
        // these chips by tools/ubench/copy_bw.hip (5.0-5.4 TB/s; one load per iteration on 4 Ki workgroups gives 4.5-4.7)
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4* src = reinterpret_cast<const f4*>(in);
        f4* dst = reinterpret_cast<f4*>(out);
        const int64_t stride = int64_t(gridDim.x) * blockDim.x;
        int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < n16; i += 4 * stride) {
            f4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
            for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
        }
        for (; i < n16; i += stride) dst[i] = src[i];
    } else {
        for (int64_t s = blockIdx.x; s < n_sym; s += gridDim.x) {
            const float4* src = in + s * (sym_in16 + gap16) + gap16;
            float4* dst = out + s * sym_out16;
            // the same cache policy as the demod kernel's streams (non-temporal loads and stores)
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
            for (int i = threadIdx.x; i < sym_in16; i += blockDim.x) acc += __builtin_nontemporal_load(reinterpret_cast<const f4*>(src) + i);
            for (int i = threadIdx.x; i < sym_out16; i += blockDim.x) __builtin_nontemporal_store(acc, reinterpret_cast<f4*>(dst) + i);
        }
    }
}

// Grid of the two probes: as many workgroups as there is work (one unrolled trip per thread / one symbol per workgroup).  With the
// 16 384 / 8 192 looping workgroups of rounds 1-2 the copy read 5.2-5.9 TB/s and the pattern 5.9-6.2 where these read 6.0-6.4 and
// 6.3-6.4 on the same chip: the probes are the yardstick the demod kernel is held against, so they get the better grid.
#ifndef OFDM_PROBE_CAP0
#define OFDM_PROBE_CAP0 4194304
#endif
#ifndef OFDM_PROBE_CAP1
#define OFDM_PROBE_CAP1 4194304
#endif
hipError_t launch_probe(const void* in, void* out, int64_t n16, int mode, int sym_in16, int gap16, int sym_out16, int64_t n_sym,
                        hipStream_t s) {
    const size_t lds = size_t(mode >> 4) * 1024;      // occupancy experiment: mode = base + 16 * KiB of (unused) dynamic LDS
    mode &= 15;
    const unsigned grid = mode == 0 ? unsigned(std::max<int64_t>(256, std::min<int64_t>((n16 / 4 + 255) / 256, OFDM_PROBE_CAP0))) : unsigned(std::min<int64_t>(n_sym, OFDM_PROBE_CAP1));
    hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(256), lds, s, static_cast<const float4*>(in), static_cast<float4*>(out), n16, mode,
                       sym_in16, gap16, sym_out16, n_sym);
    return hipGetLastError();
}

#ifndef OFDM_TX_GRID_MULT
#define OFDM_TX_GRID_MULT 16
#endif
template <int N, int KIND>
static hipError_t launch_mod_nk(const TxDev& tx, const ModArgs& a, hipStream_t s) {
    const int64_t units = int64_t(a.n_frames) * a.n_sym;
    const int64_t wgs = (units + Plan<N>::SLOTS - 1) / Plan<N>::SLOTS;
    if (wgs == 0) return hipSuccess;
    const size_t lds_b = WgLds<N>::BYTES + TX_LUT_ELEMS * sizeof(cf);      // + the constellation table
    // Exactly the workgroups that are resident at once (registers and LDS decide: asked of the runtime, once per kernel), the
    // rest is looped: a grid larger than that runs in waves of workgroups, and the last, partly filled wave of a looping
    // kernel costs a whole loop's time on a fraction of the chip (2 048 workgroups with 1 536 resident: +50 %).
    static int per_cu = 0, n_cu = 0;
    if (per_cu == 0) {
        int nb = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, tx_modulate_kernel<N, KIND>, Plan<N>::WG, lds_b) != hipSuccess || nb < 1) nb = 4;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1)
            n_cu = 256;
        else
            n_cu = prop.multiProcessorCount;
        per_cu = nb;
    }
    // ... unless the launch is large enough for MANY such waves of workgroups: then the tail is a sixteenth of the work and the
    // dispatcher's refilling of freed slots beats the fixed assignment (+2-8 % at the bench's batch sizes; a workgroup still walks
    // >= 16 symbols, so its tables stay amortised; at 512 frames the exactly-resident grid is as good or better and is kept)
    const int64_t resident = int64_t(n_cu) * per_cu;
    int64_t g0 = std::min<int64_t>(wgs, wgs >= resident * OFDM_TX_GRID_MULT * 16 ? resident * OFDM_TX_GRID_MULT : resident);
    // a workgroup walks units first, first + stride, ...: with a stride that is a multiple of the [S, D] pattern length it would
    // meet the same position of the pattern every time (a quarter of the workgroups only copying sync symbols): keep them coprime
    if (g0 < wgs) {
        const int64_t SD = tx.S + tx.D;
        auto gcd = [](int64_t x, int64_t y) { while (y) { const int64_t r_ = x % y; x = y; y = r_; } return x; };
        while (g0 > 1 && gcd(g0 * Plan<N>::SLOTS, SD) != 1) --g0;
    }
    hipLaunchKernelGGL((tx_modulate_kernel<N, KIND>), dim3(unsigned(g0)), dim3(Plan<N>::WG), lds_b, s, tx, a);
    return hipGetLastError();
}

template <int N>
static hipError_t launch_mod_n(const TxDev& tx, const ModArgs& a, hipStream_t s) {
    const bool al4 = a.bits && ((reinterpret_cast<uintptr_t>(a.bits) | uintptr_t(a.bits_stride)) & 3) == 0;
    switch (tx_fetch_kind(a.bits_mode, tx.bps, al4)) {
        case 2: return launch_mod_nk<N, 2>(tx, a, s);
        case 4: return launch_mod_nk<N, 4>(tx, a, s);
        case 6: return launch_mod_nk<N, 6>(tx, a, s);
        case 10: return launch_mod_nk<N, 10>(tx, a, s);
        case 12: return launch_mod_nk<N, 12>(tx, a, s);
        case 14: return launch_mod_nk<N, 14>(tx, a, s);
        default: return launch_mod_nk<N, 0>(tx, a, s);
    }
}

hipError_t launch_tx_modulate(const TxDev& tx, const ModArgs& a, hipStream_t s) {
    switch (tx.nfft) {
        case 64: return launch_mod_n<64>(tx, a, s);
        case 128: return launch_mod_n<128>(tx, a, s);
        case 256: return launch_mod_n<256>(tx, a, s);
        case 512: return launch_mod_n<512>(tx, a, s);
        case 1024: return launch_mod_n<1024>(tx, a, s);
        case 2048: return launch_mod_n<2048>(tx, a, s);
        case 4096: return launch_mod_n<4096>(tx, a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_tx_random_bits(uint64_t seed, uint64_t offset, uint8_t* out, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int64_t nblk = (n + 127) / 128 + 1;
    hipLaunchKernelGGL(tx_random_bits_kernel, dim3(unsigned(std::min<int64_t>((nblk + 255) / 256, 262144))), dim3(256), 0, s, seed, offset, out, n);
    return hipGetLastError();
}

hipError_t launch_tx_map(const uint8_t* bits, int bits_mode, int bps, int64_t n_sym, cf* out, hipStream_t s) {
    if (n_sym <= 0) return hipSuccess;
    hipLaunchKernelGGL(tx_map_kernel, dim3(unsigned(std::min<int64_t>((n_sym + 255) / 256, 262144))), dim3(256), 0, s, bits, bits_mode, bps, n_sym, out);
    return hipGetLastError();
}

hipError_t launch_tx_grid(const TxDev& tx, const GridArgs& a, hipStream_t s) {
    if (a.n_rows <= 0) return hipSuccess;
    const int64_t total = a.n_rows * tx.nfft;
    hipLaunchKernelGGL(tx_grid_kernel, dim3(unsigned(std::min<int64_t>((total + 255) / 256, 262144))), dim3(256), 0, s, tx, a);
    return hipGetLastError();
}

hipError_t launch_tx_sync_grid(const TxDev& tx, cf* grid, hipStream_t s) {
    hipLaunchKernelGGL(tx_sync_grid_kernel, dim3(unsigned((tx.S * tx.nfft + 255) / 256)), dim3(256), 0, s, tx, grid);
    return hipGetLastError();
}

template <int N>
static hipError_t launch_time_n(const TxDev& tx, const TimeArgs& a, hipStream_t s) {
    const unsigned grid = unsigned((a.n_rows + Plan<N>::SLOTS - 1) / Plan<N>::SLOTS);
    if (grid == 0) return hipSuccess;
    if (a.do_ifft && a.do_cp)
        hipLaunchKernelGGL((tx_time_kernel<N, true, true>), dim3(grid), dim3(Plan<N>::WG), WgLds<N>::BYTES, s, tx, a);
    else if (a.do_ifft)
        hipLaunchKernelGGL((tx_time_kernel<N, true, false>), dim3(grid), dim3(Plan<N>::WG), WgLds<N>::BYTES, s, tx, a);
    else if (a.do_cp)
        hipLaunchKernelGGL((tx_time_kernel<N, false, true>), dim3(grid), dim3(Plan<N>::WG), WgLds<N>::BYTES, s, tx, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_tx_time(const TxDev& tx, const TimeArgs& a, hipStream_t s) {
    switch (tx.nfft) {
        case 64: return launch_time_n<64>(tx, a, s);
        case 128: return launch_time_n<128>(tx, a, s);
        case 256: return launch_time_n<256>(tx, a, s);
        case 512: return launch_time_n<512>(tx, a, s);
        case 1024: return launch_time_n<1024>(tx, a, s);
        case 2048: return launch_time_n<2048>(tx, a, s);
        case 4096: return launch_time_n<4096>(tx, a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_tx_mux(const TxDev& tx, const MuxArgs& a, hipStream_t s) {
    if (a.n_out_sym <= 0) return hipSuccess;
    const unsigned gy = unsigned(std::min<int64_t>(a.n_out_sym, 65535));
    hipLaunchKernelGGL(tx_mux_kernel, dim3(unsigned((tx.L + 255) / 256), gy), dim3(256), 0, s, tx, a);
    return hipGetLastError();
}

hipError_t launch_channel(const ChanArgs& a, hipStream_t s) {
    if (a.n_frames <= 0 || a.out_len <= 0) return hipSuccess;
    // (a thread takes a PAIR of samples: the grid is sized in pairs; one pair per thread -- a looping grid was measured 12 % slower)
    const unsigned gx = unsigned(std::min<int64_t>(((a.out_len + 1) / 2 + 255) / 256, 4096));
    hipLaunchKernelGGL(channel_kernel, dim3(gx, unsigned(a.n_frames)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ofdm
