// fft_core.hpp -- register/LDS FFT building blocks for gfx950 (MI355X), N = 64 .. 4096.
//
// One OFDM symbol is transformed by T = N/P cooperating lanes, each holding P complex points in
// VGPRs (P = 16, or 8 for N = 64).  N is factored R0 x R1 [x r]; every pass is a register-resident
// radix-R DIF butterfly, passes are separated by an LDS exchange whose layouts are padded so that
// both the ds_write_b64 side (16-lane groups, 32 banks) and the ds_read_b64 side (32-lane groups,
// 64 banks) are conflict-free (MI355X_MICROARCH.md, LDS table):
//
//   3-pass  N = 16*16*r, T = 16r          (512, 1024, 2048, 4096)
//     pass0  lane t=(n1,n2)   : x[t + T*n0]            -> LDS A[k0*(T+2) + t]           * W_N^(k0*t)
//     pass1  lane t'=n2*16+k0 : A[k0*(T+2)+n1*r+n2]    -> LDS B[(k1*16+k0)*(r+1) + n2]  * W_N^(16*k1*n2)
//     pass2  lane t''         : B[c*(r+1)+n2], c=t''+T*j -> X[k = c + 256*k2]
//   2-pass  N = R0*r, T = r                (64, 128, 256)
//     pass0  lane t=n1        : x[t + T*n0]            -> LDS B[k0*(r+1) + t]           * W_N^(k0*t)
//     pass1  lane t''         : B[c*(r+1)+n1], c=t''+T*j -> X[k = c + R0*k1]
//
// Final bins land in registers as  X[c + NC*kl]  with c = lane + T*j  (NC = N/r): consecutive
// lanes hold consecutive bins, which is what the coalesced bin de-map needs.
//
// Everything here is __host__ __device__ so tests/host/fft_core_host_test.cpp can run the exact
// index logic lane-by-lane on the CPU (hipcc host compilation; no GPU, no shim macros).
#pragma once
#include <hip/hip_runtime.h>

#define OFDM_HD __host__ __device__ __forceinline__
#ifndef OFDM_B_SWZ_RL8
#define OFDM_B_SWZ_RL8 0
#endif
#ifndef OFDM_W1_COMPACT_RL8
#define OFDM_W1_COMPACT_RL8 0      // 2048-pt geometry study: the 4-entry pass-1 twiddle rows of the 4096-pt plan (256 B instead of 1 KB)
#endif

namespace ofdm {

// A complex sample is a native 2-vector: + - * map 1:1 onto v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 and the
// .yx / .xx / .yy swizzles and sign flips fold into those instructions' op_sel / neg modifiers.  (Built with
// -fno-slp-vectorize: left to itself hipcc's SLP pass pairs unrelated scalars and pays ~340 v_mov per symbol.)
typedef float cf __attribute__((ext_vector_type(2)));

OFDM_HD cf pk_fma(cf a, cf b, cf c) { return __builtin_elementwise_fma(a, b, c); }
// a*b = a*(b.x,b.x) + (a.y,a.x)*(-b.y,b.y)
OFDM_HD cf cmul(cf a, cf b) { return pk_fma(a.yx, cf{-b.y, b.y}, a * b.xx); }
// a*conj(b) = a*(b.x,b.x) + (a.y,a.x)*(b.y,-b.y)
OFDM_HD cf cmulc(cf a, cf b) { return pk_fma(a.yx, cf{b.y, -b.y}, a * b.xx); }
OFDM_HD cf cconj(cf a) { return cf{a.x, -a.y}; }
OFDM_HD cf cscale(cf a, float s) { return a * s; }
OFDM_HD float cnorm2(cf a) { return a.x * a.x + a.y * a.y; }

// In-place a_i *= b_i for three independent pairs.  hipcc cannot fold the per-lane sign of a complex product into the
// packed instruction's neg_lo/neg_hi bits (it materialises (-b.y, b.y) with v_mov + v_xor: 2 extra VALU per
// product, ~13 % of this path's VALU time), so on the device the two instructions are written out:
//     t = (a.y*b.y, a.x*b.y)               v_pk_mul_f32  op_sel:[1,1] op_sel_hi:[0,1]
//     a = (a.x*b.x - t.x, a.y*b.x + t.y)   v_pk_fma_f32  op_sel_hi:[1,0,1] neg_lo:[0,0,1]
// Plain VALU, register operands only: no memory counters or hazard padding involved (cdna_hip_programming.md 5.7).
OFDM_HD void cmul3(cf& a0, cf b0, cf& a1, cf b1, cf& a2, cf b2) {
#if defined(__HIP_DEVICE_COMPILE__)
    cf t0, t1, t2;
    asm("v_pk_mul_f32 %3, %0, %6 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %4, %1, %7 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %5, %2, %8 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %0, %6, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1]\n\t"
        "v_pk_fma_f32 %1, %1, %7, %4 op_sel_hi:[1,0,1] neg_lo:[0,0,1]\n\t"
        "v_pk_fma_f32 %2, %2, %8, %5 op_sel_hi:[1,0,1] neg_lo:[0,0,1]"
        : "+v"(a0), "+v"(a1), "+v"(a2), "=&v"(t0), "=&v"(t1), "=&v"(t2)
        : "v"(b0), "v"(b1), "v"(b2));
#else
    a0 = cmul(a0, b0);
    a1 = cmul(a1, b1);
    a2 = cmul(a2, b2);
#endif
}

// the same for two pairs
OFDM_HD void cmul2(cf& a0, cf b0, cf& a1, cf b1) {
#if defined(__HIP_DEVICE_COMPILE__)
    cf t0, t1;
    asm("v_pk_mul_f32 %2, %0, %4 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %1, %5 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %0, %4, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1]\n\t"
        "v_pk_fma_f32 %1, %1, %5, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1]"
        : "+v"(a0), "+v"(a1), "=&v"(t0), "=&v"(t1)
        : "v"(b0), "v"(b1));
#else
    a0 = cmul(a0, b0);
    a1 = cmul(a1, b1);
#endif
}

// cos/sin(2*pi*j/16), j = 0..7
constexpr float kCos16[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                             0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
constexpr float kSin16[8] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                             1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};

// d * exp(-2*pi*i*J/R), J in [0, R/2), R in {2,4,8,16}
template <int R, int J>
OFDM_HD cf mul_w(cf d) {
    static_assert(R <= 16 && J < R / 2 + (R == 1), "radix");
    if constexpr (J == 0) {
        return d;
    } else if constexpr (4 * J == R) {
        return d.yx * cf{1.f, -1.f};                            // * -j = (y, -x)
    } else if constexpr (8 * J == R) {
        constexpr float c = 0.70710678118654752f;               // * (1-j)/sqrt2 = (x+y, y-x)*c
        return pk_fma(d.yx, cf{c, -c}, d * c);
    } else if constexpr (8 * J == 3 * R) {
        constexpr float c = 0.70710678118654752f;               // * (-1-j)/sqrt2 = (y-x, -x-y)*c
        return pk_fma(d.yx, cf{c, -c}, d * (-c));
    } else {
        constexpr float c = kCos16[J * (16 / R)];               // * (c - j s) = d*c + (y,-x)*s
        constexpr float s = kSin16[J * (16 / R)];
        return pk_fma(d.yx, cf{s, -s}, d * c);
    }
}

constexpr int bitrev(int k, int R) {
    int r = 0;
    for (int b = 1; b < R; b <<= 1) {
        r = (r << 1) | (k & 1);
        k >>= 1;
    }
    return r;
}

template <int R, int B, int S, int P, int J>
OFDM_HD void dif_stage(cf (&v)[P]) {
    if constexpr (J < R / 2) {
        const cf a = v[B + S * J];
        const cf b = v[B + S * (J + R / 2)];
        v[B + S * J] = a + b;
        v[B + S * (J + R / 2)] = mul_w<R, J>(a - b);
        dif_stage<R, B, S, P, J + 1>(v);
    }
}

// In-place radix-2 DIF DFT (forward, e^{-i..}) of the R points v[B + S*j]; X[k] ends at v[B + S*bitrev(k,R)].
template <int R, int B, int S, int P>
OFDM_HD void dft_dif(cf (&v)[P]) {
    if constexpr (R > 1) {
        dif_stage<R, B, S, P, 0>(v);
        dft_dif<R / 2, B, S, P>(v);
        dft_dif<R / 2, B + S * (R / 2), S, P>(v);
    }
}

// ------------------------------------------------------------------------------------------ plans
template <int N>
struct Plan {
    static_assert(N == 64 || N == 128 || N == 256 || N == 512 || N == 1024 || N == 2048 || N == 4096,
                  "supported FFT sizes: 64..4096");
    static constexpr int P = (N == 64) ? 8 : 16;          // points per lane
    static constexpr int T = N / P;                        // lanes per symbol
    static constexpr int R0 = P;                           // first radix
    static constexpr bool THREE = (N >= 512);              // 16 x 16 x r
    static constexpr int RL = THREE ? N / 256 : N / R0;    // last radix r
    static constexpr int C = P / RL;                       // last-pass DFTs per lane
    static constexpr int NC = N / RL;                      // bins k = c + NC*kl
    static constexpr int LDS_A = THREE ? 16 * (T + 2) : 0;  // exchange A elements
    // Exchange B: rows of RL elements.  Padded to RL+1 -- or, where a third of a CU's LDS hangs on 1.8 KB (4096-pt: three
    // workgroups per CU instead of two), unpadded with the column ROTATED by the row: element (c, n) at c*RL + ((n + c + c/16) % RL).
    // Both forms are conflict-free for the ds_write_b64 of pass 1 (16-lane groups) and the ds_read_b64 of the last pass (32-lane
    // groups); tests/test_fft_core_host.py checks the bank rule on the index functions themselves.
    // The same rotation exists for rows of 8 (2048-pt): column (n + c/2 + c/16) % 8.  OFDM_B_SWZ_RL8 selects it (geometry
    // study of DESIGN.md 4.1: 4 symbol slots per workgroup, 8 symbols in flight per CU).
    static constexpr bool B_SWZ = THREE && (RL == 16 || (RL == 8 && OFDM_B_SWZ_RL8 != 0));
    static constexpr int LDS_B = B_SWZ ? NC * RL : NC * (RL + 1);   // exchange B elements
    // pass-1 twiddle table: all 15 powers per n2 (16*RL entries), or only W^1, W^2, W^4, W^8 (4*RL entries, 4096-pt: LDS again)
    static constexpr bool W1_COMPACT = THREE && (RL == 16 || (RL == 8 && OFDM_W1_COMPACT_RL8 != 0));
    static constexpr int W1_ELEMS = THREE ? (W1_COMPACT ? 4 * RL : 16 * RL) : 0;
    static constexpr int LDS_ELEMS = (LDS_A > LDS_B ? LDS_A : LDS_B) > N ? (LDS_A > LDS_B ? LDS_A : LDS_B) : N;
    static constexpr int SLOTS = (T >= 64) ? 1 : 64 / T;   // symbols handled side by side in one workgroup
    static constexpr int WG = T * SLOTS;                   // workgroup size (>= 64)
};

// twiddle table: tw[j] = exp(-2*pi*i*j/N), j in [0,N)
// Pass-0 twiddles W_N^(k0*t) are lane-specific and stay in VGPRs for the lane's lifetime.
// Pass-1 twiddles W_N^(16*k1*n2) depend only on (n2 = lane>>4, k1): they live in a small shared
// table w1tab[n2*16 + k1] (r*16 entries, LDS on the device) instead of 30 more VGPRs per lane.
template <int N>
struct LaneTwiddles {
    cf w0[Plan<N>::R0];   // indexed by k0 (entry 0 unused: it is 1)
};

template <int N>
OFDM_HD void load_twiddles(LaneTwiddles<N>& tw, const cf* __restrict__ table, int t) {
    using PL = Plan<N>;
#pragma unroll
    for (int k0 = 0; k0 < PL::R0; ++k0) tw.w0[k0] = table[(k0 * t) & (N - 1)];
}

// Compact form of the pass-0 twiddles (R0 == 16): only W^t, W^2t, W^4t, W^8t are held (8 VGPRs instead of 30); the other
// eleven are products formed right where they are used (11 extra complex products per symbol, at most two temporaries live).
template <int N>
struct CompactTwiddles {
    cf w1, w2, w4, w8;
};

template <int N>
OFDM_HD void load_twiddles(CompactTwiddles<N>& tw, const cf* __restrict__ table, int t) {
    tw.w1 = table[t & (N - 1)];
    tw.w2 = table[(2 * t) & (N - 1)];
    tw.w4 = table[(4 * t) & (N - 1)];
    tw.w8 = table[(8 * t) & (N - 1)];
}

// exchange B index of element (row c, column n)
template <int N>
OFDM_HD int b_index(int c, int n) {
    using PL = Plan<N>;
    if constexpr (PL::B_SWZ && PL::RL == 16)
        return c * 16 + ((n + c + (c >> 4)) & 15);
    else if constexpr (PL::B_SWZ)                               // RL == 8
        return c * 8 + ((n + (c >> 1) + (c >> 4)) & 7);
    else
        return c * (PL::RL + 1) + n;
}

// pass-1 table W_N^(16*k1*n2): entry e = n2*16 + k1 (3-pass plans: e < 16*RL), or compact e = n2*4 + log2(k1), k1 in {1,2,4,8}
template <int N>
OFDM_HD cf w1_entry(const cf* __restrict__ table, int e) {
    if constexpr (Plan<N>::W1_COMPACT)
        return table[(16 * (1 << (e & 3)) * (e >> 2)) & (N - 1)];
    else
        return table[(16 * (e & 15) * (e >> 4)) & (N - 1)];
}

// v[bitrev(k,16)] *= W^k for k = 1..15 with only W^1, W^2, W^4, W^8 given: the other eleven are products formed right where
// they are used (at most two temporaries live).
template <int N>
OFDM_HD void apply_compact_twiddles(cf (&v)[Plan<N>::P], const CompactTwiddles<N>& twc) {
    constexpr int R = 16;
    CompactTwiddles<N> tw = twc;
#if defined(__HIP_DEVICE_COMPILE__)
    // opaque copies: without this the products below are loop-invariant and hipcc hoists all eleven back into VGPRs
    asm volatile("" : "+v"(tw.w1), "+v"(tw.w2), "+v"(tw.w4), "+v"(tw.w8));
#endif
#define OFDM_V(k) v[bitrev(k, R)]
    cmul3(OFDM_V(1), tw.w1, OFDM_V(2), tw.w2, OFDM_V(4), tw.w4);
    OFDM_V(8) = cmul(OFDM_V(8), tw.w8);
    {
        cf a = cmul(tw.w1, tw.w2);                 // W^3
        OFDM_V(3) = cmul(OFDM_V(3), a);
        cf b = cmul(a, tw.w8);                     // W^11
        OFDM_V(11) = cmul(OFDM_V(11), b);
        a = cmul(a, tw.w4);                        // W^7
        OFDM_V(7) = cmul(OFDM_V(7), a);
        a = cmul(a, tw.w8);                        // W^15
        OFDM_V(15) = cmul(OFDM_V(15), a);
        a = cmul(tw.w1, tw.w4);                    // W^5
        OFDM_V(5) = cmul(OFDM_V(5), a);
        a = cmul(a, tw.w8);                        // W^13
        OFDM_V(13) = cmul(OFDM_V(13), a);
        a = cmul(tw.w2, tw.w4);                    // W^6
        OFDM_V(6) = cmul(OFDM_V(6), a);
        a = cmul(a, tw.w8);                        // W^14
        OFDM_V(14) = cmul(OFDM_V(14), a);
        a = cmul(tw.w1, tw.w8);                    // W^9
        OFDM_V(9) = cmul(OFDM_V(9), a);
        a = cmul(tw.w2, tw.w8);                    // W^10
        OFDM_V(10) = cmul(OFDM_V(10), a);
        a = cmul(tw.w4, tw.w8);                    // W^12
        OFDM_V(12) = cmul(OFDM_V(12), a);
    }
#undef OFDM_V
}

// ------------------------------------------------------------------------------------------ passes
// pass 0 with compact twiddles (same result up to the rounding of the twiddle products, ~2e-7 relative)
template <int N>
OFDM_HD void fft_pass0_store(cf (&v)[Plan<N>::P], cf* lds, const CompactTwiddles<N>& twc, int t) {
    using PL = Plan<N>;
    static_assert(PL::R0 == 16, "compact twiddles are for the radix-16 first pass");
    dft_dif<16, 0, 1, PL::P>(v);
    constexpr int row = PL::THREE ? (PL::T + 2) : (PL::RL + 1);
    constexpr int R = 16;
    apply_compact_twiddles<N>(v, twc);
#pragma unroll
    for (int k0 = 0; k0 < R; ++k0) lds[k0 * row + t] = v[bitrev(k0, R)];
}

// Each function is the work of ONE lane t (0..T-1) of one symbol; `lds` is that symbol's exchange
// region (Plan<N>::LDS_ELEMS elements).  The caller puts a workgroup barrier between consecutive calls.

// pass 0: v[n0] = x[t + T*n0] on entry.
template <int N>
OFDM_HD void fft_pass0_store(cf (&v)[Plan<N>::P], cf* lds, const LaneTwiddles<N>& tw, int t) {
    using PL = Plan<N>;
    dft_dif<PL::R0, 0, 1, PL::P>(v);
    constexpr int row = PL::THREE ? (PL::T + 2) : (PL::RL + 1);
    constexpr int R = PL::R0;
    if constexpr (R == 16) {
#pragma unroll
        for (int k0 = 1; k0 < 16; k0 += 3)
            cmul3(v[bitrev(k0, R)], tw.w0[k0], v[bitrev(k0 + 1, R)], tw.w0[k0 + 1], v[bitrev(k0 + 2, R)], tw.w0[k0 + 2]);
    } else {   // R == 8: k0 = 1..7 -> 3 + 3 + 1
        cmul3(v[bitrev(1, R)], tw.w0[1], v[bitrev(2, R)], tw.w0[2], v[bitrev(3, R)], tw.w0[3]);
        cmul3(v[bitrev(4, R)], tw.w0[4], v[bitrev(5, R)], tw.w0[5], v[bitrev(6, R)], tw.w0[6]);
        v[bitrev(7, R)] = cmul(v[bitrev(7, R)], tw.w0[7]);
    }
#pragma unroll
    for (int k0 = 0; k0 < R; ++k0) lds[k0 * row + t] = v[bitrev(k0, R)];
}

// pass 1 (3-pass plans): lane t' = n2*16 + k0
template <int N>
OFDM_HD void fft_pass1_load(cf (&v)[Plan<N>::P], const cf* lds, int t) {
    using PL = Plan<N>;
    const int n2 = t >> 4, k0 = t & 15;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = lds[k0 * (PL::T + 2) + n1 * PL::RL + n2];
}

// The rotated layouts cost one add + and per element address.  Those addresses do not depend on the symbol, so hipcc hoists all
// 16 of them out of the caller's symbol loop and holds (or spills) 16 VGPRs for the kernel's lifetime; an opaque copy of the
// lane index keeps the arithmetic where it is used (32 VALU per symbol and pass instead).
OFDM_HD int opaque_lane(int t) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(t));
#endif
    return t;
}

template <int N>
OFDM_HD void fft_pass1_store(cf (&v)[Plan<N>::P], cf* lds, const cf* w1tab, int t) {
    using PL = Plan<N>;
    if constexpr (PL::B_SWZ) t = opaque_lane(t);
    const int n2 = t >> 4, k0 = t & 15;
    dft_dif<16, 0, 1, PL::P>(v);
    if constexpr (PL::W1_COMPACT) {
        CompactTwiddles<N> tw;
        tw.w1 = w1tab[n2 * 4 + 0];
        tw.w2 = w1tab[n2 * 4 + 1];
        tw.w4 = w1tab[n2 * 4 + 2];
        tw.w8 = w1tab[n2 * 4 + 3];
        apply_compact_twiddles<N>(v, tw);
    } else {
#pragma unroll
        for (int k1 = 1; k1 < 16; k1 += 3)
            cmul3(v[bitrev(k1, 16)], w1tab[n2 * 16 + k1], v[bitrev(k1 + 1, 16)], w1tab[n2 * 16 + k1 + 1], v[bitrev(k1 + 2, 16)],
                  w1tab[n2 * 16 + k1 + 2]);
    }
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) lds[b_index<N>(k1 * 16 + k0, n2)] = v[bitrev(k1, 16)];
}

// last pass: loads, transforms; afterwards bin k = (t + T*j) + NC*kl sits in v[j*RL + bitrev(kl,RL)].
template <int N>
OFDM_HD void fft_last_load(cf (&v)[Plan<N>::P], const cf* lds, int t) {
    using PL = Plan<N>;
    if constexpr (PL::B_SWZ) t = opaque_lane(t);
#pragma unroll
    for (int j = 0; j < PL::C; ++j) {
        const int c = t + PL::T * j;
#pragma unroll
        for (int n = 0; n < PL::RL; ++n) v[j * PL::RL + n] = lds[b_index<N>(c, n)];
    }
}

template <int N, int J = 0>
OFDM_HD void fft_last_dft(cf (&v)[Plan<N>::P]) {
    using PL = Plan<N>;
    if constexpr (J < PL::C) {
        dft_dif<PL::RL, J * PL::RL, 1, PL::P>(v);
        fft_last_dft<N, J + 1>(v);
    }
}

// register slot of bin (j, kl) after the last pass
template <int N>
constexpr int out_slot(int j, int kl) {
    return j * Plan<N>::RL + bitrev(kl, Plan<N>::RL);
}

}  // namespace ofdm
