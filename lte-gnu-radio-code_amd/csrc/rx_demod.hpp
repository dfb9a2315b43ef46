// rx_demod.hpp -- the hot kernel of the receive path (gfx950).
//
//   rx_demod_kernel<N, MOD, BMODE, MINW>
//     per data symbol: CP strip (by offset) + N-point FFT + data-bin de-map + per-symbol power normalisation
//     + lag de-rotation + one-tap MMSE equalise + (fused) hard de-map / bit packing
//     reference: gr-utsa_ofdm/python/SynchAndChanEst.py:221-248 ("Loop B"), BitRecovery.py:105-157
//
// Data flow per symbol (N = 2048: T = 128 lanes = 2 waves, 16 points per lane):
//   HBM --16 x 8 B/lane coalesced--> VGPR --radix16--> LDS A --radix16--> LDS B --radix8--> VGPR (bins, lane ~ bin)
//   --scatter into the reference's bin-list order (LDS), the power of the listed bins summed from the registers on the way-->
//   4 consecutive list entries per lane (16 B reads) --> x * sqrt(Kd/P) * gain[i] --> 16 B/lane stores (+ packed bits)
// The frame's Kd gains (equaliser * lag de-rotation, written by the sync kernel) are copied into LDS once per
// chunk of symbols.  A symbol is read from HBM once and written once: the kernel is HBM-bound by design; DESIGN.md 4.1 has
// the measurements of what does and does not move its time.
//
// MOD   bits per symbol of the fused de-mapper (1,2,4,6)           } compile-time: the per-element
// BMODE 0 = no bits, 1 = packed MSB-first, 2 = one bit per byte      } decision code has no runtime switches
// MINW  __launch_bounds__ min waves per SIMD (register budget; 3 -> 168 VGPRs)
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "ofdm_launch.hpp"

// geometry knobs of the 2048-pt kernel (defaults = the shipped geometry; the alternatives are built by `make geom`)
#ifndef OFDM_NS_T128
#define OFDM_NS_T128 2          // symbol slots per workgroup where a symbol is owned by 128 lanes
#endif
#ifndef OFDM_MINW_T128
#define OFDM_MINW_T128 3        // __launch_bounds__ waves per SIMD of that kernel (4 -> a 128-VGPR budget)
#endif
#ifndef OFDM_FLAGS_T128
#define OFDM_FLAGS_T128 0u      // DemodFlags of that kernel
#endif
#ifndef OFDM_FLAGS_T256
#define OFDM_FLAGS_T256 0u      // ... and of the 4096-pt kernel (a symbol owned by 256 lanes)
#endif
#ifndef OFDM_LANE_TW
#define OFDM_LANE_TW 1          // 1: all 15 pass-0 twiddles of a lane in VGPRs (30) instead of 4 base values + 11 products per symbol
#endif
#ifndef OFDM_SCATTER_HOIST
#define OFDM_SCATTER_HOIST 1    // 1: the 16 list offsets of a lane may be hoisted out of the symbol loop (16 VGPRs, ~50 VALU less)
#endif
#ifndef OFDM_SCATTER_SKIP
#define OFDM_SCATTER_SKIP 1     // 1: register slots without a listed bin are skipped by a scalar branch
#endif

namespace ofdm {

// BitRecovery's hard decision for float32 inputs (oracle/ofdm_oracle.py:demap_hard has the derivation):
//   QPSK axis bit = 1  iff  -sqrt2 <= x < 0  or  x > sqrt2   ==  (x < 0) xor (|x| > sqrt2_f32)
// valid whenever NEITHER coordinate of the symbol is exactly zero.
__device__ __forceinline__ unsigned qpsk_axis_bit(float x) {
    constexpr float t = 1.41421354f;   // largest float32 below sqrt(2)
    return unsigned(x < 0.f) ^ unsigned(fabsf(x) > t);
}

// A symbol ON an axis (or at the origin) is equidistant, in exact arithmetic, from two (four) constellation points.  The
// reference decides such ties by what its fp64 arithmetic happens to produce (BitRecovery.py:45-52,82-98,105-157), so this
// slow path repeats that arithmetic literally, in double:
//   CDAT  = exp(j 2pi/8 [1,-1,3,5]) as float64 (:45-52) -- not symmetric in the last bit: (bcd,bcc) (bcd,-bcc) (-bcc,bcd) (-bce,-bcc)
//   dist  = |z - CDAT_k| the way NumPy's AVX-512 complex-abs kernel forms it (the recorded reference run, tests/golden/
//           ref_bitrecovery.npz):  max * sqrt(fma(r, r, 1)), r = min / max   -- all correctly rounded IEEE operations
//   k*    = first arg-min (:87);  e = z - CDAT_k* (:93-98)
//   quadrant of z in the reference's order ++, -+, --, +- with >= / <= (first match wins, :106-125) picks which of the
//           metrics -f/2 |e| and -f/2 (K - |e|), K = 1.414213562373095 (:57), is llrp0 / llrp1
//   bit   = int(0.5 (sign(llrp1 - llrp0) + 1)) (:155-156)  ==  [ x(llrp1) < x(llrp0) ]  with x = |e| or K - |e|
//           (the common factor -f/2 is negative; the two x differ by 0, 2 or 4 ulp on a tie, so the products never collapse)
__device__ __noinline__ unsigned qpsk_bits_on_axis(float zx, float zy) {
    constexpr double A = 0x1.6a09e667f3bcdp-1, B = 0x1.6a09e667f3bccp-1, Cc = 0x1.6a09e667f3bcep-1;
    const double cx[4] = {A, A, -B, -Cc};
    const double cy[4] = {B, -B, A, -B};
    constexpr double K = 0x1.6a09e667f3bccp+0;             // the literal 1.414213562373095
    const double x = double(zx), y = double(zy);
    int kbest = 0;
    double dbest = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double ax = fabs(x - cx[k]), ay = fabs(y - cy[k]);
        const double mx = fmax(ax, ay), mn = fmin(ay, ax);
        const double r = mn / mx;                           // mx >= 0.29: no 0/0 here
        const double d = sqrt(fma(r, r, 1.0)) * mx;
        if (k == 0 || d < dbest) {
            dbest = d;
            kbest = k;
        }
    }
    const double ex = fabs(x - cx[kbest]), ey = fabs(y - cy[kbest]);
    const bool q1 = x >= 0.0 && y >= 0.0;
    const bool q2 = !q1 && x <= 0.0 && y >= 0.0;
    const bool q3 = !q1 && !q2 && x <= 0.0 && y <= 0.0;
    const bool q4 = !q1 && !q2 && !q3 && x >= 0.0 && y <= 0.0;
    const bool re_pos = q1 || q4, im_pos = q1 || q2;
    const double fr = K - ex, fi = K - ey;
    const unsigned b0 = re_pos ? (fr < ex) : (ex < fr);
    const unsigned b1 = im_pos ? (fi < ey) : (ey < fi);
    return (b0 << 1) | b1;
}

// bits of one symbol, b0 in the most significant of MOD bits
template <int MOD>
__device__ __forceinline__ unsigned hard_bits(cf z) {
    if constexpr (MOD == 2) {
        if (z.x == 0.f || z.y == 0.f) return qpsk_bits_on_axis(z.x, z.y);      // NaN compares false: closed form below
        return (qpsk_axis_bit(z.x) << 1) | qpsk_axis_bit(z.y);
    } else if constexpr (MOD == 1) {
        return z.x > 0.f;
    } else if constexpr (MOD == 4) {
        constexpr float t = 0.63245553203367588f;   // 2/sqrt(10)
        return (unsigned(z.x < 0.f) << 3) | (unsigned(z.y < 0.f) << 2) | (unsigned(fabsf(z.x) > t) << 1) |
               unsigned(fabsf(z.y) > t);
    } else {
        constexpr float a = 0.61721339984836765f;    // 4/sqrt(42)
        constexpr float c = 0.30860669992418382f;    // 2/sqrt(42)
        return (unsigned(z.x < 0.f) << 5) | (unsigned(z.y < 0.f) << 4) | (unsigned(fabsf(z.x) > a) << 3) |
               (unsigned(fabsf(z.y) > a) << 2) | (unsigned(fabsf(fabsf(z.x) - a) > c) << 1) |
               unsigned(fabsf(fabsf(z.y) - a) > c);
    }
}
__device__ __forceinline__ unsigned hard_bits_rt(cf z, int mod) {
    return mod == 2 ? hard_bits<2>(z) : mod == 1 ? hard_bits<1>(z) : mod == 4 ? hard_bits<4>(z) : hard_bits<6>(z);
}

// Hard bits of 4 consecutive symbols, MSB-first, as one integer (4*MOD bits): the packed de-mapper's inner loop.
// Each decision is a v_cmp into its own SGPR pair and one v_addc_co_u32 that shifts the bit into the word
// (w = 2w + carry): 2 VALU per bit instead of cmp + cndmask + shift + or, and no s_nop between a compare and its
// consumer (gfx950 needs wait states between a VALU writing an SGPR mask and a VALU reading it, so compares are
// issued four to six at a time).  Same float compares as hard_bits<MOD>: results are bit-identical.
template <int MOD, bool ASMB = true>
__device__ __forceinline__ unsigned pack4(const cf (&z)[4]) {
    unsigned w = 0;
    if constexpr (!ASMB) {
        w = (((((hard_bits<MOD>(z[0]) << MOD) | hard_bits<MOD>(z[1])) << MOD) | hard_bits<MOD>(z[2])) << MOD) | hard_bits<MOD>(z[3]);
    } else if constexpr (MOD == 4) {
        constexpr float t = 0.63245553203367588f;   // 2/sqrt(10)
        unsigned long long m0, m1, m2, m3;
#define OFDM_Q16(RE, IM)                                   \
    "v_cmp_gt_f32_e64 %1, 0, " RE "\n\t"                   \
    "v_cmp_gt_f32_e64 %2, 0, " IM "\n\t"                   \
    "v_cmp_gt_f32_e64 %3, |" RE "|, %13\n\t"               \
    "v_cmp_gt_f32_e64 %4, |" IM "|, %13\n\t"               \
    "v_addc_co_u32_e64 %0, %1, %0, %0, %1\n\t"             \
    "v_addc_co_u32_e64 %0, %2, %0, %0, %2\n\t"             \
    "v_addc_co_u32_e64 %0, %3, %0, %0, %3\n\t"             \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\t"
        asm(OFDM_Q16("%5", "%6") OFDM_Q16("%7", "%8") OFDM_Q16("%9", "%10") OFDM_Q16("%11", "%12")
            : "+v"(w), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
            : "v"(z[0].x), "v"(z[0].y), "v"(z[1].x), "v"(z[1].y), "v"(z[2].x), "v"(z[2].y), "v"(z[3].x), "v"(z[3].y), "s"(t));
#undef OFDM_Q16
    } else if constexpr (MOD == 6) {
        constexpr float a = 0.61721339984836765f;    // 4/sqrt(42)
        constexpr float c = 0.30860669992418382f;    // 2/sqrt(42)
        unsigned long long m0, m1, m2, m3, m4, m5;
        float tr, ti;
#define OFDM_Q64(RE, IM)                                   \
    "v_sub_f32_e64 %7, |" RE "|, %17\n\t"                  \
    "v_sub_f32_e64 %8, |" IM "|, %17\n\t"                  \
    "v_cmp_gt_f32_e64 %1, 0, " RE "\n\t"                   \
    "v_cmp_gt_f32_e64 %2, 0, " IM "\n\t"                   \
    "v_cmp_gt_f32_e64 %3, |" RE "|, %17\n\t"               \
    "v_cmp_gt_f32_e64 %4, |" IM "|, %17\n\t"               \
    "v_cmp_gt_f32_e64 %5, |%7|, %18\n\t"                   \
    "v_cmp_gt_f32_e64 %6, |%8|, %18\n\t"                   \
    "v_addc_co_u32_e64 %0, %1, %0, %0, %1\n\t"             \
    "v_addc_co_u32_e64 %0, %2, %0, %0, %2\n\t"             \
    "v_addc_co_u32_e64 %0, %3, %0, %0, %3\n\t"             \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\t"             \
    "v_addc_co_u32_e64 %0, %5, %0, %0, %5\n\t"             \
    "v_addc_co_u32_e64 %0, %6, %0, %0, %6\n\t"
        asm(OFDM_Q64("%9", "%10") OFDM_Q64("%11", "%12") OFDM_Q64("%13", "%14") OFDM_Q64("%15", "%16")
            : "+v"(w), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&v"(tr), "=&v"(ti)
            : "v"(z[0].x), "v"(z[0].y), "v"(z[1].x), "v"(z[1].y), "v"(z[2].x), "v"(z[2].y), "v"(z[3].x), "v"(z[3].y), "s"(a), "s"(c));
#undef OFDM_Q64
    } else if constexpr (MOD == 2) {
        constexpr float t = 1.41421354f;             // largest float32 below sqrt(2): BitRecovery's outlier edge
        // a coordinate that is exactly zero (a tie of the reference's nearest-point search) takes the literal path
        if (z[0].x * z[0].y * z[1].x * z[1].y == 0.f || z[2].x * z[2].y * z[3].x * z[3].y == 0.f) {
            bool tie = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) tie |= (z[e].x == 0.f) | (z[e].y == 0.f);
            if (tie)
                return (((((hard_bits<2>(z[0]) << 2) | hard_bits<2>(z[1])) << 2) | hard_bits<2>(z[2])) << 2) | hard_bits<2>(z[3]);
        }
        unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
        // two symbols per group: bit = (x < 0) xor (|x| > sqrt2)
#define OFDM_QPSK2(RE0, IM0, RE1, IM1)                     \
    "v_cmp_gt_f32_e64 %1, 0, " RE0 "\n\t"                  \
    "v_cmp_gt_f32_e64 %2, |" RE0 "|, %17\n\t"              \
    "v_cmp_gt_f32_e64 %3, 0, " IM0 "\n\t"                  \
    "v_cmp_gt_f32_e64 %4, |" IM0 "|, %17\n\t"              \
    "v_cmp_gt_f32_e64 %5, 0, " RE1 "\n\t"                  \
    "v_cmp_gt_f32_e64 %6, |" RE1 "|, %17\n\t"              \
    "v_cmp_gt_f32_e64 %7, 0, " IM1 "\n\t"                  \
    "v_cmp_gt_f32_e64 %8, |" IM1 "|, %17\n\t"              \
    "s_xor_b64 %1, %1, %2\n\t"                             \
    "s_xor_b64 %3, %3, %4\n\t"                             \
    "s_xor_b64 %5, %5, %6\n\t"                             \
    "s_xor_b64 %7, %7, %8\n\t"                             \
    "s_nop 1\n\t"                                          \
    "v_addc_co_u32_e64 %0, %1, %0, %0, %1\n\t"             \
    "v_addc_co_u32_e64 %0, %3, %0, %0, %3\n\t"             \
    "v_addc_co_u32_e64 %0, %5, %0, %0, %5\n\t"             \
    "v_addc_co_u32_e64 %0, %7, %0, %0, %7\n\t"
        asm(OFDM_QPSK2("%9", "%10", "%11", "%12") OFDM_QPSK2("%13", "%14", "%15", "%16")
            : "+v"(w), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7)
            : "v"(z[0].x), "v"(z[0].y), "v"(z[1].x), "v"(z[1].y), "v"(z[2].x), "v"(z[2].y), "v"(z[3].x), "v"(z[3].y), "s"(t)
            : "scc");
#undef OFDM_QPSK2
    } else {
        w = (((((hard_bits<MOD>(z[0]) << MOD) | hard_bits<MOD>(z[1])) << MOD) | hard_bits<MOD>(z[2])) << MOD) | hard_bits<MOD>(z[3]);
    }
    return w;
}

// The same for 2 consecutive symbols (2*MOD bits): the dense output mapping hands a lane pairs of list entries.
template <int MOD, bool ASMB = true>
__device__ __forceinline__ unsigned pack2(const cf (&z)[2]) {
    unsigned w = 0;
    if constexpr (!ASMB || MOD == 1) {
        w = (hard_bits<MOD>(z[0]) << MOD) | hard_bits<MOD>(z[1]);
    } else if constexpr (MOD == 4) {
        constexpr float t = 0.63245553203367588f;   // 2/sqrt(10)
        unsigned long long m0, m1, m2, m3;
#define OFDM_Q16(RE, IM)                                   \
    "v_cmp_gt_f32_e64 %1, 0, " RE "\n\t"                   \
    "v_cmp_gt_f32_e64 %2, 0, " IM "\n\t"                   \
    "v_cmp_gt_f32_e64 %3, |" RE "|, %9\n\t"                \
    "v_cmp_gt_f32_e64 %4, |" IM "|, %9\n\t"                \
    "v_addc_co_u32_e64 %0, %1, %0, %0, %1\n\t"             \
    "v_addc_co_u32_e64 %0, %2, %0, %0, %2\n\t"             \
    "v_addc_co_u32_e64 %0, %3, %0, %0, %3\n\t"             \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\t"
        asm(OFDM_Q16("%5", "%6") OFDM_Q16("%7", "%8")
            : "+v"(w), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
            : "v"(z[0].x), "v"(z[0].y), "v"(z[1].x), "v"(z[1].y), "s"(t));
#undef OFDM_Q16
    } else if constexpr (MOD == 6) {
        constexpr float a = 0.61721339984836765f;    // 4/sqrt(42)
        constexpr float c = 0.30860669992418382f;    // 2/sqrt(42)
        unsigned long long m0, m1, m2, m3, m4, m5;
        float tr, ti;
#define OFDM_Q64(RE, IM)                                   \
    "v_sub_f32_e64 %7, |" RE "|, %13\n\t"                  \
    "v_sub_f32_e64 %8, |" IM "|, %13\n\t"                  \
    "v_cmp_gt_f32_e64 %1, 0, " RE "\n\t"                   \
    "v_cmp_gt_f32_e64 %2, 0, " IM "\n\t"                   \
    "v_cmp_gt_f32_e64 %3, |" RE "|, %13\n\t"               \
    "v_cmp_gt_f32_e64 %4, |" IM "|, %13\n\t"               \
    "v_cmp_gt_f32_e64 %5, |%7|, %14\n\t"                   \
    "v_cmp_gt_f32_e64 %6, |%8|, %14\n\t"                   \
    "v_addc_co_u32_e64 %0, %1, %0, %0, %1\n\t"             \
    "v_addc_co_u32_e64 %0, %2, %0, %0, %2\n\t"             \
    "v_addc_co_u32_e64 %0, %3, %0, %0, %3\n\t"             \
    "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\t"             \
    "v_addc_co_u32_e64 %0, %5, %0, %0, %5\n\t"             \
    "v_addc_co_u32_e64 %0, %6, %0, %0, %6\n\t"
        asm(OFDM_Q64("%9", "%10") OFDM_Q64("%11", "%12")
            : "+v"(w), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&v"(tr), "=&v"(ti)
            : "v"(z[0].x), "v"(z[0].y), "v"(z[1].x), "v"(z[1].y), "s"(a), "s"(c));
#undef OFDM_Q64
    } else {                                         // MOD == 2
        constexpr float t = 1.41421354f;             // largest float32 below sqrt(2): BitRecovery's outlier edge
        // a coordinate that is exactly zero (a tie of the reference's nearest-point search) takes the literal path
        if (z[0].x * z[0].y * z[1].x * z[1].y == 0.f) {
            const bool tie = (z[0].x == 0.f) | (z[0].y == 0.f) | (z[1].x == 0.f) | (z[1].y == 0.f);
            if (tie) return (hard_bits<2>(z[0]) << 2) | hard_bits<2>(z[1]);
        }
        unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
        asm("v_cmp_gt_f32_e64 %1, 0, %9\n\t"
            "v_cmp_gt_f32_e64 %2, |%9|, %13\n\t"
            "v_cmp_gt_f32_e64 %3, 0, %10\n\t"
            "v_cmp_gt_f32_e64 %4, |%10|, %13\n\t"
            "v_cmp_gt_f32_e64 %5, 0, %11\n\t"
            "v_cmp_gt_f32_e64 %6, |%11|, %13\n\t"
            "v_cmp_gt_f32_e64 %7, 0, %12\n\t"
            "v_cmp_gt_f32_e64 %8, |%12|, %13\n\t"
            "s_xor_b64 %1, %1, %2\n\t"
            "s_xor_b64 %3, %3, %4\n\t"
            "s_xor_b64 %5, %5, %6\n\t"
            "s_xor_b64 %7, %7, %8\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32_e64 %0, %1, %0, %0, %1\n\t"
            "v_addc_co_u32_e64 %0, %3, %0, %0, %3\n\t"
            "v_addc_co_u32_e64 %0, %5, %0, %0, %5\n\t"
            "v_addc_co_u32_e64 %0, %7, %0, %0, %7\n\t"
            : "+v"(w), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7)
            : "v"(z[0].x), "v"(z[0].y), "v"(z[1].x), "v"(z[1].y), "s"(t)
            : "scc");
    }
    return w;
}

// one-bit-per-byte output of the PAIR of list entries sym0, sym0 + 1 held by one lane (dense output mapping)
template <int MOD>
__device__ __forceinline__ void store_bits_pair_unpacked(uint8_t* row, unsigned sym0, const cf (&z)[2]) {
    uint8_t* o = row + sym0 * unsigned(MOD);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned hb = hard_bits<MOD>(z[e]);
#pragma unroll
        for (int b = 0; b < MOD; ++b) o[e * MOD + b] = uint8_t((hb >> (MOD - 1 - b)) & 1u);
    }
}

// the 4*MOD bits `w` of the list entries sym0 .. sym0 + 3 (sym0 % 4 == 0), MSB first, as MOD/2 bytes
// `row` = address of the row's first byte (the caller folds everything wave-uniform into it), sym0 = entry index within the row
template <int MOD>
__device__ __forceinline__ void store_packed4(uint8_t* row, unsigned sym0, unsigned w) {
    uint8_t* o = row + (sym0 >> 2) * unsigned(MOD / 2);
    if constexpr (MOD == 2) {
        o[0] = uint8_t(w);
    } else if constexpr (MOD == 4) {
        *reinterpret_cast<uint16_t*>(o) = uint16_t(((w & 0xffu) << 8) | (w >> 8));
    } else {
        o[0] = uint8_t(w >> 16);
        o[1] = uint8_t(w >> 8);
        o[2] = uint8_t(w);
    }
}

// writes the bits of 4 (or `cnt`) consecutive list entries starting at list index idx of output row `orow`
template <int MOD, int BMODE, bool ASMB = true>
__device__ __forceinline__ void store_bits(uint8_t* bits, int64_t sym0, const cf (&z)[4], int cnt) {
    if constexpr (BMODE == 1) {            // packed MSB-first: 4 symbols -> MOD/2 bytes (host guarantees Kd % 4 == 0, MOD even)
        store_packed4<MOD>(bits + (sym0 >> 2) * (MOD / 2), 0u, pack4<MOD, ASMB>(z));
    } else if constexpr (BMODE == 2) {     // one bit per byte
        uint8_t* o = bits + sym0 * MOD;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e < cnt) {
                const unsigned hb = hard_bits<MOD>(z[e]);
#pragma unroll
                for (int b = 0; b < MOD; ++b) o[e * MOD + b] = uint8_t((hb >> (MOD - 1 - b)) & 1u);
            }
        }
    }
}

// NS    symbol slots per workgroup.  All slots work on consecutive symbols of the SAME chunk (same frame), so they
//       share one LDS copy of the frame's gains and one pass-1 twiddle table: at N = 2048, NS = 2 brings the LDS
//       cost to 23.8 KB per symbol in flight -> 6 symbols per CU instead of 5 (occupancy is what bounds this kernel:
//       2/3/4/5 workgroups per CU measured 4.13/3.13/2.65/2.22 ms).
template <int N>
struct DemodGeom {
    static constexpr int T = Plan<N>::T;
    static constexpr int NS = (T >= 256) ? 1 : (T >= 128) ? OFDM_NS_T128 : (T >= 64 ? 2 : 64 / T);
    static constexpr int WG = T * NS;
    static constexpr int W1_OFF = WgLds<N>::STRIDE * NS;                    // cf units
    static constexpr int Q_OFF = W1_OFF + WgLds<N>::W1_ELEMS;               // 2 cf: the work-queue slot
    static constexpr int G_OFF = Q_OFF + 2;
    static size_t lds_bytes(int Kd, bool glds) { return (size_t(G_OFF) + (glds ? ((Kd + 3) & ~3) : 0)) * sizeof(cf); }
};

// Kernel build flags.  The product library instantiates FLAGS = 0 (batch and stream paths), DF_ROT (CFO receiver) and DF_HG
// (tracker receiver) only; every other bit exists for the A/B timing and stamping builds of the experiment library.
enum DemodFlags : unsigned {
    DF_ROT = 1u << 0,            // every window is multiplied by a carrier-offset rotator (SynchEstAndFO.py:339)
    DF_HG = 1u << 1,             // a frame is demodulated iff the host set its guard flag (tracker receiver)
    DF_GAINS_GLOBAL = 1u << 2,   // gains re-read from global memory per symbol instead of one LDS copy per chunk
    DF_TEMPORAL_LD = 1u << 3,    // plain (cached) stream loads instead of the non-temporal hint ...
    DF_TEMPORAL_ST = 1u << 11,   // ... and plain equalised-symbol stores (the pair of them is the round-1 / early round-2 kernel)
    DF_GENERIC_BITS = 1u << 4,   // plain C++ bit packing instead of the v_cmp / v_addc form
    DF_STAMP = 1u << 5,          // s_memtime stamps per phase (diagnostic, never timed)
    DF_NO_PIPE = 1u << 6,        // next symbol's loads issued at the loop top instead of right after the scatter
    DF_L2_INPUT = 1u << 7,       // every workgroup reads frame (blockIdx % 8): cache-resident input, WRONG results, timing only
    DF_LANE_TWIDDLES = 1u << 8,  // all 15 pass-0 twiddles of a lane in VGPRs instead of 4 base values + products
    DF_PSUM_READBACK = 1u << 9,  // power sum by a separate LDS read pass over the staged list (round-1 form)
    DF_GAINS_VGPR = 1u << 14,    // dense mapping only: a lane keeps the 16 gains of ITS list entries in 32 VGPRs for the whole chunk
                                 // (no LDS copy of the gains, no gain reads in the symbol loop)
    DF_FOUR_PER_LANE = 1u << 10, // output mapping of round 1 at every size: a lane owns 4 consecutive list entries (two 16 B
                                 // stores 32 B apart: every store instruction of a wave covers 2 KB half-filled)
};

template <int N, int MOD, int BMODE, int MINW, unsigned FLAGS = 0>
__global__ void __launch_bounds__(DemodGeom<N>::WG, MINW) rx_demod_kernel(RxDev rx, DemodArgs a) {
    constexpr bool ROT = (FLAGS & DF_ROT) != 0, HG = (FLAGS & DF_HG) != 0,
                   GREG = (FLAGS & DF_GAINS_VGPR) != 0 && Plan<N>::T >= 64 && (FLAGS & DF_FOUR_PER_LANE) == 0 && !(MOD == 1 && BMODE == 1),
                   GLDS = (FLAGS & DF_GAINS_GLOBAL) == 0 && !GREG,
                   // The IQ stream is read once and the equalised symbols are written once: both carry the non-temporal hint.
                   // It pays only with the dense output mapping (a store instruction covering whole lines): -2.1 % there, while
                   // on the half-filled 2 KB spans of the 4-entries-per-lane mapping it cost +7.6 %.  Below 1024-pt: plain.
                   NTL = (FLAGS & DF_TEMPORAL_LD) == 0 && Plan<N>::T >= 64, NT = (FLAGS & DF_TEMPORAL_ST) == 0 && Plan<N>::T >= 64, ASMB = (FLAGS & DF_GENERIC_BITS) == 0, STAMP = (FLAGS & DF_STAMP) != 0,
                   PIPE = (FLAGS & DF_NO_PIPE) == 0, L2IN = (FLAGS & DF_L2_INPUT) != 0,
                   CT = (FLAGS & DF_LANE_TWIDDLES) == 0 && !OFDM_LANE_TW && Plan<N>::R0 == 16, PSE = (FLAGS & DF_PSUM_READBACK) == 0,
                   // dense output mapping (N >= 1024): a lane owns two PAIRS of list entries 128 apart, so that each store
                   // instruction of a wave covers 1 KB contiguously and the LDS list is read at a 16 B lane stride
                   DENSE = Plan<N>::T >= 64 && (FLAGS & DF_FOUR_PER_LANE) == 0 && !(MOD == 1 && BMODE == 1);
    using PL = Plan<N>;
    using DG = DemodGeom<N>;
    constexpr int T = PL::T, P = PL::P, Q = P / 4, NS = DG::NS;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x;
    // a slot spans whole wavefronts when T >= 64: everything derived from it (symbol bookkeeping, row pointers) is then
    // wave-uniform, and saying so moves it from VGPRs / VALU to SGPRs / SALU
    const int slot = (T >= 64) ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T;
    const int t = tid % T;
    // One wave per symbol (T <= 64): the waves of a workgroup are independent symbols that only share the read-only gain copy
    // and twiddle table, so every per-symbol barrier is a wave-local fence (LDS is in order per wave).
    constexpr bool WL = T <= 64;
    cf* smem = reinterpret_cast<cf*>(smem_raw);
    cf* lds = smem + slot * WgLds<N>::STRIDE;
    float* red = reinterpret_cast<float*>(lds + WgLds<N>::ELEMS);
    cf* w1w = smem + DG::W1_OFF;
    if constexpr (PL::THREE) {
        for (int e = tid; e < WgLds<N>::W1_ELEMS; e += DG::WG) w1w[e] = w1_entry<N>(rx.tw, e);
    }
    const cf* w1tab = w1w;

    std::conditional_t<CT, CompactTwiddles<N>, LaneTwiddles<N>> tw;
    load_twiddles(tw, rx.tw, t);

    // STAMP (diagnostic build only, never timed): cycles per phase, summed over the chunk, one row per wave
    unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    auto stamp = [&](int i) {
        if constexpr (STAMP) {
            unsigned long long tn;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn)::"memory");
            acc[i] += unsigned(tn - tprev);
            tprev = tn;
        }
    };
    if constexpr (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
    // A chunk is a run of symbols of ONE frame.  Without a work queue (a.work == nullptr: stream blocks, small launches) workgroup
    // b owns chunk b.  With one, the launch has only as many workgroups as are resident at once and each takes chunk after chunk
    // from an atomic counter until the launch's chunks are used up: twiddles and the pass-1 table are set up once per workgroup
    // instead of once per chunk, and the CUs -- and the XCDs, whose share of a plain launch is fixed by the round-robin dispatch --
    // stay busy until the queue is empty instead of until their own last chunk is done.  The exit condition is reached by every
    // workgroup: the counter only grows.
    const int64_t n_chunks = int64_t(a.n_frames) * a.chunks_per_frame;
    int* const qslot = reinterpret_cast<int*>(smem + DG::Q_OFF);          // one LDS word: the chunk the workgroup works on
    int64_t chunk = blockIdx.x;
    if (a.work) {
        if (tid == 0) *qslot = int(atomicAdd(a.work, 1u));
        __syncthreads();
        chunk = *qslot;
    }
    for (; chunk < n_chunks;) {
    const int frame = int(chunk / a.chunks_per_frame);
    const int cidx = int(chunk % a.chunks_per_frame);
    // Balanced partition of the frame's data symbols into chunks_per_frame runs of whole trips (NS symbols): lengths differ by at
    // most one trip.  (launch_rx_demod_n explains why equal chunks matter: workgroups go to the 8 XCDs round-robin.)
    const int trips = (a.n_dsym + NS - 1) / NS;
    const int ds0 = int(int64_t(cidx) * trips / a.chunks_per_frame) * NS;
    const int ds1 = min(int(int64_t(cidx + 1) * trips / a.chunks_per_frame) * NS, a.n_dsym);
    constexpr bool active = true;

    const int Kd = rx.Kd, L = rx.L, S = rx.S, D = rx.D;
    const int tsr0 = a.tsr[frame * 4 + 0];
    const bool frame_on = HG ? (a.tsr[frame * 4 + 3] != 0) : true;     // host-applied guard (tracker receiver)
    // L2IN (experiment only, wrong results): every workgroup reads frame (blockIdx % 8) -> the input stays cache-resident
    const cf* frame_iq = a.iq + int64_t(L2IN ? (blockIdx.x & 7) : frame) * a.frame_stride;

    // the frame's gains -> LDS, once per chunk (Kd even: whole 16 B pairs); published by the FFT's first barrier
    const cf* gain = a.gain + int64_t(frame) * Kd;
    cf* glds = smem + DG::G_OFF;
    const cf* gsrc = GLDS ? glds : gain;
    if constexpr (GLDS) {
        for (int i = 2 * tid; i < Kd; i += 2 * DG::WG) *reinterpret_cast<float4*>(glds + i) = *reinterpret_cast<const float4*>(gain + i);
    }
    // GREG: the gains of the two pairs a lane owns per q (dense mapping: idx = 4 T q + 256 wave + 128 half + 2 lane)
    float4 greg[GREG ? Q : 1][2];
    if constexpr (GREG) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int hx = 0; hx < 2; ++hx) {
                const int idx = 4 * T * q + 256 * (t >> 6) + 128 * hx + 2 * (t & 63);
                greg[q][hx] = idx < Kd ? *reinterpret_cast<const float4*>(gain + idx) : float4{0.f, 0.f, 0.f, 0.f};
            }
    }

    const float sqrt_kd = sqrtf(float(Kd));
    const int n_iter = (ds1 - ds0 + NS - 1) / NS;                         // trips of THIS chunk
    // Per-symbol bookkeeping (uniform per slot).  SynchAndChanEst.py:222-223: data_ptr = tsr0 + S*L*(P+1), P = p*(S+D),
    // guarded once per pattern; :226: CP strip by offset.
    struct Sym {
        bool valid, compute;
        int64_t start, orow;
    };
    auto sym_of = [&](int it) {
        Sym sy;
        const int itr = it;
        const int ds = ds0 + itr * NS + slot;
        sy.valid = active && it < n_iter && ds < ds1;
        const int p = ds / D, n_ = ds - p * D;
        const int64_t pat_ptr = int64_t(tsr0) + int64_t(S) * L * (int64_t(p) * (S + D) + 1);
        sy.compute = sy.valid && (HG ? frame_on : (pat_ptr + N - 1 <= a.frame_len));
        sy.start = pat_ptr + int64_t(L) * n_;
        sy.orow = int64_t(frame) * a.rows_per_frame + (p * a.row_stride_pat + n_);
        return sy;
    };
    // PSE: the per-symbol power (:233) is summed from the FFT output REGISTERS while the bins are scattered into list order, and its
    // per-wave partials cross the scatter's own barrier -- no separate LDS read pass over the list and no barrier of its own
    // (PSE = false keeps the read-back form for A/B timing in the experiment build).
    cf v[P];
    auto load_sym = [&](const Sym& sy) {
        if (sy.compute && sy.start >= 0 && sy.start + N <= a.frame_len) {
            const cf* src = frame_iq + sy.start + t;
#pragma unroll
            for (int n0 = 0; n0 < P; ++n0) v[n0] = NTL ? __builtin_nontemporal_load(src + T * n0) : src[T * n0];
            if constexpr (ROT) {                                         // data_buff_time * cfo[idx]  (SynchEstAndFO.py:339)
#pragma unroll
                for (int n0 = 0; n0 < P; ++n0) v[n0] = cmul(v[n0], a.rot[t + T * n0]);
            }
        } else {                                                         // short tail: fft(x, N) zero-pads (:230)
            const int64_t last = a.frame_len > 0 ? a.frame_len - 1 : 0;
            const bool any = sy.compute && a.frame_len > 0;
            // Rare path.  Its 16 clamped element addresses are loop-invariant up to the symbol offset, and hipcc hoists what it can
            // out of the symbol loop: 26 VGPRs held (or spilled) for the whole kernel.  The lane index therefore passes through an
            // opaque copy HERE, so everything below is computed where it is used.
            int tq = t;
            asm volatile("" : "+v"(tq));
#pragma unroll
            for (int n0 = 0; n0 < P; ++n0) {
                const int64_t idx = sy.start + tq + T * n0;
                const cf x = any ? frame_iq[idx < 0 ? 0 : (idx < last ? idx : last)] : cf{0.f, 0.f};
                v[n0] = (any && idx >= 0 && idx < a.frame_len) ? x : cf{0.f, 0.f};
                if constexpr (ROT) v[n0] = cmul(v[n0], a.rot[tq + T * n0]);
            }
        }
    };

    // Software pipeline: the data VGPRs are dead once a symbol's bins are scattered into LDS, so the NEXT symbol's
    // loads are issued right there and are in flight during the power sum / equalise / store phases (~45 % of a
    // symbol's time) -- the HBM latency (~25 % of a wave's time when exposed) is hidden at no register cost.
    Sym cur = sym_of(0);
    if constexpr (PIPE) load_sym(cur);
    if constexpr (WL) wg_barrier();              // the shared gain copy and twiddle table: published once, read-only afterwards
    for (int it = 0; it < n_iter; ++it) {
        if constexpr (!PIPE) {
            cur = sym_of(it);
            load_sym(cur);
        }
        const bool sym_valid = cur.valid, compute = cur.compute;
        const int64_t orow = cur.orow;
        stamp(0);                                                        // loop top .. loads issued
        if constexpr (STAMP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(1);                                                        // .. loads landed
        if constexpr (STAMP) {
            fft_pass0_store<N>(v, lds, tw, t);
            wg_barrier();
            stamp(2);                                                    // .. pass 0 + exchange A written
            if constexpr (PL::THREE) {
                fft_pass1_load<N>(v, lds, t);
                wg_barrier();
                fft_pass1_store<N>(v, lds, w1tab, t);
                wg_barrier();
            }
            stamp(3);                                                    // .. pass 1 + exchange B written
            fft_last_load<N>(v, lds, t);
            fft_last_dft<N>(v);
        } else {
            wg_fft<N, decltype(tw), WL>(v, lds, tw, w1tab, t);           // :230
        }
        slot_sync<WL>();                                                 // exchange region -> staging region

        // Re-materialise Kd per symbol so hipcc does not hoist 16 per-slot list offsets into VGPRs held across the loop.
        int Kd_ = Kd;
#if !OFDM_SCATTER_HOIST
        asm volatile("" : "+s"(Kd_));
#endif
        const int hk = Kd_ >> 1;
        // :232 gather into bin-list order: negative half i = k-(N-Kd/2), positive half i = Kd/2+k-1.
        // Unlisted bins fall past the list (i in [Kd, N-2]) and DC is parked at index N (< LDS_ELEMS): no branches.
        const int off_neg = hk - N, off_pos = hk - 1;
        float pse = 0.f;
#pragma unroll
        for (int j = 0; j < PL::C; ++j) {
#pragma unroll
            for (int kl = 0; kl < PL::RL; ++kl) {
#if OFDM_SCATTER_SKIP
                // a register slot holds the T consecutive bins [base, base + T): where none of them is listed (wave-uniform, scalar
                // test) there is nothing to stage and nothing to add to the power sum
                if constexpr (T >= 64) {
                    const int base = T * j + PL::NC * kl;                  // a constant once the loops are unrolled
                    if (base > hk && base + T - 1 < N - hk) continue;
                }
#endif
                const int k = (t + T * j) + PL::NC * kl;
#if OFDM_SCATTER_SKIP
                if constexpr (T >= 64) {
                    // ... and where ALL of them are listed (scalar test again) the power term needs no membership mask
                    const int base = T * j + PL::NC * kl;
                    const bool all_pos = base >= 1 && base + T - 1 <= hk, all_neg = base >= N - hk && hk < N / 2;
                    if (all_pos || all_neg) {
                        lds[k + (all_neg ? off_neg : off_pos)] = v[out_slot<N>(j, kl)];
                        if constexpr (PSE) pse += cnorm2(v[out_slot<N>(j, kl)]);
                        continue;
                    }
                }
#endif
                int i = k + ((k >= N - hk) ? off_neg : off_pos);
                if (j == 0 && kl == 0) i = (k == 0) ? N : i;             // only this slot can hold DC
                lds[i] = v[out_slot<N>(j, kl)];
                if constexpr (PSE) pse += (unsigned(i) < unsigned(Kd_)) ? cnorm2(v[out_slot<N>(j, kl)]) : 0.f;
            }
        }
        if (Kd_ == N) {                                                  // K == N lists bin N/2 twice (ofdm_chain.py:83 wiring)
#pragma unroll
            for (int j = 0; j < PL::C; ++j) {
#pragma unroll
                for (int kl = 0; kl < PL::RL; ++kl)
                    if ((t + T * j) + PL::NC * kl == N / 2) {
                        lds[N - 1] = v[out_slot<N>(j, kl)];
                        if constexpr (PSE) pse += cnorm2(v[out_slot<N>(j, kl)]);
                    }
            }
        }
        if constexpr (PSE) {
            pse = lanes_sum<T>(pse);
            if constexpr (T > 64) {
                if ((t & 63) == 0) red[t >> 6] = pse;
            }
        }
        slot_sync<WL>();

        stamp(4);                                                        // .. pass 2 + scatter into list order
        if constexpr (PIPE) {
            cur = sym_of(it + 1);
            load_sym(cur);                                               // next symbol: in flight until the next FFT
        }
        float psum = 0.f;
        if constexpr (PSE) {
            if constexpr (T > 64) {
#pragma unroll
                for (int w = 0; w < T / 64; ++w) psum += red[w];
            } else {
                psum = pse;
            }
        } else {
            // each lane owns 4 consecutive list entries per q: 16 B LDS reads (read twice: power sum now, output below)
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int idx = 4 * (t + T * q);
                const float4 a01 = *reinterpret_cast<const float4*>(lds + idx);
                const float4 a23 = *reinterpret_cast<const float4*>(lds + idx + 2);
                const float p4[4] = {a01.x * a01.x + a01.y * a01.y, a01.z * a01.z + a01.w * a01.w,
                                     a23.x * a23.x + a23.y * a23.y, a23.z * a23.z + a23.w * a23.w};
                if (4 * T * (q + 1) <= Kd_) {                            // whole wavefront row inside the list (uniform)
#pragma unroll
                    for (int e = 0; e < 4; ++e) psum += p4[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) psum += (idx + e < Kd_) ? p4[e] : 0.f;
                }
            }
            psum = lanes_sum<T>(psum);                                   // :233 sum |x|^2 over the Kd listed bins
            if constexpr (T > 64) {
                if ((t & 63) == 0) red[t >> 6] = psum;
                wg_barrier();
                psum = 0.f;
#pragma unroll
                for (int w = 0; w < T / 64; ++w) psum += red[w];
            }
        }
        // :233 p_est0 = sqrt(Kd / P) = sqrt(Kd) * rsq(P): v_rsq_f32 is good to 1 ulp, far inside the 1e-5 bar; the IEEE division and
        // square root this replaces were ~30 VALU per wave and symbol.  P = 0 (an empty window) gives inf, as before.
        const float scale = sqrt_kd * __builtin_amdgcn_rsqf(psum);
        stamp(5);                                                        // .. list read + power sum

        // A pattern whose guard failed (:223) keeps the zero rows est_data_freq was created with (:88): in batch mode they are
        // WRITTEN as zeros (and the bits as the de-map of 0+0j) at every FFT size and output mapping -- the branch sits outside
        // the mapping split so that no instantiation can lose it.
        if (!compute) {
            if (sym_valid && a.zero_skipped) {
                const cf zz[4] = {cf{0.f, 0.f}, cf{0.f, 0.f}, cf{0.f, 0.f}, cf{0.f, 0.f}};
                int tq = t;                                              // rare path: nothing of it may be hoisted into VGPRs (see load_sym)
                asm volatile("" : "+v"(tq));
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int idx = 4 * (tq + T * q);
                    if (idx < Kd_) {
                        const bool four = idx + 2 < Kd_;
                        if (a.eq) {
                            float4* o = reinterpret_cast<float4*>(a.eq + orow * Kd_ + idx);
                            o[0] = float4{0.f, 0.f, 0.f, 0.f};
                            if (four) o[1] = float4{0.f, 0.f, 0.f, 0.f};
                        }
                        if constexpr (BMODE != 0) store_bits<MOD, BMODE>(a.bits, orow * Kd_ + idx, zz, four ? 4 : 2);
                    }
                }
            }
        } else if constexpr (DENSE) {
            // Dense mapping: per q a WAVE owns two blocks of 128 consecutive list entries (bases 4Tq + 256 wave + {0, 128}), a lane
            // one pair of each.  Whether a block lies inside the list is the same for the whole wave, so it is decided on the
            // scalar unit: blocks past the list are skipped by a branch, blocks wholly inside run without touching exec, and only
            // the one block that straddles Kd is lane-masked.  Row bases are scalar too: every store is `scalar base + 32-bit lane
            // offset` (the 64-bit address chains this replaces were ~60 VALU per wave and symbol).
            const int wv = __builtin_amdgcn_readfirstlane(t >> 6), ln = t & 63;      // scalar: block decisions
            const int lane_e = ((t >> 6) << 8) + 2 * ln;                                 // vector, loop-invariant: this lane's entry in block 0
            const int64_t row0 = orow * Kd_;                                             // uniform per slot
            cf* const eq_row = a.eq ? a.eq + row0 : nullptr;
            uint8_t* const bits_row = (BMODE == 0) ? nullptr : (BMODE == 1) ? a.bits + ((row0 * MOD) >> 3) : a.bits + row0 * MOD;
            const cf* const l_lane = lds + lane_e;
            const cf* const g_lane = gsrc + lane_e;
            const unsigned eq_off = unsigned(lane_e) * unsigned(sizeof(cf));             // byte offset of the lane's pair within a row
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int b0 = 4 * T * q + 256 * wv;                                      // scalar
                if (b0 < Kd_) {
                    unsigned wp[2] = {0u, 0u};                                            // packed bits of this lane's two pairs
#pragma unroll
                    for (int hx = 0; hx < 2; ++hx) {
                        const int base = b0 + 128 * hx;
                        const int cofs = 4 * T * q + 128 * hx;                            // compile-time part of the block base
                        auto pair = [&]() {
                            float4 g;
                            if constexpr (GREG)
                                g = greg[q][hx];
                            else
                                g = *reinterpret_cast<const float4*>(g_lane + cofs);
                            const float4 av = *reinterpret_cast<const float4*>(l_lane + cofs);
                            // :235-248  x * p_est0 * e^{j..} * gain
                            cf z[2] = {cf{av.x, av.y} * scale, cf{av.z, av.w} * scale};
                            cmul2(z[0], cf{g.x, g.y}, z[1], cf{g.z, g.w});
                            if (eq_row) {
                                typedef float f4 __attribute__((ext_vector_type(4)));
                                const f4 ov = f4{z[0].x, z[0].y, z[1].x, z[1].y};
                                // scalar block address + the lane's constant 32-bit byte offset: no address arithmetic on the VALU
                                cf* const blk = eq_row + cofs;
#ifdef OFDM_EXPERIMENTS
                                // cache-policy study of the output store (FLAGS bits 12-13): sc1 / sc0 sc1 / sc0 sc1 nt
                                constexpr unsigned SPOL = (FLAGS >> 12) & 3u;
                                if constexpr (SPOL == 1) {
                                    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(eq_off), "v"(ov), "s"(blk) : "memory");
                                } else if constexpr (SPOL == 2) {
                                    asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(eq_off), "v"(ov), "s"(blk) : "memory");
                                } else if constexpr (SPOL == 3) {
                                    asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt" ::"v"(eq_off), "v"(ov), "s"(blk) : "memory");
                                } else
#endif
                                if constexpr (NT) {
                                    asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(eq_off), "v"(ov), "s"(blk) : "memory");
                                } else {
                                    asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(eq_off), "v"(ov), "s"(blk) : "memory");
                                }
                            }
                            if constexpr (BMODE == 1) wp[hx] = pack2<MOD, ASMB>(z);
                            if constexpr (BMODE == 2) store_bits_pair_unpacked<MOD>(bits_row + cofs * MOD, unsigned(lane_e), z);
                        };
                        if (base + 128 <= Kd_) {                                          // scalar: the whole block is listed
                            pair();
                        } else if (base < Kd_) {                                          // scalar: the block that straddles Kd
                            if (base + 2 * ln < Kd_) pair();
                        }
                    }
                    if constexpr (BMODE == 1) {
                        // Lanes 2m, 2m+1 hold the four consecutive entries of a group in BOTH halves: the even lane writes the group
                        // of the first half, the odd lane the group of the second -- one exchange through a quad permute, then
                        // the same whole-byte stores as with four entries per lane, every lane busy.  (Kd % 4 == 0: a group is in
                        // or out as one.)
                        const bool odd = (ln & 1) != 0;
                        const unsigned recv = unsigned(__builtin_amdgcn_mov_dpp(int(odd ? wp[0] : wp[1]), 0xB1, 0xF, 0xF, false));   // quad_perm:[1,0,3,2]
                        const unsigned w4 = odd ? ((recv << (2 * MOD)) | wp[1]) : ((wp[0] << (2 * MOD)) | recv);
                        const int ge = lane_e + (odd ? 126 : 0);                          // group's first entry relative to 4 T q (vector, loop-invariant)
                        uint8_t* const brow = bits_row + (4 * T * q / 4) * (MOD / 2);     // scalar
                        if (b0 + 256 <= Kd_) {
                            store_packed4<MOD>(brow, unsigned(ge), w4);
                        } else if (4 * T * q + ge < Kd_) {
                            store_packed4<MOD>(brow, unsigned(ge), w4);
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int idx = 4 * (t + T * q);
                if (idx < Kd_) {
                    const bool four = idx + 2 < Kd_;
                    const float4 g01 = *reinterpret_cast<const float4*>(gsrc + idx);
                    const float4 g23 = (GLDS || four) ? *reinterpret_cast<const float4*>(gsrc + idx + 2) : float4{0.f, 0.f, 0.f, 0.f};
                    const float4 a01 = *reinterpret_cast<const float4*>(lds + idx);
                    const float4 a23 = *reinterpret_cast<const float4*>(lds + idx + 2);
                    cf z[4] = {cf{a01.x, a01.y} * scale, cf{a01.z, a01.w} * scale, cf{a23.x, a23.y} * scale,
                               cf{a23.z, a23.w} * scale};                 // :235-248  x * p_est0 * e^{j..} * gain
                    cmul3(z[0], cf{g01.x, g01.y}, z[1], cf{g01.z, g01.w}, z[2], cf{g23.x, g23.y});
                    z[3] = cmul(z[3], cf{g23.z, g23.w});
                    if (a.eq) {
                        float4* o = reinterpret_cast<float4*>(a.eq + orow * Kd_ + idx);
                        const float4 o0 = float4{z[0].x, z[0].y, z[1].x, z[1].y}, o1 = float4{z[2].x, z[2].y, z[3].x, z[3].y};
                        if constexpr (NT) {
                            typedef float f4 __attribute__((ext_vector_type(4)));
                            f4* on = reinterpret_cast<f4*>(o);
                            __builtin_nontemporal_store(f4{o0.x, o0.y, o0.z, o0.w}, on);
                            if (four) __builtin_nontemporal_store(f4{o1.x, o1.y, o1.z, o1.w}, on + 1);
                        } else {
                            o[0] = o0;
                            if (four) o[1] = o1;
                        }
                    }
                    if constexpr (BMODE != 0) store_bits<MOD, BMODE, ASMB>(a.bits, orow * Kd_ + idx, z, four ? 4 : 2);
                }
            }
        }
        stamp(6);                                                        // .. equalise + de-map + stores issued
        slot_sync<WL>();                                                 // staging region free for the next symbol
        stamp(7);                                                        // .. loop-end barrier
    }
    if (!a.work) break;
    // next chunk: every wave is done with the LDS of this one (gain copy, staging) before the slot word and the gains change
    __syncthreads();
    if (tid == 0) *qslot = int(atomicAdd(a.work, 1u));
    __syncthreads();
    chunk = *qslot;
    }   // chunks
    if (a.work) {
        // the last workgroup to leave re-arms the two words for the next launch (every fetch of this launch lies before its own exit)
        if (tid == 0 && atomicAdd(a.work + 1, 1u) == gridDim.x - 1) {
            atomicExch(a.work, 0u);
            atomicExch(a.work + 1, 0u);
        }
    }
    if constexpr (STAMP) {
        if ((tid & 63) == 0 && a.stamps) {
            unsigned* o = a.stamps + (int64_t(blockIdx.x) * (DG::WG / 64) + (tid >> 6)) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = acc[i];
        }
    }
}

// one translation unit per FFT size instantiates this (rx_demod_<N>.hip)
template <int N>
hipError_t launch_rx_demod_n(const RxDev& rx, const DemodArgs& a_in, hipStream_t s) {
    using DG = DemodGeom<N>;
    DemodArgs a = a_in;
    if (a.n_frames <= 0 || a.n_dsym <= 0) return hipSuccess;
    if (a.chunks_per_frame <= 0) {
        // Auto chunking.  Workgroup w works on chunk w % cpf of frame w / cpf, and the hardware hands workgroups to the 8 XCDs
        // round-robin (w % 8) -- statically: an XCD that gets less work is idle at the end, nothing is re-balanced.  With the
        // round-2 rule (chunks of 48, the remainder in the frame's last chunk: 48 48 48 36 at 180 symbols per frame, cpf = 4)
        // the short chunk of every frame landed on XCDs 3 and 7, the others carried 6.7 % more than the average, and the
        // kernel took 5.5 % longer than with equal chunks (profiles/r03_demod_chunking.txt).  So: (1) chunks of EQUAL length
        // where the frame's trips divide evenly, (2) else lengths that differ by one trip with an ODD chunk count per frame,
        // so that every XCD sees every chunk index equally often; 12-20 symbols per chunk (the frame's gains are copied
        // into LDS once per chunk; >= 20 rounds of resident workgroups at the bench sizes keep the tail short).
        const int trips = (a.n_dsym + DG::NS - 1) / DG::NS;
        const int64_t total_trips = int64_t(a.n_frames) * trips;
        // symbols per chunk aimed at.  A/B with the work queue on (profiles/r03_demod_chunking.txt): 9-10 chunks per 180-symbol frame
        // at 2048-pt, 12-15 at 4096-pt, 15 at 1024-pt (without the queue, where twiddles and tables are set up per chunk: 6 / 9 / 9)
        // Below 1024-pt a symbol is a few hundred bytes and a trip a few microseconds: a chunk is at least ~128 KB of input there
        // (a whole 180-symbol frame at 64-pt), or the per-chunk costs (queue, gain copy, unpipelined first loads) take over.
        const int per_chunk = N == 2048 ? 20 : N >= 4096 ? 14 : N == 1024 ? 12 : std::max(12, 131072 / (rx.L * 8));
        int want = int(std::max<int64_t>(1, (int64_t(trips) * DG::NS + per_chunk - 1) / per_chunk));
        if (total_trips / std::max(want, 1) < 4096) want = int(std::max<int64_t>(1, std::min<int64_t>(trips, 4096 / std::max(a.n_frames, 1))));  // few frames: finer
#ifdef OFDM_TUNE_ENV     // study builds only (make geom GEOMFLAGS=-DOFDM_TUNE_ENV): chunk count per frame from the environment
        if (const char* e = getenv("OFDM_DEMOD_CPF")) want = atoi(e);
#endif
        want = std::max(1, std::min(want, trips));
        int best = want;
        if (trips % want != 0) {                                            // nearest divisor of trips within [want/1.5, want*1.5], else odd
            best = 0;
            for (int d = 0; d <= want / 2 && !best; ++d) {
                if (want + d <= trips && trips % (want + d) == 0) best = want + d;
                else if (want - d >= 1 && trips % (want - d) == 0 && 3 * (want - d) >= 2 * want) best = want - d;
            }
            if (!best) best = (want & 1) ? want : std::min(want + 1, trips);
        }
        a.chunks_per_frame = best;
        a.spc = ((trips + best - 1) / best) * DG::NS;                       // longest chunk (informational)
    }
    unsigned grid = unsigned(int64_t(a.n_frames) * a.chunks_per_frame);
    size_t lds = DG::lds_bytes(rx.Kd, !((Plan<N>::T == 128 && (OFDM_FLAGS_T128 & (DF_GAINS_GLOBAL | DF_GAINS_VGPR))) ||
                                        (Plan<N>::T == 256 && (OFDM_FLAGS_T256 & (DF_GAINS_GLOBAL | DF_GAINS_VGPR)))));
#ifdef OFDM_EXPERIMENTS
    if (a.variant >= 100) lds += size_t(a.variant - 100) * 1024;   // occupancy experiment: pad the LDS request by (variant-100) KiB
#endif
    const int bmode = a.bits ? a.bits_mode : 0;
    if (a.host_guard) {             // tracker receiver: equalised symbols only, frames enabled by the host
        if (bmode != 0 || a.rot) return hipErrorInvalidValue;
        hipLaunchKernelGGL((rx_demod_kernel<N, 2, 0, 3, DF_HG>), dim3(grid), dim3(DG::WG), lds, s, rx, a);
        return hipGetLastError();
    }
    if (a.rot) {                    // CFO receiver: equalised symbols only
        if (bmode != 0) return hipErrorInvalidValue;
        hipLaunchKernelGGL((rx_demod_kernel<N, 2, 0, 3, DF_ROT>), dim3(grid), dim3(DG::WG), lds, s, rx, a);
        return hipGetLastError();
    }
#ifdef OFDM_EXPERIMENTS
    // Tuning / diagnostic builds (tools/experiments: libofdm_mi355x_exp.so, ofdm_exp_set_variant): 16-QAM packed only.
    // None of them is compiled into the product library; an unknown variant falls through to the shipped kernel.
    if constexpr (N == 2048) {
        if ((a.variant == 13 || a.variant == 1) && bmode == 1 && a.mod != 4) {   // the same two at QPSK / 64-QAM (16-QAM: below)
            constexpr unsigned PLAIN = DF_TEMPORAL_LD | DF_TEMPORAL_ST;
            if (a.mod == 6 && a.variant == 13)
                hipLaunchKernelGGL((rx_demod_kernel<N, 6, 1, 3, DF_FOUR_PER_LANE | PLAIN>), dim3(grid), dim3(DG::WG), lds, s, rx, a);
            else if (a.mod == 6)
                hipLaunchKernelGGL((rx_demod_kernel<N, 6, 1, 3, PLAIN>), dim3(grid), dim3(DG::WG), lds, s, rx, a);
            else if (a.variant == 13)
                hipLaunchKernelGGL((rx_demod_kernel<N, 2, 1, 3, DF_FOUR_PER_LANE | PLAIN>), dim3(grid), dim3(DG::WG), lds, s, rx, a);
            else
                hipLaunchKernelGGL((rx_demod_kernel<N, 2, 1, 3, PLAIN>), dim3(grid), dim3(DG::WG), lds, s, rx, a);
            return hipGetLastError();
        }
        if (a.variant != 0 && a.variant < 100 && bmode == 1 && a.mod == 4) {
#define OFDM_LV(V, MW, FL, LDSB)                                                                                            \
    if (a.variant == V) {                                                                                                   \
        hipLaunchKernelGGL((rx_demod_kernel<N, 4, 1, MW, FL>), dim3(grid), dim3(DG::WG), LDSB, s, rx, a);                   \
        return hipGetLastError();                                                                                           \
    }
            OFDM_LV(1, 3, DF_TEMPORAL_LD | DF_TEMPORAL_ST, lds)
            OFDM_LV(14, 3, DF_TEMPORAL_ST, lds)                      // non-temporal loads only
            OFDM_LV(21, 3, 1u << 12, lds)                            // output stores with sc1
            OFDM_LV(22, 3, 2u << 12, lds)                            // ... sc0 sc1
            OFDM_LV(23, 3, 3u << 12, lds)                            // ... sc0 sc1 nt
            OFDM_LV(15, 3, DF_TEMPORAL_LD, lds)                      // non-temporal stores only
            OFDM_LV(2, 3, DF_GAINS_GLOBAL, DG::lds_bytes(rx.Kd, false))
            OFDM_LV(3, 3, DF_GENERIC_BITS, lds)
            OFDM_LV(5, 3, DF_NO_PIPE, lds)
            OFDM_LV(6, 3, DF_L2_INPUT, lds)
            OFDM_LV(8, 4, 0u, lds)                                   // 128-VGPR budget
            OFDM_LV(9, 3, DF_STAMP, lds)
            OFDM_LV(10, 3, DF_LANE_TWIDDLES | DF_PSUM_READBACK, lds) // the round-1 kernel
            OFDM_LV(11, 3, DF_PSUM_READBACK, lds)
            OFDM_LV(12, 3, DF_LANE_TWIDDLES, lds)
            OFDM_LV(13, 3, DF_FOUR_PER_LANE | DF_TEMPORAL_LD | DF_TEMPORAL_ST, lds)   // round-1 output mapping, plain accesses
#undef OFDM_LV
        }
    }
#endif
    constexpr int MW = (Plan<N>::T == 128) ? OFDM_MINW_T128 : 3;
    constexpr unsigned FL = (Plan<N>::T == 128) ? OFDM_FLAGS_T128 : (Plan<N>::T == 256) ? OFDM_FLAGS_T256 : 0u;
    // more than 64 KB of dynamic LDS per workgroup (4 slots at 2048-pt) has to be announced once per kernel
#define OFDM_LD(M, B)                                                                                                        \
    do {                                                                                                                     \
        static std::atomic<int> announced{65536};          /* grows with Kd (gain table): announce every new maximum */      \
        if (int(lds) > announced.load(std::memory_order_relaxed)) {                                                          \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rx_demod_kernel<N, M, B, MW, FL>),             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));                       \
            if (e != hipSuccess) return e;                                                                                   \
            announced.store(int(lds), std::memory_order_relaxed);                                                            \
        }                                                                                                                    \
        hipLaunchKernelGGL((rx_demod_kernel<N, M, B, MW, FL>), dim3(grid), dim3(DG::WG), lds, s, rx, a);                     \
    } while (0)
    // below 1024-pt a chunk is a whole frame or a large part of one and the static grid measured 1-5 % ahead of the queue
    // (profiles/r03_small_sizes.txt)
    if (N < 1024) a.work = nullptr;
    if (a.work) {
        // work queue: the resident workgroups only (occupancy of the 16-QAM packed instantiation stands for all of them: the
        // register counts of the MOD / BMODE variants differ by a few VGPRs inside one occupancy step; the queue itself is
        // correct with any grid size)
        static int per_cu = 0, n_cu = 0;
        if (per_cu == 0) {
            int nb = 0, dev = 0;
            hipDeviceProp_t prop;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rx_demod_kernel<N, 4, 1, MW, FL>, DG::WG, lds) != hipSuccess || nb < 1) nb = 2;
            n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                       ? prop.multiProcessorCount : 256;
            per_cu = nb;
        }
        const int64_t resident = int64_t(per_cu) * n_cu;
        if (int64_t(grid) > 2 * resident)
            grid = unsigned(resident);
        else
            a.work = nullptr;                     // not worth a queue: one chunk per workgroup
    }
#define OFDM_LD_MOD(B)                  \
    switch (a.mod) {                    \
        case 1: OFDM_LD(1, B); break;   \
        case 2: OFDM_LD(2, B); break;   \
        case 4: OFDM_LD(4, B); break;   \
        default: OFDM_LD(6, B); break;  \
    }
    if (bmode == 0) {
        OFDM_LD(2, 0);
    } else if (bmode == 1) {
        OFDM_LD_MOD(1)
    } else {
        OFDM_LD_MOD(2)
    }
#undef OFDM_LD_MOD
#undef OFDM_LD
    return hipGetLastError();
}

#define OFDM_DECLARE_DEMOD(n) hipError_t launch_rx_demod_##n(const RxDev& rx, const DemodArgs& a, hipStream_t s);
OFDM_DECLARE_DEMOD(64)
OFDM_DECLARE_DEMOD(128)
OFDM_DECLARE_DEMOD(256)
OFDM_DECLARE_DEMOD(512)
OFDM_DECLARE_DEMOD(1024)
OFDM_DECLARE_DEMOD(2048)
OFDM_DECLARE_DEMOD(4096)
#undef OFDM_DECLARE_DEMOD

}  // namespace ofdm
