/* c_host.c -- drives the batch path of libofdm_mi355x.so from plain C (no Python, no torch): the boundary is a C ABI.
 *
 *   gcc -O2 -I include examples/c_host.c -o examples/c_host -L lte-gnu-radio-code_amd/ofdm_mi355x -lofdm_mi355x -lm \
 *       -Wl,-rpath,'$ORIGIN/../lte-gnu-radio-code_amd/ofdm_mi355x'
 *
 * bits -> ofdm_tx_modulate_frames -> ofdm_channel_apply (reference 5-tap profile + AWGN) -> ofdm_rx_demod_frames -> bits,
 * at the reference's configuration-1 numerology (64-pt FFT, 16-sample CP, QPSK, [1,3] pattern).  Exit code 0 iff every
 * demodulated bit equals the transmitted one. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ofdm_mi355x.h"

#define CK(call)                                                                    \
    do {                                                                            \
        long long rc_ = (long long)(call);                                          \
        if (rc_ < 0) {                                                              \
            fprintf(stderr, "%s failed (%lld): %s\n", #call, rc_, ofdm_last_error()); \
            return 2;                                                               \
        }                                                                           \
    } while (0)

int main(void) {
    const int N = 64, cp = 16, Ks = 62, Kd = 60, S = 1, D = 3, n_sym = 240, n_frames = 64, dev = 0;
    const int L = N + cp, fl = n_sym * L, nds = n_sym / (S + D) * D, bits_per_frame = nds * Kd * 2;

    ofdm_tx_cfg tc = {N, cp, Ks, Kd, S, D, 2, 23, dev, 0};
    ofdm_rx_cfg rc = {n_sym, N, cp, Ks, S, D, Kd, 100.0, 0.7, OFDM_COMPAT_UTSA, 2, dev, 0};
    ofdm_tx* tx = NULL;
    ofdm_rx* rx = NULL;
    CK(ofdm_tx_create(&tc, &tx));
    CK(ofdm_rx_create(&rc, &rx));

    unsigned char* bits = malloc((size_t)n_frames * bits_per_frame);
    unsigned char* got = malloc((size_t)n_frames * bits_per_frame);
    srand(7);
    for (long i = 0; i < (long)n_frames * bits_per_frame; ++i) bits[i] = (unsigned char)(rand() & 1);
    const double t5[10] = {0.3977, 0, 0.7954, -0.3977, -0.1988, 0, 0.0994, 0, -0.0398, 0};   /* MultiAntennaSystem.py:64 */
    double nrm = 0;
    for (int i = 0; i < 10; ++i) nrm += t5[i] * t5[i];
    float taps[10];
    for (int i = 0; i < 10; ++i) taps[i] = (float)(t5[i] / sqrt(nrm));

    void *d_bits, *d_tx, *d_rx, *d_taps, *d_out;
    CK(ofdm_device_malloc(dev, &d_bits, (long long)n_frames * bits_per_frame));
    CK(ofdm_device_malloc(dev, &d_tx, (long long)n_frames * fl * 8));
    CK(ofdm_device_malloc(dev, &d_rx, (long long)n_frames * fl * 8));
    CK(ofdm_device_malloc(dev, &d_taps, sizeof taps));
    CK(ofdm_device_malloc(dev, &d_out, (long long)n_frames * bits_per_frame));
    CK(ofdm_memcpy_h2d(dev, d_bits, bits, (long long)n_frames * bits_per_frame));
    CK(ofdm_memcpy_h2d(dev, d_taps, taps, sizeof taps));

    CK(ofdm_tx_modulate_frames(tx, d_bits, OFDM_BITS_UNPACKED, n_frames, n_sym, d_tx, fl, NULL));
    CK(ofdm_device_synchronize(dev));
    CK(ofdm_channel_apply(tx, d_tx, n_frames, fl, fl, d_taps, 5, 0, 1e-4f, 1234, d_rx, fl, fl, NULL));
    CK(ofdm_device_synchronize(dev));
    long long per_frame = ofdm_rx_demod_frames(rx, d_rx, n_frames, fl, fl, NULL, d_out, OFDM_BITS_UNPACKED, NULL, NULL);
    CK(per_frame);
    CK(ofdm_device_synchronize(dev));
    CK(ofdm_memcpy_d2h(dev, got, d_out, (long long)n_frames * bits_per_frame));

    long errors = 0;
    for (long i = 0; i < (long)n_frames * bits_per_frame; ++i) errors += got[i] != bits[i];
    printf("c_host: abi %d, %d frames x %d symbols, %lld data symbols per frame, bit errors %ld / %ld\n", ofdm_abi_version(),
           n_frames, n_sym, per_frame, errors, (long)n_frames * bits_per_frame);

    ofdm_device_free(dev, d_bits);
    ofdm_device_free(dev, d_tx);
    ofdm_device_free(dev, d_rx);
    ofdm_device_free(dev, d_taps);
    ofdm_device_free(dev, d_out);
    ofdm_rx_destroy(rx);
    ofdm_tx_destroy(tx);
    free(bits);
    free(got);
    return errors == 0 ? 0 : 1;
}
