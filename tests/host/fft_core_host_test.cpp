// Host-side lane-by-lane execution of csrc/fft_core.hpp (no GPU): verifies the FFT index logic,
// twiddles and exchange layouts against numpy.fft in tests/test_fft_core_host.py.
// usage: fft_core_host_test N in.bin out.bin   (complex64 little-endian, N points in, N bins out)
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../lte-gnu-radio-code_amd/csrc/fft_core.hpp"

using namespace ofdm;

template <int N>
int run(const char* fin, const char* fout) {
    using PL = Plan<N>;
    std::vector<cf> x(N), X(N), tab(N), lds(PL::LDS_ELEMS);
    FILE* f = fopen(fin, "rb");
    if (!f || fread(x.data(), sizeof(cf), N, f) != (size_t)N) return 2;
    fclose(f);
    for (int j = 0; j < N; ++j) {
        const double a = -2.0 * M_PI * j / N;
        tab[j] = cf{(float)cos(a), (float)sin(a)};
    }
    std::vector<LaneTwiddles<N>> tw(PL::T);
    std::vector<cf> w1tab(16 * 16);
    if constexpr (PL::THREE)
        for (int e = 0; e < PL::W1_ELEMS; ++e) w1tab[e] = w1_entry<N>(tab.data(), e);
    std::vector<cf> regs(PL::T * PL::P);
    auto V = [&](int t) -> cf(&)[PL::P] { return *reinterpret_cast<cf(*)[PL::P]>(&regs[t * PL::P]); };
    for (int t = 0; t < PL::T; ++t) load_twiddles<N>(tw[t], tab.data(), t);
    for (int t = 0; t < PL::T; ++t) {
        for (int n0 = 0; n0 < PL::P; ++n0) V(t)[n0] = x[t + PL::T * n0];
        fft_pass0_store<N>(V(t), lds.data(), tw[t], t);
    }
    if constexpr (PL::THREE) {
        for (int t = 0; t < PL::T; ++t) fft_pass1_load<N>(V(t), lds.data(), t);
        for (int t = 0; t < PL::T; ++t) fft_pass1_store<N>(V(t), lds.data(), w1tab.data(), t);
    }
    for (int t = 0; t < PL::T; ++t) {
        fft_last_load<N>(V(t), lds.data(), t);
        fft_last_dft<N>(V(t));
        for (int j = 0; j < PL::C; ++j)
            for (int kl = 0; kl < PL::RL; ++kl) X[(t + PL::T * j) + PL::NC * kl] = V(t)[out_slot<N>(j, kl)];
    }
    f = fopen(fout, "wb");
    fwrite(X.data(), sizeof(cf), N, f);
    fclose(f);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 4) return 1;
    switch (atoi(argv[1])) {
        case 64: return run<64>(argv[2], argv[3]);
        case 128: return run<128>(argv[2], argv[3]);
        case 256: return run<256>(argv[2], argv[3]);
        case 512: return run<512>(argv[2], argv[3]);
        case 1024: return run<1024>(argv[2], argv[3]);
        case 2048: return run<2048>(argv[2], argv[3]);
        case 4096: return run<4096>(argv[2], argv[3]);
    }
    return 3;
}
