// copy_bw.hip -- what can this chip's HBM actually do?  Variants of a streaming copy / read / write (standalone micro-benchmark).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/copy_bw.hip -o tools/ubench/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// A: grid-stride, one float4 per iteration
__global__ void __launch_bounds__(256) copy_a(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) out[i] = in[i];
}
// B: grid-stride, U independent float4 per iteration (U wave-instructions in flight per wave)
template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_b(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], out + i + u * stride); else out[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) out[i] = in[i];
}
// C: every workgroup owns a contiguous chunk of CH float4 (like a symbol), U loads in flight
template <int U>
__global__ void __launch_bounds__(256) copy_c(const f4* __restrict__ in, f4* __restrict__ out, size_t n, size_t ch) {
    for (size_t c = blockIdx.x; c * ch < n; c += gridDim.x) {
        const f4* s = in + c * ch;
        f4* d = out + c * ch;
        for (size_t i = threadIdx.x; i + (U - 1) * 256 < ch; i += U * 256) {
            f4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = s[i + u * 256];
#pragma unroll
            for (int u = 0; u < U; ++u) d[i + u * 256] = v[u];
        }
    }
}
// R: read only (sum), W: write only
template <int U>
__global__ void __launch_bounds__(256) read_r(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += in[i + u * stride];
    }
    if (acc.x == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(256) write_w(f4* __restrict__ out, size_t n) {
    const f4 v = {1, 2, 3, 4};
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) out[i] = v;
}

template <class F>
double time_ms(F&& launch, int reps = 7) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? atof(argv[1]) : 4.0) * (size_t(1) << 30);
    const size_t n = bytes / 16;
    f4 *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
    auto rep = [&](const char* name, double ms, double factor) { printf("%-34s %8.3f ms  %7.1f GB/s\n", name, ms, factor * bytes / ms / 1e6); };
    for (unsigned g : {1024u, 2048u, 4096u, 8192u, 16384u, 65536u}) {
        char nm[64];
        snprintf(nm, sizeof nm, "copy A grid %u", g);
        rep(nm, time_ms([&] { hipLaunchKernelGGL(copy_a, dim3(g), dim3(256), 0, 0, in, out, n); }), 2);
    }
    for (unsigned g : {2048u, 4096u, 16384u}) {
        char nm[64];
        snprintf(nm, sizeof nm, "copy B U=4 grid %u", g);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_b<4, false>), dim3(g), dim3(256), 0, 0, in, out, n); }), 2);
        snprintf(nm, sizeof nm, "copy B U=8 grid %u", g);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_b<8, false>), dim3(g), dim3(256), 0, 0, in, out, n); }), 2);
        snprintf(nm, sizeof nm, "copy B U=4 nt grid %u", g);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_b<4, true>), dim3(g), dim3(256), 0, 0, in, out, n); }), 2);
    }
    for (size_t ch : {size_t(1024), size_t(4096), size_t(65536)}) {
        char nm[64];
        snprintf(nm, sizeof nm, "copy C chunk %zu KB U=4 grid 8192", ch * 16 / 1024);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((copy_c<4>), dim3(8192), dim3(256), 0, 0, in, out, n, ch); }), 2);
    }
    rep("hipMemcpyDtoD", time_ms([&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); }), 2);
    for (unsigned g : {2048u, 8192u}) {
        char nm[64];
        snprintf(nm, sizeof nm, "read  U=4 grid %u", g);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((read_r<4>), dim3(g), dim3(256), 0, 0, in, out, n); }), 1);
        snprintf(nm, sizeof nm, "write     grid %u", g);
        rep(nm, time_ms([&] { hipLaunchKernelGGL(write_w, dim3(g), dim3(256), 0, 0, out, n); }), 1);
    }
    return 0;
}
