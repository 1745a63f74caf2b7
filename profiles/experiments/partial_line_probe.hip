// Does the MI355X fetch whole 128-byte lines from HBM when a kernel reads only 32 (or 64) bytes of each?  Streams a 2 GiB buffer (8 x the Infinity Cache) once per launch,
// reading P bytes of every 128-byte line (P = 128: every byte; 64; 32 = what a 16-channel chunk of a 64-channel bf16 pixel row is), 16 bytes per lane, and reports lines per
// second and useful GB/s.  If partial reads run at the line rate of the full read, HBM moves whole lines; if they run faster, it moves sectors — which decides how
// rocprofv3's FETCH_SIZE (requests x 64 B, doubled for 128-byte requests: MI355X_MICROARCH.md, HBM) is to be read for the big-tile input-gradient kernels.
//   hipcc -O3 --offload-arch=gfx950 partial_line_probe.hip -o partial_line_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int P>      // bytes read per 128-byte line
__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ src, unsigned* __restrict__ out, size_t lines) {
    constexpr int LPL = P / 16;                                   // lanes per line
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (size_t i = gid; i < lines * LPL; i += stride) {
        const size_t line = i / LPL, part = i % LPL;
        const u32x4 v = src[line * 8 + part];
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) out[0] = acc;                          // keeps the loads alive
}
template <int P> static void run(const u32x4* src, unsigned* out, size_t lines) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(rd<P>, dim3(256 * 16), dim3(256), 0, 0, src, out, lines);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double s = ms * 1e-3 / 5;
    printf("%3d bytes of every 128-byte line: %7.3f ms per pass  %6.2f G lines/s  %7.1f GB/s useful  (%7.1f GB/s if whole lines move)\n", P, s * 1e3, lines / s / 1e9, lines * (double)P / s / 1e9, lines * 128.0 / s / 1e9);
}
int main() {
    const size_t bytes = (size_t)2 << 30, lines = bytes / 128;
    u32x4* src; unsigned* out;
    (void)hipMalloc(&src, bytes); (void)hipMalloc(&out, 64);
    (void)hipMemset(src, 1, bytes);
    run<128>(src, out, lines); run<64>(src, out, lines); run<32>(src, out, lines); run<16>(src, out, lines); run<128>(src, out, lines);
    return 0;
}
