// Does a consumer kernel that walks a freshly written tensor BACKWARDS find it in the Infinity Cache (256 MiB, LRU-like) where a forward walk finds nothing?
// A producer streams N bytes out front to back; the consumer then reads them (16 bytes per lane, grid-stride over 64 KiB blocks) front to back or back to front.  N around and above the cache size
// (the step's tensors: 67 / 134 / 268 MB in bf16 mode, 134 MB in fp32 mode).
//   hipcc -O3 --offload-arch=gfx950 mall_order_probe.hip -o mall_order_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void produce(u32x4* dst, size_t nblk) {               // block = 4096 units of 16 bytes = 64 KiB
    for (size_t b = blockIdx.x; b < nblk; b += gridDim.x)
        for (int i = threadIdx.x; i < 4096; i += 256) dst[b * 4096 + i] = u32x4{(unsigned)b, (unsigned)i, 1u, 2u};
}
template <bool REV>
__global__ __launch_bounds__(256) void consume(const u32x4* src, unsigned* out, size_t nblk) {
    unsigned acc = 0;
    for (size_t k = blockIdx.x; k < nblk; k += gridDim.x) {
        const size_t b = REV ? nblk - 1 - k : k;
        for (int i = threadIdx.x; i < 4096; i += 256) { const u32x4 v = src[b * 4096 + i]; acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    u32x4* buf; unsigned* out;
    (void)hipMalloc(&buf, (size_t)1 << 30); (void)hipMalloc(&out, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (size_t mb : {67, 134, 201, 268, 402, 537}) {
        const size_t nblk = mb * 1000000 / 65536;
        for (int rev = 0; rev < 2; ++rev) {
            float tot = 0.f;
            for (int rep = 0; rep < 6; ++rep) {
                hipLaunchKernelGGL(produce, dim3(256 * 8), dim3(256), 0, 0, buf, nblk);
                (void)hipEventRecord(e0);
                if (rev) hipLaunchKernelGGL(consume<true>, dim3(256 * 8), dim3(256), 0, 0, buf, out, nblk);
                else hipLaunchKernelGGL(consume<false>, dim3(256 * 8), dim3(256), 0, 0, buf, out, nblk);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep) tot += ms;
            }
            printf("%4zu MB written front to back, then read %s: %7.1f us  %7.1f GB/s\n", mb, rev ? "BACK TO FRONT" : "front to back", tot / 5 * 1e3, nblk * 65536.0 / (tot / 5 * 1e-3) / 1e9);
        }
    }
    return 0;
}
