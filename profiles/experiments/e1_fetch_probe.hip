// What does FETCH_SIZE report for e1_fwd_bf16_kernel's frame loads?  (both E1 forward passes show 2.4x the
// 100.7 MB of x under the x2 convention that was calibrated on 16-byte-per-lane loads.)
//   hipcc -O3 --offload-arch=gfx950 e1_fetch_probe.hip -o e1_fetch_probe.bin
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./e1_fetch_probe.bin
// Kernels, all reading the same (B,3,64,64) fp32 tensor once (B = 2048: 100.66 MB):
//   k_stream      16 bytes per lane, fully coalesced                         (the guide's x2 case)
//   k_dword       4 bytes per lane, fully coalesced, no halo                 (is the factor 2 also right for dword loads?)
//   k_strip       exactly e1_fwd_bf16's loads: persistent 768 workgroups, 16x32 strips with a 20x40 halo, 4 bytes per lane,
//                 three channel planes, strip = workgroup + k * grid           (halo 800 / 512 = 1.56x the pixels requested)
//   k_strip_xcd   the same with the strips of a frame kept on ONE XCD (neighbouring strips share halo rows in one L2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int H = 64, SR = 16, SW = 32, HR_ = SR + 4, HWX = 40, SX = H / SW, SY = H / SR;

__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ x, float* out, size_t n4) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_dword(const float* __restrict__ x, float* out, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += x[i];
    if (s == 12345.678f) out[0] = s;
}
template <bool XCD>
__global__ __launch_bounds__(256) void k_strip(const float* __restrict__ x, float* out, int numStrips) {
    const int tid = threadIdx.x;
    constexpr int NIT = (HR_ * HWX + 255) / 256;
    float s = 0.f;
    const int G = gridDim.x;
    for (int it = blockIdx.x; it < numStrips; it += G) {
        // XCD: workgroup w (XCD w % 8) walks frames f = w % 8 + 8 k' ... every strip of a frame on the same XCD
        int strip = it;
        if (XCD) { const int perFrame = SX * SY, f = (it / (8 * perFrame)) * 8 + (it & 7), t = (it >> 3) % perFrame; strip = f * perFrame + t; }
        const int ib = strip / (SX * SY), t = strip % (SX * SY);
        const int ty0 = (t / SX) * SR, tx0 = (t % SX) * SW;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int q = tid + i * 256, hy = q / HWX, hx = q % HWX;
            const int gy = ty0 + hy - 2, gx = tx0 + hx - 2;
            const bool ok = q < HR_ * HWX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)H;
            const size_t e = ok ? ((size_t)(ib * 3) * H + gy) * H + gx : 0;
            const float a = x[e], b = x[e + (size_t)H * H], c = x[e + 2 * (size_t)H * H];
            s += ok ? a + b + c : 0.f;
        }
    }
    if (s == 12345.678f) out[0] = s;
}
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2048;
    const size_t n = (size_t)B * 3 * H * H;
    float *x, *out;
    hipMalloc(&x, n * 4); hipMalloc(&out, 64);
    hipMemset(x, 0, n * 4);
    float* trash; hipMalloc(&trash, 512u << 20);            // flush the 256 MB Infinity Cache between kernels
    const int numStrips = B * SX * SY;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemsetAsync(trash, rep, 512u << 20, 0);
        hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const float4*>(x), out, n / 4);
        hipMemsetAsync(trash, rep + 1, 512u << 20, 0);
        hipLaunchKernelGGL(k_dword, dim3(2048), dim3(256), 0, 0, x, out, n);
        hipMemsetAsync(trash, rep + 2, 512u << 20, 0);
        hipLaunchKernelGGL(k_strip<false>, dim3(768), dim3(256), 0, 0, x, out, numStrips);
        hipMemsetAsync(trash, rep + 3, 512u << 20, 0);
        hipLaunchKernelGGL(k_strip<true>, dim3(768), dim3(256), 0, 0, x, out, numStrips);
    }
    hipDeviceSynchronize();
    printf("tensor %.2f MB, %d strips\n", n * 4 / 1e6, numStrips);
    return 0;
}
