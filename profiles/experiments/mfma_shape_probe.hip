// Which MFMA shape sustains more FLOP/s on a power-limited MI355X?  Bare MFMA loops on RANDOM data, one wave per SIMD, the SAME 128 x 128 output
// tile per wave for both shapes of a type (so the same operand bytes per FLOP): bf16 32x32x16 (4 x 4 tiles of 16 registers) against 16x16x32 (8 x 8 tiles
// of 4), fp32 32x32x2 against 16x16x4.  Two operand sources: registers (loaded once) and LDS (every fragment re-read by ds_read_b128 each k step, the
// address rotating through 32 KB of random data).  Reports wall TFLOP/s over >= 1 s of back-to-back launches and the in-kernel clock
// (s_memtime / s_memrealtime x 100 MHz).  MI355X_MICROARCH.md, "DVFS give-back" item 7, reports 1.12-1.15 x for the bf16 pair.
//   hipcc -O3 --offload-arch=gfx950 mfma_shape_probe.hip -o mfma_shape_probe.bin && ./mfma_shape_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Stamp { long long cyc, rt; };

// SHAPE 0: bf16 32x32x16, 1: bf16 16x16x32, 2: f32 32x32x2, 3: f32 16x16x4.  LDS: fragments from LDS each step.
template <int SHAPE, bool LDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(const float* __restrict__ src, float* __restrict__ out, Stamp* st, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];            // 32 KB of random data
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) lds[i] = src[i];
    __syncthreads();
    constexpr bool BF = SHAPE < 2, BIG = (SHAPE & 1) == 0;
    constexpr int NF = BIG ? 4 : 8;                                     // fragments per operand: 128 rows / columns of the wave's output tile
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if constexpr (BF) {
        bf16x8 a[NF], b[NF];
        const bf16x8* l8 = reinterpret_cast<const bf16x8*>(lds);        // 2048 units of 16 bytes
        for (int i = 0; i < NF; ++i) { a[i] = l8[(lane + 64 * i) & 2047]; b[i] = l8[(lane + 64 * (i + NF)) & 2047]; }
        if constexpr (BIG) {
            f32x16 acc[4][4] = {};
            for (int it = 0; it < iters; ++it) {
                if constexpr (LDS) {
                    const int o = (it * 8) & 1023;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { a[i] = l8[o + lane + 64 * i]; b[i] = l8[o + 512 + lane + 64 * i]; }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int v = 0; v < 16; ++v) sum += acc[i][j][v];
        } else {
            f32x4 acc[8][8] = {};
            for (int it = 0; it < iters; ++it) {                        // one trip = k 32: the same FLOPs as TWO trips of the 32x32x16 loop
                if constexpr (LDS) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));  // all 64 tiles stay in AGPRs (the allocator otherwise shuffles some through VGPRs)
                }
                if constexpr (LDS) {
                    const int o = (it * 16) & 511;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { a[i] = l8[o + lane + 64 * i]; b[i] = l8[o + 1024 + lane + 64 * i]; }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int v = 0; v < 4; ++v) sum += acc[i][j][v];
        }
    } else {
        // fp32: a b128 read carries the fragments of four consecutive k steps
        f32x4 a[NF], b[NF];
        const f32x4* l4 = reinterpret_cast<const f32x4*>(lds);
        for (int i = 0; i < NF; ++i) { a[i] = l4[(lane + 64 * i) & 2047]; b[i] = l4[(lane + 64 * (i + NF)) & 2047]; }
        if constexpr (BIG) {
            f32x16 acc[4][4] = {};
            for (int it = 0; it < iters; ++it) {
                if constexpr (LDS) {
                    const int o = (it * 8) & 1023;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { a[i] = l4[o + lane + 64 * i]; b[i] = l4[o + 512 + lane + 64 * i]; }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int v = 0; v < 16; ++v) sum += acc[i][j][v];
        } else {
            f32x4 acc[8][8] = {};
            for (int it = 0; it < iters; ++it) {                        // one trip = k 16: the FLOPs of TWO trips of the 32x32x2 loop
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
                if constexpr (LDS) {
                    const int o = (it * 16) & 511;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { a[i] = l4[o + lane + 64 * i]; b[i] = l4[o + 1024 + lane + 64 * i]; }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
            }
            for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int v = 0; v < 4; ++v) sum += acc[i][j][v];
        }
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].rt = r1 - r0; }
}

template <int SHAPE, bool LDS>
static void run(const char* name, const float* src, float* out, Stamp* st, int grid, double flop_per_iter, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // warm up for >= 1 s, then time >= 1 s
    float ms = 0.f;
    int launches = 0;
    for (int phase = 0; phase < 2; ++phase) {
        hipEventRecord(e0);
        launches = 0;
        do {
            for (int k = 0; k < 20; ++k) { hipLaunchKernelGGL((probe<SHAPE, LDS>), dim3(grid), dim3(256), 0, 0, src, out, st, iters); ++launches; }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        } while (ms < 1000.f);
    }
    std::vector<Stamp> h(grid);
    hipMemcpy(h.data(), st, grid * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (auto& s : h) if (s.rt > 0) clk.push_back((double)s.cyc / (double)s.rt * 0.1);
    std::sort(clk.begin(), clk.end());
    const double tf = flop_per_iter * iters * 4.0 * grid * launches / (ms * 1e-3) / 1e12;
    std::vector<double> cyc;
    for (auto& s : h) cyc.push_back((double)s.cyc);
    std::sort(cyc.begin(), cyc.end());
    printf("%-34s %8.1f TFLOP/s   in-kernel clock %.3f GHz (median)   cycles per launch %.0f (median)   %.3f ms per launch\n", name, tf, clk.empty() ? 0.0 : clk[clk.size() / 2],
           cyc[cyc.size() / 2], ms / launches);
    fflush(stdout);
}

int main() {
    int dev = 0, cus = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    float *src, *out; Stamp* st;
    std::vector<float> h(8192);
    srand(1);
    // bf16 pairs viewed as floats: random bits with moderate exponents in both halves; as fp32 they are random numbers in +-[0.5, 2)
    for (auto& v : h) { unsigned hi = 0x3f00u + (rand() & 0xff) + ((rand() & 1) << 15), lo = 0x3f00u + (rand() & 0xff) + ((rand() & 1) << 15); unsigned u = (hi << 16) | lo; v = *reinterpret_cast<float*>(&u); }
    hipMalloc(&src, 8192 * 4); hipMalloc(&out, (size_t)cus * 256 * 4); hipMalloc(&st, cus * sizeof(Stamp));
    hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    printf("%d CUs, one 256-thread workgroup per CU, one wave per SIMD, 128 x 128 output tile per wave\n", cus);
    // FLOPs per loop trip and wave: bf16 32x32x16: 16 MFMAs x 32768; 16x16x32: 64 x 16384 (= 2 trips of the other); f32 32x32x2: 4 x 16 x 4096; 16x16x4: 4 x 64 x 2048
    for (int rep = 0; rep < 2; ++rep) {
        run<0, false>("bf16 32x32x16, registers", src, out, st, cus, 16 * 32768.0, 16000);
        run<1, false>("bf16 16x16x32, registers", src, out, st, cus, 64 * 16384.0, 8000);
        run<0, true>("bf16 32x32x16, LDS fragments", src, out, st, cus, 16 * 32768.0, 16000);
        run<1, true>("bf16 16x16x32, LDS fragments", src, out, st, cus, 64 * 16384.0, 8000);
        run<2, false>("f32 32x32x2, registers", src, out, st, cus, 64 * 4096.0, 2000);
        run<3, false>("f32 16x16x4, registers", src, out, st, cus, 256 * 2048.0, 1000);
        run<2, true>("f32 32x32x2, LDS fragments", src, out, st, cus, 64 * 4096.0, 2000);
        run<3, true>("f32 16x16x4, LDS fragments", src, out, st, cus, 256 * 2048.0, 1000);
    }
    return 0;
}
