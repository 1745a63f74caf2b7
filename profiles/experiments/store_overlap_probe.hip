// How much of an epilogue's global-store time can hide under MFMA work, by kernel structure?  (gfx950)
//   hipcc -O3 --offload-arch=gfx950 store_overlap_probe.hip -o store_overlap_probe.bin && ./store_overlap_probe.bin
// Model of the bf16 E2 forward conv at B = 2048: 8192 workgroup-tiles; per tile each of 4 waves runs MF
// bf16 MFMAs (operands in registers: the main loop stripped of its staging) and then stores its share of a 256-pixel x
// 64-channel bf16 tile (32 KB per workgroup: 8 x 16 B per lane per wave, 64-byte runs at a 128-byte stride, as
// epilogue_store writes them).  Two workgroups per CU (57 KB of LDS each, as the real kernel).
//   mode 0  one workgroup per tile, no stores                      (MFMA floor)
//   mode 1  one workgroup per tile, stores at the end              (the shipped structure)
//   mode 2  persistent workgroups (2 per CU), stores, no loads     (stores drain under the next tile's MFMAs)
//   mode 3  persistent, a load issued AFTER each tile's stores and consumed before the next tile (vmcnt retires in order: the
//           wait for the load is a wait for the stores)
//   mode 4  persistent, the next tile's load issued BEFORE the stores (counted wait passes the stores)
//   mode 5  persistent, 5th wave does the stores from an LDS patch (compute waves never issue a store), load after patch write
//   mode 6  stores only (no MFMAs): the store rate of this pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int LDS_BYTES = 57 * 1024;

template <int MODE>
__global__ __launch_bounds__(MODE == 5 ? 320 : 256) void probe(const f32x4* __restrict__ in, f32x4* __restrict__ out, int tiles, int mf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* patch = reinterpret_cast<f32x4*>(smem);            // mode 5: [8 stores][4 waves][64 lanes] x 16 B = 32 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr bool PERSIST = MODE >= 2 && MODE <= 5;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    f32x4 ld = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 4) ld = in[(size_t)blockIdx.x * 256 + (tid & 255)];
    for (int t = blockIdx.x; t < tiles; t += PERSIST ? gridDim.x : tiles) {
        f32x16 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[k][v] = ld[v & 3];
        if (MODE != 6 && (MODE != 5 || wave < 4)) {
            for (int i = 0; i < mf / 4; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
            }
        }
        // this lane's 8 store units: tile base + (wave*32 + px)*128 B + nb*64 B + c8*16 B, px = (it*64 + lane) >> 2, c8 = lane & 3
        f32x4* tb = out + (size_t)t * 2048;                    // 32 KB per tile = 2048 units
        if (MODE == 4) {                                       // next tile's load BEFORE this tile's stores
            const int tn = t + gridDim.x < tiles ? t + gridDim.x : t;
            ld = in[(size_t)tn * 256 + (tid & 255)];
        }
        if (MODE == 5) {
            if (wave < 4) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    f32x4 v = {acc[s >> 1][0 + (s & 1) * 8], acc[s >> 1][1 + (s & 1) * 8], acc[s >> 1][2 + (s & 1) * 8], acc[s >> 1][3 + (s & 1) * 8]};
                    patch[(s * 4 + wave) * 64 + lane] = v;
                }
            }
            __syncthreads();                                   // patch complete
            if (wave == 4) {
                f32x4 r[32];
#pragma unroll
                for (int s = 0; s < 32; ++s) r[s] = patch[s * 64 + lane];
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    const int st = s >> 2, w = s & 3, tl = st >> 2, nb = (st >> 1) & 1, it = st & 1, idx = it * 64 + lane;
                    tb[tl * 1024 + ((w * 32 + (idx >> 2)) * 8 + nb * 4 + (idx & 3))] = r[s];
                }
            } else {
                const int tn = t + gridDim.x < tiles ? t + gridDim.x : t;
                ld = in[(size_t)tn * 256 + (tid & 255)];       // the compute waves' next-tile load: nothing of theirs is ahead of it
            }
            __syncthreads();                                   // patch consumed
            continue;
        }
        if (MODE != 0) {
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int tl = st >> 2, nb = (st >> 1) & 1, it = st & 1, idx = it * 64 + lane;
                f32x4 v = {acc[st >> 1][0 + (st & 1) * 8], acc[st >> 1][1 + (st & 1) * 8], acc[st >> 1][2 + (st & 1) * 8], acc[st >> 1][3 + (st & 1) * 8]};
                tb[tl * 1024 + ((wave * 32 + (idx >> 2)) * 8 + nb * 4 + (idx & 3))] = v;
            }
        } else if (acc[0][0] == 123.456f) tb[tid] = f32x4{acc[1][0], acc[2][0], acc[3][0], 0.f};       // keeps the MFMAs alive
        if (MODE == 3) {                                       // next tile's load AFTER the stores
            const int tn = t + gridDim.x < tiles ? t + gridDim.x : t;
            ld = in[(size_t)tn * 256 + (tid & 255)];
        }
    }
    if (ld[0] == 777.f) out[tid] = ld;
}

template <int MODE>
static float run(const f32x4* in, f32x4* out, int tiles, int mf, int cus) {
    auto k = probe<MODE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const bool persist = MODE >= 2 && MODE <= 5;
    const int grid = persist ? 2 * cus : tiles, threads = MODE == 5 ? 320 : 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f, sum = 0.f;
    for (int rep = 0; rep < 12; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), LDS_BYTES, 0, in, out, tiles, mf);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    return sum / 10.f * 1e3f;
}

int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 8192, mf = argc > 2 ? atoi(argv[2]) : 200;
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    f32x4 *in, *out;
    hipMalloc(&in, (size_t)tiles * 256 * 16);
    hipMalloc(&out, (size_t)tiles * 2048 * 16);
    hipMemset(in, 0, (size_t)tiles * 256 * 16);
    printf("tiles %d, %d MFMAs per wave and tile, %d CUs; bytes stored per launch %.1f MB; MFMA floor at 2.4 GHz: %.1f us\n", tiles, mf, cus,
           tiles * 32768.0 / 1e6, tiles * 4.0 * mf * 32 / (cus * 4.0) / 2.4e3);
    printf("mode 0 (per-tile WG, no stores)            %8.1f us\n", run<0>(in, out, tiles, mf, cus));
    printf("mode 1 (per-tile WG, stores at the end)    %8.1f us\n", run<1>(in, out, tiles, mf, cus));
    printf("mode 2 (persistent, stores, no loads)      %8.1f us\n", run<2>(in, out, tiles, mf, cus));
    printf("mode 3 (persistent, load AFTER stores)     %8.1f us\n", run<3>(in, out, tiles, mf, cus));
    printf("mode 4 (persistent, load BEFORE stores)    %8.1f us\n", run<4>(in, out, tiles, mf, cus));
    printf("mode 5 (persistent, 5th wave stores)       %8.1f us\n", run<5>(in, out, tiles, mf, cus));
    printf("mode 6 (stores only)                       %8.1f us\n", run<6>(in, out, tiles, mf, cus));
    return 0;
}
