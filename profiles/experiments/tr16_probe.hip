// Probe of ds_read_b64_tr_b16 (gfx950): checks, with exact integer data, the lane -> element map the wgrad
// kernels rely on:  within each group of 16 consecutive lanes, lane 4q+p supplies the address of row q,
// columns 4p..4p+3 of a 4-row x 16-column block of 16-bit elements; lane i of the group receives column i of
// the 4 rows, row q in element q.   hipcc --offload-arch=gfx950 tr16_probe.hip -o tr16_probe && ./tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short short4v __attribute__((ext_vector_type(4)));
constexpr int ROWS = 64, STRIDE = 40;      // 40 elements = 80-byte rows (not a power of two on purpose)
__global__ void probe(short4v* out) {
    __shared__ __attribute__((aligned(16))) short lds[ROWS * STRIDE];
    for (int q = threadIdx.x; q < ROWS * STRIDE; q += 64) lds[q] = (short)((q / STRIDE) * 256 + q % STRIDE);
    __syncthreads();
    const int L = threadIdx.x, g = L >> 4, i = L & 15, q = i >> 2, p = i & 3;
    const int r0 = 8 * g + 3, c0 = 16 * (g & 1);          // a different block per group
    const short* addr = lds + (r0 + q) * STRIDE + c0 + 4 * p;
    out[L] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)addr);
}
int main() {
    short4v* d; hipMalloc(&d, 64 * sizeof(short4v));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    std::vector<short4v> h(64);
    hipMemcpy(h.data(), d, 64 * sizeof(short4v), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int L = 0; L < 64; ++L) {
        const int g = L >> 4, i = L & 15, r0 = 8 * g + 3, c0 = 16 * (g & 1);
        for (int e = 0; e < 4; ++e) {
            const int want = (r0 + e) * 256 + c0 + i;
            if (h[L][e] != want) { if (bad < 8) printf("lane %d elem %d: got row %d col %d, want row %d col %d\n", L, e, h[L][e] / 256, h[L][e] % 256, want / 256, want % 256); ++bad; }
        }
    }
    printf("tr16 probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    return bad != 0;
}
