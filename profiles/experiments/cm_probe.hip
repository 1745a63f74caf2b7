// Probe of the channel-major epilogue helpers (conv_epilogue.h): cm_pack_units and half_wave_colsum16 on exact integer data.
//   hipcc -O3 --offload-arch=gfx950 -I../../critic-vae_amd/csrc cm_probe.hip -o cm_probe.bin && ./cm_probe.bin
#include "conv_epilogue.h"
#include <cstdio>
#include <vector>
__global__ void probe(float* units, float* sums) {
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    f32x16 acc;
    float x[16];
    for (int v = 0; v < 16; ++v) {
        const int ch = (v & 3) + 8 * (v >> 2) + 4 * lh;           // channel of element v
        acc[v] = (float)(li * 32 + ch);                            // value encodes (pixel, channel); exact in bf16? up to 1023: no -> use small
        x[v] = (float)(li + 1) * (float)(ch + 1);                  // column sums: (ch+1) * sum(li+1) = (ch+1) * 528
    }
    for (int v = 0; v < 16; ++v) acc[v] = (float)((li & 7) * 32 + ((v & 3) + 8 * (v >> 2) + 4 * lh));   // <= 255: exact in bf16
    bf16x8 u[2];
    cm_pack_units(acc, u);
    for (int k = 0; k < 2; ++k) for (int e = 0; e < 8; ++e) units[(lane * 2 + k) * 8 + e] = (float)u[k][e];
    sums[lane] = half_wave_colsum16(x);
    for (int pass = 0; pass < 2; ++pass) {       // which lanes does each lane's result cover?  x = one bit per lane (16 lanes per pass)
        for (int v = 0; v < 16; ++v) x[v] = ((li >> 4) == pass) ? (float)(1 << (li & 15)) : 0.f;
        sums[64 + pass * 64 + lane] = half_wave_colsum16(x);
    }
}
int main() {
    float *du, *ds; hipMalloc(&du, 64 * 16 * 4); hipMalloc(&ds, 192 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, du, ds);
    std::vector<float> u(64 * 16), s(192);
    hipMemcpy(u.data(), du, 64 * 16 * 4, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, 192 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int li = lane & 31, lh = lane >> 5;
        for (int k = 0; k < 2; ++k) for (int e = 0; e < 8; ++e) {
            const float want = (float)((li & 7) * 32 + 16 * k + 8 * lh + e);
            if (u[(lane * 2 + k) * 8 + e] != want) { if (bad < 12) printf("lane %d unit %d elem %d: got %g want %g\n", lane, k, e, u[(lane * 2 + k) * 8 + e], want); ++bad; }
        }
        const int e16 = li >> 1, ch = (e16 & 3) + 8 * (e16 >> 2) + 4 * lh;
        const float want = (float)(ch + 1) * 528.f;
        if (s[lane] != want) { if (bad < 24) printf("lane %d colsum: got %g want %g (channel %d)\n", lane, s[lane], want, ch); ++bad; }
    }
    for (int lane = 0; lane < 64; lane += 1) printf("lane %2d covers lanes mask lo %04x hi %04x\n", lane, (unsigned)s[64 + lane], (unsigned)s[128 + lane]);
    printf("cm probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    return bad != 0;
}
