// adam.hip — fused flat-buffer Adam: torch.optim.Adam(lr) defaults (vae.py:36,58) over the whole
// parameter buffer in one HBM-bound pass (4 reads + 3 writes per element), gradient scale fused
// (1/world_size after the summing all-reduce).  Mirrors torch's single-tensor formulas:
//   m.lerp_(g, 1-b1);  v = b2*v + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.h"
#include <math.h>

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                   float omb1, float b2, float omb2, float step_size,
                                                   float sqrt_bc2, float eps, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gp[e] * gscale;
            mp[e] = mp[e] + (gr - mp[e]) * omb1;
            vp[e] = vp[e] * b2 + omb2 * gr * gr;
            const float denom = sqrtf(vp[e]) / sqrt_bc2 + eps;
            pp[e] = pp[e] - step_size * (mp[e] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float b1, float b2,
                float eps, float gscale, hipStream_t st) {
    if (n % 4 != 0) { cvae_set_error("adam: n=%lld must be a multiple of 4", (long long)n); return -1; }
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, n4, 1.0f - b1, b2, 1.0f - b2,
                       (float)((double)lr / bc1), (float)sqrt(bc2), eps, gscale);
    CVAE_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void zero_kernel(float4* __restrict__ p, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
int launch_zero(float* p, int64_t n, hipStream_t st) {
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<float4*>(p), n4);
    CVAE_CHECK_LAUNCH();
    return 0;
}

// Alignment padding between the tensors of the flat gradient buffer (<= 63 floats each): written as 0 so a
// caller may hand cvae_backward an uninitialised buffer (phase bit 3).  One workgroup per tensor gap.
__global__ __launch_bounds__(64) void zero_gaps_kernel(float* __restrict__ g, PadGaps gaps) {
    const int64_t o = gaps.off[blockIdx.x];
    if ((int)threadIdx.x < gaps.len[blockIdx.x]) g[o + threadIdx.x] = 0.f;
}
int launch_zero_gaps(float* grads, const PadGaps& gaps, hipStream_t st) {
    if (gaps.n == 0) return 0;
    hipLaunchKernelGGL(zero_gaps_kernel, dim3(gaps.n), dim3(64), 0, st, grads, gaps);
    CVAE_CHECK_LAUNCH();
    return 0;
}

// out_k = in_k * g[0] for the three loss gradients in ONE launch (the chain-rule factor autograd hands
// total_loss.backward(): a device scalar, so no host read)
__global__ __launch_bounds__(256) void scale3_kernel(Scale3 a) {
    const float g = a.g[0];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int64_t j = i; j < a.n[k]; j += (int64_t)gridDim.x * 256) a.dst[k][j] = a.src[k][j] * g;
}
int launch_scale3(const Scale3& a, hipStream_t st) {
    int64_t nmax = a.n[0] > a.n[1] ? a.n[0] : a.n[1];
    if (a.n[2] > nmax) nmax = a.n[2];
    int64_t blocks = (nmax + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale3_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    CVAE_CHECK_LAUNCH();
    return 0;
}

// fp32 <-> bf16 copies of a gradient range for the optional bf16 all-reduce (5.17 MB instead of 10.34 MB on
// the wire); RNE on the way down, exact on the way up.  n is a multiple of 64 (bucket ranges are).
__global__ __launch_bounds__(256) void grads_pack_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * 4);
    bf16x4 o;
    o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
    *reinterpret_cast<bf16x4*>(dst + i * 4) = o;
}
__global__ __launch_bounds__(256) void grads_unpack_bf16_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(src + i * 4);
    *reinterpret_cast<f32x4*>(dst + i * 4) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
int launch_grads_bf16(const float* src_f32, void* bf16_buf, float* dst_f32, int64_t n, hipStream_t st) {
    if (n % 4 != 0) { cvae_set_error("grads bf16 copy: n = %lld is not a multiple of 4", (long long)n); return -2; }
    if (n == 0) return 0;                                    // an empty range is a no-op, not a zero-block launch
    const int64_t n4 = n / 4;
    const unsigned blocks = (unsigned)((n4 + 255) / 256);
    if (src_f32) hipLaunchKernelGGL(grads_pack_bf16_kernel, dim3(blocks), dim3(256), 0, st, src_f32, reinterpret_cast<__bf16*>(bf16_buf), n4);
    else hipLaunchKernelGGL(grads_unpack_bf16_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const __bf16*>(bf16_buf), dst_f32, n4);
    CVAE_CHECK_LAUNCH();
    return 0;
}
