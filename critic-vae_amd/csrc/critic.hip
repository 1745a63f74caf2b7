// critic.hip — the frozen critic CNN that produces `preds` in the training loop, and the uint8
// frame pre-processing, as two small fused kernels.
//
// Replaces Critic.evaluate / Critic.forward (critic_net.py:5-69, eval mode: Dropout = identity):
//   Conv(3,8,3,p1) ReLU Pool2 -> Conv(8,8,3,p1) ReLU Pool2 -> Conv(8,8,3,p1) ReLU Pool2 ->
//   Conv(8,16,3,p1) ReLU Pool2 -> Conv(16,32,4) ReLU -> Flatten -> Linear(32,32) ReLU ->
//   Linear(32,1) Sigmoid                                  (3.4 MFLOP / image, 11 873 parameters)
// called once per step at vae.py:50, and adjust_values + HWC->CHW of preprocess_observation
// (vae_utility.py:324-343).  One workgroup per image, every activation lives in LDS; weights are
// read in the reference's own state_dict order / OIHW layout (the critic is never trained here).
#include "common.h"

// float offsets into the flat critic parameter block (reference state_dict order)
static constexpr int CW1 = 0, CB1 = 216, CW2 = 224, CB2 = 800, CW3 = 808, CB3 = 1384, CW4 = 1392, CB4 = 2544,
                     CW5 = 2560, CB5 = 10752, CF1W = 10784, CF1B = 11808, CF2W = 11840, CF2B = 11872;
static constexpr int CRITIC_PARAMS = 11873;

// 3x3/pad-1 conv + ReLU + 2x2 max-pool from zero-bordered LDS planes in[CI][S+2][S+2] to zero-bordered
// LDS planes out[CO][S/2+2][S/2+2] (or un-bordered when BORDER_OUT == 0)
template <int CI, int CO, int S, int BORDER_OUT>
__device__ __forceinline__ void conv3_relu_pool(const float* in, float* out, const float* __restrict__ w,
                                                const float* __restrict__ b) {
    constexpr int SO = S / 2, PI = (S + 2) * (S + 2), WO = SO + 2 * BORDER_OUT, PO = WO * WO;
    for (int q = threadIdx.x; q < CO * SO * SO; q += 256) {
        const int co = q % CO, p = q / CO, py = p / SO, px = p % SO;
        float acc[4] = {b[co], b[co], b[co], b[co]};
        for (int ci = 0; ci < CI; ++ci) {
            const float* ip = in + ci * PI + (2 * py) * (S + 2) + 2 * px;      // top-left of the 4x4 input patch
            const float* wp = w + (co * CI + ci) * 9;
            float v[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) v[r][c] = ip[r * (S + 2) + c];
#pragma unroll
            for (int kr = 0; kr < 3; ++kr)
#pragma unroll
                for (int kc = 0; kc < 3; ++kc) {
                    const float wv = wp[kr * 3 + kc];
                    acc[0] = fmaf(wv, v[kr][kc], acc[0]); acc[1] = fmaf(wv, v[kr][kc + 1], acc[1]);
                    acc[2] = fmaf(wv, v[kr + 1][kc], acc[2]); acc[3] = fmaf(wv, v[kr + 1][kc + 1], acc[3]);
                }
        }
        const float m = fmaxf(fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])), 0.f);   // ReLU then max == max then ReLU
        out[co * PO + (py + BORDER_OUT) * WO + px + BORDER_OUT] = m;
    }
}

__global__ __launch_bounds__(256) void critic_fwd_kernel(const float* __restrict__ x, const float* __restrict__ cp,
                                                         float* __restrict__ pred) {
    constexpr int X_FLOATS = 3 * 66 * 66, A1 = 8 * 34 * 34, A2 = 8 * 18 * 18, A3 = 8 * 10 * 10, A4 = 16 * 4 * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lx = smem;
    float* a1 = lx + X_FLOATS;
    float* a2 = a1 + A1;
    float* a3 = a2 + A2;
    float* a4 = a3 + A3;
    float* a5 = a4 + A4;          // 32
    float* f1 = a5 + 32;          // 32
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int q = tid; q < X_FLOATS + A1 + A2 + A3; q += 256) smem[q] = 0.f;      // zero borders
    __syncthreads();
    const float* xb = x + (size_t)b * 3 * 64 * 64;
    for (int q = tid; q < 3 * 64 * 16; q += 256) {
        const int c4 = q & 15, row = (q >> 4) & 63, c = q >> 10;
        const float4 v = *reinterpret_cast<const float4*>(xb + (c * 64 + row) * 64 + c4 * 4);
        float* d = lx + c * 66 * 66 + (row + 1) * 66 + c4 * 4 + 1;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    conv3_relu_pool<3, 8, 64, 1>(lx, a1, cp + CW1, cp + CB1);
    __syncthreads();
    conv3_relu_pool<8, 8, 32, 1>(a1, a2, cp + CW2, cp + CB2);
    __syncthreads();
    conv3_relu_pool<8, 8, 16, 1>(a2, a3, cp + CW3, cp + CB3);
    __syncthreads();
    conv3_relu_pool<8, 16, 8, 0>(a3, a4, cp + CW4, cp + CB4);
    __syncthreads();
    // Conv(16,32,4) on the 4x4 map = a 256-long dot product per output; 8 lanes per output
    {
        const int o = tid >> 3, part = tid & 7;
        float acc = 0.f;
        for (int k = part; k < 256; k += 8) acc = fmaf(cp[CW5 + o * 256 + k], a4[k], acc);
        acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
        if (part == 0) a5[o] = fmaxf(acc + cp[CB5 + o], 0.f);
    }
    __syncthreads();
    if (tid < 32) {
        float acc = cp[CF1B + tid];
        for (int k = 0; k < 32; ++k) acc = fmaf(cp[CF1W + tid * 32 + k], a5[k], acc);
        f1[tid] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    if (tid == 0) {
        float acc = cp[CF2B];
        for (int k = 0; k < 32; ++k) acc = fmaf(cp[CF2W + k], f1[k], acc);
        pred[b] = 1.0f / (1.0f + expf(-acc));
    }
}

// x[b][c][y][x] = u8[b][y][x][c] / 255   (adjust_values + transpose(2,0,1), vae_utility.py:324-343)
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const uint8_t* __restrict__ u8, float* __restrict__ x,
                                                            int64_t npix_total, int hw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // pixel index over B*H*W
    if (i >= npix_total) return;
    const int64_t b = i / hw, p = i % hw;
    const uint8_t* s = u8 + i * 3;
    float* d = x + b * 3 * hw + p;
    d[0] = (float)s[0] / 255.0f; d[hw] = (float)s[1] / 255.0f; d[2 * (int64_t)hw] = (float)s[2] / 255.0f;
}

int launch_critic_fwd(int width, int B, const float* x, const float* critic_params, float* pred, hipStream_t st) {
    if (width != 64) { cvae_set_error("critic: width %d unsupported (the reference critic is 64x64 only)", width); return -2; }
    constexpr int SMEM = (3 * 66 * 66 + 8 * 34 * 34 + 8 * 18 * 18 + 8 * 10 * 10 + 16 * 4 * 4 + 64) * 4;
    static DeviceOnce once;
    { int rc = cvae_grant_lds(once, reinterpret_cast<const void*>(critic_fwd_kernel), SMEM); if (rc) return rc; }
    hipLaunchKernelGGL(critic_fwd_kernel, dim3(B), dim3(256), SMEM, st, x, critic_params, pred);
    CVAE_CHECK_LAUNCH();
    return 0;
}

int launch_preprocess_u8(int width, int B, const uint8_t* u8, float* x, hipStream_t st) {
    const int64_t n = (int64_t)B * width * width;
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, u8, x, n, width * width);
    CVAE_CHECK_LAUNCH();
    return 0;
}

// diff[b][y][x] = grey(|a - b|) with the reference's luma weights (get_diff_image, vae_utility.py:256-277)
__global__ __launch_bounds__(256) void diff_grey_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ diff, int64_t npix_total, int hw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix_total) return;
    const int64_t img = i / hw, p = i % hw, o = img * 3 * hw + p;
    const float r = fabsf(b[o] - a[o]), g = fabsf(b[o + hw] - a[o + hw]), bl = fabsf(b[o + 2 * (int64_t)hw] - a[o + 2 * (int64_t)hw]);
    diff[i] = r * 0.2989f + g * 0.5870f + bl * 0.1140f;
}

int launch_diff_grey(int width, int B, const float* a, const float* b, float* diff, hipStream_t st) {
    const int64_t n = (int64_t)B * width * width;
    hipLaunchKernelGGL(diff_grey_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, diff, n, width * width);
    CVAE_CHECK_LAUNCH();
    return 0;
}

int critic_param_count() { return CRITIC_PARAMS; }
