// reduce.hip — fixed-order column sums of small row-major partial matrices [R][W] -> [W].
//
// Every cross-workgroup reduction of the step (BatchNorm backward sums, conv bias gradients,
// split-K slabs of the thin convs) ends here.  One or two launches, each thread reads a handful
// of rows: no serial loops over hundreds of partials, no atomics, bitwise reproducible.
#include "common.h"

// mid[ra][w] = sum over rows [ra*rowsPerBlk, +rowsPerBlk) of in[r*stride + w]; grid (cdiv(W,32), RA)
__global__ __launch_bounds__(256) void rows_sum_kernel(const float* __restrict__ in, int R, int W, int64_t stride,
                                                       float* __restrict__ out, int rowsPerBlk) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, w = blockIdx.x * 32 + cl;
    const int r0 = blockIdx.y * rowsPerBlk;
    int r1 = r0 + rowsPerBlk; if (r1 > R) r1 = R;
    float acc = 0.f;
    if (w < W) {
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += 8) acc += in[(size_t)r * stride + w];
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && w < W) {
#pragma unroll
        for (int k = 1; k < 8; ++k) acc += red[k][cl];
        out[(size_t)blockIdx.y * W + w] = acc;
    }
}

// out[w] = sum_r in[r*stride + w] in ONE launch for 64 < R <= 2048: 32 row lanes x 32 columns per workgroup, every lane's
// (at most 64) loads are independent, then a fixed-order sum over the row lanes.  One launch floor (~4.7 us) less than
// rows_sum_kernel twice.
__global__ __launch_bounds__(1024) void rows_sum_1024_kernel(const float* __restrict__ in, int R, int W, int64_t stride,
                                                             float* __restrict__ out) {
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, w = blockIdx.x * 32 + cl;
    float acc = 0.f;
    if (w < W) {
#pragma unroll 8
        for (int r = rl; r < R; r += 32) acc += in[(size_t)r * stride + w];
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && w < W) {
#pragma unroll
        for (int k = 1; k < 32; ++k) acc += red[k][cl];
        out[w] = acc;
    }
}

int64_t col_reduce_ws_floats(int W) { return (int64_t)32 * W; }

// First half of launch_col_reduce for callers whose next kernel can add up <= 64 rows itself:
// returns the rows still to be summed through (*rows_out, *R_out, *stride_out).
int launch_col_reduce_partial(const float* in, int R, int W, int64_t stride, float* ws, hipStream_t st,
                              const float** rows_out, int* R_out, int64_t* stride_out) {
    if (R > 64) {
        const int RA = 32, rpb = cdiv(R, RA);
        hipLaunchKernelGGL(rows_sum_kernel, dim3(cdiv(W, 32), cdiv(R, rpb)), dim3(256), 0, st, in, R, W, stride, ws, rpb);
        CVAE_CHECK_LAUNCH();
        in = ws; R = cdiv(R, rpb); stride = W;
    }
    *rows_out = in; *R_out = R; *stride_out = stride;
    return 0;
}

// out[w] = sum_r in[r*stride + w].  ws: col_reduce_ws_floats(W) floats (used when R > 64).
int launch_col_reduce(const float* in, int R, int W, int64_t stride, float* out, float* ws, hipStream_t st) {
    if (R > 64 && R <= 2048) {
        hipLaunchKernelGGL(rows_sum_1024_kernel, dim3(cdiv(W, 32)), dim3(1024), 0, st, in, R, W, stride, out);
        CVAE_CHECK_LAUNCH();
        return 0;
    }
    if (R > 64) {
        const int RA = 32, rpb = cdiv(R, RA);
        hipLaunchKernelGGL(rows_sum_kernel, dim3(cdiv(W, 32), cdiv(R, rpb)), dim3(256), 0, st, in, R, W, stride, ws, rpb);
        CVAE_CHECK_LAUNCH();
        in = ws; R = cdiv(R, rpb); stride = W;
    }
    hipLaunchKernelGGL(rows_sum_kernel, dim3(cdiv(W, 32), 1), dim3(256), 0, st, in, R, W, stride, out, R);
    CVAE_CHECK_LAUNCH();
    return 0;
}
