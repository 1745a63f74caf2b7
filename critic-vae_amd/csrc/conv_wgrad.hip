// conv_wgrad.hip — weight gradient of the 5x5 convolutions as a split-K GEMM on the fp32 MFMA.
//
// Replaces the weight half of ATen convolution_backward reached from loss.backward()
// (vae.py:57) for E2..E4 (vae_nets.py:74,79,84) and D0..D3 (:117,121,125,129):
//
//   dW[tap][ci][co] = sum_{b,p} in[b][p+tap-2][ci] * dout[b][p][co]
//
// GEMM view: M = ci, N = co, K = pixels; one accumulator tile per tap.
// A workgroup owns one kernel ROW (5 taps) x 32 input channels x NT output channels and a
// contiguous range of 128-pixel tiles; its 4 waves split each tile's pixels (K) and are summed
// through LDS at the end.  Partial slabs [split][tap][ci][co] are then reduced in fixed order
// by reduce_slabs_kernel -> bitwise reproducible, no atomics.
// The decoder's nearest-2x upsample is folded into the input gather (UP), as in the forward.
#include "common.h"

struct WgradArgs {
    const float* in;
    const float* dout;
    float* slab;        // [S][25*CIN*COUT + COUT]  (weight gradient partials | bias gradient partials)
    int B;
    int numTiles;       // cdiv(B, IMGS) * TILES_PER_IMG
    int tilesPerSplit;
};

// One wave's share of a staged tile: taps W, W+4, .., W+20 over every pixel, plus tap 24 over rows
// rho == W (mod 4) (the 25th tap is split over the 4 waves so all MFMA pipes carry 6.25 taps).
// MFMAs whose two input pixels both fall into the zero padding are skipped (wave-uniform test):
// at 8x8 / 4x4 images that is 23% / 44% of the work; the interleaved tap assignment keeps the
// four waves' remaining work balanced.
template <int H, int W>
__device__ __forceinline__ void wgrad_body(f32x16 (&acc)[7], float& bsum, const float* lds_in, const float* lds_d, int li,
                                           int lh, int ty0, int tx0) {
    using T = Tile<H>;
    constexpr int CS = 32, R = T::IMGS * T::TH;
    for (int rho = 0; rho < R; ++rho) {
        const int img = rho / T::TH, ty = rho % T::TH, gy = ty0 + ty;
        const float* inrow = lds_in + ((img * T::HTH + ty) * T::HTW + lh) * CS + li;
        const float* drow = lds_d + (rho * T::TW + lh) * 32 + li;
        const bool extra = (rho & 3) == W && gy + 2 < H;
        constexpr int KKU = T::TW / 2 < 8 ? T::TW / 2 : 8;      // full unroll of 16 k-steps (H >= 32) spills registers
#pragma unroll KKU
        for (int kk = 0; kk < T::TW / 2; ++kk) {
            const float bv = drow[(2 * kk) * 32];
            const int gx = tx0 + 2 * kk;                     // this k-step covers pixels gx, gx+1
            if (W == 0) bsum += bv;          // column sums of dout = the conv's bias gradient, for free
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int tap = 4 * j + W, r = tap / 5, s = tap % 5;
                if ((unsigned)(gy + r - 2) < (unsigned)H && gx + s - 1 >= 0 && gx + s - 2 < H)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(inrow[(r * T::HTW + s + 2 * kk) * CS], bv, acc[j], 0, 0, 0);
            }
            if (extra && gx + 2 < H)
                acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(inrow[(4 * T::HTW + 4 + 2 * kk) * CS], bv, acc[6], 0, 0, 0);
        }
    }
}

template <int CIN, int COUT, int H, bool UP>
__global__ __launch_bounds__(256, 2) void conv5x5_wgrad_kernel(WgradArgs a) {
    using T = Tile<H>;
    constexpr int CS = 32;
    constexpr int IN_FLOATS = T::HP * CS;                // full 5x5 halo of the tile, 32 channels
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds_in = smem;
    float* lds_d = smem + IN_FLOATS;                     // dout tile [128][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x, ci0 = blockIdx.y * 32, n0 = blockIdx.z * 32;
    constexpr int HS = UP ? H / 2 : H;

    f32x16 acc[7];
    float bsum = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;

    const int t0 = split * a.tilesPerSplit;
    int t1 = t0 + a.tilesPerSplit; if (t1 > a.numTiles) t1 = a.numTiles;

    // software pipeline: tile mt+1 travels global -> registers while the MFMAs of tile mt run
    constexpr int IQ = (T::HP * 8 + 255) / 256;
    f32x4 rin[IQ], rdo[4];
    auto fetch = [&](int mt) {
        const int tileInImg = mt % T::TILES_PER_IMG;
        const int img0 = (mt / T::TILES_PER_IMG) * T::IMGS;
        const int ty0 = (tileInImg / T::TILES_X) * T::TH, tx0 = (tileInImg % T::TILES_X) * T::TW;
#pragma unroll
        for (int i = 0; i < IQ; ++i) {
            const int q = tid + i * 256;
            const int c4 = q & 7, hp = q >> 3;
            const int img = hp / T::HPI, rem = hp % T::HPI;
            const int gy = ty0 + rem / T::HTW - 2, gx = tx0 + rem % T::HTW - 2, ib = img0 + img;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (((T::HP * 8) % 256 == 0 || q < T::HP * 8) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)H && ib < a.B) {
                const int sy = UP ? (gy >> 1) : gy, sx = UP ? (gx >> 1) : gx;
                v = *reinterpret_cast<const f32x4*>(a.in + ((size_t)(ib * HS + sy) * HS + sx) * CIN + ci0 + c4 * 4);
            }
            rin[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * 256;
            const int c4 = q & 7, mm = q >> 3;
            const int im = mm / (T::TH * T::TW), rem = mm % (T::TH * T::TW);
            const int gy = ty0 + rem / T::TW, gx = tx0 + rem % T::TW, ib = img0 + im;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ib < a.B) v = *reinterpret_cast<const f32x4*>(a.dout + ((size_t)(ib * H + gy) * H + gx) * COUT + n0 + c4 * 4);
            rdo[i] = v;
        }
    };
    if (t0 < t1) fetch(t0);
    for (int mt = t0; mt < t1; ++mt) {
        const int tileInImg = mt % T::TILES_PER_IMG;
        const int ty0 = (tileInImg / T::TILES_X) * T::TH, tx0 = (tileInImg % T::TILES_X) * T::TW;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < IQ; ++i) {
            const int q = tid + i * 256;
            if ((T::HP * 8) % 256 == 0 || q < T::HP * 8) *reinterpret_cast<f32x4*>(lds_in + (q >> 3) * CS + (q & 7) * 4) = rin[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * 256;
            *reinterpret_cast<f32x4*>(lds_d + (q >> 3) * 32 + (q & 7) * 4) = rdo[i];
        }
        __syncthreads();
        if (mt + 1 < t1) fetch(mt + 1);
        switch (wave) {
            case 0: wgrad_body<H, 0>(acc, bsum, lds_in, lds_d, li, lh, ty0, tx0); break;
            case 1: wgrad_body<H, 1>(acc, bsum, lds_in, lds_d, li, lh, ty0, tx0); break;
            case 2: wgrad_body<H, 2>(acc, bsum, lds_in, lds_d, li, lh, ty0, tx0); break;
            default: wgrad_body<H, 3>(acc, bsum, lds_in, lds_d, li, lh, ty0, tx0); break;
        }
    }

    float* out = a.slab + (size_t)split * (25 * CIN * COUT + COUT);     // slab row: [25][CIN][COUT] | bias[COUT]
    if (wave == 0 && blockIdx.y == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (lh == 0) out[(size_t)25 * CIN * COUT + n0 + li] = bsum;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int ci = ci0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            out[((size_t)(4 * j + wave) * CIN + ci) * COUT + n0 + li] = acc[j][v];
        }
    // tap 24: sum the four waves' partial tiles through LDS, fixed order
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) smem[((wave - 1) * 16 + v) * 64 + lane] = acc[6][v];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float x = ((acc[6][v] + smem[v * 64 + lane]) + smem[(16 + v) * 64 + lane]) + smem[(32 + v) * 64 + lane];
            const int ci = ci0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            out[((size_t)24 * CIN + ci) * COUT + n0 + li] = x;
        }
    }
}

// dst[i] = sum_s slab[s*stride + i] over s in [s0, s0+cnt), fixed order.  grid (n/4/256, RA): each
// row-chunk y writes its partial to dst + y*n (two passes when S is large and n small, so that
// enough loads are in flight to run at HBM speed).
template <typename AT>
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ dst,
                                                           int64_t n, int S, int64_t stride, int perChunk) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const int s0 = blockIdx.y * perChunk;
    int s1 = s0 + perChunk; if (s1 > S) s1 = S;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int s = s0; s < s1; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)s * stride + i);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    Act<AT>::st4(dst, (size_t)blockIdx.y * n + i, f32x4{acc.x, acc.y, acc.z, acc.w});
}

// One launch for many slabs: a workgroup owns 16 consecutive float4 outputs; its 16 thread groups add
// disjoint row sets (rows g, g+16, ..) and an LDS pass adds the 16 group sums in fixed order.
__global__ __launch_bounds__(256) void reduce_slabs_wide_kernel(const float* __restrict__ slab, float* __restrict__ dst,
                                                                int64_t n4, int S, int64_t stride) {
    __shared__ f32x4 red[16][16];
    const int oi = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int64_t o = (int64_t)blockIdx.x * 16 + oi;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (o < n4) {
#pragma unroll 4
        for (int r = g; r < S; r += 16) acc += *reinterpret_cast<const f32x4*>(slab + (size_t)r * stride + o * 4);
    }
    red[g][oi] = acc;
    __syncthreads();
    if (g == 0 && o < n4) {
#pragma unroll
        for (int k = 1; k < 16; ++k) acc += red[k][oi];
        *reinterpret_cast<f32x4*>(dst + o * 4) = acc;
    }
}

// `slab`, `stride` and n must keep 16-byte alignment (n % 4 == 0).  S <= 32: one thread per output adds
// all rows; larger S: the wide kernel above (`mid` is kept in the signature for callers that size it,
// it is no longer written).
int launch_reduce_slabs(const float* slab, float* dst, int64_t n, int S, int64_t stride, hipStream_t st, float* mid, bool out_bf16) {
    (void)mid;
    if (out_bf16) {          // activation-gradient slabs (split-K dgrad) in precision mode 1: few slabs, bf16 result
        const unsigned gx = (unsigned)((n / 4 + 255) / 256);
        hipLaunchKernelGGL(reduce_slabs_kernel<__bf16>, dim3(gx, 1), dim3(256), 0, st, slab, dst, n, S, stride, S);
        CVAE_CHECK_LAUNCH();
        return 0;
    }
    if (S > 32) {
        const int64_t n4 = n / 4;
        hipLaunchKernelGGL(reduce_slabs_wide_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, slab, dst, n4, S, stride);
        CVAE_CHECK_LAUNCH();
        return 0;
    }
    const unsigned gx = (unsigned)((n / 4 + 255) / 256);
    hipLaunchKernelGGL(reduce_slabs_kernel<float>, dim3(gx, 1), dim3(256), 0, st, slab, dst, n, S, stride, S);
    CVAE_CHECK_LAUNCH();
    return 0;
}

template <int H>
static int wgrad_splits(int B, int blocksPerSplit, int* tilesPerSplit) {
    using T = Tile<H>;
    const int numTiles = cdiv(B, T::IMGS) * T::TILES_PER_IMG;
    int S = cdiv(512, blocksPerSplit);                // aim at ~2 workgroups per CU
    if (S > numTiles) S = numTiles;
    if (S < 1) S = 1;
    const int tps = cdiv(numTiles, S);
    S = cdiv(numTiles, tps);
    *tilesPerSplit = tps;
    return S;
}

template <int CIN, int COUT, int H, bool UP>
static int run_wgrad(int B, const float* in, const float* dout, float* dw, float* dbias, float* ws, hipStream_t st,
                     int64_t* ws_need) {
    using T = Tile<H>;
    int tps;
    const int bps = (CIN / 32) * (COUT / 32);
    const int S = wgrad_splits<H>(B, bps, &tps);
    const int64_t n = (int64_t)25 * CIN * COUT, row = n + COUT;
    if (ws_need) { *ws_need = (int64_t)(S + 16) * row; return 0; }
    WgradArgs a{in, dout, ws, B, cdiv(B, T::IMGS) * T::TILES_PER_IMG, tps};
    constexpr int SMEM = (T::HP * 32 + 128 * 32) * 4;
    auto kern = conv5x5_wgrad_kernel<CIN, COUT, H, UP>;
    static DeviceOnce once;
    { int rc = cvae_grant_lds(once, reinterpret_cast<const void*>(kern), SMEM); if (rc) return rc; }
    cvae_probe_begin(st);
    hipLaunchKernelGGL(kern, dim3(S, CIN / 32, COUT / 32), dim3(256), SMEM, st, a);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    float* mid = ws + (size_t)S * row;
    st = cvae_reduce_stream(st);              // the slab reduction is off the critical path
    if (dbias == dw + n)                      // flat gradient buffer: bias follows its weight -> one reduction
        return launch_reduce_slabs(ws, dw, row, S, row, st, mid);
    int rc = launch_reduce_slabs(ws, dw, n, S, row, st, mid);
    if (rc || !dbias) return rc;
    return launch_reduce_slabs(ws + n, dbias, COUT, S, row, st, nullptr);
}

static int dispatch_wgrad(int layer, int width, int B, const float* in, const float* dout, float* dw, float* dbias,
                          float* ws, hipStream_t st, int64_t* need) {
    if (width == 64) {
        switch (layer) {
            case 1: return run_wgrad<32, 64, 32, false>(B, in, dout, dw, dbias, ws, st, need);
            case 2: return run_wgrad<64, 128, 16, false>(B, in, dout, dw, dbias, ws, st, need);
            case 3: return run_wgrad<128, 256, 8, false>(B, in, dout, dw, dbias, ws, st, need);
            case 4: return run_wgrad<256, 128, 4, false>(B, in, dout, dw, dbias, ws, st, need);
        }
    }
    if (width == 128) {
        switch (layer) {
            case 1: return run_wgrad<32, 64, 64, false>(B, in, dout, dw, dbias, ws, st, need);
            case 2: return run_wgrad<64, 128, 32, false>(B, in, dout, dw, dbias, ws, st, need);
            case 3: return run_wgrad<128, 256, 16, false>(B, in, dout, dw, dbias, ws, st, need);
            case 4: return run_wgrad<256, 128, 8, false>(B, in, dout, dw, dbias, ws, st, need);
        }
    }
    cvae_set_error("conv_wgrad: unsupported layer %d at width %d", layer, width);
    return -2;
}

int64_t wgrad_ws_floats(int layer, int width, int B) {
    int64_t need = 0;
    if (dispatch_wgrad(layer, width, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &need) != 0) return 0;
    return need;
}

int launch_conv_wgrad(int layer, int width, int B, const float* in, const float* dout, float* dw, float* dbias,
                      float* ws, hipStream_t st) {
    return dispatch_wgrad(layer, width, B, in, dout, dw, dbias, ws, st, nullptr);
}

// --------------------------------------------------------------------------------------------
// column sums: dst[c] = sum_rows src[row][c]  (conv bias gradients: db = sum over pixels of dout)
// two fixed-order stages -> reproducible.
// --------------------------------------------------------------------------------------------
static constexpr int CS_BLOCKS = 128;

__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ src, int64_t rows, int C,
                                                     float* __restrict__ part) {
    // thread -> (channel c = tid % C, row lane rl = tid / C); requires C <= 256 and 256 % C == 0
    __shared__ float red[256];
    const int c = threadIdx.x % C, rl = threadIdx.x / C, RL = 256 / C;
    const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = blockIdx.x * per;
    int64_t r1 = r0 + per; if (r1 > rows) r1 = rows;
    float acc = 0.f;
    for (int64_t r = r0 + rl; r < r1; r += RL) acc += src[r * C + c];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < RL; ++k) acc += red[k * C + c];
        part[(size_t)blockIdx.x * C + c] = acc;
    }
}

int64_t colsum_ws_floats(int64_t rows, int C) { (void)rows; return (int64_t)CS_BLOCKS * C + col_reduce_ws_floats(C); }

int launch_colsum(const float* src, int64_t rows, int C, float* dst, float* ws, hipStream_t st) {
    if (C > 256 || 256 % C != 0) { cvae_set_error("colsum: C=%d unsupported", C); return -2; }
    int nblk = CS_BLOCKS;
    if (nblk > rows) nblk = (int)rows;
    hipLaunchKernelGGL(colsum_stage1, dim3(nblk), dim3(256), 0, st, src, rows, C, ws);
    CVAE_CHECK_LAUNCH();
    return launch_col_reduce(ws, nblk, C, C, dst, ws + (size_t)CS_BLOCKS * C, st);
}
