// conv_wgrad.hip — weight gradient of the 5x5 convolutions as a split-K GEMM on the fp32 MFMA.
//
// Replaces the weight half of ATen convolution_backward reached from loss.backward()
// (vae.py:57) for E2..E4 (vae_nets.py:74,79,84) and D0..D3 (:117,121,125,129):
//
//   dW[tap][ci][co] = sum_{b,p} in[b][p+tap-2][ci] * dout[b][p][co]
//
// GEMM view: M = ci (32 per workgroup, one tap per accumulator tile), N = co, K = pixels.
// A workgroup owns one kernel ROW (5 taps) x 32 input channels x NT output channels and a
// contiguous range of 128-pixel tiles; its 4 waves split each tile's pixels (K) and are summed
// through LDS at the end.  Partial slabs [split][tap][ci][co] are then reduced in fixed order
// by reduce_slabs_kernel -> bitwise reproducible, no atomics.
// The decoder's nearest-2x upsample is folded into the input gather (UP), as in the forward.
#include "common.h"

struct WgradArgs {
    const float* in;
    const float* dout;
    float* slab;        // [S][25][CIN][COUT]
    int B;
    int numTiles;       // cdiv(B, IMGS) * TILES_PER_IMG
    int tilesPerSplit;
};

template <int CIN, int COUT, int H, bool UP, int NT>
__global__ __launch_bounds__(256) void conv5x5_wgrad_kernel(WgradArgs a) {
    using T = Tile<H>;
    constexpr int NB = NT / 32;
    constexpr int CS = 32;                               // LDS pixel stride of the input rows
    constexpr int IN_PIX = T::IMGS * T::TH * T::HTW;     // TH rows (one kernel row) x (TW+4) cols
    constexpr int IN_FLOATS = IN_PIX * CS;
    constexpr int D_FLOATS = 128 * NT;
    constexpr int RED_FLOATS = 2 * 5 * NB * 1024;        // two waves' accumulators
    constexpr int SMEM = (IN_FLOATS + D_FLOATS) > RED_FLOATS ? (IN_FLOATS + D_FLOATS) : RED_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds_in = smem;
    float* lds_d = smem + IN_FLOATS;
    static_assert(IN_FLOATS % 4 == 0, "alignment");
    (void)SMEM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x;
    const int r = blockIdx.y / (CIN / 32), ci0 = (blockIdx.y % (CIN / 32)) * 32;
    const int n0 = blockIdx.z * NT;
    constexpr int HS = UP ? H / 2 : H;

    f32x16 acc[5][NB];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[s][nb][v] = 0.f;

    const int t0 = split * a.tilesPerSplit;
    int t1 = t0 + a.tilesPerSplit; if (t1 > a.numTiles) t1 = a.numTiles;

    for (int mt = t0; mt < t1; ++mt) {
        const int tileInImg = mt % T::TILES_PER_IMG;
        const int img0 = (mt / T::TILES_PER_IMG) * T::IMGS;
        const int ty0 = (tileInImg / T::TILES_X) * T::TH, tx0 = (tileInImg % T::TILES_X) * T::TW;
        __syncthreads();
        // input rows ty0+r-2 .. (+TH), cols tx0-2 .. tx0+TW+1, channels ci0..ci0+31 -> [pixel][32]
        for (int q = tid; q < IN_PIX * 8; q += 256) {
            const int c4 = q & 7, hp = q >> 3;
            const int img = hp / (T::TH * T::HTW), rem = hp % (T::TH * T::HTW);
            const int hy = rem / T::HTW, hx = rem % T::HTW;
            const int gy = ty0 + hy + r - 2, gx = tx0 + hx - 2, ib = img0 + img;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)H && ib < a.B) {
                const int sy = UP ? (gy >> 1) : gy, sx = UP ? (gx >> 1) : gx;
                v = *reinterpret_cast<const float4*>(
                    a.in + ((size_t)(ib * HS + sy) * HS + sx) * CIN + ci0 + c4 * 4);
            }
            *reinterpret_cast<float4*>(lds_in + hp * CS + c4 * 4) = v;
        }
        // dout tile [128 pixels][NT]
        for (int q = tid; q < 128 * NT / 4; q += 256) {
            const int c4 = q % (NT / 4), mm = q / (NT / 4);
            const int im = mm / (T::TH * T::TW), rem = mm % (T::TH * T::TW);
            const int gy = ty0 + rem / T::TW, gx = tx0 + rem % T::TW, ib = img0 + im;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ib < a.B)
                v = *reinterpret_cast<const float4*>(
                    a.dout + ((size_t)(ib * H + gy) * H + gx) * COUT + n0 + c4 * 4);
            *reinterpret_cast<float4*>(lds_d + mm * NT + c4 * 4) = v;
        }
        __syncthreads();
        // wave w contracts pixels [32w, 32w+32) of the tile, two per MFMA (k = lh)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int mm = wave * 32 + 2 * kk + lh;                        // this lane's pixel
            const int im = mm / (T::TH * T::TW), rem = mm % (T::TH * T::TW);
            const int hp = (im * T::TH + rem / T::TW) * T::HTW + rem % T::TW;   // tap s adds +s
            float bv[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = lds_d[mm * NT + nb * 32 + li];
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const float av = lds_in[(hp + s) * CS + li];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[s][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[nb], acc[s][nb], 0, 0, 0);
            }
        }
    }

    // ---- sum the 4 waves' accumulators through LDS (tree: 2,3 -> 0,1 ; 1 -> 0) ----
    __syncthreads();
    float* red = smem;                                   // [2][5*NB][16][64]
    if (wave >= 2) {
        float* d = red + (wave - 2) * (5 * NB * 1024);
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int v = 0; v < 16; ++v) d[((s * NB + nb) * 16 + v) * 64 + lane] = acc[s][nb][v];
    }
    __syncthreads();
    if (wave < 2) {
        const float* d = red + wave * (5 * NB * 1024);
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[s][nb][v] += d[((s * NB + nb) * 16 + v) * 64 + lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int v = 0; v < 16; ++v) red[((s * NB + nb) * 16 + v) * 64 + lane] = acc[s][nb][v];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = a.slab + (size_t)split * 25 * CIN * COUT;
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float x = acc[s][nb][v] + red[((s * NB + nb) * 16 + v) * 64 + lane];
                    const int ci = ci0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                    out[((size_t)(r * 5 + s) * CIN + ci) * COUT + n0 + nb * 32 + li] = x;
                }
    }
}

// dst[i] = sum_s slab[s][i], fixed order.  n is a multiple of 4 for every conv weight here.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ dst,
                                                           int64_t n, int S, int64_t stride) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 acc = *reinterpret_cast<const float4*>(slab + i);
        for (int s = 1; s < S; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)s * stride + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(dst + i) = acc;
    } else {
        for (int64_t k = i; k < n; ++k) {
            float acc = slab[k];
            for (int s = 1; s < S; ++s) acc += slab[(size_t)s * stride + k];
            dst[k] = acc;
        }
    }
}

// `stride` (floats between consecutive slabs) and `slab` must keep 16-byte alignment.
int launch_reduce_slabs(const float* slab, float* dst, int64_t n, int S, int64_t stride, hipStream_t st) {
    const int64_t threads = (n + 3) / 4;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                       slab, dst, n, S, stride);
    CVAE_CHECK_LAUNCH();
    return 0;
}

template <int H>
static int wgrad_splits(int B, int blocksPerSplit, int* tilesPerSplit) {
    using T = Tile<H>;
    const int numTiles = cdiv(B, T::IMGS) * T::TILES_PER_IMG;
    int S = cdiv(1536, blocksPerSplit);               // aim at ~6 workgroups per CU
    if (S > numTiles) S = numTiles;
    if (S < 1) S = 1;
    const int tps = cdiv(numTiles, S);
    S = cdiv(numTiles, tps);
    *tilesPerSplit = tps;
    return S;
}

template <int CIN, int COUT, int H, bool UP, int NT>
static int run_wgrad(int B, const float* in, const float* dout, float* dw, float* ws, hipStream_t st,
                     int64_t* ws_need) {
    using T = Tile<H>;
    constexpr int NB = NT / 32;
    int tps;
    const int bps = 5 * (CIN / 32) * (COUT / NT);
    const int S = wgrad_splits<H>(B, bps, &tps);
    const int64_t n = (int64_t)25 * CIN * COUT;
    if (ws_need) { *ws_need = (int64_t)S * n; return 0; }
    WgradArgs a{in, dout, ws, B, cdiv(B, T::IMGS) * T::TILES_PER_IMG, tps};
    constexpr int IN_FLOATS = T::IMGS * T::TH * T::HTW * 32;
    constexpr int RED_FLOATS = 2 * 5 * NB * 1024;
    constexpr int SMEM = ((IN_FLOATS + 128 * NT) > RED_FLOATS ? (IN_FLOATS + 128 * NT) : RED_FLOATS) * 4;
    auto kern = conv5x5_wgrad_kernel<CIN, COUT, H, UP, NT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(S, 5 * (CIN / 32), COUT / NT), dim3(256), SMEM, st, a);
    CVAE_CHECK_LAUNCH();
    return launch_reduce_slabs(ws, dw, n, S, n, st);
}

static int dispatch_wgrad(int layer, int width, int B, const float* in, const float* dout, float* dw,
                          float* ws, hipStream_t st, int64_t* need) {
    if (width == 64) {
        switch (layer) {
            case 1: return run_wgrad<32, 64, 32, false, 32>(B, in, dout, dw, ws, st, need);
            case 2: return run_wgrad<64, 128, 16, false, 32>(B, in, dout, dw, ws, st, need);
            case 3: return run_wgrad<128, 256, 8, false, 32>(B, in, dout, dw, ws, st, need);
            case 4: return run_wgrad<256, 128, 4, false, 32>(B, in, dout, dw, ws, st, need);
            case 5: return run_wgrad<128, 64, 8, true, 32>(B, in, dout, dw, ws, st, need);
            case 6: return run_wgrad<64, 32, 16, true, 32>(B, in, dout, dw, ws, st, need);
            case 7: return run_wgrad<32, 32, 32, true, 32>(B, in, dout, dw, ws, st, need);
        }
    }
    cvae_set_error("conv_wgrad: unsupported layer %d at width %d", layer, width);
    return -2;
}

int64_t wgrad_ws_floats(int layer, int width, int B) {
    int64_t need = 0;
    if (dispatch_wgrad(layer, width, B, nullptr, nullptr, nullptr, nullptr, nullptr, &need) != 0) return 0;
    return need;
}

int launch_conv_wgrad(int layer, int width, int B, const float* in, const float* dout, float* dw,
                      float* ws, hipStream_t st) {
    return dispatch_wgrad(layer, width, B, in, dout, dw, ws, st, nullptr);
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
