// conv_up.hip — the decoder's Upsample(2) -> Conv2d(5x5) blocks D1..D3 (vae_nets.py:119-131)
// computed at the LOW resolution with phase-collapsed 3x3 kernels: 36 taps instead of 100.
//
// For output pixel (2y+py, 2x+px) of a 5x5/pad-2 conv over a nearest-2x upsampled image, kernel
// rows r=0..4 read source rows y + {-1,-1,0,0,1} (py=0) or y + {-1,0,0,1,1} (py=1): each output
// PHASE (py,px) is a 3x3 conv over the stored low-resolution tensor with pre-summed weights
//     Wc[py,px][a][b] = sum_{r in R(py,a)} sum_{s in R(px,b)} W[r][s],
//     R(0,.) = {0,1},{2,3},{4}    R(1,.) = {0},{1,2},{3,4}.
// forward : out[2y+py][2x+px] = sum_{a,b,ci} in[y+a-1][x+b-1][ci] * Wc[p][a][b][ci][co]
// dgrad   : din[y][x][ci]     = relu'(.) * sum_{p,a,b,co} dout[2(y-a+1)+py][2(x-b+1)+px][co] * Wc[p][a][b][ci][co]
//           (lands directly at the low resolution: no 2x2 pool-sum epilogue)
// wgrad   : dWc[p][a][b][ci][co] = sum_{y,x} in[y+a-1][x+b-1][ci] * dout[2y+py][2x+px][co];
//           dW[r][s] = sum_{py,px} dWc[py,px][a(py,r)][b(px,s)]
// i.e. 0.36x the MFMA work of the direct form at identical maths up to fp32 rounding of the
// pre-summed weights (~1e-7 relative).  All three are implicit GEMMs on v_mfma_f32_32x32x2_f32
// with the same staging scheme as conv_mfma.hip / conv_wgrad.hip.
#include "common.h"

static constexpr int KC = 16, KCP = 17;

template <int HS>        // low-resolution tile of 128 pixels with a +-1 halo
struct UpTile {
    static constexpr int TW = HS < 32 ? HS : 32;
    static constexpr int TH = (128 / TW) < HS ? (128 / TW) : HS;
    static constexpr int IMGS = 128 / (TW * TH);
    static constexpr int HTW = TW + 2, HTH = TH + 2, HPI = HTW * HTH, HP = IMGS * HPI;
    static constexpr int PS = ((HP + 5) / 8) * 8 + 2;
    static constexpr int TILES_X = HS / TW, TILES_Y = HS / TH, TILES_PER_IMG = TILES_X * TILES_Y;
};

// ---------------------------------------------------------------------------------------------
// weight transforms (tiny, once per step)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int phase_lo(int ph, int a) { return ph == 0 ? (a == 0 ? 0 : a == 1 ? 2 : 4) : (a == 0 ? 0 : a == 1 ? 1 : 3); }
__device__ __forceinline__ int phase_hi(int ph, int a) { return ph == 0 ? (a == 0 ? 1 : a == 1 ? 3 : 4) : (a == 0 ? 0 : a == 1 ? 2 : 4); }
__device__ __forceinline__ int phase_of(int ph, int r) { return ph == 0 ? (r >> 1) : ((r + 1) >> 1); }   // a(py, r)

// wc[((py*2+px)*9 + a*3+b)*n + i] = sum W[(r*5+s)*n + i],  n = Cin*Cout; blockIdx.z selects one of up to
// three layers so that D1..D3 collapse in ONE launch
struct CollapseJobs { const float* w[3]; float* wc[3]; int n[3]; };
__global__ __launch_bounds__(256) void collapse_w_kernel(CollapseJobs jobs) {
    const float* __restrict__ w = jobs.w[blockIdx.z];
    float* __restrict__ wc = jobs.wc[blockIdx.z];
    const int n = jobs.n[blockIdx.z];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int t = blockIdx.y, p = t / 9, a = (t % 9) / 3, b = t % 3, py = p >> 1, px = p & 1;
    float acc = 0.f;
    for (int r = phase_lo(py, a); r <= phase_hi(py, a); ++r)
        for (int s = phase_lo(px, b); s <= phase_hi(px, b); ++s) acc += w[(size_t)(r * 5 + s) * n + i];
    wc[(size_t)t * n + i] = acc;
}

// dw[(r*5+s)*n + i] = sum_{py,px} dwc[((py*2+px)*9 + a(py,r)*3 + b(px,s))*n + i], where dwc is the sum
// of the R (<= 32) partial rows left by the slab reduction; the bias gradient (the 36*n.. tail of
// each row) is summed by the first workgroup.  Fixed order throughout.
__global__ __launch_bounds__(256) void expand_dw_kernel(const float* __restrict__ rows, int R, int64_t stride,
                                                        float* __restrict__ dw, int n, float* __restrict__ dbias,
                                                        int cout) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (dbias != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < cout) {
        float b = 0.f;
        for (int q = 0; q < R; ++q) b += rows[(size_t)q * stride + (size_t)36 * n + threadIdx.x];
        dbias[threadIdx.x] = b;
    }
    if (i >= n) return;
    const int r = blockIdx.y / 5, s = blockIdx.y % 5;
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float* src = rows + (size_t)(p * 9 + phase_of(p >> 1, r) * 3 + phase_of(p & 1, s)) * n + i;
        float a = 0.f;
        for (int q = 0; q < R; ++q) a += src[(size_t)q * stride];
        acc += a;
    }
    dw[(size_t)blockIdx.y * n + i] = acc;
}

// ---------------------------------------------------------------------------------------------
// forward: acc[phase] over K = (ci chunk, a, b); WG = 128 low-res pixels x 32 output channels
// ---------------------------------------------------------------------------------------------
struct UpArgs {
    const float* in;      // fwd: stored low-res input (B,HS,HS,CIN)   dgrad: dout (B,2HS,2HS,COUT)
    const float* wc;      // [4][9][CIN][COUT]
    const float* bias;    // fwd
    const float* aux;     // dgrad: forward output of the producing layer (ReLU mask), low-res
    float* out;
    int B;
    int64_t sliceFloats;  // KSPLIT > 1: out = slab [KSPLIT][sliceFloats] of raw partial sums
};

template <int CIN, int COUT, int HS, int KSPLIT>
__global__ __launch_bounds__(256) void conv_up_fwd_kernel(UpArgs a) {
    using T = UpTile<HS>;
    constexpr int NT = 32, H = 2 * HS;
    constexpr int IN_FLOATS = KC * T::PS, W_FLOATS = 12 * KC * NT;
    __shared__ __attribute__((aligned(16))) float smem[IN_FLOATS + W_FLOATS];
    float* lds_in = smem;
    float* lds_w = smem + IN_FLOATS;           // [phase*3+b][kc][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int mt = xcd_tile(blockIdx.x, gridDim.x), n0 = blockIdx.y * NT;
    const int tileInImg = mt % T::TILES_PER_IMG, img0 = (mt / T::TILES_PER_IMG) * T::IMGS;
    const int ty0 = (tileInImg / T::TILES_X) * T::TH, tx0 = (tileInImg % T::TILES_X) * T::TW;
    const int m = wave * 32 + li;
    const int pimg = m / (T::TH * T::TW), prem = m % (T::TH * T::TW);
    const int aBase = lh * T::PS + pimg * T::HPI + (prem / T::TW) * T::HTW + (prem % T::TW);
    const int bBase = lh * NT + li;

    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[p][v] = 0.f;

    constexpr int WQ = 12 * KC * NT / 4, WPT = WQ / 256;      // 1536 float4, 6 per thread
    f32x4 wreg[WPT];
    auto load_w = [&](int st) {
        const int cc = st / 3, ar = st % 3;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int q = tid + i * 256;
            const int row = q / (NT / 4), c4 = q % (NT / 4);      // row = (p*3+b)*KC + kc
            const int pb = row / KC, kc = row % KC, p = pb / 3, b = pb % 3;
            wreg[i] = *reinterpret_cast<const f32x4*>(
                a.wc + ((size_t)(p * 9 + ar * 3 + b) * CIN + cc * KC + kc) * COUT + n0 + c4 * 4);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) *reinterpret_cast<f32x4*>(lds_w + (tid + i * 256) * 4) = wreg[i];
    };
    // low-res input halo chunk: global -> registers one chunk ahead -> LDS channel planes
    constexpr int NQ = T::HP * (KC / 4), IPT = (NQ + 255) / 256;
    f32x4 ireg[IPT];
    auto load_input = [&](int cc) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256;
            const int c4 = q & 3, hp = q >> 2;
            const int img = hp / T::HPI, rem = hp - img * T::HPI;
            const int gy = ty0 + rem / T::HTW - 1, gx = tx0 + rem % T::HTW - 1, ib = img0 + img;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((NQ % 256 == 0 || q < NQ) && (unsigned)gy < (unsigned)HS && (unsigned)gx < (unsigned)HS && ib < a.B)
                v = *reinterpret_cast<const f32x4*>(a.in + ((size_t)(ib * HS + gy) * HS + gx) * CIN + cc * KC + c4 * 4);
            ireg[i] = v;
        }
    };
    auto store_input = [&]() {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256;
            if (NQ % 256 == 0 || q < NQ) {
                float* d = lds_in + ((q & 3) * 4) * T::PS + (q >> 2);
                d[0] = ireg[i].x; d[T::PS] = ireg[i].y; d[2 * T::PS] = ireg[i].z; d[3 * T::PS] = ireg[i].w;
            }
        }
    };

    static_assert((CIN / KC) % KSPLIT == 0, "split-K must divide the channel chunks");
    constexpr int NST = (CIN / KC) / KSPLIT * 3;
    const int st0 = blockIdx.z * NST, st1 = st0 + NST;
    const float bv = KSPLIT > 1 ? 0.f : a.bias[n0 + li];          // requested ahead of every other load (conv_epilogue.h, load_bias)
    load_w(st0);
    load_input(st0 / 3);
    for (int st = st0; st < st1; ++st) {
        const int ar = st % 3;
        __syncthreads();
        if (ar == 0) store_input();
        store_w();
        if (ar == 0 && st + 3 < st1) load_input(st / 3 + 1);     // older than the weight load below: see conv_mfma.hip
        if (st + 1 < st1) load_w(st + 1);
        __syncthreads();
        const float* ap = lds_in + aBase + ar * T::HTW;
        __builtin_amdgcn_iglp_opt(0);
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int j = 0; j < KC / 2; ++j) {
                const float av = ap[(2 * j) * T::PS + b];
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, lds_w[bBase + ((p * 3 + b) * KC + 2 * j) * NT], acc[p], 0, 0, 0);
            }
    }
    // epilogue: phase p of low-res pixel (y,x) -> output pixel (2y+py, 2x+px)
    vm_drained();
    float* out = KSPLIT > 1 ? a.out + (size_t)blockIdx.z * a.sliceFloats : a.out;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int mm = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            const int im = mm / (T::TH * T::TW), rem = mm % (T::TH * T::TW);
            const int oy = 2 * (ty0 + rem / T::TW) + (p >> 1), ox = 2 * (tx0 + rem % T::TW) + (p & 1), ib = img0 + im;
            float x = acc[p][v] + bv;
            if (KSPLIT == 1) x = fmaxf(x, 0.f);
            if (ib < a.B) out[((size_t)(ib * H + oy) * H + ox) * COUT + n0 + li] = x;
        }
}

// ---------------------------------------------------------------------------------------------
// dgrad: WG = 128 low-res pixels x NT input channels; K = (co chunk, phase, a, b)
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT, int HS, int NT, int KSPLIT>
__global__ __launch_bounds__(256) void conv_up_dgrad_kernel(UpArgs a) {
    using T = UpTile<HS>;
    constexpr int NB = NT / 32, H = 2 * HS;
    constexpr int IN_FLOATS = KC * T::PS, W_FLOATS = 9 * NT * KCP;
    __shared__ __attribute__((aligned(16))) float smem[IN_FLOATS + W_FLOATS];
    float* lds_in = smem;
    float* lds_w = smem + IN_FLOATS;           // [tap][n][KCP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int mt = xcd_tile(blockIdx.x, gridDim.x), n0 = blockIdx.y * NT;
    const int tileInImg = mt % T::TILES_PER_IMG, img0 = (mt / T::TILES_PER_IMG) * T::IMGS;
    const int ty0 = (tileInImg / T::TILES_X) * T::TH, tx0 = (tileInImg % T::TILES_X) * T::TW;
    const int m = wave * 32 + li;
    const int pimg = m / (T::TH * T::TW), prem = m % (T::TH * T::TW);
    // tap (a,b) reads D_p[y-a+1][x-b+1] = halo index (pty + 2 - a, ptx + 2 - b)
    const int aBase = lh * T::PS + pimg * T::HPI + (prem / T::TW + 2) * T::HTW + (prem % T::TW) + 2;
    const int bBase = li * KCP + lh;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[nb][v] = 0.f;

    constexpr int WQ = 9 * NT * KC / 4, WPT = (WQ + 255) / 256;
    f32x4 wreg[WPT];
    auto load_w = [&](int st) {
        const int cc = st / 4, p = st % 4;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int q = tid + i * 256;
            if (WQ % 256 == 0 || q < WQ) {
                const int c4 = q & 3, rown = q >> 2, n = rown % NT, t = rown / NT;
                wreg[i] = *reinterpret_cast<const f32x4*>(
                    a.wc + ((size_t)(p * 9 + t) * CIN + n0 + n) * COUT + cc * KC + c4 * 4);
            }
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int q = tid + i * 256;
            if (WQ % 256 == 0 || q < WQ) {
                float* d = lds_w + (q >> 2) * KCP + (q & 3) * 4;
                d[0] = wreg[i].x; d[1] = wreg[i].y; d[2] = wreg[i].z; d[3] = wreg[i].w;
            }
        }
    };
    // D_p[yy][xx] = dout[2yy+py][2xx+px], halo +-1, zero outside: global -> registers one stage
    // ahead -> LDS channel planes
    constexpr int NQ = T::HP * (KC / 4), IPT = (NQ + 255) / 256;
    f32x4 ireg[IPT];
    auto load_input = [&](int st) {
        const int cc = st / 4, py = (st % 4) >> 1, px = st & 1;
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256;
            const int c4 = q & 3, hp = q >> 2;
            const int img = hp / T::HPI, rem = hp - img * T::HPI;
            const int yy = ty0 + rem / T::HTW - 1, xx = tx0 + rem % T::HTW - 1, ib = img0 + img;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((NQ % 256 == 0 || q < NQ) && (unsigned)yy < (unsigned)HS && (unsigned)xx < (unsigned)HS && ib < a.B)
                v = *reinterpret_cast<const f32x4*>(
                    a.in + ((size_t)(ib * H + 2 * yy + py) * H + 2 * xx + px) * COUT + cc * KC + c4 * 4);
            ireg[i] = v;
        }
    };
    auto store_input = [&]() {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256;
            if (NQ % 256 == 0 || q < NQ) {
                float* d = lds_in + ((q & 3) * 4) * T::PS + (q >> 2);
                d[0] = ireg[i].x; d[T::PS] = ireg[i].y; d[2 * T::PS] = ireg[i].z; d[3 * T::PS] = ireg[i].w;
            }
        }
    };

    static_assert((COUT / KC) % KSPLIT == 0, "split-K must divide the channel chunks");
    constexpr int NST = (COUT / KC) / KSPLIT * 4;
    const int st0 = blockIdx.z * NST, st1 = st0 + NST;
    load_w(st0);
    load_input(st0);
    for (int st = st0; st < st1; ++st) {
        __syncthreads();
        store_input();
        store_w();
        if (st + 1 < st1) { load_input(st + 1); load_w(st + 1); }
        __syncthreads();
        __builtin_amdgcn_iglp_opt(0);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < KC / 2; ++j) {
                const float av = lds_in[aBase + (2 * j) * T::PS - (t / 3) * T::HTW - (t % 3)];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, lds_w[bBase + (t * NT + nb * 32) * KCP + 2 * j], acc[nb], 0, 0, 0);
            }
    }
    vm_drained();
    float* out = KSPLIT > 1 ? a.out + (size_t)blockIdx.z * a.sliceFloats : a.out;
    // the ReLU mask values of ALL channel blocks are requested before the first store: vmcnt retires loads and stores together in
    // issue order, so a mask load issued behind a store waits for that store's write acknowledgement (and `out` / `aux` may alias
    // as far as the compiler knows: a load behind each store would be one memory round trip per element)
    size_t o[NB][16];
    float mk[NB][16];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int mm = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            const int im = mm / (T::TH * T::TW), rem = mm % (T::TH * T::TW), ib = img0 + im;
            o[nb][v] = ((size_t)(ib * HS + ty0 + rem / T::TW) * HS + tx0 + rem % T::TW) * CIN + n0 + nb * 32 + li;
            if (KSPLIT == 1) mk[nb][v] = a.aux[ib < a.B ? o[nb][v] : 0];
        }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int mm = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            if (img0 + mm / (T::TH * T::TW) >= a.B) continue;
            float x = acc[nb][v];
            if (KSPLIT == 1) x = mk[nb][v] > 0.f ? x : 0.f;
            out[o[nb][v]] = x;
        }
}

// split-K finish kernels
__global__ __launch_bounds__(256) void up_finish_relu_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                             float* __restrict__ out, int64_t n4, int64_t slice, int KS, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 acc = *reinterpret_cast<const float4*>(bias + (i * 4) % C);
    for (int z = 0; z < KS; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)z * slice + i * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    *reinterpret_cast<float4*>(out + i * 4) = acc;
}
__global__ __launch_bounds__(256) void up_finish_mask_kernel(const float* __restrict__ slab, const float* __restrict__ aux,
                                                             float* __restrict__ out, int64_t n4, int64_t slice, int KS) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < KS; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)z * slice + i * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float4 m = *reinterpret_cast<const float4*>(aux + i * 4);
    acc.x = m.x > 0.f ? acc.x : 0.f; acc.y = m.y > 0.f ? acc.y : 0.f;
    acc.z = m.z > 0.f ? acc.z : 0.f; acc.w = m.w > 0.f ? acc.w : 0.f;
    *reinterpret_cast<float4*>(out + i * 4) = acc;
}

// ---------------------------------------------------------------------------------------------
// wgrad: WG owns 32ci x 32co x (4 phases x 9 taps); wave w = phase w; K = low-res pixels (64/tile)
// ---------------------------------------------------------------------------------------------
template <int HS>        // 64 low-res pixels per tile
struct UpTile64 {
    static constexpr int TW = HS < 16 ? HS : 16;
    static constexpr int TH = (64 / TW) < HS ? (64 / TW) : HS;
    static constexpr int IMGS = 64 / (TW * TH);
    static constexpr int HTW = TW + 2, HTH = TH + 2, HPI = HTW * HTH, HP = IMGS * HPI;
    static constexpr int TILES_X = HS / TW, TILES_Y = HS / TH, TILES_PER_IMG = TILES_X * TILES_Y;
};

struct UpWgradArgs {
    const float* in;      // (B,HS,HS,CIN)
    const float* dout;    // (B,2HS,2HS,COUT)
    float* slab;          // [S][36*CIN*COUT + COUT]
    int B, numTiles, tilesPerSplit;
};

template <int CIN, int COUT, int HS>
__global__ __launch_bounds__(256, 2) void conv_up_wgrad_kernel(UpWgradArgs a) {
    using T = UpTile64<HS>;
    constexpr int H = 2 * HS, CS = 32, IN_FLOATS = T::HP * CS;
    __shared__ __attribute__((aligned(16))) float smem[IN_FLOATS + 4 * 64 * 32];
    float* lds_in = smem;                      // [halo pixel][32 ci]
    float* lds_d = smem + IN_FLOATS;           // [phase][64 low-res pixels][32 co]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x, ci0 = blockIdx.y * 32, n0 = blockIdx.z * 32;
    f32x16 acc[9];
    float bsum = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    const int t0 = split * a.tilesPerSplit;
    int t1 = t0 + a.tilesPerSplit; if (t1 > a.numTiles) t1 = a.numTiles;
    // software pipeline: tile mt+1 travels global -> registers while the MFMAs of tile mt run
    constexpr int IQ = (T::HP * 8 + 255) / 256;
    f32x4 rin[IQ], rdo[8];
    auto fetch = [&](int mt) {
        const int tileInImg = mt % T::TILES_PER_IMG, img0 = (mt / T::TILES_PER_IMG) * T::IMGS;
        const int ty0 = (tileInImg / T::TILES_X) * T::TH, tx0 = (tileInImg % T::TILES_X) * T::TW;
#pragma unroll
        for (int i = 0; i < IQ; ++i) {
            const int q = tid + i * 256;
            const int c4 = q & 7, hp = q >> 3;
            const int img = hp / T::HPI, rem = hp % T::HPI;
            const int gy = ty0 + rem / T::HTW - 1, gx = tx0 + rem % T::HTW - 1, ib = img0 + img;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (((T::HP * 8) % 256 == 0 || q < T::HP * 8) && (unsigned)gy < (unsigned)HS && (unsigned)gx < (unsigned)HS && ib < a.B)
                v = *reinterpret_cast<const f32x4*>(a.in + ((size_t)(ib * HS + gy) * HS + gx) * CIN + ci0 + c4 * 4);
            rin[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = tid + i * 256;
            const int c4 = q & 7, mm = (q >> 3) & 63, p = q >> 9;
            const int im = mm / (T::TH * T::TW), rem = mm % (T::TH * T::TW);
            const int oy = 2 * (ty0 + rem / T::TW) + (p >> 1), ox = 2 * (tx0 + rem % T::TW) + (p & 1), ib = img0 + im;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ib < a.B) v = *reinterpret_cast<const f32x4*>(a.dout + ((size_t)(ib * H + oy) * H + ox) * COUT + n0 + c4 * 4);
            rdo[i] = v;
        }
    };
    if (t0 < t1) fetch(t0);
    for (int mt = t0; mt < t1; ++mt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < IQ; ++i) {
            const int q = tid + i * 256;
            if ((T::HP * 8) % 256 == 0 || q < T::HP * 8) *reinterpret_cast<f32x4*>(lds_in + (q >> 3) * CS + (q & 7) * 4) = rin[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = tid + i * 256;
            *reinterpret_cast<f32x4*>(lds_d + (q >> 9) * 64 * 32 + ((q >> 3) & 63) * 32 + (q & 7) * 4) = rdo[i];
        }
        __syncthreads();
        if (mt + 1 < t1) fetch(mt + 1);
        const float* dph = lds_d + wave * 64 * 32 + li;
#pragma unroll 4
        for (int kk = 0; kk < 32; ++kk) {
            const int mm = 2 * kk + lh;
            const int im = mm / (T::TH * T::TW), rem = mm % (T::TH * T::TW);
            const float* ip = lds_in + ((im * T::HTH + rem / T::TW) * T::HTW + rem % T::TW) * CS + li;
            const float bv = dph[mm * 32];
            bsum += bv;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ip[((t / 3) * T::HTW + t % 3) * CS], bv, acc[t], 0, 0, 0);
        }
    }
    float* out = a.slab + (size_t)split * (36 * CIN * COUT + COUT);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int ci = ci0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            out[((size_t)(wave * 9 + t) * CIN + ci) * COUT + n0 + li] = acc[t][v];
        }
    // bias gradient partial = column sums of dout over the four phases
    __syncthreads();
    bsum += __shfl_xor(bsum, 32, 64);
    if (lh == 0) smem[wave * 32 + li] = bsum;
    __syncthreads();
    if (wave == 0 && lh == 0 && blockIdx.y == 0)
        out[(size_t)36 * CIN * COUT + n0 + li] = (smem[li] + smem[32 + li]) + (smem[64 + li] + smem[96 + li]);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct UpCfg { int cin, cout, hs; };
static UpCfg up_cfg(int layer, int width) {
    return UpCfg{kLayers[layer].cin, kLayers[layer].cout, kLayers[layer].h * (width / 64) / 2};
}
int64_t conv_up_wc_floats(int layer) { return (int64_t)36 * kLayers[layer].cin * kLayers[layer].cout; }

int launch_collapse_w(int layer, const float* w, float* wc, hipStream_t st) {
    const int n = kLayers[layer].cin * kLayers[layer].cout;
    CollapseJobs jobs{{w, nullptr, nullptr}, {wc, nullptr, nullptr}, {n, 0, 0}};
    hipLaunchKernelGGL(collapse_w_kernel, dim3(cdiv(n, 256), 36, 1), dim3(256), 0, st, jobs);
    CVAE_CHECK_LAUNCH();
    return 0;
}
// D1..D3 (layers 5..7) in one launch
int launch_collapse_w3(const float* const w[3], float* const wc[3], hipStream_t st) {
    CollapseJobs jobs;
    int nmax = 0;
    for (int i = 0; i < 3; ++i) {
        jobs.w[i] = w[i]; jobs.wc[i] = wc[i]; jobs.n[i] = kLayers[5 + i].cin * kLayers[5 + i].cout;
        if (jobs.n[i] > nmax) nmax = jobs.n[i];
    }
    hipLaunchKernelGGL(collapse_w_kernel, dim3(cdiv(nmax, 256), 36, 3), dim3(256), 0, st, jobs);
    CVAE_CHECK_LAUNCH();
    return 0;
}

template <int CIN, int COUT, int HS, int KSPLIT>
static int run_up_fwd(int B, const float* in, const float* wc, const float* bias, float* out, float* ws, hipStream_t st) {
    using T = UpTile<HS>;
    const int64_t slice = (int64_t)B * 4 * HS * HS * COUT;
    UpArgs a{in, wc, bias, nullptr, KSPLIT > 1 ? ws : out, B, slice};
    cvae_probe_begin(st);
    hipLaunchKernelGGL((conv_up_fwd_kernel<CIN, COUT, HS, KSPLIT>), dim3(cdiv(B, T::IMGS) * T::TILES_PER_IMG, COUT / 32, KSPLIT),
                       dim3(256), 0, st, a);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    if (KSPLIT > 1) {
        hipLaunchKernelGGL(up_finish_relu_kernel, dim3((unsigned)((slice / 4 + 255) / 256)), dim3(256), 0, st, ws, bias, out,
                           slice / 4, slice, KSPLIT, COUT);
        CVAE_CHECK_LAUNCH();
    }
    return 0;
}

template <int CIN, int COUT, int HS, int NT, int KSPLIT>
static int run_up_dgrad(int B, const float* dout, const float* wc, const float* aux, float* din, float* ws, hipStream_t st) {
    using T = UpTile<HS>;
    const int64_t slice = (int64_t)B * HS * HS * CIN;
    UpArgs a{dout, wc, nullptr, aux, KSPLIT > 1 ? ws : din, B, slice};
    cvae_probe_begin(st);
    hipLaunchKernelGGL((conv_up_dgrad_kernel<CIN, COUT, HS, NT, KSPLIT>), dim3(cdiv(B, T::IMGS) * T::TILES_PER_IMG, CIN / NT, KSPLIT),
                       dim3(256), 0, st, a);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    if (KSPLIT > 1) {
        hipLaunchKernelGGL(up_finish_mask_kernel, dim3((unsigned)((slice / 4 + 255) / 256)), dim3(256), 0, st, ws, aux, din,
                           slice / 4, slice, KSPLIT);
        CVAE_CHECK_LAUNCH();
    }
    return 0;
}

static constexpr int UP_KS5 = 4, UP_KS6 = 2;       // split-K of D1 / D2 (few low-res pixels at B=256)

int64_t conv_up_ws_floats(int layer, int width, int B) {
    const UpCfg c = up_cfg(layer, width);
    const int64_t fwd = (int64_t)B * 4 * c.hs * c.hs * c.cout, dg = (int64_t)B * c.hs * c.hs * c.cin;
    const int ks = layer == 5 ? UP_KS5 : layer == 6 ? UP_KS6 : 0;
    return ks * (fwd > dg ? fwd : dg);
}

int launch_conv_up_fwd(int layer, int width, int B, const float* in, const float* wc, const float* bias, float* out,
                       float* ws, hipStream_t st) {
    if (width == 64) {
        switch (layer) {
            case 5: return run_up_fwd<128, 64, 4, UP_KS5>(B, in, wc, bias, out, ws, st);
            case 6: return run_up_fwd<64, 32, 8, UP_KS6>(B, in, wc, bias, out, ws, st);
            case 7: return run_up_fwd<32, 32, 16, 1>(B, in, wc, bias, out, ws, st);
        }
    }
    if (width == 128) {
        switch (layer) {
            case 5: return run_up_fwd<128, 64, 8, UP_KS5>(B, in, wc, bias, out, ws, st);
            case 6: return run_up_fwd<64, 32, 16, UP_KS6>(B, in, wc, bias, out, ws, st);
            case 7: return run_up_fwd<32, 32, 32, 1>(B, in, wc, bias, out, ws, st);
        }
    }
    cvae_set_error("conv_up_fwd: unsupported layer %d at width %d", layer, width);
    return -2;
}

int launch_conv_up_dgrad(int layer, int width, int B, const float* dout, const float* wc, const float* aux, float* din,
                         float* ws, hipStream_t st) {
    if (width == 64) {
        switch (layer) {
            case 5: return run_up_dgrad<128, 64, 4, 64, UP_KS5>(B, dout, wc, aux, din, ws, st);
            case 6: return run_up_dgrad<64, 32, 8, 64, UP_KS6>(B, dout, wc, aux, din, ws, st);
            case 7: return run_up_dgrad<32, 32, 16, 32, 1>(B, dout, wc, aux, din, ws, st);
        }
    }
    if (width == 128) {
        switch (layer) {
            case 5: return run_up_dgrad<128, 64, 8, 64, UP_KS5>(B, dout, wc, aux, din, ws, st);
            case 6: return run_up_dgrad<64, 32, 16, 64, UP_KS6>(B, dout, wc, aux, din, ws, st);
            case 7: return run_up_dgrad<32, 32, 32, 32, 1>(B, dout, wc, aux, din, ws, st);
        }
    }
    cvae_set_error("conv_up_dgrad: unsupported layer %d at width %d", layer, width);
    return -2;
}

template <int CIN, int COUT, int HS>
static int run_up_wgrad(int B, const float* in, const float* dout, float* dw, float* dbias, float* ws, hipStream_t st,
                        int64_t* need, int bf16_layer = 0, int bf16_width = 64) {
    using T = UpTile64<HS>;
    const int numTiles = cdiv(B, T::IMGS) * T::TILES_PER_IMG;
    const int bps = (CIN / 32) * (COUT / 32);
    // splits: a whole number of workgroups per CU (one for the fp32 kernel, two for the bf16 one) with equal tile
    // counts -- 1.5 per CU leaves a third of the CUs with twice the work of the rest
    const int cus = cvae_num_cus();
    int Scap = cdiv(2 * cus, bps);
    if (Scap > numTiles) Scap = numTiles;
    int S = cdiv(cus, bps);
    if (S > numTiles) S = numTiles;
    const int tps = cdiv(numTiles, S);
    S = cdiv(numTiles, tps);
    const int64_t nc = (int64_t)36 * CIN * COUT, row = nc + COUT;
    // ws = [S slabs | 16 mid rows | reduced row]
    if (need) { *need = (int64_t)(Scap + 17) * row; return 0; }
    if (bf16_layer) {          // precision mode 1: same slab rows from the bf16-MFMA kernel (conv_bf16.hip), S' <= S of them
        int rc = launch_up_wgrad_bf16_main(bf16_layer, bf16_width, B, in, dout, ws, Scap, &S, st);
        if (rc) return rc;
    } else {
        UpWgradArgs a{in, dout, ws, B, numTiles, tps};
        cvae_probe_begin(st);
        hipLaunchKernelGGL((conv_up_wgrad_kernel<CIN, COUT, HS>), dim3(S, CIN / 32, COUT / 32), dim3(256), 0, st, a);
        cvae_probe_end(st);
        CVAE_CHECK_LAUNCH();
    }
    float* mid = ws + (size_t)S * row;
    float* red = mid + (size_t)16 * row;
    st = cvae_reduce_stream(st);              // reduce + expand are off the critical path
    { int rc = launch_reduce_slabs(ws, red, row, S, row, st, mid); if (rc) return rc; }
    const float* rows = red;
    const int R = 1;
    static_assert(COUT <= 256, "expand_dw_kernel sums the bias gradient in one workgroup");
    hipLaunchKernelGGL(expand_dw_kernel, dim3(cdiv(CIN * COUT, 256), 25), dim3(256), 0, st, rows, R, row, dw, CIN * COUT, dbias, COUT);
    CVAE_CHECK_LAUNCH();
    return 0;
}

static int dispatch_up_wgrad(int layer, int width, int B, const float* in, const float* dout, float* dw, float* dbias,
                             float* ws, hipStream_t st, int64_t* need, bool bf16 = false) {
    if (width == 64) {
        switch (layer) {
            case 5: return run_up_wgrad<128, 64, 4>(B, in, dout, dw, dbias, ws, st, need, bf16 ? 5 : 0);
            case 6: return run_up_wgrad<64, 32, 8>(B, in, dout, dw, dbias, ws, st, need, bf16 ? 6 : 0);
            case 7: return run_up_wgrad<32, 32, 16>(B, in, dout, dw, dbias, ws, st, need, bf16 ? 7 : 0);
        }
    }
    if (width == 128) {
        switch (layer) {
            case 5: return run_up_wgrad<128, 64, 8>(B, in, dout, dw, dbias, ws, st, need, bf16 ? 5 : 0, 128);
            case 6: return run_up_wgrad<64, 32, 16>(B, in, dout, dw, dbias, ws, st, need, bf16 ? 6 : 0, 128);
            case 7: return run_up_wgrad<32, 32, 32>(B, in, dout, dw, dbias, ws, st, need, bf16 ? 7 : 0, 128);
        }
    }
    cvae_set_error("conv_up_wgrad: unsupported layer %d at width %d", layer, width);
    return -2;
}
int64_t conv_up_wgrad_ws_floats(int layer, int width, int B) {
    int64_t need = 0;
    if (dispatch_up_wgrad(layer, width, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &need)) return 0;
    return need;
}
int launch_conv_up_wgrad(int layer, int width, int B, const float* in, const float* dout, float* dw, float* dbias,
                         float* ws, hipStream_t st, bool bf16) {
    return dispatch_up_wgrad(layer, width, B, in, dout, dw, dbias, ws, st, nullptr, bf16);
}
