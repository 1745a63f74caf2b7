// conv_mfma.hip — 5x5 / stride 1 / pad 2 convolution as implicit GEMM on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain), NHWC activations.
//
// Replaces the ATen call sites nn.Conv2d forward (vae_nets.py:74,79,84 encoder E2..E4;
// :117,121,125,129 decoder D0..D3) and their input-gradients in loss.backward() (vae.py:57).
//
//   forward : out[p][n]  = sum_{tap,c} in[p+tap-2][c] * W[tap][c][n]            (+bias, epilogue)
//   dgrad   : din[p][n]  = sum_{tap,c} dout[p+tap-2][c] * W[24-tap][n][c]      (flipped taps,
//             transposed weight read from the SAME [tap][Cin][Cout] array)
//
// One workgroup = 4 waves = 128 output pixels (Tile<H>) x NT output channels; wave w owns pixel
// rows [32w,32w+32) of the tile and NB=NT/32 accumulator tiles.  K loop = 16-channel chunks x
// 5 kernel rows: the input halo of a chunk is staged once in LDS as channel planes and reused
// by all 25 taps; the weight slab of one kernel row is register-prefetched one stage ahead.
// The decoder's nearest-2x Upsample (vae_nets.py:119,...) is never materialised: UP folds
// src=(y>>1,x>>1) into the halo gather; its backward (2x2 sum) and the ReLU mask are the
// POOLSUM epilogue of dgrad.
#include "common.h"
#include <stdlib.h>
#include "conv_epilogue.h"


struct ConvArgs {
    const float* in;
    const float* w;
    const float* bias;
    const float* aux;   // POOLSUM: forward output of the producing layer (ReLU mask source)
    float* out;
    float* bnpart;      // BNSTAT: [2][numTiles][NCH] (sum, M2 about the tile mean)
    int B;
    int64_t sliceFloats;   // KSPLIT > 1: out is a slab [KSPLIT][sliceFloats] of raw partial sums
};

// channels per K chunk: the NT=32 kernels run half as many MFMAs per stage, so they take 32-channel
// chunks to keep ~80 MFMAs per wave between barrier pairs; KCP = padded row of the transposed (dgrad) slab
template <int NT> struct KChunk { static constexpr int KC = NT == 32 ? 32 : 16, KCP = KC + 1; };

#ifdef CONVF_TIMING     // experiment builds only (see conv_bf16.hip CONV_TIMING): stage timing of one fp32 instantiation
__device__ long long convf_dbg[16 * 4 * 10];
extern "C" int cvae_convf_dbg_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(convf_dbg), sizeof(convf_dbg)); }
#define CF_ON (KCH == CONVF_TIMING_KCH && NCH == CONVF_TIMING_NCH && H == CONVF_TIMING_H && !UP)
#define CF_STAMP(v) do { if (CF_ON) v = clock64(); } while (0)
#else
#define CF_ON false
#define CF_STAMP(v)
#endif

template <int KCH, int NCH, int H, bool UP, bool DGRAD, int NT, int EPI, int KSPLIT>
__global__ __launch_bounds__(256) void conv5x5_mfma_kernel(ConvArgs a) {
    using T = Tile<H>;
    constexpr int NB = NT / 32;
    constexpr int KC = KChunk<NT>::KC, KCP = KChunk<NT>::KCP, QPP = KC / 4;     // QPP: float4 per pixel per chunk
    constexpr int IN_FLOATS = KC * T::PS;
    constexpr int W_FLOATS = DGRAD ? 5 * NT * KCP : 5 * KC * NT;
    constexpr int EPI_FLOATS = (EPI == EPI_POOLSUM_MASK) ? 128 * (NT + 1) : 4 * 32 * 36;
    constexpr int SMEM = (IN_FLOATS + W_FLOATS) > EPI_FLOATS ? (IN_FLOATS + W_FLOATS) : EPI_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* lds_in = smem;
    float* lds_w = smem + IN_FLOATS;     // IN_FLOATS is a multiple of 4 (KC*PS, KC=16)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int mt = xcd_tile(blockIdx.x, gridDim.x), n0 = blockIdx.y * NT;
    const int tileInImg = mt % T::TILES_PER_IMG;
    const int img0 = (mt / T::TILES_PER_IMG) * T::IMGS;
    const int ty0 = (tileInImg / T::TILES_X) * T::TH, tx0 = (tileInImg % T::TILES_X) * T::TW;
    constexpr int HS = UP ? H / 2 : H;   // stored spatial size of the input tensor

    // A operand: lane (li, lh) reads pixel m = 32*wave + li, channel 2j+lh of the chunk
    const int m = wave * 32 + li;
    const int pimg = m / (T::TH * T::TW), prem = m % (T::TH * T::TW);
    const int aBase = lh * T::PS + pimg * T::HPI + (prem / T::TW) * T::HTW + (prem % T::TW);
    const int bBase = DGRAD ? (li * KCP + lh) : (lh * NT + li);

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[nb][v] = 0.f;

    constexpr int WQ = 5 * KC * NT / 4;              // float4 per weight slab
    constexpr int WPT = (WQ + 255) / 256;
    f32x4 wreg[WPT];

    auto load_w = [&](int st) {
        const int cc = st / 5, r = st % 5;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int q = tid + i * 256;
            if (WQ % 256 == 0 || q < WQ) {
                const float* src;
                if (!DGRAD) {
                    const int row = q / (NT / 4), c4 = q % (NT / 4);
                    const int s = row / KC, kc = row % KC;
                    src = a.w + (size_t)((r * 5 + s) * KCH + cc * KC + kc) * NCH + n0 + c4 * 4;
                } else {
                    const int c4 = q % QPP, rown = q / QPP;
                    const int n = rown % NT, s = rown / NT;
                    src = a.w + (size_t)((24 - (r * 5 + s)) * NCH + n0 + n) * KCH + cc * KC + c4 * 4;
                }
                wreg[i] = *reinterpret_cast<const f32x4*>(src);
            }
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int q = tid + i * 256;
            if (WQ % 256 == 0 || q < WQ) {
                if (!DGRAD) {
                    *reinterpret_cast<f32x4*>(lds_w + q * 4) = wreg[i];
                } else {
                    const int c4 = q % QPP, rown = q / QPP;
                    float* d = lds_w + rown * KCP + c4 * 4;   // rown = s*NT + n
                    d[0] = wreg[i].x; d[1] = wreg[i].y; d[2] = wreg[i].z; d[3] = wreg[i].w;
                }
            }
        }
    };
    // input halo chunk: global -> registers (issued one chunk ahead, in flight while five stages of
    // MFMAs run) -> LDS channel planes
    constexpr int NQ = T::HP * (KC / 4), IPT = (NQ + 255) / 256;
    f32x4 ireg[IPT];
    auto load_input = [&](int cc) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256;
            const int c4 = q % QPP, hp = q / QPP;
            const int img = hp / T::HPI, rem = hp - img * T::HPI;
            const int hy = rem / T::HTW, hx = rem - hy * T::HTW;
            const int gy = ty0 + hy - 2, gx = tx0 + hx - 2, ib = img0 + img;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((NQ % 256 == 0 || q < NQ) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)H && ib < a.B) {
                const int sy = UP ? (gy >> 1) : gy, sx = UP ? (gx >> 1) : gx;
                v = *reinterpret_cast<const f32x4*>(a.in + ((size_t)(ib * HS + sy) * HS + sx) * KCH + cc * KC + c4 * 4);
            }
            ireg[i] = v;
        }
    };
    auto store_input = [&]() {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256;
            if (NQ % 256 == 0 || q < NQ) {
                float* d = lds_in + ((q % QPP) * 4) * T::PS + (q / QPP);
                d[0] = ireg[i].x; d[T::PS] = ireg[i].y; d[2 * T::PS] = ireg[i].z; d[3 * T::PS] = ireg[i].w;
            }
        }
    };

    // split-K (small-spatial layers): workgroup z contracts channel chunks [z*CPS, (z+1)*CPS)
    static_assert((KCH / KC) % KSPLIT == 0, "split-K must divide the channel chunks");
    constexpr int NST = (KCH / KC) / KSPLIT * 5;
    const int st0 = blockIdx.z * NST, st1 = st0 + NST;
    [[maybe_unused]] float biasv[NB];              // fetched here, ahead of every other load: consumed before the first store
    if constexpr (KSPLIT == 1 && (EPI == EPI_BIAS_BNSTAT || EPI == EPI_BIAS_RELU)) load_bias<NT, EPI>(biasv, a.bias, n0);
    load_w(st0);
    load_input(st0 / 5);
    [[maybe_unused]] long long ct0 = 0, ct1 = 0, ct2 = 0, ct3 = 0, ct4 = 0, cd[6] = {0, 0, 0, 0, 0, 0}, ctb = 0, cta = 0, ctw = 0;      // CONVF_TIMING builds
    CF_STAMP(ctb);
    for (int st = st0; st < st1; ++st) {
        const int r = st % 5;
        CF_STAMP(ct0);
        __syncthreads();                       // everyone finished reading the previous stage
        CF_STAMP(ct1);
        if (r == 0) store_input();
        CF_STAMP(cta);
        store_w();
        CF_STAMP(ctw);
        // issue order matters: vmcnt retires in order, so the (older) halo loads must not sit
        // between a weight load and the store_w that waits for it
        if (r == 0 && st + 5 < st1) load_input(st / 5 + 1);     // lands during this stage's MFMAs
        if (st + 1 < st1) load_w(st + 1);      // in flight while this stage computes
        CF_STAMP(ct2);
        __syncthreads();
        CF_STAMP(ct3);
        const float* ap = lds_in + aBase + r * T::HTW;
        __builtin_amdgcn_iglp_opt(0);
#pragma unroll
        for (int s = 0; s < 5; ++s) {
#pragma unroll
            for (int j = 0; j < KC / 2; ++j) {
                const float av = ap[(2 * j) * T::PS + s];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float bv = DGRAD ? lds_w[bBase + (s * NT + nb * 32) * KCP + 2 * j]
                                           : lds_w[bBase + (s * KC + 2 * j) * NT + nb * 32];
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nb], 0, 0, 0);
                }
            }
        }
#ifdef CONVF_TIMING
        if (CF_ON) {
            __builtin_amdgcn_sched_barrier(0);
            long long te = clock64();
            cd[0] += ct1 - ct0; cd[1] += ct2 - ct1; cd[2] += ct3 - ct2; cd[3] += te - ct3; cd[4] += cta - ct1; cd[5] += ctw - cta;
        }
#endif
    }
    CF_STAMP(ct4);
    vm_drained();

    // ------------------------------- epilogue -------------------------------
    if (KSPLIT > 1) {
        epilogue_store<H, NT, NCH, EPI_PLAIN>(acc, a.out + (size_t)blockIdx.z * a.sliceFloats, nullptr, smem, a.B,
                                              mt, n0, img0, ty0, tx0);
    } else if (EPI == EPI_BIAS_BNSTAT || EPI == EPI_BIAS_RELU || EPI == EPI_PLAIN) {
        apply_bias<NT, EPI>(acc, biasv);           // both channel blocks before the first store (conv_epilogue.h)
        __builtin_amdgcn_sched_barrier(0);
        epilogue_store<H, NT, NCH, EPI>(acc, a.out, a.bnpart, smem, a.B, mt, n0, img0, ty0, tx0);
    } else {   // EPI_POOLSUM_MASK: 2x2 sum (upsample backward) then ReLU mask of the producer
        __syncthreads();
        float* lo = smem;                            // [128][NT+1]
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int mm = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                lo[mm * (NT + 1) + nb * 32 + li] = acc[nb][v];
            }
        __syncthreads();
        constexpr int PTW = T::TW / 2, PTH = T::TH / 2, HO = H / 2;
        for (int q = tid; q < 32 * NT; q += 256) {
            const int c = q % NT, pp = q / NT;
            const int im = pp / (PTW * PTH), rem = pp % (PTW * PTH);
            const int py = rem / PTW, px = rem % PTW, ib = img0 + im;
            if (ib >= a.B) continue;
            const int m00 = im * T::TH * T::TW + (2 * py) * T::TW + 2 * px;
            const float sum = (lo[m00 * (NT + 1) + c] + lo[(m00 + 1) * (NT + 1) + c]) +
                              (lo[(m00 + T::TW) * (NT + 1) + c] + lo[(m00 + T::TW + 1) * (NT + 1) + c]);
            const size_t o = ((size_t)(ib * HO + ty0 / 2 + py) * HO + tx0 / 2 + px) * NCH + n0 + c;
            a.out[o] = a.aux[o] > 0.f ? sum : 0.f;
        }
    }
#ifdef CONVF_TIMING
    if (CF_ON && (blockIdx.x & 63) == 0 && blockIdx.x < 1024 && blockIdx.y == 0 && lane == 0) {
        long long tend = clock64();
        long long* o = convf_dbg + ((blockIdx.x >> 6) * 4 + wave) * 10;
        o[0] = cd[0]; o[1] = cd[1]; o[2] = cd[2]; o[3] = cd[3]; o[4] = ct4 - ctb; o[5] = tend - ct4; o[6] = st1 - st0; o[7] = tend - ctb; o[8] = cd[4]; o[9] = cd[5];
    }
#endif
}

template <int KCH, int NCH, int H, bool UP, bool DGRAD, int NT, int EPI, int KSPLIT = 1>
static int run(const ConvArgs& a, hipStream_t st) {
    using T = Tile<H>;
    static_assert(KCH % KChunk<NT>::KC == 0 && NCH % NT == 0, "channel tiling");
    dim3 grid(cdiv(a.B, T::IMGS) * T::TILES_PER_IMG, NCH / NT, KSPLIT);
    cvae_probe_begin(st);
    hipLaunchKernelGGL((conv5x5_mfma_kernel<KCH, NCH, H, UP, DGRAD, NT, EPI, KSPLIT>), grid, dim3(256), 0, st, a);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// 4x4 images (decoder block D0 at 64x64 frames, vae_nets.py:117, and its input gradient): 51 % of the 25-tap MACs of a
// 5x5 / pad-2 conv on a 4x4 image multiply zero padding.  Here the M tile is 32 IMAGES x two image rows (0,1) or (2,3);
// wave w = column w of the first image row and column w ^ 1 of the second (32 MFMA rows = 32 images at ONE pixel
// position), so "tap outside the image" is uniform: a kernel row r runs only for the image rows y with y+r-2 inside
// (workgroup-uniform: 7 of 10 row-stages per chunk, the same for both row pairs; the kernel row that neither needs is not
// staged), taps s with x+s-2 outside are skipped by the wave (3 of 5 in columns 0 / 3, 4 in columns 1 / 2: the column
// swap between the rows gives every wave 7 taps per two-row stage).  One staged input chunk and one weight slab serve
// both image rows.  LDS holds the 16 real pixels of the 32 images as [channel plane][pixel][image] (no halo: every tap
// that runs is inside; lanes = consecutive images: conflict-free operand reads).  Weight slabs, register prefetch, k order
// inside a tap (channels ascending) as in conv5x5_mfma_kernel;
// always split-K over channel chunks (raw partial sums to slab z, the callers' finish kernels add them in fixed order).
// ---------------------------------------------------------------------------------------------
template <int KCH, int NCH, bool DGRAD, int NT, int KSPLIT>
__global__ __launch_bounds__(256) void conv4x4_row_kernel(ConvArgs a) {
    constexpr int NB = NT / 32;
    constexpr int KC = KChunk<NT>::KC, KCP = KChunk<NT>::KCP, QPP = KC / 4;
    constexpr int PXS = 33, PS = 16 * PXS + 2;                 // [pixel][32 images + 1], plane stride
    constexpr int IN_FLOATS = KC * PS;
    constexpr int W_FLOATS = DGRAD ? 5 * NT * KCP : 5 * KC * NT;
    __shared__ __attribute__((aligned(16))) float smem[IN_FLOATS + W_FLOATS];
    float* lds_in = smem;
    float* lds_w = smem + IN_FLOATS;
    static_assert(IN_FLOATS % 4 == 0, "weight slab stays 16-byte aligned");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int ya = (blockIdx.x & 1) * 2, img0 = (blockIdx.x >> 1) * 32, n0 = blockIdx.y * NT;   // image rows ya, ya + 1
    const int r0 = ya == 0 ? 1 : 0, r1 = ya == 0 ? 4 : 3;              // kernel rows that at least one of the two image rows needs
    // image column of this wave: w in the first image row, w ^ 1 in the second -- columns 0 / 3 have three taps inside the image,
    // columns 1 / 2 four, so every wave carries 3 + 4 = 7 taps per stage in which both rows run (8 on two of the waves otherwise)
    const int aBase = lh * PS + li;
    const int bBase = DGRAD ? (li * KCP + lh) : (lh * NT + li);

    f32x16 acc[2][NB];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][nb][v] = 0.f;

    constexpr int WQ = 5 * KC * NT / 4, WPT = (WQ + 255) / 256;
    f32x4 wreg[WPT];
    auto load_w = [&](int cc, int r) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int q = tid + i * 256;
            if (WQ % 256 == 0 || q < WQ) {
                const float* src;
                if (!DGRAD) {
                    const int row = q / (NT / 4), c4 = q % (NT / 4);
                    const int s = row / KC, kc = row % KC;
                    src = a.w + (size_t)((r * 5 + s) * KCH + cc * KC + kc) * NCH + n0 + c4 * 4;
                } else {
                    const int c4 = q % QPP, rown = q / QPP;
                    const int n = rown % NT, s = rown / NT;
                    src = a.w + (size_t)((24 - (r * 5 + s)) * NCH + n0 + n) * KCH + cc * KC + c4 * 4;
                }
                wreg[i] = *reinterpret_cast<const f32x4*>(src);
            }
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int q = tid + i * 256;
            if (WQ % 256 == 0 || q < WQ) {
                if (!DGRAD) {
                    *reinterpret_cast<f32x4*>(lds_w + q * 4) = wreg[i];
                } else {
                    const int c4 = q % QPP, rown = q / QPP;
                    float* d = lds_w + rown * KCP + c4 * 4;
                    d[0] = wreg[i].x; d[1] = wreg[i].y; d[2] = wreg[i].z; d[3] = wreg[i].w;
                }
            }
        }
    };
    constexpr int NQ = 512 * QPP, IPT = NQ / 256;              // 32 images x 16 pixels x KC channels
    f32x4 ireg[IPT];
    auto load_input = [&](int cc) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256, c4 = q % QPP, hp = q / QPP, ib = img0 + (hp >> 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ib < a.B) v = *reinterpret_cast<const f32x4*>(a.in + ((size_t)ib * 16 + (hp & 15)) * KCH + cc * KC + c4 * 4);
            ireg[i] = v;
        }
    };
    auto store_input = [&]() {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256, c4 = q % QPP, hp = q / QPP;
            float* d = lds_in + (c4 * 4) * PS + (hp & 15) * PXS + (hp >> 4);
            d[0] = ireg[i].x; d[PS] = ireg[i].y; d[2 * PS] = ireg[i].z; d[3 * PS] = ireg[i].w;
        }
    };

    static_assert((KCH / KC) % KSPLIT == 0, "split-K must divide the channel chunks");
    constexpr int CPS = (KCH / KC) / KSPLIT;
    const int c0 = blockIdx.z * CPS, c1 = c0 + CPS;
    load_w(c0, r0);
    load_input(c0);
    for (int cc = c0; cc < c1; ++cc)
        for (int r = r0; r <= r1; ++r) {
            __syncthreads();                       // everyone finished reading the previous stage
            if (r == r0) store_input();
            store_w();
            // the (older) input loads must not sit between a weight load and the store_w that waits for it
            if (r == r0 && cc + 1 < c1) load_input(cc + 1);
            if (r < r1) load_w(cc, r + 1);
            else if (cc + 1 < c1) load_w(cc + 1, r0);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int yi = ya + t + r - 2;                 // input row of image row ya + t under kernel row r
                if ((unsigned)yi > 3u) continue;               // workgroup-uniform
                const int xc = wave ^ t;                       // taps whose input column xc+s-2 exists: s0..s1
                const int s0 = xc < 2 ? 2 - xc : 0, s1 = xc > 1 ? 5 - xc : 4;
                const float* ap = lds_in + aBase + (yi * 4 + xc - 2) * PXS;
                for (int s = s0; s <= s1; ++s) {
                    const float* as = ap + s * PXS;
                    const float* bs = lds_w + bBase + (DGRAD ? s * NT * KCP : s * KC * NT);
#pragma unroll
                    for (int j = 0; j < KC / 2; ++j) {
                        const float av = as[(2 * j) * PS];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const float bv = DGRAD ? bs[nb * 32 * KCP + 2 * j] : bs[(2 * j) * NT + nb * 32];
                            acc[t][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t][nb], 0, 0, 0);
                        }
                    }
                }
            }
        }
    // raw partial sums of slab z: row (image) (v&3) + 8 (v>>2) + 4 lh of the wave's 32, pixel (ya + t, wave ^ t), channel n0 + 32 nb + li
    vm_drained();
    float* out = a.out + (size_t)blockIdx.z * a.sliceFloats;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int ib = img0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                if (ib < a.B) out[((size_t)ib * 16 + (ya + t) * 4 + (wave ^ t)) * NCH + n0 + nb * 32 + li] = acc[t][nb][v];
            }
}

template <int KCH, int NCH, bool DGRAD, int NT, int KSPLIT>
static int run4x4(const ConvArgs& a, hipStream_t st) {
    static_assert(KCH % KChunk<NT>::KC == 0 && NCH % NT == 0, "channel tiling");
    cvae_probe_begin(st);
    hipLaunchKernelGGL((conv4x4_row_kernel<KCH, NCH, DGRAD, NT, KSPLIT>), dim3(cdiv(a.B, 32) * 2, NCH / NT, KSPLIT), dim3(256), 0, st, a);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    return 0;
}

// out[i] = relu(bias[i % C] + sum_z slab[z][i])   (split-K finish of the decoder head)
template <typename AT>
__global__ __launch_bounds__(256) void splitk_bias_relu_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                               float* __restrict__ out, int64_t n4, int64_t slice, int KS, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 acc = *reinterpret_cast<const float4*>(bias + (i * 4) % C);
    for (int z = 0; z < KS; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)z * slice + i * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    Act<AT>::st4(out, (size_t)i * 4, f32x4{acc.x, acc.y, acc.z, acc.w});
}

int launch_splitk_bias_relu(const float* slab, const float* bias, float* out, int64_t slice, int KS, int C, hipStream_t st, bool out_bf16) {
    if (out_bf16) hipLaunchKernelGGL(splitk_bias_relu_kernel<__bf16>, dim3((unsigned)((slice / 4 + 255) / 256)), dim3(256), 0, st, slab, bias, out,
                                     slice / 4, slice, KS, C);
    else hipLaunchKernelGGL(splitk_bias_relu_kernel<float>, dim3((unsigned)((slice / 4 + 255) / 256)), dim3(256), 0, st, slab, bias, out,
                       slice / 4, slice, KS, C);
    CVAE_CHECK_LAUNCH();
    return 0;
}

#ifndef D0_KS
#define D0_KS 8
#endif
#ifndef D0_DKS
#define D0_DKS 4
#endif
static constexpr int D0_KSPLIT = D0_KS;
int64_t conv_fwd_ws_floats(int layer, int width, int B) {
    if (layer != 4 || width != 64) return 0;
    const int64_t h = kLayers[4].h * (width / 64);
    return (int64_t)D0_KSPLIT * B * h * h * kLayers[4].cout;
}

// conv_mfma_ps.hip: persistent form of E2..E4 forward / input gradient.  CVAE_CONVF_PS = bit mask of the layers that run on it
// (bit layer-1 forward, bit 3+layer-1 input gradient; 0 = none, for A/B runs).
int launch_conv_mfma_ps(int layer, int width, bool dgrad, int B, const float* in, const float* w, const float* bias, float* out, float* bnpart, hipStream_t st);
#ifndef CONVF_PS_DEFAULT
#define CONVF_PS_DEFAULT 55      // the 64-channel-tile layers: E2, E3 forward, E3 input gradient (-2.5 % each), and — round 5, once the kernel's operands travelled
                                 // as buffer loads — E4 forward / input gradient on 64-channel tiles (step 87.90 -> 88.32 k img/s, masks alternating on one box,
                                 // profiles/r05_h_ps_buffer_loads.txt); the 32-channel-tile instantiations (E2 input gradient) spill and lose
#endif
static bool use_f32_ps(int layer, bool dgrad) {
    static const int mask = [] { const char* e = getenv("CVAE_CONVF_PS"); return e ? atoi(e) : CONVF_PS_DEFAULT; }();
    return ((mask >> ((dgrad ? 3 : 0) + (layer >= 4 ? 2 : layer - 1))) & 1) != 0;
}

int conv_f32_route(int layer, int width, bool dgrad, int B) {
    if (!use_f32_ps(layer, dgrad)) return 0;
    g_conv_dry = true;
    const int rc = launch_conv_mfma_ps(layer, width, dgrad, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    g_conv_dry = false;
    return rc == 0 ? 1 : 0;
}

int launch_conv_fwd(int layer, int width, int B, const float* in, const float* w, const float* bias,
                    float* out, float* bnpart, float* ws, hipStream_t st) {
    if (use_f32_ps(layer, false)) { const int rc = launch_conv_mfma_ps(layer, width, false, B, in, w, bias, out, bnpart, st); if (rc != -100) return rc; }
    ConvArgs a{in, w, bias, nullptr, out, bnpart, B, 0};
    if (width == 64) {
        switch (layer) {
            case 1: return run<32, 64, 32, false, false, 64, EPI_BIAS_BNSTAT>(a, st);
            case 2: return run<64, 128, 16, false, false, 64, EPI_BIAS_BNSTAT>(a, st);
            case 3: return run<128, 256, 8, false, false, 32, EPI_BIAS_BNSTAT>(a, st);
            case 4: {     // D0: 4x4 images, K = 6400 -> split-K over channel chunks to fill the chip
                const int64_t slice = (int64_t)B * 4 * 4 * 128;
                a.out = ws; a.sliceFloats = slice;
                int rc = run4x4<256, 128, false, 64, D0_KSPLIT>(a, st);
                if (rc) return rc;
                hipLaunchKernelGGL(splitk_bias_relu_kernel<float>, dim3((unsigned)((slice / 4 + 255) / 256)), dim3(256), 0, st,
                                   ws, bias, out, slice / 4, slice, D0_KSPLIT, 128);
                CVAE_CHECK_LAUNCH();
                return 0;
            }
        }
    }
    if (width == 128) {
        switch (layer) {
            case 1: return run<32, 64, 64, false, false, 64, EPI_BIAS_BNSTAT>(a, st);
            case 2: return run<64, 128, 32, false, false, 64, EPI_BIAS_BNSTAT>(a, st);
            case 3: return run<128, 256, 16, false, false, 64, EPI_BIAS_BNSTAT>(a, st);
            case 4: return run<256, 128, 8, false, false, 64, EPI_BIAS_RELU>(a, st);
        }
    }
    cvae_set_error("conv_fwd: unsupported layer %d at width %d", layer, width);
    return -2;
}

// D0 dgrad at 4x4 images is 256 workgroups only: split-K x2 over the co chunks when the caller
// provides scratch (the training step does; the single-op entry point runs the unsplit kernel)
static constexpr int D0_DGRAD_KSPLIT = D0_DKS;
int64_t conv_dgrad_ws_floats(int layer, int width, int B) {
    if (layer != 4 || width != 64) return 0;
    return (int64_t)D0_DGRAD_KSPLIT * B * 16 * kLayers[4].cin;
}

int launch_conv_dgrad(int layer, int width, int B, const float* dout, const float* w,
                      const float* mask_src, float* din, float* ws, hipStream_t st) {
    // KCH = layer Cout (channels of dout), NCH = layer Cin (channels of din)
    if (!mask_src && use_f32_ps(layer, true)) { const int rc = launch_conv_mfma_ps(layer, width, true, B, dout, w, nullptr, din, nullptr, st); if (rc != -100) return rc; }
    ConvArgs a{dout, w, nullptr, mask_src, din, nullptr, B, 0};
    if (width == 64 && layer == 4 && ws != nullptr) {
        const int64_t slice = (int64_t)B * 16 * 256;
        a.out = ws; a.sliceFloats = slice;
        int rc = run4x4<128, 256, true, 64, D0_DGRAD_KSPLIT>(a, st);
        if (rc) return rc;
        return launch_reduce_slabs(ws, din, slice, D0_DGRAD_KSPLIT, slice, st, nullptr);
    }
    if (width == 64) {
        switch (layer) {
            case 1: return run<64, 32, 32, false, true, 32, EPI_PLAIN>(a, st);
            case 2: return run<128, 64, 16, false, true, 64, EPI_PLAIN>(a, st);
            case 3: return run<256, 128, 8, false, true, 32, EPI_PLAIN>(a, st);
            case 4: return run<128, 256, 4, false, true, 32, EPI_PLAIN>(a, st);
        }
    }
    if (width == 128) {
        switch (layer) {
            case 1: return run<64, 32, 64, false, true, 32, EPI_PLAIN>(a, st);
            case 2: return run<128, 64, 32, false, true, 64, EPI_PLAIN>(a, st);
            case 3: return run<256, 128, 16, false, true, 64, EPI_PLAIN>(a, st);
            case 4: return run<128, 256, 8, false, true, 64, EPI_PLAIN>(a, st);
        }
    }
    cvae_set_error("conv_dgrad: unsupported layer %d at width %d", layer, width);
    return -2;
}
