// conv_wgrad_split.hip — weight gradients of the fp32-emulation modes on the bf16 MFMA (round 4).
#include "common.h"
#include <stdlib.h>
#include "conv_bf16.h"

// ---------------------------------------------------------------------------------------------
// Weight gradients of E2..E4 / D0 in the fp32-EMULATION modes (precision 2 / 3; round 4): the transposed-read kernel of conv_bf16.hip
// (conv5x5_wgrad_tr_kernel) with both fp32 operands split exactly into three bf16 parts while they are staged (x = hi + mid + lo, split3), three LDS images per
// operand, and 9 (DMAX = 4) or the 6 leading (DMAX = 2) bf16 x bf16 partial products per 16-pixel block, smallest first, into one
// fp32 accumulator — exactly what conv5x5_bf16_kernel<NS = 3> does for the forward / input-gradient passes, so all
// three passes of these layers run on the bf16 MFMA in those modes (E1, D1..D3 and D4 keep their fp32 kernels).  Every partial product is exact in the accumulator; what differs from the fp32-MFMA
// weight-gradient kernel (conv_wgrad.hip) is the order of the additions.  The bias gradient is ones x (hi + mid + lo).
// 128-pixel tiles (64 on 4x4 images): 3 x (halo + tile) x 64 B = 62-80 KB of LDS, two workgroups per CU; four waves, wave w owns taps
// w, w+4, .., w+20 and every 4th pixel group of tap 24 and of the bias row (8 accumulator tiles).  One staged fragment triple serves
// 6 / 9 MFMAs: 0.6 / 0.4 KB of LDS reads per MFMA where the plain bf16 kernel needs 2 KB — this kernel is MFMA-bound.
// ---------------------------------------------------------------------------------------------
#ifndef SPLIT_PIPE
#define SPLIT_PIPE 0      // 1 (experiment): the next tap's fragment triple is requested before this tap's MFMAs — 12 more live registers, 20-76 B of
#endif                    // spills at the 256-register budget, 131 / 127.5 / 127 -> 140 / 130 / 131 us (E2 / E3 / E4, six products, B = 256)
template <int H> using SplitTile = WtTile<H, 2, (H == 4 ? 64 : 128)>;

// W = wave index as a RUNTIME (wave-uniform) value: the six tap offsets of the wave sit in registers, everything else of an LDS
// address is a compile-time immediate — one body instead of four (the four-way switch of the bf16 kernel quadruples the compile time
// of these 6 / 9-MFMA blocks for nothing)
template <int H, int DMAX>
__device__ __forceinline__ void wgrad_split_body(f32x16 (&acc)[8], const __bf16* lds_in, const __bf16* lds_d, int ibase, int dbase, bf16x8 ones, int W) {
    using T = SplitTile<H>;
    constexpr int JT = 6, IN_IMG = T::HP * 32, D_IMG = T::NPX * 32;
    static_assert(T::KG % 4 == 0, "pixel groups per tile must split evenly over the four waves");
    const __bf16* tapp[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) { const int tap = 4 * j + W; tapp[j] = lds_in + ibase + ((tap / 5) * T::HTW + tap % 5) * 32; }
    const __bf16* tap24 = lds_in + ibase + (4 * T::HTW + 4) * 32;
    auto load_a = [&](bf16x8 (&av)[3], const __bf16* q0) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) av[sp] = tr_frag(q0 + sp * IN_IMG, q0 + sp * IN_IMG + T::IT * 32);
    };
    auto mfmas = [&](f32x16& c, const bf16x8 (&av)[3], const bf16x8 (&bv)[3]) {
#pragma unroll
        for (int d = DMAX; d >= 0; --d)               // 0 = hi, 1 = mid, 2 = lo; d = ia + ib: smallest products first
#pragma unroll
            for (int ia = 0; ia < 3; ++ia) {
                const int ib = d - ia;
                if (ib >= 0 && ib < 3) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ia], bv[ib], c, 0, 0, 0);
            }
    };
#pragma unroll
    for (int kg = 0; kg < T::KG; ++kg) {
        bf16x8 bv[3], av[2][3];
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) {
            const __bf16* dp = lds_d + sp * D_IMG + dbase + T::pixbase(kg) * 32;
            bv[sp] = tr_frag(dp, dp + T::DT * 32);
        }
        load_a(av[0], tapp[0] + T::halobase(kg) * 32);
        // the compiler's own order is read -> wait -> 6 / 9 MFMAs per tap; the co-resident workgroup's wave covers the LDS latency
#pragma unroll
        for (int j = 0; j < JT; ++j) {
#if SPLIT_PIPE
            if (j + 1 < JT) load_a(av[(j + 1) & 1], tapp[j + 1] + T::halobase(kg) * 32);
            else if ((kg % 4) == W) load_a(av[(j + 1) & 1], tap24 + T::halobase(kg) * 32);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(acc[j], av[j & 1], bv);
            __builtin_amdgcn_sched_barrier(0);
#else
            if (j > 0) load_a(av[j & 1], tapp[j] + T::halobase(kg) * 32);
            mfmas(acc[j], av[j & 1], bv);
#endif
        }
        if ((kg % 4) == W) {
#if !SPLIT_PIPE
            load_a(av[JT & 1], tap24 + T::halobase(kg) * 32);
#endif
            mfmas(acc[JT], av[JT & 1], bv);
        }
        if ((kg % 4) == ((W + 1) & 3)) {
#pragma unroll
            for (int sp = 2; sp >= 0; --sp) acc[JT + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bv[sp], acc[JT + 1], 0, 0, 0);
        }
    }
}

template <int CIN, int COUT, int H, int DMAX>
__global__ __launch_bounds__(256, 2) void conv5x5_wgrad_split_kernel(WgradBf16Args a) {          // a.in / a.dout: fp32 tensors
    using T = SplitTile<H>;
    constexpr int JT = 6, IN_IMG = T::HP * 32, D_IMG = T::NPX * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* lds_in = reinterpret_cast<__bf16*>(smem_raw);       // 3 x [halo pixel][32 ci] (hi | mid | lo), 64-byte rows
    __bf16* lds_d = lds_in + 3 * IN_IMG;                        // 3 x [pixel][32 co]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x, ci0 = blockIdx.y * 32, n0 = blockIdx.z * 32;
    const int g = lane >> 4, h = g >> 1, laneoff = ((lane & 15) >> 2) * 32 + 16 * (g & 1) + 4 * (lane & 3);
    const int ibase = h * T::IH * 32 + laneoff, dbase = h * T::DH * 32 + laneoff;

    f32x16 acc[JT + 2];
#pragma unroll
    for (int j = 0; j < JT + 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int c = 0; c < 8; ++c) ones[c] = (__bf16)1.f;

    constexpr int IU = T::HP * 4, DU = T::NPX * 4, NI = (IU + 255) / 256, ND = (DU + 255) / 256;
    f32x4 rin[NI][2], rdo[ND][2];
    unsigned okin = 0, okd = 0;                                  // validity bits: the zero select happens at the LDS store (the loads stay in flight)
    auto fetch = [&](int mt) {
        const int grp = mt / T::TPI, t = mt % T::TPI;
        const int img0 = grp * T::IMGS, ty0 = (t / T::TILES_X) * T::TH, tx0 = (t % T::TILES_X) * T::TW;
        okin = 0; okd = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = tid + i * 256, hp = q >> 2, oc = q & 3;
            const int img = hp / T::HPI, rem = hp % T::HPI;
            const int gy = ty0 + rem / T::HTW - 2, gx = tx0 + rem % T::HTW - 2, ib = img0 + img;
            const bool ok = (IU % 256 == 0 || q < IU) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)H && ib < a.B;
            const size_t e = ok ? ((size_t)(ib * H + gy) * H + gx) * CIN + ci0 + oc * 8 : 0;
            rin[i][0] = *reinterpret_cast<const f32x4*>(a.in + e);
            rin[i][1] = *reinterpret_cast<const f32x4*>(a.in + e + 4);
            okin |= (unsigned)ok << i;
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int q = tid + i * 256, px = q >> 2, oc = q & 3;
            const int img = px / (T::TH * T::TW), rem = px % (T::TH * T::TW);
            const int gy = ty0 + rem / T::TW, gx = tx0 + rem % T::TW, ib = img0 + img;
            const bool ok = (DU % 256 == 0 || q < DU) && ib < a.B;
            const size_t e = ok ? ((size_t)(ib * H + gy) * H + gx) * COUT + n0 + oc * 8 : 0;
            rdo[i][0] = *reinterpret_cast<const f32x4*>(a.dout + e);
            rdo[i][1] = *reinterpret_cast<const f32x4*>(a.dout + e + 4);
            okd |= (unsigned)ok << i;
        }
    };
    auto store3 = [&](__bf16* img0, int imgStride, int q, f32x4 lo4, f32x4 hi4, bool ok) {
        bf16x8 p[3];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float x = ok ? (c < 4 ? lo4[c] : hi4[c - 4]) : 0.f;
            const Split3 sp = split3(x);
            p[0][c] = sp.hi; p[1][c] = sp.mid; p[2][c] = sp.lo;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<bf16x8*>(img0 + (size_t)k * imgStride + (size_t)q * 8) = p[k];
    };
    const int t0 = split * a.tilesPerSplit;
    int t1 = t0 + a.tilesPerSplit; if (t1 > a.numTiles) t1 = a.numTiles;
    if (t0 < t1) fetch(t0);
    for (int mt = t0; mt < t1; ++mt) {
        __syncthreads();                        // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = tid + i * 256;
            if (IU % 256 == 0 || q < IU) store3(lds_in, IN_IMG, q, rin[i][0], rin[i][1], (okin >> i) & 1);
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int q = tid + i * 256;
            if (DU % 256 == 0 || q < DU) store3(lds_d, D_IMG, q, rdo[i][0], rdo[i][1], (okd >> i) & 1);
        }
        __syncthreads();
        if (mt + 1 < t1) fetch(mt + 1);         // in flight while this tile computes
        __builtin_amdgcn_sched_barrier(0);
        wgrad_split_body<H, DMAX>(acc, lds_in, lds_d, ibase, dbase, ones, __builtin_amdgcn_readfirstlane(wave));
    }

    float* out = a.slab + (size_t)split * (25 * CIN * COUT + COUT);     // slab row: [25][CIN][COUT] | bias[COUT]
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int ci = ci0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            out[((size_t)(4 * j + wave) * CIN + ci) * COUT + n0 + li] = acc[j][v];
        }
    // tap 24 and the bias row: every wave holds the partial of its pixel groups -> fixed-order sum through LDS
    float* red = reinterpret_cast<float*>(smem_raw);                    // [3 waves][16][64]
    float* bred = red + 3 * 16 * 64;                                    // [4 waves][32]
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) red[((wave - 1) * 16 + v) * 64 + lane] = acc[JT][v];
    }
    if (lh == 0) bred[wave * 32 + li] = acc[JT + 1][0];                 // row 0 of (ones x dy) = column sums of dy
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float x = acc[JT][v];
#pragma unroll
            for (int w2 = 0; w2 < 3; ++w2) x += red[(w2 * 16 + v) * 64 + lane];
            const int ci = ci0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            out[((size_t)24 * CIN + ci) * COUT + n0 + li] = x;
        }
        if (blockIdx.y == 0 && lh == 0) {
            float b = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) b += bred[w2 * 32 + li];
            out[(size_t)25 * CIN * COUT + n0 + li] = b;
        }
    }
}

template <int CIN, int COUT, int H>
static int run_wgrad_split(int products, int B, const float* in, const float* dout, float* dw, float* dbias, float* ws, hipStream_t st,
                           int64_t* need) {
    using T = SplitTile<H>;
    const int numTiles = cdiv(B, T::IMGS) * T::TPI;
    int S = cdiv(2 * cvae_num_cus(), (CIN / 32) * (COUT / 32));        // two workgroups per CU
    if (S > numTiles) S = numTiles;
    if (S < 1) S = 1;
    const int tps = cdiv(numTiles, S);
    S = cdiv(numTiles, tps);
    const int64_t n = (int64_t)25 * CIN * COUT, row = n + COUT;
    if (need) { *need = (int64_t)(S + 16) * row; return 0; }
    WgradBf16Args a{in, dout, ws, B, numTiles, tps};
    constexpr int STAGE = 3 * (T::HP + T::NPX) * 64, RED = (3 * 16 * 64 + 4 * 32) * 4;
    constexpr int SMEM = STAGE > RED ? STAGE : RED;
    static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");
    void (*kern)(WgradBf16Args) = products == 6 ? conv5x5_wgrad_split_kernel<CIN, COUT, H, 2> : conv5x5_wgrad_split_kernel<CIN, COUT, H, 4>;
    static DeviceOnce once[2];
    { int rc = cvae_grant_lds(once[products == 6], reinterpret_cast<const void*>(kern), SMEM); if (rc) return rc; }
    cvae_probe_begin(st);
    hipLaunchKernelGGL(kern, dim3(S, CIN / 32, COUT / 32), dim3(256), SMEM, st, a);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    float* mid = ws + (size_t)S * row;
    st = cvae_reduce_stream(st);
    if (dbias == dw + n) return launch_reduce_slabs(ws, dw, row, S, row, st, mid);
    int rc = launch_reduce_slabs(ws, dw, n, S, row, st, mid);
    if (rc || !dbias) return rc;
    return launch_reduce_slabs(ws + n, dbias, COUT, S, row, st, nullptr);
}

static int dispatch_wgrad_split(int layer, int width, int products, int B, const float* in, const float* dout, float* dw, float* dbias,
                                float* ws, hipStream_t st, int64_t* need) {
    if (width == 64) {
        switch (layer) {
            case 1: return run_wgrad_split<32, 64, 32>(products, B, in, dout, dw, dbias, ws, st, need);
            case 2: return run_wgrad_split<64, 128, 16>(products, B, in, dout, dw, dbias, ws, st, need);
            case 3: return run_wgrad_split<128, 256, 8>(products, B, in, dout, dw, dbias, ws, st, need);
            case 4: return run_wgrad_split<256, 128, 4>(products, B, in, dout, dw, dbias, ws, st, need);
        }
    } else if (width == 128) {
        switch (layer) {
            case 1: return run_wgrad_split<32, 64, 64>(products, B, in, dout, dw, dbias, ws, st, need);
            case 2: return run_wgrad_split<64, 128, 32>(products, B, in, dout, dw, dbias, ws, st, need);
            case 3: return run_wgrad_split<128, 256, 16>(products, B, in, dout, dw, dbias, ws, st, need);
            case 4: return run_wgrad_split<256, 128, 8>(products, B, in, dout, dw, dbias, ws, st, need);
        }
    }
    cvae_set_error("conv_wgrad_split: unsupported layer %d at width %d", layer, width);
    return -2;
}
bool conv_wgrad_split_supported(int products) { return products == 6 || products == 9; }
int64_t wgrad_split_ws_floats(int layer, int width, int B) {
    int64_t need = 0;
    if (dispatch_wgrad_split(layer, width, 6, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &need) != 0) return 0;
    return need;
}
// products: 9 (precision 2) or 6 (precision 3); in / dout are the fp32 tensors of those modes
int launch_conv_wgrad_split(int layer, int width, int products, int B, const float* in, const float* dout, float* dw, float* dbias,
                            float* ws, hipStream_t st) {
    return dispatch_wgrad_split(layer, width, products, B, in, dout, dw, dbias, ws, st, nullptr);
}

