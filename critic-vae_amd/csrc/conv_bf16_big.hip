// conv_bf16_big.hip — bf16-MFMA 5x5 conv forward / input-gradient kernel on a 4 x 4 WAVE TILE (round 4; CVAE_BF16_BIG bits 2..5).
//
// Same call sites as conv5x5_bf16_kernel<.., MODE_STD, NS = 1> (nn.Conv2d E3 / E4, vae_nets.py:79,84, and their input gradients under
// loss.backward(), vae.py:57), same packed weights, same LDS images, same k order.  What differs is the register tile: a workgroup
// owns FOUR 128-pixel tiles x 128 channels, every wave 32 pixels of each tile x all 128 channels = 16 accumulator tiles (256 AGPRs),
// so one k-step reads 4 + 4 fragments for 16 MFMAs — 0.5 KB of LDS per MFMA where the two-tile / 64-channel kernels read 1 KB and
// are bound by exactly that ("Round 4").  One workgroup per CU, one wave per SIMD: nothing else is resident to hide
// the staging, so the staging lives INSIDE the MFMA stream, with few registers live at a time:
//   * the weight slab is double-buffered in LDS; the slab of stage st + 1 is requested at step 0 of stage st and written into the
//     other buffer at steps 5 .. 9 (two 16-byte units per step), one barrier per stage;
//   * the next chunk's input tiles are requested in the last stage of a chunk and written behind a barrier at its end (the tiles
//     are single-buffered: 4 x 19 KB + 2 x 41 KB = 157 KB of LDS);
//   * fragments of step i + 1 are requested at the top of step i (two register sets).
// Epilogues, both from channel-major accumulators (v_cvt_pk + two 16-byte stores per tile, no transpose): plain (input gradient), or
// bias + ONE BatchNorm partial per workgroup (its four tiles summed in the lane first).  Shipped for E4 (forward and input gradient);
// the E3 instantiations are slower than the two-workgroup kernels and stay off (conv_bf16.hip, BF16_BIG_DEFAULT).
#include "common.h"
#include <stdlib.h>
#include "conv_epilogue.h"
#include "conv_bf16.h"
#ifndef BIG_PF
#define BIG_PF 1          // fragment sets requested ahead of the MFMAs that use them
#endif
#ifndef BIG_FENCE
#define BIG_FENCE 1       // a scheduling fence after every step
#endif

template <int KCH, int NCH, int H, int NT, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv5x5_bf16_big_kernel(ConvBf16Args a) {
    using T = Tile<H>;
    static_assert(EPI == EPI_PLAIN || EPI == EPI_BIAS_BNSTAT, "epilogues: plain (input gradient) or bias + BatchNorm partials (forward)");
    constexpr bool BN = EPI == EPI_BIAS_BNSTAT;
    constexpr int MT = 4, NB = NT / 32, KS = 5, KCB = 32, KB = KCB / 16, OCT = KCB / 8;
    constexpr int PSP = Bf16Geom<H, OCT>::PSP, A_UNITS = OCT * PSP, W_UNITS = KS * KB * 2 * NT;
    constexpr int NCHUNK = KCH / KCB, NST = NCHUNK * KS, NSTEP = KS * KB;
    static_assert(KCH % KCB == 0 && NCH % NT == 0 && W_UNITS % (5 * 256) == 0 && NSTEP == 10, "tiling");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* lds_a = reinterpret_cast<bf16x8*>(smem_raw);       // [tile][octet][halo pixel]
    bf16x8* lds_w = lds_a + MT * A_UNITS;                      // [buffer][tap][kb][half][n]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int mt0 = xcd_tile(blockIdx.x, gridDim.x) * MT, n0 = blockIdx.y * NT;
    int img0v[MT], ty0v[MT], tx0v[MT];
#pragma unroll
    for (int tl = 0; tl < MT; ++tl) {          // a tile index past the end maps to images >= B: loads give 0, stores are skipped
        const int tileInImg = (mt0 + tl) % T::TILES_PER_IMG;
        img0v[tl] = ((mt0 + tl) / T::TILES_PER_IMG) * T::IMGS;
        ty0v[tl] = (tileInImg / T::TILES_X) * T::TH; tx0v[tl] = (tileInImg % T::TILES_X) * T::TW;
    }
    const int m = wave * 32 + lane_pix<H, true>(li);
    const int pimg = m / (T::TH * T::TW), prem = m % (T::TH * T::TW);
    const int aPix = pimg * T::HPI + (prem / T::TW) * T::HTW + (prem % T::TW);

    f32x16 acc[MT][NB];
#pragma unroll
    for (int tl = 0; tl < MT; ++tl)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[tl][nb][v] = 0.f;

    // staging tables (fixed for the launch): weight unit q = tid + 256 i of a slab, input unit q of a tile
    constexpr int WPT = W_UNITS / 256;
    int wbase[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int q = tid + i * 256;
        const int n = q % NT, row = q / NT, half = row & 1, kb = (row >> 1) % KB, s = row / (2 * KB);
        wbase[i] = ((s * (KCH / 16) + kb) * 2 + half) * NCH + n0 + n;
    }
    constexpr int NQ = T::HP * OCT, IPT = (NQ + 255) / 256;
    static_assert(MT * IPT <= 32, "one validity bit per staged unit");
    int ebase[MT * IPT];
#pragma unroll
    for (int tl = 0; tl < MT; ++tl)
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256;
            const int oct = q % OCT, hp = q / OCT;
            const int img = hp / T::HPI, rem = hp - img * T::HPI;
            const int gy = ty0v[tl] + rem / T::HTW - 2, gx = tx0v[tl] + rem % T::HTW - 2, ib = img0v[tl] + img;
            const bool ok = (NQ % 256 == 0 || q < NQ) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)H && ib < a.B;
            ebase[tl * IPT + i] = ok ? ((ib * H + gy) * H + gx) * KCH + oct * 8 : -1;
        }
    // forward: the bias of this thread's channel, requested ahead of every other load (consumed after the loop, before the first store)
    [[maybe_unused]] float bias_stash = 0.f;
    if constexpr (BN) { if (tid < NT) bias_stash = a.bias[n0 + tid]; }
    bf16x8 wreg[WPT], breg[MT * IPT];
    bf16x8 z8;
#pragma unroll
    for (int k = 0; k < 8; ++k) z8[k] = (__bf16)0.f;
    auto ldw = [&](int i, int st) {
        const int cc = st / KS, r = st % KS;
        wreg[i] = a.wp[(size_t)(r * KS * (KCH / 16) + cc * KB) * 2 * NCH + wbase[i]];
    };
    auto stw = [&](int i, int buf) { lds_w[buf * W_UNITS + tid + i * 256] = wreg[i]; };
    auto ldin = [&](int j, int cc) {            // raw value from a clamped address; the zero padding is selected at the LDS store
        const int e = ebase[j];
        breg[j] = Act<__bf16>::ld8(a.in, e >= 0 ? (size_t)(e + cc * KCB) : 0);
    };
    auto stin = [&](int j) {
        const int tl = j / IPT, i = j % IPT, q = tid + i * 256;
        if (NQ % 256 == 0 || q < NQ) lds_a[tl * A_UNITS + (q % OCT) * PSP + q / OCT] = ebase[j] >= 0 ? breg[j] : z8;
    };

    // prologue: chunk 0's tiles and slab 0
#pragma unroll
    for (int i = 0; i < WPT; ++i) ldw(i, 0);
#pragma unroll
    for (int j = 0; j < MT * IPT; ++j) ldin(j, 0);
#pragma unroll
    for (int j = 0; j < MT * IPT; ++j) stin(j);
#pragma unroll
    for (int i = 0; i < WPT; ++i) stw(i, 0);
    __syncthreads();

    for (int st = 0; st < NST; ++st) {
        const int r = st % KS, buf = st & 1;
        const bool nextw = st + 1 < NST, nextin = r == KS - 1 && nextw;
        const bf16x8* ap = lds_a + lh * PSP + aPix + r * T::HTW;
        const bf16x8* bp = lds_w + buf * W_UNITS + lh * NT + li;
        constexpr int NSET = BIG_PF + 1;
        bf16x8 wf[NSET][NB], xf[NSET][MT];
        auto ldf = [&](int i, int b) {
            const int s = i / KB, kb = i % KB;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wf[b][nb] = bp[((s * KB + kb) * 2) * NT + nb * 32];
#pragma unroll
            for (int tl = 0; tl < MT; ++tl) xf[b][tl] = ap[tl * A_UNITS + (kb * 2) * PSP + s];
        };
#pragma unroll
        for (int i = 0; i < BIG_PF; ++i) ldf(i, i % NSET);
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            if (i + BIG_PF < NSTEP) ldf(i + BIG_PF, (i + BIG_PF) % NSET);
            if (i == 0 && nextw) {
#pragma unroll
                for (int k = 0; k < WPT; ++k) ldw(k, st + 1);
            }
            if (i == 1 && nextin) {
#pragma unroll
                for (int j = 0; j < MT * IPT; ++j) ldin(j, st / KS + 1);
            }
            if (i >= 5 && nextw) {
#pragma unroll
                for (int k = 0; k < WPT / 5; ++k) stw((WPT / 5) * (i - 5) + k, buf ^ 1);
            }
#pragma unroll
            for (int tl = 0; tl < MT; ++tl)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)          // weights as the A operand: D[channel][pixel] (channel-major, conv_epilogue.h)
                    acc[tl][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i % NSET][nb], xf[i % NSET][tl], acc[tl][nb], 0, 0, 0);
            if (BIG_FENCE) __builtin_amdgcn_sched_barrier(0);
        }
        if (nextin) {                               // chunk boundary: every wave is done with the tiles, the next chunk's take their place
            __syncthreads();
#pragma unroll
            for (int j = 0; j < MT * IPT; ++j) stin(j);
        }
        __syncthreads();                            // slab st + 1 (and the tiles) visible to all; everyone is done with slab st
    }
    vm_drained();

    // epilogue: lane = pixel m of each tile, element v of block nb = channel 32 nb + (v & 3) + 8 (v >> 2) + 4 lh (channel-major accumulators);
    // u[k] = channels 16 k + 8 lh .. + 7 of a 32-channel block.  Forward: bias, then ONE BatchNorm partial per workgroup and channel —
    // (sum, M2 about the mean) of its four tiles' valid pixels, which launch_bn_fwd_finalize(.., tilesPerPartial = 4) merges
    // (nn.BatchNorm2d train-mode statistics, vae_nets.py:75,80,85): the four tiles are added up in the lane first, so the cross-lane
    // column sums (half_wave_colsum16) run once per channel block instead of once per accumulator tile.
    bool validv[MT];
    int gyv[MT], gxv[MT], ibv[MT];
#pragma unroll
    for (int tl = 0; tl < MT; ++tl) {
        gyv[tl] = ty0v[tl] + prem / T::TW; gxv[tl] = tx0v[tl] + prem % T::TW; ibv[tl] = img0v[tl] + pimg;
        validv[tl] = ibv[tl] < a.B;
    }
    [[maybe_unused]] float* lds_x = reinterpret_cast<float*>(lds_w);      // [NT bias][S | Q][4 waves][NT] over the weight slabs (every wave is past the last barrier)
    if constexpr (BN) {
        if (tid < NT) lds_x[tid] = bias_stash;
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 bq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(lds_x + nb * 32 + 8 * g + 4 * lh);
#pragma unroll
            for (int tl = 0; tl < MT; ++tl)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[tl][nb][v] += bq[v >> 2][v & 3];
        }
    }
#pragma unroll
    for (int tl = 0; tl < MT; ++tl) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            bf16x8 u[2];
            cm_pack_units(acc[tl][nb], u);
            const size_t base = ((size_t)(ibv[tl] * H + gyv[tl]) * H + gxv[tl]) * NCH + n0 + nb * 32 + 8 * lh;
            if (validv[tl]) { Act<__bf16>::st8(a.out, base, u[0]); Act<__bf16>::st8(a.out, base + 16, u[1]); }
        }
    }
    if constexpr (BN) {
        float* red = lds_x + NT;
        const int e16 = li >> 1, chE = (e16 & 3) + 8 * (e16 >> 2) + 4 * lh;       // the element half_wave_colsum16 leaves in this lane
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float sv[16], qv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) { sv[v] = 0.f; qv[v] = 0.f; }
#pragma unroll
            for (int tl = 0; tl < MT; ++tl)
#pragma unroll
                for (int v = 0; v < 16; ++v) { const float x = validv[tl] ? acc[tl][nb][v] : 0.f; sv[v] += x; qv[v] = fmaf(x, x, qv[v]); }
            const float S = half_wave_colsum16(sv), Q = half_wave_colsum16(qv);
            if ((lane & 1) == 0) { red[(0 * 4 + wave) * NT + nb * 32 + chE] = S; red[(1 * 4 + wave) * NT + nb * 32 + chE] = Q; }
        }
        __syncthreads();
        if (tid < NT) {
            float S = 0.f, Q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { S += red[(0 * 4 + w) * NT + tid]; Q += red[(1 * 4 + w) * NT + tid]; }
            int cnt = 0;
#pragma unroll
            for (int tl = 0; tl < MT; ++tl) { int ni = a.B - img0v[tl]; ni = ni < 0 ? 0 : (ni > T::IMGS ? T::IMGS : ni); cnt += ni * T::TH * T::TW; }
            const int numBig = gridDim.x, big = mt0 / MT;
            const double m2 = cnt > 0 ? (double)Q - (double)S * (double)S / (double)cnt : 0.0;
            a.bnpart[(size_t)big * NCH + n0 + tid] = S;
            a.bnpart[((size_t)numBig + big) * NCH + n0 + tid] = (float)(m2 > 0.0 ? m2 : 0.0);
        }
    }
}

template <int KCH, int NCH, int H, int NT, int EPI>
static int run_big(const ConvBf16Args& a, hipStream_t st) {
    using T = Tile<H>;
    constexpr int SMEM = (4 * 4 * Bf16Geom<H, 4>::PSP + 2 * 5 * 2 * 2 * NT) * 16;
    static_assert(SMEM <= 160 * 1024, "LDS");
    auto kern = conv5x5_bf16_big_kernel<KCH, NCH, H, NT, EPI>;
    static DeviceOnce once;
    { int rc = cvae_grant_lds(once, reinterpret_cast<const void*>(kern), SMEM); if (rc) return rc; }
    dim3 grid(cdiv(cdiv(a.B, T::IMGS) * T::TILES_PER_IMG, 4), NCH / NT);
    cvae_probe_begin(st);
    hipLaunchKernelGGL(kern, grid, dim3(256), SMEM, st, a);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    return 0;
}

// which layers this file serves.  Input gradients: mask bit 0 = the 128-channel one (E4), bit 1 = the 64-channel one (E3).  Forward (bias +
// BatchNorm partials of FOUR tiles each — the caller must tell launch_bn_fwd_finalize): mask bit 0 = E3 (64 -> 128), bit 1 = E4 (128 -> 256)
bool conv_bf16_big_has(int layer, int width, bool dgrad, int mask) {
    if (width != 64 && width != 128) return false;
    if (dgrad) return (layer == 3 && (mask & 1)) || (layer == 2 && (mask & 2));
    return (layer == 2 && (mask & 1)) || (layer == 3 && (mask & 2));
}
// returns -100 when the layer has no instantiation
int launch_conv_bf16_big(int layer, int width, bool dgrad, int mask, const ConvBf16Args& a, hipStream_t st) {
    if (!conv_bf16_big_has(layer, width, dgrad, mask)) return -100;
    if (dgrad) {
        if (width == 64 && layer == 3) return run_big<256, 128, 8, 128, EPI_PLAIN>(a, st);
        if (width == 64 && layer == 2) return run_big<128, 64, 16, 64, EPI_PLAIN>(a, st);
        if (width == 128 && layer == 3) return run_big<256, 128, 16, 128, EPI_PLAIN>(a, st);
        if (width == 128 && layer == 2) return run_big<128, 64, 32, 64, EPI_PLAIN>(a, st);
    } else {
        if (width == 64 && layer == 2) return run_big<64, 128, 16, 128, EPI_BIAS_BNSTAT>(a, st);
        if (width == 64 && layer == 3) return run_big<128, 256, 8, 128, EPI_BIAS_BNSTAT>(a, st);
        if (width == 128 && layer == 2) return run_big<64, 128, 32, 128, EPI_BIAS_BNSTAT>(a, st);
        if (width == 128 && layer == 3) return run_big<128, 256, 16, 128, EPI_BIAS_BNSTAT>(a, st);
    }
    return -100;
}
