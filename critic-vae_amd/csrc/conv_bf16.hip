// conv_bf16.hip — bf16-MFMA variant of the 5x5 conv forward / input-gradient kernels
// (precision mode 1 of cvae_config; BASELINE.json configs 3-5 ask for a bf16 implicit GEMM).
//
// Same call sites as conv_mfma.hip (nn.Conv2d E2..E4 and D0, vae_nets.py:74,79,84,117, and their
// input gradients in loss.backward(), vae.py:57), same tiling (Tile<H>: 128 output pixels x NT
// channels per workgroup, wave = 32 pixels), same fp32 epilogues (conv_epilogue.h).  What changes:
//   * precision mode 1 (NS == 1): activations and activation gradients ARE bf16 in HBM (written by the producing
//     kernel's epilogue, RNE), so a 16-byte global load is already one LDS unit of 8 channels — no conversion
//     while staging; BatchNorm statistics (taken from the fp32 accumulators), loss scalars, master weights and the
//     flat gradient stay fp32.  The fp32-emulation modes (NS == 3) keep fp32 tensors and split them while staging;
//   * the contraction runs on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): one instruction covers
//     a 16-channel block of one tap — 8x fewer instructions at half the cycles each;
//   * a K stage = one kernel row x up to 64 channels; LDS holds 16-byte units of 8 channels,
//     [channel octet][pixel] for the input halo (tap shifts stay 16-byte aligned, lanes read
//     consecutive units) and [tap][octet][n] for the weight slab;
//   * weights are re-packed once per step by pack_w_bf16_kernel into that unit order, for the
//     forward orientation (k = ci) and the dgrad orientation (k = co, taps flipped), so ONE kernel
//     body serves both passes.
// The file also holds the bf16 weight-gradient kernels (k runs over images so that tap shifts keep
// 16-byte alignment) and the bf16 variants of the phase-collapsed up-convs of conv_up.hip.
// Results differ from the fp32 path at the bf16 rounding level (~3e-3 relative per product);
// tests/test_gpu_bf16.py states the tolerances.
#include "common.h"
#include <stdlib.h>
#include "conv_epilogue.h"
#include "conv_bf16.h"


// MODE_STD: 5x5 (or 3x3) conv of an NHWC tensor, epilogues of conv_epilogue.h.
// MODE_UP_FWD: Upsample(2)->Conv5x5 (vae_nets.py:119-131) as the phase-collapsed 3x3 conv of
//   conv_up.hip over the stored low-resolution tensor with N = 4 phases x COUT columns; the epilogue
//   adds the bias, applies ReLU and scatters column (p, co) of low-res pixel (y, x) to output pixel
//   (2y+py, 2x+px).
// MODE_UP_DGRAD: its input gradient: K = 4 phases x COUT channels gathered from dout by phase
//   (space-to-depth view), taps flipped in the packed weights, ReLU mask of the producer in the epilogue.
enum { MODE_STD = 0, MODE_UP_FWD = 1, MODE_UP_DGRAD = 2 };
#ifndef BF16_MT
#define BF16_MT 2          // 128-pixel tiles per workgroup of the plain bf16 kernels (1 = round-1 structure)
#endif
#ifndef BF16_BIG_DEFAULT
// CVAE_BF16_BIG: layers on the persistent big-tile kernel (conv_bf16_big.hip) — bit 2 = E4 input gradient, bit 3 = E3 input gradient, bit 4 = E3 forward,
// bit 5 = E4 forward, bit 6 = E2 forward and bit 7 = E2 input gradient at 64 x 64 (image-high items on an 8 x 1 wave tile; one BatchNorm partial per item of
// four / eight tiles); 0 = the two-workgroup / per-tile kernels (A/B runs).  Defaults from the un-profiled step on one box, masks alternating
// (profiles/r05_g_big_mask_sweep.txt, r05_k_big_image_layout.txt, r05_r_e2_on_big_kernel.txt): every bit at 64 x 64 (252: E2 forward 196 vs 209 us, E2 input gradient
// 192 vs 209 us, step +0.8 % over mask 60) and at 128 x 128 (E2's 64-row images as 16-row strips with real halo rows: input gradient 379 vs 415 us, forward 424 vs 431, step +0.9 %).
#define BF16_BIG_DEFAULT 252
#endif
#ifndef BF16_BIG_DEFAULT_W128
#define BF16_BIG_DEFAULT_W128 252
#endif
#ifndef BF16_WDMA
#define BF16_WDMA 0        // 1 (experiment, round 3): weight slabs travel HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) into a double
#endif                     // buffer, one barrier per K stage.  Correct (tests green) but SLOWER on the MI355X: E2 fwd 227 -> 237 us, E2 dgrad
                           // 223 -> 261, E4 dgrad 206 -> 227, E3 fwd 206 -> 216 at B = 2048 (0: registers -> ds_write_b128, two barriers per stage)

__device__ __forceinline__ bf16x8 to_bf16x8(f32x4 lo, f32x4 hi) {
    bf16x8 r;
    r[0] = (__bf16)lo.x; r[1] = (__bf16)lo.y; r[2] = (__bf16)lo.z; r[3] = (__bf16)lo.w;
    r[4] = (__bf16)hi.x; r[5] = (__bf16)hi.y; r[6] = (__bf16)hi.z; r[7] = (__bf16)hi.w;
    return r;
}

// channels per K chunk: 16 for the 3-way-split emulation (3x the LDS per channel), 64 for single-tile bf16 workgroups,
// 32 when a workgroup holds two tiles
template <int KCH, int NS, int MT> struct Bf16Chunk {
    static constexpr int CAP = NS == 3 ? 16 : (MT > 1 ? 32 : 64);
    static constexpr int KCB = KCH < CAP ? KCH : CAP;
};

// MT = 128-pixel tiles per workgroup (bf16 mode: 2): one weight slab staged into LDS — and one weight fragment read
// from LDS — serves MT times as many MFMAs; the K chunk shrinks to 32 channels so that the LDS footprint (and with
// it the number of resident workgroups) stays where it was.
#ifdef EPI_TIMING
extern "C" int cvae_epi_dbg_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(epi_dbg), sizeof(epi_dbg)); }
#endif
#ifdef CONV_TIMING     // experiment builds only: where a wave of ONE instantiation (-DCONV_TIMING_KCH/NCH/H) spends its stages
__device__ long long conv_dbg[16 * 4 * 12];
extern "C" int cvae_conv_dbg_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(conv_dbg), sizeof(conv_dbg)); }
#define CT_ON (KCH == CONV_TIMING_KCH && NCH == CONV_TIMING_NCH && H == CONV_TIMING_H && NS == 1 && MODE == MODE_STD)
#define CT_STAMP(v) do { if (CT_ON) v = clock64(); } while (0)
#else
#define CT_ON false
#define CT_STAMP(v)
#endif

#ifndef BF16_WAVES_ATTR
#define BF16_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(MT > 2 ? 1 : 2, MT > 2 ? 1 : 2)))      // MT = 4 (big-tile experiment): one wave per SIMD, 512 registers
#endif
#ifndef BF16_WAVES_PER_SIMD
#define BF16_WAVES_PER_SIMD 1      // experiment: 3 = cap the kernel at 168 VGPRs so that three workgroups fit a CU where the LDS allows it
#endif
template <int KCH, int NCH, int H, int NT, int EPI, int KSPLIT, int KS = 5, int MODE = MODE_STD, int NS = 1, int DMAX = 4, int MT = 1>
__global__ __launch_bounds__(256, BF16_WAVES_PER_SIMD) BF16_WAVES_ATTR void conv5x5_bf16_kernel(ConvBf16Args a) {
    using T = Tile<H>;
    static_assert(NS == 1 || NS == 3, "operand splits: 1 (bf16) or 3 (fp32 emulation, 9 MFMAs per product block)");
    static_assert(MT == 1 || NS == 1, "multi-tile workgroups are a bf16-mode feature");
    constexpr int OFF = 2 - KS / 2;                   // a 3x3 window sits one pixel inside the 5x5 halo
    constexpr int COUT_UP = (MODE == MODE_UP_FWD) ? NCH / 4 : KCH / 4;   // conv channels of the upsampled layer
    constexpr int KCB = Bf16Chunk<KCH, NS, MT>::KCB;                    // channels per K chunk
    constexpr int KB = KCB / 16, OCT = KCB / 8, NB = NT / 32;
    constexpr int PSP = Bf16Geom<H, OCT>::PSP;
    constexpr int A_UNITS = OCT * PSP, W_UNITS = KS * KB * 2 * NT;
    // WDMA (bf16 mode): the weight slab of stage st+1 is copied HBM -> LDS by LDS-DMA while stage st computes — no staging
    // registers, no ds_write_b128 (13 issue cycles each), and with two slab buffers ONE barrier per stage: the slab's unit
    // order in LDS is the staging thread order (unit q <- thread q), which is exactly the DMA's "wave base + lane * 16".
    constexpr bool WDMA = NS == 1 && BF16_WDMA != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* lds_a = reinterpret_cast<bf16x8*>(smem_raw);      // [tile][split][octet][halo pixel]
    bf16x8* lds_w = lds_a + MT * NS * A_UNITS;                 // [buffer (WDMA: 2)][split][tap][kb][half][n]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int mt0 = xcd_tile(blockIdx.x, gridDim.x) * MT, n0 = blockIdx.y * NT;
#ifdef CONV_STAGGER        // timing experiment: the workgroups of the second residency slot start CONV_STAGGER x 4 k cycles late
    if (blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
        for (int i = 0; i < CONV_STAGGER; ++i) __builtin_amdgcn_s_sleep(64);
    }
#endif
    int img0v[MT], ty0v[MT], tx0v[MT];
#pragma unroll
    for (int tl = 0; tl < MT; ++tl) {          // a tile index past the end maps to images >= B: loads give 0, stores are skipped
        const int tileInImg = (mt0 + tl) % T::TILES_PER_IMG;
        img0v[tl] = ((mt0 + tl) / T::TILES_PER_IMG) * T::IMGS;
        ty0v[tl] = (tileInImg / T::TILES_X) * T::TH; tx0v[tl] = (tileInImg % T::TILES_X) * T::TW;
    }
    const int m = wave * 32 + lane_pix<H, NS == 1>(li);      // conflict-free lane -> pixel map (conv_epilogue.h)
    const int pimg = m / (T::TH * T::TW), prem = m % (T::TH * T::TW);
    const int aPix = pimg * T::HPI + (prem / T::TW) * T::HTW + (prem % T::TW);

    f32x16 acc[MT][NB];
#pragma unroll
    for (int tl = 0; tl < MT; ++tl)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[tl][nb][v] = 0.f;

    // weight slab of stage (cc, r): units [s][kb][half][n] <- wp[(r*5+s)][cc*KB + kb][half][n0 + n]
    constexpr int WPT = (W_UNITS + 255) / 256;
    static_assert(!WDMA || W_UNITS % 64 == 0, "whole wave-instructions of 64 units");
    bf16x8 wreg[WDMA ? 1 : NS * WPT];
    // per-thread part of every staging address is fixed for the whole launch: computed once, so a stage adds one
    // wave-uniform term per load instead of redoing the div/mod chains (they were ~3 VALU instructions per MFMA)
    int wbase[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int q = tid + i * 256;
        const int n = q % NT, row = q / NT, half = row & 1, kb = (row >> 1) % KB, s = row / (2 * KB);
        wbase[i] = ((s * (KCH / 16) + kb) * 2 + half) * NCH + n0 + n;
    }
    auto load_w = [&](int st) {
        const int cc = st / KS, r = st % KS;
        const bf16x8* wst = a.wp + (size_t)(r * KS * (KCH / 16) + cc * KB) * 2 * NCH;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int q = tid + i * 256;
            if (W_UNITS % 256 == 0 || q < W_UNITS) {
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) wreg[sp * WPT + i] = wst[wbase[i] + sp * a.splitStride];
            }
        }
    };
    auto dma_w = [&](int st, int buf) {          // WDMA: stage st's slab -> lds_w buffer `buf`
        const int cc = st / KS, r = st % KS;
        const bf16x8* wst = a.wp + (size_t)(r * KS * (KCH / 16) + cc * KB) * 2 * NCH;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            if (W_UNITS % 256 == 0 || i * 256 + wave * 64 < W_UNITS)          // wave-uniform
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wst + wbase[i]),
                                                 (__attribute__((address_space(3))) void*)(lds_w + buf * W_UNITS + i * 256 + wave * 64), 16, 0, 0);
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int q = tid + i * 256;
            if (W_UNITS % 256 == 0 || q < W_UNITS) {
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) lds_w[sp * W_UNITS + q] = wreg[sp * WPT + i];
            }
        }
    };
    // input halo chunk: fp32 NHWC -> registers (one chunk ahead, in flight during the MFMAs of the
    // current chunk) -> bf16 units [octet][halo pixel] (split into hi/mid/lo planes when NS == 3)
    constexpr int NQ = T::HP * OCT, IPT = (NQ + 255) / 256;
    f32x4 ireg[NS == 1 ? 1 : 2 * IPT];
    bf16x8 breg[NS == 1 ? MT * IPT : 1];
    unsigned okmask = 0u;                       // NS == 1: which staged units of breg are real (the others are zero padding)
    static_assert(NS != 1 || MT * IPT <= 32, "one validity bit per staged unit");
    // element index of each staged unit's first channel at chunk 0 (-1: zero padding / past the batch / no unit)
    int ebase[MODE == MODE_UP_DGRAD ? 1 : MT * IPT];
    if constexpr (MODE != MODE_UP_DGRAD) {
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
    }
    auto load_input = [&](int cc) {
#pragma unroll
      for (int tl = 0; tl < MT; ++tl)
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            bool ok;
            size_t e = 0;                           // element index of the unit's first channel
            if constexpr (MODE == MODE_UP_DGRAD) {
                const int q = tid + i * 256;
                const int oct = q % OCT, hp = q / OCT;
                const int img = hp / T::HPI, rem = hp - img * T::HPI;
                const int gy = ty0v[tl] + rem / T::HTW - 2, gx = tx0v[tl] + rem % T::HTW - 2, ib = img0v[tl] + img;
                ok = (NQ % 256 == 0 || q < NQ) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)H && ib < a.B;
                if (ok) {                           // channel k = p*COUT + co of the low-res view = dout[2y+py][2x+px][co]
                    const int k0 = cc * KCB + oct * 8, p = k0 / COUT_UP, co = k0 % COUT_UP;
                    e = ((size_t)(ib * 2 * H + 2 * gy + (p >> 1)) * (2 * H) + 2 * gx + (p & 1)) * COUT_UP + co;
                }
            } else {
                ok = ebase[tl * IPT + i] >= 0;
                e = ok ? (size_t)(ebase[tl * IPT + i] + cc * KCB) : 0;
            }
            if constexpr (NS == 1) {
                // e = 0 when !ok: the load itself is unconditional (no branch per unit).  The zero-select happens in store_input,
                // a chunk later: selecting here makes the compiler wait for the loads right behind their issue (s_waitcnt
                // vmcnt(0) in front of the MFMA loop) — the whole HBM latency of the next chunk's tiles was exposed at every
                // chunk boundary instead of travelling under five stages of MFMAs
                breg[tl * IPT + i] = Act<__bf16>::ld8(a.in, e);
                okmask = ok ? (okmask | (1u << (tl * IPT + i))) : (okmask & ~(1u << (tl * IPT + i)));
            } else {
                f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
                if (ok) { lo = *reinterpret_cast<const f32x4*>(a.in + e); hi = *reinterpret_cast<const f32x4*>(a.in + e + 4); }
                ireg[2 * i] = lo; ireg[2 * i + 1] = hi;
            }
        }
    };
    auto store_input = [&]() {
#pragma unroll
      for (int tl = 0; tl < MT; ++tl)
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256;
            if (!(NQ % 256 == 0 || q < NQ)) continue;
            const int oct = q % OCT, hp = q / OCT;
            if constexpr (NS == 1) {
                bf16x8 z;
#pragma unroll
                for (int k = 0; k < 8; ++k) z[k] = (__bf16)0.f;
                lds_a[tl * A_UNITS + oct * PSP + hp] = ((okmask >> (tl * IPT + i)) & 1u) ? breg[tl * IPT + i] : z;
            } else {
                const f32x4 lo = ireg[2 * i], hi = ireg[2 * i + 1];
                bf16x8 u0, u1, u2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const Split3 sl = split3(lo[e]), sh = split3(hi[e]);
                    u0[e] = sl.hi; u1[e] = sl.mid; u2[e] = sl.lo;
                    u0[4 + e] = sh.hi; u1[4 + e] = sh.mid; u2[4 + e] = sh.lo;
                }
                lds_a[oct * PSP + hp] = u0; lds_a[A_UNITS + oct * PSP + hp] = u1; lds_a[2 * A_UNITS + oct * PSP + hp] = u2;
            }
        }
    };

    static_assert(KCH % KCB == 0 && (KCH / KCB) % KSPLIT == 0, "channel chunking");
    constexpr int NST = (KCH / KCB) / KSPLIT * KS;
    const int st0 = blockIdx.z * NST, st1 = st0 + NST;
    // Every value the epilogue needs from memory is requested HERE, ahead of all other loads, and consumed before the first global
    // store (conv_epilogue.h, load_bias): the bias of this lane's output columns, and for the up-conv input gradient the producer's
    // forward activations (ReLU mask) of exactly the units this lane will store.
    constexpr bool HAS_BIAS = (MODE == MODE_STD && KSPLIT == 1 && (EPI == EPI_BIAS_BNSTAT || EPI == EPI_BIAS_RELU)) || MODE == MODE_UP_FWD;
    constexpr bool BNSTAT = MODE == MODE_STD && KSPLIT == 1 && EPI == EPI_BIAS_BNSTAT;
    // bf16 mode (NS == 1): channel-major accumulators.  Behind the staging buffers: [NT bias][MT][2][4 waves][NT] BatchNorm rows
    float* lds_x = reinterpret_cast<float*>(smem_raw) + (size_t)(MT * NS * A_UNITS + (WDMA ? 2 : 1) * NS * W_UNITS) * 4;
    [[maybe_unused]] float biasv[NB] = {};
    [[maybe_unused]] float bias_stash = 0.f;    // requested first, written to LDS with the first weight slab (no wait of its own)
    if constexpr (HAS_BIAS && NS == 1) {
        if (tid < NT) bias_stash = a.bias[MODE == MODE_UP_FWD ? (n0 + tid) % COUT_UP : n0 + tid];
    } else if constexpr (HAS_BIAS) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = n0 + nb * 32;
            biasv[nb] = a.bias[(MODE == MODE_UP_FWD ? n % COUT_UP : n) + li];
        }
    }
    // up-conv input gradient, bf16 tensors: the ReLU-mask units this lane will store (pixel = its MFMA column, unit k = channels
    // 16k + 8lh .. +7 of block nb), all tiles, requested ahead of every other load
    [[maybe_unused]] bf16x8 mkv[(MODE == MODE_UP_DGRAD && NS == 1) ? MT * NB * 2 : 1];
    if constexpr (MODE == MODE_UP_DGRAD && NS == 1) {
#pragma unroll
        for (int tl = 0; tl < MT; ++tl) {
            const int im = m / (T::TH * T::TW), rem = m % (T::TH * T::TW);
            const int gy = ty0v[tl] + rem / T::TW, gx = tx0v[tl] + rem % T::TW, ib = img0v[tl] + im;
            const size_t base = ((size_t)(ib * H + gy) * H + gx) * NCH + n0 + 8 * lh;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int k = 0; k < 2; ++k) mkv[(tl * NB + nb) * 2 + k] = Act<__bf16>::ld8(a.aux, ib < a.B ? base + nb * 32 + 16 * k : 0);
        }
    }
    if constexpr (WDMA) dma_w(st0, 0); else load_w(st0);
    load_input(st0 / KS);
    [[maybe_unused]] long long ct0 = 0, ct1 = 0, ct2 = 0, ct3 = 0, ct4 = 0, cd[6] = {0, 0, 0, 0, 0, 0}, ctb = 0, cta = 0, ctw = 0;      // CONV_TIMING builds
    CT_STAMP(ctb);
    [[maybe_unused]] const long long crt0 = CT_ON ? (long long)wall_clock64() : 0;
    for (int st = st0; st < st1; ++st) {
        const int r = st % KS;
        int wbuf = 0;
        CT_STAMP(ct0);
        if constexpr (WDMA) {
            wbuf = (st - st0) & 1;
            if (r == 0) {
                __syncthreads();                   // chunk boundary: everyone finished reading the previous chunk's input tiles
                store_input();
            }
            CT_STAMP(ct1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part of slab st has landed (and the next chunk's inputs)
            CT_STAMP(cta); CT_STAMP(ctw);
            __syncthreads();                       // slab st (and the input tiles) visible to all; everyone is done with slab st-1
            CT_STAMP(ct2);
            if (r == 0 && st + KS < st1) load_input(st / KS + 1);
            if (st + 1 < st1) dma_w(st + 1, wbuf ^ 1);            // lands in the buffer stage st-1 read, while this stage computes
            CT_STAMP(ct3);
        } else {
        __syncthreads();                       // everyone finished reading the previous stage
        CT_STAMP(ct1);
        if (r == 0) store_input();
        CT_STAMP(cta);
        store_w();
        if constexpr (HAS_BIAS && NS == 1) { if (st == st0 && tid < NT) lds_x[tid] = bias_stash; }
        CT_STAMP(ctw);
        // issue order: vmcnt retires in order, so the (older) halo loads must not sit between a weight
        // load and the store_w that waits for it (see conv_mfma.hip)
        if (r == 0 && st + KS < st1) load_input(st / KS + 1);
        if (st + 1 < st1) load_w(st + 1);      // in flight while this stage computes
        CT_STAMP(ct2);
        __syncthreads();
        CT_STAMP(ct3);
        }
        const bf16x8* ap = lds_a + lh * PSP + aPix + (r + OFF) * T::HTW + OFF;
        const bf16x8* bp = lds_w + wbuf * W_UNITS + lh * NT + li;
#ifndef BF16_LOOP
#define BF16_LOOP 0       // timing experiments (results WRONG unless 0): bit 0 = compiler-scheduled loop (iglp_opt) instead of the pinned pipeline, bit 1 = operands not swapped
#endif
        if constexpr (NS == 1 && (BF16_LOOP & 1)) {
            __builtin_amdgcn_iglp_opt(0);
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    bf16x8 bv[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) bv[nb] = bp[((s * KB + kb) * 2) * NT + nb * 32];
#pragma unroll
                    for (int tl = 0; tl < MT; ++tl) {
                        const bf16x8 av = ap[tl * A_UNITS + (kb * 2) * PSP + s];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[tl][nb] = (BF16_LOOP & 2) ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[nb], acc[tl][nb], 0, 0, 0)
                                                          : __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[nb], av, acc[tl][nb], 0, 0, 0);
                    }
                }
        } else if constexpr (NS == 1) {
            // Software pipeline, written out: the fragments of step i + PF are requested before the MFMAs of step i (a step = one tap x one
            // 16-channel block: NB weight + MT pixel fragments, MT x NB MFMAs), three register sets in rotation, and the order is PINNED with
            // sched_group_barrier — left to itself (iglp_opt) the scheduler of hipcc 7.2 serialised `ds_read -> s_waitcnt lgkmcnt(0) -> MFMA`
            // on one reused register quad as soon as the operand roles were swapped (E2 forward: MFMA phase 7.7 k -> 10.1 k cycles).
            constexpr int NSTEP = KS * KB, PF = 2, NSET = 3;
            bf16x8 wf[NSET][NB], xf[NSET][MT];
            auto ld = [&](int i, int b) {
                const int s = i / KB, kb = i % KB;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wf[b][nb] = bp[((s * KB + kb) * 2) * NT + nb * 32];
#pragma unroll
                for (int tl = 0; tl < MT; ++tl) xf[b][tl] = ap[tl * A_UNITS + (kb * 2) * PSP + s];
            };
#pragma unroll
            for (int i = 0; i < PF && i < NSTEP; ++i) ld(i, i % NSET);
            __builtin_amdgcn_sched_group_barrier(0x100, (PF < NSTEP ? PF : NSTEP) * (NB + MT), 0);
#pragma unroll
            for (int i = 0; i < NSTEP; ++i) {
                if (i + PF < NSTEP) ld(i + PF, (i + PF) % NSET);
#pragma unroll
                for (int tl = 0; tl < MT; ++tl)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)      // weights as the A operand: D[channel][pixel], see conv_epilogue.h (channel-major)
                        acc[tl][nb] = (BF16_LOOP & 2) ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[i % NSET][tl], wf[i % NSET][nb], acc[tl][nb], 0, 0, 0)
                                                      : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i % NSET][nb], xf[i % NSET][tl], acc[tl][nb], 0, 0, 0);
                // one fragment read behind each MFMA while there are reads left in this step, then the remaining MFMAs back to back
                constexpr int NM = MT * NB, NR = NB + MT;
#pragma unroll
                for (int k = 0; k < NM; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i + PF < NSTEP && k < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if constexpr (NR > NM) { if (i + PF < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, NR - NM, 0); }
            }
        }
#pragma unroll
        for (int s = 0; s < (NS == 1 ? 0 : KS); ++s)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                if (NS == 1) {
                } else {
                    bf16x8 av[3];
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) av[sp] = ap[sp * A_UNITS + (kb * 2) * PSP + s];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        bf16x8 bv[3];
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) bv[sp] = bp[sp * W_UNITS + ((s * KB + kb) * 2) * NT + nb * 32];
                        // exact partial products, smallest magnitudes first (0 = hi, 1 = mid, 2 = lo); DMAX = 4: all
                        // nine; DMAX = 2: the six of relative weight >= 2^-16 (drops mid*lo, lo*mid, lo*lo <= 3*2^-24)
#pragma unroll
                        for (int d = DMAX; d >= 0; --d)
#pragma unroll
                            for (int ia = 0; ia < 3; ++ia) {
                                const int ib = d - ia;
                                if (ib >= 0 && ib < 3) acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ia], bv[ib], acc[0][nb], 0, 0, 0);
                            }
                    }
                }
            }
#ifdef CONV_TIMING
        if (CT_ON) {
            __builtin_amdgcn_sched_barrier(0);
            long long te = clock64();
            cd[0] += ct1 - ct0; cd[1] += ct2 - ct1; cd[2] += ct3 - ct2; cd[3] += te - ct3; cd[4] += cta - ct1; cd[5] += ctw - cta;
        }
#endif
    }
    vm_drained();
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int numTiles = cdiv(a.B, T::IMGS) * T::TILES_PER_IMG;
#ifdef CONV_TIMING
    if (CT_ON) { CT_STAMP(ct4); cd[0] += 0; }
#endif
    if constexpr (NS == 1) {
        // ---- channel-major epilogue (conv_epilogue.h): every wave on its own, straight from the accumulators ----
        [[maybe_unused]] f32x4 bq[NB][4];                 // bias of the lane's 16 channels per block: quads 8g + 4lh .. +3
        if constexpr (HAS_BIAS) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[nb][g] = *reinterpret_cast<const f32x4*>(lds_x + nb * 32 + 8 * g + 4 * lh);
        }
        [[maybe_unused]] float* red = lds_x + NT;
        const int e16 = li >> 1, chE = (e16 & 3) + 8 * (e16 >> 2) + 4 * lh;       // the element half_wave_colsum16 leaves in this lane
#pragma unroll
        for (int tl = 0; tl < MT; ++tl) {
            const int im = m / (T::TH * T::TW), rem = m % (T::TH * T::TW);
            const int gy = ty0v[tl] + rem / T::TW, gx = tx0v[tl] + rem % T::TW, ib = img0v[tl] + im;
            const bool valid = ib < a.B;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x16& c = acc[tl][nb];
                if constexpr (HAS_BIAS) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const float x = c[v] + bq[nb][v >> 2][v & 3];
                        c[v] = (MODE == MODE_UP_FWD || EPI == EPI_BIAS_RELU) ? fmaxf(x, 0.f) : x;
                    }
                }
                if constexpr (MODE == MODE_STD && KSPLIT > 1) {                   // fp32 partial sums to slab z: 4 channels = 16 bytes per quad
                    float* out = a.out + (size_t)blockIdx.z * a.sliceFloats + ((size_t)(ib * H + gy) * H + gx) * NCH + n0 + nb * 32 + 4 * lh;
                    if (valid) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(out + 8 * g) = f32x4{c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
                    }
                } else {
                    bf16x8 u[2];
                    cm_pack_units(c, u);
                    size_t base;
                    if constexpr (MODE == MODE_UP_FWD) {                          // column block nb = phase p, channels c0..c0+31 of the upsampled layer
                        const int n = n0 + nb * 32, p = n / COUT_UP, c0 = n % COUT_UP;
                        base = ((size_t)(ib * 2 * H + 2 * gy + (p >> 1)) * (2 * H) + 2 * gx + (p & 1)) * COUT_UP + c0 + 8 * lh;
                    } else {
                        base = ((size_t)(ib * H + gy) * H + gx) * NCH + n0 + nb * 32 + 8 * lh;
                    }
                    if constexpr (MODE == MODE_UP_DGRAD) {
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const bf16x8 mk = mkv[(tl * NB + nb) * 2 + k];
#pragma unroll
                            for (int e = 0; e < 8; ++e) u[k][e] = (float)mk[e] > 0.f ? u[k][e] : (__bf16)0.f;
                        }
                    }
                    if (valid && !(EPI_EXPERIMENT & 1)) { Act<__bf16>::st8(a.out, base, u[0]); Act<__bf16>::st8(a.out, base + 16, u[1]); }
                    if ((EPI_EXPERIMENT & 1) && u[0][0] == (__bf16)123.f && u[1][7] == (__bf16)77.f) Act<__bf16>::st8(a.out, base, u[0]);     // timing builds: keeps the values alive
                }
                if constexpr (BNSTAT && !(EPI_EXPERIMENT & 2)) {          // per-wave column sums of y and y*y over its 32 pixels (pixels past the batch count as 0)
                    float sv[16], qv[16];
#pragma unroll
                    for (int v = 0; v < 16; ++v) { sv[v] = valid ? c[v] : 0.f; qv[v] = sv[v] * sv[v]; }
                    const float S = half_wave_colsum16(sv), Q = half_wave_colsum16(qv);
                    if ((lane & 1) == 0) {
                        red[((tl * 2 + 0) * 4 + wave) * NT + nb * 32 + chE] = S;
                        red[((tl * 2 + 1) * 4 + wave) * NT + nb * 32 + chE] = Q;
                    }
                }
            }
        }
        if constexpr (BNSTAT && !(EPI_EXPERIMENT & 2)) {
            // the four waves' rows meet once: per tile and channel (sum, M2 about the tile mean) as bn_fwd_finalize merges them
            // (nn.BatchNorm2d train-mode statistics, vae_nets.py:70,75,80,85); M2 = Q - S*S/n in double
            __syncthreads();
            for (int idx = tid; idx < MT * NT; idx += 256) {
                const int tl = idx / NT, cc = idx % NT, mt = mt0 + tl;
                if (mt >= numTiles) continue;
                float S = 0.f, Q = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { S += red[((tl * 2 + 0) * 4 + w) * NT + cc]; Q += red[((tl * 2 + 1) * 4 + w) * NT + cc]; }
                int nvalid_img = a.B - img0v[tl];
                if (nvalid_img > T::IMGS) nvalid_img = T::IMGS;
                const double cnt = (double)(nvalid_img * T::TH * T::TW);
                const double m2 = (double)Q - (double)S * (double)S / cnt;
                a.bnpart[(size_t)mt * NCH + n0 + cc] = S;
                a.bnpart[((size_t)numTiles + mt) * NCH + n0 + cc] = (float)(m2 > 0.0 ? m2 : 0.0);
            }
        }
    } else {
    if constexpr (HAS_BIAS) {                    // bias (+ ReLU) of EVERY tile in registers before the first store of any of them
#pragma unroll
        for (int tl = 0; tl < MT; ++tl) apply_bias<NT, (MODE == MODE_UP_FWD ? EPI_BIAS_RELU : EPI)>(acc[tl], biasv);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tl = 0; tl < MT; ++tl) {
    const int mt = mt0 + tl, img0 = img0v[tl], ty0 = ty0v[tl], tx0 = tx0v[tl];
    if (MODE != MODE_STD) {
        // per-wave transpose through LDS, then 16-byte stores (as epilogue_store)
        __syncthreads();
        float* patch = smem + wave * (32 * 36);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = n0 + nb * 32;                               // 32 columns of one phase (COUT >= 32)
            const int p = MODE == MODE_UP_FWD ? n / COUT_UP : 0, c0 = MODE == MODE_UP_FWD ? n % COUT_UP : n;
#pragma unroll
            for (int v = 0; v < 16; ++v) patch[((v & 3) + 8 * (v >> 2) + 4 * lh) * 36 + li] = acc[tl][nb][v];     // bias + ReLU applied above
            if constexpr (NS == 1) {          // bf16 tensors: 8 channels = one 16-byte unit per lane
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int idx = it * 64 + lane, px = idx >> 2, c8 = idx & 3;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + px * 36 + c8 * 8);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + px * 36 + c8 * 8 + 4);
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (__bf16)lo[e]; o[4 + e] = (__bf16)hi[e]; }
                    const int mm = wave * 32 + lane_pix<H, NS == 1>(px);
                    const int im = mm / (T::TH * T::TW), rem = mm % (T::TH * T::TW);
                    const int gy = ty0 + rem / T::TW, gx = tx0 + rem % T::TW, ib = img0 + im;
                    if (ib >= a.B) continue;
                    if (MODE == MODE_UP_FWD) {
                        const size_t oo = ((size_t)(ib * 2 * H + 2 * gy + (p >> 1)) * (2 * H) + 2 * gx + (p & 1)) * COUT_UP + c0 + c8 * 8;
                        Act<__bf16>::st8(a.out, oo, o);
                    } else {
                        const size_t oo = ((size_t)(ib * H + gy) * H + gx) * NCH + c0 + c8 * 8;
                        const bf16x8 mk = mkv[(tl * NB + nb) * 2 + it];            // fetched before the first store
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (float)mk[e] > 0.f ? o[e] : (__bf16)0.f;
                        Act<__bf16>::st8(a.out, oo, o);
                    }
                }
                continue;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = it * 64 + lane, px = idx >> 3, c4 = idx & 7;
                f32x4 val = *reinterpret_cast<const f32x4*>(patch + px * 36 + c4 * 4);
                const int mm = wave * 32 + lane_pix<H, NS == 1>(px);
                const int im = mm / (T::TH * T::TW), rem = mm % (T::TH * T::TW);
                const int gy = ty0 + rem / T::TW, gx = tx0 + rem % T::TW, ib = img0 + im;
                if (ib >= a.B) continue;
                if (MODE == MODE_UP_FWD) {
                    const size_t o = ((size_t)(ib * 2 * H + 2 * gy + (p >> 1)) * (2 * H) + 2 * gx + (p & 1)) * COUT_UP + c0 + c4 * 4;
                    *reinterpret_cast<f32x4*>(a.out + o) = val;
                } else {
                    const size_t o = ((size_t)(ib * H + gy) * H + gx) * NCH + c0 + c4 * 4;
                    const f32x4 mk = *reinterpret_cast<const f32x4*>(a.aux + o);
                    val.x = mk.x > 0.f ? val.x : 0.f; val.y = mk.y > 0.f ? val.y : 0.f;
                    val.z = mk.z > 0.f ? val.z : 0.f; val.w = mk.w > 0.f ? val.w : 0.f;
                    *reinterpret_cast<f32x4*>(a.out + o) = val;
                }
            }
        }
    } else if (KSPLIT > 1) {
        epilogue_store<H, NT, NCH, EPI_PLAIN, float, NS == 1>(acc[tl], a.out + (size_t)blockIdx.z * a.sliceFloats, nullptr, smem, a.B,
                                                              mt, n0, img0, ty0, tx0, numTiles);
    } else {
        if constexpr (NS == 1) epilogue_store<H, NT, NCH, EPI, __bf16, true>(acc[tl], a.out, a.bnpart, smem, a.B, mt, n0, img0, ty0, tx0, numTiles, CT_ON ? tl * 12 : 32);
        else epilogue_store<H, NT, NCH, EPI>(acc[tl], a.out, a.bnpart, smem, a.B, mt, n0, img0, ty0, tx0, numTiles);
    }
    }
    }
#ifdef CONV_TIMING
    if (CT_ON && (blockIdx.x & 63) == 0 && blockIdx.x < 1024 && blockIdx.y == 0 && lane == 0) {
        long long tend = clock64();
        long long* o = conv_dbg + ((blockIdx.x >> 6) * 4 + wave) * 12;
        o[0] = cd[0]; o[1] = cd[1]; o[2] = cd[2]; o[3] = cd[3]; o[4] = ct4 - ctb; o[5] = tend - ct4; o[6] = st1 - st0; o[7] = tend - ctb; o[8] = cd[4]; o[9] = cd[5];
        o[10] = (long long)wall_clock64() - crt0;        // 100 MHz ticks over the same span: in-kernel clock = o[7] / o[10] * 100 MHz
    }
#endif
}

// ---- weight packing into bf16 units: unit ((tap*(K/16) + kb)*2 + half)*N + n holds k = kb*16 + half*8 .. +7.
//   PACK_FWD    : fp32 W[25][CIN][COUT]            -> k = ci, n = co
//   PACK_DGRAD  : same source, taps flipped        -> k = co, n = ci
//   PACK_UPFWD  : collapsed wc[4][9][CIN][COUT]    -> k = ci, n = p*COUT + co          (9 taps)
//   PACK_UPDGRAD: same source, taps flipped        -> k = p*COUT + co, n = ci          (9 taps)
enum { PACK_FWD = 0, PACK_DGRAD = 1, PACK_UPFWD = 2, PACK_UPDGRAD = 3 };
struct PackJob { const float* w; bf16x8* dst; int cin, cout, mode, splits; };
struct PackJobs { PackJob j[8]; };

__global__ __launch_bounds__(256) void pack_w_bf16_kernel(PackJobs jobs) {
    const PackJob jb = jobs.j[blockIdx.y];
    if (jb.w == nullptr) return;
    const int cin = jb.cin, cout = jb.cout, mode = jb.mode;
    const int taps = mode >= PACK_UPFWD ? 9 : 25;
    const int K = mode == PACK_FWD || mode == PACK_UPFWD ? cin : (mode == PACK_DGRAD ? cout : 4 * cout);
    const int N = mode == PACK_FWD ? cout : (mode == PACK_UPFWD ? 4 * cout : cin);
    const int units = taps * (K / 8) * N;
    // every channel count of the model is a power of two: shifts and masks instead of three runtime divisions per unit
    // (the kernel ran 16 us for 2 MB: address arithmetic, not bytes)
    const int nsh = 31 - __builtin_clz(N), ksh = 31 - __builtin_clz(K / 16), csh = 31 - __builtin_clz(cout);
    for (int u = blockIdx.x * 256 + threadIdx.x; u < units; u += gridDim.x * 256) {
        const int n = u & (N - 1), row = u >> nsh, half = row & 1, kb = (row >> 1) & ((K / 16) - 1), tap = row >> (ksh + 1);
        const int k0 = kb * 16 + half * 8;
        bf16x8 r0, r1, r2;
        // the dgrad orientations contract over the source's FASTEST index (co): the unit's 8 values are 32 contiguous bytes
        // -> two 16-byte loads (they were eight 4-byte loads a whole row apart between neighbouring lanes)
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
        if (mode == PACK_DGRAD || mode == PACK_UPDGRAD) {
            const float* src = mode == PACK_DGRAD ? jb.w + ((size_t)(24 - tap) * cin + n) * cout + k0
                                                  : jb.w + ((size_t)((k0 >> csh) * 9 + 8 - tap) * cin + n) * cout + (k0 & (cout - 1));
            c0 = *reinterpret_cast<const f32x4*>(src); c1 = *reinterpret_cast<const f32x4*>(src + 4);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            float v;
            if (mode == PACK_FWD) v = jb.w[((size_t)tap * cin + k) * cout + n];
            else if (mode == PACK_UPFWD) v = jb.w[((size_t)((n >> csh) * 9 + tap) * cin + k) * cout + (n & (cout - 1))];
            else v = e < 4 ? c0[e & 3] : c1[e & 3];
            if (jb.splits == 1) r0[e] = (__bf16)v;
            else { const Split3 sv = split3(v); r0[e] = sv.hi; r1[e] = sv.mid; r2[e] = sv.lo; }
        }
        jb.dst[u] = r0;
        if (jb.splits == 3) { jb.dst[units + u] = r1; jb.dst[2 * (size_t)units + u] = r2; }
    }
}

// packed copies live in one workspace block: [fwd L1 | dgrad L1 | ... | dgrad L4 | upfwd L5 | updgrad L5 | ... L7],
// each entry = ns (1 or 3) consecutive split copies of pack_units(layer) units
static int64_t pack_units(int layer) { return (int64_t)(layer >= 5 ? 36 : 25) * kLayers[layer].cin * kLayers[layer].cout / 8; }
int64_t conv_bf16_pack_floats(int ns) {
    int64_t u = 0;
    for (int l = 1; l <= 7; ++l) u += 2 * ns * pack_units(l);
    return u * 4;                               // 16-byte units -> floats
}
static bf16x8* pack_ptr(float* packed, int layer, int dgrad, int ns) {
    int64_t u = 0;
    for (int l = 1; l < layer; ++l) u += 2 * ns * pack_units(l);
    if (dgrad) u += ns * pack_units(layer);
    return reinterpret_cast<bf16x8*>(packed) + u;
}

int launch_pack_w_bf16(const float* const w[4], float* packed, int ns, hipStream_t st) {
    PackJobs jobs;
    for (int l = 1; l <= 4; ++l)
        for (int d = 0; d < 2; ++d)
            jobs.j[(l - 1) * 2 + d] = PackJob{w[l - 1], pack_ptr(packed, l, d, ns), kLayers[l].cin, kLayers[l].cout, d ? PACK_DGRAD : PACK_FWD, ns};
    hipLaunchKernelGGL(pack_w_bf16_kernel, dim3(400, 8), dim3(256), 0, st, jobs);      // one unit per thread for the largest layer (102 400 units): a chain of dependent loads per unit
    CVAE_CHECK_LAUNCH();
    return 0;
}
// collapsed weights wc[i] of D1..D3 (conv_up.hip: collapse_w_kernel must have run on `st` before)
int launch_pack_up_bf16(const float* const wc[3], float* packed, int ns, hipStream_t st) {
    PackJobs jobs;
    for (int i = 0; i < 8; ++i) jobs.j[i] = PackJob{nullptr, nullptr, 0, 0, 0, 1};
    for (int l = 5; l <= 7; ++l)
        for (int d = 0; d < 2; ++d)
            jobs.j[(l - 5) * 2 + d] = PackJob{wc[l - 5], pack_ptr(packed, l, d, ns), kLayers[l].cin, kLayers[l].cout, d ? PACK_UPDGRAD : PACK_UPFWD, ns};
    hipLaunchKernelGGL(pack_w_bf16_kernel, dim3(144, 6), dim3(256), 0, st, jobs);      // 36 864 units for D1
    CVAE_CHECK_LAUNCH();
    return 0;
}

template <int KCH, int NCH, int H, int NT, int EPI, int KSPLIT = 1, int KS = 5, int MODE = MODE_STD, int NS = 1, int DMAX = 4, int MT = 1>
static int run_bf16_ns(const ConvBf16Args& a, hipStream_t st) {
    using T = Tile<H>;
    constexpr int KCB = Bf16Chunk<KCH, NS, MT>::KCB;
    constexpr int WBUF = (NS == 1 && BF16_WDMA != 0) ? 2 : 1;          // LDS-DMA weight slabs are double-buffered
    constexpr int STAGE = (MT * NS * (KCB / 8) * Bf16Geom<H, KCB / 8>::PSP + WBUF * NS * KS * (KCB / 16) * 2 * NT) * 16;
    constexpr int EPI_BYTES = (8 * NT > 4 * 32 * 36 ? 8 * NT : 4 * 32 * 36) * 4;
    // bf16 mode: no transpose patch; bias row + BatchNorm rows of the channel-major epilogue sit behind the staging buffers
    constexpr int SMEM = NS == 1 ? STAGE + (NT + MT * 8 * NT) * 4 : (STAGE > EPI_BYTES ? STAGE : EPI_BYTES);
    auto kern = conv5x5_bf16_kernel<KCH, NCH, H, NT, EPI, KSPLIT, KS, MODE, NS, DMAX, MT>;
    static DeviceOnce once;
    { int rc = cvae_grant_lds(once, reinterpret_cast<const void*>(kern), SMEM); if (rc) return rc; }
    dim3 grid(cdiv(cdiv(a.B, T::IMGS) * T::TILES_PER_IMG, MT), NCH / NT, KSPLIT);
    cvae_probe_begin(st);
    hipLaunchKernelGGL(kern, grid, dim3(256), SMEM, st, a);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    return 0;
}

// a.splitStride != 0 selects the 3-split fp32-emulation instantiation; plain bf16 runs two tiles per workgroup
template <int KCH, int NCH, int H, int NT, int EPI, int KSPLIT = 1, int KS = 5, int MODE = MODE_STD>
static int run_bf16(const ConvBf16Args& a, hipStream_t st) {
    if (a.splitStride && a.products == 6) return run_bf16_ns<KCH, NCH, H, NT, EPI, KSPLIT, KS, MODE, 3, 2>(a, st);
    if (a.splitStride) return run_bf16_ns<KCH, NCH, H, NT, EPI, KSPLIT, KS, MODE, 3, 4>(a, st);
    return run_bf16_ns<KCH, NCH, H, NT, EPI, KSPLIT, KS, MODE, 1, 4, BF16_MT>(a, st);
}

// ---------------------------------------------------------------------------------------------
// 4x4 images in bf16 mode (D0 forward and input gradient at 64x64 frames): the padding-skipping kernel of
// conv_mfma.hip (conv4x4_row_kernel) on the bf16 MFMA.  M tile = 32 images x two image rows, wave = image column; a
// kernel row runs only for the image rows it reaches, a wave skips the taps that leave its column; LDS holds the 16
// real pixels as [octet][pixel][image] planes of 16-byte units (lanes = consecutive images).  Packed weights, K chunk
// (32 channels), slab layout and k order as in conv5x5_bf16_kernel; always split-K (fp32 partial sums to slab z).
// ---------------------------------------------------------------------------------------------
template <int KCH, int NCH, int NT, int KSPLIT>
__global__ __launch_bounds__(256) void conv4x4_row_bf16_kernel(ConvBf16Args a) {
    constexpr int KCB = 32, KB = KCB / 16, OCT = KCB / 8, NB = NT / 32;
    constexpr int PXS = 33, PSP = 16 * PXS + 2;                 // [pixel][32 images + 1] units, plane stride
    constexpr int A_UNITS = OCT * PSP, W_UNITS = 5 * KB * 2 * NT;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[(A_UNITS + W_UNITS) * 16];
    bf16x8* lds_a = reinterpret_cast<bf16x8*>(smem_raw);
    bf16x8* lds_w = lds_a + A_UNITS;                            // [tap s][kb][half][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int ya = (blockIdx.x & 1) * 2, img0 = (blockIdx.x >> 1) * 32, n0 = blockIdx.y * NT;   // image rows ya, ya + 1
    const int r0 = ya == 0 ? 1 : 0, r1 = ya == 0 ? 4 : 3;              // kernel rows that at least one of the two image rows needs
    // image column of this wave: w in the first image row, w ^ 1 in the second (7 taps per two-row stage on every wave, see conv4x4_row_kernel)

    f32x16 acc[2][NB];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[t][nb][v] = 0.f;

    constexpr int WPT = (W_UNITS + 255) / 256;
    static_assert(W_UNITS % 256 == 0, "whole staging rounds");
    bf16x8 wreg[WPT];
    int wbase[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int q = tid + i * 256;
        const int n = q % NT, row = q / NT, half = row & 1, kb = (row >> 1) % KB, s = row / (2 * KB);
        wbase[i] = ((s * (KCH / 16) + kb) * 2 + half) * NCH + n0 + n;
    }
    auto load_w = [&](int cc, int r) {
        const bf16x8* wst = a.wp + (size_t)(r * 5 * (KCH / 16) + cc * KB) * 2 * NCH;
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[i] = wst[wbase[i]];
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) lds_w[tid + i * 256] = wreg[i];
    };
    constexpr int NQ = 512 * OCT, IPT = NQ / 256;              // 32 images x 16 pixels x OCT octets
    bf16x8 breg[IPT];
    unsigned okmask = 0u;                                       // zero-select deferred to store_input (see conv5x5_bf16_kernel)
    auto load_input = [&](int cc) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256, oct = q % OCT, hp = q / OCT, ib = img0 + (hp >> 4);
            const bool ok = ib < a.B;
            breg[i] = Act<__bf16>::ld8(a.in, ok ? ((size_t)ib * 16 + (hp & 15)) * KCH + cc * KCB + oct * 8 : 0);
            okmask = ok ? (okmask | (1u << i)) : (okmask & ~(1u << i));
        }
    };
    auto store_input = [&]() {
        bf16x8 z;
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = (__bf16)0.f;
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256, oct = q % OCT, hp = q / OCT;
            lds_a[oct * PSP + (hp & 15) * PXS + (hp >> 4)] = (okmask >> i) & 1u ? breg[i] : z;
        }
    };

    static_assert(KCH % KCB == 0 && (KCH / KCB) % KSPLIT == 0, "channel chunking");
    constexpr int CPS = (KCH / KCB) / KSPLIT;
    const int c0 = blockIdx.z * CPS, c1 = c0 + CPS;
    load_w(c0, r0);
    load_input(c0);
    for (int cc = c0; cc < c1; ++cc)
        for (int r = r0; r <= r1; ++r) {
            __syncthreads();                       // everyone finished reading the previous stage
            if (r == r0) store_input();
            store_w();
            if (r == r0 && cc + 1 < c1) load_input(cc + 1);        // older than the weight load below (vmcnt retires in order)
            if (r < r1) load_w(cc, r + 1);
            else if (cc + 1 < c1) load_w(cc + 1, r0);
            __syncthreads();
            const bf16x8* bp = lds_w + lh * NT + li;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int yi = ya + t + r - 2;                 // input row of image row ya + t under kernel row r
                if ((unsigned)yi > 3u) continue;               // workgroup-uniform
                const int xc = wave ^ t;                       // taps whose input column xc+s-2 exists: s0..s1
                const int s0 = xc < 2 ? 2 - xc : 0, s1 = xc > 1 ? 5 - xc : 4;
                const bf16x8* ap = lds_a + lh * PSP + (yi * 4 + xc - 2) * PXS + li;
                for (int s = s0; s <= s1; ++s)
#pragma unroll
                    for (int kb = 0; kb < KB; ++kb) {
                        const bf16x8 av = ap[(kb * 2) * PSP + s * PXS];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[t][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bp[((s * KB + kb) * 2) * NT + nb * 32], acc[t][nb], 0, 0, 0);
                    }
            }
        }
    // KSPLIT > 1: fp32 partial sums to slab z (the callers' finish kernels add them, add the bias, round);  KSPLIT == 1: the plain
    // result straight to the bf16 tensor (input gradient at large batches: no slab, no finish launch)
    vm_drained();
    float* out = a.out + (size_t)blockIdx.z * a.sliceFloats;
    __bf16* out16 = reinterpret_cast<__bf16*>(a.out);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int ib = img0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                const size_t o = ((size_t)ib * 16 + (ya + t) * 4 + (wave ^ t)) * NCH + n0 + nb * 32 + li;
                if (ib < a.B) { if (KSPLIT > 1) out[o] = acc[t][nb][v]; else out16[o] = (__bf16)acc[t][nb][v]; }
            }
}

template <int KCH, int NCH, int NT, int KSPLIT>
static int run4x4_bf16(const ConvBf16Args& a, hipStream_t st) {
    static_assert(NCH % NT == 0, "channel tiling");
    cvae_probe_begin(st);
    hipLaunchKernelGGL((conv4x4_row_bf16_kernel<KCH, NCH, NT, KSPLIT>), dim3(cdiv(a.B, 32) * 2, NCH / NT, KSPLIT), dim3(256), 0, st, a);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    return 0;
}

bool conv_bf16_supported(int layer, int width) { return (width == 64 || width == 128) && layer >= 1 && layer <= 7; }

// bf16 mode: E2..E4 forward / input gradient run on the persistent kernel of conv_bf16_ps.hip (CVAE_CONV_PS=0: the per-tile kernel, for A/B runs)
// CVAE_CONV_PS = bit mask of the layers that run on it: bit (layer - 1) forward, bit (3 + layer - 1) input gradient (0 = none: A/B runs)
#ifndef CONV_PS_DEFAULT
#define CONV_PS_DEFAULT 7        // forward E2..E4 on the persistent kernel; the input gradients measured level or slower on it
#endif
static bool use_ps_kernel(int layer, bool dgrad) {
    static const int mask = [] { const char* e = getenv("CVAE_CONV_PS"); return e ? atoi(e) : CONV_PS_DEFAULT; }();
    const int bit = (dgrad ? 3 : 0) + (layer >= 4 ? 2 : layer - 1);       // layer 4 (128-wide frames only) shares E4's bit
    return ((mask >> bit) & 1) != 0;
}

static int bf16_big_mask(int width) {
    static const int big = [] { const char* e = getenv("CVAE_BF16_BIG"); return e ? atoi(e) : -1; }();
    return big >= 0 ? big : (width == 128 ? BF16_BIG_DEFAULT_W128 : BF16_BIG_DEFAULT);
}
// E2..E4 forward / input gradient in bf16 mode: the persistent kernel families in the order the launchers try them.  *family = 2 (the big-tile
// kernel of conv_bf16_big.hip took the layer), 1 (the two-workgroup kernel of conv_bf16_ps.hip did) or 0 (neither serves this layer at this size — no
// instantiation, masked out, or a tensor of 2 GiB and more: the caller runs the per-tile kernel).  Returns the launch's error code.
static int try_persistent_bf16(int layer, int width, bool dgrad, const ConvBf16Args& a, hipStream_t st, int* family) {
    *family = 0;
    const int big = bf16_big_mask(width);
    const int m = dgrad ? ((big >> 2) & 3) | ((big >> 5) & 4) : (big >> 4) & 7;      // bits 2 / 3 / 7: E4 / E3 / E2 input gradient; bits 4 / 5 / 6: E3 / E4 / E2 forward
    if (m) { const int rc = launch_conv_bf16_big(layer, width, dgrad, m, a, st); if (rc != -100) { *family = 2; return rc; } }
    if (use_ps_kernel(layer, dgrad)) { const int rc = launch_conv_bf16_ps(layer, width, dgrad, a, st); if (rc != -100) { *family = 1; return rc; } }
    return 0;
}
int conv_bf16_route(int layer, int width, bool dgrad, int B) {
    ConvBf16Args a{};
    a.B = B;
    int family = 0;
    g_conv_dry = true;
    (void)try_persistent_bf16(layer, width, dgrad, a, nullptr, &family);
    g_conv_dry = false;
    return family;
}
// `tilesPerPartial` (out, may be null): how many 128-pixel tiles one BatchNorm partial row of `bnpart` covers — 1 for the per-tile and the
// two-workgroup persistent kernels, 4 for the items of conv_bf16_big.hip.  The caller hands it to launch_bn_fwd_finalize: the kernel that
// actually ran decides, not a second reading of the switches.
int launch_conv_fwd_bf16(int layer, int width, int ns, int B, const float* in, const float* packed, const float* bias, float* out,
                         float* bnpart, float* ws, hipStream_t st, int* tilesPerPartial) {
    ConvBf16Args a{in, pack_ptr(const_cast<float*>(packed), layer, 0, ns >= 3 ? 3 : 1), bias, out, bnpart, B, 0, nullptr, ns >= 3 ? pack_units(layer) : 0, ns == 6 ? 6 : 9};
    if (tilesPerPartial) *tilesPerPartial = 1;
    if (ns == 1) {       // the persistent kernels; the big-tile one emits one BatchNorm partial per item
        int family;
        const int rc = try_persistent_bf16(layer, width, false, a, st, &family);
        if (family == 2 && tilesPerPartial) *tilesPerPartial = conv_bf16_big_tiles(layer, width, false);
        if (family) return rc;
    }
    if (width == 64) {
        switch (layer) {
            case 1: return run_bf16<32, 64, 32, 64, EPI_BIAS_BNSTAT>(a, st);
            case 2: return run_bf16<64, 128, 16, 64, EPI_BIAS_BNSTAT>(a, st);
            case 3: return run_bf16<128, 256, 8, 64, EPI_BIAS_BNSTAT>(a, st);
            case 4: {     // 4x4 images: split-K x4 over the channel chunks so that 256 workgroups exist at B=256
                const int64_t slice = (int64_t)B * 16 * 128;
                a.out = ws; a.sliceFloats = slice;
                // bf16 mode: the padding-skipping kernel; large batches need only two K slices to fill the chip (half the slab traffic)
                const int ks = (ns == 1 && B >= 1024) ? 2 : 4;
                int rc = ns != 1 ? run_bf16<256, 128, 4, 64, EPI_PLAIN, 4>(a, st)
                                 : (ks == 2 ? run4x4_bf16<256, 128, 64, 2>(a, st) : run4x4_bf16<256, 128, 64, 4>(a, st));
                if (rc) return rc;
                return launch_splitk_bias_relu(ws, bias, out, slice, ks, 128, st, ns == 1);
            }
        }
    } else if (width == 128) {
        switch (layer) {
            case 1: return run_bf16<32, 64, 64, 64, EPI_BIAS_BNSTAT>(a, st);
            case 2: return run_bf16<64, 128, 32, 64, EPI_BIAS_BNSTAT>(a, st);
            case 3: return run_bf16<128, 256, 16, 64, EPI_BIAS_BNSTAT>(a, st);
            case 4: return run_bf16<256, 128, 8, 64, EPI_BIAS_RELU>(a, st);
        }
    }
    cvae_set_error("conv_fwd_bf16: unsupported layer %d at width %d", layer, width);
    return -2;
}

int launch_conv_dgrad_bf16(int layer, int width, int ns, int B, const float* dout, const float* packed, float* din, float* ws, hipStream_t st) {
    ConvBf16Args a{dout, pack_ptr(const_cast<float*>(packed), layer, 1, ns >= 3 ? 3 : 1), nullptr, din, nullptr, B, 0, nullptr, ns >= 3 ? pack_units(layer) : 0, ns == 6 ? 6 : 9};
    if (ns == 1) {
        int family;
        const int rc = try_persistent_bf16(layer, width, true, a, st, &family);
        if (family) return rc;
    }
    if (width == 64) {
        switch (layer) {
            case 1: return run_bf16<64, 32, 32, 32, EPI_PLAIN>(a, st);
            case 2: return run_bf16<128, 64, 16, 64, EPI_PLAIN>(a, st);
            // bf16: 64 output channels per workgroup (1.0 LDS fragment reads per MFMA; with 32 the MFMA phase of a stage ran at 40 % of
            // the MFMA rate, LDS-bound at 1.5 reads per MFMA and 3 waves per SIMD: 207 -> 185 us at B = 2048); the emulation modes keep 32
            case 3: if (ns == 1) return run_bf16<256, 128, 8, 64, EPI_PLAIN>(a, st); else return run_bf16<256, 128, 8, 32, EPI_PLAIN>(a, st);
            case 4: {
                const int64_t slice = (int64_t)B * 16 * 256;
                const ConvBf16Args a2 = a;                 // out = din
                a.out = ws; a.sliceFloats = slice;
                if (ns == 1 && B >= 1024) return run4x4_bf16<128, 256, 64, 1>(a2, st);     // enough workgroups without split-K: bf16 result directly
                int rc = ns == 1 ? run4x4_bf16<128, 256, 64, 2>(a, st) : run_bf16<128, 256, 4, 64, EPI_PLAIN, 2>(a, st);
                if (rc) return rc;
                return launch_reduce_slabs(ws, din, slice, 2, slice, st, nullptr, ns == 1);
            }
        }
    } else if (width == 128) {
        switch (layer) {
            case 1: return run_bf16<64, 32, 64, 32, EPI_PLAIN>(a, st);
            case 2: return run_bf16<128, 64, 32, 64, EPI_PLAIN>(a, st);
            case 3: return run_bf16<256, 128, 16, 64, EPI_PLAIN>(a, st);
            case 4: return run_bf16<128, 256, 8, 64, EPI_PLAIN>(a, st);
        }
    }
    cvae_set_error("conv_dgrad_bf16: unsupported layer %d at width %d", layer, width);
    return -2;
}

// Upsample(2)->Conv5x5 of D1..D3 (layers 5..7) on the bf16 MFMA: phase-collapsed 3x3 conv at the stored
// low resolution (see conv_up.hip for the algebra); `in` = stored low-res activation (B,HS,HS,CIN).
int launch_conv_up_fwd_bf16(int layer, int width, int ns, int B, const float* in, const float* packed, const float* bias, float* out, hipStream_t st) {
    ConvBf16Args a{in, pack_ptr(const_cast<float*>(packed), layer, 0, ns >= 3 ? 3 : 1), bias, out, nullptr, B, 0, nullptr, ns >= 3 ? pack_units(layer) : 0, ns == 6 ? 6 : 9};
    if (width == 64) {
        switch (layer) {
            case 5: return run_bf16<128, 256, 4, 64, EPI_PLAIN, 1, 3, MODE_UP_FWD>(a, st);
            case 6: return run_bf16<64, 128, 8, 64, EPI_PLAIN, 1, 3, MODE_UP_FWD>(a, st);
            case 7: return run_bf16<32, 128, 16, 64, EPI_PLAIN, 1, 3, MODE_UP_FWD>(a, st);
        }
    } else if (width == 128) {
        switch (layer) {
            case 5: return run_bf16<128, 256, 8, 64, EPI_PLAIN, 1, 3, MODE_UP_FWD>(a, st);
            case 6: return run_bf16<64, 128, 16, 64, EPI_PLAIN, 1, 3, MODE_UP_FWD>(a, st);
            case 7: return run_bf16<32, 128, 32, 64, EPI_PLAIN, 1, 3, MODE_UP_FWD>(a, st);
        }
    }
    cvae_set_error("conv_up_fwd_bf16: unsupported layer %d at width %d", layer, width);
    return -2;
}
// d_in (B,HS,HS,CIN) = relu'(aux) * sum over phases/taps of dout (B,2HS,2HS,COUT)
int launch_conv_up_dgrad_bf16(int layer, int width, int ns, int B, const float* dout, const float* packed, const float* aux, float* din, hipStream_t st) {
    ConvBf16Args a{dout, pack_ptr(const_cast<float*>(packed), layer, 1, ns >= 3 ? 3 : 1), nullptr, din, nullptr, B, 0, aux, ns >= 3 ? pack_units(layer) : 0, ns == 6 ? 6 : 9};
    if (width == 64) {
        switch (layer) {
            case 5: return run_bf16<256, 128, 4, 32, EPI_PLAIN, 1, 3, MODE_UP_DGRAD>(a, st);
            case 6: return run_bf16<128, 64, 8, 32, EPI_PLAIN, 1, 3, MODE_UP_DGRAD>(a, st);
            case 7: return run_bf16<128, 32, 16, 32, EPI_PLAIN, 1, 3, MODE_UP_DGRAD>(a, st);
        }
    } else if (width == 128) {
        switch (layer) {
            case 5: return run_bf16<256, 128, 8, 32, EPI_PLAIN, 1, 3, MODE_UP_DGRAD>(a, st);
            case 6: return run_bf16<128, 64, 16, 32, EPI_PLAIN, 1, 3, MODE_UP_DGRAD>(a, st);
            case 7: return run_bf16<128, 32, 32, 32, EPI_PLAIN, 1, 3, MODE_UP_DGRAD>(a, st);
        }
    }
    cvae_set_error("conv_up_dgrad_bf16: unsupported layer %d at width %d", layer, width);
    return -2;
}

// ---------------------------------------------------------------------------------------------
// weight gradients on the bf16 MFMA: dW[tap][ci][co] = sum_{img,y,x} in[img][y+r-2][x+s-2][ci] * dy[img][y][x][co]
// (the wgrad half of loss.backward() for nn.Conv2d E2..E4 / D0).  GEMM M = ci (32), N = co (32), K = pixels.
// Both operands want 8 consecutive k (pixels) per lane for a FIXED channel, i.e. a column of the [pixel][channel]
// tensors as they sit in HBM.  ds_read_b64_tr_b16 does that transposition inside the LDS read (4 pixels x 16
// channels per 16-lane group; map verified by profiles/experiments/tr16_probe.hip), so staging is a plain
// 16-byte copy HBM -> registers (one tile ahead, in flight under the MFMAs) -> LDS, with no conversion and no
// shuffling, and a tap shift is just a different row offset of the same halo image.
// One MFMA contracts 16 pixels = 4 quads of 4 x-consecutive pixels (lane half h takes quads 2h, 2h+1).
// Workgroup = 32 ci x 32 co, all 25 taps, over its share of the 256-pixel tiles (split-K slabs, fixed-order
// reduce_slabs as in the fp32 kernel); 8 waves: wave w owns taps w, w+8, w+16 and every 8th pixel group of tap 24
// (80 accumulator registers, so two workgroups = 4 waves per SIMD fit without spilling).
// The bias gradient (column sums of dy) is one more MFMA per 8 pixel groups against an all-ones A operand.
// ---------------------------------------------------------------------------------------------

// WT_NW waves per workgroup (4: two workgroups per CU, 8 accumulator tiles per wave; 8: one workgroup per CU whose
// 8 waves share every staged tile, 5 accumulator tiles per wave — better for the 32-channel layer E2)
#ifndef WGRAD_PIPE
#define WGRAD_PIPE 2       // input fragments requested this many MFMA slots ahead (0: round-4 form — request, wait, MFMA)
#endif
template <int H, int WT_NW, int W, int COB>
__device__ __forceinline__ void wgrad_tr_body(f32x16 (&acc)[COB][24 / WT_NW + 2], const __bf16* lds_in, const __bf16* lds_d, int ibase, int dbase,
                                              bf16x8 ones) {
    using T = WtTile<H>;
    constexpr int JT = 24 / WT_NW;
    static_assert(T::KG % WT_NW == 0, "pixel groups per tile must split evenly over the waves");
    // A slot = one input fragment (two transposed reads) and its COB MFMAs: the JT taps of this wave for every pixel group.
    // Round 5: the fragment of slot n + WGRAD_PIPE is requested before the MFMAs of slot n and the dy fragments of the
    // next pixel group half a group ahead, pinned (one MFMA, then its share of reads).  Round 4 left the order to the compiler, which emitted
    // request -> s_waitcnt lgkmcnt(0) -> MFMA for every slot: a wave's MFMA sat behind the full latency of its own transposed reads and the
    // matrix pipe was busy 47-58 % with two waves per SIMD (profiles/r04_m_pmc_summary_bf16.csv).
    auto in_frag = [&](int kg, int tap) {
        const int r = tap / 5, s = tap % 5;
        const __bf16* q = lds_in + ibase + T::halobase(kg) * 32 + (r * T::HTW + s) * 32;
        return tr_frag(q, q + T::IT * 32);
    };
    auto dy_frag = [&](int kg, int cb) {
        const __bf16* dp = lds_d + cb * T::NPX * 32 + dbase + T::pixbase(kg) * 32;
        return tr_frag(dp, dp + T::DT * 32);
    };
    constexpr int NSLOT = T::KG * JT, PD = WGRAD_PIPE, RING = PD + 1;        // the regular slots: n = kg * JT + j (tap 24 rides outside the pipeline)
    bf16x8 av[RING], bv[2][COB];
#pragma unroll
    for (int cb = 0; cb < COB; ++cb) bv[0][cb] = dy_frag(0, cb);
#pragma unroll
    for (int k = 0; k < PD; ++k) av[k % RING] = in_frag(k / JT, WT_NW * (k % JT) + W);
#pragma unroll
    for (int kg = 0; kg < T::KG; ++kg) {
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const int n = kg * JT + j, na = n + PD;
            if (na < NSLOT) av[na % RING] = in_frag(na / JT, WT_NW * (na % JT) + W);
            if (j == JT / 2 && kg + 1 < T::KG) {
#pragma unroll
                for (int cb = 0; cb < COB; ++cb) bv[(kg + 1) & 1][cb] = dy_frag(kg + 1, cb);
            }
#pragma unroll
            for (int cb = 0; cb < COB; ++cb)
                acc[cb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[n % RING], bv[kg & 1][cb], acc[cb][j], 0, 0, 0);
            if constexpr (PD > 0) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (j == JT / 2 && kg + 1 < T::KG) __builtin_amdgcn_sched_group_barrier(0x100, 2 * COB, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, COB, 0);
            }
        }
        if ((kg % WT_NW) == W) {                                // tap 24: every WT_NW-th pixel group of this wave (1 / 25 of the MFMAs, unpipelined)
            const bf16x8 a24 = in_frag(kg, 24);
#pragma unroll
            for (int cb = 0; cb < COB; ++cb) acc[cb][JT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a24, bv[kg & 1][cb], acc[cb][JT], 0, 0, 0);
        }
        if ((kg % WT_NW) == ((W + 1) % WT_NW)) {
#pragma unroll
            for (int cb = 0; cb < COB; ++cb) acc[cb][JT + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bv[kg & 1][cb], acc[cb][JT + 1], 0, 0, 0);
        }
    }
}

// COB = 32-channel blocks of dy per workgroup.  2 for E2 (32 -> 64 channels, the layer with the largest tensors): the
// staged input tile then serves both halves of the output channels instead of being fetched and staged again by a second
// workgroup (round 2: 805 MB fetched for 403 MB of operands), and every input fragment read from LDS feeds two MFMAs.
template <int CIN, int COUT, int H, int WT_NW, int COB>
__global__ __launch_bounds__(WT_NW * 64, 2) void conv5x5_wgrad_tr_kernel(WgradBf16Args a) {
    using T = WtTile<H>;
    constexpr int WT_NT = WT_NW * 64, JT = 24 / WT_NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* lds_in = reinterpret_cast<__bf16*>(smem_raw);       // [halo pixel][32 ci], 64-byte rows
    __bf16* lds_d = lds_in + T::HP * 32;                        // COB x [pixel][32 co]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x, ci0 = blockIdx.y * 32, n0 = blockIdx.z * 32 * COB;
    // transposed-read address of this lane inside a block: row (pixel) (lane&15)>>2, columns 16*(group&1) + 4*(lane&3)
    const int g = lane >> 4, h = g >> 1, laneoff = ((lane & 15) >> 2) * 32 + 16 * (g & 1) + 4 * (lane & 3);
    const int ibase = h * T::IH * 32 + laneoff, dbase = h * T::DH * 32 + laneoff;

    f32x16 acc[COB][JT + 2];
#pragma unroll
    for (int cb = 0; cb < COB; ++cb)
#pragma unroll
        for (int j = 0; j < JT + 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[cb][j][v] = 0.f;
    bf16x8 zero8, ones;
#pragma unroll
    for (int c = 0; c < 8; ++c) { zero8[c] = (__bf16)0.f; ones[c] = (__bf16)1.f; }

    constexpr int IU = T::HP * 4, DU = T::NPX * 4 * COB, NI = (IU + WT_NT - 1) / WT_NT, ND = (DU + WT_NT - 1) / WT_NT;
    bf16x8 rin[NI], rdo[ND];
    auto fetch = [&](int mt) {
        const int grp = mt / T::TPI, t = mt % T::TPI;
        const int img0 = grp * T::IMGS, ty0 = (t / T::TILES_X) * T::TH, tx0 = (t % T::TILES_X) * T::TW;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = tid + i * WT_NT, hp = q >> 2, oc = q & 3;
            const int img = hp / T::HPI, rem = hp % T::HPI;
            const int gy = ty0 + rem / T::HTW - 2, gx = tx0 + rem % T::HTW - 2, ib = img0 + img;
            const bool ok = (IU % WT_NT == 0 || q < IU) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)H && ib < a.B;
            const size_t e = ok ? ((size_t)(ib * H + gy) * H + gx) * CIN + ci0 + oc * 8 : 0;
            const bf16x8 l = Act<__bf16>::ld8(a.in, e);          // unconditional load from a clamped address + select
            rin[i] = ok ? l : zero8;
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int q = tid + i * WT_NT, px = q / (4 * COB), oc = q % (4 * COB);      // oc: 16-byte unit of the pixel's 32*COB channels
            const int img = px / (T::TH * T::TW), rem = px % (T::TH * T::TW);
            const int gy = ty0 + rem / T::TW, gx = tx0 + rem % T::TW, ib = img0 + img;
            const bool ok = (DU % WT_NT == 0 || q < DU) && ib < a.B;
            const size_t e = ok ? ((size_t)(ib * H + gy) * H + gx) * COUT + n0 + oc * 8 : 0;
            const bf16x8 l = Act<__bf16>::ld8(a.dout, e);
            rdo[i] = ok ? l : zero8;
        }
    };
    const int t0 = split * a.tilesPerSplit;
    int t1 = t0 + a.tilesPerSplit; if (t1 > a.numTiles) t1 = a.numTiles;
    if (t0 < t1) fetch(t0);
    for (int mt = t0; mt < t1; ++mt) {
        __syncthreads();                        // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = tid + i * WT_NT;
            if (IU % WT_NT == 0 || q < IU) *reinterpret_cast<bf16x8*>(lds_in + (size_t)q * 8) = rin[i];
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int q = tid + i * WT_NT;
            if (DU % WT_NT == 0 || q < DU) {
                const int px = q / (4 * COB), oc = q % (4 * COB);
                *reinterpret_cast<bf16x8*>(lds_d + ((size_t)(oc >> 2) * T::NPX + px) * 32 + (oc & 3) * 8) = rdo[i];
            }
        }
        __syncthreads();
        if (mt + 1 < t1) fetch(mt + 1);         // in flight while this tile computes
        switch (wave) {
            case 0: wgrad_tr_body<H, WT_NW, 0, COB>(acc, lds_in, lds_d, ibase, dbase, ones); break;
            case 1: wgrad_tr_body<H, WT_NW, 1, COB>(acc, lds_in, lds_d, ibase, dbase, ones); break;
            case 2: wgrad_tr_body<H, WT_NW, 2, COB>(acc, lds_in, lds_d, ibase, dbase, ones); break;
            case 3: wgrad_tr_body<H, WT_NW, 3, COB>(acc, lds_in, lds_d, ibase, dbase, ones); break;
            case 4: if constexpr (WT_NW == 8) wgrad_tr_body<H, WT_NW, 4, COB>(acc, lds_in, lds_d, ibase, dbase, ones); break;
            case 5: if constexpr (WT_NW == 8) wgrad_tr_body<H, WT_NW, 5, COB>(acc, lds_in, lds_d, ibase, dbase, ones); break;
            case 6: if constexpr (WT_NW == 8) wgrad_tr_body<H, WT_NW, 6, COB>(acc, lds_in, lds_d, ibase, dbase, ones); break;
            default: if constexpr (WT_NW == 8) wgrad_tr_body<H, WT_NW, 7, COB>(acc, lds_in, lds_d, ibase, dbase, ones); break;
        }
    }

    float* out = a.slab + (size_t)split * (25 * CIN * COUT + COUT);     // slab row: [25][CIN][COUT] | bias[COUT]
#pragma unroll
    for (int cb = 0; cb < COB; ++cb)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int ci = ci0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                out[((size_t)(WT_NW * j + wave) * CIN + ci) * COUT + n0 + cb * 32 + li] = acc[cb][j][v];
            }
    // tap 24 and the bias row: every wave holds the partial of its pixel groups -> fixed-order sum through LDS
    float* red = reinterpret_cast<float*>(smem_raw);                    // [NW-1 waves][16][64]
    float* bred = red + (WT_NW - 1) * 16 * 64;                          // [NW waves][32]
#pragma unroll
    for (int cb = 0; cb < COB; ++cb) {
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) red[((wave - 1) * 16 + v) * 64 + lane] = acc[cb][JT][v];
        }
        if (lh == 0) bred[wave * 32 + li] = acc[cb][JT + 1][0];                   // row 0 of (ones x dy) = column sums of dy
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float x = acc[cb][JT][v];
#pragma unroll
                for (int w2 = 0; w2 < WT_NW - 1; ++w2) x += red[(w2 * 16 + v) * 64 + lane];
                const int ci = ci0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                out[((size_t)24 * CIN + ci) * COUT + n0 + cb * 32 + li] = x;
            }
            if (blockIdx.y == 0 && lh == 0) {
                float b = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < WT_NW; ++w2) b += bred[w2 * 32 + li];
                out[(size_t)25 * CIN * COUT + n0 + cb * 32 + li] = b;
            }
        }
    }
}

template <int H>
static int wgrad_bf16_splits(int B, int blocksPerSplit, int targetWgs, int* tilesPerSplit, int* numTilesOut) {
    using T = WtTile<H>;
    const int numTiles = cdiv(B, T::IMGS) * T::TPI;
    int S = cdiv(targetWgs, blocksPerSplit);
    if (S > numTiles) S = numTiles;
    if (S < 1) S = 1;
    const int tps = cdiv(numTiles, S);
    S = cdiv(numTiles, tps);
    *tilesPerSplit = tps; *numTilesOut = numTiles;
    return S;
}

template <int CIN, int COUT, int H>
static int run_wgrad_bf16(int B, const float* in, const float* dout, float* dw, float* dbias, float* ws, hipStream_t st,
                          int64_t* need) {
    using T = WtTile<H>;
    int tps, numTiles;
    constexpr int COB = (CIN == 32 && COUT == 64) ? 2 : 1;              // E2: both output-channel halves in one workgroup
    // split-K workgroups: two per CU for the 4-wave kernels (256 of them: E3 227 / E4 267 us instead of 197 / 202); ONE per CU for E2's 8-wave workgroups
    // (195.6 instead of 207.1 us at B = 2048, and half the slab traffic: 52 instead of 105 MB written and re-read; profiles/r05_o_wgrad_split_count.txt)
    const int S = wgrad_bf16_splits<H>(B, (CIN / 32) * (COUT / (32 * COB)), CIN == 32 ? 256 : 512, &tps, &numTiles);
    const int64_t n = (int64_t)25 * CIN * COUT, row = n + COUT;
    if (need) { *need = (int64_t)(S + 16) * row; return 0; }
    WgradBf16Args a{in, dout, ws, B, numTiles, tps};
    constexpr int WT_NW = CIN == 32 ? 8 : 4, WT_NT = WT_NW * 64;
    constexpr int STAGE = (T::HP + COB * T::NPX) * 64, RED = ((WT_NW - 1) * 16 * 64 + WT_NW * 32) * 4;
    constexpr int SMEM = STAGE > RED ? STAGE : RED;
    auto kern = conv5x5_wgrad_tr_kernel<CIN, COUT, H, WT_NW, COB>;
    static DeviceOnce once;
    { int rc = cvae_grant_lds(once, reinterpret_cast<const void*>(kern), SMEM); if (rc) return rc; }
    cvae_probe_begin(st);
    hipLaunchKernelGGL(kern, dim3(S, CIN / 32, COUT / (32 * COB)), dim3(WT_NT), SMEM, st, a);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    float* mid = ws + (size_t)S * row;
    st = cvae_reduce_stream(st);
    if (dbias == dw + n) return launch_reduce_slabs(ws, dw, row, S, row, st, mid);
    int rc = launch_reduce_slabs(ws, dw, n, S, row, st, mid);
    if (rc || !dbias) return rc;
    return launch_reduce_slabs(ws + n, dbias, COUT, S, row, st, nullptr);
}

static int dispatch_wgrad_bf16(int layer, int width, int B, const float* in, const float* dout, float* dw, float* dbias, float* ws,
                               hipStream_t st, int64_t* need) {
    if (width == 64) {
        switch (layer) {
            case 1: return run_wgrad_bf16<32, 64, 32>(B, in, dout, dw, dbias, ws, st, need);
            case 2: return run_wgrad_bf16<64, 128, 16>(B, in, dout, dw, dbias, ws, st, need);
            case 3: return run_wgrad_bf16<128, 256, 8>(B, in, dout, dw, dbias, ws, st, need);
            case 4: return run_wgrad_bf16<256, 128, 4>(B, in, dout, dw, dbias, ws, st, need);
        }
    } else if (width == 128) {
        switch (layer) {
            case 1: return run_wgrad_bf16<32, 64, 64>(B, in, dout, dw, dbias, ws, st, need);
            case 2: return run_wgrad_bf16<64, 128, 32>(B, in, dout, dw, dbias, ws, st, need);
            case 3: return run_wgrad_bf16<128, 256, 16>(B, in, dout, dw, dbias, ws, st, need);
            case 4: return run_wgrad_bf16<256, 128, 8>(B, in, dout, dw, dbias, ws, st, need);
        }
    }
    cvae_set_error("conv_wgrad_bf16: unsupported layer %d at width %d", layer, width);
    return -2;
}
int64_t wgrad_bf16_ws_floats(int layer, int width, int B) {
    int64_t need = 0;
    if (dispatch_wgrad_bf16(layer, width, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &need) != 0) return 0;
    return need;
}
int launch_conv_wgrad_bf16(int layer, int width, int B, const float* in, const float* dout, float* dw, float* dbias, float* ws, hipStream_t st) {
    return dispatch_wgrad_bf16(layer, width, B, in, dout, dw, dbias, ws, st, nullptr);
}

// ---------------------------------------------------------------------------------------------
// weight gradients of the phase-collapsed up-convs (D1..D3) on the bf16 MFMA:
//   dwc[p][t][ci][co] = sum_{img,y,x} in[img][y+a-1][x+b-1][ci] * dout[img][2y+py][2x+px][co],  t = a*3+b
// Same scheme as conv5x5_wgrad_tr_kernel: the contraction runs over PIXELS, both operands are transposed LDS reads
// (ds_read_b64_tr_b16) of tiles staged as plain 16-byte copies, one tile ahead in registers.  Workgroup = 128 low-res
// pixels x 32 ci x 32 co; the dy tile holds the four phase images [p][pixel][32 co] (dout[2y+py][2x+px]); wave w = output
// phase w with its 9 taps (as conv_up_wgrad_kernel).  Round 3: the earlier version contracted over 8 images, transposed
// 8x8 blocks in registers while staging and had no tile in flight (83 / 45 / 45 us at B = 2048).
// The slab row [36][CIN][COUT] | bias[COUT] is what conv_up.hip's reduce + expand_dw_kernel consume.
// ---------------------------------------------------------------------------------------------
#ifndef UPW_NP
#define UPW_NP 64
#endif
template <int HS> using UpWgTile = WtTile<HS, 1, UPW_NP>;        // low-res pixels per tile, halo 1

template <int CIN, int COUT, int HS>
__global__ __launch_bounds__(256, 2) void conv_up_wgrad_bf16_kernel(WgradBf16Args a) {
    using T = UpWgTile<HS>;
    constexpr int H = 2 * HS;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[(T::HP + 4 * T::NPX) * 64];
    __bf16* lds_in = reinterpret_cast<__bf16*>(smem_raw);       // [halo pixel][32 ci], 64-byte rows
    __bf16* lds_d = lds_in + T::HP * 32;                        // [phase][low-res pixel][32 co]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x, ci0 = blockIdx.y * 32, n0 = blockIdx.z * 32;
    const int g = lane >> 4, h = g >> 1, laneoff = ((lane & 15) >> 2) * 32 + 16 * (g & 1) + 4 * (lane & 3);
    const int ibase = h * T::IH * 32 + laneoff, dbase = wave * T::NPX * 32 + h * T::DH * 32 + laneoff;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float bsum[8];                                              // column sums of this thread's dout units (channels 8*(tid&3)..+7)
#pragma unroll
    for (int c = 0; c < 8; ++c) bsum[c] = 0.f;
    bf16x8 zero8;
#pragma unroll
    for (int c = 0; c < 8; ++c) zero8[c] = (__bf16)0.f;

    constexpr int IU = T::HP * 4, NI = (IU + 255) / 256, ND = 4 * T::NPX * 4 / 256;
    constexpr int PPT = T::NPX / 64;                            // low-res pixels per thread: dout unit i = PPT * phase + j
    static_assert(ND == 4 * PPT && T::NPX % 64 == 0, "unit q = tid + 256 i: phase i / PPT, pixel (tid >> 2) + 64 (i % PPT), octet tid & 3");
    bf16x8 rin[NI], rdo[ND];
    unsigned okm = 0u;                          // bit i: rin[i] valid, bit 8 + i: rdo[i] valid.  Out-of-range units are fetched from
                                                // element 0 and zeroed when staged (a select here waits for the loads on the spot)
    auto fetch = [&](int mt) {
        const int grp = mt / T::TPI, t = mt % T::TPI;
        const int img0 = grp * T::IMGS, ty0 = (t / T::TILES_X) * T::TH, tx0 = (t % T::TILES_X) * T::TW;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = tid + i * 256, hp = q >> 2, oc = q & 3;
            const int img = hp / T::HPI, rem = hp % T::HPI;
            const int gy = ty0 + rem / T::HTW - 1, gx = tx0 + rem % T::HTW - 1, ib = img0 + img;
            const bool ok = (IU % 256 == 0 || q < IU) && (unsigned)gy < (unsigned)HS && (unsigned)gx < (unsigned)HS && ib < a.B;
            rin[i] = Act<__bf16>::ld8(a.in, ok ? ((size_t)(ib * HS + gy) * HS + gx) * CIN + ci0 + oc * 8 : 0);
            okm = ok ? (okm | (1u << i)) : (okm & ~(1u << i));
        }
        // dout unit i = PPT p + j: low-res pixel (tid >> 2) + 64 j of phase p -- one address per pixel, the phase is a constant offset
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int px = (tid >> 2) + 64 * j, img = px / (T::TH * T::TW), rem = px % (T::TH * T::TW), ib = img0 + img;
            const bool ok = ib < a.B;
            const size_t e = ok ? ((size_t)(ib * H + 2 * (ty0 + rem / T::TW)) * H + 2 * (tx0 + rem % T::TW)) * COUT + n0 + (tid & 3) * 8 : 0;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                rdo[PPT * ph + j] = Act<__bf16>::ld8(a.dout, e + (size_t)(((ph >> 1) * H + (ph & 1)) * COUT));
                okm = ok ? (okm | (256u << (PPT * ph + j))) : (okm & ~(256u << (PPT * ph + j)));
            }
        }
    };
    const int t0 = split * a.tilesPerSplit;
    int t1 = t0 + a.tilesPerSplit; if (t1 > a.numTiles) t1 = a.numTiles;
    if (t0 < t1) fetch(t0);
    for (int mt = t0; mt < t1; ++mt) {
        __syncthreads();                        // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = tid + i * 256;
            if (IU % 256 == 0 || q < IU) *reinterpret_cast<bf16x8*>(lds_in + (size_t)q * 8) = (okm >> i) & 1u ? rin[i] : zero8;
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const bf16x8 d = (okm >> (8 + i)) & 1u ? rdo[i] : zero8;
            *reinterpret_cast<bf16x8*>(lds_d + (size_t)(tid + i * 256) * 8) = d;            // unit order IS [phase][pixel][octet]
#pragma unroll
            for (int c = 0; c < 8; ++c) bsum[c] += (float)d[c];
        }
        __syncthreads();
        if (mt + 1 < t1) fetch(mt + 1);         // in flight while this tile computes
#pragma unroll
        for (int kg = 0; kg < T::KG; ++kg) {    // 16 low-res pixels per MFMA; wave = output phase, nine taps each
            const __bf16* dp = lds_d + dbase + T::pixbase(kg) * 32;
            const bf16x8 bv = tr_frag(dp, dp + T::DT * 32);
            const __bf16* ip = lds_in + ibase + T::halobase(kg) * 32;
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const __bf16* q = ip + ((t9 / 3) * T::HTW + t9 % 3) * 32;
                acc[t9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(q, q + T::IT * 32), bv, acc[t9], 0, 0, 0);
            }
        }
    }
    float* out = a.slab + (size_t)split * (36 * CIN * COUT + COUT);
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int ci = ci0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            out[((size_t)(wave * 9 + t9) * CIN + ci) * COUT + n0 + li] = acc[t9][v];
        }
    __syncthreads();
    f32x4* bred = reinterpret_cast<f32x4*>(smem_raw);
    bred[tid * 2] = f32x4{bsum[0], bsum[1], bsum[2], bsum[3]};
    bred[tid * 2 + 1] = f32x4{bsum[4], bsum[5], bsum[6], bsum[7]};
    __syncthreads();
    if (blockIdx.y == 0 && tid < 8) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 64; ++k) s += bred[(k * 4 + (tid >> 1)) * 2 + (tid & 1)];
        *reinterpret_cast<f32x4*>(out + (size_t)36 * CIN * COUT + n0 + tid * 4) = s;
    }
}

template <int CIN, int COUT, int HS>
static int run_up_wgrad_bf16_main(int B, const float* in, const float* dout, float* slab, int Smax, int* S_out, hipStream_t st) {
    using T = UpWgTile<HS>;
    const int numTiles = cdiv(B, T::IMGS) * T::TPI;
    int S = cdiv(2 * cvae_num_cus(), (CIN / 32) * (COUT / 32));     // two resident workgroups per CU, equal tile counts
    if (S > Smax) S = Smax;
    if (S > numTiles) S = numTiles;
    if (S < 1) S = 1;
    const int tps = cdiv(numTiles, S);
    S = cdiv(numTiles, tps);
    WgradBf16Args a{in, dout, slab, B, numTiles, tps};
    cvae_probe_begin(st);
    hipLaunchKernelGGL((conv_up_wgrad_bf16_kernel<CIN, COUT, HS>), dim3(S, CIN / 32, COUT / 32), dim3(256), 0, st, a);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    *S_out = S;
    return 0;
}

// main kernel only: writes S (<= Smax) slab rows [36][CIN][COUT] | bias[COUT]; conv_up.hip reduces and expands them
int launch_up_wgrad_bf16_main(int layer, int width, int B, const float* in, const float* dout, float* slab, int Smax, int* S_out, hipStream_t st) {
    if (width == 64) {
        switch (layer) {
            case 5: return run_up_wgrad_bf16_main<128, 64, 4>(B, in, dout, slab, Smax, S_out, st);
            case 6: return run_up_wgrad_bf16_main<64, 32, 8>(B, in, dout, slab, Smax, S_out, st);
            case 7: return run_up_wgrad_bf16_main<32, 32, 16>(B, in, dout, slab, Smax, S_out, st);
        }
    } else if (width == 128) {
        switch (layer) {
            case 5: return run_up_wgrad_bf16_main<128, 64, 8>(B, in, dout, slab, Smax, S_out, st);
            case 6: return run_up_wgrad_bf16_main<64, 32, 16>(B, in, dout, slab, Smax, S_out, st);
            case 7: return run_up_wgrad_bf16_main<32, 32, 32>(B, in, dout, slab, Smax, S_out, st);
        }
    }
    cvae_set_error("conv_up_wgrad_bf16: unsupported layer %d at width %d", layer, width);
    return -2;
}
