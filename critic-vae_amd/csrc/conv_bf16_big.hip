// conv_bf16_big.hip — PERSISTENT bf16-MFMA 5x5 conv forward / input-gradient kernel on a 16-accumulator-tile WAVE TILE (rounds 4-5).
//
// Same call sites as conv5x5_bf16_kernel<.., MODE_STD, NS = 1> (nn.Conv2d E2..E4, vae_nets.py:74,79,84, and their input gradients under
// loss.backward(), vae.py:57), same packed weights, same LDS images, same k order.  What differs is the register tile: a workgroup
// owns MT 128-pixel tiles x NT channels, every wave 32 pixels of each tile x all NT channels = MT x NT/32 = 16 accumulator tiles
// (256 AGPRs), so one k-step reads MT + NT/32 fragments for 16 MFMAs — 0.5 KB (4 x 4) or 0.625 KB (8 x 2) of LDS per MFMA where the
// two-tile / 64-channel kernels read 1 KB and are bound by exactly that ("Round 4").  One workgroup per CU, one wave per
// SIMD: nothing else is resident to hide anything, so everything that is not an MFMA lives INSIDE the MFMA stream, in fixed slots:
//   * fragments of step i + 1 are requested in the first slots of step i — one ds_read_b128 behind each of the first MFMAs, pinned with
//     sched_group_barrier (round 4 left the placement to the compiler, which sank the reads to three MFMAs in front of their use:
//     6.3 k cycles per stage for 5.1 k of MFMAs, profiles/r05_a_big_timing_round4_kernel.txt);
//   * the weight slab is double-buffered in LDS: the slab of stage g + 1 is requested at step 0 of stage g and written into the other
//     buffer a few steps later; ONE barrier per stage, placed in FRONT of the stage's last step, so that the first fragments of the next
//     stage are requested behind it and travel under that step's 16 MFMAs;
//   * the next chunk's input tiles are requested in the chunk's last stage and written — behind the same barrier — between the MFMAs
//     of its last step (the tiles are single-buffered), a second barrier closes the chunk;
//   * PERSISTENT workgroups (round 5) walk (tile group, channel block) items in an XCD-contiguous order: the next item's tiles, its
//     slab 0 AND its slab 1 are requested before the current item's epilogue (vmcnt retires loads and stores in issue order: a load
//     consumed behind the epilogue's stores would wait for their write acknowledgements), so an item starts without a cold prologue
//     and the epilogue's stores drain under the next item's MFMAs.
// IMGL (round 5): an item's tiles are whole images in LDS — borders zeroed once, only the interiors staged (see the kernel).  S16 (round 5, opt-in through
// CVAE_BIG_S16=1): the same kernel on v_mfma_f32_16x16x32_bf16, on which the power-limited chip holds a 15 % higher clock (profiles/r05_l_mfma_shape_probe.txt,
// r05_m_big_s16.txt: level with the 32x32x16 form so far; see stage16).
// Epilogues, both from channel-major accumulators (v_cvt_pk + two 16-byte buffer stores per tile, no transpose, invalid lanes as
// out-of-range offsets): plain (input gradient), or bias (= the accumulators' initial value) + ONE BatchNorm partial per item — its MT
// tiles summed in the lane before the cross-lane column sums (launch_bn_fwd_finalize(.., tilesPerPartial = MT)).
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "conv_epilogue.h"
#include "conv_bf16.h"

// -DBIG_TIMING (timing builds): cycles of ONE instantiation (-DBIG_T_KCH/NCH/H), wave 0 lane 0 of the sampled workgroups (blockIdx.x a
// multiple of 16 below 256): [items, prologue, stages (MFMA stream with everything interleaved), chunk-closing barriers + first fragments,
// epilogue, whole kernel, s_memrealtime ticks of the whole kernel, stages per item, entry tick, exit tick]
#ifdef BIG_TIMING
__device__ long long big_dbg[16 * 12];
extern "C" int cvae_big_dbg_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(big_dbg), sizeof(big_dbg)); }
#define BT_ON (KCH == BIG_T_KCH && NCH == BIG_T_NCH && H == BIG_T_H)
#define BT(v) do { if (BT_ON) { __builtin_amdgcn_sched_barrier(0); v = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define BT_ON false
#define BT(v)
#endif
// -DBIG_STEPTIME (with -DBIG_TIMING): s_memtime at the top of every step of ONE chunk pair (the second trip of the first item) of workgroup 0, wave 0
#ifdef BIG_STEPTIME
__device__ long long big_steps[128];
extern "C" int cvae_big_steps_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(big_steps), sizeof(big_steps)); }
#endif
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
// Timing experiments (WRONG results, never shipped; profiles/experiments/variant.sh -DBIG_EXP=n): what the stage loop pays for each of its
// parts.  bit 0: no barrier in front of a stage's last step;  bit 1: no slab writes;  bit 2: no slab requests;  bit 3: no tile requests / writes;
// bit 4: no BatchNorm sums in the epilogue;  bit 5: no column sums / partial rows;  bit 6: no output stores;  bit 7: workgroups start staggered
#ifndef BIG_EXP
#define BIG_EXP 0
#endif
#ifndef BIG_ST_AUX
#define BIG_ST_AUX (BN ? 0 : 2)      // cache policy of the output stores: nt (2) for the input gradients — the written lines do not push the input lines the next chunks re-read out of
                                      // L2 (counter fetch of E3's input gradient 260 -> 234 MB, E2's 318 -> 299; kernels -1 %; profiles/r05_r_e2_on_big_kernel.txt); forward outputs stay cached
#endif
#ifndef BIG_ALLC4
#define BIG_ALLC4 false    // experiment builds: the same for E4's input gradient (4 lines per pixel, groups of four chunks)
#endif
#ifndef BIG_ALLC3
#define BIG_ALLC3 false    // ... and E3's (2 lines per pixel)
#endif
#ifndef BIG_ALLC
#define BIG_ALLC true      // false (A/B builds): E2's input gradient requests its tiles chunk by chunk like the others
#endif
#ifndef BIG_IMGL
#define BIG_IMGL true      // false (A/B builds): the per-tile halo layout for the instantiations that existed before the image layout
#endif

// raw buffer descriptor over a whole tensor (gfx9 word 3: DATA_FORMAT_32): a lane whose byte offset is >= bytes is dropped by the
// bounds check, so "store if valid" needs no branch (tensors of 2 GiB and more do not take this kernel: run_big returns -100)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t big_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
static constexpr unsigned BIG_OOB = 0x80000000u;

// S16: which of the wave's 32 pixels (row-major inside the 128-pixel tile) sits behind column c of pixel block jh of a 16x16x32 fragment.  A
// ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32): with k-block q = lane / 16 reading octet plane q & 1
// (plane stride == 8 units mod 16) a group is conflict-free when lanes {0-3, 12-15} hold pixels at unit residues S u (S + 8) and lanes {4-11} the rest:
// lanes 8-11 and 12-15 trade places.  8-pixel rows: a block is rows {0, 2} or {1, 3} of the wave's four (unit offsets 0..7 and 24..31 = residues 0..15).
template <int H>
__device__ __forceinline__ int s16_pix(int jh, int c) {
    const int j = c < 8 ? c : (c < 12 ? c + 4 : c - 4);
    if constexpr (Tile<H>::TW == 8) return ((j >> 3) * 2 + jh) * 8 + (j & 7);
    else return 16 * jh + j;
}
// Sum x[k] (k = 0..15) over the 16 lanes of each DPP row, 45 instructions: lane L returns the total of element L & 15.  Fixed pairing, fixed order.
__device__ __forceinline__ float row_colsum16(const float (&x)[16]) {
    const int lane = threadIdx.x & 63;
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
    float r8[8], r4[4], r2[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) r8[j] = (b3 ? x[j + 8] : x[j]) + dpp_mov<0x128>(b3 ? x[j] : x[j + 8]);            // partner i ^ 8 (row_ror:8)
#pragma unroll
    for (int j = 0; j < 4; ++j) r4[j] = (b2 ? r8[j + 4] : r8[j]) + dpp_mov<0x141>(b2 ? r8[j] : r8[j + 4]);        // partner 7 - i of the 8 (row_half_mirror: bit 2 flipped)
#pragma unroll
    for (int j = 0; j < 2; ++j) r2[j] = (b1 ? r4[j + 2] : r4[j]) + dpp_mov<0x4E>(b1 ? r4[j] : r4[j + 2]);         // partner i ^ 2
    return (b0 ? r2[1] : r2[0]) + dpp_mov<0xB1>(b0 ? r2[0] : r2[1]);                                                // partner i ^ 1
}

template <int KCH, int NCH, int H, int NT, int MT, int KB, int EPI, bool TDB, bool IMGL, bool S16 = false, bool ALLC = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv5x5_bf16_big_kernel(ConvBf16Args a, int numGroups) {
    using T = Tile<H>;
    static_assert(EPI == EPI_PLAIN || EPI == EPI_BIAS_BNSTAT, "epilogues: plain (input gradient) or bias + BatchNorm partials (forward)");
    constexpr bool BN = EPI == EPI_BIAS_BNSTAT;
    constexpr int NB = NT / 32, KS = 5, KCB = 16 * KB, OCT = KCB / 8, NY = NCH / NT;
    constexpr int PSP = Bf16Geom<H, OCT>::PSP, A_UNITS = OCT * PSP, W_UNITS = KS * KB * 2 * NT;
    constexpr int NCHUNK = KCH / KCB, NST = NCHUNK * KS, NSTEP = KS * KB;
    static_assert(KCH % KCB == 0 && NCH % NT == 0 && (MT * NB == 16 || (MT == 8 && NB == 1)) && (KB == 1 || KB == 2), "tiling");       // 8 x 1: a 32-channel layer (E2's input gradient) fills half the accumulators
    // IMGL (image layout, round 5): an item's MT tiles are NIMG WHOLE images, kept in LDS as ONE plane per octet of (H + 4) x (H + 4) halo images.
    // Every halo pixel of a whole image is zero padding: the planes are zeroed once per kernel and only the H x H interiors are ever staged —
    // 128 instead of 288 units per 8 x 8-image tile, 1024 instead of 2304 for a 32 x 32 image cut into eight 4-row tiles (the tiles of the
    // per-tile layout carry a halo each: 98 KB of requests per 16-channel chunk, which is what kept E2 off this kernel).
    // STRIP (64-row images, E2 at 128 x 128 frames): an image does not fit, so the item is a full-width STRIP of SR rows (its MT tiles, TILES_X side by side) with two real
    // halo rows above and below — staged with the interior (out-of-range offsets = zeros at the image's top / bottom edge) — and the two zero columns left and right.
    constexpr int HW = H + 4, NIMG = MT * 128 / (H * H);
    constexpr bool STRIP = IMGL && T::TW != H;
    constexpr int SR = STRIP ? MT * 128 / H : 0;
    static_assert(!IMGL || STRIP || (NIMG >= 1 && NIMG * H * H == MT * 128 && T::TW == H && T::HTW == HW), "image layout: the item's tiles are whole images");
    static_assert(!STRIP || (H % SR == 0 && MT % T::TILES_X == 0 && T::TH * (MT / T::TILES_X) == SR && T::IMGS == 1 && !S16 && !ALLC && 256 / OCT == 2 * H), "strip layout");
    constexpr int PLPX = STRIP ? (SR + 4) * HW : NIMG * HW * HW;          // halo pixels of one octet plane
    constexpr int PSPI = ((PLPX + 15 - Bf16Geom<H, OCT>::PAD) / 16) * 16 + Bf16Geom<H, OCT>::PAD;      // plane stride, == 16 / OCT (mod 16) like PSP
    constexpr int PSX = IMGL ? PSPI : PSP;                     // octet-plane stride of the layout in use
    constexpr int TILE_UNITS = IMGL ? OCT * PSPI : MT * A_UNITS;       // one tile buffer
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* lds_a = reinterpret_cast<bf16x8*>(smem_raw);       // [tile][octet][halo pixel]; IMGL: [octet][image][halo pixel]
    constexpr int TBUFS = TDB ? 2 : 1;                         // TDB: the input tiles are double-buffered too (tile buffer = chunk parity)
    bf16x8* lds_w = lds_a + TBUFS * TILE_UNITS;                // [buffer][tap][kb][half][n]
    constexpr int NSLAB = S16 ? 3 : 2;                         // slab buffers (S16: a ring of three, see stage16)
    // behind the slabs: 256 dump units (4 KB) for the staging stores of units that do not exist, then (forward) the [S | Q][wave][NT] rows of the
    // BatchNorm partials (they live from an item's epilogue until the next item's first stage barrier: not in the dump area) and [NCH] bias
    [[maybe_unused]] float* red = reinterpret_cast<float*>(lds_w + NSLAB * W_UNITS + 256);
    [[maybe_unused]] float* lds_bias = red + 2 * 4 * NT;
    bf16x8* const lds_patch = lds_w + NSLAB * W_UNITS + 256 + (BN ? (2 * 4 * NT + NCH) / 4 : 0);              // [wave][32 rows][8 + 1] units: the epilogue's transposing patch

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    [[maybe_unused]] long long b0 = 0, b1 = 0, b2 = 0, tpro = 0, tstage = 0, tclose = 0, tepi = 0, nit = 0;
    [[maybe_unused]] const long long brt0 = BT_ON ? (long long)wall_clock64() : 0;
    BT(b0);
    [[maybe_unused]] const long long bt_entry = b0;

    // item -> (tile group, channel block): XCD x (= item & 7 while gridDim.x % 8 == 0) owns the contiguous range of groups
    // [x PP, (x+1) PP) — neighbouring tiles share halo rows in ONE L2 — and walks it group by group, the NY channel blocks of a group
    // back to back (the input tiles are re-read from L2, not from HBM); conv_bf16_ps.hip
    const int G = gridDim.x, PP = cdiv(numGroups, 8), numItems = 8 * PP * NY;
    auto decode = [&](int it, int& grp, int& n0) { const int x = it & 7, j = it >> 3, jp = j / NY; n0 = (j - jp * NY) * NT; grp = jp < PP ? x * PP + jp : numGroups; };
    int it = blockIdx.x, grp0, n00;
    decode(it, grp0, n00);
    if (it >= numItems || grp0 >= numGroups) return;           // whole workgroup (G % 8 == 0: once past the end, always past the end)

    // ---- per-thread tables that do not depend on the item ----
    const int m = wave * 32 + lane_pix<H, true>(li);           // pixel (of the 128 of a tile) behind MFMA column li of this wave
    const int pimg = m / (T::TH * T::TW), prem = m % (T::TH * T::TW);
    const int aPix = STRIP ? (prem / T::TW) * HW + (prem % T::TW) : pimg * T::HPI + (prem / T::TW) * T::HTW + (prem % T::TW);
    // epilogue patch of this wave: [32 pixel rows][RU0 = 8 units of 8 channels + 1 pad] = the wave's pixels x 64 channels (two channel blocks: one
    // 128-byte line per pixel); lane (li, lh) writes units 4 nb' + 2 k + lh of row li, and reads back unit (lane % 8) of rows (lane / 8) + 8 k:
    // orow[k] = byte offset of that row's pixel from the tile's first pixel, pcol = of the unit
    constexpr int HB = NB % 2 == 0 ? 2 : 1, NH = NB / HB, RU0 = HB * 4, PRS = RU0 + 1, PPI = 64 / RU0, PIT = 32 / PPI;     // HB = 1 (NB = 1): 64-byte rows, 16 pixels per store instruction
    static_assert(NB % HB == 0, "the epilogue walks the channel blocks in pairs");
    bf16x8* const patch = lds_patch + wave * (32 * PRS);
    bf16x8* const patch_w = patch + li * PRS + lh;
    const bf16x8* const patch_r = patch + (lane / RU0) * PRS + lane % RU0;
    const unsigned pcol = (unsigned)(lane % RU0) * 16u;
    unsigned orow[PIT];
#pragma unroll
    for (int k = 0; k < PIT; ++k) {
        const int mr = wave * 32 + lane_pix<H, true>(lane / RU0 + PPI * k), ir = mr / (T::TH * T::TW), rr = mr % (T::TH * T::TW);
        orow[k] = (unsigned)(((ir * H + rr / T::TW) * H + rr % T::TW) * NCH * 2);
    }
    constexpr int WPT = (W_UNITS + 255) / 256;
    unsigned wbase[WPT];                                       // byte offset inside a stage's slab of the packed weights (without n0)
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int q = tid + i * 256, n = q % NT, row = q / NT, half = row & 1, kb = (row >> 1) % KB, s = row / (2 * KB);
        wbase[i] = (W_UNITS % 256 == 0 || q < W_UNITS) ? (((s * (KCH / 16) + kb) * 2 + half) * NCH + n) * 16 : 0;      // past the slab: unit 0 (never staged)
    }
    // staged 16-byte units per thread and tile buffer: per tile IPT units of its halo image (tables shared by the tiles), or (IMGL) the item's
    // NU units of image interiors
    constexpr int NQ = STRIP ? (SR + 4) * H * OCT : (IMGL ? NIMG * H * H * OCT : T::HP * OCT), IPT = (NQ + 255) / 256;
    constexpr int NU = IMGL ? IPT : MT * IPT;                  // units a thread stages per tile buffer
    static_assert(NU <= 32 && (!IMGL || NQ % 256 == 0), "staged units");
    constexpr int IPTT = IMGL ? 1 : IPT;
    int irel[IPTT];                                            // byte offset of the unit relative to the tile's first pixel, chunk 0
    unsigned ipk[IPTT];                                        // halo row | halo column << 8 | image << 16 | unit exists << 31
    // IMGL: unit u of a thread is pixel (tid / OCT) + u UPX of the item — UPX = 256 / OCT pixels are whole rows, so unit u sits a compile-time
    // number of images and rows behind unit 0 in memory (cgl, carried by the scalar offset of the load) and in LDS (clds, the write's immediate):
    // one lane offset, one LDS address and one image number per thread instead of a table per unit
    constexpr int UPX = 256 / OCT;
    static_assert(!IMGL || (UPX % H == 0 && (H * H) % UPX == 0) || UPX % (H * H) == 0, "a unit step is whole rows of one image, or whole images");
    [[maybe_unused]] unsigned irel0 = 0;
    [[maybe_unused]] int ilds0 = 0, pimg0 = 0;
    [[maybe_unused]] bool top_req = false, bot_req = false;     // STRIP: the item being requested touches the image's top / bottom edge (its halo rows there are padding)
    if constexpr (STRIP) {                                     // unit u = strip rows 2 u, 2 u + 1 (row 0 = two rows above the strip's first), interior columns
        const int oct = tid % OCT, px = tid / OCT, row0 = px / H, x = px % H;
        irel0 = (unsigned)((((row0 - 2) * H + x) * KCH + oct * 8) * 2 + (2 * H + 2) * KCH * 2);
        ilds0 = oct * PSPI + row0 * HW + x + 2;
    } else if constexpr (IMGL) {
        const int oct = tid % OCT, px = tid / OCT, img = px / (H * H), rem = px - img * (H * H), y = rem / H, x = rem % H;
        irel0 = (unsigned)((px * KCH + oct * 8) * 2 + (2 * H + 2) * KCH * 2);     // + IBIAS (the input descriptor starts that far in front of the tensor)
        ilds0 = oct * PSPI + img * (HW * HW) + (y + 2) * HW + x + 2;
        pimg0 = img;
    } else {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256, oct = q % OCT;
            const int hp = q / OCT, img = hp / T::HPI, rem = hp - img * T::HPI;
            const int hy = rem / T::HTW, hx = rem % T::HTW;
            irel[i] = (((img * H + hy - 2) * H + hx - 2) * KCH + oct * 8) * 2;
            ipk[i] = (unsigned)hy | ((unsigned)hx << 8) | ((unsigned)img << 16) | ((NQ % 256 == 0 || q < NQ) ? 0x80000000u : 0u);
        }
    }
    const __amdgpu_buffer_rsrc_t rs_out = big_rsrc(a.out, (unsigned)((size_t)a.B * H * H * NCH * 2));
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_bn = big_rsrc(a.bnpart, BN ? (unsigned)((size_t)2 * numGroups * NCH * 4) : 0u);
    // Operands travel as buffer loads (uniform descriptor + 32-bit lane offset + uniform SGPR offset: no 64-bit lane addresses, vmcnt only —
    // the flat loads hipcc emits for a laundered pointer also count in lgkmcnt and so sit in every fragment wait).  The input descriptor
    // starts IBIAS bytes in front of the tensor so that the halo's negative offsets are non-negative lane offsets (the bounds check of a raw
    // buffer looks at the lane offset alone); zero padding = a lane offset past the end: the load returns 0 without touching memory.
    constexpr int IBIAS = (2 * H + 2) * KCH * 2;
    const __amdgpu_buffer_rsrc_t rs_w = big_rsrc(a.wp, (unsigned)(25 * (KCH / 16) * 2 * NCH * 16));
    const __amdgpu_buffer_rsrc_t rs_in = big_rsrc(reinterpret_cast<const char*>(a.in) - IBIAS, (unsigned)((size_t)a.B * H * H * KCH * 2 + IBIAS));

    // ---- item state (wave-uniform) ----
    struct Item { int n0, grp, img0[MT], ty0[MT], tx0[MT], ibase[MT]; };
    auto setup = [&](int g, int nn0) {
        Item x; x.n0 = nn0; x.grp = g;
#pragma unroll
        for (int tl = 0; tl < MT; ++tl) {                      // a tile index past the end maps to images >= B: loads give 0, stores are dropped
            const int mt = g * MT + tl, tin = mt % T::TILES_PER_IMG;
            x.img0[tl] = (mt / T::TILES_PER_IMG) * T::IMGS;
            x.ty0[tl] = (tin / T::TILES_X) * T::TH; x.tx0[tl] = (tin % T::TILES_X) * T::TW;
            x.ibase[tl] = ((x.img0[tl] * H + x.ty0[tl]) * H + x.tx0[tl]) * KCH * 2;    // bytes
        }
        return x;
    };
    bf16x8 wreg[WPT], breg[NU];
    constexpr int NVO = IMGL ? 1 : NU;
    unsigned voff[NVO];                                        // lane offsets of the staged units of the item whose tiles are being requested (padding: out of range)
    [[maybe_unused]] int nv_req = NIMG;                        // IMGL: images of that item that exist (fewer than NIMG only at the ragged end)
    auto slab_soff = [&](int n0, int st) {                     // byte offset of the slab of stage st = (chunk st / 5, kernel row st % 5), channel block n0
        const int cc = st / KS, r = st - cc * KS;
        return (unsigned)(((r * KS * (KCH / 16) + cc * KB) * 2 * NCH + n0) * 16);
    };
    auto load_w = [&](unsigned soff) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wbase[i], soff, 0));
    };
    auto store_w = [&](int i0, int i1, int buf) {                // units past the slab land in the dump slots behind the slabs (no branch around a store)
#pragma unroll
        for (int i = i0; i < i1; ++i)
            if (i < WPT) lds_w[(W_UNITS % 256 == 0 || tid + i * 256 < W_UNITS) ? buf * W_UNITS + tid + i * 256 : NSLAB * W_UNITS + tid] = wreg[i < WPT ? i : 0];
    };
    auto set_voff = [&](const Item& x) {
        if constexpr (STRIP) {
            top_req = x.ty0[0] == 0; bot_req = x.ty0[0] + SR == H;
        } else if constexpr (IMGL) {                           // interiors only: a unit is missing only when its image is (ragged last item)
            nv_req = a.B - x.img0[0];
        } else {
#pragma unroll
            for (int tl = 0; tl < MT; ++tl)
#pragma unroll
                for (int i = 0; i < IPT; ++i) {
                    const int hy = ipk[i] & 255, hx = (ipk[i] >> 8) & 255, img = (ipk[i] >> 16) & 255;
                    const bool ok = (ipk[i] >> 31) && (unsigned)(x.ty0[tl] + hy - 2) < (unsigned)H && (unsigned)(x.tx0[tl] + hx - 2) < (unsigned)H && x.img0[tl] + img < a.B;
                    voff[tl * IPT + i] = ok ? (unsigned)(irel[i] + IBIAS) : BIG_OOB;
                }
        }
    };
    // unit u of a tile buffer: where it comes from (tsoff[tl] = byte offset of tile tl's first pixel, chunk included; IMGL: tile 0's = the item's)
    // and where it goes (units that do not exist land in the dump slots behind the slabs: no branch around a store)
    auto load_unit = [&](int u, const unsigned (&tsoff)[MT]) {
        if (u < NU) {
            if constexpr (STRIP) {
                const unsigned v = ((u == 0 && top_req) || (u == NU - 1 && bot_req)) ? BIG_OOB : irel0;
                breg[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v, tsoff[0] + (unsigned)(u * UPX * KCH * 2), 0));
            } else if constexpr (IMGL) {
                const int cimg = (u * UPX) / (H * H);
                const unsigned v = (NIMG == 1 || pimg0 < nv_req - cimg) ? irel0 : BIG_OOB;      // the unit's image number stays on the scalar side
                breg[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v, tsoff[0] + (unsigned)(u * UPX * KCH * 2), 0));
            } else breg[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[u], tsoff[u / IPT], 0));
        }
    };
    auto store_unit = [&](int u, int tb) {
        if (u < NU) {
            if constexpr (IMGL) {
                const int clds = STRIP ? (u * UPX / H) * HW : ((u * UPX) / (H * H)) * (HW * HW) + (((u * UPX) % (H * H)) / H) * HW;
                lds_a[tb * TILE_UNITS + clds + ilds0] = breg[u];
            } else {
                const int tl = u / IPT, i = u % IPT, q = tid + i * 256;
                lds_a[(NQ % 256 == 0 || q < NQ) ? (tb * MT + tl) * A_UNITS + (q % OCT) * PSP + q / OCT : TBUFS * TILE_UNITS + NSLAB * W_UNITS + tid] = breg[u];
            }
        }
    };
    auto load_input = [&](const unsigned (&tsoff)[MT]) {
#pragma unroll
        for (int u = 0; u < NU; ++u) load_unit(u, tsoff);
    };
    auto store_input = [&]() {                                 // tile buffer 0
#pragma unroll
        for (int u = 0; u < NU; ++u) store_unit(u, 0);
    };

    // ALLC (round 5, E2's input gradient: KCH = 64, one 128-byte line per pixel): the 16-channel chunks take 32 bytes of every pixel's line, the item walks all its
    // pixels once per chunk, and HBM moves whole lines (profiles/experiments/partial_line_probe.hip) — chunk by chunk each line was fetched 2.4 times.  All NCHUNK
    // chunks of the NEXT item are therefore requested together, in the item's last chunk: chunk 0 goes to LDS as before, chunks 1.. wait in registers (hreg: the 8 x 1
    // wave tile leaves 290 of them unused) and are written into the tile buffer that has just become free, one chunk later each.
    // More than four chunks (KCH > 64: several lines per pixel): the same in GROUPS of four chunks = one line — the group's last chunk requests the next group.
    static_assert(!ALLC || (TDB && IMGL && !S16 && NCHUNK % 4 == 0 && NU <= 8), "ALLC: image layout, double-buffered tiles, chunks in groups of four");
    [[maybe_unused]] bf16x8 hreg[ALLC ? 3 * NU : 1];
    [[maybe_unused]] auto load_unit_to = [&](bf16x8& dst, int uu, unsigned soff) {
        const int cimg = (uu * UPX) / (H * H);
        const unsigned v = (NIMG == 1 || pimg0 < nv_req - cimg) ? irel0 : BIG_OOB;
        dst = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v, soff + (unsigned)(uu * UPX * KCH * 2), 0));
    };
    [[maybe_unused]] auto store_unit_from = [&](const bf16x8& src, int uu, int tb) {
        const int clds = ((uu * UPX) / (H * H)) * (HW * HW) + (((uu * UPX) % (H * H)) / H) * HW;
        lds_a[tb * TILE_UNITS + clds + ilds0] = src;
    };
    f32x16 acc[MT][NB];
    bf16x8 wf[2][NB], xf[2][MT];                               // two fragment sets, alternating per step (they live across stages)
    auto ldf = [&](int set, int i, int r, int buf, int tb = 0) {   // fragments of step i of a stage (kernel row r, slab buffer buf, tile buffer tb)
        const int s = i / KB, kb = i % KB;
        const bf16x8* ap = lds_a + (TDB ? tb : 0) * TILE_UNITS + lh * PSX + aPix + r * (STRIP ? HW : T::HTW);
        const bf16x8* bp = lds_w + buf * W_UNITS + lh * NT + li;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wf[set][nb] = bp[((s * KB + kb) * 2) * NT + nb * 32];
#pragma unroll
        for (int tl = 0; tl < MT; ++tl) {                      // IMGL: tile tl = 128 consecutive pixels of the item's images, its first one at halo pixel toff
            constexpr int TPX = 128;
            const int toff = STRIP ? (tl / T::TILES_X) * T::TH * HW + (tl % T::TILES_X) * T::TW
                                   : (IMGL ? (tl * TPX / (H * H)) * (HW * HW) + ((tl * TPX % (H * H)) / H) * HW : tl * A_UNITS);
            xf[set][tl] = ap[toff + (kb * 2) * PSX + s];
        }
    };

    // One stage = one kernel row x KCB channels = NSTEP steps of 16 MFMAs, straight-line code (every step is ONE scheduling region: a branch
    // inside it would cut the pinned MFMA / fragment-read interleave; what is conditional is made harmless instead — a request of a slab
    // nobody will read, tiles written where no wave looks any more).  P0 = fragment set of step 0 (filled by the previous stage's last step),
    // R = kernel row; gst = running stage count of the workgroup (slab buffer = gst & 1).  The slab stream runs two stages ahead:
    //   steps 0 .. WN - 1: the slab travelling in wreg (requested a stage ago) is written into the other buffer;
    //   step WN:           the slab at `wsoff` is requested into wreg — it is written during the NEXT stage;
    //   R == KS - 3:       the tiles at tsoff[] are requested behind that stage's slab request — vmcnt retires in issue order, so whatever is
    //                      consumed first must be requested first: the two slabs requested after the tiles are written (stage starts of R = KS - 1
    //                      and of the next chunk) 7 and 12 steps later, by when the tiles have landed; requested at step 0 of stage KS - 2 the
    //                      tiles sat in front of a slab that is needed 5 steps later (215 cycles per stage, profiles/r05_c_big_ablation.txt);
    //   R == KS - 1:       they are written in the last step behind the barrier (chunk / item boundary; the tiles are single-buffered).
    constexpr int WN = KB == 2 ? 5 : 2, WU = (WPT + WN - 1) / WN;
    static_assert(WN < NSTEP - 1, "the slab request sits in front of the stage's barrier");
#ifdef BIG_STEPTIME
    bool steptime_on = false, steptime_first = false;
    int steptime_idx = 0;
    const __amdgpu_buffer_rsrc_t rs_steps = big_rsrc(big_steps, (unsigned)sizeof(big_steps));
#endif
    static_assert(!TDB || (KB == 1 && NU <= 12), "the unit schedule of the double-buffered tiles: 6 slots of two units");
    auto stage = [&](auto p0c, auto rc, auto tbc, int gst, unsigned wsoff, const unsigned (&tsoff)[MT], [[maybe_unused]] int ci = 0) {
        constexpr int P0 = decltype(p0c)::value, R = decltype(rc)::value, TB = decltype(tbc)::value;       // TB: tile buffer of this chunk (TDB)
        constexpr int RN = R == KS - 1 ? 0 : R + 1;
        const int buf = gst & 1;
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            constexpr int NR = NB + MT;
            const int set = (P0 + i) & 1;
#ifdef BIG_STEPTIME
            if (BT_ON) {                                       // branch-free: every lane but one carries an out-of-range offset
                __builtin_amdgcn_sched_barrier(0);
                const long long t = clock64();
                const unsigned so = (steptime_on && tid == 0) ? (unsigned)(steptime_idx * 8) : BIG_OOB;
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)t, rs_steps, so, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)(t >> 32), rs_steps, so == BIG_OOB ? BIG_OOB : so + 4u, 0, 0);
                ++steptime_idx;
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            if (i == NSTEP - 1) {
                if (!(BIG_EXP & 1)) __syncthreads();           // slab gst + 1 visible; every wave holds its last fragments of slab gst (and of the tiles)
                if constexpr (TDB) ldf(set ^ 1, 0, RN, buf ^ 1, R == KS - 1 ? TB ^ 1 : TB);      // the next chunk's tiles were written in stages 2..4
                else if constexpr (R == KS - 1) { if (!(BIG_EXP & 8)) store_input(); }
                else ldf(set ^ 1, 0, RN, buf ^ 1);
            } else ldf(set ^ 1, i + 1, R, buf, TB);
            if (i < WN && !(BIG_EXP & 2)) store_w(WU * i, WU * (i + 1), buf ^ 1);
            if (i == WN && !(BIG_EXP & 4)) load_w(wsoff);
            if constexpr (ALLC) {
                // ci = this chunk's index in its group of four (a constant: the group loop is unrolled).  Last chunk of a group: the next group's 4 x NU units (the
                // next item's first group behind the item's last chunk) are requested four per step in rows 0..3 (its first chunk first: that one is written below, in
                // this chunk); other chunks: no request, chunk ci + 1 of the group leaves hreg for the free tile buffer.
                if (ci == 3 && R <= 3 && (i == 3 || i == 4) && !(BIG_EXP & 8)) {
                    const int sl = R * 2 + (i - 3);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int u = 4 * sl + q, k = u / NU, uu = u % NU;
                        if (u < 4 * NU) { if (k == 0) load_unit_to(breg[uu], uu, tsoff[0]); else load_unit_to(hreg[(k - 1) * NU + uu], uu, tsoff[0] + (unsigned)(k * KCB * 2)); }
                    }
                }
                if ((((R == 2 || R == 3) && (i == 3 || i == 4)) || (R == 4 && (i == 2 || i == 3))) && !(BIG_EXP & 8)) {
                    const int sl = R == 4 ? 4 + (i - 2) : (R - 2) * 2 + (i - 3);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int uu = 2 * sl + q;
                        if (uu < NU) { if (ci == 3) store_unit_from(breg[uu], uu, TB ^ 1); else store_unit_from(hreg[(ci < 3 ? ci : 0) * NU + uu], uu, TB ^ 1); }
                    }
                }
            } else if constexpr (TDB) {
                // the tile stream, two units per step (the 12 requests of a chunk in ONE step held the wave at the texture unit for 800 cycles, and
                // their 12 LDS writes + a second barrier closed every chunk: 1.7 k of a 16.4 k-cycle chunk, profiles/r05_f_big_steptime.txt)
                if (R <= 2 && (i == 3 || i == 4) && !(BIG_EXP & 8)) { const int sl = R * 2 + (i - 3); load_unit(2 * sl, tsoff); load_unit(2 * sl + 1, tsoff); }
                if ((R == 2 || R == 3) && (i == 3 || i == 4) && !(BIG_EXP & 8)) { const int sl = (R - 2) * 2 + (i - 3); store_unit(2 * sl, TB ^ 1); store_unit(2 * sl + 1, TB ^ 1); }
                if (R == 4 && (i == 2 || i == 3) && !(BIG_EXP & 8)) { const int sl = 4 + (i - 2); store_unit(2 * sl, TB ^ 1); store_unit(2 * sl + 1, TB ^ 1); }
            } else {
                if (i == WN + 1 && R == KS - 3 && !(BIG_EXP & 8)) load_input(tsoff);
            }
#pragma unroll
            for (int tl = 0; tl < MT; ++tl)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)                // weights as the A operand: D[channel][pixel] (channel-major, conv_epilogue.h)
                    acc[tl][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[set][nb], xf[set][tl], acc[tl][nb], 0, 0, 0);
            // pinned: one fragment read behind each of the first NR MFMAs, then the step's LDS writes / requests two per gap
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (k < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                else __builtin_amdgcn_sched_group_barrier(0x220, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- S16 (round 5): the same tiles, slabs and staging on v_mfma_f32_16x16x32_bf16 — the chip holds a higher clock on that shape under its power
    // limit (profiles/r05_l_mfma_shape_probe.txt: +14 % sustained).  k = 32 of one MFMA = TWO TAPS x the chunk's 16 channels: lane (c, q) of a
    // fragment reads the unit of tap A (q < 2) or tap B (q >= 2), octet q & 1 — the LDS images stay as they are, only the lane -> address map changes.
    // The wave tile is 2 MT pixel blocks (16 pixels: rows 16 jh .. + 15 of the wave's 32 in tile tl) x 2 NB channel blocks of 16; acc16[j][i], lane
    // (c, q) = pixel c of block j, channels 16 i + 4 q .. + 3.  25 taps of a (chunk, 5 kernel rows) = 12.5 steps: a chunk PAIR is 25 steps — stage
    // g = 5 h + r (h = chunk of the pair, r = kernel row) runs the steps whose first tap lies in its slab: three (taps 0|1, 2|3, 4|next slab's 0) when g
    // is even, two (1|2, 3|4) when odd; the crossing step reads slab g + 1 and (r = 4) the next chunk's tiles, both visible behind the barrier that
    // stands in front of every stage's last step anyway.
    constexpr int NJ = S16 ? 2 * MT : 1, NI = S16 ? 2 * NB : 1;
    [[maybe_unused]] f32x4 acc16[NJ][NI];
    [[maybe_unused]] bf16x8 wf16[2][NI], xf16[2][NJ];          // two fragment sets, alternating per step
    [[maybe_unused]] const int c16 = lane & 15, q4 = lane >> 4;
    [[maybe_unused]] const bool hiq = q4 >= 2;
    [[maybe_unused]] int xb16[2], wb16 = 0;
    if constexpr (S16) {
        static_assert(!S16 || (TDB && IMGL && KB == 1 && NCHUNK % 2 == 0 && NT % 64 == 0), "S16: image layout, double-buffered tiles, 16-channel chunks");
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
            const int mj = wave * 32 + s16_pix<H>(jh, c16), pi = mj / (T::TH * T::TW), pr = mj % (T::TH * T::TW);
            xb16[jh] = (q4 & 1) * PSX + pi * T::HPI + (pr / T::TW) * T::HTW + (pr % T::TW);
        }
        wb16 = (q4 & 1) * NT + c16;
    }
    // Fragment reads = ONE lane pointer per KIND of tap pair + a compile-time offset (the ds_read's immediate): the lanes of tap B (q >= 2) sit a
    // constant distance D behind tap A's — kind 0: the next column of the same kernel row (D = 1 unit; weights: 2 NT), kind 1: tap 4 of a row | tap 0 of
    // the next row (D = HTW - 4), kind 2: tap (4, 4) of a chunk | tap (0, 0) of the next chunk in the other tile buffer (D = TILE_UNITS - 4 HTW - 4).
    // (Per-step lane pointers with both taps' offsets folded in are 50 loop invariants: hoisted, spilled, and every reload inside the loop is followed
    // by s_waitcnt vmcnt(0) — behind the slab and tile requests in flight: 5.1 k instead of 2.8 k cycles per stage.)
    [[maybe_unused]] int xbk[3][2], wbh = 0;
    if constexpr (S16) {
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
            xbk[0][jh] = xb16[jh] + (hiq ? 1 : 0);
            xbk[1][jh] = xb16[jh] + (hiq ? T::HTW - 4 : 0);
            xbk[2][jh] = xb16[jh] + (hiq ? TILE_UNITS - 4 * T::HTW - 4 : 0);
        }
        wbh = wb16 + (hiq ? 2 * NT : 0);
    }
    // weights of a tap pair (sA, sA + 1) inside ring slot `slot`
    auto ldw16 = [&](int set, int sA, int slot) {
        const bf16x8* wp = lds_w + wbh + slot * W_UNITS;
#pragma unroll
        for (int i = 0; i < NI; ++i) wf16[set][i] = wp[sA * 2 * NT + 16 * i];
    };
    // weights of the crossing pair: tap 4 of ring slot b0 | tap 0 of ring slot b1
    auto ldw16x = [&](int set, int b0, int b1) {
        const bf16x8* wp = lds_w + wb16 + (hiq ? b1 * W_UNITS : b0 * W_UNITS + 8 * NT);
#pragma unroll
        for (int i = 0; i < NI; ++i) wf16[set][i] = wp[16 * i];
    };
    // pixel fragments: kind of the pair, cA = tap A's compile-time unit offset (tile buffer, kernel row, column)
    auto ldx16 = [&](int set, int kind, int cA) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int tl = j >> 1, toff = (tl * 128 / (H * H)) * (HW * HW) + ((tl * 128 % (H * H)) / H) * HW;
            xf16[set][j] = (lds_a + xbk[kind][j & 1])[cA + toff];
        }
    };
    // One stage on the 16x16x32 shape.  HC = chunk of the pair (= tile buffer), R = kernel row, P0 = fragment set of its first step; sb = ring slot of
    // the stage's slab (THREE slab buffers: g in sb, g + 1 in sb + 1, and slab g + 2 — travelling in wreg — is written into sb + 2 during this stage).
    //   step 0: fragments of step 1 (own slab) are read, nothing else;
    //   step 1: BARRIER in front of it — the crossing step's fragments (THREE: tap 4 | tap 0 of slab g + 1, at R = 4 of the next chunk's tiles) or the next
    //           stage's first fragments (two-step stage) lie behind it; then slab g + 2 goes from wreg into ring slot sb + 2 (its previous tenant g - 1 was
    //           last read right behind the PREVIOUS stage's barrier: every wave is past that) and slab g + 3 (at wsoff) is requested;
    //   step 2 (THREE): the next stage's first fragments.
    //   Tile units of the next chunk: requested two per step at (R, step) = (0,1) (1,0) (1,1) (2,0), written two stages later at (2,1) (3,0) (3,1) (4,0) —
    //   all in front of stage 4's barrier, behind which the crossing step reads them.
    auto stage16 = [&](auto hc, auto rc, auto p0c, int sb, unsigned wsoff, const unsigned (&tsoff)[MT]) {
        constexpr int HC = decltype(hc)::value, R = decltype(rc)::value, P0 = decltype(p0c)::value, TB = HC;
        constexpr bool THREE = ((HC * KS + R) & 1) == 0;
        constexpr int NSTP = THREE ? 3 : 2;
        constexpr int RN = R == KS - 1 ? 0 : R + 1, TBN = R == KS - 1 ? TB ^ 1 : TB;
        const int b0 = sb, b1 = sb == 2 ? 0 : sb + 1, b2 = b1 == 2 ? 0 : b1 + 1;
#pragma unroll
        for (int i = 0; i < NSTP; ++i) {
            const int set = (P0 + i) & 1;
#ifdef BIG_STEPTIME
            if (BT_ON) {
                __builtin_amdgcn_sched_barrier(0);
                const long long t = clock64();
                const unsigned so = (steptime_on && tid == 0) ? (unsigned)(steptime_idx * 8) : BIG_OOB;
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)t, rs_steps, so, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)(t >> 32), rs_steps, so == BIG_OOB ? BIG_OOB : so + 4u, 0, 0);
                ++steptime_idx;
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            // the NEXT step's taps: step 0 -> own slab (THREE: 2|3, else 3|4); step 1 -> behind the barrier: THREE: 4 | next slab's 0, else the next stage's
            // first step (0|1 of slab g + 1); step 2 (THREE) -> the next stage's first step (1|2 of slab g + 1)
            if (i == 0) { ldw16(set ^ 1, THREE ? 2 : 3, b0); ldx16(set ^ 1, 0, TB * TILE_UNITS + R * T::HTW + (THREE ? 2 : 3)); }
            if (i == 1) {
                if (!(BIG_EXP & 1)) __syncthreads();
                if constexpr (THREE) { ldw16x(set ^ 1, b0, b1); ldx16(set ^ 1, R == KS - 1 ? 2 : 1, TB * TILE_UNITS + R * T::HTW + 4); }
                else { ldw16(set ^ 1, 0, b1); ldx16(set ^ 1, 0, TBN * TILE_UNITS + RN * T::HTW); }
                if (!(BIG_EXP & 2)) store_w(0, WPT, b2);
                if (!(BIG_EXP & 4)) load_w(wsoff);
            }
            if (i == 2) { ldw16(set ^ 1, 1, b1); ldx16(set ^ 1, 0, TBN * TILE_UNITS + RN * T::HTW + 1); }
            if (!(BIG_EXP & 8)) {
                constexpr int RQ = R == 0 ? 0 : (R == 1 ? 1 : 3);             // first request slot of row R (rows 0..2), first write slot of row R + 2
                if ((R == 0 && i == 1) || (R == 1 && i <= 1) || (R == 2 && i == 0)) { const int sl = RQ + (R == 1 ? i : 0); load_unit(2 * sl, tsoff); load_unit(2 * sl + 1, tsoff); }
                if ((R == 2 && i == 1) || (R == 3 && i <= 1) || (R == 4 && i == 0)) {
                    const int sl = (R == 2 ? 0 : (R == 3 ? 1 : 3)) + (R == 3 ? i : 0);
                    store_unit(2 * sl, TB ^ 1); store_unit(2 * sl + 1, TB ^ 1);
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int ii = 0; ii < NI; ++ii)                // weights as the A operand: D[channel][pixel]
                    acc16[j][ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf16[set][ii], xf16[set][j], acc16[j][ii], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {                     // pinned: a fragment read per four MFMAs (LDS at half its rate), the step's writes / requests between
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x220, 2, 0);      // (one per gap stretches wreg's live range: 92 spills)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Forward: the finished item's per-wave rows [S | Q][wave][NT] -> ONE (sum, M2) partial per channel.  Deferred: a barrier of its own in the
    // epilogue waited for the slowest wave's store issue (8.5 k of the 16 k-cycle epilogue, profiles/r05_d_big_epilogue_ablation.txt); the rows are
    // combined behind the first stage barrier of the NEXT item instead (or behind one barrier at the very end), no branch: without a finished item
    // the two stores carry an out-of-range offset.
    [[maybe_unused]] int pd_grp = 0, pd_n0 = 0, pd_cnt = 0;
    [[maybe_unused]] bool have_pd = false;
    auto bn_combine = [&]() {
        if constexpr (BN && !(BIG_EXP & 32)) {
            const int c = tid < NT ? tid : 0;
            float S = 0.f, Q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { S += red[(0 * 4 + w) * NT + c]; Q += red[(1 * 4 + w) * NT + c]; }
            // M2 = Q - S^2 / n in double; a full item's n is a constant (no double-precision division on the path every item takes)
            const double rn = pd_cnt == MT * 128 ? 1.0 / (MT * 128) : (pd_cnt > 0 ? 1.0 / (double)pd_cnt : 0.0);
            const double m2 = (double)Q - (double)S * (double)S * rn;
            const bool ok = have_pd && tid < NT;
            const unsigned o = (unsigned)(pd_grp * NCH + pd_n0 + c) * 4u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, S), rs_bn, ok ? o : BIG_OOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(m2 > 0.0 ? m2 : 0.0)), rs_bn,
                                                  ok ? o + (unsigned)(numGroups * NCH) * 4u : BIG_OOB, 0, 0);
        }
    };
    if (BIG_EXP & 128) { for (int k = 0; k < ((int)(blockIdx.x >> 3) & 7); ++k) __builtin_amdgcn_s_sleep(30); }      // 8 phases, ~1.9 k cycles apart
    // ---- prologue: the first item's tiles and slab 0 into LDS, its slab 1 on the way ----
    if constexpr (BN) { for (int c = tid; c < NCH; c += 256) lds_bias[c] = a.bias[c]; }
    if constexpr (IMGL) {                                      // the zero padding around every image, once: staging never writes there
        for (int i = tid; i < TBUFS * TILE_UNITS; i += 256) lds_a[i] = bf16x8{};
        __syncthreads();
    }
    Item cur = setup(grp0, n00);
    set_voff(cur);
    {
        unsigned ts[MT];
#pragma unroll
        for (int tl = 0; tl < MT; ++tl) ts[tl] = (unsigned)cur.ibase[tl];
        load_w(slab_soff(cur.n0, 0));
        load_input(ts);
        if constexpr (ALLC) {
#pragma unroll
            for (int k = 1; k < 4; ++k)
#pragma unroll
                for (int uu = 0; uu < NU; ++uu) load_unit_to(hreg[(k - 1) * NU + uu], uu, ts[0] + (unsigned)(k * KCB * 2));
        }
    }
    store_input();
    store_w(0, WPT, 0);
    load_w(slab_soff(cur.n0, 1));
    if constexpr (S16) {                                       // ring slots 0 and 1 filled, slab 2 on the way (stage 0 writes it into slot 2)
        store_w(0, WPT, 1);
        load_w(slab_soff(cur.n0, 2));
        __syncthreads();
        ldw16(0, 0, 0);
        ldx16(0, 0, 0);
    } else {
    __syncthreads();
    ldf(0, 0, 0, 0);
    }
    BT(b1);
#ifdef BIG_TIMING
    if (BT_ON) tpro = b1 - bt_entry;
#endif

    int gst = 0;
    [[maybe_unused]] int sb16 = 0;                             // S16: ring slot of the current stage's slab
#ifdef BIG_STEPTIME
    steptime_first = true;
#endif
    for (;;) {
        int itn = it + G, grpn, n0n;
        decode(itn, grpn, n0n);
        const bool have_next = itn < numItems && grpn < numGroups;
        if constexpr (S16) {
#pragma unroll
            for (int ii = 0; ii < NI; ++ii) {
                f32x4 bq4 = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (BN) bq4 = *reinterpret_cast<const f32x4*>(lds_bias + cur.n0 + 16 * ii + 4 * q4);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc16[j][ii] = bq4;
            }
        } else
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 bq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (BN) bq[g] = *reinterpret_cast<const f32x4*>(lds_bias + cur.n0 + nb * 32 + 8 * g + 4 * lh);
                else bq[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int tl = 0; tl < MT; ++tl)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[tl][nb][v] = bq[v >> 2][v & 3];
        }
        BT(b1);
        // Two chunks per trip (the fragment sets alternate per step, a chunk has KS x NSTEP steps), kernel rows unrolled.  Slab st + 1 is
        // travelling in wreg when stage st starts (requested by stage st - 1, by the previous item's last stage, or by the prologue).
        static_assert(NCHUNK % 2 == 0, "chunks are walked in pairs");
        constexpr int GP = ALLC ? 2 : 1;                       // ALLC: a trip is a GROUP of four chunks (two pairs), so that the chunk's index in its group is a constant
        for (int cq = 0; cq < NCHUNK; cq += 2 * GP) {
#pragma unroll
        for (int g2 = 0; g2 < GP; ++g2) {
            const int cp = cq + 2 * g2;
#ifdef BIG_STEPTIME
            steptime_on = BT_ON && blockIdx.x == 0 && steptime_first && cp == (NCHUNK >= 4 ? 2 : 0);
            steptime_idx = 0;
#endif
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cc = cp + h;
                const bool lastc = h == 1 && cc == NCHUNK - 1;
                unsigned ts[MT];                               // where the next tiles come from: the next chunk of this item, or chunk 0 of the next item
#pragma unroll
                for (int tl = 0; tl < MT; ++tl) ts[tl] = (unsigned)(cur.ibase[tl] + (lastc ? 0 : cc + 1) * KCB * 2);
                if (lastc && have_next) {                      // between two stages (its own block): the next item's lane offsets and tile bases
                    const Item nx = setup(grpn, n0n);
                    set_voff(nx);
#pragma unroll
                    for (int tl = 0; tl < MT; ++tl) ts[tl] = (unsigned)nx.ibase[tl];
                }
                [[maybe_unused]] long long b3 = 0, b4 = 0;
                BT(b2);
#pragma unroll
                for (int r = 0; r < KS; ++r) {
                    const int s2 = cc * KS + r + (S16 ? 3 : 2);    // the slab requested in this stage: two (S16: three) stages ahead, across the item boundary
                    const bool own = s2 < NST;
                    const unsigned wsoff = slab_soff(own || !have_next ? cur.n0 : n0n, own ? s2 : (have_next ? s2 - NST : 0));
                    constexpr int P0v[2][5] = {{0, NSTEP & 1, 0, NSTEP & 1, 0}, {NSTEP & 1, 0, NSTEP & 1, 0, NSTEP & 1}};
                    if constexpr (S16) {
                        // fragment set of a stage's first step = parity of the steps before it in the pair (3, 2, 3, 2 ... steps per stage)
                        constexpr int P16[2][5] = {{0, 1, 1, 0, 0}, {1, 1, 0, 0, 1}};
                        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
                        if (h == 0) {
                            if (r == 0) stage16(I0{}, std::integral_constant<int, 0>{}, std::integral_constant<int, P16[0][0]>{}, sb16, wsoff, ts);
                            if (r == 1) stage16(I0{}, std::integral_constant<int, 1>{}, std::integral_constant<int, P16[0][1]>{}, sb16, wsoff, ts);
                            if (r == 2) stage16(I0{}, std::integral_constant<int, 2>{}, std::integral_constant<int, P16[0][2]>{}, sb16, wsoff, ts);
                            if (r == 3) stage16(I0{}, std::integral_constant<int, 3>{}, std::integral_constant<int, P16[0][3]>{}, sb16, wsoff, ts);
                            if (r == 4) stage16(I0{}, std::integral_constant<int, 4>{}, std::integral_constant<int, P16[0][4]>{}, sb16, wsoff, ts);
                        } else {
                            if (r == 0) stage16(I1{}, std::integral_constant<int, 0>{}, std::integral_constant<int, P16[1][0]>{}, sb16, wsoff, ts);
                            if (r == 1) stage16(I1{}, std::integral_constant<int, 1>{}, std::integral_constant<int, P16[1][1]>{}, sb16, wsoff, ts);
                            if (r == 2) stage16(I1{}, std::integral_constant<int, 2>{}, std::integral_constant<int, P16[1][2]>{}, sb16, wsoff, ts);
                            if (r == 3) stage16(I1{}, std::integral_constant<int, 3>{}, std::integral_constant<int, P16[1][3]>{}, sb16, wsoff, ts);
                            if (r == 4) stage16(I1{}, std::integral_constant<int, 4>{}, std::integral_constant<int, P16[1][4]>{}, sb16, wsoff, ts);
                        }
                        sb16 = sb16 == 2 ? 0 : sb16 + 1;
                    } else
                    if (h == 0) {
                        if (r == 0) stage(std::integral_constant<int, P0v[0][0]>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, gst, wsoff, ts, 2 * g2 + h);
                        if (r == 1) stage(std::integral_constant<int, P0v[0][1]>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, gst, wsoff, ts, 2 * g2 + h);
                        if (r == 2) stage(std::integral_constant<int, P0v[0][2]>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, gst, wsoff, ts, 2 * g2 + h);
                        if (r == 3) stage(std::integral_constant<int, P0v[0][3]>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}, gst, wsoff, ts, 2 * g2 + h);
                        if (r == 4) stage(std::integral_constant<int, P0v[0][4]>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 0>{}, gst, wsoff, ts, 2 * g2 + h);
                    } else {
                        if (r == 0) stage(std::integral_constant<int, P0v[1][0]>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, gst, wsoff, ts, 2 * g2 + h);
                        if (r == 1) stage(std::integral_constant<int, P0v[1][1]>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, gst, wsoff, ts, 2 * g2 + h);
                        if (r == 2) stage(std::integral_constant<int, P0v[1][2]>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, gst, wsoff, ts, 2 * g2 + h);
                        if (r == 3) stage(std::integral_constant<int, P0v[1][3]>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, gst, wsoff, ts, 2 * g2 + h);
                        if (r == 4) stage(std::integral_constant<int, P0v[1][4]>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{}, gst, wsoff, ts, 2 * g2 + h);
                    }
                    ++gst;
                    if (BN && h == 0 && r == 0 && cp == 0) bn_combine();      // behind stage 0's barrier: the previous item's rows are complete
                }
                BT(b3);
                if constexpr (!TDB) {
                    __syncthreads();                           // chunk / item boundary: the new tiles are visible behind this barrier
                    if (!lastc) ldf(h == 0 ? (NSTEP & 1) : 0, 0, 0, gst & 1);
                }
                BT(b4);
#ifdef BIG_TIMING
                if (BT_ON) { tstage += b3 - b2; tclose += b4 - b3; }
#endif
            }
            if constexpr (S16) {                               // a chunk pair is 25 steps: its last step left the next first fragments in set 1 — every pair starts from set 0
#pragma unroll
                for (int ii = 0; ii < NI; ++ii) wf16[0][ii] = wf16[1][ii];
#pragma unroll
                for (int j = 0; j < NJ; ++j) xf16[0][j] = xf16[1][j];
            }
        }
        }
        BT(b1);

        // ---- epilogue: lane = pixel m of each tile, element v of block nb = channel 32 nb + (v & 3) + 8 (v >> 2) + 4 lh (channel-major accumulators);
        // u[k] = channels 16 k + 8 lh .. + 7 of a 32-channel block.  Forward: ONE BatchNorm partial per item and channel — (sum, M2 about the
        // mean) of its MT tiles' valid pixels, which launch_bn_fwd_finalize(.., tilesPerPartial = MT) merges (nn.BatchNorm2d train-mode
        // statistics, vae_nets.py:75,80,85): the tiles are added up in the lane first, so the cross-lane column sums (half_wave_colsum16)
        // run once per channel block instead of once per accumulator tile.
        bool validv[MT];
        bool allv = true;
#pragma unroll
        for (int tl = 0; tl < MT; ++tl) { validv[tl] = cur.img0[tl] + pimg < a.B; allv = allv && (cur.img0[tl] + T::IMGS <= a.B); }
        // Two channel blocks (64 channels = one 128-byte line per pixel) at a time, tile by tile: the wave's 32 pixels x 64 channels go through its
        // private LDS patch (rows = pixels, 16-byte units of 8 channels, one pad unit per row: conflict-free ds_write_b128) and leave as WHOLE lines —
        // 8 consecutive lanes store the 128 contiguous bytes of one pixel, so a store instruction touches 8 full lines.  Straight from the accumulator
        // layout a lane stores 16 bytes of ITS pixel: 64 lines per instruction, and the 16-tile epilogue was bound by exactly that (9.5 k cycles for
        // 128 KB per CU, profiles/r05_b_big_timing_persistent.txt).  A wave's LDS operations execute in order: no wait between the patch's writes
        // and reads.  Forward: the BatchNorm sums of the pair's 2 x 16 channels per lane are taken from the same register copies.
        if constexpr (S16) {
            // 16x16 accumulator tiles: lane (c, q) holds channels 16 i + 4 q .. + 3 of pixel c of block j.  64 channels (four channel blocks = one 128-byte
            // line per pixel) at a time, block by block, through the wave's patch [16 pixel rows][8 + 1 units]: the lane writes its 8 bytes of every
            // channel block into row c (conflict-free ds_write_b64), reads back unit lane % 8 of rows lane / 8 and lane / 8 + 8, and 8 consecutive lanes
            // store one pixel's whole line.  Forward: per-lane sums of the 16 (channel block, e) values over the wave's blocks, then row_colsum16 over
            // the 16 pixel lanes of each DPP row: lane (c, q) ends with channel 64 hp + 16 (c >> 2) + 4 q + (c & 3) — 64 lanes, 64 channels.
            constexpr int NPASS = NT / 64;
            // the epilogue's lane tables are rebuilt here from an opaque copy of the lane id: as loop invariants they would be hoisted over the stage loop,
            // where the register file has no room for them (spilled, and every reload inside the loop waits for vmcnt(0))
            int lz = lane;
            asm volatile("" : "+v"(lz));
            const int cz = lz & 15, qz = lz >> 4;
            char* const pw = reinterpret_cast<char*>(lds_patch + wave * (16 * 9)) + cz * 144 + 8 * qz;
            const bf16x8* const pr = lds_patch + wave * (16 * 9) + (lz >> 3) * 9 + (lz & 7);
            const unsigned pcol16 = (unsigned)(lz & 7) * 16u;
            unsigned orow16[2][2];                             // byte offset (from the tile's first pixel) of the pixel rows lane / 8 and lane / 8 + 8 of block jh
#pragma unroll
            for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int mr = wave * 32 + s16_pix<H>(jh, (lz >> 3) + 8 * k), ir = mr / (T::TH * T::TW), rr = mr % (T::TH * T::TW);
                    orow16[jh][k] = (unsigned)(((ir * H + rr / T::TW) * H + rr % T::TW) * NCH * 2);
                }
            if constexpr (BN) {
                if (!allv) {
#pragma unroll
                    for (int tl = 0; tl < MT; ++tl)
                        if (!validv[tl]) {
#pragma unroll
                            for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                                for (int ii = 0; ii < NI; ++ii) acc16[2 * tl + jh][ii] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                }
            }
#pragma unroll
            for (int hp = 0; hp < NPASS; ++hp) {
                [[maybe_unused]] float sv[16], qv[16];
                if constexpr (BN) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) { sv[k] = 0.f; qv[k] = 0.f; }
                }
                bf16x8 rows[2];
                unsigned rbase = BIG_OOB;
                auto flush16 = [&](int jhp) {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (!(BIG_EXP & 64)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, rows[k]), rs_out, rbase == BIG_OOB ? BIG_OOB : rbase + orow16[jhp][k], 0, 0);
                };
#pragma unroll
                for (int tl = 0; tl < MT; ++tl)
#pragma unroll
                    for (int jh = 0; jh < 2; ++jh) {
                        const int j = 2 * tl + jh;
#pragma unroll
                        for (int il = 0; il < 4; ++il) {
                            const int ii = hp * 4 + il;
                            asm volatile("" : "+a"(acc16[j][ii]));      // one tile at a time out of the AGPRs (see the 32x32 epilogue)
                            const f32x4 cv = acc16[j][ii];
                            u32x2v pk;
                            pk[0] = pack_bf16x2(cv[0], cv[1]); pk[1] = pack_bf16x2(cv[2], cv[3]);
                            *reinterpret_cast<u32x2v*>(pw + 32 * il) = pk;
                            if constexpr (BN && !(BIG_EXP & 16)) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { sv[il * 4 + e] += cv[e]; qv[il * 4 + e] = __builtin_fmaf(cv[e], cv[e], qv[il * 4 + e]); }
                            }
                        }
                        asm volatile("" ::: "memory");
                        if (j > 0) flush16(jh ^ 1);
#pragma unroll
                        for (int k = 0; k < 2; ++k) rows[k] = pr[k * 8 * 9];
                        rbase = validv[tl] ? ((unsigned)(cur.ibase[tl] / (KCH * 2)) * NCH + cur.n0 + hp * 64) * 2u + pcol16 : BIG_OOB;
                        asm volatile("" ::: "memory");
                    }
                if constexpr (BN && !(BIG_EXP & 32)) {
                    const float S = row_colsum16(sv), Q = row_colsum16(qv);
                    const int ch = hp * 64 + 16 * (cz >> 2) + 4 * qz + (cz & 3);
                    red[(0 * 4 + wave) * NT + ch] = S; red[(1 * 4 + wave) * NT + ch] = Q;
                }
                flush16(1);
            }
        } else {
        if constexpr (BN) {
            // ragged end only (wave-uniform, rare): the accumulators of tiles that do not exist hold the bias — zeroed, so that the sums below need no mask
            if (!allv) {
#pragma unroll
                for (int tl = 0; tl < MT; ++tl)
                    if (!validv[tl]) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                            for (int v = 0; v < 16; ++v) acc[tl][nb][v] = 0.f;
                    }
            }
        }
#pragma unroll
        for (int hp = 0; hp < NH; ++hp) {
            [[maybe_unused]] f32x2 sv[HB][8], qv[HB][8];       // packed pairs: v_pk_add_f32 / v_pk_fma_f32 — two channels per instruction
            if constexpr (BN) {
#pragma unroll
                for (int j = 0; j < HB; ++j)
#pragma unroll
                    for (int v = 0; v < 8; ++v) { sv[j][v] = f32x2{0.f, 0.f}; qv[j][v] = f32x2{0.f, 0.f}; }
            }
            bf16x8 rows[PIT];                                  // the previous tile's rows, stored one tile later: the patch reads travel under the next tile's VALU work
            unsigned rbase = BIG_OOB;
            auto flush_rows = [&]() {
#pragma unroll
                for (int k = 0; k < PIT; ++k)
                    if (!(BIG_EXP & 64)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, rows[k]), rs_out, rbase == BIG_OOB ? BIG_OOB : rbase + orow[k], 0, BIG_ST_AUX);
            };
#pragma unroll
            for (int tl = 0; tl < MT; ++tl) {
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    const int nb = hp * HB + j;
                    // one accumulator tile at a time: the opaque statement pins the tile in its AGPRs until here — without it the compiler copies
                    // all 256 accumulators out in front of the first store and spills what lives across the epilogue to make room
                    asm volatile("" : "+a"(acc[tl][nb]));
                    const f32x16 c = acc[tl][nb];
                    bf16x8 u[2];
                    cm_pack_units(c, u);
                    patch_w[4 * j] = u[0];
                    patch_w[4 * j + 2] = u[1];
                    if constexpr (BN && !(BIG_EXP & 16)) {
#pragma unroll
                        for (int v = 0; v < 8; ++v) {
                            const f32x2 x = f32x2{c[2 * v], c[2 * v + 1]};
                            sv[j][v] += x;
                            qv[j][v] = __builtin_elementwise_fma(x, x, qv[j][v]);
                        }
                    }
                }
                asm volatile("" ::: "memory");
                if (tl > 0) flush_rows();
#pragma unroll
                for (int k = 0; k < PIT; ++k) rows[k] = patch_r[k * PPI * PRS];
                rbase = validv[tl] ? ((unsigned)(cur.ibase[tl] / (KCH * 2)) * NCH + cur.n0 + hp * HB * 32) * 2u + pcol : BIG_OOB;
                asm volatile("" ::: "memory");
            }
            if constexpr (BN && !(BIG_EXP & 32)) {
#pragma unroll
                for (int j = 0; j < HB; ++j) {
                    const int nb = hp * HB + j;
                    const int e16 = li >> 1, chE = (e16 & 3) + 8 * (e16 >> 2) + 4 * lh;   // the element half_wave_colsum16 leaves in this lane
                    float s1[16], q1[16];
#pragma unroll
                    for (int v = 0; v < 8; ++v) { s1[2 * v] = sv[j][v][0]; s1[2 * v + 1] = sv[j][v][1]; q1[2 * v] = qv[j][v][0]; q1[2 * v + 1] = qv[j][v][1]; }
                    const float S = half_wave_colsum16(s1), Q = half_wave_colsum16(q1);
                    if ((lane & 1) == 0) { red[(0 * 4 + wave) * NT + nb * 32 + chE] = S; red[(1 * 4 + wave) * NT + nb * 32 + chE] = Q; }
                }
            }
            flush_rows();                                      // the pair's last tile (behind the column sums: its patch reads have long landed)
        }
        }
        if constexpr (BN) {                                    // the four waves' rows meet behind the NEXT barrier the workgroup passes anyway (bn_combine)
            pd_cnt = 0;
#pragma unroll
            for (int tl = 0; tl < MT; ++tl) { int ni = a.B - cur.img0[tl]; ni = ni < 0 ? 0 : (ni > T::IMGS ? T::IMGS : ni); pd_cnt += ni * T::TH * T::TW; }
            pd_grp = cur.grp; pd_n0 = cur.n0; have_pd = true;
        }
        BT(b2);
#ifdef BIG_TIMING
        if (BT_ON) { tepi += b2 - b1; ++nit; }
#endif
#ifdef BIG_STEPTIME
        steptime_first = false;
#endif
        if (!have_next) break;
        cur = setup(grpn, n0n); it = itn;
        set_voff(cur);                                         // recomputed (not kept) across the epilogue: ~100 VALU per item for 12-16 registers
        if constexpr (!TDB) ldf(0, 0, 0, gst & 1);             // first fragments of the next item (its tiles and slab 0 are in LDS: the last stage's barriers);
                                                               // TDB: the last stage's last step has requested them already (tile buffer 0, kernel row 0)
    }
    if constexpr (BN) { __syncthreads(); bn_combine(); }       // the last item's rows
#ifdef BIG_TIMING
    if (BT_ON && (blockIdx.x & 15) == 0 && blockIdx.x < 256 && tid == 0) {
        __builtin_amdgcn_sched_barrier(0);
        const long long e = clock64();
        long long* o = big_dbg + (blockIdx.x >> 4) * 12;
        o[0] = nit; o[1] = tpro; o[2] = tstage; o[3] = tclose; o[4] = tepi; o[5] = e - bt_entry;
        o[6] = (long long)wall_clock64() - brt0; o[7] = NST; o[8] = brt0; o[9] = (long long)wall_clock64();
    }
#endif
}

template <int KCH, int NCH, int H, int NT, int MT, int KB, int EPI, bool IMGL = false, bool S16 = false, bool ALLC = false>
static int run_big(const ConvBf16Args& a, hipStream_t st) {
    // (8 x 1 wave tile, E2's input gradient: 8 accumulator tiles = 256 registers would let TWO workgroups share a CU with single-buffered tiles — measured:
    //  194.0 us against 192.1 for one workgroup with double-buffered tiles, step level; profiles/r05_r_e2_on_big_kernel.txt)
    constexpr bool TDB = (MT == 4 || IMGL) && KB == 1;         // the 4 x 4 tile / the image layout leave LDS for a second set of input tiles
    using T = Tile<H>;
    constexpr int OCT = 2 * KB;
    constexpr int NY = NCH / NT;
    constexpr int NIMG = MT * 128 / (H * H), PAD = Bf16Geom<H, OCT>::PAD;
    constexpr int PLPX = (IMGL && T::TW != H) ? (MT * 128 / H + 4) * (H + 4) : NIMG * (H + 4) * (H + 4);      // strip layout: SR + 4 rows of H + 4
    constexpr int TILE_UNITS = IMGL ? OCT * (((PLPX + 15 - PAD) / 16) * 16 + PAD) : MT * OCT * Bf16Geom<H, OCT>::PSP;
    constexpr int SMEM = ((TDB ? 2 : 1) * TILE_UNITS + (S16 ? 3 : 2) * 5 * KB * 2 * NT + 256 + (S16 ? 4 * 16 * 9 : 4 * 32 * 9)) * 16 + (EPI == EPI_BIAS_BNSTAT ? (2 * 4 * NT + NCH) * 4 : 0);
    static_assert(SMEM <= 160 * 1024, "LDS");
    // 32-bit byte offsets and buffer descriptors inside: larger tensors take the per-tile kernels (size_t addressing)
    if ((size_t)a.B * H * H * KCH * 2 >= (1ull << 31) || (size_t)a.B * H * H * NCH * 2 >= (1ull << 31)) return -100;
    if (g_conv_dry) return 0;
    auto kern = conv5x5_bf16_big_kernel<KCH, NCH, H, NT, MT, KB, EPI, TDB, IMGL, S16, ALLC>;
    static DeviceOnce once;
    { int rc = cvae_grant_lds(once, reinterpret_cast<const void*>(kern), SMEM); if (rc) return rc; }
    const int numTiles = cdiv(a.B, T::IMGS) * T::TILES_PER_IMG, numGroups = cdiv(numTiles, MT);
    const int numItems = 8 * cdiv(numGroups, 8) * NY;
    // CVAE_BIG_MAXWG (tests): cap the persistent grid so that small batches walk several items per workgroup
    static const int maxwg = [] { const char* e = getenv("CVAE_BIG_MAXWG"); return e ? atoi(e) : 0; }();
    int G = cvae_num_cus();
    if (maxwg > 0 && G > maxwg) G = maxwg;
    G -= G % 8;
    if (G < 8) G = 8;
    if (G > numItems) G = numItems;
    cvae_probe_begin(st);
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), SMEM, st, a, numGroups);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    return 0;
}

// Which layers this file serves, by mask bit: input gradients — bit 0 = E4 (256 -> 128, 4 x 4 wave tile), bit 1 = E3 (128 -> 64, 8 x 2), bit 2 = E2 at 64 x 64 (64 -> 32:
// an image-high item on an 8 x 1 wave tile, 9 fragment reads per 8 MFMAs where the per-tile kernel reads 12: 192 vs 209 us); forward (bias + ONE BatchNorm partial per
// item of conv_bf16_big_tiles(..) tiles — the kernel that ran tells launch_bn_fwd_finalize) — bit 0 = E3 (64 -> 128), bit 1 = E4 (128 -> 256), both 4 x 4, bit 2 = E2 at
// 64 x 64 (32 -> 64 as two 32-channel halves of an image-high item, 8 x 1: 196 vs 209 us on the two-workgroup persistent kernel; as ONE 8 x 2 item it lost, 244 us:
// its 16-tile BatchNorm epilogue was half of a 10-stage item, profiles/r05_k_big_image_layout.txt).
// CVAE_BIG_S16 (experiment): the instantiations built on v_mfma_f32_16x16x32_bf16 (S16)
static bool big_s16() { static const bool on = [] { const char* e = getenv("CVAE_BIG_S16"); return e && atoi(e) != 0; }(); return on; }
bool conv_bf16_big_has(int layer, int width, bool dgrad, int mask) {
    if (width != 64 && width != 128) return false;
    if (dgrad) return (layer == 3 && (mask & 1)) || (layer == 2 && (mask & 2)) || (layer == 1 && (mask & 4));
    return (layer == 2 && (mask & 1)) || (layer == 3 && (mask & 2)) || (layer == 1 && (mask & 4));
}
int conv_bf16_big_tiles(int layer, int width, bool dgrad) { (void)width; return ((dgrad && layer == 2) || (!dgrad && layer == 1)) ? 8 : 4; }
// returns -100 when the layer has no instantiation (or the tensors are too large for its 32-bit offsets)
int launch_conv_bf16_big(int layer, int width, bool dgrad, int mask, const ConvBf16Args& a, hipStream_t st) {
    if (!conv_bf16_big_has(layer, width, dgrad, mask)) return -100;
    if (dgrad) {
        if (width == 64 && layer == 3 && big_s16()) return run_big<256, 128, 8, 128, 4, 1, EPI_PLAIN, true, true>(a, st);
        if (width == 64 && layer == 3) return run_big<256, 128, 8, 128, 4, 1, EPI_PLAIN, BIG_IMGL, false, BIG_ALLC4>(a, st);
        if (width == 64 && layer == 2) return run_big<128, 64, 16, 64, 8, 1, EPI_PLAIN, BIG_IMGL, false, BIG_ALLC3>(a, st);
        if (width == 64 && layer == 1) return run_big<64, 32, 32, 32, 8, 1, EPI_PLAIN, true, false, BIG_ALLC>(a, st);      // E2: image-high item, 8 x 1 wave tile, whole lines fetched once
        if (width == 128 && layer == 3) return run_big<256, 128, 16, 128, 4, 1, EPI_PLAIN, BIG_IMGL>(a, st);
        if (width == 128 && layer == 2) return run_big<128, 64, 32, 64, 8, 1, EPI_PLAIN, BIG_IMGL>(a, st);
        if (width == 128 && layer == 1) return run_big<64, 32, 64, 32, 8, 1, EPI_PLAIN, true>(a, st);       // 16-row strips of the 64-row image, 8 x 1
    } else {
        if (width == 64 && layer == 1) return run_big<32, 64, 32, 32, 8, 1, EPI_BIAS_BNSTAT, true>(a, st);      // 8 x 1 (two channel halves per image); 8 x 2: 244 us
        if (width == 64 && layer == 2) return run_big<64, 128, 16, 128, 4, 1, EPI_BIAS_BNSTAT, BIG_IMGL>(a, st);
        if (width == 64 && layer == 3 && big_s16()) return run_big<128, 256, 8, 128, 4, 1, EPI_BIAS_BNSTAT, true, true>(a, st);
        if (width == 64 && layer == 3) return run_big<128, 256, 8, 128, 4, 1, EPI_BIAS_BNSTAT, BIG_IMGL>(a, st);
        if (width == 128 && layer == 2) return run_big<64, 128, 32, 128, 4, 1, EPI_BIAS_BNSTAT>(a, st);       // half an image per item: per-tile layout
        if (width == 128 && layer == 3) return run_big<128, 256, 16, 128, 4, 1, EPI_BIAS_BNSTAT, BIG_IMGL>(a, st);
        if (width == 128 && layer == 1) return run_big<32, 64, 64, 32, 8, 1, EPI_BIAS_BNSTAT, true>(a, st);
    }
    return -100;
}
