// conv_bf16_ps.hip — persistent bf16-MFMA 5x5 conv forward / input-gradient kernel (precision mode 1, round 4).
//
// Same call sites as conv5x5_bf16_kernel<.., MODE_STD, NS = 1> (nn.Conv2d E2..E4, vae_nets.py:74,79,84, and their input
// gradients in loss.backward(), vae.py:57), same tiling (two 128-pixel tiles x NT channels per workgroup-step, 32-channel K
// chunks, one kernel row per stage), same packed weights, same LDS images.  What changes is everything AROUND the MFMA loop —
// stage timing of the round-3 kernel (profiles/r04_stage_timing.txt): of a 25 k-cycle E2-forward workgroup 7.5 k are MFMAs,
// 3 k wait for its (cold) input tiles, 8-10 k are the epilogue at ~15 cycles per VALU instruction beside the co-resident
// workgroup's MFMAs:
//   * PERSISTENT workgroups (two per CU) loop over (tile pair, channel block) items; the input tiles of the NEXT item are
//     requested during the last chunk of the current one, the weight slab of its first stage during the last stage: no
//     cold prologue after the first item;
//   * the epilogue is reduced to what must happen before the accumulators are reused — bias is the accumulators' INITIAL value
//     (no add), BatchNorm partial sums in the lane (pixel-major: a lane holds 16 pixels of one channel), 32 v_cvt_pk, the
//     tile parked as bf16 rows [channel][32 pixels] in an LDS patch (tile 0) / in 16 registers (tile 1) — about 200 VALU
//     instructions instead of ~600;
//   * the global stores are DEFERRED: during the first five stages of the next item every wave drains its part of the patch
//     with transposing reads (ds_read_b64_tr_b16: lane = pixel, 4 consecutive channels per read -> two reads = one 16-byte
//     NHWC unit) — a few instructions per staging slot, under the other workgroup's MFMA phase.  vmcnt retires loads and
//     stores in issue order, so inside a slot the next stage's loads are issued BEFORE the drain's stores, and the stores are
//     buffer stores whose invalid lanes carry an out-of-range offset: no branch around a store, every wait stays counted
//     (profiles/experiments/vm_after_store.py).
// BatchNorm partials: per tile and channel (sum, M2 about the tile mean) as bn_fwd_finalize merges them, M2 = Q - S*S/n in
// double from the per-wave fp32 sums S, Q of the biased fp32 accumulators (same inputs as round 3, different summation order).
#include "common.h"
#include "conv_epilogue.h"
#include "conv_bf16.h"

// -DPS_TIMING (timing builds): cycles per phase of ONE instantiation (-DPS_T_KCH/NCH/H), summed over the items of the sampled
// workgroups (blockIdx.x a multiple of 32), wave 0 lane 0: [items, barrier 1, staging, drain, barrier 2, MFMA, epilogue, whole loop]
#ifdef PS_TIMING
__device__ long long ps_dbg[16 * 12];
extern "C" int cvae_ps_dbg_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ps_dbg), sizeof(ps_dbg)); }
#define PT_ON (KCH == PS_T_KCH && NCH == PS_T_NCH && H == PS_T_H)
#define PT(v) do { if (PT_ON) { __builtin_amdgcn_sched_barrier(0); v = clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define PT_ON false
#define PT(v)
#endif
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

// raw buffer descriptor over a whole tensor (gfx9 word 3: DATA_FORMAT_32): a lane whose byte offset is >= bytes is dropped by
// the bounds check, so "store if valid" needs no branch
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
static constexpr unsigned OOB = 0x80000000u;

template <int KCH, int NCH, int H, int NT, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv5x5_bf16_ps_kernel(ConvBf16Args a, int numPairs, int numTiles) {
    using T = Tile<H>;
    constexpr int MT = 2, KS = 5, KCB = 32, KB = KCB / 16, OCT = KCB / 8, NB = NT / 32, NY = NCH / NT, NCHUNK = KCH / KCB;
    static_assert(KCH % KCB == 0 && NCH % NT == 0 && (NT == 32 || NT == 64), "channel tiling");
    constexpr int PSP = Bf16Geom<H, OCT>::PSP;
    constexpr int A_UNITS = OCT * PSP, W_UNITS = KS * KB * 2 * NT;
    constexpr bool HAS_BIAS = EPI != EPI_PLAIN, BNSTAT = EPI == EPI_BIAS_BNSTAT, RELU = EPI == EPI_BIAS_RELU;
    constexpr int PRS = 36;                                    // patch row: 32 pixels + 4 (72 bytes: conflict-free 8-byte row writes)
    constexpr int NU = NT / 16;                                // 16-byte units a lane drains per tile (its wave's 32 pixels x NT channels)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16x8* lds_a = reinterpret_cast<bf16x8*>(smem_raw);       // [tile][octet][halo pixel]
    bf16x8* lds_w = lds_a + MT * A_UNITS;                      // [tap][kb][half][n]
    float* lds_bias = reinterpret_cast<float*>(lds_w + W_UNITS);           // [NCH]
    float* red = lds_bias + NCH;                               // [tile][S | Q][wave][NT]
    __bf16* patch = reinterpret_cast<__bf16*>(red + MT * 2 * 4 * NT);      // [wave][NT channel rows][PRS]

    [[maybe_unused]] const long long rt_entry = PT_ON ? (long long)wall_clock64() : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int G = gridDim.x;
    // item -> (tile pair, channel block).  Workgroups are dealt to the 8 XCDs round-robin (item & 7 = the XCD while G % 8 == 0): XCD x
    // owns the contiguous range of pairs [x PP, (x+1) PP) — neighbouring pairs share halo rows in ONE L2 — and walks it pair by
    // pair, the NY channel blocks of a pair back to back (the input tile is re-read from L2, not from HBM).  A first version dealt
    // consecutive pairs to consecutive XCDs: 470 MB fetched for the 403 MB of E2 forward (profiles/r04_a), the E1 mistake again.
    const int PP = cdiv(numPairs, 8), numItems = 8 * PP * NY;
    auto decode = [&](int it, int& pair, int& n0) { const int x = it & 7, j = it >> 3, jp = j / NY; n0 = (j - jp * NY) * NT; pair = jp < PP ? x * PP + jp : numPairs; };
    int it = blockIdx.x, pair, n0;
    decode(it, pair, n0);
    if (it >= numItems || pair >= numPairs) return;            // whole workgroup; (G % 8 == 0: once past the end, always past the end)

    // ---- per-thread tables that do not depend on the item ----
    const int m = wave * 32 + lane_pix<H, true>(li);           // pixel (of the 128 of a tile) behind MFMA row li of this wave
    const int aPix = (m / (T::TH * T::TW)) * T::HPI + ((m % (T::TH * T::TW)) / T::TW) * T::HTW + (m % T::TW);
    constexpr int WPT = (W_UNITS + 255) / 256;
    unsigned wbase[WPT];                                       // byte offset inside a stage's slab of the packed weights, without n0: uniform base + 32-bit lane offset = one SGPR pair for all loads
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int q = tid + i * 256, n = q % NT, row = q / NT, half = row & 1, kb = (row >> 1) % KB, s = row / (2 * KB);
        wbase[i] = (W_UNITS % 256 == 0 || q < W_UNITS) ? (((s * (KCH / 16) + kb) * 2 + half) * NCH + n) * 16 : 0;     // BYTES; past the slab: re-read unit 0 (never staged)
    }
    constexpr int NQ = T::HP * OCT, IPT = (NQ + 255) / 256;
    int irel[IPT];                                             // byte offset of the unit relative to the tile's first pixel, chunk 0
    unsigned ipk[IPT];                                         // halo row | halo column << 8 | image << 16 | unit exists << 31
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int q = tid + i * 256, oct = q % OCT, hp = q / OCT, img = hp / T::HPI, rem = hp - img * T::HPI;
        const int hy = rem / T::HTW, hx = rem % T::HTW;
        irel[i] = (((img * H + hy - 2) * H + hx - 2) * KCH + oct * 8) * 2;      // BYTES
        ipk[i] = (unsigned)hy | ((unsigned)hx << 8) | ((unsigned)img << 16) | ((NQ % 256 == 0 || q < NQ) ? 0x80000000u : 0u);
    }
    // drain: this lane's pixel column r of its wave's block, and where that pixel sits relative to the tile's first pixel
    const int grp = lane >> 4, dcol = 16 * (grp & 1) + (lane & 15);
    const int dm = wave * 32 + lane_pix<H, true>(dcol), dimg = dm / (T::TH * T::TW), drem = dm % (T::TH * T::TW);
    const int drel = (dimg * H + drem / T::TW) * H + drem % T::TW;                     // pixels
    const __bf16* dsrc = patch + (size_t)(wave * NT + ((lane & 15) >> 2)) * PRS + 16 * (grp & 1) + 4 * (lane & 3);   // + (8 o [+ 4]) * PRS
    // epilogue: rows (channel li of block nb) of this wave in the patch, columns 8 g + 4 lh
    __bf16* pdst = patch + (size_t)(wave * NT + li) * PRS + 4 * lh;

    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(a.out, (unsigned)((size_t)a.B * H * H * NCH * 2));
    const __amdgpu_buffer_rsrc_t rs_bn = make_rsrc(a.bnpart, BNSTAT ? (unsigned)((size_t)2 * numTiles * NCH * 4) : 0u);

    // ---- item state ----
    struct Item { int n0, img0[MT], ty0[MT], tx0[MT], ibase[MT], obase[MT], mt0; };
    auto setup = [&](int pr, int nn0) {
        Item x; x.n0 = nn0; x.mt0 = pr * MT;
#pragma unroll
        for (int tl = 0; tl < MT; ++tl) {
            const int mt = pr * MT + tl, tin = mt % T::TILES_PER_IMG;
            x.img0[tl] = (mt / T::TILES_PER_IMG) * T::IMGS;
            x.ty0[tl] = (tin / T::TILES_X) * T::TH; x.tx0[tl] = (tin % T::TILES_X) * T::TW;
            x.ibase[tl] = ((x.img0[tl] * H + x.ty0[tl]) * H + x.tx0[tl]) * KCH * 2;    // bytes
            x.obase[tl] = (x.img0[tl] * H + x.ty0[tl]) * H + x.tx0[tl];               // pixels
        }
        return x;
    };
    bf16x8 wreg[WPT], breg[MT * IPT];
    unsigned okmask = 0u;
    // Operands travel as BUFFER loads (uniform descriptor + 32-bit lane offset + uniform SGPR offset).  Round 4 used a laundered uniform pointer
    // + lane offset, which hipcc emits as FLAT loads: they count in lgkmcnt too and return out of order with the LDS reads, so every fragment
    // wait of the MFMA phase became lgkmcnt(0) and its first MFMA waited for the next stage's slab and tiles — the loads the phase was meant
    // to hide (round 5: 1052 flat_load in this file's listing; conv_bf16_big.hip header).
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.wp, (unsigned)(25u * (KCH / 16) * 2u * NCH * 16u));
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in, (unsigned)((size_t)a.B * H * H * KCH * 2));
    auto load_w = [&](const Item& x, int cc, int r) {
        const unsigned soff = (unsigned)((r * KS * (KCH / 16) + cc * KB) * 2 * NCH + x.n0) * 16u;
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wbase[i], soff, 0));
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            if (W_UNITS % 256 == 0 || tid + i * 256 < W_UNITS) lds_w[tid + i * 256] = wreg[i];
    };
    auto load_input = [&](const Item& x, int cc) {
#pragma unroll
        for (int tl = 0; tl < MT; ++tl)
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int hy = ipk[i] & 255, hx = (ipk[i] >> 8) & 255, img = (ipk[i] >> 16) & 255;
                // branch-free: a short-circuit here becomes a branch per unit around the load, and the waits stop being counted
                const unsigned okb = (ipk[i] >> 31) & (unsigned)((unsigned)(x.ty0[tl] + hy - 2) < (unsigned)H) &
                                     (unsigned)((unsigned)(x.tx0[tl] + hx - 2) < (unsigned)H) & (unsigned)(x.img0[tl] + img < a.B);
                const unsigned e = okb ? (unsigned)(x.ibase[tl] + irel[i] + cc * KCB * 2) : 0u;      // bytes from a.in (uniform base + 32-bit lane offset)
                breg[tl * IPT + i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_in, e, 0, 0));    // unconditional load (unit 0 when padding), select when staged
                okmask = (okmask & ~(1u << (tl * IPT + i))) | (okb << (tl * IPT + i));
            }
    };
    auto store_input = [&]() {
        bf16x8 z;
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = (__bf16)0.f;
#pragma unroll
        for (int tl = 0; tl < MT; ++tl)
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                const int q = tid + i * 256;
                if (!(NQ % 256 == 0 || q < NQ)) continue;
                lds_a[tl * A_UNITS + (q % OCT) * PSP + q / OCT] = ((okmask >> (tl * IPT + i)) & 1u) ? breg[tl * IPT + i] : z;
            }
    };

    f32x16 acc[MT][NB];
    unsigned p1[NB][8];                                        // tile 1 of the finished item, packed bf16 pixel pairs, until the patch is free
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int k = 0; k < 8; ++k) p1[nb][k] = 0u;

    // ---- drain of the finished item `pd` (valid only if have_pd): tile `tl`, units u0, u0 + 1 ----
    auto drain_units = [&](const Item& pd, bool have_pd, int tl, int u0) {
        const bool ok = have_pd && pd.img0[tl] + dimg < a.B;
        const unsigned pix = (unsigned)(pd.obase[tl] + drel);
#pragma unroll
        for (int u = u0; u < u0 + 2 && u < NU; ++u) {
            const int o = 2 * u + (grp >> 1);                  // channel octet of the tile's NT channels
            const __bf16* s0 = dsrc + (size_t)(8 * o) * PRS;
            const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)s0);
            const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(s0 + 4 * PRS));
            const u32x2v l2 = __builtin_bit_cast(u32x2v, lo), h2 = __builtin_bit_cast(u32x2v, hi);
            const unsigned off = ok ? (pix * NCH + pd.n0 + 8 * o) * 2u : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4v{l2[0], l2[1], h2[0], h2[1]}, rs_out, off, 0, 0);
        }
    };
    auto park_tile1 = [&]() {                                  // tile 1's packed rows -> patch (tile 0's drain is complete)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<u32x2v*>(pdst + (size_t)(nb * 32) * PRS + 8 * g) = u32x2v{p1[nb][2 * g], p1[nb][2 * g + 1]};
    };
    auto bn_combine = [&](const Item& pd, bool have_pd) {      // the four waves' rows of the finished item -> (sum, M2) per tile and channel
        if constexpr (BNSTAT) {
            const int tl = tid / NT, cc = tid % NT, mt = pd.mt0 + (tl < MT ? tl : 0);
            float S = 0.f, Q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { S += red[(((tl & 1) * 2 + 0) * 4 + w) * NT + cc]; Q += red[(((tl & 1) * 2 + 1) * 4 + w) * NT + cc]; }
            int nvalid_img = a.B - pd.img0[tl & 1];
            if (nvalid_img > T::IMGS) nvalid_img = T::IMGS;
            if (nvalid_img < 1) nvalid_img = 1;
            const double cnt = (double)(nvalid_img * T::TH * T::TW);
            const double m2 = (double)Q - (double)S * (double)S / cnt;
            const bool ok = have_pd && tid < MT * NT && mt < numTiles;
            const unsigned o = (unsigned)(mt * NCH + pd.n0 + cc) * 4u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, S), rs_bn, ok ? o : OOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(m2 > 0.0 ? m2 : 0.0)), rs_bn, ok ? o + (unsigned)(numTiles * NCH) * 4u : OOB, 0, 0);
        }
    };

    // ---- MFMA phase of stage (kernel row r): pinned software pipeline (see conv5x5_bf16_kernel) ----
    auto mfma_phase = [&](int r) {
        const bf16x8* ap = lds_a + lh * PSP + aPix + r * T::HTW;
        const bf16x8* bp = lds_w + lh * NT + li;
        constexpr int NSTEP = KS * KB, PF = 1, NSET = 2;      // one step ahead: the registers go to the deferred tile and the input prefetch
        bf16x8 wf[NSET][NB], xf[NSET][MT];
        auto ld = [&](int i, int b) {
            const int s = i / KB, kb = i % KB;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wf[b][nb] = bp[((s * KB + kb) * 2) * NT + nb * 32];
#pragma unroll
            for (int tl = 0; tl < MT; ++tl) xf[b][tl] = ap[tl * A_UNITS + (kb * 2) * PSP + s];
        };
#pragma unroll
        for (int i = 0; i < PF; ++i) ld(i, i % NSET);
        __builtin_amdgcn_sched_group_barrier(0x100, PF * (NB + MT), 0);
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            if (i + PF < NSTEP) ld(i + PF, (i + PF) % NSET);
#pragma unroll
            for (int tl = 0; tl < MT; ++tl)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)      // pixels as the A operand: D[pixel][channel] — a lane holds 16 pixels of channel li
                    acc[tl][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[i % NSET][tl], wf[i % NSET][nb], acc[tl][nb], 0, 0, 0);
            constexpr int NM = MT * NB, NR = NB + MT;
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i + PF < NSTEP && k < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if constexpr (NR > NM) { if (i + PF < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, NR - NM, 0); }
        }
    };

    // ---- prologue ----
    if constexpr (HAS_BIAS) { for (int c = tid; c < NCH; c += 256) lds_bias[c] = a.bias[c]; }
    Item cur = setup(pair, n0), pd = cur;
    bool have_pd = false;
    load_w(cur, 0, 0);
    load_input(cur, 0);
    __syncthreads();                                           // bias row visible

    [[maybe_unused]] long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, tb1 = 0, tst = 0, tdr = 0, tb2 = 0, tmf = 0, tep = 0, nit = 0, tall = 0;
    PT(tall);
    [[maybe_unused]] const long long rt0 = PT_ON ? (long long)wall_clock64() : 0;
#ifndef PS_PRIO
#define PS_PRIO 1
#endif
    int itemNo = 0;
    for (;;) {
        // The two workgroups of a CU share each SIMD's issue ports, arbitrated by priority, then AGE: left alone, the workgroup that
        // was dispatched second loses every tie and finishes its 16 items 50 us after the first (173 vs 225 us, ps_timing.py), running
        // the tail alone.  Alternating the priority per item (the halves in opposite phase) shares the ports evenly.
        if (PS_PRIO) { if (((itemNo++ ^ (blockIdx.x >= (unsigned)(G / 2))) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        int itn = it + G, pairn, n0n;
        decode(itn, pairn, n0n);
        const bool have_next = itn < numItems && pairn < numPairs;
        const Item nxt = have_next ? setup(pairn, n0n) : cur;  // no next item: the prefetches re-read this one (valid addresses, values unused)
#pragma unroll
        for (int tl = 0; tl < MT; ++tl)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float b0 = HAS_BIAS ? lds_bias[cur.n0 + nb * 32 + li] : 0.f;
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[tl][nb][v] = b0;
            }
        // chunk 0: five stages, each with its share of the previous item's drain
#pragma unroll
        for (int r = 0; r < KS; ++r) {
            PT(c0);
            __syncthreads();                                   // everyone finished reading the previous stage (and its drain reads)
            PT(c1);
            if (r == 0) store_input();
            store_w();
            // the loads of the coming stage / chunk are issued BEFORE this slot's stores: waiting for them later does not wait for the stores
            if (r == 0) { if (NCHUNK > 1) load_input(cur, 1); else load_input(nxt, 0); }
            if (r < KS - 1) load_w(cur, 0, r + 1); else if (NCHUNK > 1) load_w(cur, 1, 0); else load_w(nxt, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                 // pinned: the scheduler otherwise sinks the loads below the drain's stores
            PT(c2);
            if (r == 0) { bn_combine(pd, have_pd); drain_units(pd, have_pd, 0, 0); }
            if (r == 1) drain_units(pd, have_pd, 0, 2);
            if (r == 2) park_tile1();
            if (r == 3) drain_units(pd, have_pd, 1, 0);
            if (r == 4) drain_units(pd, have_pd, 1, 2);
            PT(c3);
            __syncthreads();
            PT(c4);
            mfma_phase(r);
            PT(c5);
#ifdef PS_TIMING
            if (PT_ON) { tb1 += c1 - c0; tst += c2 - c1; tdr += c3 - c2; tb2 += c4 - c3; tmf += c5 - c4; }
#endif
        }
        for (int cc = 1; cc < NCHUNK; ++cc) {
            const bool lastc = cc == NCHUNK - 1;
#pragma unroll
            for (int r = 0; r < KS; ++r) {
                PT(c0);
                __syncthreads();
                PT(c1);
                if (r == 0) store_input();
                store_w();
                if (r == 0) { if (lastc) load_input(nxt, 0); else load_input(cur, cc + 1); }
                if (r < KS - 1) load_w(cur, cc, r + 1); else if (lastc) load_w(nxt, 0, 0); else load_w(cur, cc + 1, 0);
                PT(c2);
                __syncthreads();
                PT(c4);
                mfma_phase(r);
                PT(c5);
#ifdef PS_TIMING
                if (PT_ON) { tb1 += c1 - c0; tst += c2 - c1; tb2 += c4 - c2; tmf += c5 - c4; }
#endif
            }
        }
        // ---- epilogue: what must happen before the accumulators are reused ----
        PT(c0);
#pragma unroll
        for (int tl = 0; tl < MT; ++tl)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x16& c = acc[tl][nb];
                if constexpr (RELU) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) c[v] = fmaxf(c[v], 0.f);
                }
                if constexpr (BNSTAT) {        // this lane: channel li of block nb, its 16 pixels (rows (v&3) + 8 (v>>2) + 4 lh of the wave's 32)
                    float s = 0.f, q = 0.f;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int mm = wave * 32 + lane_pix<H, true>((v & 3) + 8 * (v >> 2) + 4 * lh);
                        const float y = (T::IMGS == 1 || cur.img0[tl] + mm / (T::TH * T::TW) < a.B) ? c[v] : 0.f;
                        s += y; q += y * y;
                    }
                    if (T::IMGS == 1 && cur.img0[tl] >= a.B) { s = 0.f; q = 0.f; }
                    s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
                    if (lh == 0) { red[((tl * 2 + 0) * 4 + wave) * NT + nb * 32 + li] = s; red[((tl * 2 + 1) * 4 + wave) * NT + nb * 32 + li] = q; }
                }
                // bf16, four consecutive MFMA rows (pixels) per 8-byte unit: columns 8 g + 4 lh .. +3 of row (channel) li
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned w0 = pack_bf16x2(c[4 * g], c[4 * g + 1]), w1 = pack_bf16x2(c[4 * g + 2], c[4 * g + 3]);
                    if (tl == 0) *reinterpret_cast<u32x2v*>(pdst + (size_t)(nb * 32) * PRS + 8 * g) = u32x2v{w0, w1};
                    else { p1[nb][2 * g] = w0; p1[nb][2 * g + 1] = w1; }
                }
            }
        PT(c1);
#ifdef PS_TIMING
        if (PT_ON) { tep += c1 - c0; ++nit; }
#endif
        pd = cur; have_pd = true;
        if (!have_next) break;
        cur = nxt; it = itn;
    }
#ifdef PS_TIMING
    if (PT_ON && (blockIdx.x & 31) == 0 && blockIdx.x < 512 && tid == 0) {
        long long* o = ps_dbg + (blockIdx.x >> 5) * 12;
        o[10] = rt_entry;
        o[0] = nit; o[1] = tb1; o[2] = tst; o[3] = tdr; o[4] = tb2; o[5] = tmf; o[6] = tep; o[7] = clock64() - tall;
        o[9] = rt0;                                       // absolute start (100 MHz ticks): do the workgroups 256..511 start with 0..255?
        o[8] = (long long)wall_clock64() - rt0;          // s_memrealtime ticks (100 MHz) over the same span: in-kernel clock = o[7] / o[8] * 100 MHz
    }
#endif
    // ---- tail: the last item's tiles ----
    __syncthreads();
    bn_combine(pd, true);
    drain_units(pd, true, 0, 0); drain_units(pd, true, 0, 2);
    __syncthreads();
    park_tile1();
    __syncthreads();
    drain_units(pd, true, 1, 0); drain_units(pd, true, 1, 2);
#ifdef PS_TIMING
    if (PT_ON && (blockIdx.x & 31) == 0 && blockIdx.x < 512 && tid == 0) ps_dbg[(blockIdx.x >> 5) * 12 + 11] = (long long)wall_clock64();
#endif
}

template <int KCH, int NCH, int H, int NT, int EPI>
static int run_ps(const ConvBf16Args& a, hipStream_t st) {
    using T = Tile<H>;
    constexpr int NY = NCH / NT;
    constexpr int SMEM = (2 * 4 * Bf16Geom<H, 4>::PSP + 5 * 2 * 2 * NT) * 16 + (NCH + 2 * 2 * 4 * NT) * 4 + 4 * NT * 36 * 2;
    static_assert(SMEM <= 80 * 1024, "two workgroups per CU");
    // 32-bit byte offsets, a 32-bit num_records and 0x80000000 as the "skip this lane" offset inside: tensors of 2 GiB and more take the
    // per-tile kernel (size_t addressing) — the caller falls back on -100
    if ((size_t)a.B * H * H * KCH * 2 >= (1ull << 31) || (size_t)a.B * H * H * NCH * 2 >= (1ull << 31)) return -100;
    if (g_conv_dry) return 0;
    auto kern = conv5x5_bf16_ps_kernel<KCH, NCH, H, NT, EPI>;
    static DeviceOnce once;
    { int rc = cvae_grant_lds(once, reinterpret_cast<const void*>(kern), SMEM); if (rc) return rc; }
    const int numTiles = cdiv(a.B, T::IMGS) * T::TILES_PER_IMG, numPairs = cdiv(numTiles, 2);
    const int numItems = 8 * cdiv(numPairs, 8) * NY;
    int G = 2 * cvae_num_cus();
    G -= G % 8;
    if (G < 8) G = 8;
    if (G > numItems) G = numItems;
    cvae_probe_begin(st);
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), SMEM, st, a, numPairs, numTiles);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    return 0;
}

int launch_conv_bf16_ps(int layer, int width, bool dgrad, const ConvBf16Args& a, hipStream_t st) {
    if (!dgrad) {
        if (width == 64) {
            switch (layer) {
                case 1: return run_ps<32, 64, 32, 64, EPI_BIAS_BNSTAT>(a, st);
                case 2: return run_ps<64, 128, 16, 64, EPI_BIAS_BNSTAT>(a, st);
                case 3: return run_ps<128, 256, 8, 64, EPI_BIAS_BNSTAT>(a, st);
            }
        } else if (width == 128) {
            switch (layer) {
                case 1: return run_ps<32, 64, 64, 64, EPI_BIAS_BNSTAT>(a, st);
                case 2: return run_ps<64, 128, 32, 64, EPI_BIAS_BNSTAT>(a, st);
                case 3: return run_ps<128, 256, 16, 64, EPI_BIAS_BNSTAT>(a, st);
                case 4: return run_ps<256, 128, 8, 64, EPI_BIAS_RELU>(a, st);
            }
        }
    } else {
        if (width == 64) {
            switch (layer) {
                case 1: return run_ps<64, 32, 32, 32, EPI_PLAIN>(a, st);
                case 2: return run_ps<128, 64, 16, 64, EPI_PLAIN>(a, st);
                case 3: return run_ps<256, 128, 8, 64, EPI_PLAIN>(a, st);
            }
        } else if (width == 128) {
            switch (layer) {
                case 1: return run_ps<64, 32, 64, 32, EPI_PLAIN>(a, st);
                case 2: return run_ps<128, 64, 32, 64, EPI_PLAIN>(a, st);
                case 3: return run_ps<256, 128, 16, 64, EPI_PLAIN>(a, st);
                case 4: return run_ps<128, 256, 8, 64, EPI_PLAIN>(a, st);
            }
        }
    }
    return -100;
}
