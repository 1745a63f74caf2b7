// conv_mfma_ps.hip — persistent form of the fp32-MFMA 5x5 conv forward / input-gradient kernel (round 4).
//
// Same call sites, tiling, LDS images and k order as conv5x5_mfma_kernel (nn.Conv2d E2..E4, vae_nets.py:74,79,84, and their input
// gradients in loss.backward(), vae.py:57): 128 output pixels x NT channels per step, 16- or 32-channel K chunks, one kernel row
// per stage, v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate in k order).  What changes is what happens between the
// MFMA loops of consecutive tiles — stage timing of the per-tile kernel: of a 120.5 k-cycle E2-forward workgroup
// 8.7 k are the epilogue and ~4 k the cold prologue, during which only the co-resident workgroup feeds the matrix pipe:
//   * persistent workgroups (two per CU) walk (tile, channel block) items; the next item's first weight slab and input chunk are
//     requested during the current item's last stage / chunk;
//   * channel-major accumulators (weights as the MFMA's A operand, conv_epilogue.h): a lane holds ONE pixel and four consecutive
//     channels per register quad, so a finished tile is 4 x 16-byte stores per 32-channel block straight from registers — no
//     LDS transpose patch, no barrier.  The bias is the accumulators' INITIAL value (no add).  The finished tile's values are MOVED to
//     a second register set (32 values); its stores and its BatchNorm partials (a per-wave halving butterfly, fixed order) are
//     issued in the staging slots of the next item's first five stages, after that slot's loads (vmcnt retires loads and stores in
//     issue order; buffer stores with an out-of-range offset for invalid lanes keep every wait counted).
// Measured (MI355X, B = 256, rocprofv3): the 64-channel-tile layers gain 2-3 % (E2 forward 222.5 -> 216.7 us, E3 forward 215.2 ->
// 209.4, E3 input gradient 212.4 -> 208.0); the 32-channel-tile instantiations (KC = 32: 36 staging registers more) spill under
// the 256-register budget and lose (E2 input gradient 222 -> 262 us) — they stay on the per-tile kernel (conv_mfma.hip, CONVF_PS_DEFAULT).
// BatchNorm partials: per tile and channel (sum, M2 about the tile mean), M2 = Q - S*S/n in double from the per-wave fp32 sums of
// the biased accumulators; the sums themselves are exact-order-fixed (bitwise reproducible), the values summed are the same fp32
// accumulators as in the per-tile kernel.
#include "common.h"
#include "conv_epilogue.h"

struct ConvPsArgs {
    const float* in;
    const float* w;
    const float* bias;
    float* out;
    float* bnpart;
    int B;
};

typedef unsigned u32x4p __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_f(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
static constexpr unsigned OOBF = 0x80000000u;

template <int NT> struct KChunkPs { static constexpr int KC = NT == 32 ? 32 : 16, KCP = KC + 1; };

template <int KCH, int NCH, int H, bool DGRAD, int NT, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv5x5_mfma_ps_kernel(ConvPsArgs a, int numTiles) {
    using T = Tile<H>;
    constexpr int NB = NT / 32, NY = NCH / NT;
    constexpr int KC = KChunkPs<NT>::KC, KCP = KChunkPs<NT>::KCP, QPP = KC / 4, NCHUNK = KCH / KC;
    constexpr int IN_FLOATS = KC * T::PS;
    constexpr int W_FLOATS = DGRAD ? 5 * NT * KCP : 5 * KC * NT;
    constexpr bool HAS_BIAS = EPI != EPI_PLAIN, BNSTAT = EPI == EPI_BIAS_BNSTAT, RELU = EPI == EPI_BIAS_RELU;
    static_assert(KCH % KC == 0 && NCH % NT == 0, "channel tiling");
    __shared__ __attribute__((aligned(16))) float smem[IN_FLOATS + W_FLOATS + NCH + 2 * 4 * NT];
    float* lds_in = smem;
    float* lds_w = smem + IN_FLOATS;
    float* lds_bias = lds_w + W_FLOATS;                        // [NCH]
    float* red = lds_bias + NCH;                               // [S | Q][wave][NT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int G = gridDim.x;
    // item -> (tile, channel block): XCD x (= item & 7) owns the contiguous tile range [x PP, (x+1) PP) and walks it tile by tile, the
    // NY channel blocks of a tile back to back (conv_bf16_ps.hip)
    const int PP = cdiv(numTiles, 8), numItems = 8 * PP * NY;
    auto decode = [&](int it, int& tile, int& n0) { const int x = it & 7, j = it >> 3, jp = j / NY; n0 = (j - jp * NY) * NT; tile = jp < PP ? x * PP + jp : numTiles; };
    int it = blockIdx.x, tile, n0;
    decode(it, tile, n0);
    if (it >= numItems || tile >= numTiles) return;

    // ---- item-independent per-thread tables ----
    const int m = wave * 32 + li;                              // pixel of the tile behind MFMA column li of this wave
    const int pimg = m / (T::TH * T::TW), prem = m % (T::TH * T::TW);
    const int aBase = lh * T::PS + pimg * T::HPI + (prem / T::TW) * T::HTW + (prem % T::TW);
    const int bBase = DGRAD ? (li * KCP + lh) : (lh * NT + li);
    const int orel = (pimg * H + prem / T::TW) * H + prem % T::TW;                 // pixels, relative to the tile's first pixel
    constexpr int WQ = 5 * KC * NT / 4, WPT = (WQ + 255) / 256;
    unsigned wbase[WPT];                                       // byte offset of the unit inside the stage's slab source (without n0 / chunk / row)
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int q = tid + i * 256;
        unsigned o = 0;
        if (WQ % 256 == 0 || q < WQ) {
            if (!DGRAD) { const int row = q / (NT / 4), c4 = q % (NT / 4), s = row / KC, kc = row % KC; o = (unsigned)((s * KCH + kc) * NCH + c4 * 4) * 4u; }
            else { const int c4 = q % QPP, rown = q / QPP, n = rown % NT, s = rown / NT; o = (unsigned)(((4 - s) * NCH + n) * KCH + c4 * 4) * 4u; }     // from the row's LAST tap (lowest address)
        }
        wbase[i] = o;
    }
    constexpr int NQ = T::HP * QPP, IPT = (NQ + 255) / 256;
    int irel[IPT];
    unsigned ipk[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int q = tid + i * 256, c4 = q % QPP, hp = q / QPP, img = hp / T::HPI, rem = hp - img * T::HPI;
        const int hy = rem / T::HTW, hx = rem - hy * T::HTW;
        irel[i] = (((img * H + hy - 2) * H + hx - 2) * KCH + c4 * 4) * 4;          // bytes
        ipk[i] = (unsigned)hy | ((unsigned)hx << 8) | ((unsigned)img << 16) | ((NQ % 256 == 0 || q < NQ) ? 0x80000000u : 0u);
    }
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc_f(a.out, (unsigned)((size_t)a.B * H * H * NCH * 4));
    const __amdgpu_buffer_rsrc_t rs_bn = make_rsrc_f(a.bnpart, BNSTAT ? (unsigned)((size_t)2 * numTiles * NCH * 4) : 0u);

    struct Item { int n0, img0, ty0, tx0, ibase, obase, mt; };
    auto setup = [&](int t, int nn0) {
        Item x; x.n0 = nn0; x.mt = t;
        const int tin = t % T::TILES_PER_IMG;
        x.img0 = (t / T::TILES_PER_IMG) * T::IMGS;
        x.ty0 = (tin / T::TILES_X) * T::TH; x.tx0 = (tin % T::TILES_X) * T::TW;
        x.ibase = ((x.img0 * H + x.ty0) * H + x.tx0) * KCH * 4;               // bytes
        x.obase = (x.img0 * H + x.ty0) * H + x.tx0;                           // pixels
        return x;
    };
    f32x4 wreg[WPT], ireg[IPT];
    unsigned okmask = 0u;
    // Operands travel as BUFFER loads: uniform descriptor + 32-bit lane offset + uniform SGPR offset.  Rounds 3-4 used a laundered uniform
    // pointer + lane offset, which hipcc emits as FLAT loads: those count in lgkmcnt as well as vmcnt and return out of order with the LDS
    // reads, so every fragment wait of the MFMA phase became lgkmcnt(0) — the phase's first MFMA waited for the NEXT stage's weight slab and
    // input tile to arrive from L2 / HBM, the very loads it was meant to hide (found in round 5: 414 flat_load in this file's listing).
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc_f(a.w, (unsigned)(25u * KCH * NCH * 4u));
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc_f(a.in, (unsigned)((size_t)a.B * H * H * KCH * 4));
    auto load_w = [&](const Item& x, int cc, int r) {
        // forward: W[(r*5+s)][cc*KC+kc][n0 + ..]; dgrad: W[24-(r*5+s)][n0+n][cc*KC + ..] (flipped taps, transposed read)
        const unsigned soff = (unsigned)(DGRAD ? ((20 - r * 5) * NCH + x.n0) * KCH + cc * KC : ((r * 5) * KCH + cc * KC) * NCH + x.n0) * 4u;
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wbase[i], soff, 0));
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int q = tid + i * 256;
            if (WQ % 256 == 0 || q < WQ) {
                if (!DGRAD) *reinterpret_cast<f32x4*>(lds_w + q * 4) = wreg[i];
                else { float* d = lds_w + (q / QPP) * KCP + (q % QPP) * 4; d[0] = wreg[i].x; d[1] = wreg[i].y; d[2] = wreg[i].z; d[3] = wreg[i].w; }
            }
        }
    };
    auto load_input = [&](const Item& x, int cc) {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int hy = ipk[i] & 255, hx = (ipk[i] >> 8) & 255, img = (ipk[i] >> 16) & 255;
            const unsigned okb = (ipk[i] >> 31) & (unsigned)((unsigned)(x.ty0 + hy - 2) < (unsigned)H) &
                                 (unsigned)((unsigned)(x.tx0 + hx - 2) < (unsigned)H) & (unsigned)(x.img0 + img < a.B);
            const unsigned e = okb ? (unsigned)(x.ibase + irel[i] + cc * KC * 4) : 0u;
            ireg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, e, 0, 0));
            okmask = (okmask & ~(1u << i)) | (okb << i);
        }
    };
    auto store_input = [&]() {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256;
            if (NQ % 256 == 0 || q < NQ) {
                const bool ok = (okmask >> i) & 1u;
                float* d = lds_in + ((q % QPP) * 4) * T::PS + (q / QPP);
                d[0] = ok ? ireg[i].x : 0.f; d[T::PS] = ok ? ireg[i].y : 0.f; d[2 * T::PS] = ok ? ireg[i].z : 0.f; d[3 * T::PS] = ok ? ireg[i].w : 0.f;
            }
        }
    };

    f32x16 acc[NB], pend[NB];                                  // pend: the finished tile (bias / ReLU applied), channel-major
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int v = 0; v < 16; ++v) pend[nb][v] = 0.f;

    // the finished tile `pd`: block nb's stores (quads g0, g0 + 1) / its BatchNorm partial rows / the rows' combination
    auto drain_store = [&](const Item& pd, bool have_pd, int nb, int g0) {
        const bool ok = have_pd && pd.img0 + pimg < a.B;
        const unsigned base = ((unsigned)(pd.obase + orel) * NCH + pd.n0 + nb * 32 + 4 * lh) * 4u;
#pragma unroll
        for (int g = g0; g < g0 + 2; ++g)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4p, f32x4{pend[nb][4 * g], pend[nb][4 * g + 1], pend[nb][4 * g + 2], pend[nb][4 * g + 3]}),
                                                   rs_out, ok ? base + 32u * g : OOBF, 0, 0);
    };
    auto bn_rows = [&](const Item& pd, int nb) {
        if constexpr (BNSTAT) {
            const bool valid = pd.img0 + pimg < a.B;
            float sv[16], qv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) { sv[v] = valid ? pend[nb][v] : 0.f; qv[v] = sv[v] * sv[v]; }
            const float S = half_wave_colsum16(sv), Q = half_wave_colsum16(qv);
            const int e16 = li >> 1, ch = (e16 & 3) + 8 * (e16 >> 2) + 4 * lh;
            if ((lane & 1) == 0) { red[(0 * 4 + wave) * NT + nb * 32 + ch] = S; red[(1 * 4 + wave) * NT + nb * 32 + ch] = Q; }
        }
    };
    auto bn_combine = [&](const Item& pd, bool have_pd) {
        if constexpr (BNSTAT) {
            const int cc = tid % NT;
            float S = 0.f, Q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { S += red[(0 * 4 + w) * NT + cc]; Q += red[(1 * 4 + w) * NT + cc]; }
            int nvalid_img = a.B - pd.img0;
            if (nvalid_img > T::IMGS) nvalid_img = T::IMGS;
            if (nvalid_img < 1) nvalid_img = 1;
            const double cnt = (double)(nvalid_img * T::TH * T::TW);
            const double m2 = (double)Q - (double)S * (double)S / cnt;
            const bool ok = have_pd && tid < NT && pd.mt < numTiles;
            const unsigned o = (unsigned)(pd.mt * NCH + pd.n0 + cc) * 4u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, S), rs_bn, ok ? o : OOBF, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(m2 > 0.0 ? m2 : 0.0)), rs_bn, ok ? o + (unsigned)(numTiles * NCH) * 4u : OOBF, 0, 0);
        }
    };
    auto mfma_phase = [&](int r) {
        const float* ap = lds_in + aBase + r * T::HTW;
        __builtin_amdgcn_iglp_opt(0);
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int j = 0; j < KC / 2; ++j) {
                const float av = ap[(2 * j) * T::PS + s];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float bv = DGRAD ? lds_w[bBase + (s * NT + nb * 32) * KCP + 2 * j] : lds_w[bBase + (s * KC + 2 * j) * NT + nb * 32];
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[nb], 0, 0, 0);       // weights as A: D[channel][pixel]
                }
            }
    };

    if constexpr (HAS_BIAS) { for (int c = tid; c < NCH; c += 256) lds_bias[c] = a.bias[c]; }
    Item cur = setup(tile, n0), pd = cur;
    bool have_pd = false;
    load_w(cur, 0, 0);
    load_input(cur, 0);
    __syncthreads();
    int itemNo = 0;
    for (;;) {
        // equal shares of the issue ports for the two co-resident workgroups (conv_bf16_ps.hip)
        if (((itemNo++ ^ (blockIdx.x >= (unsigned)(G / 2))) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        int itn = it + G, tilen, n0n;
        decode(itn, tilen, n0n);
        const bool have_next = itn < numItems && tilen < numTiles;
        const Item nxt = have_next ? setup(tilen, n0n) : cur;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = HAS_BIAS ? *reinterpret_cast<const f32x4*>(lds_bias + cur.n0 + nb * 32 + 8 * g + 4 * lh) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[nb][4 * g + e] = b4[e];
            }
        // chunk 0: five stages, each with its share of the previous item's tile (stores after the slot's loads)
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            __syncthreads();
            if (r == 0) store_input();
            store_w();
            if (r == 0) { if (NCHUNK > 1) load_input(cur, 1); else load_input(nxt, 0); }
            if (r < 4) load_w(cur, 0, r + 1); else if (NCHUNK > 1) load_w(cur, 1, 0); else load_w(nxt, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (r == 0) drain_store(pd, have_pd, 0, 0);
            if (r == 1) { drain_store(pd, have_pd, 0, 2); bn_rows(pd, 0); }
            if (r == 2 && NB > 1) { drain_store(pd, have_pd, NB - 1, 0); bn_rows(pd, NB - 1); }
            if (r == 3 && NB > 1) drain_store(pd, have_pd, NB - 1, 2);
            if (r == 4) bn_combine(pd, have_pd);              // the rows were written in stages 1 / 2: two barriers ago
            __syncthreads();
            mfma_phase(r);
        }
        for (int cc = 1; cc < NCHUNK; ++cc) {
            const bool lastc = cc == NCHUNK - 1;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                __syncthreads();
                if (r == 0) store_input();
                store_w();
                if (r == 0) { if (lastc) load_input(nxt, 0); else load_input(cur, cc + 1); }
                if (r < 4) load_w(cur, cc, r + 1); else if (lastc) load_w(nxt, 0, 0); else load_w(cur, cc + 1, 0);
                __syncthreads();
                mfma_phase(r);
            }
        }
        // the finished tile moves to the second register set (ReLU applied); everything else about it happens in the next item's slots
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) pend[nb][v] = RELU ? fmaxf(acc[nb][v], 0.f) : acc[nb][v];
        pd = cur; have_pd = true;
        if (!have_next) break;
        cur = nxt; it = itn;
    }
    // ---- tail: the last item's tile ----
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { drain_store(pd, true, nb, 0); drain_store(pd, true, nb, 2); bn_rows(pd, nb); }
    __syncthreads();
    bn_combine(pd, true);
}

template <int KCH, int NCH, int H, bool DGRAD, int NT, int EPI>
static int run_mfma_ps(const ConvPsArgs& a, hipStream_t st) {
    using T = Tile<H>;
    constexpr int NY = NCH / NT;
    // the kernel addresses both tensors with 32-bit byte offsets and marks skipped lanes with the offset 0x80000000: tensors of 2 GiB and
    // more take the per-tile kernel (size_t addressing) — the caller falls back on -100
    if ((size_t)a.B * H * H * KCH * 4 >= (1ull << 31) || (size_t)a.B * H * H * NCH * 4 >= (1ull << 31)) return -100;
    if (g_conv_dry) return 0;
    const int numTiles = cdiv(a.B, T::IMGS) * T::TILES_PER_IMG;
    const int numItems = cdiv(numTiles, 8) * 8 * NY;
    int G = 2 * cvae_num_cus();
    G -= G % 8;
    if (G < 8) G = 8;
    if (G > numItems) G = numItems;
    cvae_probe_begin(st);
    hipLaunchKernelGGL((conv5x5_mfma_ps_kernel<KCH, NCH, H, DGRAD, NT, EPI>), dim3(G), dim3(256), 0, st, a, numTiles);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    return 0;
}

// returns -100 when the layer has no persistent instantiation (the caller falls back to the per-tile kernel).  Built by default: the
// 64-channel-tile layers the default mask selects (E2 / E3 / E4 forward, E3 / E4 input gradient at 64 x 64; E2 / E3 forward and E3 input gradient
// at 128 x 128).  -DCONVF_PS_ALL adds the others — the 32-channel-tile ones spill and lose (header comment), the rest were never the default —
// at 8 minutes of compile time.
int launch_conv_mfma_ps(int layer, int width, bool dgrad, int B, const float* in, const float* w, const float* bias, float* out, float* bnpart, hipStream_t st) {
    const ConvPsArgs a{in, w, bias, out, bnpart, B};
    if (!dgrad) {
        if (width == 64) {
            switch (layer) {
                case 1: return run_mfma_ps<32, 64, 32, false, 64, EPI_BIAS_BNSTAT>(a, st);
                case 2: return run_mfma_ps<64, 128, 16, false, 64, EPI_BIAS_BNSTAT>(a, st);
#ifdef CONVF_PS_ALL
                case 3: return run_mfma_ps<128, 256, 8, false, 32, EPI_BIAS_BNSTAT>(a, st);
#else
                case 3: return run_mfma_ps<128, 256, 8, false, 64, EPI_BIAS_BNSTAT>(a, st);      // round 5: E4 on 64-channel tiles (no spills): 215 -> 207 us
#endif
            }
        } else if (width == 128) {
            switch (layer) {
                case 1: return run_mfma_ps<32, 64, 64, false, 64, EPI_BIAS_BNSTAT>(a, st);
                case 2: return run_mfma_ps<64, 128, 32, false, 64, EPI_BIAS_BNSTAT>(a, st);
#ifdef CONVF_PS_ALL
                case 3: return run_mfma_ps<128, 256, 16, false, 64, EPI_BIAS_BNSTAT>(a, st);
                case 4: return run_mfma_ps<256, 128, 8, false, 64, EPI_BIAS_RELU>(a, st);
#endif
            }
        }
    } else {
        if (width == 64) {
            switch (layer) {
                case 2: return run_mfma_ps<128, 64, 16, true, 64, EPI_PLAIN>(a, st);
#ifdef CONVF_PS_ALL
                case 1: return run_mfma_ps<64, 32, 32, true, 32, EPI_PLAIN>(a, st);
                case 3: return run_mfma_ps<256, 128, 8, true, 32, EPI_PLAIN>(a, st);
#else
                case 3: return run_mfma_ps<256, 128, 8, true, 64, EPI_PLAIN>(a, st);              // 212 -> 211 us
#endif
            }
        } else if (width == 128) {
            switch (layer) {
                case 2: return run_mfma_ps<128, 64, 32, true, 64, EPI_PLAIN>(a, st);
#ifdef CONVF_PS_ALL
                case 1: return run_mfma_ps<64, 32, 64, true, 32, EPI_PLAIN>(a, st);
                case 3: return run_mfma_ps<256, 128, 16, true, 64, EPI_PLAIN>(a, st);
                case 4: return run_mfma_ps<128, 256, 8, true, 64, EPI_PLAIN>(a, st);
#endif
            }
        }
    }
    return -100;
}
