// conv_thin.hip — the two 3-channel convolutions, restructured so the MFMA never sees N=3.
//
// E1  (vae_nets.py:69, Conv2d(3,32,5,1,2) on the NCHW frame x):
//     forward  = GEMM  M=pixels, N=32, K=75(+1 zero)   — x halo staged as 3 LDS planes
//     wgrad    = GEMM  M=75(+21), N=32, K=pixels       — split-K slabs, fixed-order reduce
// D4  (vae_nets.py:133-134, Upsample(2) -> Conv2d(32,3,5,1,2) -> Tanh, NCHW recon out):
//     forward  : Q[src][(tap,co)] = o3[src][:] . W[tap][:][co]  (GEMM M=src pixels, N=75, K=32,
//                computed at the LOW resolution) then recon[co][y][x] = tanh(b + sum_tap
//                Q[((y+r-2)>>1,(x+s-2)>>1)][(tap,co)])  — the upsample never exists.
//     backward : with dOut = d_recon*(1-recon^2) and G[src][(tap,co)] = sum over the 2x2 block q
//                of src of dOut[co][q-(tap-2)]:   d_o3[src][ci] = relu'(o3) * sum_k G[src][k] W[k][ci]
//                (GEMM N=32, K=75) and dW[k][ci] = sum_src G[src][k] o3[src][ci] (GEMM M=75, N=32,
//                K=src pixels) — one fused kernel builds each G tile once in LDS for both.
#include "common.h"
#include "conv_epilogue.h"
#ifndef THIN_F32_OCC
#define THIN_F32_OCC 2      // waves per SIMD promised to the compiler for the fp32 E1 / D4 kernels (e1_fwd 160 -> 110 VGPRs; -1..2 us each)
#endif

// LDS offset of GEMM-k = tap*3 + ci inside the 3-plane x halo (k = 75 is the zero pad row)
template <int PS, int HTW>
__device__ __forceinline__ constexpr int e1_off(int k) {
    return k >= 75 ? 0 : (k % 3) * PS + ((k / 3) / 5) * HTW + (k / 3) % 5;
}

// E1 forward.  WG = a 16-row x 32-column strip (512 pixels) of one image; wave w owns rows 4w..4w+3
// (four 32-pixel accumulator tiles).  The whole B operand (76 x 32 weights) lives in 38 registers per
// lane, read once from global; the x halo (3 planes) is staged once per strip.  Epilogue: bias,
// NHWC store, and the strip's BatchNorm partial (sum, M2 about the strip mean).
// (precision mode 1 runs e1_fwd_bf16_kernel below instead.)
template <int H>
__global__ __launch_bounds__(256, THIN_F32_OCC) void e1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     float* __restrict__ bnpart, int B) {
    constexpr int SR = 16, SW = 32, HW_ = SW + 4, HR_ = SR + 4, PS = ((3 * 0 + HW_ * HR_ + 5) / 8) * 8 + 2;
    constexpr int SX = H / SW, SY = H / SR;
    __shared__ float lds_x[3 * PS];
    __shared__ __attribute__((aligned(16))) float patch_all[4 * 32 * 36];
    __shared__ float red[2][4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int strip = xcd_tile(blockIdx.x, gridDim.x);      // each XCD owns a contiguous range of strips: shared halo rows meet in ONE L2
    const int ib = strip / (SX * SY), t = strip % (SX * SY);
    const int ty0 = (t / SX) * SR, tx0 = (t % SX) * SW;
    float bw[38];
#pragma unroll
    for (int j = 0; j < 38; ++j) bw[j] = (2 * j + lh < 75) ? w[(2 * j + lh) * 32 + li] : 0.f;
    {       // all halo loads are issued before the first LDS write (clamped address + select: no branch per element)
        constexpr int NQ = 3 * HW_ * HR_, NIT = (NQ + 255) / 256;
        float xv[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int q = tid + i * 256, c = q / (HW_ * HR_), hp = q % (HW_ * HR_);
            const int gy = ty0 + hp / HW_ - 2, gx = tx0 + hp % HW_ - 2;
            const bool ok = q < NQ && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)H;
            const float l = x[ok ? ((size_t)(ib * 3 + c) * H + gy) * H + gx : 0];
            xv[i] = ok ? l : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int q = tid + i * 256;
            if (q < NQ) lds_x[(q / (HW_ * HR_)) * PS + q % (HW_ * HR_)] = xv[i];
        }
    }
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[r][v] = 0.f;
    const int aBase = (wave * 4) * HW_ + li;
#pragma unroll
    for (int j = 0; j < 38; ++j) {
        const int off = lh ? e1_off<PS, HW_>(2 * j + 1) : e1_off<PS, HW_>(2 * j);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(lds_x[aBase + r * HW_ + off], bw[j], acc[r], 0, 0, 0);
    }
    // epilogue: element v of lane (li, lh) in tile r = pixel column (v&3)+8*(v>>2)+4*lh of row 4*wave+r, channel li
    const float bv = bias[li];
    float s = 0.f;
    float* patch = patch_all + wave * (32 * 36);       // per-wave transpose patch -> 16-byte stores
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float val = acc[r][v] + bv;
            acc[r][v] = val;
            s += val;
            patch[((v & 3) + 8 * (v >> 2) + 4 * lh) * 36 + li] = val;
        }
        const int gy = ty0 + wave * 4 + r;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 64 + lane, px = idx >> 3, c4 = idx & 7;
            const float4 val = *reinterpret_cast<const float4*>(patch + px * 36 + c4 * 4);
            *reinterpret_cast<float4*>(y + ((size_t)(ib * H + gy) * H + tx0 + px) * 32 + c4 * 4) = val;
        }
    }
    s += __shfl_xor(s, 32, 64);
    if (lh == 0) red[0][wave][li] = s;
    __syncthreads();
    const float mean = ((red[0][0][li] + red[0][1][li]) + (red[0][2][li] + red[0][3][li])) * (1.0f / (SR * SW));
    float m2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int v = 0; v < 16; ++v) { const float d = acc[r][v] - mean; m2 += d * d; }
    m2 += __shfl_xor(m2, 32, 64);
    if (lh == 0) red[1][wave][li] = m2;
    __syncthreads();
    if (tid < 32) {
        const size_t nt = gridDim.x;
        bnpart[(size_t)strip * 32 + tid] = (red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]);
        bnpart[(nt + strip) * 32 + tid] = (red[1][0][tid] + red[1][1][tid]) + (red[1][2][tid] + red[1][3][tid]);
    }
    (void)B;
}

// precision mode 1, E1 forward.  The frame strip is staged ONCE as bf16 pixels of 4 channels (r, g, b, 0) — 8 bytes per
// pixel, rows of 40 pixels — so the 4 channels of one tap are ONE 8-byte LDS unit: an A fragment (8 consecutive k) is
// two 8-byte LDS reads instead of 8 scalar reads + 8 conversions.  K = 25 taps x 4 = 100 padded to 112 (zero weights):
// 7 MFMAs per 32 pixels (e1_wk_* above), 28 weight registers per lane.
// y1 leaves as bf16 (16-byte units), the BatchNorm partials come from the fp32 accumulators.
// PASS (the bf16 training step runs the kernel TWICE instead of reading y1 back — the 75-tap conv is ~1 % of the
// step's MFMA work, the tensor it produces is the step's largest):
//   E1_Y     : y1 + BatchNorm partials (the stand-alone conv op).
//   E1_STATS : BatchNorm partials only — nothing but x is read, 2 x 32 floats per strip are written.
//   E1_POOL  : after the statistics are merged: on the bf16-rounded conv values, exactly as bn_pool_act_fwd_bf16_kernel<0>
//              would compute from a stored y1, scale/shift -> first maximum of each 2x2 window in scan order -> ReLU -> a1.
//              y1 ITSELF IS NOT WRITTEN (round 3): the only reader left in the step, E1's weight-gradient kernel, runs the
//              75-tap conv a third time on the tile it stages (e1_wgrad_bf16_kernel<H, true>) — B*H*H*32*2 bytes less
//              written here and read there (1.07 GB per step at B = 2048).  Exceptions: keepY != 0 (the CVAE_FUSE_E1=0
//              A/B path, whose separate apply pass reads y1), and steps in which a channel has |gamma| < 1e-2: the
//              BatchNorm-backward statistics kernel then takes xhat of that channel from y1 (bn.hip).
// K-packed form of the 75-tap contraction (round 3).  The staged strip holds bf16 pixels of 4 channels (r, g, b, 0), so the
// four channels of one tap are one 8-byte LDS unit and an MFMA's K = 16 is FOUR taps (two per lane half): 25 taps -> 7
// MFMAs (the kernel-row form padded each row's 5 taps to 8: 10 MFMAs).  Tap assignment — chosen so that lane half 1 reads
// at a FIXED pixel offset from lane half 0 (two extra base pointers instead of per-lane offset tables):
//   MFMA j = 0..4 : kernel row j, columns (0, 1) | (2, 3)           half 1 = half 0 + 2 columns
//   MFMA 5        : column 4, rows (0, 1) | (2, 3)                  half 1 = half 0 + 2 rows
//   MFMA 6        : tap (4, 4), then zero weights                   (both halves read tap (4,4): finite values x 0)
// The forward passes AND the weight-gradient kernel's recompute run exactly this sequence (j = 0..6 into one
// accumulator), so they produce the same fp32 sums to the bit.
__device__ __forceinline__ constexpr int e1_wk_tap(int j, int lh, int u) {        // -> r*5 + s, or -1 (zero weight)
    return j < 5 ? j * 5 + 2 * lh + u : (j == 5 ? (2 * lh + u) * 5 + 4 : (lh == 0 && u == 0 ? 24 : -1));
}
__device__ __forceinline__ void e1_wk_load(const float* __restrict__ w, int li, int lh, bf16x8 (&bw)[7]) {
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = lh ? e1_wk_tap(j, 1, e >> 2) : e1_wk_tap(j, 0, e >> 2), c = e & 3;
            bw[j][e] = (__bf16)((t >= 0 && c < 3) ? w[(t * 3 + c) * 32 + li] : 0.f);
        }
}
// p = this lane's output pixel in the strip image (tap (0,0)), row stride HWX units
template <int HWX>
__device__ __forceinline__ f32x16 e1_wk_conv(const bf16x4* __restrict__ p, int lh, const bf16x8 (&bw)[7]) {
    const bf16x4* pc = p + 2 * lh;                  // lane half 1: + 2 columns
    const bf16x4* pr = p + 2 * HWX * lh;            // lane half 1: + 2 rows
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const bf16x4 lo = j < 5 ? pc[j * HWX] : (j == 5 ? pr[4] : p[4 * HWX + 4]);
        const bf16x4 hi = j < 5 ? pc[j * HWX + 1] : (j == 5 ? pr[HWX + 4] : p[4 * HWX + 4]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7), bw[j], acc, 0, 0, 0);
    }
    return acc;
}

typedef float f32x2t __attribute__((ext_vector_type(2)));
enum { E1_Y = 0, E1_STATS = 1, E1_POOL = 2, E1_POOL_X = 3 };      // E1_POOL_X: the pool pass staging its strips from the fp32 frame (eval mode: no statistics pass ran)
template <int H, int PASS>
__global__ __launch_bounds__(256, PASS == 1 ? 3 : 2) void e1_fwd_bf16_kernel(      // statistics pass: 3 workgroups per CU (<= 168 VGPRs)
                                                         const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          float* __restrict__ bnpart, int B,
                                                          const float* __restrict__ coef, float* __restrict__ a1, int numStrips, int keepY,
                                                          bf16x4* __restrict__ xp) {
    // xp (round 5): the frame as packed bf16 pixels (r, g, b, 0) = the LDS unit of this kernel, 8 bytes per pixel, in the workspace.  The
    // statistics pass writes it from the strip it stages anyway (interior pixels: one 8-byte store per pixel and step); the pool pass and
    // E1's weight-gradient kernel then stage their strips as plain 8-byte units — buffer loads whose padding lanes carry an out-of-range
    // offset (the load returns 0) — instead of three 4-byte plane loads + three conversions + a select per pixel, three times per step.
    // The operands are bit-identical (x is rounded to bf16 exactly once either way).
    constexpr int SR = 16, SW = 32, HR_ = SR + 4, HWX = 40;
    constexpr bool XP_RD = PASS == E1_POOL, XP_WR = PASS == E1_STATS, POOL = PASS == E1_POOL || PASS == E1_POOL_X;
    constexpr int SX = H / SW, SY = H / SR;
    __shared__ __attribute__((aligned(16))) bf16x4 lds_x[HR_ * HWX];
    __shared__ __attribute__((aligned(16))) float patch_all[4 * 32 * 36];
    __shared__ float red[2][4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    // persistent: the 80 weights of this lane are fetched and converted ONCE per workgroup (one workgroup per strip spent
    // more instructions on them than on the strip), and the next strip's frame values travel during this strip's MFMAs
    bf16x8 bw[7];                       // K-packed B fragments (e1_wk_load)
    e1_wk_load(w, li, lh, bw);
    const float bv = bias[li];
    bool wr_y = PASS != E1_STATS;
    if constexpr (POOL) {
        const float gam = coef[li * 4] / coef[li * 4 + 3];             // gamma = scale / invstd, as bn.hip's backward tests it
        wr_y = keepY != 0 || __any(!(fabsf(gam) >= 1e-2f));           // the same answer in every wave (a wave spans all 32 channels)
    }
    constexpr int NIT = (HR_ * HWX + 255) / 256;
    float v0[NIT], v1[NIT], v2[NIT];
    typedef unsigned u32x2x __attribute__((ext_vector_type(2)));
    [[maybe_unused]] u32x2x pk[NIT];            // XP_RD: the strip's packed units
    unsigned okm = 0u;                          // validity bit per staged unit (frame pixels vs zero padding)
    constexpr unsigned XP_BIAS = (2 * H + 2) * 8;       // the descriptor starts this far in front of xp: halo offsets are non-negative lane offsets
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_xp =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(xp) - XP_BIAS, 0, (int)((size_t)B * H * H * 8 + XP_BIAS), 0x00020000);
    auto fetch = [&](int strip) {                // all loads first (clamped address), the LDS writes (+ zero select) follow later
        const int ib = strip / (SX * SY), t = strip % (SX * SY);
        const int ty0 = (t / SX) * SR, tx0 = (t % SX) * SW;
        if constexpr (XP_RD) {
            const unsigned soff = (unsigned)(((ib * H + ty0) * H + tx0) * 8);
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int q = tid + i * 256, hy = q / HWX, hx = q % HWX;
                const bool ok = q < HR_ * HWX && (unsigned)(ty0 + hy - 2) < (unsigned)H && (unsigned)(tx0 + hx - 2) < (unsigned)H;
                pk[i] = __builtin_amdgcn_raw_buffer_load_b64(rs_xp, ok ? (unsigned)(((hy - 2) * H + hx - 2) * 8 + (int)XP_BIAS) : 0x80000000u, soff, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int q = tid + i * 256, hy = q / HWX, hx = q % HWX;
            const int gy = ty0 + hy - 2, gx = tx0 + hx - 2;
            const bool ok = q < HR_ * HWX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)H;
            const size_t e = ok ? ((size_t)(ib * 3) * H + gy) * H + gx : 0;
            // raw values now, the zero padding is selected when the strip is staged (a turn later): a select HERE makes the
            // compiler wait for the loads right behind their issue — 2.0-2.6 k of a strip's 6.6-7.5 k cycles were that wait
            // (clock64 stamps around the strip phases in a throw-away build), the prefetch hid nothing
            v0[i] = x[e]; v1[i] = x[e + (size_t)H * H]; v2[i] = x[e + 2 * (size_t)H * H];
            okm = ok ? (okm | (1u << i)) : (okm & ~(1u << i));
        }
    };
    // XCD-aware strip order (round 4): workgroups are dealt to the 8 XCDs round-robin, and with strip = workgroup + k * grid the
    // neighbouring strips of a frame — which share 4 of their 20 halo rows and 8 of their 40 columns — ran on different XCDs:
    // every L2 fetched its own copy.  Measured with the kernel's exact loads (profiles/experiments/e1_fetch_probe.hip,
    // FETCH_SIZE): 246.6 MB for the 100.7 MB tensor, 100.7 MB once each XCD owns a contiguous range of strips.
    const int G = gridDim.x, b = blockIdx.x;
    const bool xcd = !((G & 7) || (numStrips & 7));
    const int sbase = xcd ? (b & 7) * (numStrips >> 3) + (b >> 3) : b, sstep = xcd ? G >> 3 : G;      // strip of turn k = sbase + k * sstep
    if (b < numStrips) fetch(sbase);
    for (int n = b, strip = sbase; n < numStrips; n += G, strip += sstep) {
        const int ib = strip / (SX * SY), t = strip % (SX * SY);
        const int ty0 = (t / SX) * SR, tx0 = (t % SX) * SW;
        __syncthreads();                         // every wave is done with the previous strip's LDS image
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int q = tid + i * 256;
            if constexpr (XP_RD) {
                if (q < HR_ * HWX) lds_x[q] = __builtin_bit_cast(bf16x4, pk[i]);
            } else {
                const bool ok = (okm >> i) & 1u;
                bf16x4 u; u[0] = (__bf16)(ok ? v0[i] : 0.f); u[1] = (__bf16)(ok ? v1[i] : 0.f); u[2] = (__bf16)(ok ? v2[i] : 0.f); u[3] = (__bf16)0.f;
                if (q < HR_ * HWX) lds_x[q] = u;
                if constexpr (XP_WR) {               // the strip's own 16 x 32 pixels (every pixel of the frame is interior to exactly one strip)
                    const int hy = q / HWX, hx = q % HWX;
                    const bool mine = xp != nullptr && q < HR_ * HWX && hy >= 2 && hy < SR + 2 && hx >= 2 && hx < SW + 2;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2x, u), rs_xp,
                        mine ? (unsigned)(((ib * H + ty0 + hy - 2) * H + tx0 + hx - 2) * 8 + (int)XP_BIAS) : 0x80000000u, 0, 0);
                }
            }
        }
        __syncthreads();
        if (n + G < numStrips) fetch(strip + sstep);
        f32x16 acc[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[rr] = e1_wk_conv<HWX>(lds_x + (wave * 4 + rr) * HWX + li, lh, bw);
        // epilogue: element v of lane (li, lh) in tile r = pixel column (v&3)+8*(v>>2)+4*lh of row 4*wave+r, channel li
        // BatchNorm partial of the strip (sum, M2 about its mean) — round 5: ONE pass of packed sums over the raw accumulators (S0, Q0 of
        // conv without the bias: M2 = Q0 - S0^2 / n is shift-invariant and better conditioned without it, the bias re-enters the sum as
        // n * bias), one barrier; rounds 2-4 added the bias first, summed, met, subtracted the strip mean and summed squares behind a
        // second barrier: ~190 more VALU instructions per wave and strip in a kernel that is issue-bound at three waves per SIMD.
        if constexpr (!POOL) {
            f32x2t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int v = 0; v < 16; v += 2) {
                    const f32x2t xx = {acc[r][v], acc[r][v + 1]};
                    s2 += xx;
                    q2 = __builtin_elementwise_fma(xx, xx, q2);
                }
            float S0 = s2.x + s2.y, Q0 = q2.x + q2.y;
            S0 += __shfl_xor(S0, 32, 64); Q0 += __shfl_xor(Q0, 32, 64);
            if (lh == 0) { red[0][wave][li] = S0; red[1][wave][li] = Q0; }
        }
        float* patch = patch_all + wave * (32 * 36);
        if (POOL || wr_y) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float val = acc[r][v] + bv;
                    acc[r][v] = val;
                    if (wr_y) patch[((v & 3) + 8 * (v >> 2) + 4 * lh) * 36 + li] = val;
                }
                if (!wr_y) continue;
                const int gy = ty0 + wave * 4 + r;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int idx = it * 64 + lane, px = idx >> 2, c8 = idx & 3;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + px * 36 + c8 * 8);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + px * 36 + c8 * 8 + 4);
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (__bf16)lo[e]; o[4 + e] = (__bf16)hi[e]; }
                    Act<__bf16>::st8(y, ((size_t)(ib * H + gy) * H + tx0 + px) * 32 + c8 * 8, o);
                }
            }
        }
        if constexpr (POOL) {
            // BatchNorm + 2x2 max + ReLU of this wave's 4 rows x 32 columns: both rows and both columns of a window sit in
            // this lane's accumulators (rows 2q / 2q+1, elements v / v+1 for even v)
            const float sc = coef[li * 4], sh = coef[li * 4 + 1];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int v = 0; v < 16; v += 2) {
                    float m = 0.f;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float yb = (float)(__bf16)acc[2 * q + (p >> 1)][v + (p & 1)];       // the value y1 holds
                        const float n = fmaf(yb, sc, sh);
                        m = (p == 0 || n > m) ? n : m;
                    }
                    const int pcol = ((v & 3) + 8 * (v >> 2) + 4 * lh) >> 1;
                    patch[(q * 16 + pcol) * 36 + li] = fmaxf(m, 0.f);
                }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = it * 64 + lane, pp = idx >> 2, c8 = idx & 3;                  // pooled pixel pp = q*16 + pcol
                const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + pp * 36 + c8 * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + pp * 36 + c8 * 8 + 4);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (__bf16)lo[e]; o[4 + e] = (__bf16)hi[e]; }
                const int prow = ty0 / 2 + wave * 2 + (pp >> 4), pc = tx0 / 2 + (pp & 15);
                Act<__bf16>::st8(a1, ((size_t)(ib * (H / 2) + prow) * (H / 2) + pc) * 32 + c8 * 8, o);
            }
            continue;                                        // the statistics came from the E1_STATS pass
        }
        __syncthreads();
        if (tid < 32) {
            const float S = (red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]);
            const float Q = (red[1][0][tid] + red[1][1][tid]) + (red[1][2][tid] + red[1][3][tid]);
            const double m2 = (double)Q - (double)S * (double)S * (1.0 / (SR * SW));
            bnpart[(size_t)strip * 32 + tid] = S + (float)(SR * SW) * bias[tid];
            bnpart[((size_t)numStrips + strip) * 32 + tid] = (float)(m2 > 0.0 ? m2 : 0.0);
        }
    }
    (void)B;
}

struct ThinWgradArgs {
    const float* a0;     // E1: x (NCHW)          D4: dOut planes (NCHW); bf16 mode: d_recon (the Tanh backward is applied while staging)
    const float* a2;     //                       D4, bf16 mode: recon (NCHW)
    const float* a1;     // E1: dy (NHWC, 32)     D4: o3 (NHWC, 32)
    const float* w;      // D4: W4 [25][32][3]
    float* din;          // D4: d_o3 (NHWC, 32)
    float* slab;         // [S][96][32]
    int B, numTiles, tilesPerSplit;
    const void* xp = nullptr;      // E1, bf16 mode: the packed bf16 frame (r, g, b, 0) the forward's statistics pass wrote, or null
};

// sum acc[3] over the 4 waves (through LDS, fixed order) and write the block's slab
__device__ __forceinline__ void thin_slab_out(f32x16 (&acc)[3], float* red, float* slab_blk, float bsum = 0.f, bool with_bias = false) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
            for (int v = 0; v < 16; ++v) red[(((wave - 1) * 3 + mb) * 16 + v) * 64 + lane] = acc[mb][v];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float x = acc[mb][v];
#pragma unroll
                for (int w2 = 0; w2 < 3; ++w2) x += red[((w2 * 3 + mb) * 16 + v) * 64 + lane];
                const int k = mb * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                // E1: row 75 is the K pad (its accumulator saw real pixels through offset 0 -> write the
                // zero it stands for), row 76 is reserved for the bias-gradient partial below
                if (!with_bias) slab_blk[k * 32 + li] = x;
                else if (k < 75) slab_blk[k * 32 + li] = x;
                else if (k == 75) slab_blk[k * 32 + li] = 0.f;
            }
        if (with_bias) {                       // row 76 of the slab carries the bias-gradient partial
            bsum += __shfl_xor(bsum, 32, 64);
            if (lh == 0) slab_blk[76 * 32 + li] = bsum;
        }
    }
}

// FUSE: the weight-gradient kernels take E1's BatchNorm/MaxPool/ReLU backward in while they stage a tile, instead of
// reading a dy tensor that bn.hip's apply pass wrote (nothing else reads dy of block 0 — E1 has no input gradient):
// per 2x2 window and channel, dy[p] = scale*((p == argmax ? g : 0) - k1 - xhat[p]*k2) with g = da*[a > 0],
// xhat = (y - mean)*invstd and (k1, k2) = bcoef — the arithmetic of bn_bwd_kernel<0,1>, so the result is the
// same to the bit; the step saves one write and one read of the largest activation gradient (B x 64 x 64 x 32).
struct E1Fuse { const float *y, *a, *da, *coef, *bcoef, *w, *bias; };     // w, bias: E1's conv parameters (bf16 mode recomputes y)

// tile mt of E1 wgrad into registers: x halo (3 planes, zero padded) and the 128x32 dy tile
// (FUSE: thread = (channel quad, window column, window row): the window's four y quads in rd, a and da quads in rf)
template <int H, bool FUSE>
__device__ __forceinline__ void e1_wgrad_fetch(const ThinWgradArgs& a, const E1Fuse& fu, int mt, float (&rx)[(3 * Tile<H>::HPI + 255) / 256],
                                               f32x4 (&rd)[4], f32x4 (&rf)[2], unsigned& okm) {
    using T = Tile<H>;
    const int tid = threadIdx.x;
    const int ib = mt / T::TILES_PER_IMG, tileInImg = mt % T::TILES_PER_IMG;
    const int ty0 = (tileInImg / T::TILES_X) * T::TH, tx0 = (tileInImg % T::TILES_X) * T::TW;
#pragma unroll
    for (int i = 0; i < (3 * T::HPI + 255) / 256; ++i) {
        const int q = tid + i * 256;
        const int c = q / T::HPI, hp = q % T::HPI;
        const int gy = ty0 + hp / T::HTW - 2, gx = tx0 + hp % T::HTW - 2;
        const bool ok = q < 3 * T::HPI && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)H;
        // no branch around the load, and no select behind it either: the zero padding is selected when the tile is staged
        // (a select here makes the compiler wait for the load at once, and the prefetch hides nothing)
        rx[i] = a.a0[ok ? ((size_t)(ib * 3 + c) * H + gy) * H + gx : 0];
        okm = ok ? (okm | (1u << i)) : (okm & ~(1u << i));
    }
    if constexpr (FUSE) {
        const int c4 = tid & 7, gy = ty0 + 2 * (tid >> 7), gx = tx0 + 2 * ((tid >> 3) & 15);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            rd[p] = *reinterpret_cast<const f32x4*>(fu.y + ((size_t)(ib * H + gy + (p >> 1)) * H + gx + (p & 1)) * 32 + c4 * 4);
        const size_t pe = ((size_t)(ib * (H / 2) + gy / 2) * (H / 2) + gx / 2) * 32 + c4 * 4;
        rf[0] = *reinterpret_cast<const f32x4*>(fu.a + pe);
        rf[1] = *reinterpret_cast<const f32x4*>(fu.da + pe);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * 256, c4 = q & 7, mm = q >> 3;
            const int gy = ty0 + mm / T::TW, gx = tx0 + mm % T::TW;
            rd[i] = *reinterpret_cast<const f32x4*>(a.a1 + ((size_t)(ib * H + gy) * H + gx) * 32 + c4 * 4);
        }
    }
}

// (precision mode 1 runs e1_wgrad_bf16_kernel below instead.)
template <int H, bool FUSE>
__global__ __launch_bounds__(256, THIN_F32_OCC) void e1_wgrad_kernel(ThinWgradArgs a, E1Fuse fu) {
    using T = Tile<H>;
    static_assert(!FUSE || (T::TW == 32 && T::TH == 4 && T::IMGS == 1), "fused staging: 2 x 16 windows per tile");
    constexpr int X_FLOATS = ((3 * T::PS + 3) / 4) * 4;
    constexpr int RED = 3 * 3 * 1024;
    constexpr int SM = (X_FLOATS + 128 * 32) > RED ? (X_FLOATS + 128 * 32) : RED;
    __shared__ __attribute__((aligned(16))) float smem[SM];
    float* lds_x = smem;
    float* lds_d = smem + X_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    int aoff[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) {
        const int k = mb * 32 + li;
        aoff[mb] = k >= 75 ? 0 : (k % 3) * T::PS + ((k / 3) / 5) * T::HTW + (k / 3) % 5;
    }
    f32x16 acc[3];
    float bsum = 0.f;
#pragma unroll
    for (int mb = 0; mb < 3; ++mb)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[mb][v] = 0.f;
    const int t0 = blockIdx.x * a.tilesPerSplit;
    int t1 = t0 + a.tilesPerSplit; if (t1 > a.numTiles) t1 = a.numTiles;
    // software pipeline: the next tile's x halo and dy tile are fetched into registers while the
    // MFMAs of the current tile run; LDS is refilled between two barriers
    constexpr int XQ = (3 * T::HPI + 255) / 256;
    float rx[XQ];
    unsigned okm = 0u;                     // validity bit per staged x unit (frame pixel vs zero padding)
    f32x4 rd[4], rf[2];
    float bsc[4], bsh[4], bmean[4], binv[4], bk1[4], bk2[4];          // FUSE: this thread's four channels
    if constexpr (FUSE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (tid & 7) * 4 + e;
            bsc[e] = fu.coef[c * 4]; bsh[e] = fu.coef[c * 4 + 1]; bmean[e] = fu.coef[c * 4 + 2]; binv[e] = fu.coef[c * 4 + 3];
            bk1[e] = fu.bcoef[c * 2]; bk2[e] = fu.bcoef[c * 2 + 1];
        }
    }
    if (t0 < t1) e1_wgrad_fetch<H, FUSE>(a, fu, t0, rx, rd, rf, okm);
    for (int mt = t0; mt < t1; ++mt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < XQ; ++i) {
            const int q = tid + i * 256;
            if (q < 3 * T::HPI) lds_x[(q / T::HPI) * T::PS + q % T::HPI] = ((okm >> i) & 1u) ? rx[i] : 0.f;
        }
        if constexpr (FUSE) {
            f32x4 d[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m = 0.f; int pos = 0;
#pragma unroll
                for (int p = 0; p < 4; ++p) {                           // first maximum in scan order, like the forward
                    const float n = fmaf(rd[p][e], bsc[e], bsh[e]);
                    if (p == 0 || n > m) { m = n; pos = p; }
                }
                const float g = rf[0][e] > 0.f ? rf[1][e] : 0.f;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float xhat = (rd[p][e] - bmean[e]) * binv[e];
                    d[p][e] = bsc[e] * ((p == pos ? g : 0.f) - bk1[e] - xhat * bk2[e]);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int mm = (2 * (tid >> 7) + (p >> 1)) * 32 + 2 * ((tid >> 3) & 15) + (p & 1);
                *reinterpret_cast<f32x4*>(lds_d + mm * 32 + (tid & 7) * 4) = d[p];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = tid + i * 256;
                *reinterpret_cast<f32x4*>(lds_d + (q >> 3) * 32 + (q & 7) * 4) = rd[i];
            }
        }
        __syncthreads();
        if (mt + 1 < t1) e1_wgrad_fetch<H, FUSE>(a, fu, mt + 1, rx, rd, rf, okm);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int mm = wave * 32 + 2 * kk + lh;
            const int poff = (mm / T::TW) * T::HTW + mm % T::TW;
            const float bv = lds_d[mm * 32 + li];
            bsum += bv;
#pragma unroll
            for (int mb = 0; mb < 3; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(lds_x[aoff[mb] + poff], bv, acc[mb], 0, 0, 0);
        }
    }
    // each wave summed its own 32 pixels per tile: total over the 4 waves
    __shared__ float bred[4][32];
    bsum += __shfl_xor(bsum, 32, 64);
    if (lh == 0) bred[wave][li] = bsum;
    __syncthreads();
    bsum = 0.5f * ((bred[0][li] + bred[1][li]) + (bred[2][li] + bred[3][li]));   // thin_slab_out re-adds lane halves
    thin_slab_out(acc, smem, a.slab + (size_t)blockIdx.x * 96 * 32, bsum, true);
}

// precision mode 1, E1 weight gradient:  dW[m'][co] = sum_px P[px][m'] * dy[px][co],  m' = 4t + c, tap t = 5r + s,
// P[px][4t + c] = x[c][y + r - 2][px + s - 2].  The frame strip is staged as bf16 pixels of 4 channels (as in the forward), so
// the four channels of (pixel, tap) are ONE 8-byte LDS unit and a column block of P is reached by transposed LDS reads
// (ds_read_b64_tr_b16: every lane supplies the address of one (pixel, tap) unit) — no per-element gather, no conversion
// in the loop.  M-packed (round 3): the 25 taps + one all-ones unit (bias row = column sums of dy) + zero units fill
// M = 128 = 4 MFMAs per 16 pixels (the kernel-row form padded each row's 20 values to 32: 5 + 1 MFMAs, 96 accumulator
// registers instead of 64).  Wave w owns tile row w (32 pixels, 2 k-steps): 8 MFMAs per tile.
// Slab row of a workgroup: [128 m'][32 co]; m' = 100 is the bias row (101..103 repeat it, 104.. are zero).
static constexpr int E1W_ROW = 128 * 32;
// FUSE (see E1Fuse): dy of block 0 is produced HERE from x, a0 and d_a0 alone — y1 is not read, it is not even stored by
// the forward any more.  Per tile (4 rows x 32 columns):
//   1. each wave runs the forward's 75-tap conv (e1_wk_conv: the forward's operand fragments and accumulation order, so
//      the fp32 sums, hence the bf16-rounded y values and the window argmax, are the forward's to the bit) on its 2-row x
//      16-column quadrant: accumulator element v of lane (li, lh) is channel li of quadrant pixel m = (v&3) + 8(v>>2) + 4lh
//      = (row m>>4, column m&15), so the four values of a 2x2 pooling window are elements v, v+1, v+8, v+9 of ONE lane;
//   2. in the lane, on packed fp32 (two columns of a window per instruction): yb = bf16(acc + bias); first maximum of
//      fmaf(yb, scale, shift) in scan order; dy[p] = (p == argmax ? g*scale : 0) - (A + Bc*yb[p]) with g = d_a0*[a0 > 0],
//      Bc = scale*k2*invstd, A = scale*k1 - Bc*mean (the apply pass's formula with the per-channel constants folded);
//      a0 / d_a0 tiles arrive by one 16-byte load per thread, a tile ahead, and are picked up per (window, channel) from LDS;
//   3. dy never leaves the registers: the weight-gradient MFMA contracts over pixels in any order, so accumulator elements
//      8ks .. 8ks+7 of every lane ARE its B fragment of k-step ks (quadrant row ks), and the transposed reads of the A operand
//      fetch the pixels in that order (an earlier form stored dy to LDS channel-major and read it back behind a third barrier).
#ifndef E1W_OCC
#define E1W_OCC 3          // workgroups per CU of the bf16 E1 weight-gradient kernel (VGPR budget 512 / (E1W_OCC) per lane; splits = E1W_OCC * CUs)
#endif
template <int H, bool FUSE, bool XP>          // XP: the strip comes from the packed bf16 frame the forward's statistics pass left in the workspace (a.xp)
__global__ __launch_bounds__(256, E1W_OCC) void e1_wgrad_bf16_kernel(ThinWgradArgs a, E1Fuse fu) {
    using T = Tile<H>;
    static_assert(T::TW == 32 && T::TH == 4 && T::IMGS == 1, "one tile row per wave");
    constexpr int HWX = 40, HR_ = T::TH + 4, NPXH = HR_ * HWX, U_ONES = NPXH + 8, U_ZERO = NPXH + 9;
    __shared__ __attribute__((aligned(16))) bf16x4 lds_x[NPXH + 10];          // strip | 8 over-read pad units | ones | zeros
    __shared__ __attribute__((aligned(16))) __bf16 lds_d[FUSE ? 8 : 128 * 32];               // !FUSE: dy tile [px][32] (FUSE: dy never leaves registers)
    __shared__ __attribute__((aligned(16))) __bf16 lds_p[FUSE ? 2 * 32 * 32 : 8];             // FUSE: a0 | d_a0 tiles [pooled px][32]
    __shared__ float red[3 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int g = lane >> 4, h = g >> 1, qrow = (lane & 15) >> 2, cb = 16 * (g & 1) + 4 * (lane & 3);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
    if (tid < 10) {
        const __bf16 fill = (__bf16)(tid == 8 ? 1.f : 0.f);
        bf16x4 z; z[0] = z[1] = z[2] = z[3] = fill;
        lds_x[NPXH + tid] = z;
    }
    // transposed-read unit of this lane in M block b: tap t = 8b + 4(g&1) + (lane&3) -> element offset of (row r_t, column s_t)
    // relative to the lane's pixel; t = 25: the ones unit (bias row), t > 25: the zeros unit
    int moff[4];
    bool mconst[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int t = 8 * b + 4 * (g & 1) + (lane & 3);
        mconst[b] = t >= 25;
        moff[b] = t < 25 ? ((t / 5) * HWX + t % 5) * 4 : (t == 25 ? U_ONES * 4 : U_ZERO * 4);
    }
    const int t0 = blockIdx.x * a.tilesPerSplit;
    int t1 = t0 + a.tilesPerSplit; if (t1 > a.numTiles) t1 = a.numTiles;
    constexpr int XQ = (NPXH + 255) / 256;
    float rx[XQ][3];
    typedef unsigned u32x2x __attribute__((ext_vector_type(2)));
    [[maybe_unused]] u32x2x pk[XQ];
    constexpr unsigned XP_BIAS = (2 * H + 2) * 8;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_xp =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.xp)) - XP_BIAS, 0, (int)((size_t)a.B * H * H * 8 + XP_BIAS), 0x00020000);
    unsigned okm = 0u;                               // validity bit per staged strip unit
    bf16x8 rd[2];
    bf16x8 rp;                                       // FUSE: 8 channels of one pooled pixel of a0 (threads 0..127) / d_a0 (128..255)
    bf16x8 bw[7];                                    // FUSE: the forward's K-packed B fragments
    float bv = 0.f, bsc = 0.f, bsh = 0.f, bA = 0.f, bB = 0.f;         // this lane's channel li
    if constexpr (FUSE) {
        e1_wk_load(fu.w, li, lh, bw);
        bv = fu.bias[li];
        const float sc = fu.coef[li * 4], mean = fu.coef[li * 4 + 2], invstd = fu.coef[li * 4 + 3];
        bsc = sc; bsh = fu.coef[li * 4 + 1];
        bB = sc * fu.bcoef[li * 2 + 1] * invstd;
        bA = sc * fu.bcoef[li * 2] - bB * mean;
    }
    auto fetch = [&](int mt) {
        const int ib = mt / T::TILES_PER_IMG, tileInImg = mt % T::TILES_PER_IMG;
        const int ty0 = (tileInImg / T::TILES_X) * T::TH, tx0 = (tileInImg % T::TILES_X) * T::TW;
        if constexpr (XP) {                        // 8-byte units, zero padding = an out-of-range lane offset (e1_fwd_bf16_kernel)
            const unsigned soff = (unsigned)(((ib * H + ty0) * H + tx0) * 8);
#pragma unroll
            for (int i = 0; i < XQ; ++i) {
                const int q = tid + i * 256, hy = q / HWX, hx = q % HWX;
                const bool ok = q < NPXH && (unsigned)(ty0 + hy - 2) < (unsigned)H && (unsigned)(tx0 + hx - 2) < (unsigned)H;
                pk[i] = __builtin_amdgcn_raw_buffer_load_b64(rs_xp, ok ? (unsigned)(((hy - 2) * H + hx - 2) * 8 + (int)XP_BIAS) : 0x80000000u, soff, 0);
            }
        } else {
#pragma unroll
        for (int i = 0; i < XQ; ++i) {
            const int q = tid + i * 256, hy = q / HWX, hx = q % HWX;
            const int gy = ty0 + hy - 2, gx = tx0 + hx - 2;
            const bool ok = q < NPXH && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)H;
            const size_t e = ok ? ((size_t)(ib * 3) * H + gy) * H + gx : 0;
            // raw values; the zero padding is selected at staging time (see e1_fwd_bf16_kernel: a select here exposes the load latency)
            rx[i][0] = a.a0[e]; rx[i][1] = a.a0[e + (size_t)H * H]; rx[i][2] = a.a0[e + 2 * (size_t)H * H];
            okm = ok ? (okm | (1u << i)) : (okm & ~(1u << i));
        }
        }
        if constexpr (FUSE) {
            const int t = tid & 127, pp = t >> 2, c8 = t & 3;            // pooled pixel pp = prow*16 + pcol of the tile's 2 x 16
            const size_t pe = ((size_t)(ib * (H / 2) + ty0 / 2 + (pp >> 4)) * (H / 2) + tx0 / 2 + (pp & 15)) * 32 + c8 * 8;
            rp = Act<__bf16>::ld8(tid < 128 ? fu.a : fu.da, pe);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = tid + i * 256, c8 = q & 3, mm = q >> 2;
                const int gy = ty0 + mm / T::TW, gx = tx0 + mm % T::TW;
                rd[i] = Act<__bf16>::ld8(a.a1, ((size_t)(ib * H + gy) * H + gx) * 32 + c8 * 8);
            }
        }
    };
    if (t0 < t1) fetch(t0);
    const __bf16* xs = reinterpret_cast<const __bf16*>(lds_x);
    for (int mt = t0; mt < t1; ++mt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < XQ; ++i) {
            const int q = tid + i * 256;
            if constexpr (XP) {
                if (q < NPXH) lds_x[q] = __builtin_bit_cast(bf16x4, pk[i]);
            } else {
                const bool ok = (okm >> i) & 1u;
                if (q < NPXH) { bf16x4 u; u[0] = (__bf16)(ok ? rx[i][0] : 0.f); u[1] = (__bf16)(ok ? rx[i][1] : 0.f); u[2] = (__bf16)(ok ? rx[i][2] : 0.f); u[3] = (__bf16)0.f; lds_x[q] = u; }
            }
        }
        if constexpr (FUSE) {
            *reinterpret_cast<bf16x8*>(lds_p + (size_t)tid * 8) = rp;       // [a0 | d_a0][pp][32]: thread order IS the layout
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<bf16x8*>(lds_d + (size_t)(tid + i * 256) * 8) = rd[i];
        }
        __syncthreads();
        if (mt + 1 < t1) fetch(mt + 1);
        if constexpr (FUSE) {
            // 1. the forward conv of this wave's quadrant: rows 2(wave>>1) + (li>>4), columns 16(wave&1) + (li&15)
            const int qr = 2 * (wave >> 1), qc = 16 * (wave & 1);
            const f32x16 cacc = e1_wk_conv<HWX>(lds_x + (qr + (li >> 4)) * HWX + qc + (li & 15), lh, bw);
            // 2. BatchNorm / pool / ReLU backward of the lane's four windows (quadrant columns 4lh + {0,2,8,10}), channel li;
            //    the two columns of a window row travel as one packed pair
            const f32x2t bv2 = {bv, bv}, sc2 = {bsc, bsc}, sh2 = {bsh, bsh}, bA2 = {bA, bA}, bB2 = {bB, bB};
            bf16x8 dyf[2];                           // [quadrant row = k-step]: dy of accumulator elements 8ks .. 8ks+7
            // dy of the quadrant goes straight into the weight-gradient MFMA as its B operand (k = pixel, n = channel li): the
            // contraction order over pixels is free, so k-step ks takes accumulator elements 8ks .. 8ks+7 of every lane as they
            // are — lane half lh, element jj = quadrant row ks, column 4lh + (jj&3) + 8(jj>>2) — and the transposed reads of
            // the A operand below fetch the pixels in that same order.  (Round 3 first wrote dy to LDS channel-major and read
            // it back behind a third barrier.)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = 2 * j, m0 = (v & 3) + 8 * (v >> 2) + 4 * lh;          // first column of the window
                const int pp = (wave >> 1) * 16 + 8 * (wave & 1) + (m0 >> 1);
                const float av = (float)lds_p[pp * 32 + li], gv = (float)lds_p[1024 + pp * 32 + li];
                // the values the forward pooled: bf16(acc + bias), rows 0 / 1 of the window
                const f32x2t s0 = f32x2t{cacc[v], cacc[v + 1]} + bv2, s1 = f32x2t{cacc[v + 8], cacc[v + 9]} + bv2;
                const f32x2t y0 = {(float)(__bf16)s0.x, (float)(__bf16)s0.y}, y1 = {(float)(__bf16)s1.x, (float)(__bf16)s1.y};
                const f32x2t n0 = __builtin_elementwise_fma(y0, sc2, sh2), n1 = __builtin_elementwise_fma(y1, sc2, sh2);
                // first maximum in scan order (0, 1 | 2, 3): tournament with strict comparisons = the forward's sequential scan
                const bool c01 = n0.y > n0.x, c23 = n1.y > n1.x;
                const float m01 = c01 ? n0.y : n0.x, m23 = c23 ? n1.y : n1.x;
                const bool hi = m23 > m01;
                const float gs = av > 0.f ? gv * bsc : 0.f;
                const f32x2t t0v = __builtin_elementwise_fma(bB2, y0, bA2), t1v = __builtin_elementwise_fma(bB2, y1, bA2);
                const f32x2t sel0 = {(!hi && !c01) ? gs : 0.f, (!hi && c01) ? gs : 0.f};
                const f32x2t sel1 = {(hi && !c23) ? gs : 0.f, (hi && c23) ? gs : 0.f};
                const f32x2t d0 = sel0 - t0v, d1 = sel1 - t1v;
                dyf[0][v] = (__bf16)d0.x; dyf[0][v + 1] = (__bf16)d0.y;          // elements v, v+1 (row 0) and v+8, v+9 (row 1)
                dyf[1][v] = (__bf16)d1.x; dyf[1][v + 1] = (__bf16)d1.y;
            }
            // 3. weight gradient over the quadrant: k-step ks = quadrant row ks; transposed-read lane address: pixel column
            //    4h + qrow for read 0, +8 for read 1 (K index 8h + 4*read + qrow <-> column 4h + qrow + 8*read, as dyf)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const __bf16* pb = xs + ((qr + ks) * HWX + qc + 4 * h + qrow) * 4;       // this lane's pixel in the strip (tap (0,0))
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const __bf16* ap = mconst[b] ? xs + moff[b] : pb + moff[b];
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(ap, mconst[b] ? ap : ap + 32), dyf[ks], acc[b], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int px = 16 * ks + 8 * h + qrow;                       // this lane's block row (pixel) for read 0; read 1: +4
            const __bf16* dp = lds_d + (wave * 32 + px) * 32 + cb;
            const bf16x8 bvv = tr_frag(dp, dp + 4 * 32);
            const __bf16* pb = xs + (wave * HWX + px) * 4;               // this lane's pixel in the strip (tap (0,0))
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const __bf16* ap = mconst[b] ? xs + moff[b] : pb + moff[b];
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(ap, mconst[b] ? ap : ap + 16), bvv, acc[b], 0, 0, 0);
            }
        }
        }
    }
    // every wave contracted its own tile rows: fixed-order sum over the 4 waves, one accumulator at a time
    float* out = a.slab + (size_t)blockIdx.x * E1W_ROW;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) red[((wave - 1) * 16 + v) * 64 + lane] = acc[j][v];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float x = ((acc[j][v] + red[v * 64 + lane]) + red[(16 + v) * 64 + lane]) + red[(32 + v) * 64 + lane];
                const int m = (v & 3) + 8 * (v >> 2) + 4 * lh;
                out[(j * 32 + m) * 32 + li] = x;
            }
        }
    }
}

// dW1[(5r+s)*3+c][co] <- reduced slab row m' = 4(5r+s) + c; db1 <- row 100 (the ones unit)
__global__ __launch_bounds__(256) void e1_perm_kernel(const float* __restrict__ red, float* __restrict__ dw, float* __restrict__ db) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2400) {
        const int n = i & 31, k = i >> 5, c = k % 3, tap = k / 3;
        dw[i] = red[(4 * tap + c) * 32 + n];
    } else if (i < 2432 && db) {
        db[i - 2400] = red[100 * 32 + i - 2400];
    }
}

// dst (W4 [tap][ci][co]) <- reduced slab row (k = tap*3+co, col ci); the last workgroup also sums the
// per-plane dOut sums into the three bias gradients (one launch instead of two)
__global__ __launch_bounds__(256) void d4_perm_kernel(const float* __restrict__ red, float* __restrict__ dst,
                                                      const float* __restrict__ part, float* __restrict__ db, int B) {
    if (blockIdx.x == gridDim.x - 1) {
        const int co = threadIdx.x >> 6, lane = threadIdx.x & 63;      // waves 0..2 = the three output channels
        if (!part) {                        // bf16 mode: the fused backward left the sums in row 3 of the slab
            if (threadIdx.x < 3) db[threadIdx.x] = red[3 * 32 + threadIdx.x];
            return;
        }
        if (co < 3) {
            float acc = 0.f;
            for (int b = lane; b < B; b += 64) acc += part[b * 3 + co];
            acc = wave_sum(acc);
            if (lane == 0) db[co] = acc;
        }
        return;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2400) return;
    const int co = i % 3, ci = (i / 3) % 32, tap = i / 96;
    if (!part) {                            // bf16 mode: red = dV[i][j][co][ci] of the phase-collapsed form; tap (r, s) is reached from
        const int r = tap / 5, s5 = tap % 5;        // the window positions i in {4-r, 5-r}, j in {4-s, 5-s} (d4_bwd_bf16_kernel)
        float acc = 0.f;
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) acc += red[(((4 - r + di) * 8 + (4 - s5 + dj)) * 4 + co) * 32 + ci];
        dst[i] = acc;
        return;
    }
    dst[i] = red[(tap * 3 + co) * 32 + ci];
}

// Split counts of the E1 weight-gradient / D4 backward kernels.  THIN_SPLIT_CAP is the ONE bound shared by the workspace
// sizing (e1_wgrad_ws_floats / d4_bwd_ws_floats) and the launches: whatever occupancy switch (-DE1W_OCC, -DD4B_OCC) or
// device CU count asks for, a launch never writes more slab rows than the workspace was sized for.
static constexpr int THIN_SPLIT_CAP = 1024;
static int thin_splits(int numTiles, int* tps, int want = 512) {
    if (want > THIN_SPLIT_CAP) want = THIN_SPLIT_CAP;
    int S = numTiles < want ? numTiles : want;
    *tps = cdiv(numTiles, S);
    return cdiv(numTiles, *tps);
}

int64_t e1_wgrad_ws_floats(int width, int B) {
    int tps; const int tiles = B * (width / 4) * (width / 32);
    const int64_t S = thin_splits(tiles, &tps, THIN_SPLIT_CAP);   // upper bound of the split counts used below
    const int64_t f32 = S * 3072 + col_reduce_ws_floats(3072);
    const int64_t b16 = S * E1W_ROW + E1W_ROW + 32 + col_reduce_ws_floats(E1W_ROW);     // slabs | reduced row | column-reduce scratch
    return f32 > b16 ? f32 : b16;
}

int launch_e1_fwd(int width, int B, const float* x, const float* w, const float* bias, float* y,
                  float* bnpart, hipStream_t st, bool bf16, int pass, const float* coef, float* a1, bool keep_y, float* xpf) {
    bf16x4* xp = reinterpret_cast<bf16x4*>(xpf);          // packed bf16 frame (workspace): written by pass 1, read by pass 2; null: pass 2 is not available
    const int keepY = keep_y ? 1 : 0;
    if (pass != 0 && !bf16) { cvae_set_error("e1_fwd: passes 1/2 exist in bf16 mode only"); return -2; }
    if (pass < 0 || pass > 2) { cvae_set_error("e1_fwd: pass %d", pass); return -2; }
    if (pass == 2 && !xp) pass = 3;                       // no packed frame (eval mode: no statistics pass ran): the pool pass stages from the fp32 frame
    const int ns64 = B * 8, ns128 = B * 32, cap = cvae_num_cus() * 3;      // persistent: 3 workgroups per CU (<= 168 VGPRs), one strip each per turn
    const dim3 g64(ns64 < cap ? ns64 : cap), g128(ns128 < cap ? ns128 : cap);
    cvae_probe_begin(st);
    if (width == 64 && bf16 && pass == 1) hipLaunchKernelGGL((e1_fwd_bf16_kernel<64, E1_STATS>), g64, dim3(256), 0, st, x, w, bias, y, bnpart, B, coef, a1, ns64, keepY, xp);
    else if (width == 64 && bf16 && pass == 2) hipLaunchKernelGGL((e1_fwd_bf16_kernel<64, E1_POOL>), g64, dim3(256), 0, st, x, w, bias, y, bnpart, B, coef, a1, ns64, keepY, xp);
    else if (width == 128 && bf16 && pass == 1) hipLaunchKernelGGL((e1_fwd_bf16_kernel<128, E1_STATS>), g128, dim3(256), 0, st, x, w, bias, y, bnpart, B, coef, a1, ns128, keepY, xp);
    else if (width == 128 && bf16 && pass == 2) hipLaunchKernelGGL((e1_fwd_bf16_kernel<128, E1_POOL>), g128, dim3(256), 0, st, x, w, bias, y, bnpart, B, coef, a1, ns128, keepY, xp);
    else if (width == 64 && bf16 && pass == 3) hipLaunchKernelGGL((e1_fwd_bf16_kernel<64, E1_POOL_X>), g64, dim3(256), 0, st, x, w, bias, y, bnpart, B, coef, a1, ns64, keepY, xp);
    else if (width == 128 && bf16 && pass == 3) hipLaunchKernelGGL((e1_fwd_bf16_kernel<128, E1_POOL_X>), g128, dim3(256), 0, st, x, w, bias, y, bnpart, B, coef, a1, ns128, keepY, xp);
    else if (width == 64 && bf16) hipLaunchKernelGGL((e1_fwd_bf16_kernel<64, E1_Y>), g64, dim3(256), 0, st, x, w, bias, y, bnpart, B, coef, a1, ns64, keepY, xp);
    else if (width == 128 && bf16) hipLaunchKernelGGL((e1_fwd_bf16_kernel<128, E1_Y>), g128, dim3(256), 0, st, x, w, bias, y, bnpart, B, coef, a1, ns128, keepY, xp);
    else if (width == 64) hipLaunchKernelGGL(e1_fwd_kernel<64>, dim3(B * 8), dim3(256), 0, st, x, w, bias, y, bnpart, B);
    else if (width == 128) hipLaunchKernelGGL(e1_fwd_kernel<128>, dim3(B * 32), dim3(256), 0, st, x, w, bias, y, bnpart, B);
    else { cvae_set_error("e1_fwd: width %d unsupported", width); return -2; }
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    return 0;
}

int launch_e1_wgrad(int width, int B, const float* x, const float* dy, float* dw, float* dbias, float* ws, hipStream_t st, bool bf16,
                    const float* const* fuse, const float* xp) {
    if (width != 64 && width != 128) { cvae_set_error("e1_wgrad: width %d unsupported", width); return -2; }
    int tps; const int tiles = B * (width / 4) * (width / 32);
    const int S = thin_splits(tiles, &tps, bf16 ? E1W_OCC * cvae_num_cus() : 512);
    ThinWgradArgs a{x, nullptr, dy, nullptr, nullptr, ws, B, tiles, tps, xp};
    // fuse = {y0, a0, d_a0, coef0, bcoef0, w1, b1}: block 0's BatchNorm/pool/ReLU backward is applied while staging (no dy
    // tensor); the bf16 kernel recomputes y0 from x, w1, b1 and never reads fuse[0]
    const E1Fuse fu = fuse ? E1Fuse{fuse[0], fuse[1], fuse[2], fuse[3], fuse[4], fuse[5], fuse[6]} : E1Fuse{};
    cvae_probe_begin(st);
    if (bf16) {          // precision mode 1: transposed-read kernel, its own slab layout + a permuting finish
        if (width == 64 && fuse && xp) hipLaunchKernelGGL((e1_wgrad_bf16_kernel<64, true, true>), dim3(S), dim3(256), 0, st, a, fu);
        else if (width == 64 && fuse) hipLaunchKernelGGL((e1_wgrad_bf16_kernel<64, true, false>), dim3(S), dim3(256), 0, st, a, fu);
        else if (width == 64) hipLaunchKernelGGL((e1_wgrad_bf16_kernel<64, false, false>), dim3(S), dim3(256), 0, st, a, fu);
        else if (fuse && xp) hipLaunchKernelGGL((e1_wgrad_bf16_kernel<128, true, true>), dim3(S), dim3(256), 0, st, a, fu);
        else if (fuse) hipLaunchKernelGGL((e1_wgrad_bf16_kernel<128, true, false>), dim3(S), dim3(256), 0, st, a, fu);
        else hipLaunchKernelGGL((e1_wgrad_bf16_kernel<128, false, false>), dim3(S), dim3(256), 0, st, a, fu);
        cvae_probe_end(st);
        CVAE_CHECK_LAUNCH();
        st = cvae_reduce_stream(st);
        float* red = ws + (size_t)S * E1W_ROW;
        { int rc = launch_col_reduce(ws, S, E1W_ROW, E1W_ROW, red, red + E1W_ROW + 32, st); if (rc) return rc; }
        hipLaunchKernelGGL(e1_perm_kernel, dim3(10), dim3(256), 0, st, red, dw, dbias);
        CVAE_CHECK_LAUNCH();
        return 0;
    }
    if (width == 64 && fuse) hipLaunchKernelGGL((e1_wgrad_kernel<64, true>), dim3(S), dim3(256), 0, st, a, fu);
    else if (width == 64) hipLaunchKernelGGL((e1_wgrad_kernel<64, false>), dim3(S), dim3(256), 0, st, a, fu);
    else if (fuse) hipLaunchKernelGGL((e1_wgrad_kernel<128, true>), dim3(S), dim3(256), 0, st, a, fu);
    else hipLaunchKernelGGL((e1_wgrad_kernel<128, false>), dim3(S), dim3(256), 0, st, a, fu);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    st = cvae_reduce_stream(st);
    // slab row = [75 x 32 weights | 32 zeros (K pad) | 32 bias partials]: in the flat buffer enc0.b sits at
    // enc0.w + 2432 (2400 weights padded to 64 floats), so ONE column reduction fills both
    if (dbias == dw + 2432)
        return launch_col_reduce(ws, S, 2464, 3072, dw, ws + (size_t)S * 3072, st);
    int rc = launch_col_reduce(ws, S, 2400, 3072, dw, ws + (size_t)S * 3072, st);
    if (rc || !dbias) return rc;
    return launch_col_reduce(ws + 2432, S, 32, 3072, dbias, ws + (size_t)S * 3072, st);
}

// ------------------------------------------ D4 ------------------------------------------------

// tanh through one exp: |error| ~1e-7 absolute (the reference's tanh differs from this by less than
// the fp32 rounding of its own input sum)
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f / (__expf(2.0f * x) + 1.0f); }

// fp32 mode (round 3): the phase-collapsed form (see the bf16 kernel below for the algebra) on the exact fp32 MFMA
// (v_mfma_f32_16x16x4_f32: N = 16 columns for the 12 real ones, K = 9 taps x 32 channels = 72 MFMAs per 16 source pixels, A = one
// float per lane read in place from the staged 10 x 18 halo tile, the 72 collapsed-weight values of a lane in registers, four
// independent accumulator chains per wave).  Rounds 1-2 computed Q[source][tap, co] = o3 . W on 32x32x2 MFMAs (28 of 128 rows and 21
// of 96 columns padding) and had every output pixel gather its 25 taps from LDS (75 reads): 48 -> 39 us at B = 256.  Sums of up to
// four 5x5 taps are formed in fp32 before the contraction (as D1..D3 do, conv_up.hip): the summation order differs from the direct
// form at the 1e-7 level.  Persistent, XCD-aware tile order, next tile prefetched into registers.
template <int H>
__global__ __launch_bounds__(256, 2) void d4_fwd_pc_f32_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ recon, int B) {
    constexpr int HS = H / 2, TXN = HS / 16, TPI = (HS / 8) * TXN;
    constexpr int HW = 18, HP = 10 * HW, AS = 36;               // halo 10 x 18 source pixels, row stride 36 floats: 16-byte rows, and
    constexpr int OS = 34;                                       // the (pixel, k) reads of a wave hit 64 different banks
    __shared__ __attribute__((aligned(16))) float lds_a[HP * AS];
    __shared__ __attribute__((aligned(16))) float lds_o[3 * 16 * OS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lg = lane >> 4;
    // collapsed weights: B value of k-step (tap t, j): k = 4j + lg (channel), column n = lc = p*3 + co (n >= 12: zero)
    float bw[72];
    {
        static_assert(9 * 32 * 16 <= HP * AS, "collapsed weights fit in the halo buffer");
        // the 2400 weights travel to LDS first (coalesced, all loads in flight at once): summing them straight from global
        // memory was a chain of up to 72 dependent scattered loads per thread — most of the kernel's time at B = 256
        for (int q = tid; q < 2400; q += 256) lds_a[q] = w[q];
        __syncthreads();
        float cw[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int q = tid + i * 256, n = q & 15, ci = (q >> 4) & 31, t = q >> 9;
            const int pz = n / 3, co = n % 3, py = pz >> 1, px = pz & 1, ta = t / 3, tb = t % 3;
            const int r0 = py ? (ta == 0 ? 0 : 2 * ta - 1) : 2 * ta, r1 = py ? 2 * ta : (ta == 2 ? 4 : 2 * ta + 1);
            const int s0 = px ? (tb == 0 ? 0 : 2 * tb - 1) : 2 * tb, s1 = px ? 2 * tb : (tb == 2 ? 4 : 2 * tb + 1);
            float acc = 0.f;
            if (n < 12)
                for (int r = r0; r <= r1; ++r)
                    for (int s5 = s0; s5 <= s1; ++s5) acc += lds_a[((r * 5 + s5) * 32 + ci) * 3 + co];
            cw[i] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 18; ++i) { const int q = tid + i * 256; lds_a[((q >> 9) * 32 + ((q >> 4) & 31)) * 16 + (q & 15)] = cw[i]; }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) bw[t * 8 + j] = lds_a[(t * 32 + 4 * j + lg) * 16 + lc];
        __syncthreads();
    }
    const float bv = lc < 12 ? bias[lc % 3] : 0.f;
    const int numTiles = B * TPI;
    constexpr int NQ = HP * 8, IPT = (NQ + 255) / 256;          // 16-byte units of the halo tile
    f32x4 ra[IPT];
    unsigned okm = 0u;                          // validity bit per staged unit (see e1_fwd_bf16_kernel)
    auto fetch = [&](int tile) {
        const int ib = tile / TPI, t = tile % TPI;
        const int sy0 = (t / TXN) * 8 - 1, sx0 = (t % TXN) * 16 - 1;
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256, c4 = q & 7, sp = q >> 3;
            const int sy = sy0 + sp / HW, sx = sx0 + sp % HW;
            const bool ok = sp < HP && (unsigned)sy < (unsigned)HS && (unsigned)sx < (unsigned)HS;
            ra[i] = *reinterpret_cast<const f32x4*>(in + (ok ? ((size_t)(ib * HS + sy) * HS + sx) * 32 + c4 * 4 : 0));      // raw; zero select at staging
            okm = ok ? (okm | (1u << i)) : (okm & ~(1u << i));
        }
    };
    const int G = gridDim.x;
    auto tile_of = [&](int n) {          // XCD-aware order, as the bf16 kernel
        if ((G & 7) || (numTiles & 7)) return n;
        const int b = n % G, k = n / G;
        return (b & 7) * (numTiles >> 3) + k * (G >> 3) + (b >> 3);
    };
    if ((int)blockIdx.x < numTiles) fetch(tile_of(blockIdx.x));
    for (int n = blockIdx.x; n < numTiles; n += G) {
        const int tile = tile_of(n);
        const int ib = tile / TPI, t = tile % TPI;
        const int ty0 = (t / TXN) * 16, tx0 = (t % TXN) * 32;
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const int q = tid + i * 256;
            if (q < NQ) *reinterpret_cast<f32x4*>(lds_a + (q >> 3) * AS + (q & 7) * 4) = ((okm >> i) & 1u) ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();       // also: every thread is past the previous tile's output rows, lds_o is free
        if (n + G < numTiles) fetch(tile_of(n + G));
        f32x4 acc[2], acc2[2];               // two source rows x two chains (even / odd k-steps): four independent MFMA chains
#pragma unroll
        for (int g = 0; g < 2; ++g) { acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const float* ap0 = lds_a + ((2 * wave) * HW + lc) * AS + lg;             // source row 2w, column lc; + tap and 4j; row 2w+1: + HW*AS
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; j += 2)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float* ap = ap0 + g * HW * AS + ((t / 3) * HW + t % 3) * AS + 4 * j;
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[0], bw[t * 8 + j], acc[g], 0, 0, 0);
                    acc2[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4], bw[t * 8 + j + 1], acc2[g], 0, 0, 0);
                }
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[g] += acc2[g];
        if (lc < 12) {
            const int pz = lc / 3, co = lc % 3;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    lds_o[(co * 16 + 2 * (2 * wave + g) + (pz >> 1)) * OS + 2 * (4 * lg + i) + (pz & 1)] = fast_tanh(acc[g][i] + bv);
        }
        __syncthreads();
        {
            const int oy = tid >> 4, ox = (tid & 15) * 2;
#pragma unroll
            for (int co = 0; co < 3; ++co)
                *reinterpret_cast<float2*>(recon + ((size_t)(ib * 3 + co) * H + ty0 + oy) * H + tx0 + ox) =
                    *reinterpret_cast<const float2*>(lds_o + (co * 16 + oy) * OS + ox);
        }
    }
}

// precision mode 1 (round 3): Upsample(2) -> Conv5x5 -> Tanh as the PHASE-COLLAPSED 3x3 conv of conv_up.hip over the stored
// low-resolution o3, straight on the MFMA.  Output pixel (2y+py, 2x+px) reads source pixel (y+a-1, x+b-1) through the 5x5
// taps r with floor((py+r-2)/2) = a-1 (and s likewise), so with wc[p][a][b] = the sum of those taps' weights
//     out[2y+py][2x+px][co] = bias[co] + sum_{a,b < 3} sum_ci o3[y+a-1][x+b-1][ci] * wc[p][a][b][ci][co]
// = a GEMM with M = source pixels, N = 4 phases x 3 channels = 12 (one 16-column v_mfma_f32_16x16x32_bf16 block: 75 % of
// the accumulator lanes carry an output, against 37 % for a 32-column block) and K = 9 taps x 32 channels: 9 MFMAs per
// 16 source pixels, their A fragments read in place from the staged halo tile (a tap is an address offset), the collapsed
// weights in registers for the whole launch.  The round-2 kernel computed Q[source][tap, co] for the 10x10 window and had
// every output pixel gather its 25 taps from LDS: 48 ds_write_b32 + 75 ds_read_b32 per thread and tile, which bound it
// (134 us at B = 2048 for 235 MB); here a wave issues 18 ds_read_b128 + 8 ds_write_b32 per 32 source pixels.
// Workgroup = 8 x 16 source pixels (16 x 32 outputs), wave w = source rows 2w, 2w+1; Tanh in the accumulator layout, the
// NCHW fp32 rows leave through an LDS transpose as 8-byte stores.
template <int H>
__global__ __launch_bounds__(256, 4) void d4_fwd_bf16_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ recon, int B) {
    constexpr int HS = H / 2, TXN = HS / 16, TPI = (HS / 8) * TXN;
    constexpr int HW = 18, HP = 10 * HW, PSP = 180;            // halo tile 10 x 18 source pixels; plane stride == 4 (mod 16) units:
    static_assert(PSP >= HP && PSP % 16 == 4, "plane stride");  // the 4 octets x 4 pixels of a store phase hit 16 different 16-byte slots
    constexpr int OS = 34;                                       // row stride (floats) of the output tile [3][16][OS]
    __shared__ __attribute__((aligned(16))) bf16x8 lds_a[4 * PSP];      // [channel octet][halo pixel]
    __shared__ __attribute__((aligned(16))) float lds_o[3 * 16 * OS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, lg = lane >> 4;
    // collapsed weights as B fragments: column n = p*3 + co (n >= 12: zero), k = ci; built once per workgroup in LDS (the
    // halo buffer, before the first tile), then this lane's 9 fragments (tap t = a*3 + b, k = 8*lg + j) stay in registers
    bf16x8 bw[9];
    {
        __bf16* wl = reinterpret_cast<__bf16*>(lds_a);           // [t][n][lg][8]
        static_assert(9 * 16 * 4 <= 4 * PSP && 2400 * 4 <= 4 * PSP * 16, "weight fragments / raw weights fit in the halo buffer");
        // the 2400 weights travel to LDS first (coalesced, all loads in flight at once): summing them straight from global
        // memory was a chain of dependent scattered loads per thread
        float* wraw = reinterpret_cast<float*>(lds_a);
        for (int q = tid; q < 2400; q += 256) wraw[q] = w[q];
        __syncthreads();
        float cw[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int q = tid + i * 256, ci = q & 31, n = (q >> 5) & 15, t = q >> 9;
            const int pz = n / 3, co = n % 3, py = pz >> 1, px = pz & 1, ta = t / 3, tb = t % 3;
            // 5x5 taps that reach source row offset ta-1 from output phase py: py = 0: {0,1},{2,3},{4}; py = 1: {0},{1,2},{3,4}
            const int r0 = py ? (ta == 0 ? 0 : 2 * ta - 1) : 2 * ta, r1 = py ? 2 * ta : (ta == 2 ? 4 : 2 * ta + 1);
            const int s0 = px ? (tb == 0 ? 0 : 2 * tb - 1) : 2 * tb, s1 = px ? 2 * tb : (tb == 2 ? 4 : 2 * tb + 1);
            float acc = 0.f;
            if (n < 12)
                for (int r = r0; r <= r1; ++r)
                    for (int s5 = s0; s5 <= s1; ++s5) acc += wraw[((r * 5 + s5) * 32 + ci) * 3 + co];
            cw[i] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int q = tid + i * 256, ci = q & 31, n = (q >> 5) & 15, t = q >> 9;
            wl[((t * 16 + n) * 4 + (ci >> 3)) * 8 + (ci & 7)] = (__bf16)cw[i];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) bw[t] = lds_a[(t * 16 + lc) * 4 + lg];
        __syncthreads();
    }
    const float bv = lc < 12 ? bias[lc % 3] : 0.f;
    const int numTiles = B * TPI;
    bf16x8 zero8;
#pragma unroll
    for (int c = 0; c < 8; ++c) zero8[c] = (__bf16)0.f;
    bf16x8 ra[3];
    auto fetch = [&](int tile) {
        const int ib = tile / TPI, t = tile % TPI;
        const int sy0 = (t / TXN) * 8 - 1, sx0 = (t % TXN) * 16 - 1;           // 10 x 18 source window
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = tid + i * 256, c8 = q & 3, sp = q >> 2;
            const int sy = sy0 + sp / HW, sx = sx0 + sp % HW;
            const bool ok = sp < HP && (unsigned)sy < (unsigned)HS && (unsigned)sx < (unsigned)HS;
            const bf16x8 l = Act<__bf16>::ld8(in, ok ? ((size_t)(ib * HS + sy) * HS + sx) * 32 + c8 * 8 : 0);
            ra[i] = ok ? l : zero8;
        }
    };
    // XCD-aware tile order (common.h, xcd_tile): workgroups b, b+8, ... share an XCD, so each XCD walks ONE contiguous
    // eighth of the tiles — the source windows of neighbouring tiles overlap by two rows / columns, and the overlap is
    // then re-read from that XCD's L2 instead of from HBM by another XCD
    const int G = gridDim.x;
    auto tile_of = [&](int n) {
        if ((G & 7) || (numTiles & 7)) return n;
        const int b = n % G, k = n / G;
        return (b & 7) * (numTiles >> 3) + k * (G >> 3) + (b >> 3);
    };
    if ((int)blockIdx.x < numTiles) fetch(tile_of(blockIdx.x));
    for (int n = blockIdx.x; n < numTiles; n += G) {
        const int tile = tile_of(n);
        const int ib = tile / TPI, t = tile % TPI;
        const int ty0 = (t / TXN) * 16, tx0 = (t % TXN) * 32;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = tid + i * 256;
            if (q < HP * 4) lds_a[(q & 3) * PSP + (q >> 2)] = ra[i];
        }
        __syncthreads();       // also: every thread is past the previous tile's output rows, lds_o is free
        if (n + G < numTiles) fetch(tile_of(n + G));
        f32x4 acc[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bf16x8* ap = lds_a + lg * PSP + (2 * wave + g) * HW + lc;      // source row 2w+g, column lc of the tile: halo (row, col) + tap
#pragma unroll
            for (int t = 0; t < 9; ++t)
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[(t / 3) * HW + t % 3], bw[t], acc[g], 0, 0, 0);
        }
        if (lc < 12) {
            const int pz = lc / 3, co = lc % 3;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i)       // accumulator row 4*lg + i = source column of row 2w+g
                    lds_o[(co * 16 + 2 * (2 * wave + g) + (pz >> 1)) * OS + 2 * (4 * lg + i) + (pz & 1)] = fast_tanh(acc[g][i] + bv);
        }
        __syncthreads();
        {
            const int oy = tid >> 4, ox = (tid & 15) * 2;
#pragma unroll
            for (int co = 0; co < 3; ++co)
                *reinterpret_cast<float2*>(recon + ((size_t)(ib * 3 + co) * H + ty0 + oy) * H + tx0 + ox) =
                    *reinterpret_cast<const float2*>(lds_o + (co * 16 + oy) * OS + ox);
        }
    }
}

// dOut = d_recon * (1 - recon^2) (Tanh backward, vae_nets.py:134) + per-plane sums for db
__global__ __launch_bounds__(256) void d4_actbwd_kernel(const float* __restrict__ d_recon, const float* __restrict__ recon,
                                                        float* __restrict__ dout, float* __restrict__ part, int hw) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * hw;
    float acc = 0.f;
    for (int i = threadIdx.x * 4; i < hw; i += 1024) {
        const float4 g = *reinterpret_cast<const float4*>(d_recon + base + i);
        const float4 r = *reinterpret_cast<const float4*>(recon + base + i);
        float4 o;
        o.x = g.x * (1.f - r.x * r.x); o.y = g.y * (1.f - r.y * r.y);
        o.z = g.z * (1.f - r.z * r.z); o.w = g.w * (1.f - r.w * r.w);
        *reinterpret_cast<float4*>(dout + base + i) = o;
        acc += (o.x + o.y) + (o.z + o.w);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// tile mt of the D4 backward into registers: dOut halo planes [3][20][36] (zero padded) and the o3 tile
template <int H, typename AT>
__device__ __forceinline__ void d4_bwd_fetch(const ThinWgradArgs& a, int mt, float (&rg)[(3 * 720 + 255) / 256], f32x4 (&ro)[4], unsigned& okm) {
    constexpr int HS = H / 2, TPI = (HS / 8) * (HS / 16), G0 = 3 * 720;
    const int tid = threadIdx.x;
    const int ib = mt / TPI, t = mt % TPI;
    const int sy0 = (t / (HS / 16)) * 8, sx0 = (t % (HS / 16)) * 16;
#pragma unroll
    for (int i = 0; i < (G0 + 255) / 256; ++i) {
        const int q = tid + i * 256, c = q / 720, rem = q % 720;
        const int uy = 2 * sy0 - 2 + rem / 36, ux = 2 * sx0 - 2 + rem % 36;
        const bool ok = q < G0 && (unsigned)uy < (unsigned)H && (unsigned)ux < (unsigned)H;
        rg[i] = a.a0[ok ? ((size_t)(ib * 3 + c) * H + uy) * H + ux : 0];       // raw; the zero padding is selected at staging (see e1_wgrad_fetch)
        okm = ok ? (okm | (1u << i)) : (okm & ~(1u << i));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + i * 256, c4 = q & 7, sp = q >> 3;
        ro[i] = Act<AT>::ld4(a.a1, ((size_t)(ib * HS + sy0 + sp / 16) * HS + sx0 + sp % 16) * 32 + c4 * 4);
    }
}

template <int H, typename AT>   // H = output size (64); src tiles of 8 rows x 16 cols at HS = H/2; o3 / d_o3 stored as AT
__global__ __launch_bounds__(256, THIN_F32_OCC) void d4_bwd_kernel(ThinWgradArgs a) {
    constexpr int HS = H / 2, TPI = (HS / 8) * (HS / 16);
    constexpr int G0 = 3 * 720, O = 128 * 32, G = 128 * 77 + 32, WR = 76 * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds_g0 = smem;              // dOut halo planes [3][20][36]
    float* lds_o = smem + G0;          // o3 tile [128][32]
    float* lds_G = lds_o + O;          // [128][77] (+32 pad)
    float* lds_wr = lds_G + G;         // W4r[k=tap*3+co][ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int q = tid; q < WR; q += 256) {
        const int k = q / 32, ci = q % 32;
        lds_wr[q] = k < 75 ? a.w[((k / 3) * 32 + ci) * 3 + k % 3] : 0.f;
    }
    for (int q = tid; q < 32; q += 256) lds_G[128 * 77 + q] = 0.f;
    if (tid < 128) lds_G[tid * 77 + 75] = 0.f;               // K = 75 padded to 76: the pad column stays zero
    f32x16 accw[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb)
#pragma unroll
        for (int v = 0; v < 16; ++v) accw[mb][v] = 0.f;
    const int t0 = blockIdx.x * a.tilesPerSplit;
    int t1 = t0 + a.tilesPerSplit; if (t1 > a.numTiles) t1 = a.numTiles;
    // software pipeline: the next tile's dOut halo and o3 tile travel through registers while this
    // tile's G build and MFMAs run
    constexpr int GQ = (G0 + 255) / 256;
    float rg[GQ];
    unsigned okm = 0u;
    f32x4 ro[4];
    if (t0 < t1) d4_bwd_fetch<H, AT>(a, t0, rg, ro, okm);
    const int gsp = tid & 127, ghalf = tid >> 7, gsy = gsp >> 4, gsx = gsp & 15;
    for (int mt = t0; mt < t1; ++mt) {
        const int ib = mt / TPI, t = mt % TPI;
        const int sy0 = (t / (HS / 16)) * 8, sx0 = (t % (HS / 16)) * 16;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < GQ; ++i) {
            const int q = tid + i * 256;
            if (q < G0) lds_g0[q] = ((okm >> i) & 1u) ? rg[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = tid + i * 256;
            *reinterpret_cast<f32x4*>(lds_o + (q >> 3) * 32 + (q & 7) * 4) = ro[i];
        }
        __syncthreads();
        if (mt + 1 < t1) d4_bwd_fetch<H, AT>(a, mt + 1, rg, ro, okm);
        __builtin_amdgcn_sched_barrier(0);            // keep the loads HERE (the scheduler sank them below the MFMAs: nothing was prefetched)
        // G[src][(r*5+s)*3+co] = sum of dOut over the 2x2 block of src shifted by the tap: thread =
        // (src pixel, half of the 15 (co, r) pairs); the five s taps of a pair share six column sums
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pr = ghalf * 8 + i;                     // wave-uniform
            if (pr < 15) {
                const int co = pr / 5, r = pr % 5;
                const float* p = lds_g0 + co * 720 + (2 * gsy - r + 4) * 36 + 2 * gsx;
                float cs[6];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float2 u = *reinterpret_cast<const float2*>(p + 2 * j);
                    const float2 d = *reinterpret_cast<const float2*>(p + 36 + 2 * j);
                    cs[2 * j] = u.x + d.x; cs[2 * j + 1] = u.y + d.y;
                }
                float* g = lds_G + gsp * 77 + r * 15 + co;
#pragma unroll
                for (int sI = 0; sI < 5; ++sI) g[sI * 3] = cs[4 - sI] + cs[5 - sI];
            }
        }
        __syncthreads();
        // dgrad: d_o3[src][ci] = relu'(o3) * sum_k G[src][k] * W4r[k][ci]
        f32x16 accd;
#pragma unroll
        for (int v = 0; v < 16; ++v) accd[v] = 0.f;
#pragma unroll
        for (int j = 0; j < 38; ++j)
            accd = __builtin_amdgcn_mfma_f32_32x32x2f32(lds_G[(wave * 32 + li) * 77 + 2 * j + lh],
                                                        lds_wr[(2 * j + lh) * 32 + li], accd, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int sp = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            const float x = lds_o[sp * 32 + li] > 0.f ? accd[v] : 0.f;
            Act<AT>::st(a.din, ((size_t)(ib * HS + sy0 + sp / 16) * HS + sx0 + sp % 16) * 32 + li, x);
        }
        // wgrad: dW[k][ci] += sum_src G[src][k] * o3[src][ci]
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int sp = wave * 32 + 2 * kk + lh;
            const float bv = lds_o[sp * 32 + li];
#pragma unroll
            for (int mb = 0; mb < 3; ++mb)
                accw[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(lds_G[sp * 77 + mb * 32 + li], bv, accw[mb], 0, 0, 0);
        }
    }
    thin_slab_out(accw, smem, a.slab + (size_t)blockIdx.x * 96 * 32);
}

// precision mode 1 (round 3): the fused D4 backward on the bf16 MFMA in the phase-collapsed form of the forward.  Source pixel
// (y, x) is read by the 6 x 6 output pixels (2y-2+i, 2x-2+j), i, j < 6, output (i, j) reaching it through the 5x5 taps
// r in {4-i, 5-i}, s in {4-j, 5-j} (those inside 0..4), so with V[i][j] = the sum of those taps' weights
//     d_o3[y][x][ci]   = relu'(o3) * sum_{i,j,co} dOut[2y-2+i][2x-2+j][co] * V[i][j][ci][co]
//     dV[i][j][ci][co] = sum_{y,x} dOut[2y-2+i][2x-2+j][co] * o3[y][x][ci],   dW[r][s] = sum_{i in {4-r,5-r}, j in {4-s,5-s}} dV[i][j]
// The A-side operand of BOTH contractions is dOut itself: staged once per tile as a bf16 image [row][column][4] (channel 3 =
// 0), a source pixel's 6 x 6 window is 6 runs of 8 columns x 4 channels = 32 contiguous elements (columns 6, 7 of a run meet
// zero weights), i.e. K = 6 x 32 = 192 read in place — 16-byte reads for the input gradient, transposed 8-byte reads
// (ds_read_b64_tr_b16, every lane supplies its own source pixel's address) for the weight gradient.  The round-2 kernel built
// G[source][tap, co] (sums of up to 2 x 2 outputs) in LDS first: ~100 LDS operations per thread and tile, which bound it
// (175 us at B = 2048 for 485 MB); here the tile costs a wave 12 + 12 MFMAs, 12 ds_read_b128 and 32 transposed reads.
//   input gradient : C[ci][source pixel] = V^T . dOut^T, 12 k-steps (weights in registers as A fragments, the wave's 32
//                    source pixels as columns): lane = pixel, registers = 4 consecutive channels x 4 -> 8-byte bf16 stores;
//   weight gradient: dV[k][ci] over K = source pixels; wave w takes k-blocks {w>>1, (w>>1)+2, (w>>1)+4} over the tile half w&1
//                    (3 accumulators per wave instead of 6); halves are added through LDS at the end.
// Slab row of a workgroup: [192 k][32 ci], k = i*32 + j*4 + co; row 3 (a zero row: co = 3) carries the bias-gradient partials.
#ifndef D4B_OCC
#define D4B_OCC 2          // workgroups per CU of the bf16 D4 backward (splits = D4B_OCC * CUs)
#endif
#ifndef D4B_VWLDS
#define D4B_VWLDS 0
#endif
#ifndef D4B_ACC2
#define D4B_ACC2 1          // input gradient as two chains of 6 MFMAs (119 -> 116 us); D4B_VWLDS=1 (weights re-read from LDS, 3 workgroups per CU): 134-142 us
#endif
static constexpr int D4P_ROW = 192 * 32;
static constexpr int D4P_IW = 40;                 // image row stride in pixels (36 staged + 4 zero columns for the 8-wide runs)
static constexpr int D4_BWD_BF16_SMEM_BYTES = 2 * 3 * 16 * 64 * 4;      // dynamic LDS of d4_bwd_bf16_kernel (see D4_BWD_BF16_SMEM)
template <int H>
__global__ __launch_bounds__(256, D4B_OCC) void d4_bwd_bf16_kernel(ThinWgradArgs a) {
    constexpr int HS = H / 2, TPI = (HS / 8) * (HS / 16);
    constexpr int IMG = 20 * D4P_IW * 4;                                        // elements of the dOut image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* img = reinterpret_cast<__bf16*>(smem_raw);                          // [20 rows][40 columns][4]
    __bf16* lds_o = img + IMG;                                                  // o3 tile [128][32]
#if D4B_VWLDS
    __shared__ __attribute__((aligned(16))) __bf16 lds_v[12 * 64 * 8];          // the input gradient's weight fragments [k-step][lane]
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    // collapsed weights as A fragments of the input gradient: row ci = li, k-step m covers k = 16m + 8lh + jj
    bf16x8 vw[D4B_VWLDS ? 1 : 12];
    {
        static_assert(12 * 32 * 2 * 8 * 2 + 2400 * 4 <= D4_BWD_BF16_SMEM_BYTES, "weight fragments + raw weights fit in the dynamic LDS");
        // the 2400 weights travel to LDS first (coalesced, all loads in flight at once; behind the fragment table): summing
        // them straight from global memory was a chain of dependent scattered loads per thread
        float* wraw = reinterpret_cast<float*>(smem_raw + 12 * 32 * 2 * 8 * 2);
        for (int q = tid; q < 2400; q += 256) wraw[q] = a.w[q];
        __syncthreads();
        for (int q = tid; q < 192 * 32; q += 256) {
            const int ci = q & 31, k = q >> 5, co = k & 3, j = (k >> 2) & 7, i = k >> 5;
            float acc = 0.f;
            if (co < 3 && j < 6)
                for (int r = (4 - i < 0 ? 0 : 4 - i); r <= (5 - i > 4 ? 4 : 5 - i); ++r)
                    for (int s5 = (4 - j < 0 ? 0 : 4 - j); s5 <= (5 - j > 4 ? 4 : 5 - j); ++s5) acc += wraw[((r * 5 + s5) * 32 + ci) * 3 + co];
            img[(((k >> 4) * 32 + ci) * 2 + ((k >> 3) & 1)) * 8 + (k & 7)] = (__bf16)acc;
        }
        __syncthreads();
#if D4B_VWLDS
        for (int q = tid; q < 12 * 64; q += 256) reinterpret_cast<bf16x8*>(lds_v)[q] = reinterpret_cast<const bf16x8*>(img)[q];
#else
#pragma unroll
        for (int m = 0; m < 12; ++m) vw[m] = *reinterpret_cast<const bf16x8*>(img + ((m * 32 + li) * 2 + lh) * 8);
#endif
        __syncthreads();
    }
    for (int q = tid; q < 20 * 2; q += 256)       // columns 36..39 of every image row stay zero (16 elements = two 16-byte units per row)
        *reinterpret_cast<f32x4*>(img + ((q >> 1) * D4P_IW + 36) * 4 + (q & 1) * 8) = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 accw[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb)
#pragma unroll
        for (int v = 0; v < 16; ++v) accw[mb][v] = 0.f;
    const int t0 = blockIdx.x * a.tilesPerSplit;
    int t1 = t0 + a.tilesPerSplit; if (t1 > a.numTiles) t1 = a.numTiles;
    // staging: item q = (image row, column pair) carries both pixels x 3 channels: 6 + 6 8-byte loads, one 16-byte LDS unit
    float2 rg[2][3], rr[2][3];
    bf16x8 ro[2];
    auto fetch = [&](int mt) {
        const int ib = mt / TPI, t = mt % TPI;
        const int sy0 = (t / (HS / 16)) * 8, sx0 = (t % (HS / 16)) * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + i * 256, row = q / 18, cp = q % 18;
            const int uy = 2 * sy0 - 2 + row, ux = 2 * sx0 - 2 + 2 * cp;
            const bool ok = q < 360 && (unsigned)uy < (unsigned)H && (unsigned)ux < (unsigned)H;     // ux even: ux + 1 is inside with it
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t e = ok ? ((size_t)(ib * 3 + c) * H + uy) * H + ux : 0;
                const float2 g = *reinterpret_cast<const float2*>(a.a0 + e), r = *reinterpret_cast<const float2*>(a.a2 + e);
                rg[i][c] = ok ? g : make_float2(0.f, 0.f);
                rr[i][c] = r;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + i * 256, c8 = q & 3, sp = q >> 2;
            ro[i] = Act<__bf16>::ld8(a.a1, ((size_t)(ib * HS + sy0 + sp / 16) * HS + sx0 + sp % 16) * 32 + c8 * 8);
        }
    };
    // bias gradient = sum of dOut over the tile's own 16 x 32 output pixels (rows / columns 2.. of the staged halo): which
    // staged items those are is a per-thread constant
    float bsum[3] = {0.f, 0.f, 0.f}, bmask[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + i * 256, row = q / 18, cp = q % 18;
        bmask[i] = (q < 360 && row >= 2 && row < 18 && cp >= 1 && cp < 17) ? 1.f : 0.f;
    }
    if (t0 < t1) fetch(t0);
    // this lane's source pixel in the two contractions
    const int dy = 2 * wave + (li >> 4), dx = li & 15;                          // input gradient: wave w = tile rows 2w, 2w+1
    const __bf16* dsrc = img + ((2 * dy) * D4P_IW + 2 * dx + 2 * lh) * 4;       // + (i*IW + 4*(m&1)) * 4 for k-step m = 2i + (m&1)
    const int g = lane >> 4, h = g >> 1, qrow = (lane & 15) >> 2, cb = 16 * (g & 1) + 4 * (lane & 3);      // transposed-read lane address
    const int whalf = wave & 1, wblk = wave >> 1;
    for (int mt = t0; mt < t1; ++mt) {
        const int ib = mt / TPI, t = mt % TPI;
        const int sy0 = (t / (HS / 16)) * 8, sx0 = (t % (HS / 16)) * 16;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + i * 256;
            // dOut = d_recon * (1 - recon^2): Tanh backward (vae_nets.py:134), d4_actbwd_kernel's expression
            float d[3][2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                d[c][0] = rg[i][c].x * (1.f - rr[i][c].x * rr[i][c].x);
                d[c][1] = rg[i][c].y * (1.f - rr[i][c].y * rr[i][c].y);
                bsum[c] = fmaf(bmask[i], d[c][0] + d[c][1], bsum[c]);
            }
            bf16x8 u;
            u[0] = (__bf16)d[0][0]; u[1] = (__bf16)d[1][0]; u[2] = (__bf16)d[2][0]; u[3] = (__bf16)0.f;
            u[4] = (__bf16)d[0][1]; u[5] = (__bf16)d[1][1]; u[6] = (__bf16)d[2][1]; u[7] = (__bf16)0.f;
            if (q < 360) *reinterpret_cast<bf16x8*>(img + ((q / 18) * D4P_IW + 2 * (q % 18)) * 4) = u;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<bf16x8*>(lds_o + (size_t)(tid + i * 256) * 8) = ro[i];
        __syncthreads();
        if (mt + 1 < t1) fetch(mt + 1);
        // input gradient
        f32x16 accd, accd2;
#pragma unroll
        for (int v = 0; v < 16; ++v) { accd[v] = 0.f; accd2[v] = 0.f; }
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(dsrc + ((m >> 1) * D4P_IW + 4 * (m & 1)) * 4);
#if D4B_VWLDS
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(lds_v + ((m * 32 + li) * 2 + lh) * 8);
#else
            const bf16x8 av = vw[m];
#endif
            if (D4B_ACC2 && (m & 1)) accd2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, accd2, 0, 0, 0);      // two chains of 6
            else accd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, accd, 0, 0, 0);
        }
        if (D4B_ACC2) {
#pragma unroll
            for (int v = 0; v < 16; ++v) accd[v] += accd2[v];
        }
        {
            const int sp = dy * 16 + dx;                     // accumulator column = this lane's source pixel; rows = channels
            const size_t o = ((size_t)(ib * HS + sy0 + dy) * HS + sx0 + dx) * 32;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int c0 = 8 * q4 + 4 * lh;
                const bf16x4 ov = *reinterpret_cast<const bf16x4*>(lds_o + sp * 32 + c0);
                bf16x4 out;
#pragma unroll
                for (int e = 0; e < 4; ++e) out[e] = (float)ov[e] > 0.f ? (__bf16)accd[4 * q4 + e] : (__bf16)0.f;
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.din) + o + c0) = out;
            }
        }
        // weight gradient: this wave's three k-blocks over its half of the tile's source pixels
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int row = 64 * whalf + 16 * ks + 8 * h + qrow;                 // source pixel (tile row row>>4, column row&15); row + 4: same tile row
            const bf16x8 bv = tr_frag(lds_o + row * 32 + cb, lds_o + (row + 4) * 32 + cb);
            const __bf16* ip = img + ((2 * (row >> 4)) * D4P_IW + 2 * (row & 15)) * 4 + cb;
#pragma unroll
            for (int mb = 0; mb < 3; ++mb) {
                const __bf16* rp = ip + (wblk + 2 * mb) * D4P_IW * 4;
                const bf16x8 av = tr_frag(rp, rp + 8 * 4);                       // source pixel + 4 = 8 image columns on
                accw[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, accw[mb], 0, 0, 0);
            }
        }
    }
    // add the two tile halves (waves w, w^1 hold the same k-blocks) and write the slab rows
    float* red = reinterpret_cast<float*>(smem_raw);
    __syncthreads();
    if (whalf == 1) {
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
            for (int v = 0; v < 16; ++v) red[((wblk * 3 + mb) * 16 + v) * 64 + lane] = accw[mb][v];
    }
    __syncthreads();
    float* slab = a.slab + (size_t)blockIdx.x * D4P_ROW;
    if (whalf == 0) {
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int k = (wblk + 2 * mb) * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                if (k != 3) slab[k * 32 + li] = accw[mb][v] + red[((wblk * 3 + mb) * 16 + v) * 64 + lane];
            }
    }
    // row 3 of the slab (k = (0, 0, co = 3): a zero row) carries this workgroup's three bias-gradient partials; the column
    // reduction of the slabs sums them with the weights' (fixed order), d4_perm_kernel picks them up
    __shared__ float bred[4][3];
#pragma unroll
    for (int co = 0; co < 3; ++co) { const float v = wave_sum(bsum[co]); if (lane == 0) bred[wave][co] = v; }
    __syncthreads();
    if (tid < 32) slab[3 * 32 + tid] = tid < 3 ? (bred[0][tid] + bred[1][tid]) + (bred[2][tid] + bred[3][tid]) : 0.f;
}
static constexpr int D4_BWD_BF16_SMEM = D4_BWD_BF16_SMEM_BYTES;       // the final half-add of the accumulators (24 KB) exceeds the staging buffers (14.6 KB)
static_assert(D4_BWD_BF16_SMEM >= (20 * D4P_IW * 4 + 128 * 32) * 2, "staging buffers");

static constexpr int D4_BWD_SMEM = (3 * 720 + 128 * 32 + 128 * 77 + 32 + 76 * 32) * 4;

static int d4_splits(int width, int B, int* tps, bool bf16io = false) {
    return thin_splits(B * (width / 16) * (width / 32), tps, bf16io ? D4B_OCC * cvae_num_cus() : 512);
}
// ws of launch_d4_bwd = [S*3072 split-K slab | B*3 plane sums of dOut]
int64_t d4_bwd_ws_floats(int width, int B) {
    int tps;
    const int64_t S = thin_splits(B * (width / 16) * (width / 32), &tps, THIN_SPLIT_CAP);   // upper bound of the split counts used
    const int64_t row = D4P_ROW;                                                        // bf16 mode's slab row (fp32: 3072)
    return S * row + align_up((int64_t)B * 3, 64) + row + col_reduce_ws_floats((int)row);
}

int launch_d4_fwd(int width, int B, const float* in, const float* w, const float* bias, float* recon, hipStream_t st, bool bf16io) {
    // persistent: 16 x 32-output tiles; bf16 mode 4 workgroups per CU (18 KB of LDS), fp32 2 (their prologue builds the 72 collapsed
    // weight values of every lane: fewer, longer-lived workgroups); each loops over its share
    const int want = (bf16io ? 4 : 2) * cvae_num_cus();
    const int tiles = B * (width / 16) * (width / 32), grid = tiles < want ? tiles : want;
    cvae_probe_begin(st);
    if (width == 64 && bf16io) hipLaunchKernelGGL(d4_fwd_bf16_kernel<64>, dim3(grid), dim3(256), 0, st, in, w, bias, recon, B);
    else if (width == 128 && bf16io) hipLaunchKernelGGL(d4_fwd_bf16_kernel<128>, dim3(grid), dim3(256), 0, st, in, w, bias, recon, B);
    else if (width == 64) hipLaunchKernelGGL(d4_fwd_pc_f32_kernel<64>, dim3(grid), dim3(256), 0, st, in, w, bias, recon, B);
    else if (width == 128) hipLaunchKernelGGL(d4_fwd_pc_f32_kernel<128>, dim3(grid), dim3(256), 0, st, in, w, bias, recon, B);
    else { cvae_set_error("d4_fwd: width %d unsupported", width); return -2; }
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    return 0;
}

// Fused D4 backward: Tanh backward -> dout planes (B,3,W,W), then d_o3 (ReLU-masked), dW4, db4.
int launch_d4_bwd(int width, int B, const float* o3, const float* d_recon, const float* recon, const float* w,
                  float* dout, float* d_o3, float* dw, float* db, float* ws, hipStream_t st, bool bf16io) {
    if (width != 64 && width != 128) { cvae_set_error("d4_bwd: width %d unsupported", width); return -2; }
    int tps;
    const int tiles = B * (width / 16) * (width / 32);
    const int S = d4_splits(width, B, &tps, bf16io);
    const int row = bf16io ? D4P_ROW : 3072;                  // floats of one workgroup's slab
    float* plane_sums = ws + (size_t)S * row;
    if (!bf16io) {      // fp32: Tanh backward as its own pass (writes dOut); bf16 mode applies it while the backward kernel stages
        hipLaunchKernelGGL(d4_actbwd_kernel, dim3(B * 3), dim3(256), 0, st, d_recon, recon, dout, plane_sums, width * width);
        CVAE_CHECK_LAUNCH();
    }
    ThinWgradArgs a{bf16io ? d_recon : dout, bf16io ? recon : nullptr, o3, w, d_o3, ws, B, tiles, tps};
    static DeviceOnce once[4];
    void (*kern)(ThinWgradArgs) = width == 64 ? (bf16io ? d4_bwd_bf16_kernel<64> : d4_bwd_kernel<64, float>)
                                              : (bf16io ? d4_bwd_bf16_kernel<128> : d4_bwd_kernel<128, float>);
    const int smem_bytes = bf16io ? D4_BWD_BF16_SMEM : D4_BWD_SMEM;
    { int rc = cvae_grant_lds(once[(width == 128) * 2 + bf16io], reinterpret_cast<const void*>(kern), smem_bytes); if (rc) return rc; }
    cvae_probe_begin(st);
    hipLaunchKernelGGL(kern, dim3(S), dim3(256), smem_bytes, st, a);
    cvae_probe_end(st);
    CVAE_CHECK_LAUNCH();
    float* red = plane_sums + align_up((int64_t)B * 3, 64);
    { int rc = launch_col_reduce(ws, S, row, row, red, red + row, st); if (rc) return rc; }
    hipLaunchKernelGGL(d4_perm_kernel, dim3(cdiv(2400, 256) + 1), dim3(256), 0, st, red, dw, bf16io ? nullptr : plane_sums, db, B);
    CVAE_CHECK_LAUNCH();
    return 0;
}
