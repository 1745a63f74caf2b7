// fc.hip — the three Linear layers around the latent, fused with reparametrize and the concat.
//
// Replaces (vae_nets.py):  :105-109 flatten + fc_mu + fc_var,  :48-51 reparametrize (eps given),
// :143 cat((z,pred),1) + decoder_input Linear(33, bottleneck),  :144 view(-1,256,s,s), and their
// autograd.  Tiny GEMMs (0.8 MFLOP/img) -> VALU kernels, LDS-broadcast operands, fixed-order
// two-stage reductions.  Native layouts: Wfc [K][64] (cols 0..31 = fc_mu, 32..63 = fc_var, K in
// (h,w,c) order), Wd [33][K] (K in (h,w,c) order) so that `flat`/`h` are the NHWC activations.
#include "common.h"

static constexpr int FC_KS = 32;         // K splits of the fc_mu|fc_var GEMM
static constexpr int FC_IMGS = 16;       // images per workgroup

// slab[ks][m][64] = sum_{k in K-slice ks} A[m][k] * Bm(k, n)   on v_mfma_f32_32x32x2_f32.
// A is [M][K] row-major.  B_KMAJOR == false: Bm(k,n) = Bp[k*64 + n] (fc_mu|fc_var weights, N = 64);
// B_KMAJOR == true : Bm(k,n) = Bp[n*K + k] for n < NV, 0 otherwise (decoder_input transposed, NV = 33).
// WG = 128 rows x 64 columns x one K-slice; wave w owns rows 32w..32w+31 and both 32-column tiles.
template <bool B_KMAJOR, typename AT>       // AT = storage type of A (an activation / activation gradient)
__global__ __launch_bounds__(256) void latent_gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bp,
                                                          float* __restrict__ slab, int M, int K, int kslice, int NV) {
    __shared__ float lds_a[128 * 33];
    __shared__ float lds_b[32 * 65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 128, k0 = blockIdx.y * kslice;
    f32x16 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[nb][v] = 0.f;
    for (int kc = k0; kc < k0 + kslice; kc += 32) {
        __syncthreads();
        for (int q = tid; q < 128 * 8; q += 256) {
            const int c4 = q & 7, r = q >> 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m0 + r < M) v = Act<AT>::ld4(A, (size_t)(m0 + r) * K + kc + c4 * 4);
            float* d = lds_a + r * 33 + c4 * 4;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
        if (!B_KMAJOR) {
            for (int q = tid; q < 32 * 16; q += 256) {
                const int c4 = q & 15, kk = q >> 4;
                const float4 v = *reinterpret_cast<const float4*>(Bp + (size_t)(kc + kk) * 64 + c4 * 4);
                float* d = lds_b + kk * 65 + c4 * 4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int q = tid; q < 64 * 8; q += 256) {
                const int c4 = q & 7, n = q >> 3;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < NV) v = *reinterpret_cast<const float4*>(Bp + (size_t)n * K + kc + c4 * 4);
                lds_b[(c4 * 4 + 0) * 65 + n] = v.x; lds_b[(c4 * 4 + 1) * 65 + n] = v.y;
                lds_b[(c4 * 4 + 2) * 65 + n] = v.z; lds_b[(c4 * 4 + 3) * 65 + n] = v.w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float av = lds_a[(wave * 32 + li) * 33 + 2 * j + lh];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, lds_b[(2 * j + lh) * 65 + nb * 32 + li], acc[nb], 0, 0, 0);
        }
    }
    float* out = slab + (size_t)blockIdx.y * M * 64;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int m = m0 + wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            if (m < M) out[(size_t)m * 64 + nb * 32 + li] = acc[nb][v];
        }
}

// dzcat[b][i] = sum_ks slab[ks][b][i], i < 33
__global__ __launch_bounds__(256) void decin_dz_finish_kernel(const float* __restrict__ slab, float* __restrict__ dzcat, int B, int KS) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * 33) return;
    const int b = idx / 33, i = idx % 33;
    float acc = 0.f;
    if (KS == FC_KS) {                      // the split count the launchers use: all loads in flight at once
#pragma unroll
        for (int ks = 0; ks < FC_KS; ++ks) acc += slab[((size_t)ks * B + b) * 64 + i];
    } else {
        for (int ks = 0; ks < KS; ++ks) acc += slab[((size_t)ks * B + b) * 64 + i];
    }
    dzcat[idx] = acc;
}

// part[ks][b][64] = sum_{k in split ks} flat[b][k] * Wfc[k][n]
__global__ __launch_bounds__(256) void fc_fwd_partial_kernel(const float* __restrict__ flat, const float* __restrict__ wfc,
                                                             float* __restrict__ part, int B, int K) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];   // [FC_IMGS][kchunk]
    const int kchunk = K / FC_KS;
    const int b0 = blockIdx.x * FC_IMGS, ks = blockIdx.y, k0 = ks * kchunk;
    const int n = threadIdx.x & 63, bq = threadIdx.x >> 6;
    for (int q = threadIdx.x; q < FC_IMGS * kchunk / 4; q += 256) {
        const int i = q / (kchunk / 4), k4 = q % (kchunk / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b0 + i < B) v = *reinterpret_cast<const float4*>(flat + (size_t)(b0 + i) * K + k0 + k4 * 4);
        *reinterpret_cast<float4*>(lds_f + i * kchunk + k4 * 4) = v;
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* f = lds_f + (bq * 4) * kchunk;
#pragma unroll 8
    for (int k = 0; k < kchunk; ++k) {
        const float wv = wfc[(size_t)(k0 + k) * 64 + n];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaf(f[i * kchunk + k], wv, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = b0 + bq * 4 + i;
        if (b < B) part[((size_t)ks * B + b) * 64 + n] = acc[i];
    }
}

// mu, logvar = bias + sum of partials; z = mu + eps*exp(0.5*logvar); zcat = [z | pred]
__global__ __launch_bounds__(256) void fc_finish_kernel(const float* __restrict__ part, const float* __restrict__ bfc,
                                                        const float* __restrict__ eps, const float* __restrict__ pred,
                                                        float* __restrict__ mu, float* __restrict__ logvar,
                                                        float* __restrict__ zcat, int B) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * 32) return;
    const int b = idx >> 5, d = idx & 31;
    float m = bfc[d], lv = bfc[32 + d];
    for (int ks = 0; ks < FC_KS; ++ks) {
        m += part[((size_t)ks * B + b) * 64 + d];
        lv += part[((size_t)ks * B + b) * 64 + 32 + d];
    }
    mu[idx] = m;
    logvar[idx] = lv;
    zcat[b * 33 + d] = m + eps[idx] * expf(0.5f * lv);
    if (d == 0) zcat[b * 33 + 32] = pred[b];
}

// h[b][j] = bd[j] + sum_i zcat[b][i] * Wd[i][j].  Workgroup = 1024 columns x DI_IMGS images: every 16-byte load of Wd
// serves DI_IMGS images (one image per workgroup re-read the 33 x K weights B times from L2: 1.1 GB at B = 2048,
// 44 us; now 69 MB).  Per output the sum still runs i = 0..32 from the bias.
static constexpr int DI_IMGS = 16;
template <typename AT>
__global__ __launch_bounds__(256) void decin_fwd_kernel(const float* __restrict__ zcat, const float* __restrict__ wd,
                                                        const float* __restrict__ bd, float* __restrict__ h, int K, int B) {
    __shared__ float z[DI_IMGS][33];
    const int b0 = blockIdx.y * DI_IMGS, j = (blockIdx.x * 256 + threadIdx.x) * 4;
    for (int q = threadIdx.x; q < DI_IMGS * 33; q += 256) z[q / 33][q % 33] = b0 + q / 33 < B ? zcat[(size_t)b0 * 33 + q] : 0.f;
    __syncthreads();
    const f32x4 bias = *reinterpret_cast<const f32x4*>(bd + j);
    f32x4 acc[DI_IMGS];
#pragma unroll
    for (int m = 0; m < DI_IMGS; ++m) acc[m] = bias;
#pragma unroll 3
    for (int i = 0; i < 33; ++i) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(wd + (size_t)i * K + j);
#pragma unroll
        for (int m = 0; m < DI_IMGS; ++m) {
            const float zv = z[m][i];                         // wave-uniform LDS broadcast
            acc[m][0] = fmaf(zv, w[0], acc[m][0]); acc[m][1] = fmaf(zv, w[1], acc[m][1]);
            acc[m][2] = fmaf(zv, w[2], acc[m][2]); acc[m][3] = fmaf(zv, w[3], acc[m][3]);
        }
    }
#pragma unroll
    for (int m = 0; m < DI_IMGS; ++m)
        if (b0 + m < B) Act<AT>::st4(h, (size_t)(b0 + m) * K + j, acc[m]);
}

// dzcat[b][i] = sum_j dh[b][j] * Wd[i][j].  One wave = 2 images x all 33 rows of Wd: lanes split K,
// 66 per-lane partial sums, one shuffle reduction per output at the end (no LDS, no barriers).
__global__ __launch_bounds__(256) void decin_bwd_dz_kernel(const float* __restrict__ dh, const float* __restrict__ wd,
                                                           float* __restrict__ dzcat, int B, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = (blockIdx.x * 4 + wave) * 2;
    if (b0 >= B) return;
    const bool two = b0 + 1 < B;
    float acc0[33], acc1[33];
#pragma unroll
    for (int i = 0; i < 33; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    for (int j = lane * 4; j < K; j += 256) {
        const float4 g0 = *reinterpret_cast<const float4*>(dh + (size_t)b0 * K + j);
        const float4 g1 = two ? *reinterpret_cast<const float4*>(dh + (size_t)(b0 + 1) * K + j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 33; ++i) {
            const float4 w = *reinterpret_cast<const float4*>(wd + (size_t)i * K + j);
            acc0[i] += (g0.x * w.x + g0.y * w.y) + (g0.z * w.z + g0.w * w.w);
            acc1[i] += (g1.x * w.x + g1.y * w.y) + (g1.z * w.z + g1.w * w.w);
        }
    }
#pragma unroll
    for (int i = 0; i < 33; ++i) {
        const float s0 = wave_sum(acc0[i]), s1 = wave_sum(acc1[i]);
        if (lane == 0) { dzcat[b0 * 33 + i] = s0; if (two) dzcat[(b0 + 1) * 33 + i] = s1; }
    }
}

// slab[bs][i][j] (i<33: dWd, i==33: dbd) = sum over the batch slice bs
template <typename AT>
__global__ __launch_bounds__(256) void decin_bwd_dw_kernel(const float* __restrict__ zcat, const float* __restrict__ dh,
                                                           float* __restrict__ slab, int B, int K, int bPerSplit) {
    constexpr int ZB = 16;                          // batch rows staged per barrier pair
    __shared__ float z[ZB * 33];
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int b0 = blockIdx.y * bPerSplit;
    int b1 = b0 + bPerSplit; if (b1 > B) b1 = B;
    float acc[34];
#pragma unroll
    for (int i = 0; i < 34; ++i) acc[i] = 0.f;
    for (int bb = b0; bb < b1; bb += ZB) {
        const int nb = b1 - bb < ZB ? b1 - bb : ZB;
        __syncthreads();
        for (int q = threadIdx.x; q < nb * 33; q += 256) z[q] = zcat[bb * 33 + q];
        __syncthreads();
        for (int r = 0; r < nb; ++r) {
            const float g = Act<AT>::ld(dh, (size_t)(bb + r) * K + j);
#pragma unroll
            for (int i = 0; i < 33; ++i) acc[i] = fmaf(z[r * 33 + i], g, acc[i]);
            acc[33] += g;
        }
    }
    float* o = slab + (size_t)blockIdx.y * 34 * K;
#pragma unroll
    for (int i = 0; i < 34; ++i) o[(size_t)i * K + j] = acc[i];
}

// dml[b][0..31] = dz + dmu_loss ; dml[b][32..63] = dz*eps*0.5*exp(0.5*logvar) + dlv_loss
__global__ __launch_bounds__(256) void fc_bwd_prep_kernel(const float* __restrict__ dzcat, const float* __restrict__ eps,
                                                          const float* __restrict__ logvar, const float* __restrict__ dmu_loss,
                                                          const float* __restrict__ dlv_loss, float* __restrict__ dml, int B) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * 32) return;
    const int b = idx >> 5, d = idx & 31;
    const float dz = dzcat[b * 33 + d];
    dml[b * 64 + d] = dz + dmu_loss[idx];
    dml[b * 64 + 32 + d] = dz * eps[idx] * 0.5f * expf(0.5f * logvar[idx]) + dlv_loss[idx];
}

// dflat[b][k] = sum_n dml[b][n] * Wfc[k][n].  Thread = one k (its 64 weights in registers), workgroup = 256 k x DF_IMGS
// images (8 images per workgroup re-read the 1 MB of Wfc 256 times at B = 2048 and ran at 42 us); n = 0..63 in order.
static constexpr int DF_IMGS = 32;
template <typename AT>
__global__ __launch_bounds__(256) void fc_bwd_dflat_kernel(const float* __restrict__ dml, const float* __restrict__ wfc,
                                                           float* __restrict__ dflat, int B, int K) {
    __shared__ __attribute__((aligned(16))) float g[DF_IMGS][64];
    const int b0 = blockIdx.x * DF_IMGS, k = blockIdx.y * 256 + threadIdx.x;
    for (int q = threadIdx.x; q < DF_IMGS * 16; q += 256) {
        const int r = q >> 4, c4 = q & 15;
        *reinterpret_cast<f32x4*>(&g[r][c4 * 4]) = b0 + r < B ? *reinterpret_cast<const f32x4*>(dml + (size_t)(b0 + r) * 64 + c4 * 4)
                                                              : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    float w[64];
#pragma unroll
    for (int n4 = 0; n4 < 16; ++n4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wfc + (size_t)k * 64 + n4 * 4);
        w[n4 * 4] = v[0]; w[n4 * 4 + 1] = v[1]; w[n4 * 4 + 2] = v[2]; w[n4 * 4 + 3] = v[3];
    }
#pragma unroll 4
    for (int i = 0; i < DF_IMGS; ++i) {
        if (b0 + i >= B) break;
        float acc = 0.f;
#pragma unroll
        for (int n4 = 0; n4 < 16; ++n4) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(&g[i][n4 * 4]);      // wave-uniform 16-byte LDS broadcast
            acc = fmaf(gv[0], w[n4 * 4], acc); acc = fmaf(gv[1], w[n4 * 4 + 1], acc);
            acc = fmaf(gv[2], w[n4 * 4 + 2], acc); acc = fmaf(gv[3], w[n4 * 4 + 3], acc);
        }
        Act<AT>::st(dflat, (size_t)(b0 + i) * K + k, acc);
    }
}

// dWfc[k][n] = sum_b flat[b][k] * dml[b][n]    (16 k-rows per workgroup: each wave 4 rows x 64 columns, all of
// the batch; the four flat values of a row group are one 16-byte wave-uniform load per image)
template <typename AT>
__global__ __launch_bounds__(256) void fc_bwd_dw_kernel(const float* __restrict__ flat, const float* __restrict__ dml,
                                                        float* __restrict__ dwfc, int B, int K) {
    const int n = threadIdx.x & 63, k = blockIdx.x * 16 + (threadIdx.x >> 6) * 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) {
        const f32x4 f = Act<AT>::ld4(flat, (size_t)b * K + k);
        const float d = dml[(size_t)b * 64 + n];
        a0 = fmaf(f.x, d, a0); a1 = fmaf(f.y, d, a1); a2 = fmaf(f.z, d, a2); a3 = fmaf(f.w, d, a3);
    }
    dwfc[(size_t)k * 64 + n] = a0; dwfc[(size_t)(k + 1) * 64 + n] = a1;
    dwfc[(size_t)(k + 2) * 64 + n] = a2; dwfc[(size_t)(k + 3) * 64 + n] = a3;
}

// precision mode 1: the two batch-contracted weight gradients  C[m][n] = sum_b A[b][m] * Bm[b][n]  on the bf16 MFMA
// (fc_mu|fc_var: A = flat (bf16 activations), Bm = dml;  decoder_input: A = [zcat | 1] (the ones column yields the
// bias gradient), Bm = dh (bf16)).  Both inputs are row-major in b, i.e. the contraction index is the ROW: tiles of BG_BT
// images are copied to LDS as they are (fp32 inputs rounded to bf16 on the way) and both MFMA operands are transposed
// LDS reads (ds_read_b64_tr_b16).  Workgroup = (32*MBLK) x (32*NBLK) outputs over the whole batch (no split-K slabs);
// wave w contracts images [w*BG_BT/4, (w+1)*BG_BT/4) of every tile, the four partial tiles are summed through LDS.
struct BGemmArgs {
    const float* A; const float* Bm;      // opaque: bf16 or fp32 per template flags
    int lda, ldb;                         // row strides in elements
    int a_cols;                           // valid columns of A (fp32 A only); column a_cols reads as 1.0, beyond as 0
    float* out; int ldo;                  // C rows [0, out_rows) -> out[m*ldo + n0 + n]
    int out_rows;
    float* out_last;                      // row `out_rows` (the ones column) -> out_last[n0 + n], may be null
    int B;
};

// BT images per LDS tile: each tile costs two workgroup barriers, and at 64 images a wave had two MFMAs between them (41 / 35 us
// at B = 2048 for 1 GFLOP); 256 images per tile = 4 k-steps per wave and tile
static constexpr int BG_BT = 256;
template <int MBLK, int NBLK, bool A_F32, bool B_F32>
__global__ __launch_bounds__(256) void bgemm_tr_kernel(BGemmArgs a) {
    constexpr int MC = 32 * MBLK, NC = 32 * NBLK, BT = BG_BT;
    __shared__ __attribute__((aligned(16))) __bf16 lds_a[BT * MC];
    __shared__ __attribute__((aligned(16))) __bf16 lds_b[BT * NC];
    __shared__ float red[3 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int g = lane >> 4, h = g >> 1, qrow = (lane & 15) >> 2, cb = 16 * (g & 1) + 4 * (lane & 3);
    const int m0 = A_F32 ? 0 : blockIdx.x * MC, n0 = (A_F32 ? blockIdx.x : 0) * NC;   // fp32-A form tiles N, bf16-A form tiles M
    f32x16 acc[MBLK][NBLK];
#pragma unroll
    for (int i = 0; i < MBLK; ++i)
#pragma unroll
        for (int j = 0; j < NBLK; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    // fp32 A (decoder_input: [zcat | 1], 34 real columns of MC = 64): only the real columns travel — item q = (row q / AC, column
    // q % AC); the LDS columns past them are zeroed once and never written again (half the loads and stores of a dense tile)
    constexpr int AC = 34;
    constexpr int AU = A_F32 ? (BT * AC) / 256 : (BT * MC / 8 + 255) / 256;        // per-thread staging items
    constexpr int BU = B_F32 ? (BT * NC / 4) / 256 : (BT * NC / 8 + 255) / 256;
    float ra32[A_F32 ? AU : 1];
    bf16x8 ra16[A_F32 ? 1 : AU];
    f32x4 rb32[B_F32 ? BU : 1];
    bf16x8 rb16[B_F32 ? 1 : BU];
    bf16x8 zero8;
#pragma unroll
    for (int c = 0; c < 8; ++c) zero8[c] = (__bf16)0.f;
    auto fetch = [&](int b0) {
        if constexpr (A_F32) {
#pragma unroll
            for (int i = 0; i < AU; ++i) {
                const int q = tid + i * 256, r = q / AC, c = q % AC, b = b0 + r;
                const bool ok = b < a.B && c < a.a_cols;
                const float l = a.A[ok ? (size_t)b * a.lda + c : 0];
                ra32[i] = ok ? l : ((b < a.B && c == a.a_cols) ? 1.0f : 0.f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < AU; ++i) {
                const int q = tid + i * 256, r = q / (MC / 8), c8 = q % (MC / 8), b = b0 + r;
                const bool ok = (BT * MC / 8) % 256 == 0 || q < BT * MC / 8;
                const bool inb = ok && b < a.B;
                const bf16x8 l = Act<__bf16>::ld8(a.A, inb ? (size_t)b * a.lda + m0 + c8 * 8 : 0);
                ra16[i] = inb ? l : zero8;
            }
        }
        if constexpr (B_F32) {
#pragma unroll
            for (int i = 0; i < BU; ++i) {
                const int q = tid + i * 256, r = q / (NC / 4), c4 = q % (NC / 4), b = b0 + r;
                const bool ok = b < a.B;
                const f32x4 l = *reinterpret_cast<const f32x4*>(a.Bm + (ok ? (size_t)b * a.ldb + n0 + c4 * 4 : 0));
                rb32[i] = ok ? l : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int i = 0; i < BU; ++i) {
                const int q = tid + i * 256, r = q / (NC / 8), c8 = q % (NC / 8), b = b0 + r;
                const bool ok = ((BT * NC / 8) % 256 == 0 || q < BT * NC / 8) && b < a.B;
                const bf16x8 l = Act<__bf16>::ld8(a.Bm, ok ? (size_t)b * a.ldb + n0 + c8 * 8 : 0);
                rb16[i] = ok ? l : zero8;
            }
        }
    };
    if constexpr (A_F32) {
        static_assert((BT * AC) % 256 == 0 && AC <= MC, "whole staging rounds of the real columns");
        for (int q = tid; q < BT * MC; q += 256) lds_a[q] = (__bf16)0.f;      // columns >= AC stay zero
    }
    fetch(0);
    for (int b0 = 0; b0 < a.B; b0 += BT) {
        __syncthreads();
        if constexpr (A_F32) {
#pragma unroll
            for (int i = 0; i < AU; ++i) { const int q = tid + i * 256; lds_a[(q / AC) * MC + q % AC] = (__bf16)ra32[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < AU; ++i) {
                const int q = tid + i * 256;
                if ((BT * MC / 8) % 256 == 0 || q < BT * MC / 8) *reinterpret_cast<bf16x8*>(lds_a + (size_t)q * 8) = ra16[i];
            }
        }
        if constexpr (B_F32) {
#pragma unroll
            for (int i = 0; i < BU; ++i) {
                bf16x4 u;
                u[0] = (__bf16)rb32[i][0]; u[1] = (__bf16)rb32[i][1]; u[2] = (__bf16)rb32[i][2]; u[3] = (__bf16)rb32[i][3];
                *reinterpret_cast<bf16x4*>(lds_b + (size_t)(tid + i * 256) * 4) = u;
            }
        } else {
#pragma unroll
            for (int i = 0; i < BU; ++i) {
                const int q = tid + i * 256;
                if ((BT * NC / 8) % 256 == 0 || q < BT * NC / 8) *reinterpret_cast<bf16x8*>(lds_b + (size_t)q * 8) = rb16[i];
            }
        }
        __syncthreads();
        if (b0 + BT < a.B) fetch(b0 + BT);
#pragma unroll
        for (int ks = 0; ks < BT / 64; ++ks) {
            const int row = (BT / 4) * wave + 16 * ks + 8 * h + qrow;      // this wave's BT/4 images of the tile, 16 per k-step
            bf16x8 bv[NBLK];
#pragma unroll
            for (int j = 0; j < NBLK; ++j) bv[j] = tr_frag(lds_b + row * NC + j * 32 + cb, lds_b + (row + 4) * NC + j * 32 + cb);
#pragma unroll
            for (int i = 0; i < MBLK; ++i) {
                const bf16x8 av = tr_frag(lds_a + row * MC + i * 32 + cb, lds_a + (row + 4) * MC + i * 32 + cb);
#pragma unroll
                for (int j = 0; j < NBLK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MBLK; ++i)
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
            __syncthreads();
            if (wave > 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) red[((wave - 1) * 16 + v) * 64 + lane] = acc[i][j][v];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float x = ((acc[i][j][v] + red[v * 64 + lane]) + red[(16 + v) * 64 + lane]) + red[(32 + v) * 64 + lane];
                    const int m = m0 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh, n = n0 + j * 32 + li;
                    if (m < a.out_rows) a.out[(size_t)m * a.ldo + n] = x;
                    else if (m == a.out_rows && a.out_last) a.out_last[n] = x;
                }
            }
        }
}

// fp32 mode: the same batch-contracted weight gradients on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: one A and one B
// value per lane and k-step, read as conflict-free LDS rows — no transposition needed).  Replaces the serial VALU loops
// over the batch (fc_bwd_dw_kernel) and the batch-split slabs + two slab reductions (decin_bwd_dw_kernel).
// A_PAD: A is [b][a_cols] fp32 with an implicit ones column at a_cols (decoder_input); else A is [b][lda] and the
// workgroup takes columns m0..m0+MC.
template <int MBLK, int NBLK, bool A_PAD>
__global__ __launch_bounds__(256) void bgemm_f32_kernel(BGemmArgs a) {
    constexpr int MC = 32 * MBLK, NC = 32 * NBLK;
    __shared__ __attribute__((aligned(16))) float lds_a[64 * MC];
    __shared__ __attribute__((aligned(16))) float lds_b[64 * NC];
    __shared__ float red[3 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int m0 = A_PAD ? 0 : blockIdx.x * MC, n0 = (A_PAD ? blockIdx.x : 0) * NC;
    f32x16 acc[MBLK][NBLK];
#pragma unroll
    for (int i = 0; i < MBLK; ++i)
#pragma unroll
        for (int j = 0; j < NBLK; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    constexpr int AU = A_PAD ? (64 * MC) / 256 : (64 * MC / 4) / 256, BU = (64 * NC / 4) / 256;
    float ra1[A_PAD ? AU : 1];
    f32x4 ra4[A_PAD ? 1 : AU], rb4[BU];
    auto fetch = [&](int b0) {
        if constexpr (A_PAD) {
#pragma unroll
            for (int i = 0; i < AU; ++i) {
                const int q = tid + i * 256, r = q / MC, c = q % MC, b = b0 + r;
                const bool ok = b < a.B && c < a.a_cols;
                const float l = a.A[ok ? (size_t)b * a.lda + c : 0];
                ra1[i] = ok ? l : ((b < a.B && c == a.a_cols) ? 1.0f : 0.f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < AU; ++i) {
                const int q = tid + i * 256, r = q / (MC / 4), c4 = q % (MC / 4), b = b0 + r;
                const bool ok = b < a.B;
                const f32x4 l = *reinterpret_cast<const f32x4*>(a.A + (ok ? (size_t)b * a.lda + m0 + c4 * 4 : 0));
                ra4[i] = ok ? l : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < BU; ++i) {
            const int q = tid + i * 256, r = q / (NC / 4), c4 = q % (NC / 4), b = b0 + r;
            const bool ok = b < a.B;
            const f32x4 l = *reinterpret_cast<const f32x4*>(a.Bm + (ok ? (size_t)b * a.ldb + n0 + c4 * 4 : 0));
            rb4[i] = ok ? l : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(0);
    for (int b0 = 0; b0 < a.B; b0 += 64) {
        __syncthreads();
        if constexpr (A_PAD) {
#pragma unroll
            for (int i = 0; i < AU; ++i) lds_a[tid + i * 256] = ra1[i];
        } else {
#pragma unroll
            for (int i = 0; i < AU; ++i) *reinterpret_cast<f32x4*>(lds_a + (size_t)(tid + i * 256) * 4) = ra4[i];
        }
#pragma unroll
        for (int i = 0; i < BU; ++i) *reinterpret_cast<f32x4*>(lds_b + (size_t)(tid + i * 256) * 4) = rb4[i];
        __syncthreads();
        if (b0 + 64 < a.B) fetch(b0 + 64);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {                       // this wave's 16 images of the tile, two per MFMA
            const int row = 16 * wave + 2 * kk + lh;
            float bv[NBLK];
#pragma unroll
            for (int j = 0; j < NBLK; ++j) bv[j] = lds_b[row * NC + j * 32 + li];
#pragma unroll
            for (int i = 0; i < MBLK; ++i) {
                const float av = lds_a[row * MC + i * 32 + li];
#pragma unroll
                for (int j = 0; j < NBLK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MBLK; ++i)
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
            __syncthreads();
            if (wave > 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) red[((wave - 1) * 16 + v) * 64 + lane] = acc[i][j][v];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float x = ((acc[i][j][v] + red[v * 64 + lane]) + red[(16 + v) * 64 + lane]) + red[(32 + v) * 64 + lane];
                    const int m = m0 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh, n = n0 + j * 32 + li;
                    if (m < a.out_rows) a.out[(size_t)m * a.ldo + n] = x;
                    else if (m == a.out_rows && a.out_last) a.out_last[n] = x;
                }
            }
        }
}

static inline int bott(int width) { return 256 * (width / 16) * (width / 16); }
static inline int decin_splits(int B) { int s = cdiv(B, 16); return s > 16 ? 16 : s; }

int64_t fc_ws_floats(int width, int B) {
    const int K = bott(width);
    const int64_t a = (int64_t)FC_KS * B * 64;                 // fc forward partials
    const int64_t b = (int64_t)decin_splits(B) * 34 * K;       // decoder_input dW slabs
    const int64_t c = (int64_t)B * 64 + colsum_ws_floats(B, 64);   // dml + its column sums
    return (a > b ? a : b) + c;
}

int launch_fc_fwd(int width, int B, const float* flat, const float* wfc, const float* bfc, const float* eps,
                  const float* pred, float* mu, float* logvar, float* zcat, float* ws, hipStream_t st, bool bf16io) {
    const int K = bott(width);
    if (bf16io) hipLaunchKernelGGL((latent_gemm_kernel<false, __bf16>), dim3(cdiv(B, 128), FC_KS), dim3(256), 0, st, flat, wfc, ws, B, K, K / FC_KS, 64);
    else hipLaunchKernelGGL((latent_gemm_kernel<false, float>), dim3(cdiv(B, 128), FC_KS), dim3(256), 0, st, flat, wfc, ws, B, K, K / FC_KS, 64);
    CVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(fc_finish_kernel, dim3(cdiv(B * 32, 256)), dim3(256), 0, st, ws, bfc, eps, pred, mu, logvar, zcat, B);
    CVAE_CHECK_LAUNCH();
    return 0;
}

int launch_decin_fwd(int width, int B, const float* zcat, const float* wd, const float* bd, float* h, hipStream_t st, bool bf16io) {
    const int K = bott(width);
    if (bf16io) hipLaunchKernelGGL(decin_fwd_kernel<__bf16>, dim3(K / 1024, cdiv(B, DI_IMGS)), dim3(256), 0, st, zcat, wd, bd, h, K, B);
    else hipLaunchKernelGGL(decin_fwd_kernel<float>, dim3(K / 1024, cdiv(B, DI_IMGS)), dim3(256), 0, st, zcat, wd, bd, h, K, B);
    CVAE_CHECK_LAUNCH();
    return 0;
}

int launch_decin_bwd(int width, int B, const float* zcat, const float* dh, const float* wd, float* dwd, float* dbd,
                     float* dzcat, float* ws, hipStream_t st, bool bf16io) {
    const int K = bott(width);
    if (bf16io) hipLaunchKernelGGL((latent_gemm_kernel<true, __bf16>), dim3(cdiv(B, 128), FC_KS), dim3(256), 0, st, dh, wd, ws, B, K, K / FC_KS, 33);
    else hipLaunchKernelGGL((latent_gemm_kernel<true, float>), dim3(cdiv(B, 128), FC_KS), dim3(256), 0, st, dh, wd, ws, B, K, K / FC_KS, 33);
    CVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(decin_dz_finish_kernel, dim3(cdiv(B * 33, 256)), dim3(256), 0, st, ws, dzcat, B, FC_KS);
    CVAE_CHECK_LAUNCH();
    if (bf16io) {        // [zcat | 1]^T . dh on the bf16 MFMA, whole batch per workgroup: dWd and dbd written directly
        BGemmArgs g{zcat, dh, 33, K, 33, dwd, K, 33, dbd, B};
        hipLaunchKernelGGL((bgemm_tr_kernel<2, 1, true, false>), dim3(K / 32), dim3(256), 0, st, g);
        CVAE_CHECK_LAUNCH();
        return 0;
    }
    {                    // fp32: [zcat | 1]^T . dh on the fp32 MFMA, whole batch per workgroup
        BGemmArgs g{zcat, dh, 33, K, 33, dwd, K, 33, dbd, B};
        hipLaunchKernelGGL((bgemm_f32_kernel<2, 1, true>), dim3(K / 32), dim3(256), 0, st, g);
        CVAE_CHECK_LAUNCH();
        return 0;
    }
    const int S = decin_splits(B), bps = cdiv(B, S);
    hipLaunchKernelGGL(decin_bwd_dw_kernel<float>, dim3(K / 256, S), dim3(256), 0, st, zcat, dh, ws, B, K, bps);
    CVAE_CHECK_LAUNCH();
    int rc = launch_reduce_slabs(ws, dwd, (int64_t)33 * K, S, (int64_t)34 * K, st);
    if (rc) return rc;
    return launch_reduce_slabs(ws + (size_t)33 * K, dbd, K, S, (int64_t)34 * K, st);
}

int launch_fc_bwd(int width, int B, const float* flat, const float* wfc, const float* dzcat, const float* eps,
                  const float* logvar, const float* dmu_loss, const float* dlv_loss, float* dwfc, float* dbfc,
                  float* dflat, float* ws, hipStream_t st, bool bf16io) {
    const int K = bott(width);
    const int64_t a = (int64_t)FC_KS * B * 64, b = (int64_t)decin_splits(B) * 34 * K;
    float* dml = ws + (a > b ? a : b);
    float* csws = dml + (size_t)B * 64;
    hipLaunchKernelGGL(fc_bwd_prep_kernel, dim3(cdiv(B * 32, 256)), dim3(256), 0, st, dzcat, eps, logvar, dmu_loss, dlv_loss, dml, B);
    CVAE_CHECK_LAUNCH();
    int rc = launch_colsum(dml, B, 64, dbfc, csws, st);
    if (rc) return rc;
    if (bf16io) hipLaunchKernelGGL(fc_bwd_dflat_kernel<__bf16>, dim3(cdiv(B, DF_IMGS), K / 256), dim3(256), 0, st, dml, wfc, dflat, B, K);
    else hipLaunchKernelGGL(fc_bwd_dflat_kernel<float>, dim3(cdiv(B, DF_IMGS), K / 256), dim3(256), 0, st, dml, wfc, dflat, B, K);
    CVAE_CHECK_LAUNCH();
    if (bf16io) {        // flat^T . dml on the bf16 MFMA
        BGemmArgs g{flat, dml, K, 64, 0, dwfc, 64, K, nullptr, B};
        hipLaunchKernelGGL((bgemm_tr_kernel<1, 2, false, true>), dim3(K / 32), dim3(256), 0, st, g);
    } else {             // fp32: flat^T . dml on the fp32 MFMA
        BGemmArgs g{flat, dml, K, 64, 0, dwfc, 64, K, nullptr, B};
        hipLaunchKernelGGL((bgemm_f32_kernel<1, 2, false>), dim3(K / 32), dim3(256), 0, st, g);
    }
    CVAE_CHECK_LAUNCH();
    return 0;
}
