// conv_epilogue.h — accumulator epilogues shared by the MFMA conv kernels.
#pragma once
#include "common.h"

enum { EPI_BIAS_BNSTAT = 0, EPI_BIAS_RELU = 1, EPI_PLAIN = 2, EPI_POOLSUM_MASK = 3 };

// Row r (0..31) of a wave's 32-pixel MFMA sub-tile -> pixel of that sub-tile (row-major tile order).  Identity for
// the fp32 kernels.  The bf16 kernels read their A fragments as 16-byte units `[octet][halo pixel]` with
// ds_read_b128, whose 16-lane groups are {0-3,12-15,20-27} and {4-11,16-19,28-31}: with halo rows TW+4 units apart,
// the identity map puts two lanes of a group on the same 16-byte bank slot for the 16- and 8-pixel-wide tiles
// (25-38 % of the LDS cycles of those kernels were conflicts).  These permutations give every group 16 distinct slots.
template <int H, bool PERM>
__device__ __forceinline__ int lane_pix(int r) {
    if constexpr (!PERM) return r;
    else if constexpr (Tile<H>::TW == 16) {        // rows of 16: second row takes columns 8-11 | 0-7 | 12-15
        if (r < 16) return r;
        const int j = r - 16;
        return 16 + (j < 4 ? j + 8 : (j < 12 ? j - 4 : j));
    } else if constexpr (Tile<H>::TW == 8) {       // rows of 8: group A gets rows 0 and 2, group B rows 1 and 3
        return r < 4 ? r : (r < 12 ? r + 4 : (r < 16 ? r - 8 : (r < 20 ? r + 8 : (r < 28 ? r - 4 : r))));
    } else return r;
}

// Store a 128-pixel x NT-channel tile held as NT/32 32x32 accumulators per wave (wave w = tile
// rows [32w, 32w+32); element v of lane (li, lh) is row (v&3)+8*(v>>2)+4*lh, column li) to an
// NHWC tensor, adding the bias (and ReLU), and — for the encoder convs — emit the per-tile,
// per-channel BatchNorm partials (sum, M2 about the tile mean) that bn_fwd_finalize merges
// (train-mode batch statistics of nn.BatchNorm2d, vae_nets.py:70,75,80,85).
// `smem` is reused (a barrier precedes the first write) and must hold >= max(8*NT, 4*32*36) floats.
template <int H, int NT, int NCH, int EPI, typename AT = float, bool PERM = false>      // AT: storage type of `out`; PERM: lane_pix
__device__ __forceinline__ void epilogue_store(f32x16 (&acc)[NT / 32], const float* bias, float* out,
                                               float* bnpart, float* smem, int B, int mt, int n0,
                                               int img0, int ty0, int tx0, int numTiles = -1) {
    using T = Tile<H>;
    constexpr int NB = NT / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    float bv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bv[nb] = (EPI == EPI_PLAIN) ? 0.f : bias[n0 + nb * 32 + li];
    // Stores are issue-bound (one instruction per accumulator register moves only 4 B per lane), so each
    // wave transposes its 32x32 tile through a private LDS patch and writes 16 B per lane: 4 store
    // instructions per tile instead of 16 (cdna guide T21: the store tail is bound by instruction count).
    __syncthreads();                                   // every wave is done with the staging buffers
    float* patch = smem + wave * (32 * 36);            // [32 pixels][36] floats, rows 16-byte aligned
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float x = acc[nb][v] + bv[nb];
            if (EPI == EPI_BIAS_RELU) x = fmaxf(x, 0.f);
            acc[nb][v] = x;
            patch[((v & 3) + 8 * (v >> 2) + 4 * lh) * 36 + li] = x;
        }
        if constexpr (Act<AT>::BF16) {        // 8 channels = one 16-byte unit per lane: 2 store instructions per tile
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = it * 64 + lane, px = idx >> 2, c8 = idx & 3;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + px * 36 + c8 * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + px * 36 + c8 * 8 + 4);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (__bf16)lo[e]; o[4 + e] = (__bf16)hi[e]; }
                const int mm = wave * 32 + lane_pix<H, PERM>(px);
                const int im = mm / (T::TH * T::TW), rem = mm % (T::TH * T::TW);
                const int gy = ty0 + rem / T::TW, gx = tx0 + rem % T::TW, ib = img0 + im;
                if (ib < B) Act<AT>::st8(out, ((size_t)(ib * H + gy) * H + gx) * NCH + n0 + nb * 32 + c8 * 8, o);
            }
        } else {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 64 + lane, px = idx >> 3, c4 = idx & 7;
            const float4 val = *reinterpret_cast<const float4*>(patch + px * 36 + c4 * 4);
            const int mm = wave * 32 + lane_pix<H, PERM>(px);
            const int im = mm / (T::TH * T::TW), rem = mm % (T::TH * T::TW);
            const int gy = ty0 + rem / T::TW, gx = tx0 + rem % T::TW, ib = img0 + im;
            if (ib < B)
                *reinterpret_cast<float4*>(out + ((size_t)(ib * H + gy) * H + gx) * NCH + n0 + nb * 32 + c4 * 4) = val;
        }
        }
    }
    if (EPI == EPI_BIAS_BNSTAT) {
        __syncthreads();
        float* red = smem;                       // [2][4][NT]
        int nvalid_img = B - img0;
        if (nvalid_img > T::IMGS) nvalid_img = T::IMGS;
        const float cnt = (float)(nvalid_img * T::TH * T::TW);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float t = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int mm = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                if (img0 + mm / (T::TH * T::TW) < B) t += acc[nb][v];
            }
            t += __shfl_xor(t, 32, 64);
            if (lh == 0) red[wave * NT + nb * 32 + li] = t;
        }
        __syncthreads();
        float mean[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int c = nb * 32 + li;
            mean[nb] = (red[c] + red[NT + c] + red[2 * NT + c] + red[3 * NT + c]) / cnt;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float t = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int mm = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                const float d = acc[nb][v] - mean[nb];
                if (img0 + mm / (T::TH * T::TW) < B) t += d * d;
            }
            t += __shfl_xor(t, 32, 64);
            if (lh == 0) red[4 * NT + wave * NT + nb * 32 + li] = t;
        }
        __syncthreads();
        const size_t nt = numTiles < 0 ? gridDim.x : numTiles;     // multi-tile workgroups pass the true tile count
        if (tid < NT && (size_t)mt < nt) {
            const int c = tid;
            const float sum = red[c] + red[NT + c] + red[2 * NT + c] + red[3 * NT + c];
            const float m2 = red[4 * NT + c] + red[5 * NT + c] + red[6 * NT + c] + red[7 * NT + c];
            bnpart[(size_t)mt * NCH + n0 + c] = sum;
            bnpart[(nt + mt) * NCH + n0 + c] = m2;
        }
    }
}
