// conv_epilogue.h — accumulator epilogues shared by the MFMA conv kernels.
#pragma once
#include "common.h"

enum { EPI_BIAS_BNSTAT = 0, EPI_BIAS_RELU = 1, EPI_PLAIN = 2, EPI_POOLSUM_MASK = 3 };

// Timing experiments (WRONG results, never shipped; profiles/experiments/variant.sh -DEPI_EXPERIMENT=n): which part of the
// epilogue a kernel's time sits in.  bit 0: no global stores of the tile;  bit 1: no BatchNorm partials;  bit 2: no LDS transpose
// (stores whatever the patch holds).
#ifndef EPI_EXPERIMENT
#define EPI_EXPERIMENT 0
#endif
// -DEPI_TIMING (timing builds): s_memtime stamps inside epilogue_store of the sampled workgroups (blockIdx.x a multiple of 64
// below 1024, blockIdx.y == 0; lane 0 of every wave), read back through cvae_epi_dbg_read: [16 workgroups][4 waves][32 stamps]
#ifdef EPI_TIMING
__device__ long long epi_dbg[16 * 4 * 32];
__device__ __forceinline__ void epi_stamp(int& k) {
    if ((blockIdx.x & 63) == 0 && blockIdx.x < 1024 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && k < 32) {
        __builtin_amdgcn_sched_barrier(0);
        epi_dbg[((blockIdx.x >> 6) * 4 + (threadIdx.x >> 6)) * 32 + k] = clock64();
        __builtin_amdgcn_sched_barrier(0);
    }
    ++k;
}
#define EPI_STAMP(k) epi_stamp(k)
#else
#define EPI_STAMP(k)
#endif

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

// The conv bias of this lane's output columns (n0 + 32 nb + li), fetched in the kernel PROLOGUE.  vmcnt counts a wave's
// loads and stores together and retires them in issue order: a bias value first used behind the previous block's global
// stores makes the wave sit out those stores' write acknowledgements (the round-3 epilogues did exactly that — `s_waitcnt
// vmcnt(0)` between the two 32-channel blocks of every tile and again at the second tile: 5 k of the 8.4 k epilogue
// cycles of the bf16 E2 forward workgroup).  Rule for every epilogue here: all loads are issued AND consumed before the
// first global store (profiles/experiments/vm_after_store.py lists violations from the hipcc -S listings).
template <int NT, int EPI>
__device__ __forceinline__ void load_bias(float (&bv)[NT / 32], const float* bias, int n0) {
    const int li = threadIdx.x & 31;
#pragma unroll
    for (int nb = 0; nb < NT / 32; ++nb) bv[nb] = (EPI == EPI_PLAIN) ? 0.f : bias[n0 + nb * 32 + li];
}
// bias (+ ReLU) in the accumulator registers, for every tile of the workgroup, before any of them is stored
template <int NT, int EPI>
__device__ __forceinline__ void apply_bias(f32x16 (&acc)[NT / 32], const float (&bv)[NT / 32]) {
    if constexpr (EPI == EPI_PLAIN) return;
#pragma unroll
    for (int nb = 0; nb < NT / 32; ++nb)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float x = acc[nb][v] + bv[nb];
            acc[nb][v] = (EPI == EPI_BIAS_RELU) ? fmaxf(x, 0.f) : x;
        }
}

// Store a 128-pixel x NT-channel tile held as NT/32 32x32 accumulators per wave (wave w = tile
// rows [32w, 32w+32); element v of lane (li, lh) is row (v&3)+8*(v>>2)+4*lh, column li) to an
// NHWC tensor — the accumulators already carry bias (+ ReLU), see apply_bias — and, for the encoder
// convs, emit the per-tile, per-channel BatchNorm partials (sum, M2 about the tile mean) that
// bn_fwd_finalize merges (train-mode batch statistics of nn.BatchNorm2d, vae_nets.py:70,75,80,85).
// No global load in here.  `smem` is reused (a barrier precedes the first write) and must hold
// >= max(8*NT, 4*32*36) floats.
template <int H, int NT, int NCH, int EPI, typename AT = float, bool PERM = false>      // AT: storage type of `out`; PERM: lane_pix
__device__ __forceinline__ void epilogue_store(f32x16 (&acc)[NT / 32], float* out,
                                               float* bnpart, float* smem, int B, int mt, int n0,
                                               int img0, int ty0, int tx0, int numTiles = -1, [[maybe_unused]] int ek = 0) {
    using T = Tile<H>;
    constexpr int NB = NT / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // Stores are issue-bound (one instruction per accumulator register moves only 4 B per lane), so each
    // wave transposes its 32x32 tile through a private LDS patch and writes 16 B per lane: 4 store
    // instructions per tile instead of 16 (cdna guide T21: the store tail is bound by instruction count).
    EPI_STAMP(ek);
    __syncthreads();                                   // every wave is done with the staging buffers
    EPI_STAMP(ek);
    float* patch = smem + wave * (32 * 36);            // [32 pixels][36] floats, rows 16-byte aligned
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        if (!(EPI_EXPERIMENT & 4)) {
#pragma unroll
        for (int v = 0; v < 16; ++v) patch[((v & 3) + 8 * (v >> 2) + 4 * lh) * 36 + li] = acc[nb][v];
        }
        EPI_STAMP(ek);
        if (EPI_EXPERIMENT & 1) continue;
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
        EPI_STAMP(ek);
    }
    if (EPI == EPI_BIAS_BNSTAT && !(EPI_EXPERIMENT & 2)) {
        __syncthreads();
        EPI_STAMP(ek);
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
        EPI_STAMP(ek);
        __syncthreads();
        EPI_STAMP(ek);
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
        EPI_STAMP(ek);
        __syncthreads();
        EPI_STAMP(ek);
        const size_t nt = numTiles < 0 ? gridDim.x : numTiles;     // multi-tile workgroups pass the true tile count
        if (tid < NT && (size_t)mt < nt) {
            const int c = tid;
            const float sum = red[c] + red[NT + c] + red[2 * NT + c] + red[3 * NT + c];
            const float m2 = red[4 * NT + c] + red[5 * NT + c] + red[6 * NT + c] + red[7 * NT + c];
            bnpart[(size_t)mt * NCH + n0 + c] = sum;
            bnpart[(nt + mt) * NCH + n0 + c] = m2;
        }
        EPI_STAMP(ek);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Channel-major accumulators (bf16 mode, round 4).  With the MFMA's operands swapped — weights as A, pixels as B — the
// 32x32 accumulator of a wave is D[channel][pixel]: lane (li, lh) holds ONE pixel (column li) and the 16 channels
// (v&3) + 8*(v>>2) + 4*lh of the block.  Nothing about the staging changes (the A and B fragments of
// v_mfma_f32_32x32x16_bf16 have the same register image: 8 consecutive k of row / column li), but the epilogue does:
//   * a lane owns 4 consecutive channels per register quad, so after one v_permlane32_swap per packed register pair it owns
//     two whole 16-byte units (channels 16k + 8lh .. +7) of its pixel: the tile leaves as global_store_dwordx4 straight from
//     registers — no LDS transpose patch (16 ds_write_b32 + 4 ds_read_b128 per block), no barrier in front of it;
//   * the BatchNorm partials (per-channel sums over the pixels) become a reduction across the 32 lanes of a wave half: a
//     halving butterfly (v_permlane16_swap, then DPP row_ror:8 / row_half_mirror / quad_perm) that adds 16 values per lane
//     down to one in 38 instructions — per WAVE, no barrier; the four waves' rows meet once, at the very end.
// The round-3 epilogue (transpose patch, two barrier-separated passes for the statistics) measured 8-10 k cycles of a 22 k-cycle
// E2-forward workgroup at ~15 cycles per instruction (profiles/r04_epilogue_timing.txt).
// ---------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
// v_permlane32_swap: lanes 32-63 of x <-> lanes 0-31 of y.  v_permlane16_swap: rows (16 lanes) 1 and 3 of x <-> rows 0 and 2 of y.
// Written as inline asm: hipcc 7.2 miscompiles the __builtin_amdgcn_permlane16_swap pair when both results feed one instruction
// (it emitted v_add v, x', x' — the second result replaced by the first; profiles/experiments/cm_probe.hip shows it).  The s_nop pads
// cover the VALU-write -> permlane-swap and permlane-swap -> VALU-read wait states the assembler does not insert for inline asm.
__device__ __forceinline__ void permlane32_swap(unsigned& x, unsigned& y) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ void permlane16_swap(float& x, float& y) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    bf16x2 p; p[0] = (__bf16)a; p[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, p);
}
// The lane's 16 accumulator values (channels (v&3) + 8(v>>2) + 4lh) -> its two 16-byte bf16 units: u[k] = channels 16k + 8lh .. +7.
__device__ __forceinline__ void cm_pack_units(const f32x16& acc, bf16x8 (&u)[2]) {
    unsigned p[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) { p[2 * g] = pack_bf16x2(acc[4 * g], acc[4 * g + 1]); p[2 * g + 1] = pack_bf16x2(acc[4 * g + 2], acc[4 * g + 3]); }
    // v_permlane32_swap(X, Y): lanes 32-63 of X <-> lanes 0-31 of Y.  X = quad 2j (lh0: channels 16j..+3 | lh1: 16j+4..+7),
    // Y = quad 2j+1 (lh0: 16j+8..+11 | lh1: 16j+12..+15)  ->  X = (lh0: 16j..+3 | lh1: 16j+8..+11), Y = (lh0: 16j+4..+7 | lh1: 16j+12..+15)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) permlane32_swap(p[4 * j + h], p[4 * j + 2 + h]);
    u[0] = __builtin_bit_cast(bf16x8, u32x4{p[0], p[1], p[2], p[3]});
    u[1] = __builtin_bit_cast(bf16x8, u32x4{p[4], p[5], p[6], p[7]});
}
// Sum x[j] (j = 0..15) over the 32 lanes of each wave half, 38 instructions: returns in lane L the total of element
// e = (L & 31) >> 1 (lanes L and L^1 hold the same element).  Fixed pairing, fixed order: bitwise reproducible.
__device__ __forceinline__ float half_wave_colsum16(const float (&x)[16]) {
    const int lane = threadIdx.x & 63;
    float r8[8], r4[4], r2[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {          // lanes i, i + 16: rows 1 / 3 of x[j] <-> rows 0 / 2 of x[j + 8]; afterwards element j + 8 * bit4
        float a = x[j], b = x[j + 8];
        permlane16_swap(a, b);
        r8[j] = a + b;
    }
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)            // partner i ^ 8 (row_ror:8): keep element j + 4 * bit3, hand the other one over
        r4[j] = (b3 ? r8[j + 4] : r8[j]) + dpp_mov<0x128>(b3 ? r8[j] : r8[j + 4]);
#pragma unroll
    for (int j = 0; j < 2; ++j)            // partner 7 - i within the 8 lanes (row_half_mirror: bit2 flipped): element j + 2 * bit2
        r2[j] = (b2 ? r4[j + 2] : r4[j]) + dpp_mov<0x141>(b2 ? r4[j] : r4[j + 2]);
    const float r1 = (b1 ? r2[1] : r2[0]) + dpp_mov<0x4E>(b1 ? r2[0] : r2[1]);        // partner i ^ 2 (quad_perm 2,3,0,1): element bit1
    return r1 + dpp_mov<0xB1>(r1);                                                     // partner i ^ 1 (quad_perm 1,0,3,2): same element
}
