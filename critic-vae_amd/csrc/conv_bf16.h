// conv_bf16.h — declarations shared by conv_bf16.hip, conv_bf16_ps.hip and conv_wgrad_split.hip (bf16-MFMA conv kernels)
#pragma once
#include "common.h"

struct ConvBf16Args {
    const float* in;        // NHWC, KCH channels: bf16 when NS == 1 (opaque pointer), fp32 when NS == 3
    const bf16x8* wp;       // packed weights [25][KCH/16][2][NCH] units of 8 bf16
    const float* bias;
    float* out;
    float* bnpart;
    int B;
    int64_t sliceFloats;    // KSPLIT > 1: out = slab [KSPLIT][sliceFloats]
    const float* aux;       // MODE_UP_DGRAD: forward output of the producing layer (ReLU mask), layout of `out`
    int64_t splitStride;    // NS == 3: units between the hi / mid / lo copies of the packed weights
    int products;           // NS == 3: 9 (all partial products) or 6
};


template <int H, int OCT> struct Bf16Geom {
    // halo plane stride in 16-byte units: >= HP and == 16/OCT (mod 16) so that the 16 lanes of one b128
    // write phase (OCT octets x 16/OCT pixels) land in 16 different 16-byte bank groups
    static constexpr int PAD = OCT >= 8 ? 2 : (OCT == 4 ? 4 : 8);
    static constexpr int PSP = ((Tile<H>::HP + 15 - PAD) / 16) * 16 + PAD;
};


// conv_bf16_ps.hip: persistent forward / input-gradient kernel of E2..E4 in bf16 mode (returns -100 if the layer has no instantiation)
int launch_conv_bf16_ps(int layer, int width, bool dgrad, const ConvBf16Args& a, hipStream_t st);
// conv_bf16_big.hip: persistent big-tile kernel (16 accumulator tiles per wave; `mask`: the layer bits of CVAE_BF16_BIG for this pass); -100 if the
// layer has no instantiation, is masked out, or a tensor is too large for its 32-bit offsets
int launch_conv_bf16_big(int layer, int width, bool dgrad, int mask, const ConvBf16Args& a, hipStream_t st);
bool conv_bf16_big_has(int layer, int width, bool dgrad, int mask);
int conv_bf16_big_tiles(int layer, int width, bool dgrad);      // 128-pixel tiles per item (= per BatchNorm partial of its forward passes)

// Exact 3-way bf16 split of an fp32 value: x == hi + mid + lo (each difference is exact in fp32, RNE
// leaves at most 8 significant bits per step), used by the fp32-emulation mode (NS == 3): the nine
// bf16 x bf16 partial products of two split operands are exact in the fp32 accumulator, so only the
// summation order differs from an fp32 fma chain.
struct Split3 { __bf16 hi, mid, lo; };
__device__ __forceinline__ Split3 split3(float x) {
    Split3 s;
    s.hi = (__bf16)x;
    const float r = x - (float)s.hi;
    s.mid = (__bf16)r;
    s.lo = (__bf16)(r - (float)s.mid);
    return s;
}


// tile geometry of the transposed-read weight-gradient kernels (conv_bf16.hip: conv5x5_wgrad_tr_kernel, conv_up_wgrad_bf16_kernel;
// conv_wgrad_split.hip)
template <int H, int HALO = 2, int NP = 256> struct WtTile {          // HALO 2 / 256 pixels: 5x5 layers;  HALO 1 / 128: collapsed up-convs
    static constexpr int TW = H < 32 ? H : 32;
    static constexpr int TH = H < NP / TW ? H : NP / TW;
    static constexpr int IMGS = H == 4 ? (NP >= 128 ? 8 : NP / 16) : NP / (TW * TH);
    static constexpr int NPX = IMGS * TH * TW, KG = NPX / 16;
    static constexpr int HTW = TW + 2 * HALO, HTH = TH + 2 * HALO, HPI = HTW * HTH, HP = IMGS * HPI;
    static constexpr int TILES_X = H / TW, TPI = TILES_X * (H / TH);
    // lane half h (k = 8h..8h+7) and the second read t (k += 4) move by these many dy pixels / halo pixels
    static constexpr int DH = 8, DT = 4;
    static constexpr int IH = TW >= 16 ? 8 : (TW == 8 ? HTW : 2 * HTW), IT = TW >= 8 ? 4 : HTW;
    static constexpr int pixbase(int kg) {
        return TW >= 16 ? (kg / (TW / 16)) * TW + (kg % (TW / 16)) * 16 : (TW == 8 ? (kg / 4) * 64 + (kg % 4) * 16 : kg * 16);
    }
    static constexpr int halobase(int kg) {
        return TW >= 16 ? (kg / (TW / 16)) * HTW + (kg % (TW / 16)) * 16
                        : (TW == 8 ? (kg / 4) * HPI + (kg % 4) * 2 * HTW : kg * HPI);
    }
};

struct WgradBf16Args {
    const float* in;     // (B,H,H,CIN) bf16 (opaque pointer)
    const float* dout;   // (B,H,H,COUT) bf16
    float* slab;         // [S][25*CIN*COUT + COUT]
    int B, numTiles, tilesPerSplit;
};

