// conv_bf16.h — declarations shared by conv_bf16.hip and conv_bf16_ps.hip (bf16-MFMA conv kernels, precision mode 1)
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
