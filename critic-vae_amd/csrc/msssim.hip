// msssim.hip — fused MS-SSIM pyramid loss + KLD, forward and backward, on NCHW planes.
//
// Replaces MSSIM.forward / MSSIM.ssim (vae_nets.py:181-247: per level 5 depthwise 11x11
// F.conv2d + ~12 elementwise ATen ops + 2 means + avg_pool2d, window rebuilt on every call),
// the KLD of VariationalAutoencoder.vae_loss (vae_nets.py:57-60) and their autograd.
//
// Semantics kept exactly: the window is exp(+d^2/2s^2)/sum ("anti-Gaussian",
// vae_nets.py:171), zero padding 5 at every level, C1=1e-4, C2=9e-4, GLOBAL means over
// B*3*H*W, final 1 - prod_{l<4}(cs_l^w_l * ssim_4^w_4); negative cs -> NaN, no clamping.
// The 2-D window is the outer product of the 1-D one (vae_nets.py:176-177) so the filter is run
// as 11+11 taps (summation order differs from the 121-tap ATen conv by ~1e-7 relative).
//
// Per level one forward kernel: horizontal pass of {x, y, x^2, y^2, xy} into LDS, vertical pass,
// SSIM/CS maps, per-workgroup partial sums, the 2x2 average for the next level, and — for the
// backward — the three derivative maps of the level's contributing map w.r.t. (mu1, E[x^2],
// E[xy]).  One finalize workgroup reduces everything in fp64 in a fixed order, evaluates the
// loss, the KLD (+ its gradients) and the per-level gradient coefficients; the backward kernels
// run the same separable filter over the derivative maps, top level first, adding the
// average-pool backward of the level above.  Only cs_0..3 and ssim_4 carry gradient.
#include "common.h"
#include <math.h>

// The 11 normalised taps travel BY VALUE in every kernel's argument struct (scalar registers): no
// __constant__ symbol, no upload, no process-global device state — handles on different devices are
// independent and nothing here synchronises.  gaussian_window (vae_nets.py:169-173): fp32 exp, fp32 sum.
struct MsWin { float w[11]; };
static MsWin make_window() {
    MsWin m;
    float s = 0.f;
    for (int i = 0; i < 11; ++i) { m.w[i] = (float)exp((double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += m.w[i]; }
    for (int i = 0; i < 11; ++i) m.w[i] = m.w[i] / s;
    return m;
}

template <int S>
struct MsGeom {
    static constexpr int RS = S < 16 ? S : 16;                    // rows per workgroup
    static constexpr int PPB = (RS * S >= 256) ? 1 : 256 / (RS * S);   // planes per workgroup
    static constexpr int HR = RS + 10, HC = S + 10;
    static constexpr int STRIPS = S / RS;
};

// zero-padded load of NMAP planes' strips into LDS: dst[map][plane][HR][HC]
template <int S, int NMAP>
__device__ __forceinline__ void ms_load(const float* const (&src)[NMAP], float* dst, int plane0, int P, int r0) {
    using G = MsGeom<S>;
    constexpr int PER = G::PPB * G::HR * G::HC;
    if constexpr (S >= 32) {          // one plane per workgroup: 16-byte row loads of the interior + zeroed 5-column borders
        constexpr int C4 = S / 4;
        const bool plane_ok = plane0 < P;
        for (int q = threadIdx.x; q < G::HR * C4; q += 256) {
            const int rr = q / C4, c = (q % C4) * 4, r = r0 - 5 + rr;
            const bool ok = plane_ok && (unsigned)r < (unsigned)S;
#pragma unroll
            for (int mI = 0; mI < NMAP; ++mI) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4*>(src[mI] + ((size_t)plane0 * S + r) * S + c);
                float* d = dst + mI * PER + rr * G::HC + 5 + c;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
        for (int q = threadIdx.x; q < G::HR * 10; q += 256) {
            const int rr = q / 10, k = q % 10, col = k < 5 ? k : S + k;       // columns 0..4 and S+5..S+9
#pragma unroll
            for (int mI = 0; mI < NMAP; ++mI) dst[mI * PER + rr * G::HC + col] = 0.f;
        }
        return;
    }
    for (int q = threadIdx.x; q < PER; q += 256) {
        const int pl = q / (G::HR * G::HC), rem = q % (G::HR * G::HC);
        const int r = r0 - 5 + rem / G::HC, c = rem % G::HC - 5;
        const bool ok = (unsigned)r < (unsigned)S && (unsigned)c < (unsigned)S && plane0 + pl < P;
#pragma unroll
        for (int mI = 0; mI < NMAP; ++mI)
            dst[mI * PER + q] = ok ? src[mI][((size_t)(plane0 + pl) * S + r) * S + c] : 0.f;
    }
}

struct MsFwdArgs {
    const float* x;       // img1 at this level (carries grad)
    const float* y;       // img2
    float* nx; float* ny; // next level (2x2 averages) or null
    float* um; float* u11; float* u12;   // derivative maps or null
    float* part;          // [numBlocks][2]
    int P;
    int last;             // level 4: the contributing map is ssim_map instead of cs_map
    MsWin win;
};

template <int S>
__global__ __launch_bounds__(256) void msssim_fwd_kernel(MsFwdArgs a) {
    using G = MsGeom<S>;
    constexpr int PER_IN = G::PPB * G::HR * G::HC, PER_T = G::PPB * G::HR * S;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // 2*PER_IN + 5*PER_T floats
    __shared__ float red[8];
    float* lin = smem;
    float* tmp = smem + 2 * PER_IN;
    const int plane0 = (blockIdx.x / G::STRIPS) * G::PPB, r0 = (blockIdx.x % G::STRIPS) * G::RS;
    const float* const srcs[2] = {a.x, a.y};
    ms_load<S, 2>(srcs, lin, plane0, a.P, r0);
    __syncthreads();
    float w[11];
#pragma unroll
    for (int t = 0; t < 11; ++t) w[t] = a.win.w[t];
    // horizontal pass.  Large levels (one plane per workgroup): 4 adjacent outputs per work item from
    // a 14-wide register window (3.5 LDS reads per output instead of 22); same fma order per output.
    if constexpr (S >= 32) {
        constexpr int CG = S / 4;
        for (int it = threadIdx.x; it < G::HR * CG; it += 256) {
            const int r = it / CG, c = (it % CG) * 4;
            const float* px = lin + r * G::HC + c;
            const float* py = px + PER_IN;
            float xs[14], ys[14];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const float2 u = *reinterpret_cast<const float2*>(px + 2 * i);
                const float2 v = *reinterpret_cast<const float2*>(py + 2 * i);
                xs[2 * i] = u.x; xs[2 * i + 1] = u.y; ys[2 * i] = v.x; ys[2 * i + 1] = v.y;
            }
            f32x4 o0, o1, o2, o3, o4;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float hx = 0.f, hy = 0.f, hxx = 0.f, hyy = 0.f, hxy = 0.f;
#pragma unroll
                for (int t = 0; t < 11; ++t) {
                    const float xv = xs[o + t], yv = ys[o + t];
                    hx = fmaf(w[t], xv, hx); hy = fmaf(w[t], yv, hy);
                    hxx = fmaf(w[t], xv * xv, hxx); hyy = fmaf(w[t], yv * yv, hyy); hxy = fmaf(w[t], xv * yv, hxy);
                }
                o0[o] = hx; o1[o] = hy; o2[o] = hxx; o3[o] = hyy; o4[o] = hxy;
            }
            float* d = tmp + r * S + c;
            *reinterpret_cast<f32x4*>(d) = o0; *reinterpret_cast<f32x4*>(d + PER_T) = o1;
            *reinterpret_cast<f32x4*>(d + 2 * PER_T) = o2; *reinterpret_cast<f32x4*>(d + 3 * PER_T) = o3;
            *reinterpret_cast<f32x4*>(d + 4 * PER_T) = o4;
        }
    } else {
    for (int q = threadIdx.x; q < PER_T; q += 256) {
        const int pl = q / (G::HR * S), rem = q % (G::HR * S), r = rem / S, c = rem % S;
        const float* px = lin + (pl * G::HR + r) * G::HC + c;
        const float* py = px + PER_IN;
        float hx = 0.f, hy = 0.f, hxx = 0.f, hyy = 0.f, hxy = 0.f;
#pragma unroll
        for (int t = 0; t < 11; ++t) {
            const float xv = px[t], yv = py[t];
            hx = fmaf(w[t], xv, hx); hy = fmaf(w[t], yv, hy);
            hxx = fmaf(w[t], xv * xv, hxx); hyy = fmaf(w[t], yv * yv, hyy); hxy = fmaf(w[t], xv * yv, hxy);
        }
        tmp[q] = hx; tmp[PER_T + q] = hy; tmp[2 * PER_T + q] = hxx; tmp[3 * PER_T + q] = hyy; tmp[4 * PER_T + q] = hxy;
    }
    }
    __syncthreads();
    // vertical pass + maps
    const float C1 = 0.0001f, C2 = 0.0009f;
    float s_ssim = 0.f, s_cs = 0.f;
    auto pixel = [&](int pl, int r, int c, float mu1, float mu2, float a11, float a22, float a12) {
        const float mu1sq = mu1 * mu1, mu2sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float v1 = 2.0f * (a12 - mu12) + C2;
        const float v2 = (a11 - mu1sq) + (a22 - mu2sq) + C2;
        const float cs = v1 / v2;
        const float num = 2.0f * mu12 + C1, den = mu1sq + mu2sq + C1;
        const float lum = num / den;
        s_cs += cs;
        s_ssim += (num * v1) / (den * v2);
        if (a.um) {
            // d cs / d(mu1, A11, A12);  level 4: d (lum*cs) / d(...)
            const float inv2 = 1.0f / v2;
            float dm = (2.0f * mu1 * cs - 2.0f * mu2) * inv2;
            float d11 = -cs * inv2;
            float d12 = 2.0f * inv2;
            if (a.last) {
                const float dlum = (2.0f * mu2 - 2.0f * mu1 * lum) / den;
                dm = dlum * cs + lum * dm; d11 *= lum; d12 *= lum;
            }
            const size_t o = ((size_t)(plane0 + pl) * S + r0 + r) * S + c;
            a.um[o] = dm; a.u11[o] = d11; a.u12[o] = d12;
        }
    };
    if constexpr (S >= 32) {       // 4 vertically adjacent outputs per work item from a 14-deep register window
        if (plane0 < a.P) {
            for (int it = threadIdx.x; it < (G::RS / 4) * S; it += 256) {
                const int c = it % S, rb = (it / S) * 4;
                float res[5][4];
#pragma unroll
                for (int mI = 0; mI < 5; ++mI) {
                    const float* t0 = tmp + mI * PER_T + rb * S + c;
                    float v[14];
#pragma unroll
                    for (int i = 0; i < 14; ++i) v[i] = t0[i * S];
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        float acc = 0.f;
#pragma unroll
                        for (int t = 0; t < 11; ++t) acc = fmaf(w[t], v[o + t], acc);
                        res[mI][o] = acc;
                    }
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) pixel(0, rb + o, c, res[0][o], res[1][o], res[2][o], res[3][o], res[4][o]);
            }
        }
    } else {
    for (int q = threadIdx.x; q < G::PPB * G::RS * S; q += 256) {
        const int pl = q / (G::RS * S), rem = q % (G::RS * S), r = rem / S, c = rem % S;
        if (plane0 + pl >= a.P) continue;
        const float* t0 = tmp + (pl * G::HR + r) * S + c;
        float mu1 = 0.f, mu2 = 0.f, a11 = 0.f, a22 = 0.f, a12 = 0.f;
#pragma unroll
        for (int t = 0; t < 11; ++t) {
            mu1 = fmaf(w[t], t0[t * S], mu1); mu2 = fmaf(w[t], t0[PER_T + t * S], mu2);
            a11 = fmaf(w[t], t0[2 * PER_T + t * S], a11); a22 = fmaf(w[t], t0[3 * PER_T + t * S], a22);
            a12 = fmaf(w[t], t0[4 * PER_T + t * S], a12);
        }
        pixel(pl, r, c, mu1, mu2, a11, a22, a12);
    }
    }
    s_ssim = wave_sum(s_ssim); s_cs = wave_sum(s_cs);
    if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = s_ssim; red[(threadIdx.x >> 6) * 2 + 1] = s_cs; }
    // 2x2 average for the next level (avg_pool2d, vae_nets.py:232-233)
    if (a.nx) {
        constexpr int SO = S / 2;
        for (int q = threadIdx.x; q < G::PPB * (G::RS / 2) * SO; q += 256) {
            const int pl = q / ((G::RS / 2) * SO), rem = q % ((G::RS / 2) * SO), pr = rem / SO, pc = rem % SO;
            if (plane0 + pl >= a.P) continue;
            const float* px = lin + (pl * G::HR + 2 * pr + 5) * G::HC + 2 * pc + 5;
            const float* py = px + PER_IN;
            const size_t o = ((size_t)(plane0 + pl) * SO + r0 / 2 + pr) * SO + pc;
            a.nx[o] = ((px[0] + px[1]) + (px[G::HC] + px[G::HC + 1])) * 0.25f;
            a.ny[o] = ((py[0] + py[1]) + (py[G::HC] + py[G::HC + 1])) * 0.25f;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.part[blockIdx.x * 2] = (red[0] + red[2]) + (red[4] + red[6]);
        a.part[blockIdx.x * 2 + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}

struct MsFinArgs {
    const float* part;       // all levels' partials, level l at partOff[l], nblk[l] pairs
    int partOff[5], nblk[5];
    double count[5];         // B*3*S_l*S_l
    const float* mu; const float* logvar; int B;
    float* scalars;          // CVAE_N_SCALARS
    float* coef;             // [5] per-pixel gradient coefficient of each level's map
    float* d_mu; float* d_logvar;
};

__global__ __launch_bounds__(1024) void msssim_finalize_kernel(MsFinArgs a) {
    __shared__ double red[11][16];
    // every thread accumulates all 11 sums (5 x ssim, 5 x cs, KLD) over its strided share, then ONE
    // block reduction (wave shuffles + 16 wave partials), all in a fixed order
    double acc[11];
#pragma unroll
    for (int q = 0; q < 11; ++q) acc[q] = 0.0;
#pragma unroll
    for (int l = 0; l < 5; ++l)
        for (int i = threadIdx.x; i < a.nblk[l]; i += 1024) {
            acc[l] += (double)a.part[a.partOff[l] + i * 2];
            acc[5 + l] += (double)a.part[a.partOff[l] + i * 2 + 1];
        }
    const float kw = 0.001f, invB = a.B > 0 ? 1.0f / (float)a.B : 0.f;
    for (int i = threadIdx.x; i < a.B * 32; i += 1024) {
        const float m = a.mu[i], lv = a.logvar[i], e = expf(lv);
        acc[10] += (double)(1.0f + lv - m * m - e);
        if (a.d_mu) { a.d_mu[i] = kw * m * invB; a.d_logvar[i] = kw * 0.5f * (e - 1.0f) * invB; }
    }
#pragma unroll
    for (int q = 0; q < 11; ++q) {
        const double v = wave_sum_d(acc[q]);
        if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        // lanes 0-4: ssim_l, lanes 5-9: cs_l, lane 10: KLD sum — every lane finishes its own scalar
        // (same operations and order as a serial evaluation), shuffles bring them together
        const int lane = threadIdx.x, l = lane % 5;
        const float wts[5] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
        double t = 0.0;
        if (lane < 11) for (int wv = 0; wv < 16; ++wv) t += red[lane][wv];
        const double cnt = a.count[l];
        const float meanf = (float)(t / cnt);
        const float pw = powf(meanf, wts[l]);
        const float p2 = __shfl(pw, 4, 64);
        float out = 1.0f;
        for (int q = 0; q < 4; ++q) out *= __shfl(pw, 5 + q, 64) * p2;              // vae_nets.py:243-246
        const float recon = 1.0f - out;
        const double k = __shfl(t, 10, 64);
        const float kld = a.B > 0 ? (float)(-0.5 * k / (double)a.B) * kw : 0.0f;
        if (lane == 0) { a.scalars[0] = recon + kld; a.scalars[1] = recon; a.scalars[2] = kld; }
        if (lane < 5) a.scalars[3 + lane] = meanf;
        else if (lane < 10) a.scalars[8 + l] = meanf;
        else if (lane < 13) a.scalars[13 + lane - 10] = 0.f;
        if (lane >= 5 && lane < 9) a.coef[l] = (float)((double)(-out * wts[l] / meanf) / cnt);
        if (lane == 4) a.coef[4] = (float)((double)(-out * 4.0f * wts[4] / meanf) / cnt);
    }
}

struct MsBwdArgs {
    const float* um; const float* u11; const float* u12;
    const float* x; const float* y;
    const float* gup;     // gradient of the level above (S/2) or null
    const float* coef;    // device scalar for this level
    float* dx;
    int P;
    MsWin win;
};

template <int S>
__global__ __launch_bounds__(256) void msssim_bwd_kernel(MsBwdArgs a) {
    using G = MsGeom<S>;
    constexpr int PER_IN = G::PPB * G::HR * G::HC, PER_T = G::PPB * G::HR * S;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // 3*PER_IN + 3*PER_T floats
    float* lin = smem;
    float* tmp = smem + 3 * PER_IN;
    const int plane0 = (blockIdx.x / G::STRIPS) * G::PPB, r0 = (blockIdx.x % G::STRIPS) * G::RS;
    const float* const srcs[3] = {a.um, a.u11, a.u12};
    ms_load<S, 3>(srcs, lin, plane0, a.P, r0);
    __syncthreads();
    float w[11];
#pragma unroll
    for (int t = 0; t < 11; ++t) w[t] = a.win.w[t];
    if constexpr (S >= 32) {
        constexpr int CG = S / 4;
        for (int it = threadIdx.x; it < G::HR * CG; it += 256) {
            const int r = it / CG, c = (it % CG) * 4;
#pragma unroll
            for (int mI = 0; mI < 3; ++mI) {
                const float* p0 = lin + mI * PER_IN + r * G::HC + c;
                float xs[14];
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    const float2 u = *reinterpret_cast<const float2*>(p0 + 2 * i);
                    xs[2 * i] = u.x; xs[2 * i + 1] = u.y;
                }
                f32x4 o4v;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float h = 0.f;
#pragma unroll
                    for (int t = 0; t < 11; ++t) h = fmaf(w[t], xs[o + t], h);
                    o4v[o] = h;
                }
                *reinterpret_cast<f32x4*>(tmp + mI * PER_T + r * S + c) = o4v;
            }
        }
    } else {
    for (int q = threadIdx.x; q < PER_T; q += 256) {
        const int pl = q / (G::HR * S), rem = q % (G::HR * S), r = rem / S, c = rem % S;
        const float* p0 = lin + (pl * G::HR + r) * G::HC + c;
        float h0 = 0.f, h1 = 0.f, h2 = 0.f;
#pragma unroll
        for (int t = 0; t < 11; ++t) {
            h0 = fmaf(w[t], p0[t], h0); h1 = fmaf(w[t], p0[PER_IN + t], h1); h2 = fmaf(w[t], p0[2 * PER_IN + t], h2);
        }
        tmp[q] = h0; tmp[PER_T + q] = h1; tmp[2 * PER_T + q] = h2;
    }
    }
    __syncthreads();
    const float coef = a.coef[0];
    auto emit = [&](int pl, int r, int c, float f0, float f1, float f2) {
        const size_t o = ((size_t)(plane0 + pl) * S + r0 + r) * S + c;
        float g = coef * (f0 + 2.0f * a.x[o] * f1 + a.y[o] * f2);
        if (a.gup) g += 0.25f * a.gup[((size_t)(plane0 + pl) * (S / 2) + (r0 + r) / 2) * (S / 2) + c / 2];
        a.dx[o] = g;
    };
    if constexpr (S >= 32) {
        if (plane0 < a.P) {
            for (int it = threadIdx.x; it < (G::RS / 4) * S; it += 256) {
                const int c = it % S, rb = (it / S) * 4;
                float res[3][4];
#pragma unroll
                for (int mI = 0; mI < 3; ++mI) {
                    const float* t0 = tmp + mI * PER_T + rb * S + c;
                    float v[14];
#pragma unroll
                    for (int i = 0; i < 14; ++i) v[i] = t0[i * S];
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        float acc = 0.f;
#pragma unroll
                        for (int t = 0; t < 11; ++t) acc = fmaf(w[t], v[o + t], acc);
                        res[mI][o] = acc;
                    }
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) emit(0, rb + o, c, res[0][o], res[1][o], res[2][o]);
            }
        }
    } else {
    for (int q = threadIdx.x; q < G::PPB * G::RS * S; q += 256) {
        const int pl = q / (G::RS * S), rem = q % (G::RS * S), r = rem / S, c = rem % S;
        if (plane0 + pl >= a.P) continue;
        const float* t0 = tmp + (pl * G::HR + r) * S + c;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f;
#pragma unroll
        for (int t = 0; t < 11; ++t) {
            f0 = fmaf(w[t], t0[t * S], f0); f1 = fmaf(w[t], t0[PER_T + t * S], f1); f2 = fmaf(w[t], t0[2 * PER_T + t * S], f2);
        }
        emit(pl, r, c, f0, f1, f2);
    }
    }
}

template <int S> static int ms_blocks(int P) { using G = MsGeom<S>; return cdiv(P, G::PPB) * G::STRIPS; }

// workspace carve (floats).  sizes for width W: level sizes W, W/2, .., W/16.
struct MsWs {
    int64_t pyrx[5], pyry[5], um[5], u11[5], u12[5], gp[5], part[5], coef, total;
    int nblk[5];
};
static MsWs ms_carve(int width, int B) {
    MsWs w{};
    const int P = B * 3;
    int64_t off = 0;
    auto take = [&](int64_t n) { int64_t o = off; off += align_up(n, 64); return o; };
    for (int l = 0; l < 5; ++l) {
        const int S = width >> l;
        const int64_t n = (int64_t)P * S * S;
        if (l > 0) { w.pyrx[l] = take(n); w.pyry[l] = take(n); w.gp[l] = take(n); }
        w.um[l] = take(n); w.u11[l] = take(n); w.u12[l] = take(n);
    }
    for (int l = 0; l < 5; ++l) {
        switch (width >> l) {
            case 128: w.nblk[l] = ms_blocks<128>(P); break;
            case 64: w.nblk[l] = ms_blocks<64>(P); break;
            case 32: w.nblk[l] = ms_blocks<32>(P); break;
            case 16: w.nblk[l] = ms_blocks<16>(P); break;
            case 8: w.nblk[l] = ms_blocks<8>(P); break;
            default: w.nblk[l] = ms_blocks<4>(P); break;
        }
    }
    for (int l = 0; l < 5; ++l) w.part[l] = take((int64_t)w.nblk[l] * 2);
    w.coef = take(8);
    w.total = off;
    return w;
}
int64_t msssim_ws_floats(int width, int B) { return ms_carve(width, B).total; }

template <int S>
static int ms_fwd(const MsFwdArgs& a, hipStream_t st) {
    using G = MsGeom<S>;
    constexpr int SMEM = (2 * G::PPB * G::HR * G::HC + 5 * G::PPB * G::HR * S) * 4;
    static DeviceOnce once;
    { int rc = cvae_grant_lds(once, reinterpret_cast<const void*>(msssim_fwd_kernel<S>), SMEM); if (rc) return rc; }
    hipLaunchKernelGGL(msssim_fwd_kernel<S>, dim3(ms_blocks<S>(a.P)), dim3(256), SMEM, st, a);
    CVAE_CHECK_LAUNCH();
    return 0;
}
template <int S>
static int ms_bwd(const MsBwdArgs& a, hipStream_t st) {
    using G = MsGeom<S>;
    constexpr int SMEM = (3 * G::PPB * G::HR * G::HC + 3 * G::PPB * G::HR * S) * 4;
    static DeviceOnce once;
    { int rc = cvae_grant_lds(once, reinterpret_cast<const void*>(msssim_bwd_kernel<S>), SMEM); if (rc) return rc; }
    hipLaunchKernelGGL(msssim_bwd_kernel<S>, dim3(ms_blocks<S>(a.P)), dim3(256), SMEM, st, a);
    CVAE_CHECK_LAUNCH();
    return 0;
}
static int ms_fwd_size(int S, const MsFwdArgs& a, hipStream_t st) {
    switch (S) {
        case 128: return ms_fwd<128>(a, st);
        case 64: return ms_fwd<64>(a, st);
        case 32: return ms_fwd<32>(a, st);
        case 16: return ms_fwd<16>(a, st);
        case 8: return ms_fwd<8>(a, st);
        case 4: return ms_fwd<4>(a, st);
    }
    cvae_set_error("msssim: level size %d unsupported", S);
    return -2;
}
static int ms_bwd_size(int S, const MsBwdArgs& a, hipStream_t st) {
    switch (S) {
        case 128: return ms_bwd<128>(a, st);
        case 64: return ms_bwd<64>(a, st);
        case 32: return ms_bwd<32>(a, st);
        case 16: return ms_bwd<16>(a, st);
        case 8: return ms_bwd<8>(a, st);
        case 4: return ms_bwd<4>(a, st);
    }
    cvae_set_error("msssim: level size %d unsupported", S);
    return -2;
}

int launch_msssim(int width, int B, const float* img1, const float* img2, const float* mu, const float* logvar,
                  float* ws, float* scalars, float* d_img1, float* d_mu, float* d_logvar, hipStream_t st) {
    if (width != 64 && width != 128) { cvae_set_error("msssim: width %d unsupported", width); return -2; }
    int rc = 0;
    const MsWin win = make_window();
    const MsWs w = ms_carve(width, B);
    const int P = B * 3;
    const bool grad = d_img1 != nullptr;
    const float* lx[5]; const float* ly[5];
    lx[0] = img1; ly[0] = img2;
    for (int l = 1; l < 5; ++l) { lx[l] = ws + w.pyrx[l]; ly[l] = ws + w.pyry[l]; }
    for (int l = 0; l < 5; ++l) {
        MsFwdArgs a{lx[l], ly[l], l < 4 ? ws + w.pyrx[l + 1] : nullptr, l < 4 ? ws + w.pyry[l + 1] : nullptr,
                    grad ? ws + w.um[l] : nullptr, grad ? ws + w.u11[l] : nullptr, grad ? ws + w.u12[l] : nullptr,
                    ws + w.part[l], P, l == 4, win};
        rc = ms_fwd_size(width >> l, a, st);
        if (rc) return rc;
    }
    MsFinArgs f{};
    f.part = ws;
    for (int l = 0; l < 5; ++l) {
        f.partOff[l] = (int)w.part[l]; f.nblk[l] = w.nblk[l];
        f.count[l] = (double)P * (width >> l) * (width >> l);
    }
    f.mu = mu; f.logvar = logvar; f.B = mu ? B : 0; f.scalars = scalars; f.coef = ws + w.coef;
    f.d_mu = d_mu; f.d_logvar = d_logvar;
    hipLaunchKernelGGL(msssim_finalize_kernel, dim3(1), dim3(1024), 0, st, f);
    CVAE_CHECK_LAUNCH();
    if (!grad) return 0;
    for (int l = 4; l >= 0; --l) {
        MsBwdArgs a{ws + w.um[l], ws + w.u11[l], ws + w.u12[l], lx[l], ly[l], l < 4 ? ws + w.gp[l + 1] : nullptr,
                    ws + w.coef + l, l == 0 ? d_img1 : ws + w.gp[l], P, win};
        rc = ms_bwd_size(width >> l, a, st);
        if (rc) return rc;
    }
    return 0;
}
