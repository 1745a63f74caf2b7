// common.h — shared host/device declarations for libcvae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));     // one v_mfma_f32_32x32x16_bf16 operand fragment

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Activation storage type of a run: fp32 (precision modes 0, 2, 3) or bf16 (precision mode 1: y_l, a_l, h, o_i and
// their gradients live in HBM as bf16; BatchNorm statistics, loss scalars, master weights and the flat gradient
// stay fp32).  Tensor pointers cross the launchers as opaque float*; kernels index them in ELEMENTS through Act<AT>.
template <typename AT> struct Act;
template <> struct Act<float> {
    static constexpr bool BF16 = false;
    static __device__ __forceinline__ float ld(const float* p, size_t i) { return p[i]; }
    static __device__ __forceinline__ void st(float* p, size_t i, float v) { p[i] = v; }
    static __device__ __forceinline__ f32x4 ld4(const float* p, size_t i) { return *reinterpret_cast<const f32x4*>(p + i); }
    static __device__ __forceinline__ void st4(float* p, size_t i, f32x4 v) { *reinterpret_cast<f32x4*>(p + i) = v; }
};
template <> struct Act<__bf16> {
    static constexpr bool BF16 = true;
    static __device__ __forceinline__ float ld(const float* p, size_t i) { return (float)reinterpret_cast<const __bf16*>(p)[i]; }
    static __device__ __forceinline__ void st(float* p, size_t i, float v) { reinterpret_cast<__bf16*>(p)[i] = (__bf16)v; }
    static __device__ __forceinline__ f32x4 ld4(const float* p, size_t i) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p) + i);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    static __device__ __forceinline__ void st4(float* p, size_t i, f32x4 v) {
        bf16x4 o;
        o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p) + i) = o;
    }
    static __device__ __forceinline__ bf16x4 ld4raw(const float* p, size_t i) { return *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p) + i); }
    // 8 consecutive elements = one 16-byte unit
    static __device__ __forceinline__ bf16x8 ld8(const float* p, size_t i) { return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(p) + i); }
    static __device__ __forceinline__ void st8(float* p, size_t i, bf16x8 v) { *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p) + i) = v; }
};

// 8 k-values of this lane for one bf16 MFMA operand out of a row-major [k rows][channel columns] LDS image: two
// transposed 4x16 reads (ds_read_b64_tr_b16; within each group of 16 lanes, lane 4q+p supplies the address of row q,
// columns 4p..4p+3 of a 4 x 16 block and lane i receives column i of the 4 rows — profiles/experiments/tr16_probe.hip).
// EXEC must be all ones; addresses 8-byte aligned.
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* p0, const __bf16* p1) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p0);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p1);
    const short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

#define CVAE_LATENT 32
#define CVAE_ZCAT 33

// Conv layers of the path.  0..3 encoder (vae_nets.py:69,74,79,84), 4..8 decoder (:117-133).
// h = output spatial size at width 64 (scaled by width/64 at run time); up = input is the
// nearest-2x upsample of the stored tensor (vae_nets.py:119,123,127,131).
struct LayerDesc { int cin, cout, h, up; };
static constexpr LayerDesc kLayers[9] = {
    {3, 32, 64, 0}, {32, 64, 32, 0}, {64, 128, 16, 0}, {128, 256, 8, 0},
    {256, 128, 4, 0}, {128, 64, 8, 1}, {64, 32, 16, 1}, {32, 32, 32, 1}, {32, 3, 64, 1}};

// ---- output-tile geometry shared by the conv kernels: 128 output pixels per workgroup ----
template <int H>
struct Tile {
    static constexpr int TW = H < 32 ? H : 32;
    static constexpr int TH = (128 / TW) < H ? (128 / TW) : H;
    static constexpr int IMGS = 128 / (TW * TH);        // whole images per tile when H*H < 128
    static constexpr int HTW = TW + 4, HTH = TH + 4;     // halo for the 5x5 window
    static constexpr int HPI = HTW * HTH;                // halo pixels per image
    static constexpr int HP = IMGS * HPI;
    static constexpr int PS = ((HP + 5) / 8) * 8 + 2;    // LDS plane stride, == 2 (mod 8), >= HP
    static constexpr int TILES_X = H / TW, TILES_Y = H / TH;
    static constexpr int TILES_PER_IMG = TILES_X * TILES_Y;
    static_assert(TW * TH * IMGS == 128, "tile must hold 128 pixels");
};

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// `s_waitcnt vmcnt(0)` the compiler can SEE (the builtin, not inline asm), placed where a register-prefetch loop ends and nothing
// is in flight any more.  The loop's staging registers are load destinations; without this the compiler protects their reuse in the
// epilogue with counted waits (vmcnt(3), (2), (1) ...) that it derives from the loop body — and since vmcnt counts stores too and
// retires in order, such a wait placed after the epilogue's first global stores waits for THOSE (profiles/experiments/vm_after_store.py).
__device__ __forceinline__ void vm_drained() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // gfx9 encoding: vmcnt = 0, expcnt / lgkmcnt untouched

// Workgroups are dealt to the 8 XCDs round-robin by linear id, and every XCD has its own L2.  Map
// workgroup x of a grid of G (G % 8 == 0) to tile (x % 8) * (G / 8) + x / 8: each XCD then owns a
// contiguous range of tiles — neighbouring tiles share their halo rows in ONE L2, and the workgroups of
// the other output-channel tiles (blockIdx.y) that re-read the same input tile sit on the same XCD.
__device__ __forceinline__ int xcd_tile(int x, int G) { return (G & 7) ? x : (x & 7) * (G >> 3) + (x >> 3); }
inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// In-launch "last workgroup to arrive" (cdna guide §6 Guideline 16, counter form; placement-independent): every wave
// drains its stores, the workgroup meets, ONE lane releases at agent scope, drains again and draws a ticket; the
// workgroup that draws total-1 acquires at agent scope and may then read, with plain loads, everything the other
// workgroups stored before they arrived.  Returns the same answer in every thread.  *counter must be 0 when the
// launch starts (zeroed by an EARLIER launch on the stream); the last arriver puts it back to 0.
__device__ __forceinline__ bool wg_arrive_last(unsigned* counter, unsigned total, unsigned* lds_word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t == total - 1;
        if (last) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *lds_word = last ? 1u : 0u;
    }
    __syncthreads();
    return *lds_word != 0u;
}

// Optional in-step kernel probe (api.hip): the launchers bracket their MAIN kernel with a HIP event
// pair on the stream it is launched on when the orchestration armed a slot (bench.py roofline).
void cvae_probe_begin(hipStream_t st);
void cvae_probe_end(hipStream_t st);

// Split-K slab reductions of the weight gradients are off the critical path (nothing reads dW before the phase
// ends): when the orchestration arms a side stream (api.hip), a wgrad launcher calls cvae_reduce_stream(st) right
// after its main kernel and enqueues its small reduce / expand kernels on the stream it returns (ordered behind
// the main kernel by an event), so their ~5 us launch floors overlap the next input-gradient kernel.
hipStream_t cvae_reduce_stream(hipStream_t st);

// error plumbing (api.hip)
void cvae_set_error(const char* fmt, ...);
#define CVAE_CHECK_LAUNCH()                                                        \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            cvae_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,          \
                           hipGetErrorString(e_));                                 \
            return (int)e_;                                                        \
        }                                                                          \
    } while (0)

// The opt-in to more than 64 KB of dynamic LDS is a PER-DEVICE function attribute: remember, per kernel,
// the set of devices it has been granted on (idempotent, thread-safe; one handle per device may live in
// the same process).  `once` is a function-local static of the launcher that owns the kernel.
struct DeviceOnce { std::atomic<uint64_t> mask{0}; };
inline int cvae_grant_lds(DeviceOnce& once, const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (e == hipSuccess && (once.mask.load(std::memory_order_acquire) & bit)) return 0;
    if (e == hipSuccess) e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { cvae_set_error("dynamic LDS opt-in (%d bytes) failed: %s", bytes, hipGetErrorString(e)); return (int)e; }
    once.mask.fetch_or(bit, std::memory_order_release);
    return 0;
}

// Compute units of the current device (256 on the MI355X), queried once per device: sizes the persistent grids.
int cvae_num_cus();

// ---- launchers implemented across the .hip files (all asynchronous on `st`) ----
// conv_mfma.hip
int launch_conv_fwd(int layer, int width, int B, const float* in, const float* w, const float* bias,
                    float* out, float* bnpart, float* ws, hipStream_t st);
int64_t conv_fwd_ws_floats(int layer, int width, int B);
int launch_conv_dgrad(int layer, int width, int B, const float* dout, const float* w,
                      const float* mask_src, float* din, float* ws, hipStream_t st);   // ws may be null (no split-K)
int64_t conv_dgrad_ws_floats(int layer, int width, int B);
// conv_bf16.hip — precision mode 1: bf16-MFMA forward / dgrad / wgrad of layers 1..7
bool conv_bf16_supported(int layer, int width);
int64_t conv_bf16_pack_floats(int ns);
int launch_pack_w_bf16(const float* const w[4], float* packed, int ns, hipStream_t st);     // ns = 1 (bf16) or 3 (fp32 emulation)
// cvae_conv_route (host logic only): while set, the persistent launchers return 0 behind their size guards WITHOUT touching the device
extern thread_local bool g_conv_dry;
// kernel family the conv launchers pick for E2..E4 (layer 1..3; 4 = D0 at 128 x 128) at batch B: 0 per-tile, 1 two-workgroup persistent, 2 big-tile persistent
int conv_bf16_route(int layer, int width, bool dgrad, int B);
int conv_f32_route(int layer, int width, bool dgrad, int B);
int launch_conv_fwd_bf16(int layer, int width, int ns, int B, const float* in, const float* packed, const float* bias, float* out,
                         float* bnpart, float* ws, hipStream_t st, int* tilesPerPartial = nullptr);    // out: 128-pixel tiles per BatchNorm partial row
int launch_conv_dgrad_bf16(int layer, int width, int ns, int B, const float* dout, const float* packed, float* din, float* ws, hipStream_t st);
int launch_pack_up_bf16(const float* const wc[3], float* packed, int ns, hipStream_t st);
int launch_conv_up_fwd_bf16(int layer, int width, int ns, int B, const float* in, const float* packed, const float* bias, float* out, hipStream_t st);
int launch_conv_up_dgrad_bf16(int layer, int width, int ns, int B, const float* dout, const float* packed, const float* aux, float* din, hipStream_t st);
int64_t wgrad_bf16_ws_floats(int layer, int width, int B);
int launch_conv_wgrad_bf16(int layer, int width, int B, const float* in, const float* dout, float* dw, float* dbias, float* ws, hipStream_t st);
bool conv_wgrad_split_supported(int products);
int64_t wgrad_split_ws_floats(int layer, int width, int B);        // fp32-emulation modes: exact 3-way operand splits on the bf16 MFMA
int launch_conv_wgrad_split(int layer, int width, int products, int B, const float* in, const float* dout, float* dw, float* dbias,
                            float* ws, hipStream_t st);
int launch_splitk_bias_relu(const float* slab, const float* bias, float* out, int64_t slice, int KS, int C, hipStream_t st, bool out_bf16 = false);
// conv_wgrad.hip
int64_t wgrad_ws_floats(int layer, int width, int B);
int launch_conv_wgrad(int layer, int width, int B, const float* in, const float* dout, float* dw,
                      float* dbias, float* ws, hipStream_t st);
int launch_reduce_slabs(const float* slab, float* dst, int64_t n, int S, int64_t stride, hipStream_t st,
                        float* mid = nullptr, bool out_bf16 = false);
int launch_colsum(const float* src, int64_t rows, int C, float* dst, float* ws, hipStream_t st);
int64_t colsum_ws_floats(int64_t rows, int C);
// reduce.hip
int64_t col_reduce_ws_floats(int W);
int launch_col_reduce_partial(const float* in, int R, int W, int64_t stride, float* ws, hipStream_t st,
                              const float** rows_out, int* R_out, int64_t* stride_out);
int launch_col_reduce(const float* in, int R, int W, int64_t stride, float* out, float* ws, hipStream_t st);
// conv_thin.hip (E1 / D4)
int launch_e1_fwd(int width, int B, const float* x, const float* w, const float* bias, float* y,
                  float* bnpart, hipStream_t st, bool bf16 = false, int pass = 0, const float* coef = nullptr, float* a1 = nullptr,
                  bool keep_y = true, float* xp = nullptr);
                  // bf16 mode: pass 1 = BatchNorm partials only (+ writes the packed bf16 frame xp, B*width*width*8 bytes of workspace),
                  // pass 2 = fused BatchNorm/pool/ReLU -> a1 (needs coef; stages its strips from xp); y is written
                  // by pass 2 only if keep_y or a channel has |gamma| < 1e-2 (the backward's statistics then read it)
int launch_e1_wgrad(int width, int B, const float* x, const float* dy, float* dw, float* dbias, float* ws,
                    hipStream_t st, bool bf16 = false, const float* const* fuse = nullptr,    // fuse: {y0, a0, d_a0, coef0, bcoef0, w1, b1}, dy unused
                    const float* xp = nullptr);                                               // bf16 mode: the packed frame of launch_e1_fwd's pass 1, or null
                    // (bf16 mode recomputes y0 from x, w1, b1 and never reads fuse[0])
const float* bn_bwd_bcoef(int layer, int width, int B, const float* ws);                       // where launch_bn_pool_act_bwd left (k1, k2)
int64_t e1_wgrad_ws_floats(int width, int B);
int launch_d4_fwd(int width, int B, const float* in, const float* w, const float* bias, float* recon,
                  hipStream_t st, bool bf16io = false);
int64_t d4_bwd_ws_floats(int width, int B);
int launch_d4_bwd(int width, int B, const float* o3, const float* d_recon, const float* recon,
                  const float* w, float* dout, float* d_o3, float* dw, float* db, float* ws,
                  hipStream_t st, bool bf16io = false);
// conv_up.hip (phase-collapsed Upsample->Conv blocks D1..D3, layers 5..7)
int64_t conv_up_wc_floats(int layer);
int launch_collapse_w(int layer, const float* w, float* wc, hipStream_t st);
int launch_collapse_w3(const float* const w[3], float* const wc[3], hipStream_t st);     // D1..D3 in one launch
int64_t conv_up_ws_floats(int layer, int width, int B);
int launch_conv_up_fwd(int layer, int width, int B, const float* in, const float* wc, const float* bias,
                       float* out, float* ws, hipStream_t st);
int launch_conv_up_dgrad(int layer, int width, int B, const float* dout, const float* wc, const float* aux,
                         float* din, float* ws, hipStream_t st);
int64_t conv_up_wgrad_ws_floats(int layer, int width, int B);
int launch_conv_up_wgrad(int layer, int width, int B, const float* in, const float* dout, float* dw,
                         float* dbias, float* ws, hipStream_t st, bool bf16 = false);
int launch_up_wgrad_bf16_main(int layer, int width, int B, const float* in, const float* dout, float* slab, int Smax, int* S_out, hipStream_t st);
// bn.hip
int bn_num_tiles(int layer, int width, int B);
int launch_bn_fwd_finalize(int layer, int width, int B, const float* bnpart, const float* gamma,
                           const float* beta, float* run_mean, float* run_var, float* coef,
                           float* ws, int train, hipStream_t st, int tilesPerPartial = 1);
int64_t bn_fwd_ws_floats(int layer, int width);
int launch_bn_pool_act_fwd(int layer, int width, int B, const float* y, const float* coef, float* a,
                           hipStream_t st, bool bf16io = false);
int64_t bn_bwd_ws_floats(int layer, int width, int B);
int launch_bn_pool_act_bwd(int layer, int width, int B, const float* y, const float* a,
                           const float* da, const float* coef, const float* gamma, float* dy,
                           float* dgamma, float* dbeta, float* dbias, float* ws, hipStream_t st, bool bf16io = false);
// fc.hip
int64_t fc_ws_floats(int width, int B);
int launch_fc_fwd(int width, int B, const float* flat, const float* wfc, const float* bfc,
                  const float* eps, const float* pred, float* mu, float* logvar, float* zcat,
                  float* ws, hipStream_t st, bool bf16io = false);
int launch_decin_fwd(int width, int B, const float* zcat, const float* wd, const float* bd, float* h,
                     hipStream_t st, bool bf16io = false);
int launch_decin_bwd(int width, int B, const float* zcat, const float* dh, const float* wd, float* dwd,
                     float* dbd, float* dzcat, float* ws, hipStream_t st, bool bf16io = false);
int launch_fc_bwd(int width, int B, const float* flat, const float* wfc, const float* dzcat,
                  const float* eps, const float* logvar, const float* dmu_loss,
                  const float* dlv_loss, float* dwfc, float* dbfc, float* dflat, float* ws,
                  hipStream_t st, bool bf16io = false);
// msssim.hip
int64_t msssim_ws_floats(int width, int B);
int launch_msssim(int width, int B, const float* img1, const float* img2, const float* mu,
                  const float* logvar, float* ws, float* scalars, float* d_img1, float* d_mu,
                  float* d_logvar, hipStream_t st);
// critic.hip
int critic_param_count();
int launch_critic_fwd(int width, int B, const float* x, const float* critic_params, float* pred, hipStream_t st);
int launch_preprocess_u8(int width, int B, const uint8_t* u8, float* x, hipStream_t st);
int launch_diff_grey(int width, int B, const float* a, const float* b, float* diff, hipStream_t st);
// adam.hip
int launch_grads_bf16(const float* src_f32, void* bf16_buf, float* dst_f32, int64_t n, hipStream_t st);   // src set: pack; else unpack
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, int step, float lr, float b1,
                float b2, float eps, float gscale, hipStream_t st);
int launch_zero(float* p, int64_t n, hipStream_t st);
struct PadGaps { int64_t off[32]; int len[32]; int n; };
int launch_zero_gaps(float* grads, const PadGaps& gaps, hipStream_t st);
struct Scale3 { const float* g; const float* src[3]; float* dst[3]; int64_t n[3]; };
int launch_scale3(const Scale3& a, hipStream_t st);
