/*
 * cvae.h — C-ABI of libcvae_hip.so: the Critic-VAE training step on MI355X (gfx950).
 *
 * The reference (lcicek/Critic-VAE) has no FFI for this path: its "interface" is the set of
 * PyTorch calls made by vae.py:44-58.  This header is the boundary a maintainer would bind
 * (ctypes stub in INTEGRATION.md); each entry point names the reference call it replaces.
 *
 * Conventions
 *   - Plain C types only.  Every pointer is a DEVICE pointer unless stated; every tensor that crosses this
 *     boundary is fp32 (precision mode 1 keeps bf16 tensors only inside the workspace it is handed).
 *   - The caller owns every buffer (parameters, gradients, inputs, outputs, workspace); the
 *     library owns only the opaque handle.  No allocation, no device synchronisation inside.
 *   - Process-wide state is limited to two caches that never change results: a THREAD-LOCAL error string
 *     (cvae_last_error() reports the calling thread's last failure) and a per-device compute-unit count queried
 *     once (it sizes persistent grids).  Everything else lives in the handle; distinct handles are independent.
 *   - All work is enqueued on the caller's hipStream_t (passed as void*).
 *   - Return: 0 = OK, <0 = library error (cvae_last_error()), >0 = hipError_t passthrough.
 *   - Layouts: frames x / recon / d_recon are NCHW (B,3,W,W) exactly as the reference holds
 *     them (vae_utility.py:337-343); mu/logvar/eps (B,32), pred (B,1) row-major.  Parameters
 *     and gradients live in ONE flat fp32 buffer in the library's native layout described by
 *     cvae_param_*(): conv weights [kh*5+kw][Cin][Cout], fc_mu|fc_var fused as [k][64] with k
 *     in (h,w,c) order, decoder_input as [33][bottleneck] with columns in (h,w,c) order.
 *     critic-vae_amd/layout.py converts to/from the reference's state_dict layouts.
 */
#ifndef CVAE_H
#define CVAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cvae_handle_s* cvae_handle;

typedef struct cvae_config {
    int32_t width;        /* frame width == height: 64 (vae_parameters.py:5) or 128 (BASELINE config 5 shape) */
    int32_t max_batch;    /* largest per-call batch the workspace is sized for                 */
    int32_t overlap_wgrad;    /* != 0: run weight-gradient kernels on an internal low-priority side stream */
    int32_t precision;    /* 0 = fp32 everywhere (the 1e-4-parity path, bench default);
                           * 1 = bf16 mode (BASELINE.json configs 3-5): every contraction on the bf16 MFMA (fp32 accumulate) AND
                           *     activations / activation gradients stored as bf16 inside the workspace; x, recon, mu, logvar, the loss
                           *     gradients, parameters, gradients, BatchNorm statistics and Adam state stay fp32 at this boundary;
                           * 2 = fp32 emulation: forward/dgrad of E2..E4 and D0 on the bf16 MFMA with exact 3-way bf16 operand splits
                           *     (9 MFMAs per block), everything else as 0;  3 = as 2 with the six leading partial products only;
                           * other values are rejected */
} cvae_config;

enum { CVAE_OK = 0, CVAE_EINVAL = -1, CVAE_EUNSUPPORTED = -2, CVAE_ENOWS = -3 };

/* number of loss scalars written by cvae_loss: [0]=total [1]=recon(MS-SSIM) [2]=KLD(weighted)
 * [3..7]=ssim level means [8..12]=cs level means [13..15]=reserved                           */
#define CVAE_N_SCALARS 16

const char* cvae_version(void);
const char* cvae_last_error(void);

/* lifecycle */
int  cvae_create(const cvae_config* cfg, cvae_handle* out);
void cvae_destroy(cvae_handle h);

/* flat parameter / gradient buffer description (replaces nn.Module.parameters(), vae.py:36) */
int64_t     cvae_param_total(cvae_handle h);                 /* floats incl. alignment padding */
int32_t     cvae_param_count(cvae_handle h);                 /* number of tensors (30)          */
const char* cvae_param_name(cvae_handle h, int32_t i);       /* NATIVE tensor name: enc{0..3}.{w,b,gamma,beta}, fc.{w,b}
                                                              * (fc_mu|fc_var fused), dec{0..4}.{w,b}, decin.{w,b} — not a
                                                              * reference state_dict key; critic-vae_amd/layout.py holds the
                                                              * name + layout map to encoder./decoder. keys            */
int64_t     cvae_param_offset(cvae_handle h, int32_t i);     /* float offset in the flat buffer */
int64_t     cvae_param_numel(cvae_handle h, int32_t i);

/* workspace (saved activations, gradients of activations, split-K slabs, reduction partials) */
int64_t cvae_workspace_bytes(cvae_handle h, int32_t batch);

/* BatchNorm running statistics buffer: 2*480 floats [mean(32,64,128,256) | var(...)] */
int64_t cvae_bn_state_floats(cvae_handle h);

/*
 * Forward: VariationalAutoencoder.forward (vae_nets.py:14-19) = encoder (:101-111) ->
 * reparametrize with caller-supplied eps (:48-51) -> decoder (:139-147).
 * train != 0: BatchNorm uses batch statistics and updates bn_state (momentum 0.1, unbiased var).
 * Saves what backward needs in ws.
 */
int cvae_forward(cvae_handle h, int32_t batch, const float* x, const float* pred, const float* eps,
                 const float* params, float* bn_state, float* mu, float* logvar, float* recon,
                 void* ws, int32_t train, void* stream);

/*
 * Decoder.forward alone (vae_nets.py:139-147) from zcat = cat((z, pred), 1), shape (B,33).
 * zcat == NULL reuses the one cvae_forward left in ws.  cvae_forward with recon == NULL stops
 * after the encoder + reparametrize (VariationalEncoder.forward, vae_nets.py:101-111).
 */
int cvae_decode(cvae_handle h, int32_t batch, const float* zcat, const float* params, float* recon,
                void* ws, void* stream);

/*
 * Loss: VariationalAutoencoder.vae_loss (vae_nets.py:53-62) = MSSIM.forward (:217-247) + KLD.
 * Writes CVAE_N_SCALARS floats to `scalars` and the gradients of total_loss w.r.t. recon, mu,
 * logvar (d_* may be NULL to skip the backward half).
 */
int cvae_loss(cvae_handle h, int32_t batch, const float* x, const float* mu, const float* logvar,
              const float* recon, void* ws, float* scalars, float* d_recon, float* d_mu,
              float* d_logvar, void* stream);

/*
 * Backward: loss.backward() (vae.py:57) for everything cvae_forward computed, given the loss
 * gradients w.r.t. its outputs (and logvar/recon as returned by cvae_forward).  Overwrites the
 * flat gradient buffer `grads` (same layout as `params`).
 * `x`, `params` and `ws` MUST be the ones the matching cvae_forward ran on, bit for bit (no optimizer step, no other
 * forward on the same workspace in between): in precision mode 1 the first conv's output y0 is not stored — the E1
 * weight-gradient kernel recomputes it from `x` and the enc0.w / enc0.b of `params` and re-derives block 0's max-pool
 * decisions from those values — and every mode reads the saved activations of that forward from `ws`.  (Round 5: in
 * precision mode 1 a train-mode forward also leaves the frame in `ws` as packed bf16 pixels — 8 bytes per pixel, slot "xp" —
 * and the backward stages E1's strips from that copy when `ws` and `batch` are the ones of the handle's last train-mode
 * forward; after an eval-mode forward, or on another workspace, it converts the fp32 `x` itself, as rounds 3-4 did.)
 */
int cvae_backward(cvae_handle h, int32_t batch, const float* x, const float* pred, const float* eps,
                  const float* params, const float* logvar, const float* recon, const float* d_recon,
                  const float* d_mu, const float* d_logvar, void* ws, float* grads, void* stream);

/*
 * The same backward in three phases, in the order the gradients complete, so that a data-parallel
 * host can all-reduce one bucket of the flat gradient buffer while the next phase computes
 * (torch DDP's bucketed overlap; vae.py:57 under the north star's RCCL all-reduce):
 *   bit 0: decoder + decoder_input      bit 1: fc_mu|fc_var + encoder block 3      bit 2: encoder blocks 2..0
 * Phases must be issued in that order on one stream; phase_mask 7 == cvae_backward.  Bit 3 (value 8) additionally
 * writes 0 into the alignment padding between the tensors of `grads`, so an uninitialised buffer may be passed
 * (cvae_backward itself never touches the padding: FusedTrainer zeroes its buffer once).  cvae_grad_bucket
 * returns the contiguous [offset, offset+numel) range of `grads` that phase `phase` (0..2) completes.
 */
int cvae_backward_phases(cvae_handle h, int32_t batch, const float* x, const float* pred, const float* eps,
                         const float* params, const float* logvar, const float* recon, const float* d_recon,
                         const float* d_mu, const float* d_logvar, void* ws, float* grads, int32_t phase_mask,
                         void* stream);
int cvae_grad_bucket(cvae_handle h, int32_t phase, int64_t* offset, int64_t* numel);

/*
 * Chain-rule factor of total_loss.backward() (vae.py:57): d_*_out = d_* * gscale[0] for the three loss
 * gradients written by cvae_loss, in one launch; gscale is a DEVICE scalar (autograd's incoming gradient).
 */
int cvae_scale_loss_grads(cvae_handle h, int32_t batch, const float* gscale, const float* d_recon,
                          const float* d_mu, const float* d_logvar, float* d_recon_out, float* d_mu_out,
                          float* d_logvar_out, void* stream);

/*
 * Optional bf16 transport of the gradient all-reduce ("bf16 optional 5.17 MB"): round a range of the flat
 * fp32 gradient buffer to bf16 (RNE) into a caller-owned buffer of n 2-byte elements, and widen the reduced buffer
 * back.  n: a multiple of 4 (the ranges of cvae_grad_bucket are multiples of 64).  The all-reduce itself stays the
 * caller's (torch.distributed / RCCL).
 */
int cvae_grads_to_bf16(cvae_handle h, const float* grads, void* out_bf16, int64_t n, void* stream);
int cvae_grads_from_bf16(cvae_handle h, const void* in_bf16, float* grads, int64_t n, void* stream);

/*
 * Optimizer: torch.optim.Adam.step() with defaults (vae.py:36,58) on the flat buffers.
 * grad_scale multiplies the gradient first (1/world_size after a summing all-reduce).
 */
int cvae_adam_step(cvae_handle h, float* params, const float* grads, float* exp_avg,
                   float* exp_avg_sq, int64_t n, int32_t step, float lr, float beta1, float beta2,
                   float eps, float grad_scale, void* stream);

/*
 * Critic.evaluate (critic_net.py:66-69; eval mode) on frames x (B,3,64,64) in [0,1] -> pred (B,1),
 * the `preds` of vae.py:50.  critic_params: cvae_critic_param_count() (= 11 873) floats in the
 * reference's own state_dict order and layouts (features.{0,3,6,10,14}.{weight,bias},
 * crit.{1,4}.{weight,bias}); the critic is frozen, the library never writes it.
 */
int32_t cvae_critic_param_count(void);
int cvae_critic_forward(cvae_handle h, int32_t batch, const float* x, const float* critic_params,
                        float* pred, void* stream);

/* adjust_values + HWC->CHW of preprocess_observation (vae_utility.py:324-343): uint8 frames
 * (B,W,W,3) -> float (B,3,W,W) / 255, so that only 1 byte per value crosses PCIe. */
int cvae_preprocess_u8(cvae_handle h, int32_t batch, const uint8_t* frames_hwc, float* x, void* stream);

/* Difference mask of the inference path (get_diff_image, vae_utility.py:256-277), batched:
 * diff (B,W,W) = 0.2989|dR| + 0.5870|dG| + 0.1140|dB| of recon_zero - recon_one (both (B,3,W,W)). */
int cvae_diff_grey(cvae_handle h, int32_t batch, const float* recon_one, const float* recon_zero,
                   float* diff, void* stream);

/* float offset of a named saved tensor in the workspace ("y0".."y3", "a0".."a3", "o0".."o3",
 * "zcat", "h", "d_*" ...) for tests; -1 if unknown OR not allocated in this configuration: "d_y0" does not exist
 * (block 0's BatchNorm backward runs inside the E1 weight-gradient kernel; CVAE_FUSE_E1=0 restores it), and in
 * precision mode 1 "dout4" has no storage and the "y0" slot is STALE unless a block-0 |gamma| is < 1e-2 (the device
 * decides per step) or CVAE_FUSE_E1=0 — do not read it otherwise.  Slots hold bf16 elements in precision mode 1. */
int64_t cvae_ws_offset(cvae_handle h, int32_t batch, const char* name);

/* Which kernel family the forward (dgrad = 0) or input-gradient (dgrad = 1) pass of encoder conv layer 1..3 (E2..E4, nn.Conv2d at
 * vae_nets.py:74,79,84) takes at `batch` images: 0 = per-tile kernel (64-bit addressing, any size), 1 = two-workgroup persistent kernel,
 * 2 = big-tile persistent kernel.  The persistent kernels address their tensors with 32-bit byte offsets and REFUSE activations of 2 GiB
 * and more (E2's output: batch >= 8192 in fp32, >= 16384 in bf16 mode at 64 x 64); the launchers then take the per-tile kernel by
 * themselves.  Host logic only (the launchers' own decision path up to the launch; no device access, works without a GPU);
 * CVAE_EINVAL for arguments outside these ranges.  precision as in cvae_config. */
int32_t cvae_conv_route(int32_t precision, int32_t width, int32_t layer, int32_t dgrad, int64_t batch);

/*
 * In-step kernel probe (measurement only): bit (kind*9 + layer) of `mask` arms a HIP event pair
 * around that conv kernel (kind 0 forward, 1 dgrad, 2 wgrad; layer 1..7) inside cvae_forward /
 * cvae_backward, recorded on the stream the kernel is launched on.  HBM-side kernels: bit 0 = E1 forward
 * (in bf16 mode the BatchNorm/pool pass), bit 18 = E1 weight gradient (with block 0's fused BatchNorm backward),
 * bit 8 = D4 forward, bit 17 = the fused D4 backward, bits 27..30 = the BatchNorm+pool backward apply kernel of
 * encoder block 0..3, bit 31 = the MS-SSIM level-0 tile kernel (inside cvae_loss).  cvae_probe_read returns the
 * elapsed milliseconds of each recorded launch (host array) and clears the slot.
 */
int cvae_probe_config(cvae_handle h, uint32_t mask);
int cvae_probe_read(cvae_handle h, int32_t id, float* ms_host, int32_t cap);

/* ---------------------------------------------------------------------------------------- *
 * Per-op entry points (unit tests and the roofline probe in bench.py).  `layer`: 0..3 =
 * encoder conv blocks E1..E4 (vae_nets.py:69,74,79,84), 4..8 = decoder convs D0..D4
 * (vae_nets.py:117,121,125,129,133).  Activations are NHWC except x/recon (NCHW); decoder
 * layers 5..8 read the stored (pre-Upsample) tensor.  `scratch` needs cvae_op_scratch_floats().
 * ---------------------------------------------------------------------------------------- */
int64_t cvae_op_scratch_floats(cvae_handle h, int32_t batch);
int64_t cvae_op_bn_partial_floats(cvae_handle h, int32_t layer, int32_t batch);
int64_t cvae_op_msssim_ws_floats(cvae_handle h, int32_t batch);

/* nn.Conv2d forward (+bias; encoder: raw output + BatchNorm partials; decoder: +ReLU, D4: +Tanh) */
int cvae_op_conv_fwd(cvae_handle h, int32_t layer, int32_t batch, const float* in, const float* w,
                     const float* bias, float* out, float* bn_partials, void* scratch, void* stream);
/* input gradient, layers 1..7; decoder layers 5..7 also fold Upsample backward (2x2 sum) and the
 * ReLU mask of the producing layer's output `mask_src` */
int cvae_op_conv_dgrad(cvae_handle h, int32_t layer, int32_t batch, const float* dout,
                       const float* w, const float* mask_src, float* din, void* scratch, void* stream);
/* weight (+ optional bias) gradient, layers 0..7 */
int cvae_op_conv_wgrad(cvae_handle h, int32_t layer, int32_t batch, const float* in,
                       const float* dout, float* dw, float* dbias, void* scratch, void* stream);
/* D4 backward, fused: Tanh' -> dout (B,3,W,W), d_o3 (ReLU-masked, NHWC), dW4, db4 */
int cvae_op_d4_bwd(cvae_handle h, int32_t batch, const float* o3, const float* d_recon,
                   const float* recon, const float* w, float* dout, float* d_o3, float* dw,
                   float* db, void* scratch, void* stream);
/* BatchNorm2d(train) -> MaxPool2d(2) -> ReLU/Tanh of encoder block `layer` (0..3) */
int cvae_op_bn_pool_act_fwd(cvae_handle h, int32_t layer, int32_t batch, const float* y,
                            const float* bn_partials, const float* gamma, const float* beta,
                            float* run_mean, float* run_var, float* coef, float* a, void* scratch,
                            int32_t train, void* stream);
int cvae_op_bn_pool_act_bwd(cvae_handle h, int32_t layer, int32_t batch, const float* y,
                            const float* a, const float* da, const float* coef, const float* gamma,
                            float* dy, float* dgamma, float* dbeta, float* dbias, void* scratch,
                            void* stream);
/* MSSIM.forward (vae_nets.py:217-247) on NCHW planes, optional gradient w.r.t. img1 */
int cvae_op_msssim(cvae_handle h, int32_t batch, const float* img1, const float* img2, void* ws,
                   float* scalars, float* d_img1, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CVAE_H */
