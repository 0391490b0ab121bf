/*
 * mcvd_hip.h -- C ABI of libmcvd_hip.so: MI355X (gfx950) native implementation of the
 * MCVD DDPM/DDIM sampling hot path (conditional `unetmore` UNet + sampler loop).
 *
 * Plain pointers and sizes only; no torch types.  All tensor pointers are fp32,
 * contiguous NCHW, DEVICE memory unless a parameter says "host".  Every function
 * returns 0 on success or a negative MCVD_E* code and never throws across the ABI;
 * mcvd_last_error() gives the message (reference convention: Python exceptions /
 * assert, e.g. models/better/ncsnpp_more.py:382, models/better/layerspp.py:227).
 *
 * Ownership: the caller owns x / cond / eps / noise buffers; the library owns its
 * parameter blob, packed weights, caches and workspace.  Nothing is allocated inside
 * the step loop once a batch size has been seen.  One mcvd_ctx per device; calls on
 * one ctx are externally serialised; different ctxs may be used concurrently.  All
 * work is enqueued on the HIP stream given to the ctx (reference: the native op runs
 * on at::cuda::getCurrentCUDAStream, models/better/op/upfirdn2d_kernel.cu:213-215).
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   mcvd_model_create/set_param/finalize  <- get_model + load_state_dict,
 *                                            runners/ncsn_runner.py:180-195, :923-932
 *   mcvd_unet_forward                     <- UNetMore_DDPM.forward, models/better/ncsnpp_more.py:753-770
 *   mcvd_unet_forward_ft                  <- the same forward called with float timesteps (F-PNDM: t_list[1] = (t + t_next) / 2,
 *                                            models/pndm.py:42; timesteps.float() in layers.py:504-518)
 *   mcvd_pndm_transfer / mcvd_lincomb     <- transfer / the Runge-Kutta and Adams-Bashforth combinations, models/pndm.py:19-33, :15, :47
 *   mcvd_sampler_run                      <- ddpm_sampler / ddim_sampler, models/__init__.py:206-340 / :102-203
 *   mcvd_sampler_update                   <- the per-step update algebra, models/__init__.py:287-290, :165-168, :324-328
 *   mcvd_upfirdn2d                        <- pybind upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0..pad_y1),
 *                                            models/better/op/upfirdn2d.cpp:12-23
 *   mcvd_model_export_blob/import_blob    <- nn.DataParallel's per-forward replicate (runners/ncsn_runner.py:924);
 *                                            here ONE broadcast of the packed blob at load time
 */
#ifndef MCVD_HIP_H
#define MCVD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCVD_OK 0
#define MCVD_EINVAL (-1)   /* bad argument / unsupported configuration */
#define MCVD_EHIP (-2)     /* HIP runtime error */
#define MCVD_ESTATE (-3)   /* call order violated (e.g. forward before finalize, missing parameter) */
#define MCVD_ENOMEM (-4)
#define MCVD_ERANGE (-5)   /* option "f16x2" only: a network output was not finite -- an activation left the fp16 range of the two-piece
                              kernels (|x| beyond ~1e3 behind a GroupNorm), or the model diverged.  Rerun with f16x2 = 0 (the default) */

#define MCVD_MAX_LEVELS 8

typedef struct mcvd_ctx mcvd_ctx;
typedef struct mcvd_model mcvd_model;

/* The hot-path keys of the reference YAML schema (configs/[name].yml, data.* and model.*). */
typedef struct mcvd_unet_desc {
    int32_t image_size;        /* data.image_size */
    int32_t channels;          /* data.channels */
    int32_t num_frames;        /* data.num_frames (frames predicted per block) */
    int32_t num_frames_cond;   /* data.num_frames_cond + data.num_frames_future (ncsnpp_more.py:47) */
    int32_t ngf;               /* model.ngf */
    int32_t n_levels;          /* len(model.ch_mult) */
    int32_t ch_mult[MCVD_MAX_LEVELS];
    int32_t num_res_blocks;    /* model.num_res_blocks */
    int32_t n_attn;            /* len(model.attn_resolutions) */
    int32_t attn_resolutions[MCVD_MAX_LEVELS];
    int32_t n_head_channels;   /* model.n_head_channels (-1: single head) */
    int32_t spade;             /* model.spade */
    int32_t spade_dim;         /* model.spade_dim */
    int32_t num_classes;       /* model.num_classes (T) */
    int32_t sigma_dist;        /* 0 = linear, 1 = cosine (models/__init__.py:16-35) */
    float sigma_begin;         /* model.sigma_begin */
    float sigma_end;           /* model.sigma_end */
    int32_t cond_emb;          /* model.cond_emb: Embedding(2, ngf/2) of the per-sample cond_mask concatenated to temb (ncsnpp_more.py:97-99, :282-286) */
    int32_t noise_in_cond;     /* model.noise_in_cond: every forward diffuses the conditioning frames to the level of its labels (:755-768) */
    int32_t gamma;             /* model.gamma: gamma-distributed noise (ncsnpp_more.py:744-749, :761-765; models/__init__.py:273-276, :319-322) */
} mcvd_unet_desc;

/* sampler kinds / flags (models/__init__.py) */
#define MCVD_SAMPLER_DDPM 0
#define MCVD_SAMPLER_DDIM 1
#define MCVD_FLAG_DENOISE 1      /* denoise=True  (:331-333) */
#define MCVD_FLAG_CLIP_BEFORE 2  /* clip_before=True (:288-289) */
#define MCVD_FLAG_JUST_BETA 4    /* just_beta=True (:325-326) */
#define MCVD_FLAG_GAMMA 8        /* gamma=True: step / re-noise draws are standardised gamma variates (:273-276, :319-322) */

/* ---- context ---------------------------------------------------------------------- */
/* One context per (device, stream); replaces the reference's implicit "current CUDA device + current stream" (op/upfirdn2d_kernel.cu:213-215).
 * ONE PROCESS PER GPU is enforced: the first process to create a context on a device takes an advisory lock on it (a lock file keyed by
 * the PCI bus id, released when the process exits); a second process gets MCVD_EBUSY -- kernels of two processes sharing the CUs of one
 * MI355X corrupted each other's results (profiles/r04_two_process_corruption.txt; cause, round 6: a v_pk_fma_f32 that reads one VGPR pair as
 * src1 and src2 loses its low addend beside another wave's 128-bit-operand MFMA, profiles/r06_coresident_cause.txt -- this library's kernels
 * no longer hold that form, a co-tenant's may).  MCVD_ALLOW_SHARED_DEVICE=1 in the second process's environment lets it in for callers
 * that take turns on the device; such a context reports 1 from mcvd_ctx_device_shared (and, with the option "share_fence", runs attention
 * on the fp32 MFMA as rounds 5-6 did).  Several contexts of ONE process (one per stream) are allowed: measured clean, INTEGRATION.md section 4. */
#define MCVD_EBUSY (-7)    /* mcvd_ctx_create: the device is held by another process of this library */
int mcvd_ctx_create(int device, void* hip_stream, mcvd_ctx** out);
int mcvd_ctx_device_shared(mcvd_ctx* ctx);
void mcvd_ctx_destroy(mcvd_ctx* ctx);
int mcvd_ctx_set_stream(mcvd_ctx* ctx, void* hip_stream);
/* options (int values; an unknown key is MCVD_EINVAL):
 *   ARITHMETIC of the GEMM-shaped kernels (3x3 / 1x1 convs, attention).  Storage, accumulation and every elementwise op are fp32.
 *     "bf16x3" (default 1): offer the three-piece bf16 kernels -- both MFMA operands split EXACTLY into three bf16 pieces (all 24 bits,
 *         the fp32 exponent range; nothing scaled or clamped, Inf / NaN propagate), six piece products accumulated in fp32: less than one
 *         fp32 rounding per product, i.e. fp32-equivalent.  With 0 every product runs on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *     "f16x2" (default 0): ALSO offer the two-piece fp16 kernels -- operands rounded to two fp16 pieces (22-23 significant bits where fp32
 *         has 24; fp16 exponent range), three piece products: ~15 % faster end to end, NARROWER than the reference's fp32.  Range: the
 *         pieces are finite for |activation| < ~1e3 (a Winograd transform value is a sum of four, times 2^4, against 65504); values below
 *         ~1e-6 lose relative precision (fp16 denormals).  The library therefore (1) gives these kernels GroupNorm-ed inputs only -- a conv
 *         over a raw tensor (stem, shortcuts, NIN_3) always runs the fp32-range kernels -- and (2) clamps nothing: an overflow becomes Inf
 *         / NaN and is REPORTED (MCVD_ERANGE, mcvd_ctx_check_range) instead of saturating silently.
 *   KERNEL SELECTION
 *     "autotune" (1): time the candidate kernels per distinct layer shape on first use of a batch size and keep the fastest (a table
 *         imported through mcvd_model_set_tuning is used as is, with or without autotune; the arithmetic options above always win over it).
 *     "conv_shape" (-1 auto): force a kernel family for every conv (tests): 0/1/2/3 the 256/128/64-pixel / split-K direct tiles, 4 fp32
 *         Winograd F(2x2,3x3) (8: with its 2-way split of the input channels), 5 / 6 / 9 the fp32 all-DMA 1x1 GEMM (16 / 32 channels per
 *         chunk / 64 pixels per wave), 10 / 11 three-piece bf16 Winograd (11: K split), 12 / 13 two-piece fp16 Winograd (13: K split),
 *         14 / 15 the split-operand 1x1 GEMM with two fp16 / three bf16 pieces, 16 / 17 the three-piece bf16 Winograd kernel as PERSISTENT
 *         workgroups (one per CU walks a range of (region, cout tile[, K half]) items, the staging pipeline runs on across items; results
 *         bit-identical to 10 / 11; 17: K split), 18 / 19 the three-piece bf16 Winograd kernel with the input channels in 4 / 8 parts and
 *         20 the persistent kernel with 4 parts (small batches: 8x8 / 16x16 layers with fewer (region, cout tile) pairs than CUs; the
 *         parts are summed in index order by the reduce pass -- deterministic; a layer without the chunks for the depth takes the next
 *         shallower split); 22 / 23 (models only, not mcvd_op_conv2d) a 3x3
 *         conv with a handful of channels on one side as a 1x1 GEMM on the three-piece bf16 kernel: 23 = im2col of at most 10 input
 *         channels + GEMM (the stem), 22 = GEMM to 9 * Cout planes (Cout <= 7) + shift-and-add (the last conv); kernels/conv_gemm_forms.cpp.
 *         A family that does not serve a launch falls back.  "temb_table" (1): the device-loop samplers (mcvd_sampler_run) run the time MLP and all
 *         Dense_0 projections ONCE per call for the labels of all its forwards and every forward starts with a device copy of its row instead of
 *         those two launches (bit-identical; 0 = per forward, as mcvd_unet_forward always does).  "im2col_lds" (1): shape id 23 stages its im2col in LDS inside the GEMM kernel
 *         (conv1x1_h2.cpp IM: raw patch of the pixel tile + offset table; no HBM `col` tensor) where its geometry applies (W <= 128, whole
 *         image rows per 128-pixel tile); 0 = im2col3x3_kernel materialises it first (bit-identical).  "persist_grid" (0 = one workgroup
 *         per CU): number of workgroups of the persistent kernel (tests: long item ranges on small tensors).
 *     "conv_shape1" (-1): the same for the 1x1 convs only (they follow "conv_shape" otherwise), so that a test can put every 3x3 AND
 *         every 1x1 conv of a model on chosen kernels at once.  "conv_cot": cout tile (32-channel units) mcvd_op_conv2d requests with
 *         conv_shape 5 / 6 / 9 / 14 / 15.
 *     "naive_attn": 0 auto (three-piece bf16 flash kernel for head dims 32..128, two-piece fp16 when "f16x2" is on, else the fp32 flash
 *         kernel), 1 one-thread-per-query test kernel, 2 fp32 flash kernel, 3 / 4 force the two-piece fp16 / three-piece bf16 kernel.
 *     "naive_conv" (0/1): one-thread-per-output conv kernel (tests triangulate with it).
 *     "winograd" / "conv_dma1" (1): offer the Winograd / all-DMA 1x1 kernels to the autotuner.  "conv_wdma" (1: direct-conv weight chunks
 *         by LDS-DMA, 0: register staging).
 *   EXECUTION
 *     "graph" (0/1): replay each UNet forward as ONE hipGraph launch instead of ~190 kernel launches -- a forward is run eagerly the first
 *         time a (x, labels, cond, eps, B) pointer set is seen, captured on a private stream the second time and replayed afterwards;
 *         mcvd_sampler_run presents the same set on every step.  Any option change, re-tune, workspace growth or mcvd_model_finalize
 *         drops the captured graph.  Also MCVD_GRAPH=1 in the environment at mcvd_ctx_create.
 *     "gn_stats" (1): GroupNorm statistics come out of the producing conv's epilogue where the kernel supports it (0: always one pass
 *         over the normalised tensor).
 *         (The consumer-side reduction of those statistics, option "gn_inline" of rounds 3-5, measured 0.3-1 % slower on every config and was
 *         removed in round 6: profiles/r03_gn_inline_ab.txt.)  "spade_fuse" (0): 1 = the SPADE modulation inside the fp32 Winograd conv loader (gamma | beta by
 *         LDS-DMA; measured 3.5 % slower end to end than the materialising spade_apply kernel).  "side_stream" (0): ResBlock shortcut
 *         convs on a second HIP stream (measured +- 0.1 % on configs 2 / 4 / 5, profiles/r06_side_stream_ab.txt; safe since round 6 removed the
 *         instruction form that broke beside the bf16 matrix kernels from every kernel of the library: INTEGRATION.md section 4).
 *         "share_fence" (0): 1 = while the context shares its device (mcvd_ctx_device_shared) attention runs on the fp32 MFMA kernel -- the
 *         round-5 workaround for the co-residency corruption, whose cause round 6 removed from the library's kernels.
 *         "gn_producer" (1): the second pass of a K-split Winograd layer over 8 x 8 / 16 x 16 planes also writes the (A, B) table of the
 *         single-source norm over its output (one workgroup per (sample, group); bit-identical to gn_finalize), that norm's launch is
 *         skipped; 0 = two launches.
 *         "fir_form" (0): 0 = the x2 FIR resamplers stage a strip of 1024 input elements through the LDS (prologue applied once per
 *         element; power-of-two widths 8..256), 1 = the register-window forms only; bit-identical (up_or_down_sampling.py:196-258).
 *         (Rounds 5's "spade_norm_fuse" -- finalize + modulation in one launch, -1 ... +0.7 % -- and "spade_fuse_auto" -- the fused loader
 *         offered per layer as shape ids 36 / 40, chosen for 0 of 57 layers -- were removed in round 6: profiles/r05_spade_fusion_ab.txt.)
 *         "attn_presplit" (1): the fused q|k|v projection
 *         writes K and V already split into the three bf16 pieces, in the LDS-image order of the attention kernel, which then stages its
 *         tiles by LDS-DMA (head dims 32 / 64 / 96, default arithmetic, a device of its own); bit-identical to 0 (the attention kernel splits
 *         K / V itself, once per query tile).  "profile" (0/1): see mcvd_model_profile_read.
 * Environment variables read ONCE at mcvd_ctx_create set the same options: MCVD_AUTOTUNE, MCVD_SIDE_STREAM, MCVD_WINOGRAD, MCVD_CONV_DMA1,
 * MCVD_BF16X3, MCVD_F16X2, MCVD_GRAPH, MCVD_GN_STATS, MCVD_SPADE_FUSE, MCVD_NAIVE; MCVD_ALLOW_SHARED_DEVICE (see mcvd_ctx_create) is read there
 * too.  Nothing else in the production library reads the
 * environment (the diagnostics build, csrc/build.py --diag, adds timing-only ablation hooks). */
int mcvd_ctx_set_option(mcvd_ctx* ctx, const char* key, int value);
/* Option "f16x2" only (the default three-piece bf16 arithmetic has the fp32 range and needs no guard): every UNet forward run while the
 * option is on is followed by a scan of its epsilon for non-finite values (nothing is clamped in the two-piece fp16 kernels: an
 * out-of-range activation becomes Inf, then NaN).  This call synchronises the context's stream, returns MCVD_ERANGE (and a message in
 * mcvd_last_error) if any forward since the last call produced one, else 0, and clears the record.  mcvd_sampler_run / mcvd_fpndm_run
 * call it themselves before they return; a caller of mcvd_unet_forward* calls it when it wants the verdict. */
int mcvd_ctx_check_range(mcvd_ctx* ctx);
/* Runtime self-test of the hand-scheduled Winograd kernels: conv_wino3_kernel and conv_wino3p_kernel read and write
 * registers the compiler is only kept away from by a function attribute and wait for their loads with hand-counted `s_waitcnt`; the
 * build checks the generated code (tools/check_wino_isa.py), this call checks the running binary on the running device: one small
 * 3x3 conv (64 -> 96 channels, 16 x 16, GroupNorm + SiLU prologue, residual) through shape ids 10 and 16 against the fp32-MFMA
 * Winograd kernel (shape id 4: compiler-scheduled MFMAs).  10 vs 4 within 1e-4 of the output scale and 16 bit-equal to 10 -> returns 0
 * (and 1 on later calls without re-running).  On a mismatch the context option "bf16x3" is switched OFF (every product then runs on the
 * fp32 MFMA), the reason is left in mcvd_last_error and MCVD_ESELFTEST is returned.  mcvd_model_finalize runs it once per context;
 * a result is bit-reproducible either way. */
#define MCVD_ESELFTEST (-6)
int mcvd_ctx_selftest(mcvd_ctx* ctx);
/* Forgets a pending range verdict without reporting it (no synchronisation).  mcvd_sampler_run / mcvd_fpndm_run call it on entry, the
 * Python host loops at the start of every sampler call, and a change of the option "f16x2" implies it: a flag left behind by an earlier,
 * unrelated forward (or by a call that returned an error before its own check) must not fail the next run. */
int mcvd_ctx_clear_range(mcvd_ctx* ctx);
/* Diagnostics: when set (device pointer to [n_blocks][8] uint64, or NULL to disable), mcvd_op_conv2d's MFMA kernel records per
 * block the shader cycles wave 0 spent in {prologue, MFMA phases, barrier after MFMA, staging writes, second barrier, split-K
 * reduction, epilogue, total}. */
int mcvd_ctx_set_debug_buffer(mcvd_ctx* ctx, void* device_u64);
const char* mcvd_last_error(mcvd_ctx* ctx); /* ctx may be NULL: last error of this thread */
const char* mcvd_version(void);

/* ---- model ------------------------------------------------------------------------ */
/* ctx may be NULL: a plan-only model (parameter table, schedule, launch count; no device memory) for GPU-less host tests. */
int mcvd_model_create(mcvd_ctx* ctx, const mcvd_unet_desc* desc, mcvd_model** out);
void mcvd_model_destroy(mcvd_model* m);
/* Parameters are addressed by their reference state_dict names ("unet.all_modules.3.Conv_0.weight");
 * a leading "module." (DataParallel prefix, runners/ncsn_runner.py:426-433) is accepted and ignored. */
int mcvd_model_num_params(mcvd_model* m);
int mcvd_model_param_info(mcvd_model* m, int index, const char** name, int64_t shape[4], int* ndim,
                          int64_t* blob_offset_floats);
int mcvd_model_set_param(mcvd_model* m, const char* name, const float* data, const int64_t* shape, int ndim,
                         int data_on_device);
/* The raw parameter blob (all parameters, state_dict order, fp32): for the one-shot RCCL broadcast. */
int mcvd_model_blob_floats(mcvd_model* m, int64_t* n_floats);
int mcvd_model_export_blob(mcvd_model* m, float* dst_device);
int mcvd_model_import_blob(mcvd_model* m, const float* src_device); /* marks every parameter as set */
/* The one-shot weight broadcast of the data-parallel launch, directly on an RCCL communicator (replaces the per-forward
 * replicate of nn.DataParallel, runners/ncsn_runner.py:924): ncclBroadcast of the raw blob, in place, from rank `root` on the context's
 * stream.  `rccl_comm` is the caller's ncclComm_t (one process per GPU); librccl is resolved at the first call (dlopen), the library
 * has no link-time dependency on it.  Every rank must call it; afterwards every parameter counts as set and mcvd_model_finalize has
 * to run (non-root ranks need no mcvd_model_set_param at all).  Hosts that already have torch.distributed use
 * mcvd_model_export_blob / import_blob around dist.broadcast instead (mcvd_pytorch_amd/dist.py): same bytes, same single collective. */
int mcvd_model_broadcast_params(mcvd_model* m, void* rccl_comm, int root);
/* Pack/transposes weights into kernel layouts, builds the static op plan. */
int mcvd_model_finalize(mcvd_model* m);
/* Schedule buffers exactly as UNetMore_DDPM registers them (ncsnpp_more.py:735-743); host pointers, n = num_classes. */
int mcvd_model_get_schedule(mcvd_model* m, float* betas_host, float* alphas_host, float* alphas_prev_host, int n);
/* Overwrite the schedule buffers (host pointers, n = num_classes).  The library's own default restates the reference in C;
 * hosts that already hold the reference's buffers (a loaded checkpoint: state_dict keys betas/alphas/alphas_prev) pass them. */
int mcvd_model_set_schedule(mcvd_model* m, const float* betas_host, const float* alphas_host, const float* alphas_prev_host,
                            int n);
/* Sinusoid frequency table of get_timestep_embedding (layers.py:504-518); host pointer, n = ngf/2.
 * Optional: the library computes exp(-k ln(1e4)/(half-1)) itself; callers that need bit-identity with another
 * libm (torch's vectorised expf) may overwrite it. */
int mcvd_model_set_temb_freqs(mcvd_model* m, const float* freqs_host, int n);

/* eps = UNet(x, labels, cond).  x:[B, C*nf, S, S]  labels:[B] int64  cond:[B, C*nc, S, S] (NULL iff nc==0)  eps like x. */
int mcvd_unet_forward(mcvd_model* m, const float* x, const int64_t* labels, const float* cond, float* eps_out, int B);
/* Same forward with float timesteps t:[B] (device pointer), which may be fractional or negative: the F-PNDM sampler evaluates the
 * network at (t + t_next) / 2 (models/pndm.py:42) and the reference's embedding takes timesteps.float() (layers.py:504-518). */
int mcvd_unet_forward_ft(mcvd_model* m, const float* x, const float* t, const float* cond, float* eps_out, int B);
/* The same forward with the per-sample conditioning mask of `model.cond_emb` nets: cond_mask:[B] int32 (0 / 1, device) or NULL
 * (= all ones, what every sampler passes: models/__init__.py:263 binds only cond).  NULL-equivalent for nets without cond_emb. */
int mcvd_unet_forward_masked(mcvd_model* m, const float* x, const int64_t* labels, const float* cond, const int32_t* cond_mask,
                             float* eps_out, int B);
/* `model.noise_in_cond` nets draw z ~ N(0,1) shaped like cond in EVERY forward and feed sqrt(a[t]) cond + sqrt(1 - a[t]) z
 * (a = alphas, t = the row's label) to the network (ncsnpp_more.py:755-768).  This sets where the following forwards take z from:
 * z_device != NULL: [n, B, C*nc, S, S] consumed one [B, ...] slab per forward (parity runs, or z drawn by the caller the way the
 * reference does with torch); NULL: the on-device Philox stream keyed by (seed, sample_offset + row, forward counter), continuing
 * from first_draw.  With model.gamma the caller passes z already standardised ((g - k theta) / sqrt(1 - a), :761-765) or lets the
 * library draw it (mcvd_sampler_run only: all rows share the label there). */
int mcvd_model_set_cond_noise(mcvd_model* m, const float* z_device, uint64_t seed, uint64_t sample_offset, uint64_t first_draw);
/* Gamma-noise tables of `model.gamma` nets exactly as UNetMore_DDPM registers them (k_cum, theta_t: ncsnpp_more.py:745-749); host
 * pointers, n = num_classes.  Required before mcvd_sampler_run with MCVD_FLAG_GAMMA. */
int mcvd_model_set_gamma_tables(mcvd_model* m, const float* k_cum_host, const float* theta_t_host, int n);
/* SPADE models (model.spade): the gamma/beta modulation maps depend only on the conditioning frames (layerspp.py:164-168), so
 * they are computed by mcvd_model_prepare_cond and cached; later forwards that pass the SAME cond pointer and batch size reuse
 * them until mcvd_model_invalidate_cond / another prepare (the caller promises not to modify cond in between).  A forward whose
 * (cond, B) was not prepared computes the maps itself, every call.  mcvd_sampler_run prepares once per call.  No-ops for concat models. */
int mcvd_model_prepare_cond(mcvd_model* m, const float* cond, int B);
int mcvd_model_invalidate_cond(mcvd_model* m);
/* Kernel selection table of batch size B (filled by the autotuner on the first forward at B; one (tile shape id, cout tile) pair
 * per plan op, -1 / 0 for non-conv ops).  Exporting it after a warm-up and importing it into a later process pins the kernel
 * choice: no timing launches, and bit-identical results from run to run (the choice only changes fp32 summation order, but the
 * tuner's decision is timing dependent).  get: shapes == NULL returns the entry count; returns the count or a negative error. */
int mcvd_model_get_tuning(mcvd_model* m, int B, int* shapes, int* cots, int cap);
int mcvd_model_set_tuning(mcvd_model* m, int B, const int* shapes, const int* cots, int n);
/* hipGraph replay counters of this model (option "graph"): graphs instantiated / forwards served by a graph launch. */
int mcvd_model_graph_stats(mcvd_model* m, int64_t* captures, int64_t* replays);
/* Number of kernels one forward enqueues at this batch size (for tests / DESIGN.md). */
int mcvd_model_num_launches(mcvd_model* m, int B);

/* Measurement aid (bench.py roofline): with ctx option "profile"=1 the first UNet forward of every mcvd_sampler_run is bracketed
 * op by op with HIP events on the ctx stream (no extra synchronisation).  This call synchronises and returns, per op of that
 * forward: kind (0 temb,1 dense,2 groupnorm-coef,3 conv,4 fir,5 attention), conv kernel size (else 0), elapsed ms, algorithmic
 * flops and algorithmic bytes.  Call with kinds == NULL to get the op count.  Returns the op count or a negative error. */
int mcvd_model_profile_read(mcvd_model* m, int* kinds, int* ks, double* ms, double* flops, double* bytes, int cap);

/* Static description of op i of the plan: info = {kind, reference module index, conv ks, H, Cin, Cout, has_residual, has_prologue}. */
int mcvd_model_op_info(mcvd_model* m, int i, int info[8]);

/* The kernel family conv op i REALLY ran at its last launch (the ids of mcvd_last_conv_kernel: 0-3 direct implicit GEMM, 4 / 8 fp32
 * Winograd, 5 / 6 / 9 fp32 1x1 GEMM, 10 / 11 three-piece bf16 Winograd, 12 / 13 two-piece fp16 Winograd, 14 two-piece fp16 1x1 GEMM,
 * 15 three-piece bf16 1x1 GEMM; -2 the one-thread-per-output test kernel); -1 for an op that is not a conv or never ran.  Tests
 * use it to assert that a forced or imported kernel table is what executed (a graph replay re-runs what its capture recorded). */
int mcvd_model_op_kernel(mcvd_model* m, int i);

/* Launch counters of the fused forms that have no reference counterpart (diagnostics / tests), cumulative over the model's forwards:
 * what = 0: attention blocks whose K and V went from the q|k|v projection to the attention kernel pre-split (option "attn_presplit",
 * layerspp.py:236-245 computes the same products on fp32 rows); 1: unused (always 0);
 * 2: convs that took the SPADE modulation inside their loader (option "spade_fuse"); 3: norms whose table the producing
 * conv's K-split reduce pass wrote ("gn_producer"). */
long mcvd_model_fused_launches(mcvd_model* m, int what);

/* Debug/test aid: copy the output tensor of reference module `module` (index in all_modules, ncsnpp_more.py:249) from the
 * last forward at batch size B into dst_device ([B, C, H, H], capacity in floats).  The workspace keeps every intermediate of a
 * forward, so this needs no re-execution.  Module 1 returns SiLU(temb) [B, 4*ngf] (C = 4*ngf, H = 0). */
int mcvd_model_module_output(mcvd_model* m, int module, int B, float* dst_device, int64_t capacity, int* C, int* H);

/* ---- sampler ---------------------------------------------------------------------- */
/* The whole L-step loop on device: schedule subsampling (:229-237), labels, forward, x0/clip/posterior (+noise),
 * t_min re-noise (:269-280), denoise pass with label L-1 (:331-333).  x_inout:[B,C*nf,S,S] is overwritten.
 * noise: NULL -> counter-based Philox keyed by (seed, sample_offset + row, step) so results do not depend on how rows are
 * sharded over GPUs; else [n_draws, B, C*nf, S, S] consumed in draw order (t_min draw first, then one per step).
 * MCVD_FLAG_GAMMA: the draws are gamma variates g ~ Gamma(k_cum[i], scale theta_t[i]) standardised as (g - k theta) / sqrt(1 - a_i);
 * an injected `noise` then holds the RAW g (what Gamma(...).sample() returns in the reference).
 * t_min is a DOUBLE, as the Python float the reference multiplies by the table length (:269): step i is skipped iff
 * `steps[i] < t_min * L` evaluated as the reference evaluates it -- in float32 when the schedule was subsampled (there `step` is a
 * 0-dim int64 tensor and torch compares it with a Python scalar in the default dtype), in double otherwise (numpy int64 vs float).
 * A float parameter would move e.g. t_min = 0.1 to 0.100000001 and skip the step that sits exactly on the threshold. */
int mcvd_sampler_run(mcvd_model* m, int kind, float* x_inout, const float* cond, const float* noise, uint64_t seed,
                     uint64_t sample_offset, int subsample_steps, int flags, double t_min, int B);
/* The whole F-PNDM loop on the device (FPNDM_sampler, models/__init__.py:38-99 with models/pndm.py: Runge-Kutta for the first three
 * steps, 4th-order Adams-Bashforth afterwards; deterministic).  flags: MCVD_FLAG_CLIP_BEFORE.  x_inout is overwritten with the last step. */
int mcvd_fpndm_run(mcvd_model* m, float* x_inout, const float* cond, int subsample_steps, int flags, int B);
/* One fused update (host-driven loops, final_only=False / verbose paths).  Coefficients are the fp32 scalars the
 * reference computes: x0 = c_x0a*(x - c_x0b*eps); clip; x = c_mean0*x0 + c_mean1*(ddpm: x | ddim: eps) + c_noise*z. */
int mcvd_sampler_update(mcvd_ctx* ctx, int kind, float* x_inout, const float* eps, const float* noise, float c_x0a,
                        float c_x0b, float c_mean0, float c_mean1, float c_noise, int clip, int64_t n);
/* z ~ N(0,1) from the same Philox stream mcvd_sampler_run uses: out:[B, per_sample]. */
int mcvd_randn(mcvd_ctx* ctx, float* out, uint64_t seed, uint64_t sample_offset, uint64_t draw, int B,
               int64_t per_sample);
/* uint8 frame packing of the result side (runners/ncsn_runner.py:2019-2062: each frame BCHW -> HWC, `(frame * 255).astype('uint8')`):
 * frames01:[B, T*C, H, W] fp32 in [0, 1] (inverse_data_transform's output, frame-major channels) -> out:[B, T, H, W, C] uint8. */
int mcvd_pack_frames_u8(mcvd_ctx* ctx, const float* frames01, uint8_t* out, int B, int T, int C, int H, int W);
/* Standardised gamma noise of the `gamma=True` samplers (models/__init__.py:273-276, :319-322): out = (g - kt) / sd, g = raw[i]
 * when raw != NULL (a Gamma(k, rate 1/theta).sample() drawn elsewhere) else theta * Gamma(k) from the library's Philox stream;
 * kt = k * theta and sd = sqrt(1 - alpha_i) are passed as the fp32 scalars the reference computes.  out:[B, per_sample]. */
int mcvd_gamma_noise(mcvd_ctx* ctx, float* out, const float* raw, float k, float theta, float kt, float sd, uint64_t seed,
                     uint64_t sample_offset, uint64_t draw, int B, int64_t per_sample);
/* F-PNDM building blocks (models/pndm.py), fp32 with one rounding per operation in the order of the reference's tensor
 * expressions (no FMA contraction), n elements, device pointers:
 *   mcvd_lincomb:        out = scale * (((w0*in0 + w1*in1) + w2*in2) + w3*in3), the first nin (1..4) terms; out may alias an input
 *                        (runge_kutta :15: scale 1/6, w = 1,2,2,1;  gen_order_4 :47: scale 1/24, w = 55,-59,37,-9)
 *   mcvd_pndm_transfer:  out = clip?( x + d * (c1 * x - c2 * e) ) with d = a_next - a, c1 = 1/(sqrt(a)(sqrt(a)+sqrt(a_next))),
 *                        c2 = 1/(sqrt(a)(sqrt((1-a_next)a) + sqrt((1-a)a_next)))          (transfer :19-33) */
int mcvd_lincomb(mcvd_ctx* ctx, float* out, const float* in0, const float* in1, const float* in2, const float* in3, float w0,
                 float w1, float w2, float w3, float scale, int nin, int64_t n);
int mcvd_pndm_transfer(mcvd_ctx* ctx, float* out, const float* x, const float* e, float d, float c1, float c2, int clip,
                       int64_t n);

/* ---- stand-alone ops (unit parity against the oracle; also the reference's only native op) ----------------- */
/* upfirdn2d on [N, C, H, W] planes; kernel:[kh,kw] HOST pointer.  out:[N,C,oh,ow], oh=(H*up+pad0+pad1-kh)/down+1. */
int mcvd_upfirdn2d(mcvd_ctx* ctx, const float* in, const float* kernel_host, int kh, int kw, int up, int down, int pad0,
                   int pad1, float* out, int N, int C, int H, int W);
/* y = act_scale( conv_{ks}( prologue(x) ) + bias [+ res] ) with prologue(x) = silu?(coefA*x + coefB) per (b, channel).
 * x0:[B,C0,H,W], x1:[B,C1,H,W] (virtual channel concat; x1 may be NULL), w:[Cout,C0+C1,ks,ks] reference layout (device),
 * coef:[B, C0+C1, 2] or NULL, res:[B,Cout,H,W] or NULL. */
int mcvd_op_conv2d(mcvd_ctx* ctx, const float* x0, int C0, const float* x1, int C1, const float* w, const float* bias,
                   int Cout, int ks, const float* coef, int act, const float* res, float out_scale, float* y, int B, int H,
                   int W);
/* Which kernel family the calling thread's last conv launch (mcvd_op_conv2d or a model forward) was dispatched to: the ids of the
 * "conv_shape" option (0..3 direct implicit-GEMM tile shapes, 4 / 8 fp32 Winograd, 5 / 6 / 9 fp32 1x1 GEMM, 10 / 11 three-piece bf16
 * Winograd, 12 / 13 two-piece fp16 Winograd, 14 / 15 split-operand 1x1 GEMM); -1 none yet.  A forced "conv_shape" that does not apply to a
 * launch falls back -- tests use this to assert what really ran (per op of a model: mcvd_model_op_kernel). */
int mcvd_last_conv_kernel(void);
/* SPADE prologue of the Winograd conv kernel (layerspp.py:164-171, :530-535): while set (non-NULL gb), mcvd_op_conv2d computes
 * conv(silu(((A x + B)(1 + gamma) + beta) * s1 + b2)) with (A, B) = coef, gamma | beta = gb:[B, 2*Cin, H, W] and (s1, b2) =
 * coef2:[B, Cin, 2] (NULL: (1, 0)); needs conv_shape 4 or 8, ks 3, coef and act.  Inside a model forward this is what a SPADE net's
 * non-resampling norms use (ctx option "spade_fuse"). */
int mcvd_ctx_set_spade_inputs(mcvd_ctx* ctx, const float* gb, const float* coef2);
/* GroupNorm statistics from the producing conv's epilogue.  When a buffer is set (device floats, >= B*Cout*(H*W/32)*2; NULL
 * disables), mcvd_op_conv2d's MFMA kernels also write, for every (sample, cout), np partial pairs (sum, M2 about the partial's own
 * mean) over disjoint sets of H*W/np output pixels: stats[((b*Cout + co)*np + p)*2 + {0,1}].  np depends on the kernel family that
 * ran -- mcvd_last_conv_stats_np(): Winograd H*W/128 (8x8 images: 1), Winograd K split 1, all-DMA 1x1 H*W/32, 0 = that kernel
 * does not emit (direct implicit-GEMM tiles) and a consumer must take mcvd_op_gn_coef's pass over the tensor instead.
 * mcvd_op_gn_finalize folds the partials of one tensor or of a virtual channel concat of two (each with its own np) into the same
 * (A, B) coefficients mcvd_op_gn_coef produces (modes / p0 / p1 / emb_* as there).  Inside a model forward this is automatic
 * (ctx option "gn_stats", default 1). */
int mcvd_ctx_set_stats_buffer(mcvd_ctx* ctx, float* device_floats);
int mcvd_last_conv_stats_np(void);
int mcvd_op_gn_finalize(mcvd_ctx* ctx, const float* st0, int C0, int np0, const float* st1, int C1, int np1, int groups, float eps,
                        int mode, const float* p0, const float* p1, int emb_stride, int emb_off, float* coef_out, int B, int HW);
/* GroupNorm statistics folded to per-(b,c) affine coefficients: y = A*x + B.
 * mode 0: plain (A=rstd, B=-mean*rstd); mode 1: temb scale/shift, emb:[B, emb_stride] with scale at emb_off+c and shift at
 * emb_off+C+c (layerspp.py:521-535); mode 2: affine weight/bias:[C] (torch GroupNorm affine=True). */
int mcvd_op_gn_coef(mcvd_ctx* ctx, const float* x0, int C0, const float* x1, int C1, int groups, float eps, int mode,
                    const float* p0, const float* p1, int emb_stride, int emb_off, float* coef_out, int B, int HW);
/* Multi-head self-attention core on qkv:[B, 3C, HW] (q rows 0..C-1, k rows C..2C-1, v rows 2C..3C-1, heads are
 * contiguous channel chunks, layerspp.py:237-244) -> out:[B, C, HW]. */
int mcvd_op_attention(mcvd_ctx* ctx, const float* qkv, float* out, int B, int C, int heads, int HW);
/* FIR x2 resample with k=[1,3,3,1] (up: gain 4) and optional prologue silu?(A*x+B): the fused form used in ResBlocks. */
int mcvd_op_fir2(mcvd_ctx* ctx, const float* x, const float* coef, int act, int up, float* y, int B, int C, int H, int W);

#ifdef __cplusplus
}
#endif
#endif /* MCVD_HIP_H */
