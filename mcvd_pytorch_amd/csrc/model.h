// Host-side executor: config -> static op plan, parameter blob, weight packing, forward, sampler loop.
#pragma once
#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "../../include/mcvd_hip.h"
#include "common.h"

struct mcvd_ctx {
    int device = 0;
    int shared_device = 0;         // 1: another PROCESS held a context on this device when this one was created and MCVD_ALLOW_SHARED_DEVICE=1
                                   //    let it in (api.cpp: device lock): the three-piece bf16 attention kernel stays off the device
    hipStream_t stream = nullptr;
    int naive_conv = 0;
    int naive_attn = 0;
    int gn_producer = 1;       // the K-split reduce pass of a conv also finalizes the norm over its output (gn.cpp: ksplit_reduce_gn_kernel); 0: two launches
    int dbg_skip_finalize = 0; // timing-only: no gn_finalize launches inside a captured graph (wrong results)
    int fir_form = 0;          // 0: FIR x2 resamplers through the LDS where the geometry applies; 1: register forms only (A/B)
    int graph = 0;                 // 1: replay each forward as a hipGraph (captured on the second use of the same
                                   //    (x, labels, cond, out, B) pointer set; the sampler loop reuses one set for all steps)
    hipStream_t cap = nullptr;     // private capture stream (the caller's stream may be the legacy default stream, which
                                   //    cannot be captured)
    std::atomic<unsigned> epoch{0};   // bumped by every option change (and by ctx_register from another thread): invalidates captured graphs
    int conv_shape = -1;           // -1 auto, else force a conv tile shape / kernel family (tests)
    int conv_shape1 = -1;          // >= 0: the shape forced for the 1x1 convs only (they follow conv_shape otherwise): lets a test put
                                   //    every 3x3 conv AND every 1x1 conv of a model on chosen kernels at once
    int winograd = 1;              // offer the Winograd F(2x2,3x3) kernel to the autotuner (3x3 convs, H%8==0, W%16==0)
    int persist_grid = 0;          // > 0: workgroups of the persistent Winograd kernel (tests); 0 = one per CU
    int temb_table = 1;            // device-loop samplers: time MLP + Dense_0 projections once per call for all its labels (mcvd_model::prepare_temb_table)
    int im2col_lds = 1;            // shape id 23 (the stem as a GEMM): the im2col is staged in LDS by the GEMM kernel itself (conv1x1_h2.cpp IM) where its
                                   //    geometry applies; 0 = materialised in HBM by im2col3x3_kernel first (round 5's form; bit-identical results)
    int wino_selftest = 0;         // 0 not run yet, 1 passed, -1 FAILED: the hand-scheduled bf16 Winograd kernels disagree with the fp32-MFMA Winograd
                                   //    kernel on this device / driver (mcvd_ctx_selftest); bf16x3 was switched off for this context
    int conv_cot = 0;              // > 0 with conv_shape 5: cout tile (32-channel units) mcvd_op_conv2d requests (tests)
    int bf16x3 = 1;                // offer the three-piece bf16 kernels (conv_wino3.cpp, conv1x1_h2.cpp, attention_h2.cpp with NP = 3: fp32-equivalent
                                   //    arithmetic, full fp32 range) to the autotuner / the attention dispatch.  On by default.
    int f16x2 = 0;                 // offer the two-piece fp16 kernels (conv_wino2h.cpp, conv1x1_h2.cpp, attention_h2.cpp with NP = 2: 22-bit operands,
                                   //    fp16 exponent range, fp32 accumulate) as well.  OFF by default: narrower arithmetic than the reference's.
                                   //    Convs with a raw (not normalised) input never take them; see model.cpp "f16x2 range guard"
    int conv_dma1 = 1;             // offer the all-DMA 1x1 GEMM kernel (conv1x1_dma.cpp) to the autotuner
    int conv_wdma = 1;             // weight chunks by LDS-DMA (1) or register staging (0)
    int share_fence = 0;           // 1: a context that shares its device keeps the split-operand attention kernels off it (the round-5 workaround; the cause is gone)
    int side_stream = 0;           // 1: run the ResBlock shortcut 1x1 convs on a second stream (measured -2 %: off by default)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int spade_fuse = 0;            // 1: SPADE modulation inside the Winograd conv loader (gamma | beta by LDS-DMA); 0: through spade_apply.
                                   //    Off by default: measured 3.5 % SLOWER end to end (profiles/r02_spade_fusion_ab.txt)
    int attn_presplit = 1;         // the q|k|v projection writes K and V pre-split (three bf16 pieces, LDS-image order) and the attention kernel
                                   //    stages them by LDS-DMA (attn_h2p_kernel; head dims 32 / 64 / 96, default arithmetic only); 0: attn_h2_kernel
                                   //    splits them itself, once per query tile.  Bit-identical results either way.
    int gn_stats = 1;              // GroupNorm statistics from the producing conv's epilogue (0: always one pass over the tensor)
    int autotune = 1;              // time the conv tile candidates per distinct layer shape on first use of a batch size
    int profile = 0;               // record HIP events around every op of the first forward of each sampler call
    unsigned long long* dbg = nullptr;   // conv phase-timing buffer for mcvd_op_conv2d (diagnostics)
    const float* spade_gb = nullptr;     // mcvd_op_conv2d: SPADE prologue inputs of the next convs (mcvd_ctx_set_spade_inputs; tests)
    const float* spade_coef2 = nullptr;
    float* stats_buf = nullptr;          // mcvd_op_conv2d: where the conv's GroupNorm partials go (mcvd_ctx_set_stats_buffer; tests)
    int* range_flag = nullptr;     // device word: set by nonfinite_flag_kernel after a forward under f16x2 (mcvd_ctx_check_range reads + clears it)
    float* scratch = nullptr;      // small device scratch for stand-alone ops (kernel taps, packed weights)
    size_t scratch_bytes = 0;
    int ensure_scratch(size_t bytes);
};

bool mcvd_ctx_shares_device(const mcvd_ctx* c);     // api.cpp: another process, or another stream of this process, runs on the context's device

namespace mcvd {

struct ParamInfo {
    std::string name;
    int64_t shape[4] = {0, 0, 0, 0};
    int ndim = 0;
    int64_t off = 0;       // float offset in the raw parameter blob
    int64_t numel = 0;
    bool set = false;
};

// where a tensor lives: workspace arena (offset in floats PER SAMPLE), or one of the caller's buffers
enum RefKind { REF_NONE = 0, REF_ARENA, REF_X, REF_COND, REF_OUT };
struct TRef {
    RefKind kind = REF_NONE;
    int64_t off = 0;
    int C = 0;
};

enum OpKind { OP_TEMB, OP_DENSE, OP_GN, OP_CONV, OP_FIR, OP_ATTN, OP_NEAREST, OP_COEF2, OP_APPLY, OP_CONDNOISE };

struct Op {
    OpKind kind;
    int module = -1;           // index in all_modules (for diagnostics)
    TRef src0, src1, dst, res;
    int H = 0, W = 0;
    int Cout = 0, ks = 0;
    // GroupNorm -> coef
    int groups = 0;
    float eps = 0.f;
    int gn_mode = 0;
    int64_t p0 = -1, p1 = -1;      // raw blob offsets (affine weight/bias)
    int emb_off = 0;
    TRef coef;                      // [2*C] per sample
    int act = 0;
    // conv
    int64_t wp = -1, bias = -1;    // packed blob offsets
    int64_t wpw = -1;              // Winograd-transformed weights (3x3 convs at supported resolutions), else -1
    int64_t wph = -1;              // the same, pre-split into two fp16 pieces (conv_wino2h.cpp / conv1x1_h2.cpp), else -1
    int64_t wpb = -1;              // the same, pre-split into three bf16 pieces (conv_wino3.cpp / conv1x1_h2.cpp), else -1
    int CinP = 0, CoutP = 0, cot = 0;
    float out_scale = 1.f;
    // 3x3 conv that can also run as a 1x1 GEMM on the three-piece bf16 kernel (kernels/conv_gemm_forms.cpp): alt_kind = 23 im2col (few
    // input channels: the stem), 22 taps as outputs (few output channels: the last conv); the GEMM's packed matrix / pieces / zero bias,
    // its padded dims and the per-sample buffer (the im2col rows, or the 9 * Cout planes of z)
    int alt_kind = 0;
    int64_t alt_wp = -1, alt_wpb = -1, alt_bias = -1;
    int alt_CinP = 0, alt_CoutP = 0;
    TRef alt_buf;
    // fir
    int up = 0;
    // attention
    int heads = 0;
    TRef kv;                       // q|k|v projection and its attention op: K and V as three-piece bf16 LDS images (3 C HW dwords per sample;
                                   // conv1x1_h2.cpp KV epilogue -> attn_h2p_kernel); which form a forward uses: mcvd_model::kv_live
    int attn_op = -1;              // q|k|v projection: index of the attention op that consumes it
    // SPADE: gb = cached [2C] (gamma | beta) maps, coef2 = (1 + scale, shift) per (sample, channel)
    TRef gb, coef2;
    TRef tmp;                      // conv with a SPADE norm in front: where spade_apply materialises its input when the kernel cannot fuse it
    TRef dst2;                     // FIR: second output (raw-input resampling for the shortcut path)
    bool side = false;             // independent of the main chain until `join`: may run on the side stream
    bool join = false;             // must wait for the preceding side op
    bool prep = false;             // depends on the conditioning frames only: runs once per cond, not per step
    // GroupNorm statistics from the producer's epilogue: a conv whose output is normalised later writes partial (sum, M2) pairs
    // to `stats` ([C][H*W/32][2] floats per sample reserved; how many partials are used depends on the kernel that runs); a
    // GroupNorm op names the plan ops that produce its sources (-1: not a conv of this plan)
    TRef stats;
    int prod0 = -1, prod1 = -1;
    int gn_next = -1;              // conv: the plan's single-source OP_GN over this conv's output that comes first (its (A, B) table may be written by
                                   // the conv's own last pass: ConvArgs::gno), else -1
    int gn_src = -1;               // consumers of GroupNorm coefficients (conv, FIR, SPADE apply): the plan's OP_GN that computes `coef`
};

struct DenseEntry {
    std::string weight, bias;
    int ch;
    int emb_off;
};

struct ConvPack {
    std::vector<std::string> weights;   // 1 entry, or 3 for the fused q|k|v projection
    std::vector<std::string> biases;
    int Cout_each, Cin, ks, CinP, CoutP, nin;
    int64_t wp, bias;
    int64_t wpw = -1;
    int64_t wph = -1;
    int64_t wpb = -1;
    int alt_kind = 0;                   // 22 / 23: the conv's GEMM form is packed too (Op::alt_kind)
    int64_t alt_wp = -1, alt_wpb = -1;
    int alt_CinP = 0, alt_CoutP = 0;
    bool zero_bias = false;             // the packed bias stays zero (the shortcut GEMM of an up block: its bias is added elsewhere)
    std::string extra_bias;             // a second bias parameter added into this conv's packed bias (that shortcut's)
};

}  // namespace mcvd

struct mcvd_model {
    mcvd_ctx* ctx = nullptr;
    mcvd_unet_desc d;
    bool finalized = false;

    std::vector<mcvd::ParamInfo> params;
    std::map<std::string, int> pindex;
    int64_t blob_floats = 0;
    float* blob = nullptr;            // raw parameters (device)

    std::vector<mcvd::Op> ops;
    std::vector<mcvd::DenseEntry> dense;
    std::vector<mcvd::ConvPack> packs;
    int64_t packed_floats = 0;
    float* packed = nullptr;          // kernel-layout weights (device)
    long long* coef2_desc_dev = nullptr;   // SPADE nets: {arena offset, emb_off, C} of every (1 + scale, shift) table (launch_coef2_all), uploaded at finalize
    int coef2_first = -1, coef2_count = 0, coef2_cmax = 0;      // plan index of the first OP_COEF2 (it fills every table), number of tables, widest
    float* packed_h = nullptr;        // the two-piece fp16 forms (ConvPack::wph offsets), allocated + packed on first use under the option f16x2
    int64_t packed_h_floats = 0;
    bool packed_h_valid = false;      // cleared by mcvd_model_finalize (new weights)
    int ensure_f16x2_weights();
    int64_t dense_wt = -1, dense_bias = -1, freqs_off = -1;
    int NE = 0;                       // total Dense_0 outputs
    int T = 0;                        // temb width (4*ngf, + ngf/2 with cond_emb)
    int first_module = 2;             // all_modules index of the stem conv (3 with cond_emb: module 2 is the mask embedding)
    mcvd::TRef cond_src;              // what the network reads as conditioning frames: the caller's cond, or its noised copy
    const int32_t* cond_mask = nullptr;      // cond_emb: mask of the forward in flight (NULL = ones)
    float* alphas_dev = nullptr;      // [num_classes] device copy of alphas (noise_in_cond)
    bool alphas_dev_valid = false;
    float* cond_z = nullptr;          // [arena_B * C*nc*S*S] scratch for the conditioning noise of one forward
    const float* cond_noise_src = nullptr;   // injected z sequence (advances one slab per forward) or NULL = Philox
    uint64_t cond_noise_seed = 0, cond_noise_offset = 0, cond_noise_draw = 0;
    float cond_gamma_kt = 0.f, cond_gamma_sd = 1.f;
    float* noise_buf = nullptr;       // [arena_B * C*nf*S*S] standardised gamma draws of one sampler step (model.gamma)
    float cond_gamma_k = 0.f, cond_gamma_theta = 0.f;   // > 0: library-drawn conditioning noise is a standardised gamma variate (sampler loop)
    std::vector<float> k_cum, theta_t;

    int64_t arena_per_sample = 0;     // floats
    float* arena = nullptr;
    int arena_B = 0;
    int64_t* labels = nullptr;        // [arena_B] (sampler-owned labels)
    float* eps_buf = nullptr;         // [arena_B * C*nf*S*S] (sampler-owned eps)
    float* labels_f = nullptr;        // [arena_B] float labels (F-PNDM midpoints)
    float* fp_buf = nullptr;          // F-PNDM device loop: 9 state-sized buffers (4 eps history, 3 Runge-Kutta eps, x temp, combination)
    int fp_B = 0;
    float* ksplit_buf = nullptr;      // partial outputs of the K-split Winograd layers (H*W <= 256): ensure_ksplit sizes it per batch
    size_t ksplit_floats = 0;

    std::vector<float> betas, alphas, alphas_prev, freqs;

    // conv tile choice per op for the batch size it was tuned at: (shape, cot); filled by autotune()
    std::vector<int> stats_np;        // per op: partials per (sample, channel) its last launch wrote (0: none)
    std::vector<signed char> kv_live;       // per OP_ATTN of the forward in flight: 1 = its projection wrote the K / V piece images (set by the conv launch)
    long fused_launches[4] = {0, 0, 0, 0};   // mcvd_model_fused_launches: pre-split attention blocks, fused SPADE norms, convs with the SPADE loader,
                                             // norms finalized by their producer's K-split reduce pass
    std::vector<signed char> gn_done;       // per OP_GN of the forward in flight: 1 = its table was written by the producing conv's last pass
    int launch_gn(const mcvd::Op& op, const float* x, const void* labels, const float* cond, float* out, int B);
    std::vector<int> ran_kernel;      // per conv op: the kernel family its last launch REALLY ran (last_conv_kernel(); -2 the naive kernel,
                                      //    -1 never launched): what mcvd_model_op_kernel reports
    std::vector<int> tuned_shape, tuned_cot;
    int tuned_B = 0;
    int tuned_sig = -1;            // the kernel-offer options (winograd, conv_dma1, bf16x3, f16x2, spade_fuse, conv_wdma) the tables were tuned under
    // every batch size tuned (or imported through mcvd_model_set_tuning) so far: alternating batch sizes do not re-tune
    std::map<int, std::pair<std::vector<int>, std::vector<int>>> tuned_cache;
    int autotune(int B);
    void sync_tuning_options();

    // per-op HIP event timing (bench.py roofline): events are recorded on the ctx stream around each op of ONE forward
    std::vector<hipEvent_t> ev;
    bool profile_armed = false;
    int profile_B = 0;

    // SPADE modulation cache: valid for (cond pointer, batch) after prepare_cond()
    const float* prepared_cond = nullptr;
    int prepared_B = 0;
    bool cond_cache_valid = false;
    bool has_prep = false;
    int prepare_B(int B);                                   // workspace + autotune for a batch size
    int run_prep(const float* cond, int B);                 // the cond-only ops
    int prepare_cond(const float* cond, int B);             // run_prep + mark the cache valid

    // hipGraph replay of one forward (ctx option "graph")
    struct GraphKey {
        const float* x = nullptr; const void* lab = nullptr; const float* cond = nullptr; float* out = nullptr;
        const void* mask = nullptr;
        int B = 0, labels_f32 = 0; unsigned epoch = 0, ctx_epoch = 0;
        bool operator==(const GraphKey& o) const {
            return x == o.x && lab == o.lab && cond == o.cond && out == o.out && mask == o.mask && B == o.B &&
                   labels_f32 == o.labels_f32 && epoch == o.epoch && ctx_epoch == o.ctx_epoch;
        }
    };
    GraphKey graph_key, graph_seen;        // key of the instantiated graph / of the last eager forward
    hipGraphExec_t graph_exec = nullptr;
    unsigned epoch = 0;                    // bumped when the workspace, the tile choice or the packed weights change
    long graph_replays = 0, graph_captures = 0;
    void drop_graph();
    int forward_ops(const float* x, const void* labels, const float* cond, float* out, int B);   // the plain launch sequence

    hipStream_t op_stream = nullptr;                        // overrides ctx->stream for the op being launched (side stream)
    int build_plan();
    int add_param(const std::string& name, std::initializer_list<int64_t> shape);
    int find_param(const char* name) const;
    int ensure_workspace(int B);
    int ensure_ksplit(int B);                               // partial-output buffer of the K-split layers, for the deepest split selectable at B
    // Time-embedding table of a device-loop sampler call (option "temb_table"): the labels of ALL its forwards are known before the first one
    // (models/__init__.py:229-237, :283, :332), and the time MLP + every Dense_0 projection depend on nothing else -- so both run ONCE per call
    // for all L + 1 labels (two launches instead of 2 (L + 1)), and each forward starts with a 46 KB device copy of its row instead of the two
    // latency-bound launches at its head (34 + 13 us of a 12.2 ms forward on config 2, 36 + 20 us of 5.3 ms on config 4).  Same kernels, rows are
    // independent: bit-identical.
    float* temb_tab = nullptr;        // [rows][NE]   Dense_0 outputs per forward of the call
    float* temb_tmp = nullptr;        // [rows][T]    silu(temb) per forward of the call
    int64_t* temb_lab = nullptr;      // [rows]
    size_t temb_rows_cap = 0;
    int temb_row_live = 0;            // 1: OP_TEMB / OP_DENSE are skipped, ops[1].dst holds the row use_temb_row copied there
    int profile_temb_skipped = 0;     // the instrumented forward mcvd_model_profile_read describes ran with temb_row_live
    int prepare_temb_table(const std::vector<int>& labels);
    int use_temb_row(int row, int B);
    int uniform_labels = 0;        // every row of the forward in flight carries the SAME label (the sampler loops): the time MLP and
                                   //    the Dense_0 projections are evaluated for one row and read with stride 0
    int labels_f32 = 0;            // the labels of the forward in flight are float [B] instead of int64 [B] (mcvd_unet_forward_ft)
    int forward(const float* x, const void* labels, const float* cond, float* out, int B);
    int forward_unchecked(const float* x, const void* labels, const float* cond, float* out, int B);
    int launch_op(const mcvd::Op& op, const float* x, const void* labels, const float* cond, float* out, int B);
    // 3x3 conv as a 1x1 GEMM on the three-piece kernel (shape ids 22 / 23; kernels/conv_gemm_forms.cpp)
    bool gemm_form_usable(const mcvd::Op& op, const mcvd::ConvArgs& a) const;
    mcvd::ConvArgs gemm_form_args(const mcvd::Op& op, const mcvd::ConvArgs& a, float* buf) const;
    int launch_gemm_form(const mcvd::Op& op, const mcvd::ConvArgs& a, float* buf, hipStream_t s);
    float* resolve(const mcvd::TRef& r, const float* x, const float* cond, float* out, int B) const;
};
