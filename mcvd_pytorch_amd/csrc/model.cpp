// Static op plan for the `unetmore` UNet (reference: models/better/ncsnpp_more.py:70-249 construction order,
// :251-392 forward order), parameter blob, weight packing and the forward executor.
#include "model.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <tuple>

namespace mcvd {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

static int gn_groups(int ch) {      // layerspp.py:474-476 (also :212-214, :128-130)
    int g = std::min(ch / 4, 32);
    while (g > 1 && ch % g != 0) --g;
    return g < 1 ? 1 : g;
}

static int64_t align4(int64_t v) { return (v + 3) & ~(int64_t)3; }

}  // namespace mcvd

using namespace mcvd;

int mcvd_ctx::ensure_scratch(size_t bytes) {
    if (bytes <= scratch_bytes) return 0;
    if (scratch) {
        MCVD_HIP_CHECK(hipStreamSynchronize(stream));
        MCVD_HIP_CHECK(hipFree(scratch));
        scratch = nullptr;
        scratch_bytes = 0;
    }
    MCVD_HIP_CHECK(hipMalloc((void**)&scratch, bytes));
    scratch_bytes = bytes;
    return 0;
}

int mcvd_model::add_param(const std::string& name, std::initializer_list<int64_t> shape) {
    ParamInfo p;
    p.name = name;
    p.ndim = (int)shape.size();
    p.numel = 1;
    int i = 0;
    for (int64_t s : shape) {
        p.shape[i++] = s;
        p.numel *= s;
    }
    p.off = blob_floats;
    blob_floats += align4(p.numel);
    pindex[name] = (int)params.size();
    params.push_back(p);
    return (int)params.size() - 1;
}

int mcvd_model::find_param(const char* name) const {
    std::string n(name);
    if (n.rfind("module.", 0) == 0) n = n.substr(7);      // DataParallel prefix (ncsn_runner.py:426-433)
    auto it = pindex.find(n);
    return it == pindex.end() ? -1 : it->second;
}

namespace {

struct Act {
    TRef a, b;      // b.kind == REF_NONE unless this is a virtual concat [a, b]
    int H = 0;
    int C() const { return a.C + b.C; }
};

struct Builder {
    mcvd_model& m;
    int64_t arena = 0;
    int64_t packed = 0;
    explicit Builder(mcvd_model& mm) : m(mm) {}

    TRef alloc_floats(int64_t per_sample, int C) {
        TRef r;
        r.kind = REF_ARENA;
        r.off = arena;
        r.C = C;
        arena += align4(per_sample);
        return r;
    }
    TRef alloc(int C, int H) { return alloc_floats((int64_t)C * H * H, C); }
    int64_t alloc_packed(int64_t floats) {
        const int64_t o = packed;
        packed += align4(floats);
        return o;
    }
    int64_t packed_h = 0;                     // the two-piece fp16 weight forms live in a buffer of their own (mcvd_model::packed_h):
    int64_t alloc_packed_h(int64_t floats) {  // allocated and packed only when the option f16x2 is on (ensure_f16x2_weights)
        const int64_t o = packed_h;
        packed_h += align4(floats);
        return o;
    }

    int dense_entry(const std::string& prefix, int ch) {
        m.add_param(prefix + ".Dense_0.weight", {2 * ch, m.T});
        m.add_param(prefix + ".Dense_0.bias", {2 * ch});
        DenseEntry e{prefix + ".Dense_0.weight", prefix + ".Dense_0.bias", ch, m.NE};
        m.NE += 2 * ch;
        m.dense.push_back(e);
        return e.emb_off;
    }

    // ---- SPADE (layerspp.py:101-173): gamma/beta depend only on the conditioning frames -> "prep" ops, cached
    std::map<int, TRef> seg_by_res;
    TRef seg_for(int R) {
        auto it = seg_by_res.find(R);
        if (it != seg_by_res.end()) return it->second;
        const int cond_ch = m.d.channels * m.d.num_frames_cond;
        TRef seg = alloc(cond_ch, R);
        Op op{};
        op.kind = OP_NEAREST; op.module = -1; op.prep = true; op.src0 = m.cond_src; op.dst = seg;
        op.H = op.W = R;
        m.ops.push_back(op);
        seg_by_res[R] = seg;
        return seg;
    }
    // registers the MySPADE parameters of `prefix`.Norm_0 and emits the two cond-only convs; returns gb:[2ch, R, R]
    TRef spade_prep(int module, const std::string& prefix, int ch, int R) {
        const int cond_ch = m.d.channels * m.d.num_frames_cond, sd = m.d.spade_dim;
        const std::string N = prefix + ".Norm_0";
        m.add_param(N + ".mlp_shared.0.weight", {sd, cond_ch, 3, 3});
        m.add_param(N + ".mlp_shared.0.bias", {sd});
        m.add_param(N + ".mlp_gamma.weight", {ch, sd, 3, 3});
        m.add_param(N + ".mlp_gamma.bias", {ch});
        m.add_param(N + ".mlp_beta.weight", {ch, sd, 3, 3});
        m.add_param(N + ".mlp_beta.bias", {ch});
        TRef seg = seg_for(R);
        TRef sh = alloc(sd, R);
        Op c1{};
        (void)module;                      // the modulation convs are not a reference module's output: keep them out of module_output()
        c1.kind = OP_CONV; c1.module = -1; c1.prep = true; c1.src0 = seg; c1.H = c1.W = R; c1.dst = sh;
        conv_pack(c1, {N + ".mlp_shared.0.weight"}, {N + ".mlp_shared.0.bias"}, sd, cond_ch, 3, 0);
        m.ops.push_back(c1);
        TRef gb = alloc(2 * ch, R);
        Op c2{};
        c2.kind = OP_CONV; c2.module = -1; c2.prep = true; c2.src0 = sh; c2.H = c2.W = R; c2.act = 1; c2.dst = gb;   // silu(mlp_shared) :148
        conv_pack(c2, {N + ".mlp_gamma.weight", N + ".mlp_beta.weight"}, {N + ".mlp_gamma.bias", N + ".mlp_beta.bias"}, ch, sd, 3, 0);
        m.ops.push_back(c2);
        m.has_prep = true;
        return gb;
    }
    Op coef2_op(int module, int emb_off, int ch, const TRef& dst) {
        Op op{};
        op.kind = OP_COEF2; op.module = module; op.emb_off = emb_off; op.Cout = ch; op.dst = dst;
        return op;
    }

    // registers weights (possibly several fused along Cout) and fills the conv fields of `op`
    void conv_pack(Op& op, const std::vector<std::string>& wnames, const std::vector<std::string>& bnames, int Cout_each,
                   int Cin, int ks, int nin) {
        ConvPack p;
        p.weights = wnames;
        p.biases = bnames;
        p.Cout_each = Cout_each;
        p.Cin = Cin;
        p.ks = ks;
        p.nin = nin;
        const int Cout = Cout_each * (int)wnames.size();
        p.CinP = round_up(Cin, conv_chunk(ks));
        const int cot = conv_cout_tile(Cout);
        p.CoutP = round_up(Cout, 32 * cot);
        p.wp = alloc_packed((int64_t)p.CinP * ks * ks * p.CoutP);
        p.bias = alloc_packed(p.CoutP);
        if (wnames.size() == 1 && !nin && conv_wino_supported(ks, op.H, op.W))
            p.wpw = alloc_packed((int64_t)p.CinP * 16 * p.CoutP);
        if (p.wpw >= 0) p.wph = alloc_packed_h(conv_wino2h_weight_floats(p.CinP, p.CoutP));    // two fp16 pieces (conv_wino2h.cpp): offset into packed_h
        if (p.wpw >= 0) p.wpb = alloc_packed(conv_wino3_weight_floats(p.CinP, p.CoutP));       // three bf16 pieces (conv_wino3.cpp)
        if (ks == 1) p.wph = alloc_packed_h(conv1x1_h2_weight_floats(p.CinP, p.CoutP, 2)); // two fp16 pieces of the packed matrix (conv1x1_h2.cpp): offset into packed_h
        if (ks == 1) p.wpb = alloc_packed(conv1x1_h2_weight_floats(p.CinP, p.CoutP, 3));   // three bf16 pieces of it
        // the GEMM forms of a 3x3 conv with a handful of channels on one side (conv_gemm_forms.cpp), offered to the autotuner
        if (ks == 3 && wnames.size() == 1 && !nin && op.W % 4 == 0 && (op.H * op.W) % 32 == 0 && op.H == op.W) {
            if (9 * Cout <= 64) {                        // taps as outputs: Cin -> 9 * Cout, then shift-and-add
                p.alt_kind = 22;
                p.alt_CinP = round_up(Cin, 16);
                p.alt_CoutP = round_up(9 * Cout, 32);
            } else if (9 * Cin <= 256) {                 // im2col: 9 * Cin -> Cout.  Up to 10 input channels (96 rows) either staged in LDS by the GEMM
                                                         // (conv1x1_h2.cpp IM) or materialised in HBM (im2col3x3_kernel; needs the `col` buffer); up to 28
                                                         // channels (kth / bair: 15, cityscapes: 21) in the LDS form only
                p.alt_kind = 23;
                p.alt_CinP = round_up(9 * Cin, 16);
                p.alt_CoutP = p.CoutP;
            }
            if (p.alt_kind && (p.alt_kind == 22 ? Cin % 16 != 0 : false)) p.alt_kind = 0;      // (the GEMM stages whole 16-channel chunks of real data)
            if (p.alt_kind) {
                p.alt_wp = alloc_packed((int64_t)p.alt_CinP * p.alt_CoutP);
                p.alt_wpb = alloc_packed(conv1x1_h2_weight_floats(p.alt_CinP, p.alt_CoutP, 3));
                op.alt_kind = p.alt_kind;
                op.alt_wp = p.alt_wp;
                op.alt_wpb = p.alt_wpb;
                op.alt_CinP = p.alt_CinP;
                op.alt_CoutP = p.alt_CoutP;
                op.alt_bias = p.alt_kind == 22 ? alloc_packed(p.alt_CoutP) : -1;           // (the packed blob is zero-filled: a zero bias)
                if (p.alt_kind == 22) op.alt_buf = alloc(9 * Cout, op.H);
                else if (9 * Cin <= 96) op.alt_buf = alloc(p.alt_CinP, op.H);                   // (wider stems: no `col` buffer, the LDS form or nothing)
            }
        }
        op.wpw = p.wpw;
        op.wph = p.wph;
        op.wpb = p.wpb;
        m.packs.push_back(p);
        op.ks = ks;
        op.Cout = Cout;
        op.CinP = p.CinP;
        op.CoutP = p.CoutP;
        op.cot = cot;
        op.wp = p.wp;
        op.bias = p.bias;
    }

    Op gn_op(int module, const Act& x, float eps, int mode, int emb_off, int64_t p0, int64_t p1, const TRef& coef) {
        Op op{};
        op.kind = OP_GN;
        op.module = module;
        op.src0 = x.a;
        op.src1 = x.b;
        op.H = op.W = x.H;
        op.groups = gn_groups(x.C());
        op.eps = eps;
        op.gn_mode = mode;
        op.emb_off = emb_off;
        op.p0 = p0;
        op.p1 = p1;
        op.coef = coef;
        return op;
    }

    // layerspp.py:553-624 ResnetBlockBigGANppGN / :628-705 ResnetBlockBigGANppSPADE
    int res_block(int idx, const Act& x, int cin, int cout, bool up, bool down, Act* out) {
        const std::string P = "unet.all_modules." + std::to_string(idx);
        MCVD_REQUIRE(x.C() == cin, "plan: module %d expects %d channels, got %d", idx, cin, x.C());
        MCVD_REQUIRE(!(up || down) || x.b.kind == REF_NONE, "plan: resample block %d fed by a concat", idx);
        const bool conv2 = (cin != cout) || up || down;
        const bool spade = m.d.spade != 0;
        const int H = x.H, Ho = up ? 2 * H : (down ? H / 2 : H);
        // parameter registration in state_dict order (actnorm0, Conv_0, actnorm1, Conv_1, Conv_2)
        const int e0 = dense_entry(P + ".actnorm0", cin);
        TRef gb0, gb1;
        if (spade) gb0 = spade_prep(idx, P + ".actnorm0", cin, H);          // norm runs at the PRE-resample resolution
        m.add_param(P + ".Conv_0.weight", {cout, cin, 3, 3});
        m.add_param(P + ".Conv_0.bias", {cout});
        const int e1 = dense_entry(P + ".actnorm1", cout);
        if (spade) gb1 = spade_prep(idx, P + ".actnorm1", cout, Ho);
        m.add_param(P + ".Conv_1.weight", {cout, cout, 3, 3});
        m.add_param(P + ".Conv_1.bias", {cout});
        if (conv2) {
            m.add_param(P + ".Conv_2.weight", {cout, cin, 1, 1});
            m.add_param(P + ".Conv_2.bias", {cout});
        }
        const float rs2 = 1.0f / (float)sqrt(2.0);

        // ---- actnorm0
        TRef coef0 = alloc_floats(2 * cin, cin), c2_0;
        if (spade) {
            m.ops.push_back(gn_op(idx, x, 1e-6f, 0, 0, -1, -1, coef0));                 // layerspp.py:131 (eps 1e-6, no affine)
            c2_0 = alloc_floats(2 * cin, cin);
            m.ops.push_back(coef2_op(idx, e0, cin, c2_0));
        } else {
            m.ops.push_back(gn_op(idx, x, 1e-5f, 1, e0, -1, -1, coef0));
        }

        Act h1;
        h1.H = Ho;
        h1.a = alloc(cout, Ho);
        TRef shortcut_src0 = x.a, shortcut_src1 = x.b;
        std::string shortcut_bias;  // up blocks: Conv_2's bias joins Conv_1's (see below)
        TRef res = x.a;             // identity shortcut unless Conv_2 exists (in == out, no resample => single source)
        auto emit_shortcut = [&]() {
            if (!conv2) return;
            res = alloc(cout, Ho);
            Op c{};
            c.kind = OP_CONV; c.module = idx; c.src0 = shortcut_src0; c.src1 = shortcut_src1; c.H = c.W = Ho; c.dst = res;
            c.side = true;          // depends only on x: runs on the side stream, concurrently with Conv_0 / actnorm1
            conv_pack(c, {P + ".Conv_2.weight"}, {P + ".Conv_2.bias"}, cout, cin, 1, 0);
            m.ops.push_back(c);
        };
        if (up) {
            // Up block: Conv_2 is pointwise over channels, the FIR acts per channel over space -- the two commute EXACTLY
            // (zero boundary included): Conv_2(FIR_up(x)) = FIR_up(W2 x) + b2.  So the shortcut GEMM runs on the LOW-resolution x
            // (a quarter of the pixels the reference order multiplies, layerspp.py:600-601, :619-620), its result is upsampled, and
            // b2 -- which must not pass through the zero-padded FIR -- is added with Conv_1's bias in that conv's epilogue.
            TRef r_lo = alloc(cout, H);
            Op c2{};
            c2.kind = OP_CONV; c2.module = idx; c2.src0 = x.a; c2.H = c2.W = H; c2.dst = r_lo;
            conv_pack(c2, {P + ".Conv_2.weight"}, {P + ".Conv_2.bias"}, cout, cin, 1, 0);
            m.packs.back().zero_bias = true;
            m.ops.push_back(c2);
            res = alloc(cout, Ho);
            Op f2{};
            f2.kind = OP_FIR; f2.module = idx; f2.src0 = r_lo; f2.H = f2.W = H; f2.up = 1; f2.dst = res;
            m.ops.push_back(f2);
            TRef hA = alloc(cin, Ho);
            Op f{};
            f.kind = OP_FIR; f.module = idx; f.src0 = x.a; f.H = f.W = H; f.coef = coef0; f.act = 1; f.up = 1; f.dst = hA;
            if (spade) { f.gb = gb0; f.coef2 = c2_0; }
            m.ops.push_back(f);
            Op c{};
            c.kind = OP_CONV; c.module = idx; c.src0 = hA; c.H = c.W = Ho; c.dst = h1.a;
            conv_pack(c, {P + ".Conv_0.weight"}, {P + ".Conv_0.bias"}, cout, cin, 3, 0);
            m.ops.push_back(c);
            shortcut_bias = P + ".Conv_2.bias";
        } else if (down) {
            TRef hA = alloc(cin, Ho), xr = alloc(cin, Ho);
            Op f{};
            f.kind = OP_FIR; f.module = idx; f.src0 = x.a; f.H = f.W = H; f.coef = coef0; f.act = 1; f.up = 0; f.dst = hA;
            if (spade) { f.gb = gb0; f.coef2 = c2_0; }
            f.dst2 = xr;                 // one read of x produces both FIR(act(norm(x))) and FIR(x) (layerspp.py:600-601)
            m.ops.push_back(f);
            shortcut_src0 = xr;
            shortcut_src1 = TRef{};
            emit_shortcut();
            Op c{};
            c.kind = OP_CONV; c.module = idx; c.src0 = hA; c.H = c.W = Ho; c.dst = h1.a;
            conv_pack(c, {P + ".Conv_0.weight"}, {P + ".Conv_0.bias"}, cout, cin, 3, 0);
            m.ops.push_back(c);
        } else {
            emit_shortcut();
            Op c{};
            c.kind = OP_CONV; c.module = idx; c.H = c.W = H; c.dst = h1.a;
            c.src0 = x.a; c.src1 = x.b; c.coef = coef0; c.act = 1;
            if (spade) {                         // SPADE modulation in the conv loader (Winograd kernel); `tmp` serves the kernels
                c.gb = gb0; c.coef2 = c2_0;      // that cannot take it: spade_apply materialises the activated tensor there
                c.tmp = alloc(cin, H);
            }
            conv_pack(c, {P + ".Conv_0.weight"}, {P + ".Conv_0.bias"}, cout, cin, 3, 0);
            m.ops.push_back(c);
        }
        // ---- actnorm1
        TRef coef1 = alloc_floats(2 * cout, cout);
        TRef conv1_src = h1.a;
        TRef c2_1;
        if (spade) {
            m.ops.push_back(gn_op(idx, h1, 1e-6f, 0, 0, -1, -1, coef1));
            c2_1 = alloc_floats(2 * cout, cout);
            m.ops.push_back(coef2_op(idx, e1, cout, c2_1));
        } else {
            m.ops.push_back(gn_op(idx, h1, 1e-5f, 1, e1, -1, -1, coef1));
        }

        out->H = Ho;
        out->a = alloc(cout, Ho);
        out->b = TRef{};
        Op c{};
        c.kind = OP_CONV; c.module = idx; c.src0 = conv1_src; c.H = c.W = Ho; c.res = res; c.out_scale = rs2; c.dst = out->a;
        c.join = conv2;
        c.coef = coef1; c.act = 1;
        if (spade) { c.gb = gb1; c.coef2 = c2_1; c.tmp = alloc(cout, Ho); }
        conv_pack(c, {P + ".Conv_1.weight"}, {P + ".Conv_1.bias"}, cout, cout, 3, 0);
        m.packs.back().extra_bias = shortcut_bias;
        m.ops.push_back(c);
        return 0;
    }

    // layerspp.py:207-249 AttnBlockpp
    int attn_block(int idx, const Act& x, int ch, Act* out) {
        const std::string P = "unet.all_modules." + std::to_string(idx);
        MCVD_REQUIRE(x.b.kind == REF_NONE && x.C() == ch, "plan: attention block %d input", idx);
        const int gw = m.add_param(P + ".GroupNorm_0.weight", {ch});
        const int gb = m.add_param(P + ".GroupNorm_0.bias", {ch});
        for (int j = 0; j < 4; ++j) {
            m.add_param(P + ".NIN_" + std::to_string(j) + ".W", {ch, ch});
            m.add_param(P + ".NIN_" + std::to_string(j) + ".b", {ch});
        }
        int heads = 1;
        const int nhc = m.d.n_head_channels;
        if (nhc != -1 && ch >= nhc) {
            MCVD_REQUIRE(nhc > 0 && ch % nhc == 0, "attention: %d channels not divisible by n_head_channels %d", ch, nhc);
            heads = ch / nhc;
        }
        const int H = x.H;
        const float rs2 = 1.0f / (float)sqrt(2.0);
        TRef coef = alloc_floats(2 * ch, ch);
        m.ops.push_back(gn_op(idx, x, 1e-6f, 2, 0, m.params[gw].off, m.params[gb].off, coef));
        TRef qkv = alloc(3 * ch, H);
        TRef kv = alloc_floats((int64_t)3 * ch * H * H, 3 * ch);        // K and V as three-piece bf16 LDS images (6 bytes per element)
        Op c{};
        c.kind = OP_CONV; c.module = idx; c.src0 = x.a; c.H = c.W = H; c.coef = coef; c.act = 0; c.dst = qkv;
        c.kv = kv; c.heads = heads;
        c.attn_op = (int)m.ops.size() + 1;
        conv_pack(c, {P + ".NIN_0.W", P + ".NIN_1.W", P + ".NIN_2.W"}, {P + ".NIN_0.b", P + ".NIN_1.b", P + ".NIN_2.b"}, ch, ch, 1, 1);
        m.ops.push_back(c);
        TRef o = alloc(ch, H);
        Op a{};
        a.kind = OP_ATTN; a.module = idx; a.src0 = qkv; a.dst = o; a.H = a.W = H; a.heads = heads; a.Cout = ch; a.kv = kv;
        m.ops.push_back(a);
        out->H = H;
        out->a = alloc(ch, H);
        out->b = TRef{};
        Op p{};
        p.kind = OP_CONV; p.module = idx; p.src0 = o; p.H = p.W = H; p.res = x.a; p.out_scale = rs2; p.dst = out->a;
        conv_pack(p, {P + ".NIN_3.W"}, {P + ".NIN_3.b"}, ch, ch, 1, 1);
        m.ops.push_back(p);
        return 0;
    }
};

}  // namespace

int mcvd_model::build_plan() {
    const mcvd_unet_desc& c = d;
    MCVD_REQUIRE(c.n_levels >= 1 && c.n_levels <= MCVD_MAX_LEVELS, "desc: n_levels=%d", c.n_levels);
    MCVD_REQUIRE(c.image_size >= 8 && (c.image_size & (c.image_size - 1)) == 0, "desc: image_size=%d must be a power of two", c.image_size);
    MCVD_REQUIRE((c.image_size >> (c.n_levels - 1)) >= 8, "desc: lowest resolution %d < 8", c.image_size >> (c.n_levels - 1));
    MCVD_REQUIRE(c.ngf >= 8 && c.ngf % 4 == 0 && c.ngf <= 256, "desc: ngf=%d", c.ngf);
    MCVD_REQUIRE(c.channels > 0 && c.num_frames > 0 && c.num_frames_cond >= 0, "desc: frames/channels");
    MCVD_REQUIRE(!c.spade || (c.num_frames_cond > 0 && c.spade_dim > 0), "desc: spade needs conditioning frames and spade_dim");
    MCVD_REQUIRE(c.num_classes >= 2, "desc: num_classes=%d", c.num_classes);
    MCVD_REQUIRE(!c.cond_emb || c.ngf % 2 == 0, "desc: cond_emb needs an even ngf");
    const int nf = c.ngf, C = c.channels, L = c.n_levels, S = c.image_size;
    T = 4 * nf + (c.cond_emb ? nf / 2 : 0);          // temb_dim (ncsnpp_more.py:95-99)
    NE = 0;
    Builder bld(*this);

    // modules 0,1: time MLP (ncsnpp_more.py:88-95)
    add_param("unet.all_modules.0.weight", {4 * nf, nf});
    add_param("unet.all_modules.0.bias", {4 * nf});
    add_param("unet.all_modules.1.weight", {4 * nf, 4 * nf});
    add_param("unet.all_modules.1.bias", {4 * nf});
    if (c.cond_emb) add_param("unet.all_modules.2.weight", {2, nf / 2});      // nn.Embedding(2, nf // 2), ncsnpp_more.py:98
    first_module = c.cond_emb ? 3 : 2;
    Op temb{};
    temb.kind = OP_TEMB; temb.module = 1;      // output = SiLU(module 1's output), what every Dense_0 consumes
    ops.push_back(temb);
    Op dense_op{};
    dense_op.kind = OP_DENSE; dense_op.module = -1;
    ops.push_back(dense_op);
    // conditioning frames as the network sees them: the caller's tensor, or (noise_in_cond) its diffused copy made per forward
    const int cond_ch_all = C * c.num_frames_cond;
    cond_src = TRef{REF_COND, 0, cond_ch_all};
    if (c.noise_in_cond && c.num_frames_cond > 0) {
        cond_src = bld.alloc(cond_ch_all, S);
        Op cn{};
        cn.kind = OP_CONDNOISE; cn.module = -1; cn.src0 = TRef{REF_COND, 0, cond_ch_all}; cn.dst = cond_src; cn.H = cn.W = S;
        ops.push_back(cn);
    }

    auto in_attn = [&](int res) {
        for (int i = 0; i < c.n_attn; ++i)
            if (c.attn_resolutions[i] == res) return true;
        return false;
    };

    // stem conv over [x, cond] (ncsnpp_more.py:188, :257)
    int idx = first_module;
    const int cx = C * c.num_frames, cc = c.spade ? 0 : C * c.num_frames_cond;
    const std::string stem = "unet.all_modules." + std::to_string(idx);
    add_param(stem + ".weight", {nf, cx + cc, 3, 3});
    add_param(stem + ".bias", {nf});
    std::vector<Act> hs;
    {
        Act h;
        h.H = S;
        h.a = bld.alloc(nf, S);
        Op cv{};
        cv.kind = OP_CONV; cv.module = idx; cv.H = cv.W = S; cv.dst = h.a;
        cv.src0 = TRef{REF_X, 0, cx};
        if (cc > 0) cv.src1 = cond_src;
        bld.conv_pack(cv, {stem + ".weight"}, {stem + ".bias"}, nf, cx + cc, 3, 0);
        ops.push_back(cv);
        hs.push_back(h);
    }
    ++idx;
    int in_ch = nf;
    int res = S;
    for (int lv = 0; lv < L; ++lv) {
        for (int rb = 0; rb < c.num_res_blocks; ++rb) {
            const int out_ch = nf * c.ch_mult[lv];
            Act h;
            if (bld.res_block(idx++, hs.back(), in_ch, out_ch, false, false, &h)) return -1;
            in_ch = out_ch;
            if (in_attn(res)) {
                Act h2;
                if (bld.attn_block(idx++, h, in_ch, &h2)) return -1;
                h = h2;
            }
            hs.push_back(h);
        }
        if (lv != L - 1) {
            Act h;
            if (bld.res_block(idx++, hs.back(), in_ch, in_ch, false, true, &h)) return -1;
            hs.push_back(h);
            res /= 2;
        }
    }
    Act h = hs.back();
    {
        Act t;
        if (bld.res_block(idx++, h, in_ch, in_ch, false, false, &t)) return -1;
        h = t;
        if (bld.attn_block(idx++, h, in_ch, &t)) return -1;
        h = t;
        if (bld.res_block(idx++, h, in_ch, in_ch, false, false, &t)) return -1;
        h = t;
    }
    for (int lv = L - 1; lv >= 0; --lv) {
        for (int rb = 0; rb < c.num_res_blocks + 1; ++rb) {
            const int out_ch = nf * c.ch_mult[lv];
            Act skip = hs.back();
            hs.pop_back();
            MCVD_REQUIRE(skip.H == h.H, "plan: skip resolution mismatch at module %d", idx);
            Act cat;
            cat.H = h.H;
            cat.a = h.a;
            cat.b = skip.a;           // torch.cat([h, hs.pop()], dim=1)  ncsnpp_more.py:356-357
            Act t;
            if (bld.res_block(idx++, cat, in_ch + skip.a.C, out_ch, false, false, &t)) return -1;
            h = t;
            in_ch = out_ch;
        }
        if (in_attn(h.H)) {
            Act t;
            if (bld.attn_block(idx++, h, in_ch, &t)) return -1;
            h = t;
        }
        if (lv != 0) {
            Act t;
            if (bld.res_block(idx++, h, in_ch, in_ch, true, false, &t)) return -1;
            h = t;
        }
    }
    MCVD_REQUIRE(hs.empty(), "plan: skip stack not empty");
    // final norm + SiLU, conv3x3 (ncsnpp_more.py:246-247 GroupNorm affine / :584-586 SPADE without temb)
    {
        const std::string P = "unet.all_modules." + std::to_string(idx);
        TRef coef = bld.alloc_floats(2 * in_ch, in_ch);
        Op cv{};
        cv.kind = OP_CONV; cv.H = cv.W = h.H;
        if (c.spade) {
            TRef gb = bld.spade_prep(idx, P, in_ch, h.H);
            ops.push_back(bld.gn_op(idx, h, 1e-6f, 0, 0, -1, -1, coef));
            cv.src0 = h.a; cv.coef = coef; cv.act = 1; cv.gb = gb;          // no temb pair here (ncsnpp_more.py:584-585)
            cv.tmp = bld.alloc(in_ch, h.H);
        } else {
            const int gw = add_param(P + ".Norm_0.weight", {in_ch});
            const int gb = add_param(P + ".Norm_0.bias", {in_ch});
            ops.push_back(bld.gn_op(idx, h, 1e-5f, 2, 0, params[gw].off, params[gb].off, coef));
            cv.src0 = h.a; cv.coef = coef; cv.act = 1;
        }
        ++idx;
        const std::string Q = "unet.all_modules." + std::to_string(idx);
        add_param(Q + ".weight", {cx, in_ch, 3, 3});
        add_param(Q + ".bias", {cx});
        cv.module = idx;
        cv.dst = TRef{REF_OUT, 0, cx};
        bld.conv_pack(cv, {Q + ".weight"}, {Q + ".bias"}, cx, in_ch, 3, 0);
        ops.push_back(cv);
        ++idx;
    }
    // time-embedding buffers and the fused Dense_0 matrix
    ops[0].dst = bld.alloc_floats(T, T);                 // silu(temb)   [B][T]
    ops[1].src0 = ops[0].dst;
    ops[1].dst = bld.alloc_floats(NE, NE);               // all (scale, shift) pairs [B][NE]
    dense_wt = bld.alloc_packed((int64_t)T * NE);
    dense_bias = bld.alloc_packed(NE);
    freqs_off = bld.alloc_packed(nf / 2);
    arena_per_sample = bld.arena;
    packed_floats = bld.packed;
    packed_h_floats = bld.packed_h;
    // GroupNorm statistics come out of the producing conv's epilogue: find, for every norm, the conv that wrote each source
    for (size_t gi = 0; gi < ops.size(); ++gi) {
        if (ops[gi].kind != OP_GN) continue;
        for (int side = 0; side < 2; ++side) {
            const TRef& src = side ? ops[gi].src1 : ops[gi].src0;
            if (src.kind != REF_ARENA) continue;
            for (size_t pi = 0; pi < gi; ++pi) {
                Op& p = ops[pi];
                if (p.kind == OP_CONV && p.dst.kind == REF_ARENA && p.dst.off == src.off && p.H * p.W % 32 == 0) {
                    if (p.stats.kind == REF_NONE) p.stats = bld.alloc_floats((int64_t)p.Cout * (p.H * p.W / 32) * 2, p.Cout);
                    (side ? ops[gi].prod1 : ops[gi].prod0) = (int)pi;
                }
            }
        }
    }
    // ... and, for every such conv, the single-source norm over its output that comes first (ConvArgs::gno: the conv's own last pass may
    // write that norm's table)
    for (size_t gi = ops.size(); gi-- > 0;) {
        const Op& g = ops[gi];
        if (g.kind == OP_GN && g.src1.kind == REF_NONE && g.prod0 >= 0 && g.coef.kind == REF_ARENA) ops[g.prod0].gn_next = (int)gi;
    }
    // consumers of GroupNorm coefficients: which norm of the plan wrote the table they read (the latest one before them)
    for (size_t ci = 0; ci < ops.size(); ++ci) {
        Op& c = ops[ci];
        if ((c.kind != OP_CONV && c.kind != OP_FIR && c.kind != OP_APPLY) || c.coef.kind != REF_ARENA) continue;
        for (size_t gi = 0; gi < ci; ++gi)
            if (ops[gi].kind == OP_GN && ops[gi].coef.kind == REF_ARENA && ops[gi].coef.off == c.coef.off) c.gn_src = (int)gi;
    }
    arena_per_sample = bld.arena;
    stats_np.assign(ops.size(), 0);
    if (c.noise_in_cond) {            // gamma/beta depend on the noised conditioning frames: nothing can be hoisted out of the step
        for (Op& op : ops) op.prep = false;
        has_prep = false;
    }
    return 0;
}

int mcvd_model::ensure_workspace(int B) {
    if (B <= arena_B) return 0;
    hipStream_t s = ctx->stream;
    MCVD_HIP_CHECK(hipStreamSynchronize(s));
    if (arena) MCVD_HIP_CHECK(hipFree(arena));
    if (labels) MCVD_HIP_CHECK(hipFree(labels));
    if (labels_f) MCVD_HIP_CHECK(hipFree(labels_f));
    labels_f = nullptr;
    if (eps_buf) MCVD_HIP_CHECK(hipFree(eps_buf));
    if (ksplit_buf) MCVD_HIP_CHECK(hipFree(ksplit_buf));
    if (cond_z) MCVD_HIP_CHECK(hipFree(cond_z));
    if (noise_buf) MCVD_HIP_CHECK(hipFree(noise_buf));
    arena = nullptr; labels = nullptr; eps_buf = nullptr; ksplit_buf = nullptr; cond_z = nullptr; noise_buf = nullptr; arena_B = 0;
    const size_t per = (size_t)d.channels * d.num_frames * d.image_size * d.image_size;
    MCVD_HIP_CHECK(hipMalloc((void**)&arena, (size_t)arena_per_sample * B * sizeof(float)));
    MCVD_HIP_CHECK(hipMalloc((void**)&labels, (size_t)B * sizeof(int64_t)));
    MCVD_HIP_CHECK(hipMalloc((void**)&labels_f, (size_t)B * sizeof(float)));
    MCVD_HIP_CHECK(hipMalloc((void**)&eps_buf, per * B * sizeof(float)));
    ksplit_floats = 0;                            // sized by ensure_ksplit (prepare_B) for the deepest K split this batch can select
    if (d.noise_in_cond && d.num_frames_cond > 0)
        MCVD_HIP_CHECK(hipMalloc((void**)&cond_z, (size_t)d.channels * d.num_frames_cond * d.image_size * d.image_size * B * sizeof(float)));
    if (d.gamma) MCVD_HIP_CHECK(hipMalloc((void**)&noise_buf, per * B * sizeof(float)));
    arena_B = B;
    cond_cache_valid = false;
    tuned_B = 0;
    ++epoch;
    return 0;
}

float* mcvd_model::resolve(const TRef& r, const float* x, const float* cond, float* out, int B) const {
    switch (r.kind) {
        case REF_ARENA: return arena + r.off * (int64_t)B;
        case REF_X: return const_cast<float*>(x);
        case REF_COND: return const_cast<float*>(cond);
        case REF_OUT: return out;
        default: return nullptr;
    }
}

// GroupNorm -> (A, B).  Statistics already computed by the producers' epilogues (every source must have them) are finalized by a
// small kernel (or were, by the producing conv's own K-split reduce pass: gn_done).  Without producer statistics: one pass over the tensor.
int mcvd_model::launch_gn(const Op& op, const float* x, const void* lab, const float* cond, float* out, int B) {
    hipStream_t s = op_stream ? op_stream : ctx->stream;
    const size_t gi = (size_t)(&op - ops.data());
    GnArgs a{};
    a.x0 = resolve(op.src0, x, cond, out, B);
    a.x1 = resolve(op.src1, x, cond, out, B);
    a.C0 = op.src0.C;
    a.C1 = op.src1.kind == REF_NONE ? 0 : op.src1.C;
    a.groups = op.groups;
    a.eps = op.eps;
    a.mode = op.gn_mode;
    if (op.gn_mode == 1) {
        a.p0 = resolve(ops[1].dst, x, cond, out, B);
        a.emb_stride = (uniform_labels && !d.cond_emb) ? 0 : NE;
        a.emb_off = op.emb_off;
    } else if (op.gn_mode == 2) {
        a.p0 = blob + op.p0;
        a.p1 = blob + op.p1;
    }
    a.coef = resolve(op.coef, x, cond, out, B);
    a.B = B;
    a.HW = op.H * op.W;
    const int np0 = op.prod0 >= 0 ? stats_np[op.prod0] : 0;
    const int np1 = a.C1 == 0 ? 1 : (op.prod1 >= 0 ? stats_np[op.prod1] : 0);
    if (gn_done.size() == ops.size() && gn_done[gi]) {       // the producing conv's K-split reduce pass wrote this table (ConvArgs::gno)
        gn_done[gi] = 0;
        return 0;
    }
    if (ctx->gn_stats && np0 > 0 && np1 > 0) {
        // timing-only ablation (option "dbg_skip_finalize", WRONG results after the first step): the captured graph carries no finalize
        // launch -- the table keeps the values of the eager first forward.  The upper bound of what ANY scheme that removes these launches
        // (producer-side last arriver, consumer-side reduction) can gain (profiles/r05_tail_launch_bound.txt).
        if (ctx->dbg_skip_finalize && op_stream) return 0;
        return launch_gn_finalize(a, resolve(ops[op.prod0].stats, x, cond, out, B), np0,
                                  a.C1 ? resolve(ops[op.prod1].stats, x, cond, out, B) : nullptr, np1, s);
    }
    return launch_gn_coef(a, s);
}

// ---- the GEMM forms of a 3x3 conv (kernels/conv_gemm_forms.cpp; shape ids 22 / 23).  `a` = the conv's arguments as launch_op built them.
bool mcvd_model::gemm_form_usable(const Op& op, const ConvArgs& a) const {
    if (!op.alt_kind || a.shape_hint != op.alt_kind || !ctx->bf16x3 || a.gb || a.ks != 3) return false;
    if (op.alt_kind == 22 && (a.stats || a.C1 != 0)) return false;        // (the shift-and-add pass emits no GroupNorm partials; one source)
    if (op.alt_kind == 23 && (a.coef || a.act)) return false;             // (im2col copies raw values)
    ConvArgs g = gemm_form_args(op, a, nullptr);
    if (op.alt_kind == 23 && !g.im2col && op.alt_buf.kind == REF_NONE) return false;      // (a wide stem whose geometry the LDS form does not take)
    for (int c = 4; c >= 1; --c)
        if (conv1x1_h2_supported(g, c, 3)) return true;
    return false;
}

// the 1x1 GEMM's arguments; buf = the op's per-call buffer (im2col rows / z planes)
ConvArgs mcvd_model::gemm_form_args(const Op& op, const ConvArgs& a, float* buf) const {
    ConvArgs g = a;
    g.ks = 1;
    g.wp = packed + op.alt_wp;
    g.wpb = packed + op.alt_wpb;
    g.wpw = nullptr; g.wph = nullptr;
    g.part = nullptr;
    g.shape_hint = 15;
    g.CinP = op.alt_CinP;
    g.CoutP = op.alt_CoutP;
    if (op.alt_kind == 22) {                 // Cin -> 9 * Cout planes of z; bias / residual / scale belong to the shift-and-add pass
        g.Cout = 9 * a.Cout;
        g.bias = packed + op.alt_bias;
        g.res = nullptr; g.out_scale = 1.0f; g.stats = nullptr;
        g.y = buf;
    } else {                                 // the im2col rows -> Cout; the epilogue is the conv's
        g.Cin = op.alt_CinP;
        g.im2col = 1;                        // staged in LDS by the GEMM itself (g.x0 / x1 / C0 / C1 stay the conv's raw sources) ...
        bool lds_form = ctx->im2col_lds != 0;
        if (lds_form) {
            lds_form = false;
            for (int c = 4; c >= 1 && !lds_form; --c) lds_form = conv1x1_h2_supported(g, c, 3);
        }
        if (!lds_form) {                     // ... or materialised in `buf` first, where that form's geometry does not apply (or option im2col_lds = 0)
            g.im2col = 0;
            g.x0 = buf; g.x1 = nullptr;
            g.C0 = op.alt_CinP; g.C1 = 0;
        }
    }
    int cot = a.cot;
    if (cot < 1 || cot > 4 || !conv1x1_h2_supported(g, cot, 3))
        for (cot = 4; cot > 1 && !conv1x1_h2_supported(g, cot, 3); --cot) {}
    g.cot = cot;
    return g;
}

int mcvd_model::launch_gemm_form(const Op& op, const ConvArgs& a, float* buf, hipStream_t s) {
    const ConvArgs g = gemm_form_args(op, a, buf);
    if (op.alt_kind == 23 && !g.im2col)
        if (int rc = launch_im2col3x3(a.x0, a.C0, a.x1, a.C1, buf, a.B, a.H, a.W, op.alt_CinP, s)) return rc;
    if (int rc = launch_conv_mfma(g, s)) return rc;
    if (op.alt_kind == 22) return launch_taps_shift_add(buf, a.bias, a.res, a.out_scale, a.y, a.B, a.Cout, a.H, a.W, s);
    return 0;
}

int mcvd_model::launch_op(const Op& op, const float* x, const void* lab, const float* cond, float* out, int B) {
    hipStream_t s = op_stream ? op_stream : ctx->stream;
    switch (op.kind) {
        case OP_TEMB: {
            if (temb_row_live) return 0;      // this forward's row of the call's table is already in ops[1].dst (use_temb_row)
            const float* w0 = blob + params[find_param("unet.all_modules.0.weight")].off;
            const float* b0 = blob + params[find_param("unet.all_modules.0.bias")].off;
            const float* w1 = blob + params[find_param("unet.all_modules.1.weight")].off;
            const float* b1 = blob + params[find_param("unet.all_modules.1.bias")].off;
            const float* emb = d.cond_emb ? blob + params[find_param("unet.all_modules.2.weight")].off : nullptr;
            // uniform labels (and no per-row mask embedding): one row serves the batch
            return launch_temb_mlp(lab, labels_f32, packed + freqs_off, w0, b0, w1, b1, resolve(op.dst, x, cond, out, B),
                                   (uniform_labels && !d.cond_emb) ? 1 : B, d.ngf, T, emb, cond_mask, s);
        }
        case OP_DENSE:
            if (temb_row_live) return 0;
            return launch_dense_all(resolve(op.src0, x, cond, out, B), packed + dense_wt, packed + dense_bias,
                                    resolve(op.dst, x, cond, out, B), (uniform_labels && !d.cond_emb) ? 1 : B, T, NE, s);
        case OP_GN:
            return launch_gn(op, x, lab, cond, out, B);
        case OP_CONV: {
            ConvArgs a{};
            a.x0 = resolve(op.src0, x, cond, out, B);
            a.x1 = resolve(op.src1, x, cond, out, B);
            a.C0 = op.src0.C;
            a.C1 = op.src1.kind == REF_NONE ? 0 : op.src1.C;
            MCVD_REQUIRE(a.x0 && (a.C1 == 0 || a.x1), "forward: module %d needs a conditioning tensor (cond is NULL)", op.module);
            a.coef = op.coef.kind == REF_NONE ? nullptr : resolve(op.coef, x, cond, out, B);
            a.act = op.act;
            a.wp = packed + op.wp;
            a.wpw = op.wpw >= 0 ? packed + op.wpw : nullptr;
            a.wph = (op.wph >= 0 && packed_h_valid) ? packed_h + op.wph : nullptr;      // present only while the option f16x2 is on
            a.wpb = op.wpb >= 0 ? packed + op.wpb : nullptr;
            a.bias = packed + op.bias;
            a.res = op.res.kind == REF_NONE ? nullptr : resolve(op.res, x, cond, out, B);
            a.out_scale = op.out_scale;
            a.y = resolve(op.dst, x, cond, out, B);
            a.B = B;
            a.Cin = a.C0 + a.C1;
            a.CinP = op.CinP;
            a.Cout = op.Cout;
            a.CoutP = op.CoutP;
            a.H = op.H;
            a.W = op.W;
            a.ks = op.ks;
            a.cot = op.cot;
            a.shape_hint = (op.ks == 1 && ctx->conv_shape1 >= 0) ? ctx->conv_shape1 : ctx->conv_shape;
            a.wdma = ctx->conv_wdma;
            a.pgrid = ctx->persist_grid;
            a.part = (op.ks == 3 && op.H * op.W <= 256) ? ksplit_buf : nullptr;
            a.part_floats = a.part ? ksplit_floats : 0;
            const size_t oi = (size_t)(&op - ops.data());
            if (a.shape_hint == 14 || a.shape_hint == 15) {      // forced split-operand 1x1 GEMM (tests): a cout tile that kernel serves
                const int np = a.shape_hint == 14 ? 2 : 3;
                for (int c = 4; c >= 1 && !conv1x1_h2_supported(a, a.cot, np); --c)
                    if (conv1x1_h2_supported(a, c, np)) a.cot = c;
            }
            if (a.shape_hint < 0 && tuned_B == B && oi < tuned_shape.size() && tuned_shape[oi] >= 0) {
                a.shape_hint = tuned_shape[oi];
                a.cot = tuned_cot[oi];
                // an imported table from a run under other options: the options win.  f16x2 hints (12 / 13 / 14) go to their bf16x3
                // counterparts (10 / 11 / 15) when f16x2 is off -- and for convs with a raw input in any case (f16x2 range guard, below);
                // bf16x3 hints go to the fp32-MFMA kernels (4 / 8 / 5) when bf16x3 is off
                const bool raw_in = op.coef.kind == REF_NONE;
                if ((!ctx->f16x2 || raw_in) && a.shape_hint >= 12 && a.shape_hint <= 14) {
                    a.shape_hint = a.shape_hint == 12 ? 10 : a.shape_hint == 13 ? 11 : 15;
                    if (a.shape_hint == 15 && !conv1x1_h2_supported(a, a.cot, 3)) a.cot = op.cot;
                }
                if (!ctx->bf16x3 && (a.shape_hint == 10 || a.shape_hint == 11 || (a.shape_hint >= 15 && a.shape_hint <= 20))) {
                    a.shape_hint = (a.shape_hint == 10 || a.shape_hint == 16) ? 4 : a.shape_hint == 15 ? 5 : 8;
                    a.cot = op.cot;
                }
            }
            // epilogue statistics from the 3x3 (Winograd) producers; the 1x1 GEMM's epilogue can emit them too, but its 16*COT
            // 32-lane reductions per wave cost the NIN_3 launches more than the norms they spare save (measured): "gn_stats" = 2 only
            if (op.gb.kind != REF_NONE) {
                // SPADE norm in front of this conv: fused into the Winograd loader where that kernel takes the launch, otherwise
                // spade_apply materialises silu(((A x + B)(1 + gamma) + beta) s1 + b2) and the conv reads it plainly
                a.gb = resolve(op.gb, x, cond, out, B);
                a.coef2 = op.coef2.kind == REF_NONE ? nullptr : resolve(op.coef2, x, cond, out, B);
                if (a.shape_hint == 36 || a.shape_hint == 40) a.shape_hint = -1;      // (ids of a round-5 table: the per-layer fused loader is gone)
                const bool want_fused = ctx->spade_fuse && !ctx->naive_conv && ctx->winograd;
                ConvArgs t = a;
                t.shape_hint = (a.shape_hint == 8) ? 8 : 4;
                t.ksplit = t.shape_hint == 8 ? 2 : 0;
                bool fused = want_fused && conv_wino_usable(t);
                if (!fused && t.ksplit == 2) { t.ksplit = 0; t.shape_hint = 4; fused = want_fused && conv_wino_usable(t); }
                if (fused) {
                    a.shape_hint = t.shape_hint;
                    ++fused_launches[2];
                } else {
                    float* tmp = resolve(op.tmp, x, cond, out, B);
                    if (int rc = launch_spade_apply(a.x0, a.C0, a.x1, a.C1, a.coef, a.gb, a.coef2, tmp, B, op.H * op.W, s)) return rc;
                    a.x0 = tmp; a.x1 = nullptr; a.C0 = a.Cin; a.C1 = 0; a.coef = nullptr; a.act = 0; a.gb = nullptr; a.coef2 = nullptr;
                }
            }
            // (the split-operand 1x1 GEMM emits them cheaply from its transposed epilogue: shape ids 14 / 15)
            a.stats = (ctx->gn_stats && op.stats.kind != REF_NONE && !ctx->naive_conv && (op.ks == 3 || ctx->gn_stats >= 2 || a.shape_hint == 14 || a.shape_hint == 15))
                          ? resolve(op.stats, x, cond, out, B) : nullptr;
            if (ran_kernel.size() != ops.size()) ran_kernel.assign(ops.size(), -1);
            if (kv_live.size() != ops.size()) kv_live.assign(ops.size(), 0);
            if (gn_done.size() != ops.size()) gn_done.assign(ops.size(), 0);
            if (op.gn_next >= 0) {
                // the norm over this conv's output: if the launch ends in a K-split reduce pass over 8 x 8 / 16 x 16 planes, that pass writes
                // the norm's (A, B) table too (gn.cpp: ksplit_reduce_gn_kernel) and the norm's own launch is skipped
                gn_done[op.gn_next] = 0;
                const Op& g = ops[op.gn_next];
                if (ctx->gn_producer && ctx->gn_stats && a.stats && op.ks == 3 && !ctx->dbg_skip_finalize) {
                    a.gno.coef = resolve(g.coef, x, cond, out, B);
                    a.gno.groups = g.groups;
                    a.gno.eps = g.eps;
                    a.gno.mode = g.gn_mode;
                    if (g.gn_mode == 1) {
                        a.gno.p0 = resolve(ops[1].dst, x, cond, out, B);
                        a.gno.emb_stride = (uniform_labels && !d.cond_emb) ? 0 : NE;
                        a.gno.emb_off = g.emb_off;
                    } else if (g.gn_mode == 2) {
                        a.gno.p0 = blob + g.p0;
                        a.gno.p1 = blob + g.p1;
                    }
                }
            }
            if (op.attn_op >= 0) {
                // the q|k|v projection of an attention block: where the launch goes to the three-piece 1x1 GEMM and the attention op will run
                // the three-piece kernel on a device of its own, K and V leave this kernel pre-split (conv1x1_h2.cpp KV -> attn_h2p_kernel)
                kv_live[op.attn_op] = 0;
                const Op& at = ops[op.attn_op];
                const bool want = ctx->attn_presplit && !ctx->naive_conv && ctx->bf16x3 && !ctx->f16x2 && (ctx->naive_attn == 0 || ctx->naive_attn == 4) &&
                                  !(ctx->share_fence && mcvd_ctx_shares_device(ctx)) && attn_h2p_supported(at.Cout, at.heads, at.H * at.W) && a.shape_hint == 15;
                if (want) {
                    ConvArgs t = a;
                    t.kv_img = resolve(op.kv, x, cond, out, B);
                    t.kv_C = at.Cout;
                    t.kv_D = at.Cout / at.heads;
                    if (conv1x1_h2_kv_supported(t, t.cot)) {
                        a = t;
                        kv_live[op.attn_op] = 1;
                    }
                }
            }
            if (ctx->naive_conv) {
                stats_np[oi] = 0;
                ran_kernel[oi] = -2;
                return launch_conv_naive(a, s);
            }
            if (a.shape_hint == 22 || a.shape_hint == 23) {
                if (gemm_form_usable(op, a)) {
                    const int rc = launch_gemm_form(op, a, resolve(op.alt_buf, x, cond, out, B), s);
                    stats_np[oi] = a.stats ? last_conv_stats_np() : 0;
                    ran_kernel[oi] = op.alt_kind;
                    return rc;
                }
                a.shape_hint = -1;               // (forced for the whole network, or a table of another build: the dispatcher's own choice)
                a.cot = op.cot;
            }
            const int rc = launch_conv_mfma(a, s);
            stats_np[oi] = a.stats ? last_conv_stats_np() : 0;
            ran_kernel[oi] = last_conv_kernel();
            if (rc == 0 && a.gno.coef && last_conv_gn_fused()) {
                gn_done[op.gn_next] = 1;
                ++fused_launches[3];
            }
            return rc;
        }
        case OP_FIR: {
            const float* gb = op.gb.kind == REF_NONE ? nullptr : resolve(op.gb, x, cond, out, B);
            return launch_fir2(resolve(op.src0, x, cond, out, B),
                               op.coef.kind == REF_NONE ? nullptr : resolve(op.coef, x, cond, out, B), op.act, op.up,
                               resolve(op.dst, x, cond, out, B), B, op.src0.C, op.H, op.W, gb,
                               gb ? gb + (size_t)op.src0.C * op.H * op.W : nullptr,
                               op.coef2.kind == REF_NONE ? nullptr : resolve(op.coef2, x, cond, out, B),
                               op.dst2.kind == REF_NONE ? nullptr : resolve(op.dst2, x, cond, out, B), s, ctx->fir_form);
        }
        case OP_NEAREST:
            MCVD_REQUIRE(cond, "forward: SPADE model needs the conditioning tensor");
            return launch_nearest_resize(resolve(op.src0, x, cond, out, B), resolve(op.dst, x, cond, out, B), B * op.src0.C,
                                         d.image_size, d.image_size, op.H, op.W, s);
        case OP_CONDNOISE: {
            // cond <- sqrt(a[t]) cond + sqrt(1 - a[t]) z, a fresh z per forward (ncsnpp_more.py:755-768)
            MCVD_REQUIRE(cond, "forward: noise_in_cond model needs the conditioning tensor");
            MCVD_REQUIRE(!labels_f32, "forward: noise_in_cond indexes alphas with the labels; float timesteps are not valid there");
            const int64_t per = (int64_t)op.src0.C * d.image_size * d.image_size;
            if (!alphas_dev) MCVD_HIP_CHECK(hipMalloc((void**)&alphas_dev, (size_t)d.num_classes * sizeof(float)));
            if (!alphas_dev_valid) {
                MCVD_HIP_CHECK(hipMemcpyAsync(alphas_dev, alphas.data(), (size_t)d.num_classes * sizeof(float), hipMemcpyHostToDevice, s));
                alphas_dev_valid = true;
            }
            const float* z = cond_noise_src;
            if (z) {
                cond_noise_src += per * B;                        // injected sequence: one slab per forward
            } else {
                if (cond_gamma_k > 0.0f) {
                    if (int rc = launch_gamma_noise(cond_z, nullptr, cond_gamma_k, cond_gamma_theta, cond_gamma_kt, cond_gamma_sd,
                                                    cond_noise_seed, cond_noise_offset, (1ull << 32) + cond_noise_draw, B, per, s))
                        return rc;
                } else {
                    MCVD_REQUIRE(!d.gamma, "forward: a gamma model draws its conditioning noise per row (k_cum[labels]); pass z through "
                                           "mcvd_model_set_cond_noise or use mcvd_sampler_run");
                    if (int rc = launch_randn(cond_z, cond_noise_seed, cond_noise_offset, (1ull << 32) + cond_noise_draw, B, per, s)) return rc;
                }
                ++cond_noise_draw;
                z = cond_z;
            }
            return launch_cond_noise(cond, z, alphas_dev, static_cast<const int64_t*>(lab), d.num_classes,
                                     resolve(op.dst, x, cond, out, B), B, per, s);
        }
        case OP_COEF2: {
            // the FIRST table op of the plan fills every table of the forward in one launch (they all read the fused Dense_0 row, which
            // is complete before any of them); the others are no-ops.  All tables are arena tensors.
            const int self = (int)(&op - ops.data());
            if (coef2_desc_dev && coef2_count > 1) {
                if (self != coef2_first) return 0;
                return launch_coef2_all(resolve(ops[1].dst, x, cond, out, B), (uniform_labels && !d.cond_emb) ? 0 : NE, coef2_desc_dev,
                                        coef2_count, coef2_cmax, arena, B, s);
            }
            return launch_coef2(resolve(ops[1].dst, x, cond, out, B), (uniform_labels && !d.cond_emb) ? 0 : NE, op.emb_off,
                                resolve(op.dst, x, cond, out, B), B, op.Cout, s);
        }
        case OP_APPLY:
            return launch_spade_apply(resolve(op.src0, x, cond, out, B), op.src0.C, resolve(op.src1, x, cond, out, B),
                                      op.src1.kind == REF_NONE ? 0 : op.src1.C, resolve(op.coef, x, cond, out, B),
                                      resolve(op.gb, x, cond, out, B),
                                      op.coef2.kind == REF_NONE ? nullptr : resolve(op.coef2, x, cond, out, B),
                                      resolve(op.dst, x, cond, out, B), B, op.H * op.W, s);
        case OP_ATTN:
        {
            // (option share_fence, off since round 6 found and removed the cause: a context that shares its device -- another process, or another
            // stream of this one -- keeps the split-operand attention kernels off it: api.cpp, mcvd_ctx_shares_device)
            const bool shared = ctx->share_fence && mcvd_ctx_shares_device(ctx);
            const size_t ai = (size_t)(&op - ops.data());
            if (ai < kv_live.size() && kv_live[ai]) {                     // K and V were written pre-split by the projection of this forward
                kv_live[ai] = 0;
                ++fused_launches[0];
                return launch_attention_h2p(resolve(op.src0, x, cond, out, B), resolve(op.kv, x, cond, out, B), resolve(op.dst, x, cond, out, B), B,
                                            op.Cout, op.heads, op.H * op.W, s);
            }
            return launch_attention(ctx->naive_attn, shared ? 0 : ctx->f16x2, shared ? 0 : ctx->bf16x3, resolve(op.src0, x, cond, out, B), resolve(op.dst, x, cond, out, B), B,
                                    op.Cout, op.heads, op.H * op.W, s);
        }
        default:
            set_error("forward: unknown op kind %d", (int)op.kind);
            return -1;
    }
}

// Measurement-driven tile selection: for every distinct conv layer shape, time the candidate (pixel tile, cout tile)
// pairs with HIP events on the real workspace buffers and keep the fastest.  ~0.2 s once per batch size.
int mcvd_model::autotune(int B) {
    hipStream_t s = ctx->stream;
    tuned_shape.assign(ops.size(), -1);
    tuned_cot.assign(ops.size(), 0);
    hipEvent_t e0, e1;
    MCVD_HIP_CHECK(hipEventCreate(&e0));
    MCVD_HIP_CHECK(hipEventCreate(&e1));
    // benign, finite workspace contents for the timing runs
    MCVD_HIP_CHECK(hipMemsetAsync(arena, 0x3c, (size_t)arena_per_sample * B * sizeof(float), s));
    struct Key { int ks, H, cin, cout, coef, res; bool operator<(const Key& o) const {
        return std::tie(ks, H, cin, cout, coef, res) < std::tie(o.ks, o.H, o.cin, o.cout, o.coef, o.res); } };
    std::map<Key, std::pair<int, int>> best;
    float* scratch_io = arena;      // stands in for the caller's x / cond / out during tuning
    for (size_t i = 0; i < ops.size(); ++i) {
        const Op& op = ops[i];
        if (op.kind != OP_CONV) continue;
        const int cin = op.src0.C + (op.src1.kind == REF_NONE ? 0 : op.src1.C);
        Key k{op.ks, op.H, cin, op.Cout, (op.coef.kind != REF_NONE ? 1 : 0) + (op.gb.kind != REF_NONE ? 2 : 0) + (op.coef2.kind != REF_NONE ? 4 : 0),
              op.res.kind != REF_NONE};
        auto it = best.find(k);
        if (it == best.end()) {
            ConvArgs a{};
            a.x0 = resolve(op.src0, scratch_io, scratch_io, scratch_io, B);
            a.x1 = resolve(op.src1, scratch_io, scratch_io, scratch_io, B);
            a.C0 = op.src0.C;
            a.C1 = op.src1.kind == REF_NONE ? 0 : op.src1.C;
            a.coef = op.coef.kind == REF_NONE ? nullptr : resolve(op.coef, scratch_io, scratch_io, scratch_io, B);
            a.act = op.act;
            a.wp = packed + op.wp;
            a.wpw = op.wpw >= 0 ? packed + op.wpw : nullptr;
            a.wph = (op.wph >= 0 && packed_h_valid) ? packed_h + op.wph : nullptr;      // present only while the option f16x2 is on
            a.wpb = op.wpb >= 0 ? packed + op.wpb : nullptr;
            a.bias = packed + op.bias;
            a.res = op.res.kind == REF_NONE ? nullptr : resolve(op.res, scratch_io, scratch_io, scratch_io, B);
            a.out_scale = op.out_scale;
            a.y = resolve(op.dst, scratch_io, scratch_io, scratch_io, B);
            a.B = B; a.Cin = cin; a.CinP = op.CinP; a.Cout = op.Cout; a.CoutP = op.CoutP; a.H = op.H; a.W = op.W; a.ks = op.ks;
            a.wdma = ctx->conv_wdma;
            a.pgrid = ctx->persist_grid;
            a.part = (op.ks == 3 && op.H * op.W <= 256) ? ksplit_buf : nullptr;
            a.part_floats = a.part ? ksplit_floats : 0;
            const bool spade_fused = op.gb.kind != REF_NONE && ctx->spade_fuse && ctx->winograd;
            if (spade_fused) {
                a.gb = resolve(op.gb, scratch_io, scratch_io, scratch_io, B);
                a.coef2 = op.coef2.kind == REF_NONE ? nullptr : resolve(op.coef2, scratch_io, scratch_io, scratch_io, B);
            } else if (op.gb.kind != REF_NONE) {           // unfused: the conv sees the materialised tensor
                a.x0 = resolve(op.tmp, scratch_io, scratch_io, scratch_io, B);
                a.x1 = nullptr; a.C0 = cin; a.C1 = 0; a.coef = nullptr; a.act = 0;
            }
            float best_ms = 1e30f;
            std::pair<int, int> choice{-1, op.cot};
            auto time_candidate = [&](int shape, int cot) -> int {
                a.cot = cot;
                a.shape_hint = shape;
                if (launch_conv_mfma(a, s)) return 0;                     // warm-up (and validity check)
                MCVD_HIP_CHECK(hipEventRecord(e0, s));
                for (int r = 0; r < 3; ++r)
                    if (int rc = launch_conv_mfma(a, s)) return rc;
                MCVD_HIP_CHECK(hipEventRecord(e1, s));
                MCVD_HIP_CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                MCVD_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best_ms) { best_ms = ms; choice = {shape, cot}; }
                return 0;
            };
            const int cots[2] = {op.cot, 1};
            for (int ci = 0; ci < (op.cot == 1 ? 1 : 2); ++ci) {
                for (int shape = 0; shape < 5; ++shape) {
                    if (spade_fused && conv_wino_usable(a) && shape != 4) continue;   // SPADE prologue: the Winograd kernel only
                    if (shape == 3 && (op.ks != 3 || !ctx->conv_wdma)) continue;      // 3 = split-K with double-buffered weights
                    if (shape == 4 && (ci > 0 || !ctx->winograd || !conv_wino_usable(a))) continue;   // 4 = Winograd F(2x2,3x3), own cout tile
                    const int bpx = shape == 0 ? 256 : shape == 1 ? 128 : 64;
                    const bool fits = bpx % op.W == 0 && (bpx / op.W <= op.H ? op.H % (bpx / op.W) == 0 : (bpx / op.W) % op.H == 0);
                    if (shape != 4 && !fits) continue;
                    if (int rc = time_candidate(shape, cots[ci])) return rc;
                }
            }
            a.cot = op.cot;
            if (op.alt_kind && !spade_fused && op.gb.kind == REF_NONE) {         // 22 / 23 = the conv as a 1x1 GEMM on the three-piece kernel + its copy / shift pass
                ConvArgs t = a;
                t.shape_hint = op.alt_kind;
                t.stats = (ctx->gn_stats && op.stats.kind != REF_NONE) ? resolve(op.stats, scratch_io, scratch_io, scratch_io, B) : nullptr;
                float* buf = resolve(op.alt_buf, scratch_io, scratch_io, scratch_io, B);
                for (int c = 4; c >= 1; --c) {
                    t.cot = c;
                    if (!gemm_form_usable(op, t)) break;
                    if (gemm_form_args(op, t, buf).cot != c) continue;
                    if (launch_gemm_form(op, t, buf, s)) continue;            // warm-up (and validity check)
                    MCVD_HIP_CHECK(hipEventRecord(e0, s));
                    for (int r = 0; r < 3; ++r)
                        if (int rc = launch_gemm_form(op, t, buf, s)) return rc;
                    MCVD_HIP_CHECK(hipEventRecord(e1, s));
                    MCVD_HIP_CHECK(hipEventSynchronize(e1));
                    float ms = 0.f;
                    MCVD_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best_ms) { best_ms = ms; choice = {op.alt_kind, c}; }
                }
            }
            if (op.ks == 3 && ctx->winograd) {             // 8 = Winograd with a 2-way K split (more workgroups: 8x8 layers)
                ConvArgs b = a;
                b.ksplit = 2;
                if (conv_wino_usable(b))
                    if (int rc = time_candidate(8, op.cot)) return rc;
            }
            if (op.ks == 3 && ctx->winograd && ctx->bf16x3 && !spade_fused) {     // 10 / 11 = Winograd on the bf16 pipe, three exact pieces per operand
                ConvArgs b = a;
                b.ksplit = 0;
                if (conv_wino3_usable(b))
                    if (int rc = time_candidate(10, op.cot)) return rc;
                b.ksplit = 2;
                if (conv_wino3_usable(b))
                    if (int rc = time_candidate(11, op.cot)) return rc;
                // 16 / 17 = the same kernel as persistent workgroups (conv_wino3p.cpp): the staging pipeline runs on across the items of a CU
                b.ksplit = 0;
                if (conv_wino3p_usable(b))
                    if (int rc = time_candidate(16, op.cot)) return rc;
                b.ksplit = 2;
                if (conv_wino3p_usable(b))
                    if (int rc = time_candidate(17, op.cot)) return rc;
                // 18 / 19 / 20 = 4 / 8 K parts (20: persistent, 4 parts), offered where the 2-way split leaves the chip with less than
                // a few workgroups per CU (the measurement decides; above that the reduce pass only costs)
                const bool g8 = op.H == 8 && op.W == 8;
                const long pairs = (g8 ? (B + 1) / 2 : (long)B * (op.H / 8) * (op.W / 16)) * (op.CoutP / (32 * conv_wino_cout_tile(op.Cout)));
                if (2 * pairs < 1024) {
                    b.ksplit = 4;
                    if (conv_wino3_usable(b))
                        if (int rc = time_candidate(18, op.cot)) return rc;
                    if (conv_wino3p_usable(b))
                        if (int rc = time_candidate(20, op.cot)) return rc;
                    b.ksplit = 8;
                    if (4 * pairs < 1024 && conv_wino3_usable(b))
                        if (int rc = time_candidate(19, op.cot)) return rc;
                }
            }
            // f16x2 range guard: the two-piece fp16 kernels see GroupNorm-ed inputs only (a.coef set: normalised, O(1) by construction).
            // A conv over a RAW tensor (stem, shortcuts, NIN_3) has no bound on its input and stays on the fp32-range kernels.
            const bool f16_ok = ctx->f16x2 && a.coef != nullptr;
            if (op.ks == 3 && ctx->winograd && f16_ok && !spade_fused) {          // 12 / 13 = Winograd on the fp16 pipe, two-piece operands
                ConvArgs b = a;
                b.ksplit = 0;
                if (conv_wino2h_usable(b))
                    if (int rc = time_candidate(12, op.cot)) return rc;
                b.ksplit = 2;
                if (conv_wino2h_usable(b))
                    if (int rc = time_candidate(13, op.cot)) return rc;
            }
            if (op.ks == 1 && ctx->conv_dma1) {        // 5 / 6 = all-DMA 1x1 GEMM (16 / 32 channels per chunk), cout tiles of its own
                // small cout tiles first: the sweep of every 1x1 layer shape (profiles/r02_conv1x1_candidates.txt) has tiles 1-3 winning
                // everywhere; 6 / 9 never did
                static const int g1_cots[] = {2, 1, 3, 4, 6, 9};
                for (int ck = 16; ck <= 32; ck += 16) {
                    if (!conv1x1_dma_supported(a, ck)) continue;
                    int tried = 0;
                    for (int c : g1_cots) {
                        if ((op.CoutP / 32) % c != 0 || tried >= 3 || (ck == 32 && c == 9)) continue;
                        ++tried;
                        if (int rc = time_candidate(ck == 16 ? 5 : 6, c)) return rc;
                    }
                }
                if (f16_ok) {                               // 14 = the GEMM on the fp16 pipe with two-piece operands, cout tiles 1..4
                    for (int c = 4; c >= 1; --c)
                        if (conv1x1_h2_supported(a, c, 2))
                            if (int rc = time_candidate(14, c)) return rc;
                }
                if (ctx->bf16x3) {                          // 15 = the GEMM on the bf16 pipe with three exact pieces per operand
                    for (int c = 4; c >= 1; --c)
                        if (conv1x1_h2_supported(a, c, 3))
                            if (int rc = time_candidate(15, c)) return rc;
                }
                if (conv1x1_dma_supported(a, 16, 2)) {      // 9 = the same GEMM with 64 pixels per wave (256-pixel tiles), cout tiles 1 / 2
                    if (int rc = time_candidate(9, 1)) return rc;
                    if ((op.CoutP / 32) % 2 == 0)
                        if (int rc = time_candidate(9, 2)) return rc;
                }
            }
            it = best.emplace(k, choice).first;
        }
        tuned_shape[i] = it->second.first;
        tuned_cot[i] = it->second.second;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    tuned_B = B;
    ++epoch;
    return 0;
}

// The options that decide which kernels the autotuner may offer: tables tuned (or imported) under other settings are dropped --
// "f16x2" = 0 must really mean that no two-piece fp16 kernel runs.  Called before a table is looked up and before one is imported
// (mcvd_model_set_tuning stamps its table with the options in force at the import).
void mcvd_model::sync_tuning_options() {
    const int sig = (ctx->winograd ? 1 : 0) | (ctx->conv_dma1 ? 2 : 0) | (ctx->bf16x3 ? 4 : 0) | (ctx->f16x2 ? 8 : 0) |
                    (ctx->spade_fuse ? 16 : 0) | (ctx->conv_wdma ? 32 : 0);
    if (sig != tuned_sig) {
        if (tuned_sig >= 0) { tuned_cache.clear(); tuned_B = 0; }
        tuned_sig = sig;
    }
}

// The two-piece fp16 forms of the weights (1.3 x the fp32 bytes for config 2) are packed when the option f16x2 first needs them, not at
// every finalize: default-arithmetic models never allocate them.
int mcvd_model::ensure_f16x2_weights() {
    if (packed_h_valid || packed_h_floats == 0) return 0;
    hipStream_t s = ctx->stream;
    if (!packed_h) MCVD_HIP_CHECK(hipMalloc((void**)&packed_h, (size_t)packed_h_floats * sizeof(float)));
    MCVD_HIP_CHECK(hipMemsetAsync(packed_h, 0, (size_t)packed_h_floats * sizeof(float), s));
    for (const ConvPack& p : packs) {
        if (p.wph < 0) continue;
        if (p.ks == 3) {
            const ParamInfo& w = params[find_param(p.weights[0].c_str())];             // Winograd layers have one weight tensor
            if (int rc = launch_pack_wino2h_weight(blob + w.off, packed_h + p.wph, p.Cout_each, p.Cin, p.CinP, p.CoutP, s)) return rc;
        } else {
            if (int rc = launch_pack_conv1x1_h2(packed + p.wp, packed_h + p.wph, p.CinP, p.CoutP, s, 2)) return rc;
        }
    }
    MCVD_HIP_CHECK(hipStreamSynchronize(s));
    packed_h_valid = true;
    ++epoch;
    return 0;
}

// Partial-output buffer of the K-split Winograd layers (H*W <= 256): `parts` x B x Cout x HW floats for the DEEPEST split anything can
// select at this batch (it was 8 parts for every model and batch, ~200 MB for config 2 at B = 64 where no layer ever takes more
// than 2): 2 parts everywhere; 4 / 8 where the autotuner's own rule offers shape ids 18 / 20 / 19 (few (region, cout tile) pairs:
// autotune()), where a test forces one of them (conv_shape), or where an installed table for this batch names one.
int mcvd_model::ensure_ksplit(int B) {
    size_t need = 0;
    const std::vector<int>* table = nullptr;
    auto it = tuned_cache.find(B);
    if (it != tuned_cache.end() && it->second.first.size() == ops.size()) table = &it->second.first;
    for (size_t i = 0; i < ops.size(); ++i) {
        const Op& op = ops[i];
        if (!(op.kind == OP_CONV && op.ks == 3 && op.wpw >= 0 && op.H * op.W <= 256)) continue;
        int parts = 2;
        const bool g8 = op.H == 8 && op.W == 8;
        const long pairs = (g8 ? (B + 1) / 2 : (long)B * (op.H / 8) * (op.W / 16)) * (op.CoutP / (32 * conv_wino_cout_tile(op.Cout)));
        if (ctx->autotune && !table && tuned_B != B && 2 * pairs < 1024) parts = 4 * pairs < 1024 ? 8 : 4;    // the candidates autotune() is about to time
        auto deepen = [&](int shape) { if (shape == 19) parts = std::max(parts, 8); else if (shape == 18 || shape == 20) parts = std::max(parts, 4); };
        deepen(ctx->conv_shape);
        if (table) deepen((*table)[i]);
        if (tuned_B == B && i < tuned_shape.size()) deepen(tuned_shape[i]);
        need = std::max(need, (size_t)parts * B * op.Cout * op.H * op.W);
    }
    if (need <= ksplit_floats) return 0;
    MCVD_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ksplit_buf) MCVD_HIP_CHECK(hipFree(ksplit_buf));
    ksplit_buf = nullptr;
    ksplit_floats = 0;
    MCVD_HIP_CHECK(hipMalloc((void**)&ksplit_buf, need * sizeof(float)));
    ksplit_floats = need;
    ++epoch;                                      // captured graphs embed the pointer
    return 0;
}

int mcvd_model::prepare_B(int B) {
    if (int rc = ensure_workspace(B)) return rc;
    // BEFORE the K-split buffer is sized: a table tuned (or imported) under other options is dropped here, and ensure_ksplit must see
    // the cache as the lookup below will -- sized from a stale shallow table it would be too small for the 4 / 8-part candidates
    // autotune() then times (the launches carry the capacity as well, ConvArgs::part_floats)
    sync_tuning_options();
    if (int rc = ensure_ksplit(B)) return rc;
    // (a forced two-piece kernel family -- the tests' conv_shape 12 / 13 / 14 -- needs the pieces as well as the option does)
    const bool forced_h = (ctx->conv_shape >= 12 && ctx->conv_shape <= 14) || ctx->conv_shape1 == 14;
    if (ctx->f16x2 || forced_h)
        if (int rc = ensure_f16x2_weights()) return rc;
    if (!ctx->naive_conv && tuned_B != B) {
        auto it = tuned_cache.find(B);
        if (it != tuned_cache.end() && it->second.first.size() == ops.size()) {     // tuned (or imported) before: no timing launches
            tuned_shape = it->second.first;
            tuned_cot = it->second.second;
            tuned_B = B;
            ++epoch;
            return 0;
        }
        if (!ctx->autotune) return 0;             // no table for this batch size and no permission to measure: the dispatcher's heuristics
        if (int rc = autotune(B)) return rc;
        tuned_cache[B] = {tuned_shape, tuned_cot};
        cond_cache_valid = false;                 // the tuner scribbles over the workspace
    }
    return 0;
}

int mcvd_model::run_prep(const float* cond, int B) {
    for (const Op& op : ops)
        if (op.prep)
            if (int rc = launch_op(op, nullptr, nullptr, cond, nullptr, B)) return rc;
    return 0;
}

int mcvd_model::prepare_cond(const float* cond, int B) {
    MCVD_REQUIRE(finalized, "prepare_cond before mcvd_model_finalize");
    if (!has_prep) return 0;
    MCVD_REQUIRE(cond && B > 0, "prepare_cond: bad arguments");
    if (int rc = prepare_B(B)) return rc;
    if (int rc = run_prep(cond, B)) return rc;
    prepared_cond = cond;
    prepared_B = B;
    cond_cache_valid = true;
    return 0;
}

// f16x2 range guard, second half (the first: convs over raw tensors never take the two-piece fp16 kernels, autotune()): the forward's
// epsilon is scanned for Inf / NaN; mcvd_ctx_check_range turns a hit into MCVD_ERANGE.
int mcvd_model::forward(const float* x, const void* lab, const float* cond, float* out, int B) {
    if (int rc = forward_unchecked(x, lab, cond, out, B)) return rc;
    if (!ctx->f16x2) return 0;
    if (!ctx->range_flag) {
        MCVD_HIP_CHECK(hipMalloc((void**)&ctx->range_flag, sizeof(int)));
        MCVD_HIP_CHECK(hipMemsetAsync(ctx->range_flag, 0, sizeof(int), ctx->stream));
    }
    return launch_nonfinite_flag(out, (int64_t)B * d.channels * d.num_frames * d.image_size * d.image_size, ctx->range_flag, ctx->stream);
}

int mcvd_model::forward_unchecked(const float* x, const void* lab, const float* cond, float* out, int B) {
    MCVD_REQUIRE(finalized, "forward before mcvd_model_finalize");
    MCVD_REQUIRE(B > 0 && x && lab && out, "forward: bad arguments");
    if (int rc = prepare_B(B)) return rc;
    // SPADE gamma/beta: reuse the cache only for the exact (cond pointer, batch) it was prepared for
    if (has_prep && !(cond_cache_valid && prepared_cond == cond && prepared_B == B))
        if (int rc = run_prep(cond, B)) return rc;
    if (ctx->profile && profile_armed) {
        profile_armed = false;
        if (ev.size() != 2 * ops.size()) {
            for (hipEvent_t e : ev) (void)hipEventDestroy(e);
            ev.assign(2 * ops.size(), nullptr);
            for (auto& e : ev) MCVD_HIP_CHECK(hipEventCreate(&e));
        }
        profile_B = B;
        profile_temb_skipped = temb_row_live;
        for (size_t i = 0; i < ops.size(); ++i) {      // instrumented forward: everything on the main stream
            if (ops[i].prep) continue;
            MCVD_HIP_CHECK(hipEventRecord(ev[2 * i], ctx->stream));
            if (int rc = launch_op(ops[i], x, lab, cond, out, B)) return rc;
            MCVD_HIP_CHECK(hipEventRecord(ev[2 * i + 1], ctx->stream));
        }
        return 0;
    }
    if (ctx->graph && !d.noise_in_cond) {          // noise_in_cond: every forward has its own noise slab / draw index
        // Replay policy: a pointer set is run eagerly the first time it is seen (kernel attributes get set, nothing unusual
        // happens inside a capture), captured + instantiated the second time, replayed from then on.  The sampler loop
        // presents the same (x, labels, eps, cond, B) for every step of a call.
        GraphKey k;
        k.x = x; k.lab = lab; k.cond = cond; k.out = out; k.B = B; k.labels_f32 = labels_f32 | (uniform_labels << 1) | (temb_row_live << 2); k.mask = cond_mask; k.epoch = epoch; k.ctx_epoch = ctx->epoch;
        if (graph_exec && k == graph_key) {
            MCVD_HIP_CHECK(hipGraphLaunch(graph_exec, ctx->stream));
            ++graph_replays;
            return 0;
        }
        if (k == graph_seen) {
            drop_graph();
            if (!ctx->cap) MCVD_HIP_CHECK(hipStreamCreateWithFlags(&ctx->cap, hipStreamNonBlocking));
            MCVD_HIP_CHECK(hipStreamBeginCapture(ctx->cap, hipStreamCaptureModeThreadLocal));
            op_stream = ctx->cap;
            int rc = 0;
            for (const Op& op : ops) {
                if (op.prep) continue;
                if ((rc = launch_op(op, x, lab, cond, out, B))) break;
            }
            op_stream = nullptr;
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(ctx->cap, &g);
            if (rc) {
                if (g) (void)hipGraphDestroy(g);
                return rc;
            }
            if (e != hipSuccess) {
                set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
                return -2;
            }
            const hipError_t ei = hipGraphInstantiate(&graph_exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ei != hipSuccess) {
                graph_exec = nullptr;
                set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ei));
                return -2;
            }
            graph_key = k;
            ++graph_captures;
            MCVD_HIP_CHECK(hipGraphLaunch(graph_exec, ctx->stream));
            ++graph_replays;
            return 0;
        }
        graph_seen = k;
    }
    return forward_ops(x, lab, cond, out, B);
}

// silu(temb) and all Dense_0 projections for every label of a sampler call: the forward's own two kernels over `rows` rows (rows are independent;
// dense_all's summation order does not depend on the row count: temb.cpp), into a table the forwards copy their row from.
int mcvd_model::prepare_temb_table(const std::vector<int>& labels) {
    hipStream_t s = ctx->stream;
    const size_t rows = labels.size();
    if (rows > temb_rows_cap) {
        MCVD_HIP_CHECK(hipStreamSynchronize(s));
        if (temb_tab) (void)hipFree(temb_tab);
        if (temb_tmp) (void)hipFree(temb_tmp);
        if (temb_lab) (void)hipFree(temb_lab);
        temb_tab = nullptr; temb_tmp = nullptr; temb_lab = nullptr; temb_rows_cap = 0;
        MCVD_HIP_CHECK(hipMalloc((void**)&temb_tab, rows * (size_t)NE * sizeof(float)));
        MCVD_HIP_CHECK(hipMalloc((void**)&temb_tmp, rows * (size_t)T * sizeof(float)));
        MCVD_HIP_CHECK(hipMalloc((void**)&temb_lab, rows * sizeof(int64_t)));
        temb_rows_cap = rows;
    }
    std::vector<int64_t> h(labels.begin(), labels.end());
    MCVD_HIP_CHECK(hipMemcpyAsync(temb_lab, h.data(), rows * sizeof(int64_t), hipMemcpyHostToDevice, s));
    MCVD_HIP_CHECK(hipStreamSynchronize(s));           // (h is a pageable temporary; once per sampler call)
    const float* w0 = blob + params[find_param("unet.all_modules.0.weight")].off;
    const float* b0 = blob + params[find_param("unet.all_modules.0.bias")].off;
    const float* w1 = blob + params[find_param("unet.all_modules.1.weight")].off;
    const float* b1 = blob + params[find_param("unet.all_modules.1.bias")].off;
    if (int rc = launch_temb_mlp(temb_lab, 0, packed + freqs_off, w0, b0, w1, b1, temb_tmp, (int)rows, d.ngf, T, nullptr, nullptr, s)) return rc;
    if (int rc = launch_dense_all(temb_tmp, packed + dense_wt, packed + dense_bias, temb_tab, (int)rows, T, NE, s)) return rc;
    temb_row_live = 1;
    return 0;
}

// the table's row `row` -> the Dense_0 output buffer of the forward about to run (the only thing of the two skipped ops a forward reads; the
// time MLP's own output buffer keeps the values of the last forward that ran it)
int mcvd_model::use_temb_row(int row, int B) {
    MCVD_REQUIRE(temb_row_live && row >= 0 && (size_t)row < temb_rows_cap, "use_temb_row: row %d", row);
    float* dense_dst = arena + ops[1].dst.off * (int64_t)B;
    MCVD_HIP_CHECK(hipMemcpyAsync(dense_dst, temb_tab + (size_t)row * NE, (size_t)NE * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

void mcvd_model::drop_graph() {
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    graph_exec = nullptr;
    graph_key = GraphKey{};
}

int mcvd_model::forward_ops(const float* x, const void* lab, const float* cond, float* out, int B) {
    const bool use_side = ctx->side_stream && !ctx->naive_conv;
    if (use_side && !ctx->side) {
        MCVD_HIP_CHECK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        MCVD_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        MCVD_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    for (const Op& op : ops) {
        if (op.prep) continue;
        if (use_side && op.side) {                       // fork: the shortcut conv overlaps the main chain
            MCVD_HIP_CHECK(hipEventRecord(ctx->ev_fork, ctx->stream));
            MCVD_HIP_CHECK(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
            op_stream = ctx->side;
            const int rc = launch_op(op, x, lab, cond, out, B);
            op_stream = nullptr;
            if (rc) return rc;
            MCVD_HIP_CHECK(hipEventRecord(ctx->ev_join, ctx->side));
            continue;
        }
        if (use_side && op.join) MCVD_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        if (int rc = launch_op(op, x, lab, cond, out, B)) return rc;
    }
    return 0;
}
