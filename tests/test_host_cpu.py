"""CPU (no GPU): the C-ABI library loads, exports every symbol of include/mcvd_hip.h, fails loudly without a
device, and its host-side plan (parameter table, schedule) agrees with the reference-pinned oracle."""
import ctypes as C
import os
import re

import pytest
import torch

from oracle import synth, unet_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mcvd_pytorch_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mcvd_hip.h")).read()
    declared = set(re.findall(r"\b(mcvd_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mcvd_ctx", "mcvd_model", "mcvd_unet_desc"}
    assert declared, "no declarations parsed"
    raw = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in mcvd_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


def test_product_library_has_no_diagnostics_hooks():
    """The timing-only ablation kernels (wrong results by design) and every environment variable that selects them or truncates a
    sampler exist in the diagnostics build only (csrc/build.py --diag, -DMCVD_DIAG): none of their names is in libmcvd_hip.so, and
    the only getenv calls of the library are the option defaults read once in mcvd_ctx_create (api.cpp)."""
    from mcvd_pytorch_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"MCVD_WINO_EXP", b"MCVD_WINO2H_EXP", b"MCVD_WINO3_EXP", b"MCVD_Q1_EXP", b"MCVD_DBG_WAVE", b"MCVD_FPNDM_MAXSTEPS",
                 b"MCVD_CONV_SHAPE", b"MCVD_CONV_MIN_BLOCKS", b"MCVD_Q1_OCC"):
        assert name not in blob, name
    src = os.path.join(ROOT, "mcvd_pytorch_amd", "csrc")
    for dirpath, _, files in os.walk(src):
        for f in files:
            if not f.endswith((".cpp", ".h")) or "build" in dirpath:
                continue
            text = open(os.path.join(dirpath, f)).read()
            # strip the diagnostics-only regions
            text = re.sub(r"#ifdef MCVD_DIAG.*?#(?:else|endif)", "", text, flags=re.S)
            if f != "api.cpp":
                assert "getenv(" not in text, f"{f}: getenv outside #ifdef MCVD_DIAG"


def test_no_gpu_fails_loudly():
    from mcvd_pytorch_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = _lib.lib.mcvd_ctx_create(0, None, C.byref(ctx))
    assert rc != 0 and "device" in _lib.last_error().lower()
    from mcvd_pytorch_amd.scorenet import HipScoreNet
    with pytest.raises(RuntimeError):
        HipScoreNet(synth.make_config("tiny"))


def test_samplers_refuse_foreign_scorenet():
    from mcvd_pytorch_amd.samplers import ddpm_sampler
    with pytest.raises(TypeError):
        ddpm_sampler(torch.zeros(1, 2, 32, 32), lambda x, y, cond=None: x)


def _plan(name):
    from mcvd_pytorch_amd import _lib
    from mcvd_pytorch_amd.config import desc_from_config
    config = synth.make_config(name)
    desc = desc_from_config(config)
    m = C.c_void_p()
    _lib.check(_lib.lib.mcvd_model_create(None, C.byref(desc), C.byref(m)), "model_create(plan only)")
    return _lib, config, m


@pytest.mark.parametrize("name", ["tiny", "tiny_spade", "smmnist_big5", "smmnist_big5_ngf96", "kth64_big_ngf128",
                                  "bair_big_spade", "cityscapes_big", "cityscapes_big_variant", "tiny_condemb", "tiny_gamma",
                                  "tiny_spade_noisecond", "cityscapes_big_spade"])
def test_plan_parameter_table_matches_reference_names(name):
    """Names, shapes and ORDER equal the reference state_dict (oracle.param_shapes is pinned to it by gen_golden)."""
    _lib, config, m = _plan(name)
    want = unet_ref.param_shapes(unet_ref.hot_cfg(config))
    n = _lib.lib.mcvd_model_num_params(m)
    pname, shape, ndim, off = C.c_char_p(), (C.c_int64 * 4)(), C.c_int(), C.c_int64()
    have = []
    for i in range(n):
        _lib.check(_lib.lib.mcvd_model_param_info(m, i, C.byref(pname), shape, C.byref(ndim), C.byref(off)))
        have.append((pname.value.decode(), tuple(shape[k] for k in range(ndim.value))))
        assert off.value % 4 == 0
    assert have == list(want.items())
    _lib.lib.mcvd_model_destroy(m)


@pytest.mark.parametrize("name", ["tiny", "smmnist_big5"])
def test_library_default_schedule_close_to_torch(name):
    """The C restatement of linspace/cumprod (used when no host overrides it) vs torch: a few ulp at most."""
    _lib, config, m = _plan(name)
    T = config.model.num_classes
    bufs = [torch.empty(T) for _ in range(3)]
    _lib.check(_lib.lib.mcvd_model_get_schedule(m, *[C.c_void_p(b.data_ptr()) for b in bufs], T))
    ref = unet_ref.make_schedule(unet_ref.hot_cfg(config))
    for got, want in zip(bufs, ref):
        torch.testing.assert_close(got, want, rtol=2e-6, atol=0)
    _lib.lib.mcvd_model_destroy(m)


def test_unsupported_configs_raise():
    from mcvd_pytorch_amd.config import desc_from_config
    cfg = synth.make_config("tiny")
    cfg.model.time_conditional = False
    with pytest.raises(NotImplementedError):
        desc_from_config(cfg)
    cfg = synth.make_config("tiny")
    cfg.model.arch = "unetmore3d"
    with pytest.raises(NotImplementedError):
        desc_from_config(cfg)


def test_winograd_isa_checker_flags_violations():
    """tools/check_wino_isa.py (run by the build whenever conv_wino.cpp is recompiled): a register written by an in-flight
    asm load must not be touched before a vmcnt wait; spills or foreign VMEM in the K loop are rejected."""
    from tools import check_wino_isa as c
    head = "_ZN4mcvd16conv_wino_kernelILi1ELi0ELb0EEEvNS_8ConvArgsE: ; @x\n"
    loop_ok = (".LBB0_1: ; =>This Inner Loop Header: Depth=1\n"
               "\tv_mfma_f32_32x32x2_f32 v[0:15], v20, v21, v[0:15]\n"
               "\ts_waitcnt vmcnt(1)\n"
               "\tglobal_load_dwordx4 v[20:23], v30, s[2:3]\n"
               "\tglobal_load_dword v40, v31, s[4:5]\n\tglobal_load_dword v41, v31, s[4:5]\n\tglobal_load_dword v42, v31, s[4:5]\n"
               "\tglobal_load_dwordx4 v[24:27], v30, s[2:3]\n"
               "\tv_add_u32_e32 v50, v51, v52\n"
               "\ts_waitcnt vmcnt(4)\n"
               "\tv_mfma_f32_32x32x2_f32 v[0:15], v24, v21, v[0:15]\n"
               "\ts_cbranch_scc1 .LBB0_1\n.LBB0_2:\n\ts_endpgm\n.Lfunc_end0:\n")
    others = "".join(f"_ZN4mcvd16conv_wino_kernelILi{a}ELi{b}ELb0EEEvNS_8ConvArgsE: ; @x\n" + loop_ok.replace("LBB0", f"LBB{a}{b}")
                     for a, b in [(1, 1), (1, 2)])
    spade_loop = loop_ok.replace("\tglobal_load_dwordx4 v[24:27], v30, s[2:3]\n",
                                 "".join("\tglobal_load_lds_dword v[60:61], off\n" for _ in range(6)) + "\tglobal_load_dwordx4 v[24:27], v30, s[2:3]\n")
    others += "_ZN4mcvd16conv_wino_kernelILi1ELi3ELb0EEEvNS_8ConvArgsE: ; @x\n" + spade_loop.replace("LBB0", "LBB13")
    def problems(loop):
        return [p for p in c.check(head + loop + others) if "ILi1ELi0ELb0" in p]
    assert problems(loop_ok) == []
    bad_copy = loop_ok.replace("\tv_add_u32_e32 v50, v51, v52\n", "\tv_mov_b32_e32 v50, v25\n")
    assert any("touches the destination" in p for p in problems(bad_copy))
    bad_spill = loop_ok.replace("\tv_add_u32_e32 v50, v51, v52\n", "\tscratch_store_dword off, v50, off\n")
    assert any("spill" in p for p in problems(bad_spill))
    assert not [p for p in c.check(head + loop_ok + others) if "ILi1ELi3ELb0" in p]          # PRO 3: six LDS-DMA loads are expected
    assert any("LDS-DMA" in p for p in c.check(head + spade_loop + others) if "ILi1ELi0ELb0" in p)   # ... and rejected elsewhere


def test_no_kernel_of_the_library_holds_the_gfx950_coresidency_erratum_form():
    """tools/check_vop3p_dual_read.py (run by every build after the link): a packed-fp32 instruction that reads ONE VGPR pair as src1 and src2
    loses its low addend on gfx950 beside another kernel's 128-bit-operand MFMA (profiles/r06_coresident_cause.txt).  The scanner flags exactly
    that form; the built library's 22 code objects hold none."""
    import os
    from tools import check_vop3p_dual_read as c
    text = ("0000000000001900 <_ZN4mcvd1kE>:\n"
            "\tv_pk_fma_f32 v[4:5], v[100:101], v[138:139], v[138:139] op_sel:[0,0,1] op_sel_hi:[1,0,1]// 000000001900: D3B06004 0E2B1564\n"
            "\tv_pk_fma_f32 v[66:67], v[78:79], v[50:51], v[78:79] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n"          # src0 == src2: measured clean
            "\tv_pk_fma_f32 v[4:5], v[100:101], v[138:139], v[142:143] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"      # the addend from a copy: measured clean
            "\tv_pk_add_f32 v[4:5], v[100:101], v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n"                      # two operands only: measured clean
            "\tv_pk_mul_f32 v[2:3], v[2:3], v[22:23]\n")
    hits = c.scan(text)
    assert len(hits) == 1 and hits[0][0] == "_ZN4mcvd1kE" and hits[0][1].startswith("v_pk_fma_f32 v[4:5], v[100:101], v[138:139], v[138:139]")
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mcvd_pytorch_amd", "libmcvd_hip.so")
    assert os.path.exists(lib), "build first (__graft_entry__.build())"
    assert c.main(["check", lib]) == 0


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under mcvd_pytorch_amd/ (Python or native sources) may import, open or link it,
    and outside the package only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do."""
    import re
    pkg = os.path.join(ROOT, "mcvd_pytorch_amd")
    offenders = []
    for d, _, files in os.walk(pkg):
        if os.sep + "build" in d or "__pycache__" in d:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"\boracle\b", txt):
                    offenders.append(os.path.relpath(os.path.join(d, f), ROOT))
    assert not offenders, offenders
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle|import oracle", bench)]
    assert uses and all(bench.rfind("def ", 0, u) == bench.rfind("def cpu_baseline", 0, u) for u in uses), \
        "bench.py may use the oracle only inside cpu_baseline()"


def _ema_round_trip(helper_cls):
    """load_state_dict(states[0]) -> register -> load_state_dict(states[-1]) -> ema(net), as runners/ncsn_runner.py:926-932."""
    from mcvd_pytorch_amd.scorenet import HipScoreNet
    config = synth.make_config("tiny")
    ema_sd = synth.make_state_dict(config, seed=123)
    raw = {"module." + k: v + 1.0 for k, v in ema_sd.items()}
    net = HipScoreNet(config, plan_only=True)                  # parameter table only: no GPU needed, cannot compute
    assert not net.load_state_dict(raw, strict=False).missing_keys
    assert all(p.requires_grad for p in net.parameters())      # EMAHelper skips parameters that do not (models/ema.py:12-13, 26-28)
    helper = helper_cls(mu=0.999)
    helper.register(net)
    registered = dict(helper.shadow)
    helper.load_state_dict(dict(ema_sd))
    net._dirty = False
    helper.ema(net)
    return net, registered, ema_sd, raw


def test_ema_helper_protocol():
    """The EMA shadow must land in the parameters HipScoreNet uploads (requires_grad=False made the reference's
    EMAHelper a silent no-op).  Runs the restated helper always and the REAL reference class when /root/reference is present
    (build container), and demands identical outcomes."""
    import importlib.util
    from oracle.ema_ref import EMAHelper as Restated
    helpers = [Restated]
    ref_path = "/root/reference/models/ema.py"
    if os.path.exists(ref_path):
        spec = importlib.util.spec_from_file_location("ref_ema", ref_path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        helpers.append(mod.EMAHelper)
    outcomes = []
    for cls in helpers:
        net, registered, ema_sd, raw = _ema_round_trip(cls)
        assert set(registered) == set(ema_sd)                   # every parameter registered, bare names
        for k, v in registered.items():
            assert torch.equal(v, raw["module." + k])
        for k, p in net.named_parameters():
            assert torch.equal(p.data, ema_sd[k]), k            # ema() overwrote the raw weights
        outcomes.append({k: p.data.clone() for k, p in net.named_parameters()})
        with pytest.raises(RuntimeError):
            net(torch.zeros(1, 2, 32, 32), torch.zeros(1).long())    # plan-only objects cannot compute
    for o in outcomes[1:]:
        assert all(torch.equal(o[k], outcomes[0][k]) for k in o)


def test_bench_multi_gpu_flag_spawns_ranks():
    """bench.py --gpus N must start N ranks itself when no launcher did (WORLD_SIZE unset): checked on the argument plumbing only
    (no GPU here) through bench.plan_launch()."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    assert bench.plan_launch(1, env) is None                                    # N=1: run in-process
    cmd = bench.plan_launch(4, env, argv=["--gpus", "4", "--steps", "2"])
    assert cmd is not None and "--nproc-per-node" in " ".join(cmd) and "4" in cmd and "127.0.0.1" in " ".join(cmd)
    env["WORLD_SIZE"] = "4"
    assert bench.plan_launch(4, env) is None                                    # already under a launcher
    env["WORLD_SIZE"] = "2"
    with pytest.raises(SystemExit):
        bench.plan_launch(4, env)                                               # launcher / flag disagree: refuse


@pytest.mark.parametrize("name", ["tiny", "tiny_spade", "tiny_condemb", "smmnist_big5"])
def test_reset_parameters_has_the_reference_init_distributions(name, golden_dir):
    """`get_model(config)` without a checkpoint must hold what the reference's fresh `UNetMore_DDPM(config)` holds
    (models/better/layers.py:43-80 `default_init`; torch defaults for GroupNorm, SPADE convs, the cond_emb Embedding).  The fixture
    holds min / max / mean / std of every parameter tensor of the REAL reference's construction (oracle/gen_golden.py
    `gen_init_moments`); the draws differ (the reference draws in module-construction order from the global generator), the
    distributions must not: constant tensors exactly, uniform tensors by their support and their standard deviation."""
    import math
    import os
    from mcvd_pytorch_amd.scorenet import HipScoreNet
    want = torch.load(os.path.join(golden_dir, "init_moments.pt"), weights_only=False)[name]
    net = HipScoreNet(synth.make_config(name), plan_only=True)
    assert all(float(p.detach().abs().max()) == 0.0 for p in net.parameters())              # as constructed: nothing yet
    g = torch.Generator().manual_seed(11)
    assert net.reset_parameters(generator=g) is net and net._loaded
    have = dict(net.named_parameters())
    assert list(have.keys()) == list(want.keys())
    degenerate = 0
    for k, w in want.items():
        p = have[k].detach()
        assert list(p.shape) == w["shape"], k
        n = p.numel()
        if w["min"] == w["max"]:                                                    # zeros (biases) / ones (norm gains)
            assert float(p.min()) == w["min"] and float(p.max()) == w["max"], (k, float(p.min()), float(p.max()), w)
            continue
        # the reference tensor's extremes bound the support from inside: bound >= max|ref|, and for n samples of U(-b, b) the
        # largest |value| is above b (1 - 8 / n) with probability 1 - e^-8
        amax_ref, amax = max(abs(w["min"]), abs(w["max"])), float(p.abs().max())
        if k.endswith(".weight") and p.dim() == 2 and p.shape[0] == 2 and "all_modules.2." in k and name == "tiny_condemb":
            assert 0.5 < float(p.std()) < 1.6 and abs(float(p.mean())) < 0.8, k      # nn.Embedding: N(0, 1), 2 x 16 values
            continue
        slack = 1.0 + 12.0 / n
        assert amax <= amax_ref * slack * 1.02 and amax_ref <= amax * slack * 1.02, (k, amax, amax_ref)
        if n >= 256:
            assert abs(float(p.double().std()) / w["std"] - 1.0) < 4.0 / math.sqrt(n) + 0.02, (k, float(p.std()), w["std"])
            assert abs(float(p.double().mean())) < 6.0 * w["std"] / math.sqrt(n) + 1e-12, k
        degenerate += amax < 1e-6
    # init_scale = 0 -> 1e-10 leaves Conv_1 / NIN_3 / the last conv at |w| < 1e-6 (counted with the zero biases upstream)
    assert degenerate == sum(1 for w in want.values() if w["min"] != w["max"] and max(abs(w["min"]), abs(w["max"])) < 1e-6) > 0
