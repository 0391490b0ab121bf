"""Build libmcvd_hip.so (gfx950 only) with hipcc: one object per translation unit, compiled in parallel,
linked into mcvd_pytorch_amd/libmcvd_hip.so (in-tree, so it travels with the repo snapshot).

    python mcvd_pytorch_amd/csrc/build.py [--force] [-j N]
"""
import argparse
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(PKG, "libmcvd_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(ROOT, "include")]
WINO_OBJS = ("conv_wino.o", "conv_wino3.o", "conv_wino2h.o", "conv_wino3p.o")
HOST_ONLY_FLAGS = {"model.cpp": ["-ffp-contract=off"], "api.cpp": ["-ffp-contract=off"],
                   "sampler.cpp": ["-ffp-contract=off"]}      # sampler update kernels: one rounding per operation


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.cpp")) + glob.glob(os.path.join(HERE, "kernels", "*.cpp")))


def headers():
    return (glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "kernels", "*.h"))
            + glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.abspath(__file__)])


def compile_one(src, force, diag=False):
    obj = os.path.join(OBJ + ("_diag" if diag else ""), os.path.basename(src)[:-4] + ".o")
    newest_dep = max(os.path.getmtime(p) for p in [src] + headers())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest_dep:
        return obj, None
    cmd = [HIPCC] + FLAGS + (["-DMCVD_DIAG"] if diag else []) + HOST_ONLY_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
    return obj, r.stderr.strip()


def build(force=False, jobs=None, verbose=True, diag=False):
    """diag=True: the diagnostics library libmcvd_hip_diag.so (-DMCVD_DIAG: env hooks MCVD_DBG_WAVE / MCVD_WINO*_EXP / MCVD_FPNDM_MAXSTEPS and
    the timing-only ablation kernels, which produce WRONG results); tests/gpu_diag.py loads it through MCVD_LIB_PATH.  The product
    library has none of that."""
    lib = os.path.join(PKG, "libmcvd_hip_diag.so") if diag else LIB
    os.makedirs(OBJ + ("_diag" if diag else ""), exist_ok=True)
    srcs = sources()
    objs, rebuilt, wino_rebuilt = [], 0, False
    with cf.ThreadPoolExecutor(max_workers=jobs or os.cpu_count()) as ex:
        for obj, log in ex.map(lambda s: compile_one(s, force, diag), srcs):
            objs.append(obj)
            if log is not None:
                rebuilt += 1
                wino_rebuilt |= os.path.basename(obj) in WINO_OBJS
                if log and verbose:
                    print(log, file=sys.stderr)
    relink = rebuilt or not os.path.exists(lib) or force
    stamp = os.path.join(OBJ + ("_diag" if diag else ""), "wino_isa.ok")
    newest_wino = max(os.path.getmtime(os.path.join(OBJ + ("_diag" if diag else ""), o)) for o in WINO_OBJS)
    checked = os.path.exists(stamp) and os.path.getmtime(stamp) >= newest_wino
    if wino_rebuilt or (relink and not checked) or not checked:
        # The Winograd kernels manage their VMEM waits and their MFMA operand registers by hand (inline asm over registers the compiler is
        # only kept away from by amdgpu_num_vgpr): the generated code must keep the invariants that makes sound.  Checked on EVERY build
        # route -- product and diagnostics library, fresh or cached objects (the stamp is older than any unchecked object) -- before the link.
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_wino_isa.py")] + (["--diag"] if diag else []),
                           capture_output=True, text=True)
        if r.returncode != 0:
            for o in WINO_OBJS:
                try:
                    os.remove(os.path.join(OBJ + ("_diag" if diag else ""), o))
                except OSError:
                    pass
            raise RuntimeError("Winograd kernels: generated code violates the asm-load invariants\n" + r.stderr)
        open(stamp, "w").write(r.stdout)
        if verbose:
            print(r.stdout.strip())
    if relink:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stderr))
    if relink:
        # gfx950: a packed-fp32 instruction that reads ONE VGPR pair as src1 and src2 loses its low addend beside another kernel's 128-bit-operand
        # MFMA (the co-residency corruption of rounds 4-6, profiles/r06_coresident_cause.txt).  hipcc generates the form from ordinary source
        # (x * c.x + c.y over a float4), so the LINKED library is disassembled and checked; a library that contains it is not left behind.
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_vop3p_dual_read.py"), lib], capture_output=True, text=True)
        if r.returncode != 0:
            os.remove(lib)
            raise RuntimeError("the library contains the gfx950 co-residency erratum form (write the affine with fma_unpacked, common.h)\n" + r.stdout + r.stderr)
        if verbose:
            print(r.stdout.strip())
    if verbose:
        print("%s: %d/%d objects rebuilt -> %s" % (os.path.basename(lib), rebuilt, len(srcs), lib))
    return lib


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=None)
    ap.add_argument("--diag", action="store_true", help="build libmcvd_hip_diag.so (-DMCVD_DIAG) instead")
    a = ap.parse_args()
    build(force=a.force, jobs=a.j, diag=a.diag)
