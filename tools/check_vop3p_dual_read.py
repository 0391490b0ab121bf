"""Static check of the built library: no packed-fp32 instruction may read ONE VGPR pair as both src1 and src2.

Why (profiles/r06_coresident_cause.txt, tools/repro_pk_fma_beside_mfma.cpp): on gfx950

    v_pk_fma_f32 vD, vX, v[n:n+1], v[n:n+1] op_sel:[0,0,1] op_sel_hi:[1,0,1]        (hipcc's code for  x * c.x + c.y  over a float4)

returns  x.lo * c.x + 0  in its low half for one 16-lane pass when a wave of ANOTHER kernel on the same SIMD executes one of the matrix
instructions with 128-bit A / B operands (v_mfma_f32_32x32x16_bf16 / _f16, v_mfma_f32_16x16x32_bf16).  That is the whole of the
"co-residency corruption" of rounds 4-6.  The compiler cannot know; the library simply must not contain the form.  The sources write such
affines with fma_unpacked() (common.h); this script disassembles every gfx950 code object of libmcvd_hip.so and fails if the form is back.

    python tools/check_vop3p_dual_read.py [path/to/libmcvd_hip.so]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """the gfx950 code objects of every translation unit linked into the library (uncompressed clang offload bundles in .hip_fatbin)"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(tmp, "unused.so")], check=True)
        data = open(fat, "rb").read()
    pos = data.find(MAGIC)
    while pos >= 0:
        o = pos + len(MAGIC)
        (n,) = struct.unpack_from("<Q", data, o)
        o += 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size:
                yield data[pos + off:pos + off + size]
        pos = data.find(MAGIC, pos + 1)


def disassemble(blob):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(blob)
        f.flush()
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], check=True, capture_output=True, text=True).stdout


INSN = re.compile(r"^\s*(v_pk_\w+)\s+(.*)$")


def scan(text):
    """[(kernel, instruction text)] for every VOP3P instruction whose src1 and src2 are the same VGPR pair"""
    hits, kern = [], "?"
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            kern = m.group(1)
            continue
        m = INSN.match(ln.split("//")[0].rstrip())
        if not m:
            continue
        ops = m.group(2)
        cut = min([i for i in (ops.find(" op_sel"), ops.find(" neg_"), ops.find(" clamp")) if i >= 0] or [len(ops)])
        srcs = [o.strip() for o in ops[:cut].split(",")][1:]           # without the destination
        if len(srcs) == 3 and srcs[1] == srcs[2] and srcs[1].startswith("v["):
            hits.append((kern, ln.split("//")[0].strip()))
    return hits


def main(argv):
    lib = argv[1] if len(argv) > 1 else os.path.join(ROOT, "mcvd_pytorch_amd", "libmcvd_hip.so")
    n_obj, n_pk, hits = 0, 0, []
    for blob in code_objects(lib):
        n_obj += 1
        text = disassemble(blob)
        n_pk += len(re.findall(r"^\s*v_pk_", text, re.M))
        hits += scan(text)
    if not n_obj:
        print("check_vop3p_dual_read: no gfx950 code object found in", lib)
        return 2
    if hits:
        print(f"check_vop3p_dual_read: {len(hits)} packed instructions read one VGPR pair as src1 AND src2 (the gfx950 co-residency erratum form):")
        seen = {}
        for k, ins in hits:
            seen.setdefault(k, []).append(ins)
        for k, v in seen.items():
            print(f"  {k}: {len(v)}   e.g. {v[0]}")
        return 1
    print(f"check_vop3p_dual_read: ok ({n_obj} code objects, {n_pk} packed instructions, none reads one VGPR pair as src1 and src2)")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
