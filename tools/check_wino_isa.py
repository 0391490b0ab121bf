"""Static check of the invariants conv_wino.cpp's hand-managed VMEM waits rely on (run by __graft_entry__.build() and the CPU tests).

The kernel's K loop issues its global loads through inline asm and waits for them with explicit, in-order `s_waitcnt vmcnt(N)`
counts, because hipcc's own waitcnt insertion is not exact for loop-carried loads.  The compiler therefore does not know that
the destination registers of those loads are pending.  That is only sound if, in the generated code,
  1. nothing reads or overwrites a destination register between the load and a following `s_waitcnt vmcnt` (no copies, no
     spills, no reuse as a temporary), following the loop's layout order and wrapping around the back edge;
  2. the loop contains no scratch (spill) instructions: they are VMEM operations and would break the counts;
  3. the loop's VMEM instructions are exactly the asm loads (COT weight loads twice per chunk + 3 patch loads, 2 for the 8x8 form)
     plus, in the SPADE-prologue instantiations (PRO 3), two LDS-DMA loads per patch load (gamma | beta), which the vmcnt counts include.
This script compiles the file to gfx950 assembly and verifies 1-3 for every instantiation of conv_wino_kernel.

conv_wino2h.cpp (two fp16 pieces per operand, pre-split weights) and conv_wino3.cpp (three bf16 pieces, pre-split weights) count
their VMEM the same way: their K loops (two per kernel, one per phase order) must hold exactly 4*COT (6*COT) weight loads + 6 patch
loads, no scratch traffic, and keep the load destinations untouched up to a vmcnt wait that covers them.  Their MFMAs are inline
asm that read their A operand straight out of the load destinations, so in addition nothing but MFMAs may touch an accumulator
register inside a K loop (the compiler does not know they are MFMA results and would not insert the wait states).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mcvd_pytorch_amd", "csrc", "kernels", "conv_wino.cpp")
SRC3 = os.path.join(ROOT, "mcvd_pytorch_amd", "csrc", "kernels", "conv_wino3.cpp")
SRC2H = os.path.join(ROOT, "mcvd_pytorch_amd", "csrc", "kernels", "conv_wino2h.cpp")
SRC3P = os.path.join(ROOT, "mcvd_pytorch_amd", "csrc", "kernels", "conv_wino3p.cpp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
DIAG = "--diag" in sys.argv[1:]          # check the diagnostics library's code (-DMCVD_DIAG)


def _regs(text):
    r = {int(x) for x in re.findall(r"\bv(\d+)\b", text)}
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        r |= set(range(int(a), int(b) + 1))
    return r


def check(asm_text):
    problems, seen = [], 0
    for m in re.finditer(r"^(_ZN4mcvd16conv_wino_kernelILi(\d)ELi(\d)ELb(\d)E(?:Li0E)?EEvNS_8ConvArgsE):[^\n]*\n(.*?)\.Lfunc_end", asm_text, re.S | re.M):
        name, cot, pro, g8, body = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), m.group(5)
        npatch = 2 if g8 else 3                                        # patch loads per thread and chunk (MAXP)
        ndma = 2 * npatch if pro == 3 else 0                           # SPADE prologue: gamma | beta of every patch element by LDS-DMA
        seen += 1
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        starts = [i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l) and "Loop" in l]
        if not starts:
            problems.append(f"{name}: no loop found")
            continue
        end = starts[-1] + 1
        while end < len(lines) and not re.match(r"^\.LBB\d+_\d+:", lines[end]):
            end += 1
        loop = [l for l in lines[starts[0]:end] if not re.match(r"^\.LBB", l)]
        vmem = [l for l in loop if re.match(r"^(global_|buffer_|scratch_|flat_)", l)]
        if any(l.startswith("scratch_") for l in vmem):
            problems.append(f"{name}: spill code inside the K loop")
        loads = [l for l in vmem if re.match(r"^global_load_dword(x4)? v", l)]
        dmas = [l for l in vmem if re.match(r"^global_load_lds_dword ", l)]
        if len(loads) + len(dmas) != len(vmem) or len(loads) != 2 * cot + npatch or len(dmas) != ndma:
            problems.append(f"{name}: expected {2 * cot + npatch} asm loads + {ndma} LDS-DMA loads and no other VMEM in the loop, "
                            f"found {len(loads)} + {len(dmas)} of {len(vmem)}")
        n = len(loop)
        for i, l in enumerate(loop):
            mm = re.match(r"^global_load_dword(?:x4)? (v\d+|v\[\d+:\d+\]),", l)
            if not mm:
                continue
            dest = _regs(mm.group(1))
            for k in range(1, n + 1):                                   # layout order, wrapping around the back edge
                nxt = loop[(i + k) % n]
                if re.match(r"^s_waitcnt.*vmcnt", nxt):
                    break
                ops = nxt.split(None, 1)
                if len(ops) > 1 and _regs(ops[1]) & dest:
                    problems.append(f"{name}: `{nxt}` touches the destination of `{l}` before any vmcnt wait")
                    break
    if seen != 24:
        problems.append(f"expected 24 instantiations of conv_wino_kernel, found {seen}")
    return problems


def _dest_untouched(name, loop, problems):
    n = len(loop)
    for i, l in enumerate(loop):
        mm = re.match(r"^global_load_dword(?:x4)? (v\d+|v\[\d+:\d+\]),", l)
        if not mm:
            continue
        dest = _regs(mm.group(1))
        for k in range(1, n + 1):                                   # layout order, wrapping around the back edge
            nxt = loop[(i + k) % n]
            if re.match(r"^s_waitcnt.*vmcnt", nxt):
                break
            ops = nxt.split(None, 1)
            if len(ops) > 1 and _regs(ops[1]) & dest:
                problems.append(f"{name}: `{nxt}` touches the destination of `{l}` before any vmcnt wait")
                break


def _dest_untouched_covering(name, loop, problems):
    """Stricter form for loops with several wait points: a load's destination stays untouched until a vmcnt(N) wait with N <= the
    number of loads issued after it (in-order return: only then has it landed), following the back edge."""
    n = len(loop)
    for i, l in enumerate(loop):
        mm = re.match(r"^global_load_dword(?:x4)? (v\d+|v\[\d+:\d+\]),", l)
        if not mm:
            continue
        dest = _regs(mm.group(1))
        younger = 0
        for k in range(1, 2 * n + 1):
            nxt = loop[(i + k) % n]
            if nxt.startswith("global_load"):
                younger += 1
                continue
            wm = re.match(r"^s_waitcnt.*vmcnt\((\d+)\)", nxt)
            if wm:
                if int(wm.group(1)) <= younger:
                    break
                continue
            ops = nxt.split(None, 1)
            if len(ops) > 1 and _regs(ops[1]) & dest:
                problems.append(f"{name}: `{nxt}` touches the destination of `{l}` before a wait that covers it")
                break
        else:
            problems.append(f"{name}: no wait covers `{l}`")


def _inner_loop_spans(raw):
    """[(first line, end line)] of the INNERMOST loops of a function body given as raw lines (comments kept): LLVM marks a nested loop's
    header with `Parent Loop ...` on the label line and `=>This Inner Loop Header: Depth=N` on a following comment line, its other
    blocks with `in Loop: Header=BBx_y Depth=N`."""
    labels = [k for k, l in enumerate(raw) if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb.")]
    spans = []
    for k in labels:
        hm = re.match(r"^\.LBB(\d+)_(\d+):", raw[k])
        if not hm:
            continue
        head = raw[k]
        q = k + 1
        while q < len(raw) and raw[q].startswith(";") and not raw[q].startswith("; %bb.") and not raw[q].startswith(";;#"):
            head += " " + raw[q]
            q += 1
        if "Inner Loop Header" not in head:
            continue
        tag = f"Header=BB{hm.group(1)}_{hm.group(2)} "
        members = [k] + [q for q in labels if tag in raw[q] + " "]
        lo, hi_start = min(members), max(members)
        nxt = [q for q in labels if q > hi_start]
        spans.append((lo, nxt[0] if nxt else len(raw)))
    return spans


def check3(asm_text, kernel="17conv_wino3_kernel", asm_mfma=True, expect=18, wl_per_cot=6, nloops=2, npatch=(3, 1), wreg0=184, named0=172,
           targs=r"ILi(\d)ELi(\d)E(?:Lb([01])E)?Li0EEE", nested=False):
    """conv_wino3_kernel<COT, PRO, G8, 0> / conv_wino2h_kernel<COT, PRO, G8, 0>: every loop that holds MFMAs is a K loop; a K loop
    holds wl_per_cot * COT weight loads (NP pieces x 2 positions; destinations from register `wreg0` up) + npatch[G8] patch loads
    (destinations below `wreg0`)."""
    problems, seen = [], 0
    for m in re.finditer(r"^(_ZN4mcvd" + kernel + targs + r"vNS_8ConvArgsE):[^\n]*\n(.*?)\.Lfunc_end", asm_text, re.S | re.M):
        name, cot, g8, body = m.group(1), int(m.group(2)), int(m.group(4) or 0), m.group(5)
        seen += 1
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        if nested:        # the K loops are the innermost loops of a loop nest: the comment lines carry the loop structure
            raw = [l.strip() for l in body.split("\n") if l.strip()]
            raw_spans = _inner_loop_spans(raw)
            # map raw spans onto `lines` through the labels that bound them
            pos = {}
            for k, l in enumerate(lines):
                if re.match(r"^\.LBB\d+_\d+:", l):
                    pos.setdefault(l.split(":")[0], k)

            def to_line(q):          # raw index of a block boundary -> index in `lines` of the next label at or behind it
                while q < len(raw) and not re.match(r"^\.LBB\d+_\d+:", raw[q]):
                    q += 1
                return pos[raw[q].split(":")[0]] if q < len(raw) else len(lines)
            nested_spans = [(to_line(lo), to_line(hi)) for lo, hi in raw_spans]
        kloops = 0
        # loops: a header label ("Loop Header") plus every block the compiler marks "in Loop: Header=<that label>"; the compiler may lay
        # rotated blocks out BEFORE the header, so the loop is the contiguous label range that covers all of them (the in-order wait
        # analysis below walks the lines cyclically: where the cycle is entered does not matter)
        headers = [(k, re.match(r"^\.LBB(\d+)_(\d+):", l)) for k, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:.*Loop Header", l)]
        label_idx = [k for k, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l) or re.match(r"^; %bb\.", l)]
        spans = []
        for k, hm in headers:
            tag = f"Header=BB{hm.group(1)}_{hm.group(2)} "
            members = [k] + [q for q, l in enumerate(lines) if (re.match(r"^\.LBB", l) or l.startswith("; %bb.")) and tag in l + " "]
            lo, hi_start = min(members), max(members)
            nxt = [q for q in label_idx if q > hi_start]
            spans.append((lo, nxt[0] if nxt else len(lines)))
        if nested:
            spans = nested_spans
        for i0, j in spans:
            loop = [l for l in lines[i0:j] if not re.match(r"^\.LBB", l) and not l.startswith("; %bb.")]
            if not any(l.startswith("v_mfma") for l in loop):
                continue
            kloops += 1
            vmem = [l for l in loop if re.match(r"^(global_|buffer_|scratch_|flat_)", l)]
            if any(l.startswith("scratch_") for l in vmem):
                problems.append(f"{name}: spill code inside a K loop")
            if any(re.match(r"^s_waitcnt.*vmcnt\(0\)", l) for l in loop):
                problems.append(f"{name}: a vmcnt(0) wait inside a K loop (the loop's loads are meant to stay in flight)")
            ld = [l for l in vmem if re.match(r"^global_load_dword(x4)? v", l)]
            wl = [l for l in ld if min(_regs(l.split(",")[0])) >= wreg0]
            pl = [l for l in ld if min(_regs(l.split(",")[0])) < wreg0]
            if len(wl) != wl_per_cot * cot or len(pl) != npatch[g8] or len(wl) + len(pl) != len(vmem):
                problems.append(f"{name}: expected {wl_per_cot * cot} weight + {npatch[g8]} patch loads and no other VMEM in a K loop, found {len(wl)} + {len(pl)} of {len(vmem)}")
            _dest_untouched_covering(name, loop, problems)
            if asm_mfma:      # accumulators: written and read by MFMAs only
                acc = set()
                for l in loop:
                    if l.startswith("v_mfma"):
                        acc |= _regs(l.split(",")[0])
                for l in loop:
                    ops = l.split(None, 1)
                    if not l.startswith("v_mfma") and len(ops) > 1 and _regs(ops[1]) & acc:
                        problems.append(f"{name}: `{l}` touches an accumulator inside a K loop")
                        break
            # leaving the loop: its loads may still be in flight; nothing may read or overwrite their destinations before vmcnt(0)
            # (a linear walk along fall-through edges: the diagnostics build's per-phase clock stamps put conditional code between the
            # loops and their drain, which the walk cannot see through -- `--diag` checks everything but this)
            if DIAG:
                continue
            dests = set()
            for l in wl + pl:
                dests |= _regs(l.split(",")[0])
            # the exit edge: the last conditional branch of the loop that leaves it (else the code laid out behind the loop)
            inside = {re.match(r"^(\.LBB\d+_\d+):", l).group(1) for l in lines[i0:j] if re.match(r"^\.LBB\d+_\d+:", l)}
            # every conditional branch of the loop to a label outside its span is a candidate exit; a candidate whose path runs straight back
            # into the loop (a latch block the compiler laid out in front of the header) is a back edge, not an exit
            starts = []
            for l in lines[i0:j]:
                bm = re.match(r"^s_cbranch_\w+ (\.LBB\d+_\d+)", l)
                if bm and bm.group(1) not in inside:
                    tgt = [k for k, t in enumerate(lines) if t.startswith(bm.group(1) + ":")]
                    if tgt and tgt[0] not in starts:
                        starts.append(tgt[0])
            if not starts:
                starts = [j]
            for start in starts:
                pos, steps = start, 0
                while pos < len(lines) and steps < 4000:          # follow the fall-through path and unconditional branches
                    t = lines[pos]
                    pos += 1
                    steps += 1
                    lm = re.match(r"^(\.LBB\d+_\d+):", t)
                    if lm and lm.group(1) in inside:
                        break                                      # back inside the loop: this candidate was a back edge
                    if re.match(r"^\.LBB", t):
                        continue
                    bm = re.match(r"^s_(?:branch|cbranch_execnz) (\.LBB\d+_\d+)", t)      # (EXEC is never zero in this kernel)
                    if bm:
                        tgt = [k for k, u in enumerate(lines) if u.startswith(bm.group(1) + ":")]
                        if not tgt:
                            break
                        pos = tgt[0]
                        continue
                    if re.match(r"^s_waitcnt.*vmcnt\(0\)", t):
                        break
                    ops = t.split(None, 1)
                    if len(ops) > 1 and not t.startswith("s_waitcnt") and _regs(ops[1]) & dests:
                        problems.append(f"{name}: `{t}` touches a K-loop load destination after the loop, before vmcnt(0)")
                        break
        if kloops != nloops:
            problems.append(f"{name}: expected {nloops} K loop(s), found {kloops}")
        # anywhere in the kernel (prologue and epilogue included): the registers from `named0` up are touched by the kernel's own asm
        # statements only (the text between the compiler's #ASMSTART / #ASMEND markers) -- loads into them, the patch / coefficient
        # reads and the MFMAs' A operands.  The register cap that keeps the allocator away from them (amdgpu_num_vgpr) is silently
        # dropped when LLVM cannot honour it.
        in_asm = False
        for l in (x.strip() for x in body.split("\n")):
            if l.startswith(";;#ASMSTART"):
                in_asm = True
            elif l.startswith(";;#ASMEND"):
                in_asm = False
            ops = l.split(None, 1)
            if in_asm or len(ops) < 2 or l.startswith((".", ";")):
                continue
            if max(_regs(ops[1].split(";")[0]) or {0}) >= named0:
                problems.append(f"{name}: `{l}` (compiler-generated) touches a named register")
                break
    if seen != expect:
        problems.append(f"expected {expect} instantiations of {kernel}<COT, PRO, ..., 0>, found {seen}")
    return problems


def check3p(asm_text):
    """conv_wino3p_kernel<COT, PRO> (persistent workgroups): the K loops are the two innermost loops (one per phase order) of the
    run / item loop nest; same VMEM population per chunk as conv_wino3_kernel."""
    return check3(asm_text, kernel="18conv_wino3p_kernel", asm_mfma=True, expect=9, wl_per_cot=6, nloops=2, npatch=(3, 3), wreg0=184, named0=172,
                  targs=r"ILi(\d)ELi(\d)E()EE", nested=True)


def check2h(asm_text):
    return check3(asm_text, kernel="18conv_wino2h_kernel", asm_mfma=True, expect=18, wl_per_cot=4, nloops=2, npatch=(6, 6), wreg0=208, named0=202)      # x {8x16 regions, 8x8 images}; one loop per phase order


def main():
    problems = []
    with tempfile.TemporaryDirectory() as td:
        for src, fn in ((SRC, check), (SRC3, check3), (SRC2H, check2h), (SRC3P, check3p)):
            out = os.path.join(td, os.path.basename(src)[:-4] + ".s")
            cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                   src, "-o", out] + (["-DMCVD_DIAG"] if DIAG else [])
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                print(r.stderr, file=sys.stderr)
                return 2
            problems += fn(open(out).read())
    for p in problems:
        print("conv_wino ISA check:", p, file=sys.stderr)
    if not problems:
        print("conv_wino ISA check: ok (asm-load destinations untouched until their vmcnt wait, no spills in the K loop)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
