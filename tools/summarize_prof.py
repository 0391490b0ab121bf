"""Condense rocprofv3 CSV output (kernel stats + per-dispatch PMC counters) into profiles-ready text."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def short(name):
    m = re.match(r".*conv_mfma_kernel<(\d+), (\d+), (\d+), (\d+), (\w+), (\w+)(?:, (\w+))?>.*", name)
    if m:
        ks, ck, cot, pxt, split, dma, wdbf = m.groups()
        return (f"conv_mfma<{ks}x{ks},CK{ck},co{32 * int(cot)},px{(1 if split == 'true' else 4) * 32 * int(pxt)}"
                f"{',splitK' if split == 'true' else ''}{',dma' if dma == 'true' else ''}{',wdb' if wdbf == 'true' else ''}>")
    m = re.match(r".*mcvd::(\w+)(<[^>]*>)?\(.*", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    return name[:70]


def stats():
    for f in glob.glob(os.path.join(OUT, "prof", "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        print(f"== kernel stats ({os.path.relpath(f, ROOT)}): name, calls, total ms, avg us, % of GPU time")
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        agg = defaultdict(lambda: [0, 0.0])
        for r in rows:
            k = short(r["Name"])
            agg[k][0] += int(r["Calls"])
            agg[k][1] += float(r["TotalDurationNs"])
        for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            print(f"{k:58s} {c:7d} {ns / 1e6:10.2f} ms {ns / c / 1e3:9.1f} us {100 * ns / tot:6.2f}%")
        conv3 = [(c, ns) for k, (c, ns) in agg.items() if k.startswith("conv_mfma<3x3")]
        if conv3:
            c = sum(x[0] for x in conv3); ns = sum(x[1] for x in conv3)
            print(f"-- all conv_mfma<3x3,...> instantiations: {c} launches, avg {ns / c / 1e3:.1f} us")


def pmc(dirname, counter):
    for f in glob.glob(os.path.join(OUT, dirname, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        print(f"== {counter} per launch ({os.path.relpath(f, ROOT)}; rocprofv3 units: KiB as reported)")
        for k, (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
            print(f"{k:58s} {c:6d} launches  avg {v / c:14.1f}  total {v:16.1f}")


def mfma():
    """MFMA-busy and effective clock per kernel from GRBM_GUI_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES + kernel durations."""
    for f in glob.glob(os.path.join(OUT, "pmc_mfma", "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        cols = rows[0].keys() if rows else []
        print("columns:", list(cols))
        agg = defaultdict(lambda: defaultdict(float))
        for r in rows:
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if "Start_Timestamp" in r and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                agg[k]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                agg[k]["n"] += 1
        # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs
        print("== per kernel: launches, total ms, effective shader clock (GUI_ACTIVE / 8 XCDs / duration), "
              "MFMA-busy fraction of the SIMD-cycles (MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs))")
        for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("ns", 0))[:24]:
            ns, gui, mf = d.get("ns", 0), d.get("GRBM_GUI_ACTIVE", 0), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
            if ns <= 0 or gui <= 0:
                continue
            print(f"{k:58s} {int(d['n']):5d} {ns / 1e6:9.2f} ms  clk {gui / 8 / ns:5.2f} GHz  mfma_busy {mf / (gui / 8 * 1024):6.3f}")


def sq():
    """Wave-state split per kernel from the SQ counters (quad-cycle units): fractions of SQ_WAVE_CYCLES."""
    for f in glob.glob(os.path.join(OUT, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                agg[k]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                agg[k]["n"] += 1
        print("== per kernel: launches, total ms | of SQ_WAVE_CYCLES: WAIT_ANY (parked: s_waitcnt / barrier), WAIT_INST_ANY (issue-stalled), "
              "of which WAIT_INST_LDS, ACTIVE_INST_ANY (issuing) | LDS: bank-conflict cycles / LDS-active cycles")
        for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("ns", 0))[:24]:
            wc = d.get("SQ_WAVE_CYCLES", 0)
            if wc <= 0:
                continue
            print(f"{k:58s} {int(d['n']):5d} {d['ns'] / 1e6:9.2f} ms | parked {d.get('SQ_WAIT_ANY', 0) / wc:5.3f}  issue-stall "
                  f"{d.get('SQ_WAIT_INST_ANY', 0) / wc:5.3f} (lds {d.get('SQ_WAIT_INST_LDS', 0) / wc:5.3f})  issuing "
                  f"{d.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.3f} | lds conflict {d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_LDS_IDX_ACTIVE', 0), 1):5.3f}"
                  f"  lds-active/busy {d.get('SQ_LDS_IDX_ACTIVE', 0) / max(d.get('SQ_BUSY_CYCLES', 0), 1):6.3f}")


def l2():
    """L2 (TCC) and L1 (TCP) request counters per kernel: hit rate, bytes requested by the CUs' L1s (64-byte requests), per launch."""
    for d in ("pmc_l2a", "pmc_l2b"):
        for f in glob.glob(os.path.join(OUT, d, "**", "*counter_collection.csv"), recursive=True):
            agg = defaultdict(lambda: defaultdict(float))
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                agg[k]["_n_" + r["Counter_Name"]] += 1
                if "Start_Timestamp" in r:
                    agg[k]["_ns_" + r["Counter_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            print(f"== {os.path.relpath(f, ROOT)}: per kernel, per launch averages")
            names = sorted({c for d2 in agg.values() for c in d2 if not c.startswith("_")})
            for k, dd in sorted(agg.items(), key=lambda kv: -max([v for c, v in kv[1].items() if c.startswith("_ns_")] or [0]))[:12]:
                parts = []
                for c in names:
                    n = dd.get("_n_" + c, 0)
                    if n:
                        parts.append(f"{c} {dd[c] / n:14.1f}")
                n0 = max([dd.get("_n_" + c, 0) for c in names] or [1])
                ns = max([dd.get("_ns_" + c, 0) for c in names] or [0])
                extra = ""
                if dd.get("TCC_HIT_sum") or dd.get("TCC_MISS_sum"):
                    extra += f" | L2 hit rate {dd.get('TCC_HIT_sum', 0) / max(dd.get('TCC_HIT_sum', 0) + dd.get('TCC_MISS_sum', 0), 1):.3f}"
                if dd.get("TCP_TCC_READ_REQ_sum") and ns:
                    extra += f" | L1->L2 read requests x 64 B / time = {dd['TCP_TCC_READ_REQ_sum'] * 64 / ns:.1f} GB/s"
                print(f"{k:52s} {int(n0):5d} launches, {ns / max(n0, 1) / 1e3:8.1f} us avg | " + "  ".join(parts) + extra)


def family_traffic(fetch_dir, write_dir, dom_kernel_name):
    """HBM-side bytes per launch of the dominant 3x3 family from two PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE output
    directories): (2 x FETCH + WRITE) / launches -- `bench.py --pmc-traffic` measures `roofline.traffic` with this in the run it
    describes.  The x2 on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md (HBM section)."""
    if dom_kernel_name.startswith("conv_wino2h_kernel"):
        fam = lambda k: k.startswith("conv_wino2h_kernel")
    elif dom_kernel_name.startswith("conv_wino3_kernel"):
        fam = lambda k: k.startswith("conv_wino3_kernel") or k.startswith("conv_wino3p_kernel")
    else:
        fam = lambda k: k.startswith("conv_wino_kernel") or k.startswith("wino_ksplit_reduce")
    tot, n = {}, {}
    for key, d, counter in (("fetch", fetch_dir, "FETCH_SIZE"), ("write", write_dir, "WRITE_SIZE")):
        tot[key], n[key] = 0.0, 0
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == counter and fam(short(r["Kernel_Name"])):
                    tot[key] += float(r["Counter_Value"]) * 1024.0           # rocprofv3 reports KiB
                    n[key] += 1
    if not n["fetch"] or not n["write"]:
        return None
    return dict(traffic_bytes_per_launch=2.0 * tot["fetch"] / n["fetch"] + tot["write"] / n["write"],
                fetch_bytes_per_launch_corrected=2.0 * tot["fetch"] / n["fetch"], write_bytes_per_launch=tot["write"] / n["write"],
                launches_fetch_pass=n["fetch"], launches_write_pass=n["write"])


def traffic(out_json):
    """HBM-side bytes per conv launch of the Winograd kernel family from the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
    kernel-trace only), over the SAME population bench.py's `algorithmic_bytes_per_launch` averages: every Winograd-served 3x3 conv
    op of the forwards of the run (a K-split op = conv_wino_kernel + its reduce pass), no autotune launches (the runs load the
    kernel table from the tune cache).  Forwards are counted by temb_mlp_kernel dispatches.  Also a per-instantiation table."""
    import json
    # which 3x3 family dominates is the bench line's call (bench.py: roofline.kernel); the population follows it
    dom = ""
    try:
        dom = json.load(open(os.path.join(OUT, "pmc_fetch.json")))["roofline"]["kernel"]
    except Exception:
        pass
    if dom.startswith("conv_wino2h_kernel"):
        res = {"family": "wino_f16x2", "kernel_family": "conv_wino2h_kernel<*> (the reduce passes of its 3 K-split ops per forward share a kernel "
               "with the fp32 Winograd K-split ops and are left out: 3 x 8 us of 41 ops)", "per_instantiation": {}}
        fam = lambda k: k.startswith("conv_wino2h_kernel")
    elif dom.startswith("conv_wino3_kernel"):
        res = {"family": "wino_bf16x3", "kernel_family": "conv_wino3_kernel<*> + conv_wino3p_kernel<*> (the same kernel as persistent workgroups)", "per_instantiation": {}}
        fam = lambda k: k.startswith("conv_wino3_kernel") or k.startswith("conv_wino3p_kernel")
    else:
        res = {"family": "wino_f32", "kernel_family": "conv_wino_kernel<*> + wino_ksplit_reduce_kernel", "per_instantiation": {}}
        fam = lambda k: k.startswith("conv_wino_kernel") or k.startswith("wino_ksplit_reduce")
    tot = {}
    for key, dirname, counter in (("fetch", "pmc_fetch", "FETCH_SIZE"), ("write", "pmc_write", "WRITE_SIZE")):
        per = defaultdict(lambda: [0, 0.0])
        fwd = 0
        for f in glob.glob(os.path.join(OUT, dirname, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != counter:
                    continue
                k = short(r["Kernel_Name"])
                if k.startswith("temb_mlp_kernel"):
                    fwd += 1
                if fam(k):
                    per[k][0] += 1
                    per[k][1] += float(r["Counter_Value"])
        res[f"forwards_{key}"] = fwd
        tot[key] = sum(v for _, v in per.values()) * 1024.0                      # rocprofv3 reports KiB
        res[f"family_launches_{key}"] = sum(c for c, _ in per.values())
        for k, (c, v) in per.items():
            e = res["per_instantiation"].setdefault(k, {})
            e[f"launches_{key}"] = c
            e[f"{key}_kib_per_launch_raw"] = v / max(c, 1)
    ops_per_forward = None
    try:
        ops_per_forward = json.load(open(os.path.join(OUT, "pmc_fetch.json")))["roofline"]["launches"]
    except Exception:
        pass
    res["wino_ops_per_forward"] = ops_per_forward
    n_ops = (ops_per_forward or 0) * res.get("forwards_fetch", 0)
    if res["family"] != "wino_f32":
        n_ops = res.get("family_launches_fetch", 0)             # one launch per op in these families
    res["fetch_correction"] = "x2 (gfx950 FETCH_SIZE reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section)"
    if n_ops:
        res["traffic_bytes_per_launch"] = (2.0 * tot["fetch"] + tot["write"]) / n_ops
        res["fetch_bytes_per_launch_corrected"] = 2.0 * tot["fetch"] / n_ops
        res["write_bytes_per_launch"] = tot["write"] / n_ops
        try:
            res["algorithmic_bytes_per_launch"] = json.load(open(os.path.join(OUT, "pmc_fetch.json")))["roofline"]["algorithmic_bytes_per_launch"]
        except Exception:
            pass
    res["note"] = ("bench.py --subsample 5 (6 forwards), --graph 0, kernel table from the tune cache (no autotune launches); one PMC counter "
                   "per run; Infinity-Cache hits are counted as traffic (memory-side L2 counters)")
    json.dump(res, open(out_json, "w"), indent=1)
    print("wrote", out_json, {k: v for k, v in res.items() if k != "per_instantiation"})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sq":
        sq()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "traffic":
        traffic(sys.argv[2])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "l2":
        l2()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mfma":
        mfma()
        sys.exit(0)
    stats()
    pmc("pmc_fetch", "FETCH_SIZE")
    pmc("pmc_write", "WRITE_SIZE")
