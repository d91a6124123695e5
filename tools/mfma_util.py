"""MFMA utilisation, stall and HBM-traffic figures of the kernels with the most time in one train step.

    python tools/mfma_util.py <kernel-trace dir> <order.json> <pmc dir>...

Durations: rocprofv3 --kernel-trace of the HIP-graph replay (bench.py --one-stream), last 3 replayed steps.  Counters: rocprofv3
--pmc passes over an eager step of the same build (CSTS_GROUP_WGRADS=1, so that the grouped weight-gradient launches exist),
summed over a kernel's launches of that step.  Derived:
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration in the UNPROFILED graph replay x 2.4 GHz x 1024 SIMDs)
                (the counter advances 32 per v_mfma_f32_32x32x16_bf16 on the issuing SIMD and is summed over the chip; a --pmc pass
                 stretches short launches 2 x and more, so its own GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES are no time base for them; 2.4 GHz
                 is the maximum clock: the chip holds less under load, so this is a LOWER bound of the matrix pipes' busy share)
  mfma_busy_pmc = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024): the same against the profiled launch's own cycles
                (VERDICT's SQ_BUSY_CYCLES form: SQ_BUSY_CYCLES reads ~3.3 x GRBM_GUI_ACTIVE / 8 here -- it is summed over
                 shader engines, not a per-CU cycle count -- and is printed raw below)
  wait_inst_any / wait_inst_lds / wait_any = share of SQ_WAVE_CYCLES (quad-cycles, per wave)
  hbm bytes   = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction of the guide), per step
  algorithmic bytes / flop (GEMM kernels only): from the ordered launch list of the instrumented pass (A + B + C + epilogue operands)"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*$", "", n).replace(" ", "")


def main():
    tdir, order_path, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    f = glob.glob(tdir + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    steps, cur = [], []
    for r in rows:
        cur.append(r)
        if "opt_adamw_kernel" in r["Kernel_Name"]:
            steps.append(cur)
            cur = []
    use = steps[-3:]
    t = collections.defaultdict(lambda: [0, 0.0])
    for st in use:
        for r in st:
            k = short(r["Kernel_Name"])
            t[k][0] += 1
            t[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    step_ms = sum((int(st[-1]["End_Timestamp"]) - int(st[0]["Start_Timestamp"])) for st in use) / len(use) / 1e6
    pmc = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in pmc_dirs:
        for cf in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rr = list(csv.DictReader(open(cf)))
            # mean per launch over every eager launch of the pass; scaled to one step with the trace's launch count below
            per = collections.defaultdict(list)
            for r in rr:
                per[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in per.items():
                pmc[k][c] += sum(v) / len(v)
    order = json.load(open(order_path))
    alg = collections.defaultdict(lambda: [0.0, 0.0])
    for o in order:
        k = o["kernel"].replace(" ", "")
        alg[k][0] += o["flop"]
        alg[k][1] += o["bytes"] + o["epilogue_bytes"]
    top = sorted(t.items(), key=lambda kv: -kv[1][1])[:12]
    print(f"one train step (b = 4, 16 x 256^2, bf16 mode), HIP-graph replay on one stream: {step_ms:.2f} ms; kernels by time per step")
    print("kernel | launches/step | ms/step | mfma_busy | mfma_busy_pmc | wait_inst_any | wait_inst_lds | wait_any | LDS bank-conflict share | "
          "HBM GB/step (PMC) | algorithmic GB/step | TFLOP/s | HBM GB/s (PMC bytes / time)")
    for k, (n, us) in top:
        n1, ms = n / len(use), us / len(use) / 1e3
        c = pmc.get(k, {})
        gui, mf, sqb, wc = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CYCLES", 0.0), c.get("SQ_WAVE_CYCLES", 0.0)
        # counters are means per launch: per step = x launches per step; time base = the kernel's time in the replayed step
        busy = mf * n1 / (ms * 1e-3 * 2.4e9 * 1024)
        busy_sq = mf / (gui / 8 * 1024) if gui else float("nan")
        w = lambda name: (c.get(name, 0.0) / wc) if wc else float("nan")
        hbm = (2 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024 / 1e9 * n1
        lds_conf = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_INSTS_LDS", 0.0), 1.0)
        fl, by = alg.get(k, (0.0, 0.0))
        print(f"{k} | {n1:.0f} | {ms:.3f} | {busy:.3f} | {busy_sq:.3f} | {w('SQ_WAIT_INST_ANY'):.3f} | {w('SQ_WAIT_INST_LDS'):.3f} | {w('SQ_WAIT_ANY'):.3f} | "
              f"{lds_conf:.3f} | {hbm:.3f} | {by / 1e9 if by else float('nan'):.3f} | {fl / ms / 1e9 if fl else float('nan'):.1f} | {hbm / ms * 1e3:.0f}")
    # ---- kernel FAMILIES against the algorithmic work of the C-ABI entries that launch them (bench.json: roofline.per_entry,
    # tools/work_model.py: every operand once): the roofline fraction of every family can be recomputed from these columns
    bj = os.path.join(os.path.dirname(order_path), "bench.json")
    per_entry = {}
    if os.path.exists(bj):
        try:
            per_entry = json.load(open(bj)).get("roofline", {}).get("per_entry", {})
        except Exception:
            per_entry = {}
    FAM = [("LayerNorm forward", r"^ln_fwd", ["layernorm_fwd", "layernorm_fwd_add"]),
           ("LayerNorm backward", r"^ln_bwd", ["layernorm_bwd_ex", "layernorm_bwd", "layernorm_bwd2"]),
           ("conv-pool + LN(hd) forward", r"^pool_ln_fwd", ["pool_ln_fwd"]),
           ("stencil weight gradients", r"^dwconv_wgrad", ["dwconv_wgrad", "dwconv_wgrad2", "dwconv_wgrad_grouped"]),
           ("transposed / strided stencils", r"^dwconv_(transposed|strided)", ["dwconv_transposed", "dwconv_transposed2", "dwconv_strided"]),
           ("attention forward", r"^attn_fwd", ["attn_fwd"]),
           ("attention backward", r"^attn_(dq|dkv|bwd_fused)", ["attn_bwd"]),
           ("grouped weight gradients 192x384", r"^wgrad8", ["wgrad_grouped8"]),
           ("grouped weight gradients 128/256x128", r"^wgrad_grouped", ["wgrad_grouped"]),
           ("grouped weight gradients 96x96 per wave", r"^wgrad5", ["wgrad_grouped5"]),
           ("clip + AdamW", r"^(opt_|factored_)", ["adamw_step", "adamw_factored", "factored_sqnorm"]),
           ("max-pool skip", r"^maxpool", ["maxpool_fwd", "maxpool_bwd"]),
           ("trilinear", r"^trilinear", ["trilinear_fwd", "trilinear_bwd"]),
           ("all GEMM kernels", r"^(gemm|splitk_finish)", ["gemm"])]
    if per_entry:
        print("\nkernel families: time in the replayed step against the ALGORITHMIC bytes / flop of the entries that launch them")
        print("family | launches/step | ms/step | algorithmic GB/step | GFLOP/step | hbm_frac (bytes / time / 8 TB/s) | mfma_frac (flop / time / 2.5 PF) | PMC GB/step | PMC / algorithmic")
        for name, rx, entries in FAM:
            ks = [k for k in t if re.search(rx, k)]
            if not ks:
                continue
            n1 = sum(t[k][0] for k in ks) / len(use)
            ms = sum(t[k][1] for k in ks) / len(use) / 1e3
            by = sum(per_entry.get(e, {}).get("algorithmic_bytes", 0) for e in entries)
            fl = sum(per_entry.get(e, {}).get("algorithmic_flop", 0) for e in entries)
            hbm = sum((2 * pmc.get(k, {}).get("FETCH_SIZE", 0.0) + pmc.get(k, {}).get("WRITE_SIZE", 0.0)) * 1024 / 1e9 * t[k][0] / len(use) for k in ks)
            print(f"{name} | {n1:.0f} | {ms:.3f} | {by / 1e9:.3f} | {fl / 1e9:.1f} | {by / (ms * 1e-3) / 8e12 if ms else 0:.3f} | {fl / (ms * 1e-3) / 2.5e15 if ms else 0:.3f} | "
                  f"{hbm:.3f} | {hbm / (by / 1e9) if by else float('nan'):.2f}")
    print("\nraw counters (mean per launch over the eager launches of the --pmc passes):")
    for k, _ in top:
        print(k, {c: f"{v:.4g}" for c, v in sorted(pmc.get(k, {}).items())})


main()
