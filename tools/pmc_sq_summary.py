#!/usr/bin/env python
"""Condense a rocprofv3 --pmc pass with the SQ counters
    SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
    SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
into the per-dispatch table kept as profiles/rNN_pmc_sq.json (matrix-pipe utilisation and where the waves' cycles go).
Two utilisation figures: `mfma_util` divides the matrix pipe's busy cycles by ALL SIMD-cycles of the dispatch as
GRBM_GUI_ACTIVE sees it (idle CUs, tile quantisation and the profiler's own per-dispatch overhead included -- the
round-1 definition); `mfma_busy_while_resident` divides by the SIMD-cycles during which the kernel's waves were
resident (SQ_WAVE_CYCLES counts in units of 4 cycles per wave; waves per SIMD from the workgroup size and the
workgroups a CU holds), i.e. what the instruction stream itself achieves.
usage: pmc_sq_summary.py <run_dir> "<comment: the command that was profiled>" > profiles/rNN_pmc_sq.json"""
import csv
import glob
import json
import sys
from collections import defaultdict

N_XCD, N_SIMD = 8, 1024


def main(run_dir, comment):
    files = glob.glob(f"{run_dir}/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {run_dir}")
    disp = defaultdict(lambda: defaultdict(float))
    names, wg = {}, {}
    for f in files:
        for r in csv.DictReader(open(f)):
            d = int(r["Dispatch_Id"])
            names[d] = r.get("Kernel_Name", "")
            wg[d] = int(r.get("Workgroup_Size", 0) or 0)
            disp[d][r["Counter_Name"]] += float(r["Counter_Value"])
    out = []
    for d in sorted(disp):
        c = disp[d]
        name = names[d]
        if not any(k in name for k in ("gemm", "attention", "conv_halo", "conv_igemm")):
            continue
        if "GRBM_GUI_ACTIVE" not in c or not c.get("SQ_WAVE_CYCLES"):
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / N_XCD
        wave = c["SQ_WAVE_CYCLES"]
        lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        wgs_per_cu = 2 if ("gemm_bf16_kernel" in short or short.startswith("attention_kernel") or "conv_igemm" in short) else 1
        waves_per_simd = max(1.0, wg[d] / 64 / 4 * wgs_per_cu)
        out.append({"dispatch": d, "kernel": short[:48],
                    "kcycles_per_xcd": round(cyc / 1e3, 1),
                    "mfma_util": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * N_SIMD), 3),
                    "mfma_busy_while_resident": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (wave * 4 / waves_per_simd), 3),
                    "lds_conflict_frac": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / lds, 3) if lds else None,
                    "wave_parked_frac": round(c.get("SQ_WAIT_ANY", 0.0) / wave, 3),
                    "issue_stall_frac": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 3),
                    "issuing_frac": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 3)})
    json.dump({"_comment": comment, "dispatches": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
