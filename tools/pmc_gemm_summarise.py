#!/usr/bin/env python3
"""rocprofv3 counter CSVs of tools/pmc_gemm.py -> profiles/r02_pmc_summary.json: per kernel of one round-2 minibatch update
(B = 262,144) the HBM bytes per launch and per row (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md section HBM; counters
in KB), the algorithmic figure beside it, and from the SQ / GRBM pass the MFMA busy share and the effective clock.

    python tools/pmc_gemm_summarise.py <fetch.csv> <write.csv> <sq.csv> <sq_kernel_trace.csv> <out.json> [<provenance.json>]

<provenance.json> is what tools/pmc_gemm.py wrote on the GPU box next to the counter CSVs (sha256 of the libgymrl_hip.so it
loaded); the git head is added here.  bench.py attaches a summary only if that sha256 equals the library it has loaded."""
import csv
import json
import os
import subprocess
import sys
from collections import defaultdict

B = 262144
KERNELS = {  # name fragment -> (label, algorithmic HBM bytes per row, flops per row)
    "gemm_ws_kernel<256, 4, 2, false, 1, 256": ("gemm_fwd_256_tanh", 2048.0, 131072.0),
    "gemm_ws_kernel<256, 4, 2, false, 0, 512": ("gemm_fwd_512", 3072.0, 262144.0),
    "gemm_tn_kernel<8, 512": ("gemm_dw_512", 3072.0, 262144.0),
    "gemm_ws_kernel<512, 2, 2, true, 2, 256": ("gemm_dx_512_tanhbwd", 4096.0, 262144.0),
    "gemm_tn_kernel<8, 256": ("gemm_dw_256_db", 2048.0, 131072.0),
    "gemm_ws_kernel<256, 4, 2, true, 2, 256": ("gemm_dx_256_tanhbwd", 3072.0, 131072.0),
    "heads_loss_kernel": ("heads_loss_fwd_bwd", 4112.0, 0.0),
    "linear_tanh_smallk_kernel": ("linear_tanh_smallk", 1056.0, 0.0),
    "linear_smallk_bwd_kernel": ("linear_smallk_bwd", 1056.0, 0.0),
    "Cijk_": ("library mm 256 (reference point)", 2048.0, 131072.0),
}


def per_kernel(path, counters):
    out = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] not in counters:
            continue
        for frag, (label, _, _) in KERNELS.items():
            if frag in r["Kernel_Name"]:
                out[label][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return out


def mean_tail(v):      # mean over launches 2..n (the first launch pays lazy allocations)
    v = [x for _, x in sorted(v)]
    return sum(v[1:]) / max(1, len(v) - 1)


def main():
    fetch, write, sq, trace, out = sys.argv[1:6]
    rd, wr = per_kernel(fetch, {"FETCH_SIZE"}), per_kernel(write, {"WRITE_SIZE"})
    sqc = per_kernel(sq, {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"})
    dur = defaultdict(list)
    for r in csv.DictReader(open(trace)):
        for frag, (label, _, _) in KERNELS.items():
            if frag in r["Kernel_Name"]:
                dur[label].append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3))
    res = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES "
                     "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU in three separate passes (--kernel-trace only) over tools/pmc_gemm.py; FETCH_SIZE "
                     "and WRITE_SIZE are KB, FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM; a 512 MiB memset between updates "
                     "evicts the Infinity Cache; mean over launches 2..n; GRBM_GUI_ACTIVE sums the 8 XCDs", "rows": B}
    prov = json.load(open(sys.argv[6])) if len(sys.argv) > 6 else {}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        prov["git_head"] = subprocess.check_output(["git", "-C", root, "rev-parse", "HEAD"], text=True).strip()
        prov["git_dirty"] = bool(subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "gymrl_amd/csrc", "include"], text=True).strip())
    except Exception:
        pass
    res["provenance"] = prov
    for frag, (label, alg, flops) in KERNELS.items():
        if label not in rd:
            continue
        hbm = 2.0 * mean_tail(rd[label]["FETCH_SIZE"]) * 1024.0 + mean_tail(wr[label]["WRITE_SIZE"]) * 1024.0
        ent = {"hbm_bytes_per_launch": round(hbm), "hbm_bytes_per_row": round(hbm / B, 1), "algorithmic_bytes_per_row": alg}
        if label in sqc and label in dur:
            d = mean_tail(dur[label])
            gui = mean_tail(sqc[label]["GRBM_GUI_ACTIVE"]) / 8.0
            ent["duration_us_profiled"] = round(d, 1)
            ent["effective_clock_GHz"] = round(gui / d / 1e3, 3)
            if flops:
                busy = mean_tail(sqc[label]["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024.0      # per SIMD
                ent["mfma_busy_share"] = round(busy / gui, 3)
        res[label] = ent
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
