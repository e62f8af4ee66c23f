#!/usr/bin/env python3
"""Turn the two rocprofv3 counter_collection CSVs of tools/pmc_kernels.py (one --pmc FETCH_SIZE run, one
--pmc WRITE_SIZE run) into profiles/r01_pmc_summary.json: HBM bytes per launch and per work unit for
every hand-written HBM-bound kernel.  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled on
gfx950 (MI355X_MICROARCH.md, section HBM: wide coalesced reads are tallied at half their bytes).

    python tools/pmc_summarise.py <fetch.csv> <write.csv> <out.json> [<rollout_sq.csv> <rollout_trace.csv>]

The optional pair is the SQ-counter pass over the persistent rollout kernel (profiles/README.md): it adds
`rollout_lunar` = wave-instruction counts, duration and the fraction of the chip's VALU issue slots used
(a wave64 VALU instruction holds its SIMD for 4 cycles; 1024 SIMDs).
"""
import csv
import json
import sys
from collections import defaultdict

T, N, MB = 2048, 4096, 262144
UNITS = {   # kernel-name fragment -> (label, work units per launch, algorithmic bytes per unit)
    "gae_blk_carry": ("gae", T * N, None), "gae_blk_apply": ("gae", T * N, None), "moments_finalize": ("gae", T * N, None),
    "gae_blk_aggregate": ("gae_aggregate_pass(variant 1 only)", T * N, 9.0),
    "ppo_loss_kernel": ("ppo_loss", None, 56.0),
    "linear_tanh_smallk_kernel": ("linear_tanh_smallk", MB, 4.0 * (8 + 256)),
    "tanh_inplace_kernel": ("tanh_inplace", MB * 512, 8.0),
    "heads_fwd_tanh_kernel": ("heads_fwd_tanh", MB, 8.0 * 256 + 20.0),
    "heads_bwd_kernel": ("heads_bwd", MB, 16.0 * 256 + 20.0),
    "tanh_bwd_colsum_kernel": ("tanh_bwd_colsum", MB, 12.0 * 256),
    "linear_smallk_bwd_kernel": ("linear_smallk_bwd", MB, 8.0 * 256 + 32.0),
}


def collect(path, counter):
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for frag in UNITS:
            if frag in r["Kernel_Name"]:
                # the loss kernel is grid-stride (same grid for every size): tools/pmc_kernels.py alternates
                # whole-rollout and minibatch launches, told apart by launch order
                key = (frag, len(per[(frag, 0)]) % 2) if frag == "ppo_loss_kernel" else (frag, 1)
                per[(frag, 0)].append(0.0)
                per[key + ("v",)].append(float(r["Counter_Value"]) * 1024.0)
    # mean over launches, first launch dropped (cold caches / lazy allocations)
    return {k[:2]: (sum(v[1:]) / (len(v) - 1) if len(v) > 1 else v[0]) for k, v in per.items() if len(k) == 3}


def main():
    fetch, write, out = sys.argv[1:4]
    rd, wr = collect(fetch, "FETCH_SIZE"), collect(write, "WRITE_SIZE")
    res = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (with --kernel-trace only) over "
                     "tools/pmc_kernels.py; counters are KB; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM; a 512 MiB "
                     "memset between launches evicts the 256 MiB Infinity Cache; mean over launches 2..n",
           "shape": {"T": T, "N": N, "minibatch": MB}}
    agg = defaultdict(lambda: [0.0, 0.0])
    for (frag, grid), b in rd.items():
        label, units, alg = UNITS[frag]
        if frag == "ppo_loss_kernel":
            label, units = ("ppo_loss" if grid == 0 else "ppo_loss_minibatch"), None
        agg[label][0] += 2.0 * b
        agg[label][1] += wr.get((frag, grid), 0.0)
    loss_units = {"ppo_loss": T * N, "ppo_loss_minibatch": MB}
    for label, (r, w) in agg.items():
        units = loss_units.get(label) or next(u for f, (l, u, a) in UNITS.items() if l == label and u)
        alg = 17.0 if label == "gae" else next((a for f, (l, u, a) in UNITS.items() if l.startswith(label.split("_mini")[0]) and a), None)
        res[label] = {"hbm_read_bytes": r, "hbm_write_bytes": w, "units_per_launch": units,
                      "hbm_bytes_per_unit": (r + w) / units, "algorithmic_bytes_per_unit": alg}
    # keys bench.py reads
    res["gae"]["hbm_bytes_per_transition"] = res["gae"]["hbm_bytes_per_unit"]
    res["ppo_loss"]["hbm_bytes_per_sample"] = res["ppo_loss"]["hbm_bytes_per_unit"]
    if len(sys.argv) >= 6:
        sq = defaultdict(list)
        for r in csv.DictReader(open(sys.argv[4])):
            sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
        tr = [r for r in csv.DictReader(open(sys.argv[5])) if "rollout_lunar" in r["Kernel_Name"]]
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 for r in tr]
        valu, cyc = sq["SQ_INSTS_VALU"][0], sq["GRBM_GUI_ACTIVE"][0] / 8.0          # GRBM_GUI_ACTIVE sums the 8 XCDs
        res["rollout_lunar"] = {"launch": "T = 512 vector steps x 4096 envs", "duration_s": dur[0], "cycles": cyc,
                                "wave_insts_valu": valu, "wave_insts_salu": sq["SQ_INSTS_SALU"][0],
                                "wave_insts_lds": sq["SQ_INSTS_LDS"][0],
                                "valu_issue_slots_used": valu * 4.0 / (1024.0 * cyc),
                                "bound": "ALU latency: one dependent chain per workgroup, ~7 cycles per instruction"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: (round(v["hbm_bytes_per_unit"], 2), v["algorithmic_bytes_per_unit"]) for k, v in res.items()
                      if isinstance(v, dict) and "hbm_bytes_per_unit" in v}, indent=1))


if __name__ == "__main__":
    main()
