cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf /tmp/pp_$name; timeout -k 5 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pp_$name -- python /root/repo/tools/pmc_sb.py > /dev/null 2>&1; f=$(find /tmp/pp_$name -name '*counter_collection.csv' | head -1); python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "gemm_sb_ws" in k or "gemm_ws_kernel" in k:
        agg["sb" if "gemm_sb" in k else "exact"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS
run c FETCH_SIZE
run d WRITE_SIZE
run e TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run f GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
