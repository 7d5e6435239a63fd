mkdir -p gpurun_out/pmc_n10
G="SQ_BUSY_CYCLES SQ_WAVE_CYCLES,SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY,SQ_INSTS_VALU SQ_ACTIVE_INST_VALU,GRBM_GUI_ACTIVE SQ_WAIT_ANY,SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT,SQ_INSTS_SALU SQ_INSTS_VMEM_WR,SQ_WAVES SQ_INSTS_VMEM_RD"
python tools/pmc.py "env_kernel<10" gpurun_out/pmc_n10/r03_b_pmc_issue_n10_w8192_k32.json "$G" -- python tools/kbench.py --worlds 8192 --agents 10 --spl 32 > /dev/null 2> gpurun_out/pmc_n10/err.txt
python tools/pmc.py "env_kernel<10" gpurun_out/pmc_n10/r03_b_pmc_issue_n10_w262144_k1.json "$G" -- python tools/kbench.py --worlds 262144 --agents 10 --spl 1 > /dev/null 2>> gpurun_out/pmc_n10/err.txt
python - <<'PY'
import json
for f in ("gpurun_out/pmc_n10/r03_b_pmc_issue_n10_w8192_k32.json", "gpurun_out/pmc_n10/r03_b_pmc_issue_n10_w262144_k1.json"):
    d = json.load(open(f)); print(f, {k: round(v["mean_per_dispatch"]) for k, v in d["counters"].items()})
PY
