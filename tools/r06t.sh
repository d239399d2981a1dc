#!/bin/bash
# round 6: PMC passes over six MEDIUM commits; per-kernel summaries of the kernels the commit is made of (what binds each: VALU issue, LDS, waiting)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06t; mkdir -p $O; rm -rf $O/*
tools/pmc_run.sh $O/pmc python $R/tests/gpu_build_only.py "" 6 > $O/pmc.log 2>&1
for K in small_build wide_plan wide_emit top_bin top_partition top_local; do
  python tools/pmc_summary.py $O/pmc "$K" $O/pmc_$K > $O/sum_$K.log 2>&1
  echo "== $K"; grep -a "Derived" $O/pmc_$K.md | cut -c1-600
  python - <<PY
import json
d=json.load(open("$O/pmc_$K.json")); c=d["counters"]
print({k: ("%.3g" % c[k]) for k in ("SQ_WAVES","SQ_INSTS_VALU","SQ_INSTS_LDS","SQ_INSTS_SALU","SQ_INSTS_VMEM_RD","SQ_BUSY_CYCLES","SQ_WAVE_CYCLES","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_ANY","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","GRBM_GUI_ACTIVE") if k in c})
PY
done
