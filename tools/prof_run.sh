#!/bin/bash
# rocprofv3 passes over one configuration; run ON the GPU box from the repo root:
#   tools/prof_run.sh <label> <prof_config.py arguments...>
# kernel-trace/stats pass, then separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ sets), each with
# --kernel-trace only; summaries land in gpurun_out/prof_<label>/summary.json
label=$1; shift
driver=${PROF_DRIVER:-tools/prof_config.py}      # e.g. PROF_DRIVER=tools/prof_sliding.py
out=$PWD/gpurun_out/prof_$label
mkdir -p $out
export PYTHONPATH=$PWD
repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 $repo/$driver "$@" > $out/stats.log 2>&1
i=0
# PROF_SETS=short: only the HBM-byte and LDS passes (the per-point passes of the LDS-tile sweep)
if [ "$PROF_SETS" = "short" ]; then
  sets=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_LDS")
else
  sets=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
        "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE")
fi
for set in "${sets[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -o p -- python3 $repo/$driver "$@" > $out/pmc$i.log 2>&1
done
cd $repo
python3 tools/prof_summarize.py $out > $out/summary.json
python3 - <<PY
import json
d=json.load(open('$out/summary.json'))
for k,v in d['kernels'].items():
    if 'avg_us' in v and v.get('avg_us',0)>50: print('$label', k, v.get('calls'), v.get('avg_us'), {c: v['counters'][c] for c in ('FETCH_SIZE','WRITE_SIZE','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE') if c in v.get('counters',{})})
PY
