#!/bin/bash
# every profile committed under profiles/ for this round; run ON the GPU box from the repo root
set -x
R=${1:-r02}
tools/prof_run.sh ${R}_A_stream2 A 20000 0 3
tools/prof_run.sh ${R}_A_rows_full A 20000 3 3
tools/prof_run.sh ${R}_B_headline B 100000 0 3
tools/prof_run.sh ${R}_C_stream C 10000 0 3
# LDS-tile sizing sweep of BASELINE configs[4] on the generic kernel: waves per workgroup x twiddle cache
for w in 1 2 4 6 7; do for c in 1 0; do
  PROF_SETS=short tools/prof_run.sh ${R}_C_generic_w${w}_c${c} C 10000 1 2 $w $c
done; done
