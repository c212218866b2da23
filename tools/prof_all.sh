#!/bin/bash
# every profile committed under profiles/ for this round; run ON the GPU box from the repo root
set -x
R=${1:-r02}
tools/prof_run.sh ${R}_A_stream2 A 20000 0 3
tools/prof_run.sh ${R}_A_rows_full A 20000 3 3
tools/prof_run.sh ${R}_B_headline B 100000 0 3
tools/prof_run.sh ${R}_C_stream C 10000 0 3
# round 3: the ragged-corpus scan and the file batch (PROF_DRIVER selects the program behind the passes)
if [ "$R" != "r02" ]; then
  PROF_DRIVER=tools/prof_sliding.py tools/prof_run.sh ${R}_sliding_q21 1000000 21 20 70 5
  PROF_DRIVER=tools/prof_sliding.py PROF_SETS=short tools/prof_run.sh ${R}_sliding_q5 1000000 5 20 70 5
  PROF_DRIVER=tools/exp/files_time.py PROF_SETS=short tools/prof_run.sh ${R}_files 1 10
  if [ "$R" != "r03" ]; then      # round 4: the long query and the 6000-file call
    PROF_DRIVER=tools/prof_sliding.py PROF_SETS=short tools/prof_run.sh ${R}_sliding_q48 1000000 48 20 70 5
    PROF_DRIVER=tools/exp/files_time.py PROF_SETS=short tools/prof_run.sh ${R}_files6000 100 2
    if [ "$R" != "r04" ]; then    # round 5: eight queries of one length in one call
      PROF_DRIVER=tools/prof_sliding_batch.py PROF_SETS=short tools/prof_run.sh ${R}_sliding_batch8_q21 1000000 21 8 5
      PROF_DRIVER=tools/prof_sliding_batch.py PROF_SETS=short tools/prof_run.sh ${R}_sliding_batch8_q5 1000000 5 8 5
    fi
  fi
  # round 6: SWEEP=1 repeats round 2's LDS-tile sizing sweep of configs[4] on the generic kernel as it is now
  if [ "$SWEEP" != "1" ]; then exit 0; fi
fi
# LDS-tile sizing sweep of BASELINE configs[4] on the generic kernel: waves per workgroup x twiddle cache
for w in 1 2 4 6 7; do for c in 1 0; do
  PROF_SETS=short tools/prof_run.sh ${R}_C_generic_w${w}_c${c} C 10000 1 2 $w $c
done; done
