#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 2 4 8 16; do for cfg in "1 12" "1 24" "1 48" "2 12" "2 24" "2 48" "4 16" "4 24" "4 48"; do set -- $cfg; echo "b=$b nq=$1 cap=$2: $(HN_SB_NQ=$1 HN_SB_SPLITS=$2 python tools/quick_cfg2.py $b 100 2>/dev/null| tail -1)"; done; done
