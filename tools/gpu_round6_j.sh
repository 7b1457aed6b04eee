#!/bin/bash
# round 6, call J: what could ray ordering buy on the 12 M-triangle scene (the walk is HBM-bound there)?  rays sorted outside the timed region, unchanged walk
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_j; mkdir -p $O
cd $R
ORDER_SCENES=cubes12m ORDER_FIRST=0 ORDER_KINDS=none,cell3,cell4,cell5,cell6,cell7,oct_cell4,oct_cell5,oct_cell6,dir4c4,dir8c4 timeout 1200 python tools/order_potential.py 10 2>&1 | tee $O/order_potential_12m.log
