#!/bin/bash
# tools/exp_run.sh -- on the GPU box: headline rate of every pirip_amd/lib_exp/N variant (5 s each)
for d in pirip_amd/lib_exp/*/; do
  n=$(basename $d)
  echo -n "EXP $n: "
  PIRIP_HIP_LIB=$PWD/$d/libpirip_hip.so python tools/power_loop.py ${1:-4} 2>&1 | tail -1
done
