#!/bin/bash
# geometry sweep of the ABN kernels (exploration; run on the GPU box)
for who in ${WHO:-BAPP APPLY BRED STATS}; do
  for tb in ${TBS:-128 256 384 512 640}; do
    for rr in ${RRS:-0 4}; do
      printf "%s TB=%4d RR=%d : " $who $tb $rr
      env ABN_ONLY=$who UCD_TB_$who=$tb UCD_RR_$who=$rr python tools/abn_bench.py 24 2>&1 | tail -2 | grep -v amdgpu.ids | tr -d '\n'; echo
    done
  done
done
