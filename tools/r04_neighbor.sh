#!/bin/bash
N=tools/clk/neighbor
for vk in 0 1; do
echo "######## victim kind $vk (0 = MFMA + VALU, 1 = VALU)"
echo "== 1 aggressor, v_fma loop, no LDS"; $N 1 0 0 $vk
echo "== 1 aggressor, v_fma loop, whole LDS"; $N 1 1 0 $vk
echo "== 8 aggressors, v_fma loop, no LDS"; $N 8 0 0 $vk
echo "== 8 aggressors, v_fma loop, whole LDS"; $N 8 1 0 $vk
echo "== 1 aggressor, quarter-rate loop, whole LDS"; $N 1 1 1 $vk
echo "== 8 aggressors, quarter-rate loop, whole LDS"; $N 8 1 1 $vk
done
echo "######## victim 224 workgroups (one per CU with room to spare)"
$N 8 1 0 0 224
$N 8 0 0 0 224
