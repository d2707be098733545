# stream-K for the generic forward convolutions: minimum range sweep on the SSD-300 / SSD-512 tail layers (batch 32 / 16)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
for mr in off 8 16 24 32 48 96; do
  echo "== SSDK_SK_MINRANGE=$mr"
  if [ $mr = off ]; then export SSDK_CONV_NO_STREAMK_GENERIC=1; else unset SSDK_CONV_NO_STREAMK_GENERIC; export SSDK_SK_MINRANGE=$mr; fi
  timeout -k 10 200 python3 $R/tools/conv_decomp_sweep.py 32 ssd300 nosweep 2>&1 | grep -v amdgpu | cut -c1-110
done > $O/sk_sweep.txt 2>&1
cat $O/sk_sweep.txt
