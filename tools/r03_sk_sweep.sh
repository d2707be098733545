# stream-K for the generic forward convolutions: minimum range sweep on the SSD-300 / SSD-512 tail layers (batch 32 / 16).
# profiles/r03_sk_sweep.txt was taken while the library's switch was the opt-OUT SSDK_CONV_NO_STREAMK_GENERIC (commit bb12561); the library
# now reads the opt-IN SSDK_CONV_STREAMK_GENERIC (conv.hip streamk_would_take), which is what this script sets.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
for mr in off 8 16 24 32 48 96; do
  echo "== SSDK_SK_MINRANGE=$mr"
  if [ $mr = off ]; then unset SSDK_CONV_STREAMK_GENERIC; else export SSDK_CONV_STREAMK_GENERIC=1; export SSDK_SK_MINRANGE=$mr; fi
  timeout -k 10 200 python3 $R/tools/conv_decomp_sweep.py 32 ssd300 nosweep 2>&1 | grep -v amdgpu | cut -c1-110
done > $O/sk_sweep.txt 2>&1
cat $O/sk_sweep.txt
