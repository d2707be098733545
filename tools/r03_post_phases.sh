# select-pass phases of the postprocess at batch 64 (SSDK_POST_STOP: 1 = staging only, 2 = + row pass, 3 = + survivors without the key store, 0 = all)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for v in trained worst; do
  for st in 0 1 2 3; do
    SSDK_POST_STOP=$st timeout -k 5 90 rocprofv3 --kernel-trace --stats -d /tmp/pp_${v}_$st -o p -- python3 $R/tools/bench_post.py ssd_300_vgg16_voc 64 $v 20 > $O/post_b64_${v}_stop$st.log 2>&1
    echo "== $v stop=$st" >> $O/post_phases.md
    python3 $R/tools/rocpd_stats.py /tmp/pp_${v}_$st/p_results.db 9 >> $O/post_phases.md 2>&1
    rm -rf /tmp/pp_${v}_$st
  done
done
cat $O/post_phases.md
