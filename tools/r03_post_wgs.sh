# post_select2: persistent workgroups per call (SSDK_POST_WGS), trained-like and worst case at batch 64
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -f $O/post_wgs.md
for v in trained worst; do
  for w in ${WGS:-768 1024 1280 1536 2048}; do
    SSDK_POST_WGS=$w timeout -k 5 90 rocprofv3 --kernel-trace --stats -d /tmp/pw_${v}_$w -o p -- python3 $R/tools/bench_post.py ${CFG:-ssd_300_vgg16_voc} ${B:-64} $v 20 > $O/post_wgs_${v}_$w.log 2>&1
    echo "== $v wgs=$w" >> $O/post_wgs.md
    python3 $R/tools/rocpd_stats.py /tmp/pw_${v}_$w/p_results.db 7 | grep "post_\|total" >> $O/post_wgs.md 2>&1
    rm -rf /tmp/pw_${v}_$w
  done
done
cat $O/post_wgs.md
