R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/post; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for v in trained worst; do
  timeout -k 5 90 rocprofv3 --kernel-trace --stats -d /tmp/pp_$v -o p -- python3 $R/tools/bench_post.py ssd_300_vgg16_voc 64 $v 20 > $O/post_b64_$v.log 2>&1
  python3 $R/tools/rocpd_stats.py /tmp/pp_$v/p_results.db 10 > $O/post_b64_${v}_kernel_stats.md 2>&1
  rm -rf /tmp/pp_$v
done
cat $O/post_b64_trained_kernel_stats.md
