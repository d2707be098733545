# stage timings of the postprocess kernels (debug stops): bash tools/run_stops.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${CFG:-ssd_300_vgg16_voc}; B=${B:-64}
for v in trained worst; do
  timeout -k 5 90 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_full_$v -o p -- python3 $R/tools/bench_post.py $CFG $B $v 20 > $R/gpurun_out/full_$v.log 2>&1
  echo full $v rc=$?
done
for st in 1 2; do
  SSDK_POST_STOP=$st timeout -k 5 90 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stop_$st -o p -- python3 $R/tools/bench_post.py $CFG $B trained 20 > $R/gpurun_out/stop_$st.log 2>&1
  echo stop $st rc=$?
done
for st in 1 2 3; do
  SSDK_NMS_STOP=$st timeout -k 5 90 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_nstop_$st -o p -- python3 $R/tools/bench_post.py $CFG $B trained 20 > $R/gpurun_out/nstop_$st.log 2>&1
  echo nstop $st rc=$?
done
echo done
