cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${CFG:-ssd_300_vgg16_voc}; B=${B:-64}
run() { tag=$1; shift; env "$@" timeout -k 5 90 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o p -- python3 $R/tools/bench_post.py $CFG $B ${V:-trained} 20 > $R/gpurun_out/$tag.log 2>&1; echo $tag rc=$?; }
run base X=1
run ldsrow SSDK_POST_LDSROW=1
run wgs1792 SSDK_POST_WGS=1792
run wgs1792_ldsrow SSDK_POST_WGS=1792 SSDK_POST_LDSROW=1
run wgs2560 SSDK_POST_WGS=2560
V=worst run base_worst X=1
