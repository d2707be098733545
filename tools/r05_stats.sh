# kernel statistics + step timeline of one bench configuration on the GPU box: bash tools/r05_stats.sh <tag> [bench args...]   (-> gpurun_out/r05/<tag>_*)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=/tmp/r05_work_$$; rm -rf $W; mkdir -p $W
tag=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $W/prof -o p -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra-legs > $W/prof.log 2>&1; rc=$?
grep "^{" $W/prof.log | tail -1 > $O/${tag}_bench_line.json
python3 $R/tools/rocpd_stats.py $W/prof/p_results.db 90 > $O/${tag}_kernel_stats.md 2>&1
python3 $R/tools/rocpd_gaps.py $W/prof/p_results.db multi_tensor_apply 5 > $O/${tag}_step_timeline.txt 2>&1
rm -rf $W
echo "stats $tag rc=$rc"
