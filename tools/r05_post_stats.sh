# kernel statistics of ONE postprocess leg on the GPU box: bash tools/r05_post_stats.sh <tag> <config> <batch> <trained|worst>   (-> gpurun_out/r05/<tag>_post_kernel_stats.md)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=/tmp/r05_work_$$; rm -rf $W; mkdir -p $W
tag=$1; shift
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $W/prof -o p -- python3 $R/tools/post_leg.py "$@" > $W/prof.log 2>&1; rc=$?
tail -1 $W/prof.log > $O/${tag}_post_leg.txt
python3 $R/tools/rocpd_stats.py $W/prof/p_results.db 12 > $O/${tag}_post_kernel_stats.md 2>&1
rm -rf $W
echo "post stats $tag rc=$rc"; cat $O/${tag}_post_leg.txt; grep "post_" $O/${tag}_post_kernel_stats.md | cut -c1-150
