# per-call durations of the kernels whose name contains <pattern> in one bench configuration: bash tools/r05_calls.sh <tag> <pattern> [bench args...]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=/tmp/r05_calls_$$; rm -rf $W; mkdir -p $W
tag=$1; pat=$2; shift; shift
timeout -k 10 300 rocprofv3 --kernel-trace -d $W/prof -o p -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra-legs > $W/prof.log 2>&1; rc=$?
python3 $R/tools/rocpd_calls.py $W/prof/p_results.db "$pat" > $O/${tag}_calls.txt 2>&1
rm -rf $W
echo "calls $tag rc=$rc"; tail -n 24 $O/${tag}_calls.txt
