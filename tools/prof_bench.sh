# rocprofv3 kernel stats of the default bench command: bash tools/prof_bench.sh [tag] [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-bench}; shift
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o p -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
echo rc=$?
tail -1 $R/gpurun_out/prof_$TAG.log | cut -c1-300
