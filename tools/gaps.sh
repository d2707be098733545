R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gaps; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/gp -o p -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench.log 2>&1
python3 $R/tools/rocpd_gaps.py /tmp/gp/p_results.db multi_tensor_apply 5 > $O/gaps.txt 2>&1
rm -rf /tmp/gp
cat $O/gaps.txt
