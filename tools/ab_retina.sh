cd /tmp && export TMPDIR=/tmp
for e in 0 1; do
  if [ $e = 1 ]; then export SSDK_NO_RELU_BY_NORM=1; else unset SSDK_NO_RELU_BY_NORM; fi
  timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --config retina_rn50_500_coco --batch 32 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"retina no_relu_by_norm=$e\", round(d[\"ms_per_step\"],3))"
done
unset SSDK_NO_RELU_BY_NORM
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/rt -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config retina_rn50_500_coco --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > /tmp/rt.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/rt/p_results.db 14 | cut -c1-150
rm -rf /tmp/rt
for i in 1 2; do timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"ssd300\", round(d[\"ms_per_step\"],3), round(d[\"roofline\"][\"ms_per_step\"],4))"; done
