cd /tmp && export TMPDIR=/tmp
for r in 192 256 384; do
  SSDK_BN_WGS=$r timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/bnr_$r -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config retina_rn50_500_coco --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > /tmp/bnr_$r.log 2>&1
  echo "target wgs $r"; python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/bnr_$r/p_results.db 40 | grep "bn_reduce"
  grep "^{" /tmp/bnr_$r.log | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
  rm -rf /tmp/bnr_$r
done
