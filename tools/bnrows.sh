cd /tmp && export TMPDIR=/tmp
for r in 16 32; do
  SSDK_BN_ROWS=$r timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/bn_$r -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-legs > /tmp/bn_$r.log 2>&1
  echo "rows $r"; python3 $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/bn_$r/p_results.db 40 | grep "bn_reduce"
  grep "^{" /tmp/bn_$r.log | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
  rm -rf /tmp/bn_$r
done
