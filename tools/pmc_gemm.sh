cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; W=/tmp/pg; rm -rf $W; mkdir -p $W $R/gpurun_out/pg
timeout -k 10 200 python3 -m pytest $R/tests/test_heads_gpu.py -x -q -k "vs_torch or golden" 2>&1 | tail -2
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $W/$set -o p -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $W/$set.log 2>&1; echo $set rc=$?
done
python3 $R/tools/collect_pmc.py $R/gpurun_out/pg/pmc.json x $W/FETCH_SIZE $W/WRITE_SIZE > /dev/null
python3 -c "
import json; d=json.load(open('$R/gpurun_out/pg/pmc.json'))['x:igemm_fwd_heads']; print('FETCH x2 MB', 2*d['FETCH_SIZE_KiB']/1024, 'WRITE MB', d['WRITE_SIZE_KiB']/1024, 'us', d['profiled_us'])"
for i in 1 2; do timeout -k 10 120 python3 $R/bench.py --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],3), round(d['roofline']['ms_per_step'],3))"; done
