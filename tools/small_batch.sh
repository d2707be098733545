for cfg in ssd_mb2_voc ssd_300_vgg16_voc; do for b in 1 2 4 8; do
 for e in 0 1; do
  if [ $e = 1 ]; then export SSDK_HEADS_NO_SPLITK=1; else unset SSDK_HEADS_NO_SPLITK; fi
  python3 bench.py --config $cfg --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg b$b nosplit=$e', round(d['ms_per_step'],3),'ms/step gemm', round(d['roofline']['ms_per_step'],4), 'ms eval', round(d.get('eval_images_per_sec',0)))"
 done; done; done
