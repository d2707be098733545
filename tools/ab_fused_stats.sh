for i in 1 2; do
for c in "ssd_300_vgg16_voc 32" "m2det_512_vgg16_coco 16"; do set -- $c
 for e in 0 1; do
  if [ $e = 1 ]; then export SSDK_NO_FUSED_STATS=1; else unset SSDK_NO_FUSED_STATS; fi
  timeout -k 10 200 python3 bench.py --config $1 --batch $2 --steps 10 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$1 nofuse=$e\", round(d[\"ms_per_step\"],3), round(d[\"roofline\"][\"ms_per_step\"],4))"
 done; done; done
