#!/bin/bash
# A/B of one environment knob over training steps on the same box: tools/ab_step.sh VAR value1 value2 ...   (value "-" = unset)
# Configs: AB_CONFIGS="name:batch name:batch" (default: the headline step, SSD-512 and M2Det).  Prints ms per step, two rounds.
var=$1; shift
vals=("$@")
cfgs=${AB_CONFIGS:-"ssd_300_vgg16_voc:32 ssd_512_vgg16_coco:16 m2det_512_vgg16_coco:16"}
for rep in 1 2; do
  for v in "${vals[@]}"; do
    if [ "$v" = "-" ]; then unset "$var"; else export "$var=$v"; fi
    for cfg in $cfgs; do
      c=${cfg%:*}; b=${cfg#*:}
      python bench.py --config "$c" --batch "$b" --no-cpu-baseline --no-extra-legs --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', '$c', round(d['ms_per_step'],4))"
    done
  done
done
