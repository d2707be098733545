#!/bin/bash
# A/B of one environment knob over the headline step and M2Det on the same box: tools/ab_step.sh VAR value1 value2 ...
# (value "-" = unset).  Prints ms per step per config, two rounds.
var=$1; shift
vals=("$@")
for rep in 1 2; do
  for v in "${vals[@]}"; do
    if [ "$v" = "-" ]; then unset "$var"; else export "$var=$v"; fi
    for cfg in "ssd_300_vgg16_voc 32" "m2det_512_vgg16_coco 16"; do
      c=${cfg% *}; b=${cfg#* }
      python bench.py --config "$c" --batch "$b" --no-cpu-baseline --no-extra-legs --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', '$c', round(d['ms_per_step'],4))"
    done
  done
done
