# round-5 measurement pass on the GPU box: bash tools/r05_measure.sh [part]   (small summaries only are left under gpurun_out/r05m/)
#   part a: the bench line + kernel statistics of every configuration + deterministic mode + step timeline
#   part b: PMC passes, postprocess legs, step_fn leg, fast mode, 2-rank rehearsal
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=/tmp/r05_work; rm -rf $W; mkdir -p $W
part=${1:-ab}
stats() {   # tag, bench args...
  tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $W/prof_$tag -o p -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra-legs > $W/prof_$tag.log 2>&1; rc=$?
  grep "^{" $W/prof_$tag.log | tail -1 > $O/${tag}_bench_line.json
  python3 $R/tools/rocpd_stats.py $W/prof_$tag/p_results.db 50 > $O/${tag}_kernel_stats.md 2>&1
  if [ "$tag" = ssd300_b32 ]; then
    python3 $R/tools/rocpd_calls.py $W/prof_$tag/p_results.db igemm_streamk_kernel > $O/streamk_calls.txt 2>&1
    python3 $R/tools/rocpd_gaps.py $W/prof_$tag/p_results.db multi_tensor_apply 5 > $O/step_timeline_ssd300_b32.txt 2>&1
  fi
  rm -rf $W/prof_$tag
  echo "stats $tag rc=$rc"
}
case $part in *a*)
echo "== bench line"; timeout -k 10 1000 python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err; echo rc=$?
stats ssd300_b32 --steps 20 --warmup 3
stats ssd300_b64 --config ssd_300_vgg16_voc --batch 64 --steps 5 --warmup 2
stats ssd300_c21_b32 --config ssd_300_vgg16_voc_c21 --batch 32 --steps 5 --warmup 2
stats ssd512_b16 --config ssd_512_vgg16_coco --batch 16 --steps 5 --warmup 2
stats retina_b32 --config retina_rn50_500_coco --batch 32 --steps 4 --warmup 1
stats m2det_b16 --config m2det_512_vgg16_coco --batch 16 --steps 4 --warmup 1
stats mb2_b2 --config ssd_mb2_voc --batch 2 --steps 10 --warmup 2
echo "== deterministic mode: bench line + kernel stats"
SSDK_DETERMINISTIC=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $W/prof_det -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-legs > $W/prof_det.log 2>&1
grep "^{" $W/prof_det.log | tail -1 > $O/deterministic_ssd300_b32_bench_line.json
python3 $R/tools/rocpd_stats.py $W/prof_det/p_results.db 40 > $O/deterministic_ssd300_b32_kernel_stats.md 2>&1
rm -rf $W/prof_det
SSDK_DETERMINISTIC=1 timeout -k 10 200 python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/deterministic_ssd300_b32_steps100.json 2>/dev/null; echo det rc=$?
timeout -k 10 200 python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/default_ssd300_b32_steps100.json 2>/dev/null; echo def rc=$?
;; esac
case $part in *b*)
echo "== pmc passes (SSD-300 b32)"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $W/pmc_$i -o p -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $W/pmc_$i.log 2>&1; echo pmc $i rc=$?
done
python3 $R/tools/collect_pmc.py $O/pmc.json ssd_300_vgg16_voc:b32 $W/pmc_1 $W/pmc_2 $W/pmc_3 > $O/pmc_collect.log 2>&1
rm -rf $W/pmc_*
echo "== postprocess legs"
cd $R
for leg in "ssd_300_vgg16_voc 64 trained" "ssd_300_vgg16_voc 64 worst" "retina_rn50_500_coco 32 trained" "retina_rn50_500_coco 32 worst"; do
  set -- $leg
  tag=post_$(echo $1 | cut -d_ -f1)_b$2_$3
  timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $W/prof_$tag -o p -- python3 $R/tools/post_leg.py $1 $2 $3 20 > $O/$tag.log 2>&1; echo $tag rc=$?
  python3 $R/tools/rocpd_stats.py $W/prof_$tag/p_results.db 8 > $O/${tag}_kernel_stats.md 2>&1
  rm -rf $W/prof_$tag
done
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $W/pmcpost_$set -o p -- python3 $R/tools/post_leg.py ssd_300_vgg16_voc 64 worst 6 > /dev/null 2>&1
done
python3 $R/tools/collect_pmc.py $O/pmc_post.json ssd_300_vgg16_voc:b64:worst $W/pmcpost_FETCH_SIZE $W/pmcpost_WRITE_SIZE > $O/pmc_post_collect.log 2>&1
cd /tmp
echo "== step_fn leg (detection.init with and without graph_hot_path)"
timeout -k 10 400 python3 $R/bench.py --step-fn-only > $O/step_fn_line.json 2> $O/step_fn.err; echo rc=$?
echo "== fast mode block"
timeout -k 10 400 python3 $R/bench.py --fast-mode-only > $O/fast_mode_line.json 2> $O/fast_mode.err; echo rc=$?
echo "== 2-rank rehearsal of the self-launching bench (gloo, both ranks on this one GPU)"
SSDK_BENCH_ONE_GPU=1 SSDK_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo rc=$?
;; esac
rm -rf $W
du -sh $O
echo done
