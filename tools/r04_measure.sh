# round-4 measurement pass on the GPU box: bash tools/r04_measure.sh   (small summaries only are left under gpurun_out/r04/)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=/tmp/r04_work; rm -rf $W; mkdir -p $W
echo "== bench line"; timeout -k 10 900 python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err; echo rc=$?
stats() {   # tag, bench args...
  tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $W/prof_$tag -o p -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra-legs > $W/prof_$tag.log 2>&1; rc=$?
  grep "^{" $W/prof_$tag.log | tail -1 > $O/${tag}_bench_line.json
  python3 $R/tools/rocpd_stats.py $W/prof_$tag/p_results.db 45 > $O/${tag}_kernel_stats.md 2>&1
  if [ "$tag" = ssd300_b32 ]; then python3 $R/tools/rocpd_calls.py $W/prof_$tag/p_results.db igemm_streamk_kernel > $O/streamk_calls.txt 2>&1; fi
  rm -rf $W/prof_$tag
  echo "stats $tag rc=$rc"
}
stats ssd300_b32 --steps 20 --warmup 3
stats ssd300_b64 --config ssd_300_vgg16_voc --batch 64 --steps 5 --warmup 2
stats ssd300_c21_b32 --config ssd_300_vgg16_voc_c21 --batch 32 --steps 5 --warmup 2
stats ssd512_b16 --config ssd_512_vgg16_coco --batch 16 --steps 5 --warmup 2
stats retina_b32 --config retina_rn50_500_coco --batch 32 --steps 4 --warmup 1
stats m2det_b16 --config m2det_512_vgg16_coco --batch 16 --steps 4 --warmup 1
stats mb2_b2 --config ssd_mb2_voc --batch 2 --steps 10 --warmup 2
echo "== deterministic mode, kernel stats"
SSDK_DETERMINISTIC=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $W/prof_det -o p -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-legs > $W/prof_det.log 2>&1
grep "^{" $W/prof_det.log | tail -1 > $O/deterministic_ssd300_b32_bench_line.json
python3 $R/tools/rocpd_stats.py $W/prof_det/p_results.db 30 > $O/deterministic_ssd300_b32_kernel_stats.md 2>&1
rm -rf $W/prof_det
echo "== step timeline"
timeout -k 10 300 rocprofv3 --kernel-trace -d $W/gp -o p -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs > $W/gp.log 2>&1
python3 $R/tools/rocpd_gaps.py $W/gp/p_results.db multi_tensor_apply 5 > $O/step_timeline_ssd300_b32.txt 2>&1
rm -rf $W/gp
echo "== fast mode block"
timeout -k 10 400 python3 $R/bench.py --fast-mode-only > $O/fast_mode_line.json 2> $O/fast_mode.err; echo rc=$?
SSDK_FAST_MODE=bf16x3 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $W/prof_fast -o p -- python3 $R/bench.py --config retina_rn50_500_coco --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs > $W/prof_fast.log 2>&1
grep "^{" $W/prof_fast.log | tail -1 > $O/fast_mode_retina_b32_bench_line.json
python3 $R/tools/rocpd_stats.py $W/prof_fast/p_results.db 16 > $O/fast_mode_retina_b32_kernel_stats.md 2>&1
rm -rf $W/prof_fast
echo "== pmc passes"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $W/pmc_$i -o p -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs > $W/pmc_$i.log 2>&1; echo pmc $i rc=$?
done
python3 $R/tools/collect_pmc.py $O/pmc.json ssd_300_vgg16_voc:b32 $W/pmc_1 $W/pmc_2 $W/pmc_3 > $O/pmc_collect.log 2>&1
echo "== postprocess b64 stats"
for v in trained worst; do
  timeout -k 5 90 rocprofv3 --kernel-trace --stats -d $W/prof_post_$v -o p -- python3 $R/tools/bench_post.py ssd_300_vgg16_voc 64 $v 20 > $O/post_b64_$v.log 2>&1; echo post $v rc=$?
  python3 $R/tools/rocpd_stats.py $W/prof_post_$v/p_results.db 8 > $O/post_b64_${v}_kernel_stats.md 2>&1
done
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $W/pmcpost_$set -o p -- python3 $R/tools/bench_post.py ssd_300_vgg16_voc 64 worst 6 > /dev/null 2>&1
done
python3 $R/tools/collect_pmc.py $O/pmc_post.json ssd_300_vgg16_voc:b64:worst $W/pmcpost_FETCH_SIZE $W/pmcpost_WRITE_SIZE > $O/pmc_post_collect.log 2>&1
echo "== 2-rank rehearsal of the self-launching bench (gloo, both ranks on this one GPU)"
SSDK_BENCH_ONE_GPU=1 SSDK_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo rc=$?
rm -rf $W
du -sh $O
echo done
