# instruction / cycle counters of the postprocess kernels: bash tools/pmc_post.sh   (on the GPU box; rocprofv3 --pmc passes only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${CFG:-ssd_300_vgg16_voc}; B=${B:-64}; V=${V:-trained}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_post_$i -o p -- python3 $R/tools/bench_post.py $CFG $B $V 6 > $R/gpurun_out/pmc_post_$i.log 2>&1
  echo set $i rc=$?
done
