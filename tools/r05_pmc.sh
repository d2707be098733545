# PMC passes of one bench configuration on the GPU box: bash tools/r05_pmc.sh <tag> [bench args...]  -> gpurun_out/r05/<tag>_pmc.json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=/tmp/r05_pmc_$$; rm -rf $W; mkdir -p $W
tag=$1; shift
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $W/pmc_$i -o p -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extra-legs > $W/pmc_$i.log 2>&1; echo pmc $i rc=$?
done
python3 $R/tools/collect_pmc.py $O/${tag}_pmc.json $tag $W/pmc_1 $W/pmc_2 $W/pmc_3 $W/pmc_4 > $O/${tag}_pmc_collect.log 2>&1
rm -rf $W
