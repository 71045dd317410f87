#!/bin/bash
# The counter passes behind bench.py's roofline line, on the GPU box, for the tree as it is (pinned plans
# profiles/r03_plans.json): FETCH_SIZE pass, WRITE_SIZE pass, SQ matrix-pipe pass, each a rocprofv3 run of its own with the
# program directly behind `--`; summaries under gpurun_out/ (copy r03_traffic.{md,json} and r03_pmc_mfma.md to profiles/).
#   usage (from the repository root on the box): bash tools/roofline_passes.sh
R=$PWD
REPLAYS=3
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/traffic_probe.py --replays $REPLAYS --plans $R/profiles/r03_plans.json --meta $R/gpurun_out/traffic_meta.json > $R/gpurun_out/pmc_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $R/gpurun_out/pmc_$tag.log; exit 1; }; }
run fetch FETCH_SIZE &&
run write WRITE_SIZE TCC_WRITE_sum &&
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE &&
cd $R &&
python tools/traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write $REPLAYS gpurun_out/r03_traffic.md gpurun_out/r03_traffic.json gpurun_out/traffic_meta.json &&
python tools/pmc_summary.py gpurun_out/pmc_sq $REPLAYS gpurun_out/r03_pmc_mfma.md gpurun_out/traffic_meta.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
