# the round's profile set:  gpurun --timeout 2400 -- 'bash tools/jobs/round_profile.sh r06_a'
TAG=${1:-r06_x}
bash tools/profile_round.sh $TAG
OUT=gpurun_out/prof_$TAG
timeout 500 bash tools/pmc_kernels.sh timed > $OUT/pmc_kernels_timed.txt 2>&1
timeout 600 python tools/plan_report.py --measure > $OUT/plan_report.txt 2> $OUT/plan_report.err
timeout 200 bash tools/jobs/wfs_var.sh TH_WF_SPLIT=0 TH_WF_DBG=21 TH_WF_DBG=1565 TH_WF_DBG=1 TH_WF_DBG=16 TH_WF_DBG=4 > $OUT/wfs_layer_rate.txt 2>&1
timeout 300 bash tools/jobs/wfs_pmc.sh > $OUT/wfs_pmc.txt 2>&1
timeout 400 bash tools/jobs/first5_pmc.sh > $OUT/first5_pmc.txt 2>&1
