cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
bash tools/prof_workload.sh igev_refine32 r6igev > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
head -25 gpurun_out/prof_r6igev/kernel_stats.csv | cut -c1-200 > gpurun_out/r6/igev_refine32_kernel_stats_head.txt
bash tools/prof_workload.sh igev_refine32 r6igeva --amp > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
head -25 gpurun_out/prof_r6igeva/kernel_stats.csv | cut -c1-200 > gpurun_out/r6/igev_refine32_amp_kernel_stats_head.txt
