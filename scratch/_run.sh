python scratch/travel_dist.py hip profiles/r04_travel_cpu_summaries.json gpurun_out/r04_travel_ratio_distribution.txt > gpurun_out/r04_travel_log.txt 2>&1; tail -8 gpurun_out/r04_travel_ratio_distribution.txt
python scratch/other_configs_bench.py > gpurun_out/r04_other_configs.txt 2>&1; cat gpurun_out/r04_other_configs.txt
