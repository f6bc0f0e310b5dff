#!/bin/bash
# usage: scratch/final_measure.sh <tag>   -- the round's measurement set on ONE box: plain bench, interleaved A/B against the previous
# round's tree, bench under rocprofv3 (kernel stats, category summary, kernel sequence), matrix-pipe busy share per kernel, HBM traffic per kernel
tag=${1:-r05_v1}
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench: $(cut -c1-220 gpurun_out/${tag}_bench.json)"
bash scratch/ab_base.sh 3 100 > gpurun_out/${tag}_ab_vs_previous_round.txt 2>&1
cat gpurun_out/${tag}_ab_vs_previous_round.txt
bash scratch/measure.sh $tag > gpurun_out/${tag}_meas.txt 2>&1
head -20 gpurun_out/cat_$tag.txt
bash scratch/pmc_mfma.sh > gpurun_out/${tag}_mfma_busy_per_kernel.txt 2>&1
head -12 gpurun_out/${tag}_mfma_busy_per_kernel.txt
bash scratch/pmc_bench_traffic.sh > gpurun_out/${tag}_traffic_top.txt 2>&1
cp gpurun_out/traffic.json gpurun_out/${tag}_traffic.json
head -5 gpurun_out/${tag}_traffic_top.txt
