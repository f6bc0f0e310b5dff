#!/bin/bash
# usage: scratch/measure.sh <tag>   -- bench under rocprofv3 kernel trace, category summary per iteration, kernel sequence of one iteration
export GANK_BENCH_EXTRAS=0     # bench.py: the headline measurement only (no fp16 child, no other configurations)
tag=${1:-m}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag --output-format csv -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err < /dev/null
python3 scratch/cat_summary.py gpurun_out/prof_$tag 27 > gpurun_out/cat_$tag.txt 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/bench_$tag.json | head -2
head -14 gpurun_out/cat_$tag.txt
python3 scratch/seq_dump.py gpurun_out/prof_$tag 1000 > gpurun_out/seq_$tag.txt 2>&1
rm -f gpurun_out/prof_$tag/*kernel_trace.csv
