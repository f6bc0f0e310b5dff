export GANK_LIB_NAME=libgank_tune.so
echo "== defaults"; python scratch/bench_critic_wgrad.py
for pf in 2 3 4; do for tgt in 128 256 512 768; do echo "== rows PF=$pf TARGET=$tgt"; GANK_WGRAD_ROWS_PF=$pf GANK_WGRAD_ROWS_TARGET=$tgt python scratch/bench_critic_wgrad.py rows; done; done
for ms in 1 2 4 8; do for tgt in 128 256 512; do echo "== lean MIN_STEPS=$ms SPLIT_TARGET=$tgt"; GANK_WGRAD_MIN_STEPS=$ms GANK_WGRAD_SPLIT_TARGET=$tgt python scratch/bench_critic_wgrad.py lean; done; done
