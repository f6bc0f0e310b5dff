echo full; python scratch/bench_img16.py
for ab in 1 2 3; do echo "ablate $ab"; GANK_LIB_NAME=libgank_ab$ab.so python scratch/bench_img16.py; done
