cd $GRAFT_REPO_ROOT
for L in 0 84000; do python tools/conv_one.py --variant 32 --B 32 --iters 20 --min-lds $L; done
for L in 0 84000; do python tools/conv_one.py --variant 32 --B 128 --iters 10 --min-lds $L; done
for L in 0 84000; do python tools/conv_one.py --variant 33 --k 3 --B 32 --cin 256 --cout 256 --iters 20 --min-lds $L; done
