cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -2
for V in 2 29 36; do python tools/conv_one.py --variant $V --k 3 --hw 368 --cin 64 --cout 64 --B 8 --iters 10; done
for V in 2 29 36; do python tools/conv_one.py --variant $V --k 3 --hw 184 --cin 64 --cout 64 --B 16 --iters 10; done
for G in 4 5; do for B in 1 4 8; do python tools/profile_driver.py --batch $B --steps 20 --gen $G | head -1; done; done
for G in 4 5 4 5; do python tools/profile_driver.py --batch 32 --steps 5 --gen $G | head -1; done
