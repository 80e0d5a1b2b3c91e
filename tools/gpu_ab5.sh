cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -3
for V in 35 41; do timeout 120 python tools/conv_one.py --variant $V --k 3 --hw 92 --B 32 --cin 256 --cout 256 --iters 10; done
for G in 5 6 5 6; do python tools/profile_driver.py --batch 32 --steps 5 --gen $G | head -1; done
