cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -8
for V in 36 42; do timeout 120 python tools/conv_one.py --variant $V --k 3 --hw 368 --B 32 --cin 3 --cout 64 --iters 10; done
for G in 5 6 5 6; do python tools/profile_driver.py --batch 32 --steps 5 --gen $G | head -1; done
