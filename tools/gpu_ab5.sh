cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -3
for G in 5 6; do for B in 8 16 24 32 48; do python tools/profile_driver.py --batch $B --steps 5 --gen $G | head -1; done; done
