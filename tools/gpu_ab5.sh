cd $GRAFT_REPO_ROOT
for G in 6 6 6; do python tools/profile_driver.py --batch 32 --steps 5 --gen $G | head -1; done
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
