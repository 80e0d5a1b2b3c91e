cd $GRAFT_REPO_ROOT
for V in 39 130 131 132 39 130 131 132; do timeout 120 python tools/conv_one.py --variant $V --B 64 --iters 20; done
