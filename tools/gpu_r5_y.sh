# Round-5 GPU call Y: block timing build pointed at the unit-mode launches (single image): where the ~20 us outside the 1024 MFMAs go
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r05y; mkdir -p $O; cd $R
(timeout 200 python tools/block_timing.py --ks 7 --cin 128 --batch 1 --keep-tail --raw $O/u7_b1.npy --json $O/u7_b1.json) > $O/u7_b1.log 2>&1; echo "7x7 b1 rc=$?"
(timeout 200 python tools/block_timing.py --ks 7 --cin 128 --batch 2 --keep-tail --raw $O/u7_b2.npy --json $O/u7_b2.json) > $O/u7_b2.log 2>&1; echo "7x7 b2 rc=$?"
(timeout 200 python tools/block_timing.py --ks 3 --cin 512 --cout 512 --batch 1 --keep-tail --raw $O/u3_b1.npy --json $O/u3_b1.json) > $O/u3_b1.log 2>&1; echo "3x3 b1 rc=$?"
tail -30 $O/u7_b1.log
