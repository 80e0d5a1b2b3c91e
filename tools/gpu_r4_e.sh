# Round-4 GPU call E: block timing diagnostics (tools/block_timing.py) for a 7x7 layer, conv2_1, conv3_2, conv4_2
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
(timeout 600 python tools/block_timing.py --ks 7 --cin 128 --batch 64 --json $O/bt_7x7.json) > $O/bt_7x7.log 2>&1
(timeout 600 python tools/block_timing.py --ks 3 --cin 64 --cout 128 --hw 184 --batch 32 --json $O/bt_conv2_1.json) > $O/bt_conv2_1.log 2>&1
(timeout 600 python tools/block_timing.py --ks 3 --cin 256 --cout 256 --hw 92 --batch 32 --json $O/bt_conv3_2.json) > $O/bt_conv3_2.log 2>&1
(timeout 600 python tools/block_timing.py --ks 3 --cin 512 --cout 512 --hw 46 --batch 32 --json $O/bt_conv4_2.json) > $O/bt_conv4_2.log 2>&1
for f in bt_7x7 bt_conv2_1 bt_conv3_2 bt_conv4_2; do echo == $f; tail -40 $O/$f.log; done
