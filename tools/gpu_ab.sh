# A/B of two builds of the library in one GPU call: csrc/libpose_base.so.keep (baseline) vs the current build
# usage: gpurun -- 'bash tools/gpu_ab.sh <tag> [pytest -k expression]'
TAG=${1:-ab}; KEXPR=${2:-}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
C=chainer_realtime_multi-person_pose_estimation_amd/csrc
if [ -n "$KEXPR" ]; then (timeout 1200 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv.py -q -x -p no:cacheprovider -k "$KEXPR" 2>&1 | tail -15) > $O/pytest.log; tail -4 $O/pytest.log; fi
cp $C/libpose_mi355x.so $C/libpose_new.so.keep
for V in base new base new; do
  cp $C/libpose_$V.so.keep $C/libpose_mi355x.so
  for B in 32; do (timeout 300 python tools/profile_driver.py --batch $B --steps 8) 2>&1 | grep ms/step | sed "s/^/$V /"; done
done | tee $O/ab.log
cp $C/libpose_new.so.keep $C/libpose_mi355x.so
(timeout 300 python tools/profile_driver.py --batch 32 --steps 3 --profile-json $O/prof_new.json) > /dev/null 2>&1
(timeout 300 python tools/profile_driver.py --batch 1 --steps 30) 2>&1 | grep ms/step
python tools/sum_layer_profile.py $O/prof_new.json 30
