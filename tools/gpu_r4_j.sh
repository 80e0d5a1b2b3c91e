# Round-4 GPU call J: fresh fuzz examples for every conv kernel family after the epilogue / prologue rewrite + soak
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
(PMX_FUZZ=400 timeout 1200 python -m pytest "tests/test_gpu_winograd.py::test_random_run_geometry_shapes_bit_exact" "tests/test_gpu_conv.py" -m gpu -q -x -k "random or fuzz or shapes") > $O/fuzz.log 2>&1; echo "fuzz rc=$?" | tee -a $O/summary.log
(timeout 300 python tools/soak.py --steps 300 --batch 32) > $O/soak.txt 2>&1; echo "soak rc=$?" | tee -a $O/summary.log
(timeout 300 python tools/soak.py --steps 400 --batch 1) >> $O/soak.txt 2>&1
(timeout 300 python tools/soak.py --steps 200 --batch 5) >> $O/soak.txt 2>&1
tail -3 $O/fuzz.log; grep -v amdgpu.ids $O/soak.txt | tail -6
