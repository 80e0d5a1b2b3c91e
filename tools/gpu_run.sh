#!/bin/bash
# tools/gpu_run.sh -- THE script for a call on the GPU box (replaces the per-experiment gpu_r<N>_<x>.sh scripts of rounds 3-5, last present
# at commit e8c97bc; tools/README.md lists the invocations that produced the committed profiles).
#
#   gpurun --timeout 1800 -- 'bash tools/gpu_run.sh <tag> <step> [<step> ...]'
#
# Everything is written under gpurun_out/<tag>/ (summary.log = one "step rc=N" line per step); afterwards, in the authoring container:
#   python tools/summarize_profiles.py <tag> <rNN>     -> profiles/<rNN>_*
# Steps (each under its own `timeout`, none aborts the call):
#   tests[:<pytest -k expr>]   the GPU suite (or a subset)            smoke            __graft_entry__.smoke()
#   bench                      python bench.py (full line)            bench_gather     bench.py --force-gather (one-rank RCCL record path)
#   bench_fast                 bench.py --no-cpu-baseline --no-extras bench_ranks:<n>  bench.py --gpus n over gloo on this one GPU
#   census  census1  census_rect  census_modes    the margin census: 512 frames batch 32 | 128 frames batch 1 | 368x496 + 496x368 |
#                              512 frames with the conv1-direct and direct-kernel engine modes next to the default (drift attribution)
#   rp_bench rp_drv rp_b1 rp_rect rp_precise rp_mixed   rocprofv3 --kernel-trace --stats of bench.py / the batch-32 driver / batch 1 /
#                              368x496 / detect_precise / the mixed-size batch
#   pmc                        the five PMC passes of the batch-32 driver (one counter group per pass, MI355X_MICROARCH.md)
#   pmc_b1                     the same of the batch-1 driver
#   rect_batches               landscape / portrait rate per pixel vs the square case at batch 8, 16, 24, 32
#   ab:<pytest -k expr>        A/B of csrc/libpose_base.so.keep vs the current build (whole step at batch 32 + batch 1, layer profile)
#   variants[:time-conv]       tools/kernel_variants.py time (whole step, layer profile) | time-conv (single layers) over the libraries in tools/_build/
#   block_timing:<args>        tools/block_timing.py <args with , for spaces>
#   soak                       tools/soak.py + fresh-example fuzz (PMX_FUZZ)
#   py:<script and args, with , for spaces>       any tools/ script
TAG=${1:?tag}; shift
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
C=chainer_realtime_multi-person_pose_estimation_amd/csrc
note() { echo "$1 rc=$2" | tee -a $O/summary.log; }
rp() { # rp <name> <timeout> <cmd...>: rocprofv3 kernel stats of a command, trace rows dropped (only the stats summary is kept)
  local n=$1 t=$2; shift 2
  (cd /tmp && timeout $t rocprofv3 --kernel-trace --stats -d $O/rp_$n -o $n --output-format csv -- "$@") > $O/rp_$n.log 2>&1; note rp_$n $?
  rm -f $O/rp_$n/*trace.csv $O/rp_$n/*/*trace.csv
}
pmc() { # pmc <suffix> <driver args...>
  local sfx=$1; shift
  for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    N=$(echo $P | cut -d" " -f1)
    (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace -d $O/pmc${sfx}_$N -o drv --output-format csv -- python $R/tools/profile_driver.py "$@") > $O/pmc${sfx}_$N.log 2>&1
    note pmc${sfx}_$N $?
  done
  rm -f $O/pmc${sfx}_*/*agent_info.csv $O/pmc${sfx}_*/*/*agent_info.csv
}
for STEP in "$@"; do
  S=${STEP%%:*}; A=""; [ "$S" != "$STEP" ] && A=${STEP#*:}
  case $S in
    tests) if [ -n "$A" ]; then (timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$A") > $O/pytest_gpu.log 2>&1; else (timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider) > $O/pytest_gpu.log 2>&1; fi; note tests $?; tail -5 $O/pytest_gpu.log ;;
    smoke) (timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; note smoke $? ;;
    bench) (timeout 900 python bench.py --steps 20 --warmup 3 --dump-profile $O/prof_bench.json) > $O/bench.log 2> $O/bench.err; note bench $? ;;
    bench_fast) (timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras) > $O/bench_fast.log 2> $O/bench_fast.err; note bench_fast $?; tail -c 600 $O/bench_fast.log ;;
    bench_gather) (timeout 400 python bench.py --steps 10 --warmup 2 --force-gather --no-cpu-baseline --no-extras) > $O/bench_force_gather.log 2> $O/bench_force_gather.err; note bench_gather $? ;;
    bench_ranks) (timeout 900 python bench.py --gpus $A --steps 4 --warmup 1 --backend gloo --no-cpu-baseline --no-profile --no-extras) > $O/bench_${A}ranks.log 2> $O/bench_${A}ranks.err; note bench_ranks_$A $? ;;
    census) (timeout 1200 python tools/parity_census.py --frames 512 --out $O/parity_census.json) > $O/census.log 2> $O/census.err; note census $? ;;
    census1) (timeout 900 python tools/parity_census.py --frames 128 --batch 1 --out $O/parity_census_single_image.json) > $O/census1.log 2> $O/census1.err; note census1 $? ;;
    census_rect) (timeout 900 python tools/parity_census.py --frames 128 --h 368 --w 496 --out $O/parity_census_368x496.json) > $O/census_r1.log 2> $O/census_r1.err; note census_368x496 $?
                 (timeout 900 python tools/parity_census.py --frames 96 --h 496 --w 368 --out $O/parity_census_496x368.json) > $O/census_r2.log 2> $O/census_r2.err; note census_496x368 $? ;;
    census_modes) (timeout 1500 python tools/parity_census.py --frames 512 --mode conv1_direct:conv1_wino=0 --mode direct_kernels:conv_algo=0 --out $O/parity_census_modes.json) > $O/census_modes.log 2> $O/census_modes.err; note census_modes $? ;;
    rp_bench) rp bench 600 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras ;;
    rp_drv) rp drv 300 python $R/tools/profile_driver.py --batch 32 --steps 3 ;;
    rp_b1) rp b1 300 python $R/tools/profile_driver.py --batch 1 --steps 20 ;;
    rp_rect) rp rect 300 python $R/tools/rect_time.py --h 368 --w 496 --batch 32 --steps 3 ;;
    rp_precise) rp precise 400 python $R/tools/precise_bench_driver.py ;;
    rp_mixed) rp mixed 400 python $R/tools/mixed_batch_time.py --steps 3 ;;
    pmc) pmc "" --batch 32 --steps 1 ;;
    pmc_b1) pmc _b1 --batch 1 --steps 4 ;;
    rect_batches) (timeout 900 python tools/rect_batches.py --json $O/rect_batches.json) > $O/rect_batches.log 2>&1; note rect_batches $?; tail -12 $O/rect_batches.log ;;
    ab) if [ -n "$A" ]; then (timeout 1200 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_conv.py -q -x -p no:cacheprovider -k "$A" 2>&1 | tail -15) > $O/ab_pytest.log; tail -4 $O/ab_pytest.log; fi
        cp $C/libpose_mi355x.so $C/libpose_new.so.keep
        for V in base new base new; do cp $C/libpose_$V.so.keep $C/libpose_mi355x.so
          for B in 32 1; do (timeout 300 python tools/profile_driver.py --batch $B --steps $((B == 1 ? 30 : 8))) 2>&1 | grep ms/step | sed "s/^/$V b$B /"; done
        done | tee $O/ab.log
        cp $C/libpose_new.so.keep $C/libpose_mi355x.so
        (timeout 300 python tools/profile_driver.py --batch 32 --steps 3 --profile-json $O/prof_new.json) > /dev/null 2>&1
        python tools/sum_layer_profile.py $O/prof_new.json 30; note ab 0 ;;
    variants) MODE=${A:-time}      # the variant libraries were built here (tools/kernel_variants.py build ...) and travel in tools/_build/
        (timeout 1500 python tools/kernel_variants.py $MODE --json $O/variants.json) > $O/variants.log 2>&1; note variants $?; tail -30 $O/variants.log ;;
    block_timing) (timeout 400 python tools/block_timing.py $(echo $A | tr ',' ' ') --json $O/block_timing.json) > $O/block_timing.log 2>&1; note block_timing $?; tail -30 $O/block_timing.log ;;
    soak) (PMX_FUZZ=200 timeout 1500 python -m pytest tests/test_gpu_properties.py -q -p no:cacheprovider) > $O/fuzz.log 2>&1; note fuzz $?
          (PMX_FUZZ=60 timeout 1500 python -m pytest tests/test_gpu_multi.py -q -p no:cacheprovider -k fuzz) > $O/fuzz_multi.log 2>&1; note fuzz_multi $?
          (timeout 900 python tools/soak.py) > $O/soak.log 2>&1; note soak $?
          (timeout 900 python tools/soak_modes.py) >> $O/soak.log 2>&1; note soak_modes $?; tail -8 $O/soak.log ;;
    py) (timeout 1500 python tools/$(echo $A | tr ',' ' ')) > $O/py_$(echo $A | cut -d, -f1 | tr '/.' '__').log 2>&1; note "py:$A" $?; tail -25 $O/py_$(echo $A | cut -d, -f1 | tr '/.' '__').log ;;
    *) echo "unknown step $STEP" | tee -a $O/summary.log ;;
  esac
done
cat $O/summary.log; du -sh $O
if [ -f $O/bench.log ]; then python - <<PY
import json
l=[q for q in open('$O/bench.log') if q.startswith('{')]
if l:
    d=json.loads(l[-1])
    print('fps %.1f ms %.3f dom %.4f frac %.3f step %.3f | single %.3f ms | precise %.2f ms batch8 %.2f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['step_roofline']['frac'],d['single_image']['ms_per_call'],d['precise']['ms_per_image'],d['precise']['batch8']['ms_per_image']))
PY
fi
