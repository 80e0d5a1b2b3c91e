# detect_precise against GPU_MAX_HW_QUEUES on whatever box this call got: the precise leg alone in a process, and inside the full bench.py line
# usage (on the GPU box): [HWQ_LIST="1 2 3 8"] bash tools/hwq_sweep.sh <tag>   -> gpurun_out/<tag>/hwq.log
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/${1:-hwq}; mkdir -p $O
for Q in ${HWQ_LIST:-2 4 8 2 4 8}; do
  export GPU_MAX_HW_QUEUES=$Q
  A=$(timeout 300 python tools/precise_bench_leg.py 1 2>/dev/null | tail -1)
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_q$Q.log 2>/dev/null
  B=$(python - <<PY
import json
d=json.loads([l for l in open("$O/bench_q$Q.log") if l.startswith("{")][-1])
print("bench: fps %.1f single %.3f precise %.2f batch8 %.2f mixed %.1f" % (d["value"], d["single_image"]["ms_per_call"], d["precise"]["ms_per_image"], d["precise"]["batch8"]["ms_per_image"], d["mixed_sizes"]["mixed_batch_ms"]))
PY
)
  echo "Q=$Q | leg alone: $A | $B" | tee -a $O/hwq.log
done
