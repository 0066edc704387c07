# Round-5 final GPU passes: K18 sweep profile (kernel stats, HBM traffic, SQ counters, phase probe), LM kernel statistics on the bench's large graphs,
# the A/B of the sweep variants on one box, the bench at --gpus 2 on the shared GPU (gloo: exercises the sharded legs), the whole GPU suite.
# usage (gpurun): bash tools/profile_round5_final.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z; mkdir -p $O
bash $R/tools/profile_round5_sweep.sh z > $O/sweep_profile.log 2>&1; tail -32 $O/sweep_profile.log
cd /tmp && export TMPDIR=/tmp
for n in large roof; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_ba_$n -- python $R/tools/ba_variant_probe.py $n > $O/ba_$n.log 2>&1
done
cd $R
for n in large roof; do DB=$(find $O/prof_ba_$n -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/ba_${n}_kernel_stats.txt 2>&1; tail -1 $O/ba_$n.log; cut -c1-160 $O/ba_${n}_kernel_stats.txt | head -22; done
find $O -name "*.db" -size +20M -delete
{ for v in old base5; do echo "== $v"; VDO_HIP_LIB=$R/vdo_slam_amd/libvdo_hip_$v.so timeout 300 python tools/sweep_repeat_probe.py 2200000 2>&1 | grep -v "^edges" | tail -6; done
  echo "== default"; timeout 300 python tools/sweep_repeat_probe.py 2200000 2>&1 | grep -v "^edges" | tail -6; } > $O/sweep_ab.txt 2>&1; cat $O/sweep_ab.txt
( time python bench.py --gpus 2 --steps 20 --no-cpu-baseline --no-host-inputs > $O/bench_gpus2_shared.json 2> $O/bench_gpus2.err ) 2> $O/bench_gpus2_time.txt; tail -3 $O/bench_gpus2_time.txt; tail -3 $O/bench_gpus2.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_gpus2_shared.json").read().strip().splitlines()[-1]); print("gpus2 value", d["value"], d.get("sharded")); 
except Exception as e: print("gpus2 failed", e)
PY
bash tools/gpu_suite_by_file.sh $O/suite.log > $O/suite_summary.txt 2>&1; tail -45 $O/suite_summary.txt
