# Round-4 rocprofv3 kernel statistics of the batch LM (5 iterations, PCG) on the bench's three graph shapes -> gpurun_out/r04/ (run through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
for n in bench large roof; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_ba_$n -- python $R/tools/ba_variant_probe.py $n > $O/ba_$n.log 2>&1
done
cd $R
for n in bench large roof; do DB=$(find $O/prof_ba_$n -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/ba_${n}_kernel_stats.txt 2>&1; tail -1 $O/ba_$n.log; cut -c1-160 $O/ba_${n}_kernel_stats.txt | head -26; done
find $O -name "*.db" -size +20M -delete
