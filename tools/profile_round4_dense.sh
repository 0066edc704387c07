# Round 4, dense reduced-camera solver: stand-alone check of every launch sequence (tools/dense_check), kernel statistics of sequences 1 / 3 / 4
# on the 2176^2 system, the dense tests, ms per LM iteration per sequence on the bench graph, then - with the fastest sequence that passed exported
# as VDO_BA_DENSE - the bench's GPU legs and the rest of the GPU suite.  Everything under gpurun_out/r04d/ (run through gpurun from the repo root).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 $R/tools/dense_check 5 > $O/dense_check.log 2>&1; echo "dense_check rc $?" | tee -a $O/dense_check.log
cat $O/dense_check.log | tail -50
for v in 1 3 4; do
  timeout 120 rocprofv3 --kernel-trace --stats -d $O/prof_check_v$v -- $R/tools/dense_check 5 $v > $O/dense_check_prof_v$v.log 2>&1
done
cd $R
for v in 1 3 4; do DB=$(find $O/prof_check_v$v -name "*.db" | head -1); python tools/rocprof_summary.py $DB 20 > $O/dense_check_v${v}_kernel_stats.txt 2>&1; cut -c1-150 $O/dense_check_v${v}_kernel_stats.txt | head -12; done
find $O -name "*.db" -size +20M -delete
# fastest sequence whose every case passed
BEST=$(python - <<'EOF'
import re, collections
ok = collections.defaultdict(lambda: True); us = {}
for line in open("gpurun_out/r04d/dense_check.log"):
    m = re.match(r"n=(\d+) n0=\d+ \w+ v(\d): .* us ([\d.]+) flag \d+ (ok|BAD)", line)
    if not m: continue
    n, v, t, st = int(m.group(1)), int(m.group(2)), float(m.group(3)), m.group(4)
    if st != "ok": ok[v] = False
    if n == 2176: us[v] = t
good = [v for v in us if ok[v]]
print(min(good, key=lambda v: us[v]) if good else 1)
EOF
)
echo "best dense sequence: $BEST" | tee $O/best_version.txt
timeout 500 python -m pytest tests/test_dense_check_gpu.py tests/test_ba_gpu.py -m gpu -q --tb=short -rf 2>&1 | grep -v "^  File \"/usr" | tail -25 > $O/test_ba.log; tail -6 $O/test_ba.log
timeout 200 python tools/dense_probe.py > $O/dense_probe.log 2>&1; cat $O/dense_probe.log | tail -8
export VDO_BA_DENSE=$BEST
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_probe -- env DENSE_PROBE_VERSIONS=$BEST python $R/tools/dense_probe.py > $O/dense_probe_prof.log 2>&1 )
DB=$(find $O/prof_probe -name "*.db" | head -1); python tools/rocprof_summary.py $DB 40 > $O/dense_probe_kernel_stats.txt 2>&1; cut -c1-150 $O/dense_probe_kernel_stats.txt | head -30
find $O -name "*.db" -size +20M -delete
timeout 400 python bench.py --no-cpu-baseline --no-live-pmc --no-host-inputs > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-600 $O/bench.json
bash tools/gpu_suite_by_file.sh $O/suite.log --ignore=tests/test_ba_gpu.py --ignore=tests/test_dense_check_gpu.py 2>&1 | tail -40
