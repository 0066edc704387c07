# The GPU test suite file by file (one python process each: a crash in one file does not hide the others), short tracebacks.
# usage (through gpurun, from the repo root):  bash tools/gpu_suite_by_file.sh gpurun_out/r04/suite.log [pytest args]
OUT=$1; shift
mkdir -p $(dirname $OUT); : > $OUT
for f in tests/test_*.py; do
  if grep -q "mark.gpu" $f; then
    echo "=== $f" >> $OUT
    timeout 600 python -m pytest $f -m gpu -q --tb=short -rf "$@" 2>&1 | grep -v "^  File \"/usr\|^Extension modules" | tail -40 >> $OUT
  fi
done
grep -n "^=== \|passed\|failed\|^FAILED\|^ERROR" $OUT
