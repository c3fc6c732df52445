# usage: bash tools/_flake_loop.sh N [ENV=VAL ...]   — run the pipelined-vs-serial test N times, print the failure messages
n=$1; shift
fail=0
for i in $(seq 1 $n); do
  out=$(env "$@" timeout 300 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k test_pipelined_batches_equal_serial 2>&1)
  if echo "$out" | grep -q "1 passed"; then :; else fail=$((fail+1)); echo "$out" | grep "^E  " | head -2; fi
done
echo "env [$*]: $fail failures of $n"
