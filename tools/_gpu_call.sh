cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm" > $O/pytest_ops.log 2>&1; echo "ops rc=$?" > $O/rc.txt
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -s -k "backbone or classification" > $O/pytest_full.log 2>&1; echo "full rc=$?" >> $O/rc.txt
for i in 1 2; do
 for m in 0 2 1; do
  python bench.py --steps 10 --warmup 2 --clip-ln-fold $m --no-cpu-baseline --no-inclusive > $O/bench_m$m.$i.json 2> $O/bench_m$m.$i.err; python -c "import json; d=json.load(open('$O/bench_m$m.$i.json')); print('fold_mode$m', d['ms_per_step'], d['value'])" >> $O/rc.txt
 done
done
grep -E "passed|failed|Error|error" $O/pytest_ops.log $O/pytest_full.log | tail -12; cat $O/rc.txt
