cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
for i in 1 2; do python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inclusive > $O/bench$i.json 2> $O/bench$i.err; python -c "import json; d=json.load(open('$O/bench$i.json')); print('bench$i', d['ms_per_step'], d['value'], d['exchange'])" >> $O/rc.txt; done
rocprofv3 --kernel-trace -d $O/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inclusive > $O/prof_bench.json 2> $O/prof.err
python tools/lane_timeline.py $O/prof/bench_results.db > $O/lane_timeline.txt 2>&1
grep -E "passed|failed|Error" $O/pytest.log | tail -5; cat $O/rc.txt; cat $O/lane_timeline.txt
