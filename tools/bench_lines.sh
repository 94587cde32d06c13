#!/bin/bash
# The bench lines of the five configurations + kernel trace of the benchmarked step, with the GPU's clock / power sampled beside the headline run
# (boxes of this pool differ by up to 17 % in sustained step time at identical isolated-kernel rates).  Output: gpurun_out/lines/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/lines; rm -rf $O; mkdir -p $O
rocm-smi --showclocks --showpower --showtemp --showperflevel > $O/smi_before.txt 2>&1
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' ; echo; sleep 1; done ) > $O/smi_during_bench.txt 2>&1 &
SMI=$!
python bench.py > $O/bench_full_b4_1024.json 2> $O/bench_full.err; echo "bench rc=$?" > $O/rc.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-inclusive > $O/bench_full_b4_1024_steps20.json 2> $O/bench_full20.err
kill $SMI
python bench.py --stage unet --images 1 > $O/bench_unet_b1.json 2> $O/bench_unet_b1.err
python bench.py --stage unet --images 4 --no-cpu-baseline > $O/bench_unet_b4.json 2> $O/bench_unet_b4.err
python bench.py --stage unet --images 16 --no-cpu-baseline > $O/bench_unet_b16.json 2> $O/bench_unet_b16.err
python bench.py --vocab ade150 --images 8 --no-cpu-baseline --no-inclusive > $O/bench_ade150_b8_1024.json 2> $O/bench_ade150.err
python bench.py --vocab ade847 --size 1280 --images 2 --semantic-only --no-cpu-baseline --no-inclusive > $O/bench_ade847_b2_1280_semantic.json 2> $O/bench_ade847.err
python bench.py --in-flight 3 --steps 12 --no-cpu-baseline --no-inclusive > $O/bench_full_b4_1024_in_flight3.json 2> $O/bench_inflight.err
rocprofv3 --kernel-trace -d $O/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inclusive > $O/prof_bench.json 2> $O/prof.err
python tools/db_by_shape.py $O/prof/bench_results.db marker 60 > $O/bench_full_by_shape.txt 2>&1
python tools/lane_timeline.py $O/prof/bench_results.db > $O/lane_timeline.txt 2>&1
rm -rf $O/prof
cat $O/rc.txt; for f in $O/bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['ms_per_step'],2), round(d['value'],2), d['unit'], d.get('roofline',{}).get('frac'))"; done
head -3 $O/lane_timeline.txt; tail -25 $O/smi_during_bench.txt | head -12
