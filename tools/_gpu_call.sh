cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
export ODISE_HIP_LIB=$PWD/odise_amd/lib/libodise_hip_tools.so
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inclusive > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['ms_per_step'],2), round(d['value'],2))" >> $O/summary.txt
}
run new_defer X=1
run new_nodefer ODISE_NO_DEFER_JOIN=1
run old_defer ODISE_LANE_ORDER_OLD=1
run old_nodefer ODISE_LANE_ORDER_OLD=1 ODISE_NO_DEFER_JOIN=1
run new_defer_prio ODISE_LANE2_HIGH_PRIORITY=1
run old_defer_prio ODISE_LANE_ORDER_OLD=1 ODISE_LANE2_HIGH_PRIORITY=1
run new_defer_b X=1
run old_defer_b ODISE_LANE_ORDER_OLD=1
ODISE_LANE2_HIGH_PRIORITY=1 rocprofv3 --kernel-trace -d $O/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inclusive > $O/prof_bench.json 2> $O/prof.err
python tools/lane_timeline.py $O/prof/bench_results.db > $O/lane_timeline_new_defer_prio.txt 2>&1
cat $O/summary.txt; cat $O/lane_timeline_new_defer_prio.txt
