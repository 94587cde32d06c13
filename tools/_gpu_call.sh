cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
export ODISE_HIP_LIB=$PWD/odise_amd/lib/libodise_hip_tools.so
run() { name=$1; shift
  env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inclusive > $O/bench_$name.json 2> $O/bench_$name.err
  python -c "import json; d=json.load(open('$O/bench_$name.json')); print('$name', round(d['ms_per_step'],2), round(d['value'],2))" >> $O/summary.txt; }
run base X=1
run reserve8 ODISE_VAE_CU_RESERVE=8
run reserve4 ODISE_VAE_CU_RESERVE=4
run reserve3 ODISE_VAE_CU_RESERVE=3
run reserve2 ODISE_VAE_CU_RESERVE=2
run reserve4_old ODISE_VAE_CU_RESERVE=4 ODISE_LANE_ORDER_OLD=1
run base_b X=1
ODISE_VAE_CU_RESERVE=4 rocprofv3 --kernel-trace -d $O/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inclusive > $O/prof_bench.json 2> $O/prof.err
python tools/lane_timeline.py $O/prof/bench_results.db > $O/lane_timeline_reserve4.txt 2>&1
cat $O/summary.txt; cat $O/lane_timeline_reserve4.txt
