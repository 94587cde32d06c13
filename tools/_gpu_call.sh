cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3s; mkdir -p $O
T=$PWD/odise_amd/lib/libodise_hip_tools.so
: > $O/rc.txt
for i in 1 2; do
 for m in 0 4096; do
  ODISE_HIP_LIB=$T ODISE_GEMM_FLAGS=$m python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inclusive > $O/bench_m$m.$i.json 2> $O/bench_m$m.$i.err; python -c "import json; d=json.load(open('$O/bench_m$m.$i.json')); print('flags$m', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'])" >> $O/rc.txt
 done
done
cat $O/rc.txt
