cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/eff2; rm -rf $O; mkdir -p $O
T=$PWD/odise_amd/lib/libodise_hip_tools.so
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inclusive > $O/bench_check.json 2> $O/bench_check.err
ODISE_HIP_LIB=$T ODISE_GEMM_FLAGS=32 rocprofv3 --kernel-trace --output-format csv -d $O/eff -o t -- python tools/gemm_eff.py run 2> $O/eff_gemm.log > /dev/null
python tools/gemm_eff.py join $O/eff/t_kernel_trace.csv $O/eff_gemm.log > $O/gemm_efficiency_by_shape.txt 2>&1
rm -rf $O/eff
python -c "import json; d=json.load(open('$O/bench_check.json')); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'])"
head -12 $O/gemm_efficiency_by_shape.txt | cut -c1-170
