cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
ODISE_HIP_LIB=$PWD/odise_amd/lib/libodise_hip_tools.so timeout 300 python tools/post_bench.py > $O/post_bench.log 2>&1; echo "post_bench rc=$?" >> $O/rc.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-inclusive > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inclusive > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?" >> $O/rc.txt
python tools/trace_by_shape.py $(ls $O/prof/*/*kernel_trace.csv | head -1) marker 70 > $O/by_shape.txt 2>&1
rm -rf $O/prof/*/*.db 2>/dev/null; ls -la $O/prof/* | head; du -sh $O
tail -4 $O/pytest.log; cat $O/rc.txt; cat $O/post_bench.log; cat $O/bench.json | cut -c1-400
