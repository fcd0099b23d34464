# r5 GPU call 10: the head without packed math -- timed configuration x4, the whole GPU suite, default bench line
cd $GRAFT_REPO_ROOT
f() { grep -v amdgpu.ids | grep "^\[" | cut -c1-220; }
for i in 1 2 3 4; do python tools/diag_timed_config.py --tag run_$i 2>&1 | f; done
python tools/diag_timed_config.py --no-graph --tag three_streams_eager 2>&1 | f
python tools/diag_head_under_load.py --load f16x3 --iters 60 --tag "shipped head, marching load" 2>&1 | f
python -m pytest tests -m gpu -q 2>&1 | grep -v GridwiseOp | tail -40 > gpurun_out/r5_t10.log; tail -25 gpurun_out/r5_t10.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench10.json 2> gpurun_out/r5_bench10.err; head -c 1800 gpurun_out/r5_bench10.json
