cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 900 python tools/prof_train_ops.py --amp --top 10 --stacks "aten::copy_,aten::add_,aten::add,aten::fill_,aten::cat,aten::mul,aten::sum,aten::_to_copy,aten::zero_,aten::clone,aten::sub,aten::div,aten::clamp" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/train_ops_amp_stacks.txt
timeout 900 python -m pytest tests/test_gpu_channel_sums.py tests/test_gpu_autograd.py -q -x 2>&1 | tail -3
