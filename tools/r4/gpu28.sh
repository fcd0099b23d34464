#!/bin/bash
# round 4, GPU call 28: 3-D transposed convs as two half launches (4 parity classes each, 3 waves per SIMD): parity, per-layer A/B, whole-model A/B.
# RECORD ONLY: the experiment lost 0.9 % on the whole model and its code (OSA_DECONV_HALVES / osa_deconv3d_halves) was reverted -- profiles/round4/deconv_half_launches_ab.txt.
cd "$(dirname "$0")/../.."
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "deconv or redir or gwc_hourglass or gwcnet_small or gwcnet_full_size or stereobase_hourglass or igev_hourglass or split_activation" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_autograd.py -q -k "conv_transpose or gwcnet or hourglass" 2>&1 | tail -3
export OSA_PRECISION=f16x3
for B in 9 3; do
echo "== deconv layers B=$B (split chain)"
timeout 600 python tools/bench_layers.py --set 3d --batch $B --iters 10 --split --only "deconv" --envs "OSA_DECONV_HALVES=0;OSA_DECONV_HALVES=1;OSA_DECONV_HALVES=0;OSA_DECONV_HALVES=1" 2>&1 | grep -v "amdgpu.ids\|^sum"
done
unset OSA_PRECISION
echo "== whole model A/B (timed only)"
bash tools/bench_ab.sh "OSA_DECONV_HALVES=0" "OSA_DECONV_HALVES=1" 2>&1 | grep -v amdgpu.ids
