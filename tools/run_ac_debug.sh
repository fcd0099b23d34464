for w in stereobase igev lightstereo; do
  for rep in 1 2; do
    timeout 300 python -X faulthandler -m pytest tests/test_gpu_autocast.py -x -q -s --tb=short -k "training_step and $w" 2>&1 | grep -v "MIOpen(HIP)" | tail -40 > gpurun_out/ac_train_${w}_$rep.log
  done
done
timeout 600 python -m pytest tests/test_gpu_autocast.py -q --tb=short -k "not training_step" 2>&1 | grep -v "MIOpen(HIP)" | tail -80 > gpurun_out/ac_rest.log
