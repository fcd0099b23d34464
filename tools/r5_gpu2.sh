# r5 GPU call 2: whole GPU suite (incl. the new SYNC_BN / timed-config / AMP-training tests), training workloads with and without --amp,
# steady-state kernel table of the AMP whole-model training step
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r5_t2.log
for W in stereobase_train stereobase_e2e_train; do
  python bench.py --workload $W --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r5_${W}.json 2> gpurun_out/r5_${W}.err
  python bench.py --workload $W --steps 6 --warmup 3 --no-cpu-baseline --amp > gpurun_out/r5_${W}_amp.json 2> gpurun_out/r5_${W}_amp.err
done
bash tools/prof_train.sh stereobase_e2e_train r5_e2e_amp 400 3 --amp > /dev/null 2>&1
tail -45 gpurun_out/r5_t2.log
for f in gpurun_out/r5_stereobase*.json; do echo $f; head -c 500 $f; echo; done
