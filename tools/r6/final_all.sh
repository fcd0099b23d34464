# everything the round's final evidence consists of, at ONE code state: GPU suite, training-step kernel table, final pass (tools/profile_round6.sh); collect with tools/collect_round6.py
cd $GRAFT_REPO_ROOT
bash tools/r6/gpu_suite.sh
cd $GRAFT_REPO_ROOT
bash tools/r6/final_train_table.sh
