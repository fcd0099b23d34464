cd $GRAFT_REPO_ROOT
timeout 600 python tools/probe_amp_dtypes.py 2>&1 | grep "ConvGRU\|encoder\|update:" | cut -c1-600
