cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  python bench.py --workload stereobase_train --force-ddp --steps 3 --warmup 1 --no-cpu-baseline > /tmp/ddp_$i.out 2> /tmp/ddp_$i.err; echo "run $i rc=$?"; grep -o '"launch": "[^"]*"' /tmp/ddp_$i.out | head -1
  grep -v "GridwiseOp\|amdgpu.ids" /tmp/ddp_$i.err | grep -i "error\|capture\|Traceback\|raise\|what()" | head -8 | cut -c1-300
done
