cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for mode in pipe seq; do
  extra=""; [ $mode = seq ] && extra="--no-pipeline"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$mode -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --quick $extra > $GRAFT_REPO_ROOT/gpurun_out/r04c_$mode.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04c_$mode.err)
  F=$(find /tmp/tr_$mode -name '*kernel_trace.csv' | head -1)
  python tools/overlap_report.py $F 2 >> gpurun_out/r04c_overlap.jsonl
done
cat gpurun_out/r04c_overlap.jsonl; cut -c1-200 gpurun_out/r04c_pipe.json gpurun_out/r04c_seq.json
