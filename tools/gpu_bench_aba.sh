#!/bin/bash
# bench A/B: product library vs a side build copied over it ON THE BOX (scratch snapshot).  Usage: gpu_visit3.sh tag side.so
TAG=${1:-r01z}; SIDE=$2
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_a.log 2>&1
cp $P /tmp/prod.so; cp $SIDE $P
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_b.log 2>&1
cp /tmp/prod.so $P
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_a2.log 2>&1
for f in a b a2; do tail -1 gpurun_out/${TAG}_bench_$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['config']['phase_ms'], d['roofline']['achieved'])"; done
