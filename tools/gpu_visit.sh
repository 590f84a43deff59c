#!/bin/bash
# ONE parameterised GPU-box visit script (replaces the per-visit gpu_r0*.sh files of rounds 1-2):
#     gpurun --timeout 900 -- 'bash tools/gpu_visit.sh <tag> <step> [<step> ...]'
# Steps (each writes under gpurun_out/<tag>_*; nothing here is imported by the product):
#   gemm:<args>      tools/gemm_bench on the product library with <args> ("," -> " "), e.g.
#                    gemm:--cases,conv3,--variants,auto,21,22,--check  (variants themselves use "+": auto+21+22)
#   pytest[:<k>]     python -m pytest tests -m gpu -x -q [-k <k>]
#   smoke            __graft_entry__.smoke()
#   bench[:<args>]   python bench.py <args>  -> <tag>_bench_line.json
#   prof             rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras` -> <tag>_bench_kernel_stats.csv
#   pmc:<file>       HBM-side traffic per launch class (separate --pmc FETCH_SIZE / WRITE_SIZE passes over gemm_bench);
#                    <file> lists one "variant|case name" per line
#   py:<script>[:<args>]   python <script> <args>  (tools/*.py helpers), output -> <tag>_<script>.log
#   profpy:<script>[:<args>]   the same under rocprofv3 --kernel-trace --stats -> <tag>_<script>_kernel_stats.csv
#   sh:<cmd>         anything else ("," -> " ")
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=$1; shift
LIB=${EA_VISIT_LIB:-editanything_amd/csrc/libeditanything_hip.so}
n=0
for step in "$@"; do
  n=$((n+1))
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  t0=$(date +%s)
  case $kind in
    gemm)
      a=$(echo "$arg" | tr ',' ' ' | tr '+' ',')
      timeout 600 tools/gemm_bench $LIB $a --out gpurun_out/${TAG}_gemm${n}.jsonl > /dev/null 2> gpurun_out/${TAG}_gemm${n}.err
      echo "gemm${n} rc=$? lines=$(wc -l < gpurun_out/${TAG}_gemm${n}.jsonl 2>/dev/null)";;
    pytest)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -x -q -rP -m gpu -k "$arg" > gpurun_out/${TAG}_pytest.log 2>&1
      else timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; fi
      echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -3 gpurun_out/${TAG}_pytest.log;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2;;
    bench)
      a=$(echo "$arg" | tr ',' ' ')
      timeout 900 python bench.py $a > gpurun_out/${TAG}_bench_line${n}.json 2> gpurun_out/${TAG}_bench${n}.err
      echo "bench rc=$?"; tail -c 1800 gpurun_out/${TAG}_bench_line${n}.json;;
    prof)
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- \
         python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.err)
      echo "prof rc=$?"
      F=$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp "$F" gpurun_out/${TAG}_bench_kernel_stats.csv
      rm -rf gpurun_out/prof_$TAG; head -12 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-170;;
    pmc)
      D=gpurun_out/pmc_$TAG; rm -rf $D; mkdir -p $D; i=0
      while IFS='|' read -r var c; do
        [ -z "$c" ] && continue; i=$((i+1))
        for pass in fetch write; do
          if [ $pass = fetch ]; then CNT="FETCH_SIZE"; else CNT="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; fi
          (cd /tmp && timeout 60 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$D/c${i}_$pass -o p --pmc $CNT -- \
            $GRAFT_REPO_ROOT/tools/gemm_bench $GRAFT_REPO_ROOT/$LIB --cases "=$c" --variants $var --geglu 32 --iters 3 --rounds 1 > /dev/null 2>> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc.err)
        done
        echo "$i|$var|$c" >> $D/cases.txt
      done < "$arg"
      python3 tools/pmc_collect.py $D gpurun_out/${TAG}_pmc_traffic.json; rm -rf $D;;
    py)
      script=${arg%%:*}; a=""; [ "$script" != "$arg" ] && a=$(echo "${arg#*:}" | tr ',' ' ')
      timeout 900 python $script $a > gpurun_out/${TAG}_$(basename $script .py).log 2>&1; echo "py rc=$?"; tail -5 gpurun_out/${TAG}_$(basename $script .py).log;;
    profpy)
      # rocprofv3 kernel statistics of a python helper: profpy:<script>[:<args>] -> <tag>_<script>_kernel_stats.csv
      script=${arg%%:*}; a=""; [ "$script" != "$arg" ] && a=$(echo "${arg#*:}" | tr ',' ' '); b=$(basename $script .py)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$b -o p -- \
         python $GRAFT_REPO_ROOT/$script $a > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_${b}_prof.log 2>&1)
      echo "profpy rc=$?"
      F=$(find gpurun_out/prof_${TAG}_$b -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && cp "$F" gpurun_out/${TAG}_${b}_kernel_stats.csv
      rm -rf gpurun_out/prof_${TAG}_$b; head -25 gpurun_out/${TAG}_${b}_kernel_stats.csv | cut -c1-150;;
    sh)
      a=$(echo "$arg" | tr ',' ' '); timeout 900 bash -c "$a" 2>&1 | tail -20;;
    *) echo "unknown step $step";;
  esac
  echo "[$step] $(( $(date +%s) - t0 )) s"
done
