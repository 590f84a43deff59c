#!/bin/bash
# round-5 visit D: the build without packed fp32 -- GPU test-suite (incl. the extended neighbour-stream tests), bench A/B against the
# round-4 flags, contraction-kernel A/B, the remaining instruction forms of the probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 tools/probe_pk_swap editanything_amd/csrc/libeditanything_hip.so 200 > gpurun_out/r05d_probe_pk_swap_forms.jsonl 2> gpurun_out/r05d_probe.err; echo "pk rc=$?"
tail -6 gpurun_out/r05d_probe_pk_swap_forms.jsonl | cut -c1-330
timeout 200 tools/probe_pk_swap editanything_amd/csrc/libeditanything_hip.so 200 1 > gpurun_out/r05d_probe_pk_swap_neighbours.jsonl 2>> gpurun_out/r05d_probe.err; echo "pk1 rc=$?"
tail -2 gpurun_out/r05d_probe_pk_swap_neighbours.jsonl | cut -c60-330
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05d_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r05d_pytest.log
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r05d_bench_no_packed.json 2> gpurun_out/r05d_bench_no_packed.err; echo "bench rc=$?"
timeout 400 python tools/bench_with_lib.py gpurun_exp/libea_packed_f32_allowed.so --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r05d_bench_packed.json 2> gpurun_out/r05d_bench_packed.err; echo "bench2 rc=$?"
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r05d_bench_no_packed_2.json 2>> gpurun_out/r05d_bench_no_packed.err; echo "bench3 rc=$?"
python - <<'PY'
import json
for f in ["r05d_bench_no_packed.json", "r05d_bench_packed.json", "r05d_bench_no_packed_2.json"]:
    try:
        d = json.loads(open("gpurun_out/" + f).read().strip().split("\n")[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("phase_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 300 tools/gemm_bench editanything_amd/csrc/libeditanything_hip.so,gpurun_exp/libea_packed_f32_allowed.so --cases all --geglu 32 --iters 10 --rounds 3 --out gpurun_out/r05d_gemm_bench_ab.jsonl > /dev/null 2> gpurun_out/r05d_gemm_bench.err; echo "gemm_bench rc=$?"
