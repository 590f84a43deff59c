# same-box A/B/A/B/A/B of the whole step between two full libraries:  bash tools/visit_lib_ab.sh <libA.so> <libB.so> <tag> [runs]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
A=$1; B=$2; TAG=$3; N=${4:-3}
for i in $(seq 1 $N); do for m in A B; do
  L=$A; [ $m = B ] && L=$B
  python tools/bench_with_lib.py $L --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_${m}_$i.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_${m}_$i.json').read().strip().splitlines()[-1])
print(json.dumps({"lib":"$L","arm":"$m","run":$i,"value":d["value"],"sequential":d["sequential"]["value"],"merged":{k:v["value"] for k,v in d["merged"].items() if isinstance(v,dict)},"roofline_frac":d["roofline"]["frac"],"unet_only_frac":d["roofline"]["unet_only_frac"],"contraction_ms":d["roofline"]["contraction_ms_per_step"]}))
PY
done; done | tee gpurun_out/${TAG}.jsonl
