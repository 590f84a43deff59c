cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2 3; do for m in 0 256; do
  python tools/bench_with_lib.py gpurun_exp/libea_full_exp$m.so --steps 10 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/ks_ab_${m}_$i.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/ks_ab_${m}_$i.json').read().strip().splitlines()[-1])
print(json.dumps({"lib":"exp$m","run":$i,"value":d["value"],"sequential":d["sequential"]["value"],"merged":{k:v["value"] for k,v in d["merged"].items() if isinstance(v,dict)},"roofline_frac":d["roofline"]["frac"],"unet_only_frac":d["roofline"]["unet_only_frac"],"contraction_ms":d["roofline"]["contraction_ms_per_step"]}))
PY
done; done | tee gpurun_out/ks_ab.jsonl
