# A/B of the XCD-partitioned (grouped) tile order: gpurun_exp/libea_exp32.so (row-major, -DEA_EXP=32) vs libea_exp0.so, same call;
# then the fabric-side traffic of the affected classes on the product library.   gpurun -- 'bash tools/visit_tile_order_ab.sh <tag>'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r04n}
rm -f gpurun_out/${TAG}_tile_order_ab.jsonl
for c in "conv3 B8 H16" "conv3 B8 H8" "conv1 B8 H8" "conv1 B8 H16" "gemm M512" "gemm M2048" "conv3 B8 H32 c1920" "conv3 B8 H32 c1280"; do
  tools/gemm_bench gpurun_exp/libea_exp32.so,gpurun_exp/libea_exp0.so --cases "$c" --iters 20 --rounds 5 --check --out gpurun_out/${TAG}_tmp.jsonl >/dev/null 2>>gpurun_out/${TAG}.err
  cat gpurun_out/${TAG}_tmp.jsonl >> gpurun_out/${TAG}_tile_order_ab.jsonl
done
rm -f gpurun_out/${TAG}_tmp.jsonl
python3 - <<PY
import json
rows=[json.loads(l) for l in open("gpurun_out/${TAG}_tile_order_ab.jsonl")]
by={}
for r in rows: by.setdefault(r["case"],{})[r["lib"].split("/")[-1]]=r
for c,d in by.items():
    a,b=d.get("libea_exp32.so"),d.get("libea_exp0.so")
    if a and b: print("%-45s row-major %7.1f us  grouped %7.1f us  %+5.1f%%  diff-vs-generic %s" % (c, a["us"], b["us"], 100*(b["us"]/a["us"]-1), b.get("max_abs_diff_vs_generic")))
PY
EA_VISIT_LIB=editanything_amd/csrc/libeditanything_hip.so bash tools/gpu_visit.sh $TAG pmc:tools/pmc_cases_colmajor.txt | tail -2
