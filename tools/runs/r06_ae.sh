mkdir -p gpurun_out/r06_ae
for i in 1 2 3; do
  s=$(date +%s)
  timeout 900 python bench.py > gpurun_out/r06_ae/line_$i.json 2> gpurun_out/r06_ae/err_$i.log
  e=$(date +%s)
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_ae/line_$i.json").read().strip().splitlines()[-1])
oc=d["other_configs"]
print("run $i", $e-$s, "s |", d["ms_per_step"], d["value"], "fixed", d["fixed_batch"]["ms_per_step"], "| c4", oc["c4"]["ms_per_step"], oc["c4"]["median_block_ms_per_step"], [p["ms_per_step"] for p in oc["c4"]["per_seed"]], oc["c4"]["fixed_batch"]["ms_per_step"], "| c5", oc["c5"]["ms_per_step"], oc["c5"]["fixed_batch"]["ms_per_step"])
PY
done
