OUT=gpurun_out/bf16_base; mkdir -p $OUT
python bench.py --mma bf16 --no-secondary --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bf16.json 2> $OUT/bf16.err
python bench.py --mma bf16x6 --no-secondary --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bf16x6.json 2> $OUT/bf16x6.err
python - <<PY
import json
for n in ("bf16","bf16x6"):
    j=json.loads([l for l in open("$OUT/%s.json"%n) if l.startswith("{")][-1]); k=j["train_step"]["kernels"]
    print(n, "ms/step %.2f"%j["ms_per_step"], " ".join("%s %.2f"%(a,v["ms_per_step"]) for a,v in k.items()), "other %.2f"%j["train_step"]["other_ms_per_step"], "mem %.1f GB"%j["train_step"]["peak_device_memory_gb"], j["train_step"]["launch_kinds"])
PY
