cd $GRAFT_REPO_ROOT
for w in 12 16 24 32; do
timeout -k 10 300 python bench.py --dist --amp bf16 --graph --no-cpu-baseline --one-loop --workers $w --steps 60 --warmup 10 > gpurun_out/w$w.json 2> gpurun_out/w$w.err
python - $w <<'PY'
import json,sys
d=json.loads([l for l in open("gpurun_out/w%s.json"%sys.argv[1]) if l.startswith("{")][-1]); t=d["trainer_loop"]
print("workers %s: resident %.1f  loop %.1f (%.3f)  loader %.0f samples/s" % (sys.argv[1], d["value"], t["value"], t["vs_resident"], d.get("image_prep",{}).get("loader_samples_per_s_gpu_prep",0)))
PY
done
