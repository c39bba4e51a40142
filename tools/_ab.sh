run() { python bench.py --steps 100 --warmup 20 --no-trainer-loop --no-cpu-baseline "$@" > gpurun_out/t1.json 2> gpurun_out/t1.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/t1.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; }
