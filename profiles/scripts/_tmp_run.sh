python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for b in 128 16; do
python bench.py --steps 3 --batch $b --no-cpu-baseline 2> gpurun_out/r2_diet_$b.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', round(d['value'],2), round(d['ms_per_step'],1), d['hbm'])"
done
timeout 600 python tests/gpu_config3.py 21 32 32 2>&1 | tail -2
