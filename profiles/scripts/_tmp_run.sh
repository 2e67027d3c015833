python -m pytest tests/test_eig.py -m gpu -q -x 2>&1 | tail -2
for b in 128 16; do
python bench.py --steps 3 --batch $b --no-cpu-baseline 2> gpurun_out/r2_dyn_$b.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', round(d['value'],2), round(d['ms_per_step'],1))
for k in d['roofline']['kernels'][:6]: print('    ',k['kernel'],k['launches'],round(k['avg_us'],1),round(k['est_total_ms_per_step'],1), round(k['frac'],3))"
done
TRX_QR_GROUPS=2 python bench.py --steps 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('groups2', round(d['value'],2), round(d['ms_per_step'],1))"
TRX_QR_GROUPS=3 python bench.py --steps 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('groups3', round(d['value'],2), round(d['ms_per_step'],1))"
