python -m pytest tests/test_eig.py -m gpu -q -x 2>&1 | tail -1
run() { echo "== $*"; env "$@" python bench.py --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   ', round(d['value'],2), round(d['ms_per_step'],1))
for k in d['roofline']['kernels'][:2]: print('    ',k['kernel'],k['launches'],round(k['avg_us'],1),round(k['est_total_ms_per_step'],1), round(k['frac'],3))"; }
run TRX_QR_AED=64 TRX_SLAB_PIPE=1
run TRX_QR_AED=64
