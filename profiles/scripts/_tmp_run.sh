python -m pytest tests/test_eig.py tests/test_pipeline.py -m gpu -q -x 2>&1 | tail -3
for c in 3 2 1; do
  TRX_QR_CHAINS=$c python bench.py --steps 3 --no-cpu-baseline > gpurun_out/r2_chain${c}_b128.json 2> gpurun_out/r2_chain${c}_b128.err
  TRX_QR_CHAINS=$c python bench.py --steps 3 --batch 16 --no-cpu-baseline > gpurun_out/r2_chain${c}_b16.json 2> gpurun_out/r2_chain${c}_b16.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2_chain*_b*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, round(d['value'],2), round(d['ms_per_step'],1))
        for k in d['roofline']['kernels'][:5]: print('    ',k['kernel'],k['launches'],round(k['avg_us'],1),round(k['est_total_ms_per_step'],1))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-500:])
PY
