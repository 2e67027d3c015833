python -m pytest tests/test_blocks.py tests/test_eig.py -m gpu -q -x 2>&1 | tail -2
python tests/gpu_gemm_bench.py 2>&1 | grep -v amdgpu.ids
python bench.py --steps 3 --no-cpu-baseline 2> gpurun_out/r2_bk32.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', round(d['value'],2), round(d['ms_per_step'],1))
for k in d['roofline']['kernels'][:8]: print('    ',k['kernel'],k['launches'],round(k['avg_us'],1),round(k['est_total_ms_per_step'],1), round(k['frac'],3))"
