run() { echo -n "$* : "; env "$@" TRX_BENCH_NOPROF=1 timeout 90 python bench.py --steps 3 --warmup 1 --batch $B --points $B --no-cpu-baseline --no-strong-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],1))"; }
B=16
run X=0
run TRX_QR_GROUPS=1
run TRX_QR_GROUPS=4
run TRX_QR_GROUPS=8
run TRX_QR_GROUPS=4 TRX_QR_AED=64
run TRX_QR_GROUPS=4 TRX_QR_AED=32
run TRX_QR_GROUPS=2 TRX_QR_AED=64
run TRX_QR_GROUPS=2 TRX_QR_AED=32
run TRX_QR_GROUPS=4 TRX_SLAB_SPW=1
run TRX_QR_GROUPS=4 TRX_QR_CHAINS=2
B=32
run X=0
run TRX_QR_GROUPS=4
run TRX_QR_GROUPS=8
run TRX_QR_GROUPS=4 TRX_QR_AED=64
B=8
run X=0
run TRX_QR_GROUPS=4
run TRX_QR_GROUPS=8
