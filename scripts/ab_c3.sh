for v in 0 1; do echo "DEMCZ_MLB_L32=$v"; DEMCZ_MLB_L32=$v python scripts/bench_configs.py 2000 nocpu 2>&1 | grep "C3"; done
DEMCZ_MLB_L32=1 timeout -k 10 600 python -m pytest tests/test_gpu_long_oracle.py tests/test_gpu_parity.py -x -q -m gpu -k "c3 or blocks" 2>&1 | tail -3
